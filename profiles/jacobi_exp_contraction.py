"""How fast does point Jacobi contract on the velocity systems of the channel envs?  (matrix from the library, sweeps in torch)"""
import sys
sys.path.insert(0, "/root/repo")
import torch, fluidgym_amd
from fluidgym_amd import _lib as L

def run(env_id, B, forcing=2.0):
    env = fluidgym_amd.make(env_id, num_envs=B)
    env.reset(seed=5); env.seed(5)
    blk0 = env._domain.getBlock(0)
    blk0.setVelocitySource(torch.zeros_like(blk0.velocity))
    g = torch.Generator(device="cuda").manual_seed(4321)
    for _ in range(2):
        blk0.velocitySource.normal_(0.0, forcing, generator=g)
        env.step(env.sample_action())
    ns = env._domain.solver
    dt = float(env._dt)
    ns.copy_velocity_result_from_blocks()
    ns.setup_advection(dt)
    ny, nx = ns.ny, ns.nx
    A = ns.buffer(L.FG_BUF_A, (B, 1, ny, nx)).double()
    C = ns.buffer(L.FG_BUF_C_OFF, (B, 4, ny, nx)).double()
    b = ns.buffer(L.FG_BUF_ADV_RHS, (B, 2, ny, nx)).double()
    info = ns.solve_advection(tol=1e-5)
    print(env_id, "BiCGStab iterations", sorted({i.used_iterations for i in info}), "final res max", max(i.final_residual for i in info))
    ratio = (C.abs().sum(1, keepdim=True) / A).amax().item()
    print(f"  max row sum|off|/diag = {ratio:.3f}; dt {dt}; |u|max {blk0.velocity.abs().max().item():.2f}")
    def offx(x):
        return (C[:, 0:1] * torch.roll(x, 1, 3) + C[:, 1:2] * torch.roll(x, -1, 3) + C[:, 2:3] * torch.roll(x, 1, 2) + C[:, 3:4] * torch.roll(x, -1, 2))
    x = torch.zeros_like(b)
    for k in range(1, 41):
        xn = (b - offx(x)) / A
        r = A * (xn - x)      # residual of x
        rms = r.pow(2).mean(dim=(2, 3)).sqrt().amax().item()
        x = xn
        if k <= 3 or k % 2 == 0 or rms < 1e-5:
            print(f"  sweep {k:2d}: rms residual of x_{k-1} = {rms:.3e}")
        if rms < 1e-5: break
    # fp32 sweeps: where do they stall?
    A32, C32, b32 = A.float(), C.float(), b.float()
    rA = 1.0 / A32; Cs = C32 * rA; bs = b32 * rA
    x = torch.zeros_like(b32)
    for k in range(1, 41):
        xn = bs - (Cs[:, 0:1] * torch.roll(x, 1, 3) + Cs[:, 1:2] * torch.roll(x, -1, 3) + Cs[:, 2:3] * torch.roll(x, 1, 2) + Cs[:, 3:4] * torch.roll(x, -1, 2))
        x = xn
    xd = x.double()
    r = b - (A * xd + offx(xd))
    print(f"  fp32 pre-scaled sweeps x40: true residual rms {r.pow(2).mean(dim=(2,3)).sqrt().amax().item():.3e}")
    env.close()

run("ChannelJet2D-v0", 8)
run("ChannelJet2D-large-v0", 4)
