"""fp32 point Jacobi on the channel velocity systems: the estimate diag (x_new - x) against the true residual; Chebyshev variant"""
import sys, math
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
    A = ns.buffer(L.FG_BUF_A, (B, 1, ny, nx))
    C = ns.buffer(L.FG_BUF_C_OFF, (B, 4, ny, nx))
    b = ns.buffer(L.FG_BUF_ADV_RHS, (B, 2, ny, nx))
    Ad, Cd, bd = A.double(), C.double(), b.double()
    def nb(Cm, x):
        return (Cm[:, 0:1] * torch.roll(x, 1, 3) + Cm[:, 1:2] * torch.roll(x, -1, 3) + Cm[:, 2:3] * torch.roll(x, 1, 2) + Cm[:, 3:4] * torch.roll(x, -1, 2))
    def true_res(x):
        xd = x.double()
        r = bd - (Ad * xd + nb(Cd, xd))
        return r.pow(2).mean(dim=(2, 3)).sqrt().amax().item()
    rA = 1.0 / A; Cs = C * rA; bs = b * rA
    print(env_id, "diag mean", A.mean().item(), "tol 1e-5")
    x = torch.zeros_like(b)
    hit = None
    for k in range(1, 49):
        xn = bs - nb(Cs, x)
        est = (A * (xn - x)).double().pow(2).mean(dim=(2, 3)).sqrt().amax().item()
        x = xn
        if k % 4 == 0 or (hit is None and est < 1e-5):
            print(f"  sweep {k:2d}: estimate (residual of x_{k-1}) {est:.3e}   true residual of x_{k} {true_res(x):.3e}")
        if hit is None and est < 1e-5: hit = k
    print("  plain Jacobi: estimate below tol at sweep", hit)
    # Chebyshev semi-iteration on the Jacobi splitting, rho from the observed contraction of sweeps 3 -> 4
    x = torch.zeros_like(b); ests = []
    for k in range(1, 5):
        xn = bs - nb(Cs, x); ests.append((A * (xn - x)).double().pow(2).mean().sqrt().item()); x = xn
    rho = ests[3] / ests[2]
    for rho_use in (rho, 0.8 * rho, 1.2 * rho):
        x_prev = torch.zeros_like(b); x = bs.clone(); om = 1.0; hitc = None
        for k in range(2, 41):
            om = 1.0 / (1.0 - 0.25 * rho_use * rho_use * om) if k > 2 else 1.0 / (1.0 - 0.5 * rho_use * rho_use)
            g_ = bs - nb(Cs, x)
            est = (A * (g_ - x)).double().pow(2).mean(dim=(2, 3)).sqrt().amax().item()
            xn = om * (g_ - x_prev) + x_prev
            x_prev, x = x, xn
            if est < 1e-5 and hitc is None:
                hitc = k; print(f"  Chebyshev rho={rho_use:.3f}: estimate {est:.2e} at sweep {k}; true residual after it {true_res(x):.3e}")
                break
        if hitc is None: print(f"  Chebyshev rho={rho_use:.3f}: not below tol in 40 sweeps (last {est:.2e})")
    env.close()

run("ChannelJet2D-v0", 8)
run("ChannelJet2D-large-v0", 4)
