"""restarted Chebyshev(S) on the Jacobi splitting, fp32, per pass: passes to the criterion"""
import sys
sys.path.insert(0, "/root/repo")
import torch, fluidgym_amd
from fluidgym_amd import _lib as L

def run(env_id, B, S_list, forcing=2.0, steps=2):
    env = fluidgym_amd.make(env_id, num_envs=B)
    env.reset(seed=5); env.seed(5)
    blk0 = env._domain.getBlock(0)
    blk0.setVelocitySource(torch.zeros_like(blk0.velocity))
    g = torch.Generator(device="cuda").manual_seed(4321)
    for _ in range(steps):
        blk0.velocitySource.normal_(0.0, forcing, generator=g)
        env.step(env.sample_action())
    ns = env._domain.solver
    dt = float(env._dt)
    ns.copy_velocity_result_from_blocks()
    ns.setup_advection(dt)
    ny, nx = ns.ny, ns.nx
    A = ns.buffer(L.FG_BUF_A, (B, 1, ny, nx)); C = ns.buffer(L.FG_BUF_C_OFF, (B, 4, ny, nx)); b = ns.buffer(L.FG_BUF_ADV_RHS, (B, 2, ny, nx))
    rA = 1.0 / A; Cs = C * rA; bs = b * rA
    def G(x):
        return bs - (Cs[:, 0:1] * torch.roll(x, 1, 3) + Cs[:, 1:2] * torch.roll(x, -1, 3) + Cs[:, 2:3] * torch.roll(x, 1, 2) + Cs[:, 3:4] * torch.roll(x, -1, 2))
    def est(xn, x):
        return (A * (xn - x)).double().pow(2).mean(dim=(2, 3)).sqrt().amax().item()
    x = torch.zeros_like(b); e = []
    for k in range(4):
        xn = G(x); e.append(est(xn, x)); x = xn
    rho0 = e[3] / e[2]
    bound = (Cs.abs().sum(1)).amax().item()
    print(f"{env_id}: rho from plain sweeps {rho0:.3f}, row-sum bound {bound:.3f}")
    for S in S_list:
        for rho in (rho0, 0.8 * rho0, 1.25 * rho0, bound):
            x = torch.zeros_like(b); out = []
            for p in range(12):
                xp = x.clone(); om = 1.0
                for k in range(S):
                    g_ = G(x)
                    if k == S - 1: r = est(g_, x)
                    if k == 0: xn = g_; om = 1.0
                    else:
                        om = 1.0 / (1.0 - 0.5 * rho * rho) if k == 1 else 1.0 / (1.0 - 0.25 * rho * rho * om)
                        xn = om * (g_ - xp) + xp
                    xp, x = x, xn
                out.append(r)
                if r < 1e-5: break
            print(f"   S={S} rho_use={rho:.3f}: passes {len(out)}  residual estimates per pass " + " ".join(f"{v:.1e}" for v in out))
    env.close()

run("ChannelJet2D-v0", 8, (6, 8))
run("ChannelJet2D-large-v0", 4, (4, 6))
