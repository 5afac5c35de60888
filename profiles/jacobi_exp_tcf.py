"""Point-Jacobi contraction on the velocity systems of the 3-D turbulent channel (TCF3D-baseline-v0) and of RBC (wall-refined grids)"""
import sys
sys.path.insert(0, "/root/repo")
import torch, fluidgym_amd
from fluidgym_amd import _lib as L

def run(env_id, B, dims):
    env = fluidgym_amd.make(env_id, num_envs=B)
    env.reset(seed=5); env.seed(5)
    for _ in range(2): env.step(env.sample_action())
    ns = env._domain.solver
    dt = float(env._dt) * 0.5
    ns.copy_velocity_result_from_blocks()
    ns.setup_advection(dt)
    shp = (ns.nz, ns.ny, ns.nx) if dims == 3 else (ns.ny, ns.nx)
    A = ns.buffer(L.FG_BUF_A, (B, 1) + shp).double(); C = ns.buffer(L.FG_BUF_C_OFF, (B, 2 * dims) + shp).double(); b = ns.buffer(L.FG_BUF_ADV_RHS, (B, dims) + shp).double()
    ax = {0: -1, 1: -2, 2: -3}
    def offx(x):
        out = torch.zeros_like(x)
        for a in range(dims):
            out += C[:, 2 * a:2 * a + 1] * torch.roll(x, 1, ax[a]) + C[:, 2 * a + 1:2 * a + 2] * torch.roll(x, -1, ax[a])
        return out
    ratio = (C.abs().sum(1, keepdim=True) / A).amax().item()
    info = ns.solve_advection(tol=1e-5 if dims == 2 else 1e-6)
    print(f"{env_id}: dt {dt:.2e} max row sum|off|/diag {ratio:.3f}; BiCGStab iterations {sorted({i.used_iterations + 1 for i in info})}")
    x = torch.zeros_like(b)
    red = tuple(range(2, 2 + dims))
    for k in range(1, 121):
        xn = (b - offx(x)) / A
        rms = (A * (xn - x)).pow(2).mean(dim=red).sqrt().amax().item()
        x = xn
        if k in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 120) or rms < 1e-6:
            print(f"   sweep {k:3d}: rms residual {rms:.3e}")
        if rms < 1e-6: break
    env.close()

run("TCF3D-baseline-v0", 2, 3)
