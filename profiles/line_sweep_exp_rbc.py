"""Would LINE Jacobi sweeps -- x <- (D + O_y)^-1 (b - O_x x), the y-line solve of csrc/fg_linepre.hip as an iteration of its own -- settle
the velocity systems of RBC2D (512 x 128, 40 : 1 wall refinement) the way point sweeps settle the channel's?  Matrix and right-hand
side from the library, sweeps in torch (fp64), the true residual per sweep; point sweeps beside them.
    python profiles/line_sweep_exp_rbc.py [ENV_ID] [NUM_ENVS] [ENV_STEPS]"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, fluidgym_amd
from fluidgym_amd import _lib as L

env_id = sys.argv[1] if len(sys.argv) > 1 else "RBC2D-baseline-v0"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
env = fluidgym_amd.make(env_id, num_envs=B)
env.reset(seed=5); env.seed(5)
for _ in range(steps):
    env.step(env.sample_action())
ns = env._domain.solver
n_sub = max(int(getattr(env._sim, "substep_count", 1)), 1)
dt = float(env._dt) / n_sub      # (the size of the adaptive substeps the last sim step took)
print("sim step", float(env._dt), "substeps", n_sub)
ns.copy_velocity_result_from_blocks()
ns.setup_advection(dt)
ny, nx = ns.ny, ns.nx
A = ns.buffer(L.FG_BUF_A, (B, 1, ny, nx)).double()
C = ns.buffer(L.FG_BUF_C_OFF, (B, 4, ny, nx)).double()
b = ns.buffer(L.FG_BUF_ADV_RHS, (B, 2, ny, nx)).double()
info = ns.solve_advection(tol=1e-5)
print(env_id, f"{nx} x {ny}, dt {dt:.4g}: the library's solver took", sorted({i.used_iterations for i in info}), "iterations")
print(f"  max row sum|off|/diag = {(C.abs().sum(1, keepdim=True) / A).amax().item():.3f};  x part {(C[:, 0:2].abs().sum(1, keepdim=True) / A).amax().item():.3f};  y part {(C[:, 2:4].abs().sum(1, keepdim=True) / A).amax().item():.3f}")
Cxm, Cxp, Cym, Cyp = C[:, 0:1], C[:, 1:2], C[:, 2:3], C[:, 3:4]

def offx(x): return Cxm * torch.roll(x, 1, 3) + Cxp * torch.roll(x, -1, 3)
def offy(x): return Cym * torch.roll(x, 1, 2) + Cyp * torch.roll(x, -1, 2)
def res(x): return (b - (A * x + offx(x) + offy(x))).pow(2).mean(dim=(2, 3)).sqrt().amax().item()

# Thomas factorisation of T = tridiag(Cym, A, Cyp) along y, per env and column (the first / last row's outer coefficient is zero: walls)
inv = torch.zeros_like(A); cp = torch.zeros_like(A)
for j in range(ny):
    d = A[:, :, j] - (Cym[:, :, j] * cp[:, :, j - 1] if j > 0 else 0.0)
    inv[:, :, j] = 1.0 / d
    cp[:, :, j] = Cyp[:, :, j] * inv[:, :, j]
def tsolve(r):
    y = torch.zeros_like(r)
    for j in range(ny):
        y[:, :, j] = (r[:, :, j] - (Cym[:, :, j] * y[:, :, j - 1] if j > 0 else 0.0)) * inv[:, :, j]
    z = torch.zeros_like(r)
    for j in range(ny - 1, -1, -1):
        z[:, :, j] = y[:, :, j] - (cp[:, :, j] * z[:, :, j + 1] if j < ny - 1 else 0.0)
    return z

for name, sweep in (("line", lambda x: tsolve(b - offx(x))), ("point", lambda x: (b - offx(x) - offy(x)) / A)):
    for start, x in (("zero", torch.zeros_like(b)), ("u^n", ns.velocity.double().reshape(B, 2, ny, nx).clone())):
        out = [res(x)]
        for k in range(12):
            x = sweep(x)
            out.append(res(x))
        print(f"  {name:5s} sweeps from {start:4s}: rms residual " + " ".join(f"{v:.1e}" for v in out))
env.close()
