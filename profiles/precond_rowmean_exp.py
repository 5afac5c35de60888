"""CPU experiment: PCG iterations on the stirred channel's pressure system with M = L(A = 1) against M = P built from the
ROW-AVERAGED coefficient a(y) = mean_x(1/A) (still separable: x transform + per-mode tridiagonal in y with per-env factors)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse.linalg as spla
from oracle import piso_oracle as O
from fluidgym_amd.envs.channel import CHANNEL_JET_2D_DEFAULT_CONFIG as CFG, inflow_profile

dt = np.float64
nx, ny, L, H = CFG["resolution_x"], CFG["resolution_y"], 22.0, 4.1
ex, ey = np.linspace(0, L, nx + 1), np.linspace(-H / 2, H / 2, ny + 1)
g = O.Grid(O.rectilinear_coords([ex, ey], dtype=dt))
prof = inflow_profile(H, ny).astype(dt)
u = np.zeros((2, ny, nx), dt); u[0] = prof[:, None]
rng = np.random.default_rng(0)
u += 0.05 * rng.standard_normal(u.shape)
inflow = np.zeros((2, ny, 1), dt); inflow[0, :, 0] = prof
bc = {0: O.FixedBC(inflow.copy()), 1: O.FixedBC(inflow.copy()), 2: O.FixedBC(np.zeros(2, dt)), 3: O.FixedBC(np.zeros(2, dt))}
dom = O.Domain(g, 1.0 / CFG["reynolds_number"], u, np.zeros((ny, nx), dt), bc)
opts = O.SolverOptions(direct=True, non_orthogonal=True)
O.make_divergence_free(dom, O.SolverOptions(direct=True))
velm = np.array([1.0, 0.0])

def pcg(P, b, Minv, tol, maxit=20):
    x = np.zeros_like(b); r = b.copy(); hist = [np.sqrt((r * r).mean())]
    z = Minv(r); p = z.copy(); rz = r @ z
    for it in range(maxit):
        if hist[-1] < tol: break
        Ap = P @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap
        hist.append(np.sqrt((r * r).mean()))
        z = Minv(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return len(hist) - 1, hist

def solver_for(A):
    P, _, _ = O.build_pressure_matrix(dom, A)
    Pr = (P - 1e-9 * abs(P.diagonal()).max() * __import__("scipy.sparse", fromlist=["eye"]).eye(P.shape[0])).tocsc()
    lu = spla.splu(Pr)
    def M(r):
        z = lu.solve(r - r.mean()); return z - z.mean()
    return M

amp = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
for step in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    if step % 25 == 0: dom.velocity_source = amp * rng.standard_normal((2, ny, nx))
    O.update_advective_boundaries(dom, [1], velm, CFG["dt"], tol=1e-5)
    out = O.piso_split_step(dom, CFG["dt"], opts)
    A = out["A"]; P = out["P"]
    M0 = solver_for(np.ones_like(A))
    Arow = 1.0 / (1.0 / A).mean(axis=1, keepdims=True) * np.ones_like(A)
    M1 = solver_for(Arow)
    for c in (0, 1):
        b = out[f"div{c}"].ravel().copy()
        n0, h0 = pcg(P, b, M0, 1e-5); n1, h1 = pcg(P, b, M1, 1e-5)
        print(f"step {step} corr {c}: rhs {h0[0]:.2e} | M=L: {n0} its {['%.1e' % v for v in h0[1:3]]} | M=row-mean: {n1} its {['%.1e' % v for v in h1[1:3]]}  (A spread row {np.ptp(Arow):.1f} total {np.ptp(A):.1f})")
