import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers_mb as H
from tests.test_gpu_mb import _state, _load
for fn in (H.polar_ring, H.split_rotated_channel):
    for ml in (0, 1):
        spec = fn(); d = spec.oracle(); B = 2
        dom = spec.native(batch=B)
        if ml: print("tables", dom.set_pressure_multilevel())
        states = [_state(d, 10 + b) for b in range(B)]
        _load(dom, states)
        print("====", fn.__name__, "ml", ml, flush=True)
        its = dom.piso_step([0.05, 0.03], advection_tol=1e-7, pressure_tol=2e-6, pressure_use_bicgstab=1, pressure_project_mean=True, raise_on_failure=False)
        print("its", its, "ladder", dom.ladder(), flush=True)
        from fluidgym_amd import _lib as L
        N, F = dom.n_cells, 4
        diag = dom.buffer(L.FG_MB_BUF_P_DIAG).view(B, N).cpu().numpy().astype(np.float64)
        off = dom.buffer(L.FG_MB_BUF_P_OFF).view(B, F, N).cpu().numpy().astype(np.float64)
        div = dom.buffer(L.FG_MB_BUF_DIV).view(B, N).cpu().numpy().astype(np.float64)
        pr = dom.pressure.cpu().numpy().astype(np.float64)
        nbr = dom.neighbors()
        for b in range(B):
            y = diag[b] * pr[b]
            for f in range(F):
                ok = nbr[f] >= 0
                y[ok] += off[b, f][ok] * pr[b][nbr[f][ok]]
            res = div[b] - y
            if ml:
                tab = dom._multilevel_tables
                rr = np.random.default_rng(b).standard_normal(N).astype(np.float32)
                if b == 0:
                    zz = dom.multilevel_apply(torch.from_numpy(np.stack([rr] * B))).cpu().numpy().astype(np.float64)
                a4, p4 = tab["a4"], tab["parent4"]
                sinv = tab["geom_diag_sum"] / diag[b].sum()
                r4 = np.bincount(a4, weights=rr.astype(np.float64) if b == 0 else np.random.default_rng(0).standard_normal(N).astype(np.float32).astype(np.float64), minlength=tab["n4"])
                r0 = np.random.default_rng(0).standard_normal(N).astype(np.float32).astype(np.float64)
                r4 = np.bincount(a4, weights=r0, minlength=tab["n4"]); r8 = np.bincount(p4, weights=r4, minlength=tab["n8"])
                ref = r0 / diag[b] + 0.5 * sinv * (r4 / tab["d4"])[a4] + sinv * (tab["aci8"] @ r8)[p4[a4]]
                print("  env", b, "M apply rel err vs numpy", np.abs(zz[b] - ref).max() / np.abs(ref).max(), "scale_inv", sinv, "diag mean", diag[b].mean(), "geom", tab["geom_diag_sum"], flush=True)
            if ml:
                Pd = np.zeros((N, N))
                Pd[np.arange(N), np.arange(N)] = diag[b]
                for f in range(F):
                    ok = nbr[f] >= 0
                    Pd[np.nonzero(ok)[0], nbr[f][ok]] += off[b, f][ok]
                def Mf(r):
                    r4 = np.bincount(a4, weights=r, minlength=tab["n4"]); r8 = np.bincount(p4, weights=r4, minlength=tab["n8"])
                    return r / diag[b] + 0.5 * sinv * (r4 / tab["d4"])[a4] + sinv * (tab["aci8"] @ r8)[p4[a4]]
                Md = np.stack([Mf(e) for e in np.eye(N)], 1)
                ev = np.linalg.eigvals(Pd @ Md)
                evp = np.linalg.eigvals(Pd / diag[b][None, :])
                print("  env", b, "N", N, "eig(P M): re min/max", ev.real.min(), ev.real.max(), "im max", np.abs(ev.imag).max(), "| eig(P D^-1): re min/max", evp.real.min(), evp.real.max(),
                      "asym", np.abs(Pd - Pd.T).max() / np.abs(Pd).max(), flush=True)
                # the recurrence of the kernels replayed in float64 on this very matrix and right-hand side
                def bicg(Mfun, bvec, tol=2e-6, maxit=400):
                    x = np.zeros(N); r = bvec - bvec.mean(); rw = r.copy(); pp = r.copy(); rho = rw @ r; al = om = 1.0; v = np.zeros(N)
                    for it in range(maxit):
                        if np.sqrt(r @ r / N) < tol: return it, x
                        if it > 0:
                            rn = rw @ r; be = (rn / rho) * (al / om); rho = rn; pp = r + be * (pp - om * v)
                        ph = Mfun(pp) if Mfun else pp
                        v = Pd @ ph; v -= v.mean(); al = rho / (rw @ v); s_ = r - al * v
                        sh = Mfun(s_) if Mfun else s_
                        t_ = Pd @ sh; t_ -= t_.mean(); om = (t_ @ s_) / (t_ @ t_)
                        x = x + al * ph + om * sh; r = s_ - om * t_
                    return maxit, x
                i0, _ = bicg(None, div[b]); i1, x1 = bicg(Mf, div[b])
                rr1 = div[b] - Pd @ x1; rr1 -= rr1.mean()
                print("  env", b, "float64 replay on this system: plain its", i0, "multilevel its", i1, "true rms", np.sqrt((rr1 ** 2).mean()), flush=True)
            print("  env", b, "true rms residual of the last pressure solve", np.sqrt((res ** 2).mean()), "mean-free", np.sqrt(((res - res.mean()) ** 2).mean()), "rms rhs", np.sqrt((div[b] ** 2).mean()), flush=True)
        dom.close()
