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
            print("  env", b, "true rms residual of the last pressure solve", np.sqrt((res ** 2).mean()), "mean-free", np.sqrt(((res - res.mean()) ** 2).mean()), "rms rhs", np.sqrt((div[b] ** 2).mean()), flush=True)
        dom.close()
