"""The additive multilevel preconditioner of the on-chip pressure CG on the CPU: host tables (simulation/multiblock.py::
multilevel_tables) on the reference's cylinder mesh, and a NumPy replay of exactly what the kernel does with them
(csrc/fg_mb_step.hip::k_mbc_onchip, PRE branch) inside the same projected CG -- iteration counts against the plain recurrence."""
import numpy as np
import scipy.sparse as sp

from fluidgym_amd.simulation.multiblock import multilevel_tables
from tests import helpers_mb as H
from tests.test_mb_tables import HostTables


def _cylinder_spec(res):
    from fluidgym_amd.envs.cylinder_grid import make_vortex_street_mesh

    m = make_vortex_street_mesh(res)
    F = {"-x": 0, "+x": 1, "-y": 2, "+y": 3}
    s = H.Spec(2, 0.01)
    s.blocks = [c.astype(np.float64) for c in m.coords]
    s.fixed = [(b, F[f], v.astype(np.float64)) for (b, f), v in m.fixed.items()]
    s.connections = [(b1, F[f1], b2, F[f2], F[ax]) for b1, f1, b2, f2, ax in m.connections]
    return s


def _pressure_matrix(t, rA):
    N, F = t.N, 2 * t.d
    nbr = t.table(0, np.int32).reshape(F, N)
    KPp = t.table(8, np.float32).reshape(F + 1, F, N).astype(np.float64)
    KPn = t.table(9, np.float32).reshape(F + 1, F, N).astype(np.float64)
    rn = np.where(nbr >= 0, rA[np.maximum(nbr, 0)], 0.0)
    vals = (KPp * rA[None, None, :] + KPn * rn[None, :, :]).sum(1)
    rows, cols, v = [np.arange(N)], [np.arange(N)], [vals[0]]
    for f in range(F):
        ok = nbr[f] >= 0
        rows.append(np.nonzero(ok)[0]); cols.append(nbr[f][ok]); v.append(vals[1 + f][ok])
    return sp.csr_matrix((np.concatenate(v), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N))


def _pcg(P, b, M, tol, maxit=600):
    """the kernel's recurrence: residual with its mean removed, z = M r (mean removed), RMS criterion"""
    N = len(b)
    x = np.zeros(N); r = b - b.mean()
    for it in range(maxit):
        if np.sqrt(r @ r / N) < tol:
            return it
        z = M(r) if M else r.copy()
        z -= z.mean()
        rz = r @ z
        p = z if it == 0 else z + (rz / rz_prev) * p
        Ap = P @ p; Ap -= Ap.mean()
        alpha = rz / (p @ Ap)
        x += alpha * p; r -= alpha * Ap
        rz_prev = rz
    return maxit


def test_tables_and_iteration_counts_on_the_cylinder_mesh():
    spec = _cylinder_spec(24)                    # the easy envs' mesh: 5 blocks, 14 232 cells
    t = HostTables(spec)
    N = t.N
    sizes, off = [], 0
    for c in spec.blocks:
        nx, ny = c.shape[-1] - 1, c.shape[-2] - 1
        sizes.append((nx, ny, off)); off += nx * ny
    assert off == N
    rng = np.random.default_rng(0)
    A = 100.0 * (1.0 + 0.3 * rng.random(N))
    P1 = _pressure_matrix(t, np.ones(N))
    tab = multilevel_tables(P1, sizes)
    a4, p4 = tab["a4"], tab["parent4"]
    assert a4.min() == 0 and a4.max() == tab["n4"] - 1 and len(np.unique(a4)) == tab["n4"]        # every aggregate is used
    assert p4.min() == 0 and p4.max() == tab["n8"] - 1 and len(p4) == tab["n4"]
    # aggregates never straddle blocks and hold at most 5 x 5 cells; 8-aggregates hold at most 4 sub-tiles
    for nx, ny, o in sizes:
        ids = a4[o: o + nx * ny]
        assert set(ids).isdisjoint(set(np.delete(a4, np.s_[o: o + nx * ny])))
    assert np.bincount(a4).max() <= 25 and np.bincount(p4).max() <= 4
    for a, (first, w, h, stride) in enumerate(tab["rect4"]):                          # the rectangles the kernel sums by shape
        cells = (first + np.arange(h)[:, None] * stride + np.arange(w)[None, :]).reshape(-1)
        assert (a4[cells] == a).all() and len(cells) == np.count_nonzero(a4 == a)
    # the kernel's preconditioner: D^-1 r + 1/2 s^-1 Z4 D4^-1 Z4^T r + s^-1 Z8 A8^+ Z8^T r, s = trace(P) / trace(S)
    P = _pressure_matrix(t, 1.0 / A)
    D = P.diagonal()
    inv_s = tab["geom_diag_sum"] / D.sum()
    Z4 = sp.csr_matrix((np.ones(N), (np.arange(N), a4)), shape=(N, tab["n4"]))
    Z8 = sp.csr_matrix((np.ones(tab["n4"]), (np.arange(tab["n4"]), p4)), shape=(tab["n4"], tab["n8"]))

    def M(r):
        r4 = Z4.T @ r
        e8 = inv_s * (tab["aci8"] @ (Z8.T @ r4))
        c4 = 0.5 * inv_s * r4 / tab["d4"] + Z8 @ e8
        return r / D + Z4 @ c4

    cc = np.concatenate([0.25 * (c[:, :-1, :-1] + c[:, 1:, :-1] + c[:, :-1, 1:] + c[:, 1:, 1:]).reshape(2, -1) for c in spec.blocks], 1)
    xs = np.sin(0.7 * cc[0]) * np.cos(1.3 * cc[1]) + 0.3 * np.sin(2.1 * cc[0] + cc[1])
    b = P @ xs
    b -= b.mean(); b /= np.sqrt(b @ b / N)
    plain, pre = _pcg(P, b, None, 1e-2), _pcg(P, b, M, 1e-2)
    assert pre * 2.5 <= plain, (plain, pre)        # measured on this mesh: 63 against 193
    assert _pcg(P, b, M, 1e-3) < 600               # and it keeps converging on the non-symmetric matrix
    t.close()


def test_coarse_operator_never_acts_on_the_constant():
    """The coarse constant is the (approximate) null vector of the Galerkin operator; inverted it is a huge amplification of a
    mode the pressure matrix annihilates -- the failure of the right-preconditioned BiCGStab before this was fixed.
    On a matrix assembled in fp32 (row sums only ~1e-7 off) the tables must deflate it: A8^+ 1 = 0, 1^T A8^+ = 0, and P M keeps
    its spectrum on the non-negative real axis with no eigenvalue thrown far from 0 by the constant."""
    from tests import helpers_mb as H2

    for fn in (H2.split_rotated_channel, H2.polar_ring):
        spec = fn()
        t = HostTables(spec)
        N = t.N
        sizes, off = [], 0
        for c in spec.blocks:
            nx, ny = c.shape[-1] - 1, c.shape[-2] - 1
            sizes.append((nx, ny, off)); off += nx * ny
        rng = np.random.default_rng(0)
        P1 = _pressure_matrix(t, np.ones(N)).tolil()
        # what fp32 assembly on the GPU does to the diagonal: relative perturbations of 1e-7, either sign
        P1.setdiag(P1.diagonal() * (1.0 + 2e-7 * rng.standard_normal(N)))
        P1 = P1.tocsr()
        tab = multilevel_tables(P1, sizes, max_n4=65534, max_n8=2048)
        one = np.ones(tab["n8"])
        assert np.abs(tab["aci8"] @ one).max() < 1e-9 * np.abs(tab["aci8"]).max()
        assert np.abs(one @ tab["aci8"]).max() < 1e-9 * np.abs(tab["aci8"]).max()
        a4, p4 = tab["a4"], tab["parent4"]
        P = _pressure_matrix(t, 0.03 * np.exp(0.3 * rng.standard_normal(N))).toarray()
        diag = P.diagonal()
        sinv = tab["geom_diag_sum"] / diag.sum()

        def M(r):
            r4 = np.bincount(a4, weights=r, minlength=tab["n4"]); r8 = np.bincount(p4, weights=r4, minlength=tab["n8"])
            return r / diag + 0.5 * sinv * (r4 / tab["d4"])[a4] + sinv * (tab["aci8"] @ r8)[p4[a4]]

        ev = np.linalg.eigvals(P @ np.stack([M(e) for e in np.eye(N)], 1))
        assert ev.real.min() > -1e-6 and ev.real.max() < 4.0 and np.abs(ev.imag).max() < 1e-6, (fn.__name__, ev.real.min(), ev.real.max())
