"""The host-built mesh tables of the multi-block path (csrc/fg_mb_topo.hip, C++) against the per-cell oracle, on the CPU.

A handle created with device < 0 builds and serves the tables without a GPU.  Each table is replayed with NumPy exactly
as the kernels of csrc/fg_mb_step.hip use it and compared with what oracle/mb_oracle.py computes cell by cell: neighbour
and boundary-slot numbering, cell / boundary-face transforms, the viscous part of the advection-diffusion matrix, the
pressure matrix as coefficient pairs of 1/A, and the lagged corner operators of velocity and pressure."""
import ctypes

import numpy as np
import pytest

from fluidgym_amd import _lib as L
from oracle import mb_oracle as mbo
from tests import helpers_mb as H


def _cylinder_spec():
    from fluidgym_amd.envs.cylinder_grid import make_vortex_street_mesh

    m = make_vortex_street_mesh(8)
    F = {"-x": 0, "+x": 1, "-y": 2, "+y": 3}
    s = H.Spec(2, 0.01)
    s.blocks = [c.astype(np.float64) for c in m.coords]
    rng = np.random.default_rng(5)
    s.fixed = [(b, F[f], v.astype(np.float64) + 0.1 * rng.standard_normal(v.shape)) for (b, f), v in m.fixed.items()]
    s.connections = [(b1, F[f1], b2, F[f2], F[ax]) for b1, f1, b2, f2, ax in m.connections]
    return s


def _recorded_airfoil_spec():
    """The reference's OWN airfoil mesh at full resolution (45 k cells, NACA 0012 at 20 degrees; vertex coordinates, boundary
    velocities and construction calls recorded off its make_airfoil_domain, tests/golden/make_golden_airfoil.py): the smallest
    cells have 1e-7 of the typical area where front / top / bottom blocks meet."""
    import os

    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_airfoil_grid.npz"))
    F = {"-x": 0, "+x": 1, "-y": 2, "+y": 3}
    calls = [str(c).split() for c in G["aoa20_calls"]]
    s = H.Spec(2, 0.3 / 1e3)
    order = sorted(int(c[1]) for c in calls if c[0] == "block")
    s.blocks = [G[f"aoa20_block{b}"].astype(np.float64) for b in order]
    for c in calls:
        if c[0] == "velocity":
            b, f = int(c[1]), F[c[2]]
            face_cells = s.blocks[b].shape[2 if f >= 2 else 1] - 1
            v = G[f"aoa20_velocity_{b}_{c[2]}"].astype(np.float64)
            v = v.reshape(2, -1) if v.size > 2 else v.reshape(2, 1)
            s.fixed.append((b, f, np.ascontiguousarray(np.broadcast_to(v, (2, face_cells)))))
        elif c[0] == "connect":
            s.connections.append((int(c[1]), F[c[2]], int(c[3]), F[c[4]], F[c[5]]))
    return s


class HostTables:
    def __init__(self, spec, flags=25):
        self.lib = L.load()
        self.h = ctypes.c_void_p()
        L.check(self.lib.fg_mb_create(spec.dims, 1, -1, ctypes.byref(self.h)))
        if flags != 25:
            L.check(self.lib.fg_mb_set_nonortho_flags(self.h, flags))
        for c in spec.blocks:
            c32 = np.ascontiguousarray(c, np.float32)
            size = [c.shape[-1 - a] - 1 for a in range(spec.dims)] + [1] * (3 - spec.dims)
            L.check(self.lib.fg_mb_add_block(self.h, c32.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), *size, None))
        for b, a in spec.periodic:
            L.check(self.lib.fg_mb_make_periodic(self.h, b, a))
        for b1, f1, b2, f2, a1, *rest in spec.connections:
            L.check(self.lib.fg_mb_connect(self.h, b1, f1, b2, f2, a1, rest[0] if rest else 0))
        L.check(self.lib.fg_mb_finalize(self.h))
        n, nb = ctypes.c_int32(), ctypes.c_int32()
        L.check(self.lib.fg_mb_sizes(self.h, ctypes.byref(n), ctypes.byref(nb)))
        self.N, self.NB, self.d = n.value, nb.value, spec.dims

    def table(self, which, dtype):
        cnt = ctypes.c_int64()
        L.check(self.lib.fg_mb_get_host_table(self.h, which, None, ctypes.byref(cnt)))
        out = np.zeros(max(cnt.value, 1), dtype)
        L.check(self.lib.fg_mb_get_host_table(self.h, which, out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(cnt)))
        return out[: cnt.value]

    def close(self):
        self.lib.fg_mb_destroy(self.h)


SPECS = [H.split_rotated_channel, H.skewed_pair, H.twisted_ring, H.skewed_pair_3d, _cylinder_spec, H.cylinder_3d_small, H.airfoil_spec]


def test_tables_of_the_reference_airfoil_mesh_reproduce_the_oracle():
    test_tables_reproduce_the_oracle(_recorded_airfoil_spec, 25)


@pytest.mark.parametrize("flags", [25, 10])
@pytest.mark.parametrize("spec_fn", SPECS)
def test_tables_reproduce_the_oracle(spec_fn, flags):
    """flags: the reference's nonOrthoFlags -- 25 what the simulation runs, 10 every cross-metric term on the right-hand side."""
    spec = spec_fn()
    d = spec.oracle()
    t = HostTables(spec, flags)
    N, NB, dm, F = t.N, t.NB, spec.dims, 2 * spec.dims
    assert N == d.N
    nbr = t.table(0, np.int32).reshape(F, N)
    T = t.table(2, np.float32).reshape(N, dm * dm + 1)
    Tb = t.table(3, np.float32).reshape(-1, dm * dm + 1)[:NB]
    Vdiag, Voff = t.table(6, np.float32), t.table(7, np.float32).reshape(F, N)
    KPp, KPn = t.table(8, np.float32).reshape(F + 1, F, N), t.table(9, np.float32).reshape(F + 1, F, N)
    rng = np.random.default_rng(1)
    nu = d.nu

    # ---- transforms and slot numbering
    slot_of = {}
    k = 0
    for b, blk in enumerate(d.blocks):
        for f, bd in enumerate(blk.bounds):
            if bd.type == mbo.FIXED:
                slot_of[(b, f)] = k
                assert np.allclose(Tb[k: k + len(bd.det), :dm * dm], bd.Minv.reshape(-1, dm * dm), rtol=2e-5, atol=1e-6)
                assert np.allclose(Tb[k: k + len(bd.det), dm * dm], bd.det, rtol=2e-5)
                k += len(bd.det)
    assert k == NB
    for b, pos in d.cells():
        g = d.gidx(b, pos)
        Minv, det = d.Tcell(b, pos)
        assert np.allclose(T[g, :dm * dm], Minv.reshape(-1), rtol=2e-5, atol=1e-5 * np.abs(Minv).max())
        assert np.isclose(T[g, dm * dm], det, rtol=2e-5)

    # ---- viscous matrix part: with u = 0 and resting walls C = det/dt + nu V
    for b, blk in enumerate(d.blocks):
        for bd in blk.bounds:
            if bd.type == mbo.FIXED:
                bd.velocity[:] = 0.0
    dt = 0.07
    diag, off, onbr = d.build_matrix(np.zeros((dm, N)), dt, flags)
    det = T[:, dm * dm].astype(np.float64)
    assert np.array_equal(np.where(onbr >= 0, onbr, -1), np.where(nbr >= 0, nbr, -1))
    assert np.allclose((diag * det - det / dt) / nu, Vdiag, rtol=3e-5, atol=3e-5 * np.abs(Vdiag).max())
    assert np.allclose(off * det / nu, Voff, rtol=3e-5, atol=3e-5 * np.abs(Vdiag).max())

    # ---- pressure matrix from the coefficient pairs
    A = 1.0 + rng.random(N)
    rA = 1.0 / A
    Pd, Po, _ = d.build_pressure_matrix(A, flags)
    rn = np.where(nbr >= 0, rA[np.maximum(nbr, 0)], 0.0)          # [F, N]
    mine = (KPp * rA[None, None, :] + KPn * rn[None, :, :]).sum(axis=1)   # [F + 1, N]
    scale = np.abs(Pd).max()
    assert np.allclose(mine[0], Pd, rtol=5e-5, atol=5e-6 * scale)
    assert np.allclose(np.where(nbr >= 0, mine[1:], 0.0), Po, rtol=5e-5, atol=5e-6 * scale)

    # ---- lagged corner operators
    KC = t.table(10, np.int32).size // N
    KB = t.table(12, np.int32).size // N
    KPN = t.table(14, np.int32).size // N
    ci, cw = t.table(10, np.int32).reshape(KC, N), t.table(11, np.float32).reshape(KC, N)
    bi, bw = t.table(12, np.int32).reshape(KB, N), t.table(13, np.float32).reshape(KB, N)
    pi, pf = t.table(14, np.int32).reshape(KPN, N), t.table(15, np.int32).reshape(KPN, N)
    pwp, pwn = t.table(16, np.float32).reshape(KPN, N), t.table(17, np.float32).reshape(KPN, N)
    u = rng.standard_normal(N)
    ub = rng.standard_normal(max(NB, 1))
    for b, blk in enumerate(d.blocks):       # the same boundary values in the oracle's per-face arrays (component 0)
        for f, bd in enumerate(blk.bounds):
            if bd.type == mbo.FIXED:
                s0 = slot_of[(b, f)]
                bd.velocity[0] = ub[s0: s0 + bd.velocity.shape[1]]
    S_mine = (cw * u[ci]).sum(0) + ((bw * ub[bi]).sum(0) if KB else 0.0)
    S_ref = np.array([d.nonortho_rhs(b, pos, 0, lambda q: u[q], lambda bb, ff, pp: d.bound_value(bb, ff, pp, 0),
                                     flags, False) for b, pos in d.cells()]) / nu
    assert np.allclose(S_mine, S_ref, rtol=1e-4, atol=1e-5 * max(np.abs(S_ref).max(), 1e-12))
    p = rng.standard_normal(N)
    rface = np.where(np.take_along_axis(nbr, pf, axis=0) >= 0, rA[np.maximum(np.take_along_axis(nbr, pf, axis=0), 0)], 0.0)
    N_mine = ((pwp * rA[None, :] + pwn * rface) * p[pi]).sum(0) if KPN else np.zeros(N)
    N_ref = d.pressure_nonortho(p, A, flags)
    if flags == 10:  # nothing but the orthogonal Laplacian in the matrix: symmetric, rows sum to zero
        assert np.abs(mine[0] + np.where(nbr >= 0, mine[1:], 0.0).sum(0)).max() < 1e-5 * scale
    assert np.allclose(N_mine, N_ref, rtol=1e-4, atol=1e-5 * max(np.abs(N_ref).max(), 1e-12))
    t.close()


def test_host_only_handle_refuses_compute():
    t = HostTables(H.split_rotated_channel())
    rc = t.lib.fg_mb_bind(t.h, None, None, None, None)
    assert rc < 0
    assert t.lib.fg_mb_unit_pressure_matrix(t.h, None) < 0
    t.close()
