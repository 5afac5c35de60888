"""The two-kernel BiCGStab iteration of the single-block path (csrc/fg_bicgstab.hip k_bicgf_a / k_bicgf_b, the default) against
the direct solve and against the five-kernel form it replaces (FG_BICG_FUSED=0 at fg_create): the same recurrence
(bicgstab_solver_kernel.cu:63-411) with rho_{i+1} taken from rw.s - omega rw.t, so the iterates agree to rounding and the
iteration counts to a few."""
import numpy as np
import pytest

from oracle import piso_oracle as O
from tests.helpers import make_case, rel_err

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _solve(case, dt, fused, monkeypatch, tol=1e-7, max_iterations=5000, from_result=False, for_scalar=False):
    monkeypatch.setenv("FG_BICG_FUSED", "2" if fused else "0")      # read once per handle, at fg_create (2: also in 3-D)
    ns = case.native()
    ns.set_advection_start(from_result)
    ns.setup_advection(dt, for_scalar=for_scalar, channel=0)
    info = ns.solve_advection(for_scalar=for_scalar, tol=tol, max_iterations=max_iterations)
    shape = (case.B,) + case.shape if for_scalar else (case.B, case.dims) + case.shape
    x = _np(ns.buffer(7 if for_scalar else 3, shape))
    ns.close()
    return x, info


CASES = [dict(dims=2, n=(32, 24), fixed_axes=(1,), B=3, seed=5), dict(dims=2, n=(30, 17), fixed_axes=(0, 1), B=2, seed=6),
         dict(dims=3, n=(16, 12, 8), fixed_axes=(1,), B=2, seed=7), dict(dims=3, n=(9, 8, 7), fixed_axes=(), B=2, seed=8),
         dict(dims=2, n=(64, 32), fixed_axes=(0,), B=2, seed=9, through_flow_axis=0)]


@pytest.mark.parametrize("kw", CASES)
def test_fused_iteration_matches_direct_solve_and_the_five_kernel_form(kw, monkeypatch):
    case = make_case(vel_scale=0.4, nu=0.03, **kw)
    dt = 0.08
    xf, inf_f = _solve(case, dt, True, monkeypatch)
    x5, inf_5 = _solve(case, dt, False, monkeypatch)
    assert all(i.converged and i.is_finite for i in inf_f) and all(i.converged for i in inf_5)
    g = case.grid()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)
        C, _, _ = O.build_advection_matrix(dom, dt)
        rhs = O.advection_rhs_velocity(dom, dt)
        for comp in range(case.dims):
            x_ref = O.solve_direct(C, rhs[comp].ravel()).reshape(case.shape)
            assert rel_err(xf[b, comp], x_ref) < 3e-5, (b, comp)
    assert rel_err(xf, x5) < 1e-5
    for a, b in zip(inf_f, inf_5):
        # (the last iterations of a solve at 1e-7 sit at the fp32 rounding level of the residual: a count can move by a few)
        assert abs(a.used_iterations - b.used_iterations) <= max(3, b.used_iterations // 4), (a.used_iterations, b.used_iterations)
    assert max(i.used_iterations for i in inf_f) >= 3          # the case does iterate


def test_fused_iteration_cap_warm_start_and_scalar(monkeypatch):
    case = make_case(dims=2, n=(32, 24), fixed_axes=(1,), B=2, seed=11, vel_scale=0.4, nu=0.03, n_scalars=1)
    # iteration cap: both forms stop after the same two iterations and report it
    xf, inf_f = _solve(case, 0.08, True, monkeypatch, max_iterations=2)
    x5, inf_5 = _solve(case, 0.08, False, monkeypatch, max_iterations=2)
    assert not any(i.converged for i in inf_f) and not any(i.converged for i in inf_5)
    assert [i.used_iterations for i in inf_f] == [i.used_iterations for i in inf_5] == [2] * (case.B * 2)
    assert rel_err(xf, x5) < 1e-5
    assert np.allclose([i.final_residual for i in inf_f], [i.final_residual for i in inf_5], rtol=1e-3)
    # passive scalar (one system per env)
    xf, inf_f = _solve(case, 0.08, True, monkeypatch, for_scalar=True)
    x5, inf_5 = _solve(case, 0.08, False, monkeypatch, for_scalar=True)
    assert all(i.converged for i in inf_f) and rel_err(xf, x5) < 1e-5
    # start vector = the solution: no iteration (reported as -1, like the five-kernel form)
    monkeypatch.setenv("FG_BICG_FUSED", "1")
    ns = case.native()
    ns.setup_advection(0.08)
    ns.set_advection_start(False)
    first = ns.solve_advection(tol=1e-6)
    ns.set_advection_start(True)
    again = ns.solve_advection(tol=1e-5)
    assert all(i.used_iterations > 0 for i in first) and all(i.used_iterations == -1 and i.converged for i in again)
    ns.close()


def test_fused_step_is_bit_reproducible(monkeypatch):
    """Two handles, the same inputs: the same bits (order-independent reductions + a fixed kernel sequence)."""
    monkeypatch.setenv("FG_BICG_FUSED", "1")
    case = make_case(dims=2, n=(64, 32), fixed_axes=(1,), B=4, seed=3, vel_scale=0.4, with_source=True)
    outs = []
    for _ in range(2):
        ns = case.native()
        for _ in range(3):
            ok, stats = ns.piso_step(0.03, advection_tol=1e-6, pressure_tol=1e-6)
            assert ok
        outs.append((ns.velocity.clone(), ns.pressure.clone()))
        ns.close()
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()


# ---- 3-D: the two-kernel iteration as z-marching LDS-ring kernels (csrc/fg_bicgstab3d.hip; the default in 3-D when the grid
# fits the tiles and fills the chip, forced here on small grids with FG_BICG3 = planes per z-chunk)
def _solve3(case, dt, mode, monkeypatch, zc=4, bxl=None, **kw):
    """mode: 'zmarch' | 'brick' (two-kernel brick form) | 'five'."""
    monkeypatch.setenv("FG_BICG3", str(zc) if mode == "zmarch" else "0")
    if bxl:
        monkeypatch.setenv("FG_BICG3_BXL", str(bxl))
    else:
        monkeypatch.delenv("FG_BICG3_BXL", raising=False)
    return _solve(case, dt, mode != "five", monkeypatch, **kw)


ZCASES = [
    # (case kwargs, planes per chunk, tile lanes): walls in y / periodic everywhere / walls in z and x / ragged last chunk
    (dict(dims=3, n=(64, 16, 8), fixed_axes=(1,), B=2, seed=21), 4, None),
    (dict(dims=3, n=(64, 32, 9), fixed_axes=(), B=2, seed=22), 4, None),
    (dict(dims=3, n=(128, 16, 6), fixed_axes=(0, 2), B=3, seed=23), 3, None),
    (dict(dims=3, n=(128, 16, 8), fixed_axes=(1,), B=2, seed=24), 8, 32),
    (dict(dims=3, n=(128, 32, 12), fixed_axes=(2,), B=1, seed=25), 5, 16),
]


@pytest.mark.parametrize("kw,zc,bxl", ZCASES)
def test_zmarch_iteration_matches_direct_solve_and_the_brick_kernels(kw, zc, bxl, monkeypatch):
    case = make_case(vel_scale=0.4, nu=0.03, **kw)
    dt = 0.08
    xz, inf_z = _solve3(case, dt, "zmarch", monkeypatch, zc=zc, bxl=bxl)
    x5, inf_5 = _solve3(case, dt, "five", monkeypatch)
    assert all(i.converged and i.is_finite for i in inf_z) and all(i.converged for i in inf_5)
    g = case.grid()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)
        C, _, _ = O.build_advection_matrix(dom, dt)
        rhs = O.advection_rhs_velocity(dom, dt)
        for comp in range(case.dims):
            x_ref = O.solve_direct(C, rhs[comp].ravel()).reshape(case.shape)
            assert rel_err(xz[b, comp], x_ref) < 3e-5, (b, comp)
    assert rel_err(xz, x5) < 1e-5
    for a, b in zip(inf_z, inf_5):
        assert abs(a.used_iterations - b.used_iterations) <= max(3, b.used_iterations // 4), (a.used_iterations, b.used_iterations)
    assert max(i.used_iterations for i in inf_z) >= 3


def test_zmarch_cap_warm_start_scalar_and_uneven_convergence(monkeypatch):
    case = make_case(dims=3, n=(64, 16, 8), fixed_axes=(1,), B=3, seed=31, vel_scale=0.4, nu=0.03, n_scalars=1)
    # iteration cap: same two iterations, same report as the five kernels
    xz, inf_z = _solve3(case, 0.08, "zmarch", monkeypatch, max_iterations=2)
    x5, inf_5 = _solve3(case, 0.08, "five", monkeypatch, max_iterations=2)
    assert [i.used_iterations for i in inf_z] == [i.used_iterations for i in inf_5] == [2] * (case.B * 3)
    assert not any(i.converged for i in inf_z) and rel_err(xz, x5) < 1e-5
    assert np.allclose([i.final_residual for i in inf_z], [i.final_residual for i in inf_5], rtol=1e-3)
    # passive scalar: one system per env (the nc = 1 instance)
    xz, inf_z = _solve3(case, 0.08, "zmarch", monkeypatch, for_scalar=True)
    x5, inf_5 = _solve3(case, 0.08, "five", monkeypatch, for_scalar=True)
    assert all(i.converged for i in inf_z) and rel_err(xz, x5) < 1e-5
    # a loose tolerance: the systems of an env end in different iterations (the slow path of the kernels: per-system modes,
    # converged-on-s half updates) and still agree with the five kernels
    case.velocity[:, 2] *= 1e-3              # (an absolute tolerance: the small component and the slow env finish early)
    case.velocity[1] *= 0.05
    xz, inf_z = _solve3(case, 0.08, "zmarch", monkeypatch, tol=2e-4)
    x5, inf_5 = _solve3(case, 0.08, "five", monkeypatch, tol=2e-4)
    assert all(i.converged for i in inf_z)
    assert len({i.used_iterations for i in inf_z}) > 1, [i.used_iterations for i in inf_z]
    assert all(abs(a.used_iterations - b.used_iterations) <= 2 for a, b in zip(inf_z, inf_5))
    assert np.abs(xz - x5).max() < 2e-4 * 0.08 * 10            # both within the tolerance of the same answer (x ~ dt rhs)
    # the same with the init kernel kept (FG_BICG3_MIX = 7: no folded start) -- the folded start is the same recurrence
    monkeypatch.setenv("FG_BICG3_MIX", "7")
    xk, inf_k = _solve3(case, 0.08, "zmarch", monkeypatch, tol=2e-4)
    monkeypatch.delenv("FG_BICG3_MIX")
    assert all(abs(a.used_iterations - b.used_iterations) <= 2 for a, b in zip(inf_k, inf_z))
    assert np.abs(xk - xz).max() < 2e-4 * 0.08 * 10
    # start vector = the solution: no iteration
    monkeypatch.setenv("FG_BICG3", "4")
    ns = case.native()
    ns.setup_advection(0.08)
    ns.set_advection_start(False)
    first = ns.solve_advection(tol=1e-6)
    ns.set_advection_start(True)
    again = ns.solve_advection(tol=1e-5)
    assert all(i.used_iterations > 0 for i in first) and all(i.used_iterations == -1 and i.converged for i in again)
    ns.close()


def test_zmarch_step_is_bit_reproducible_and_batch_independent(monkeypatch):
    monkeypatch.setenv("FG_BICG3", "4")
    case = make_case(dims=3, n=(64, 16, 8), fixed_axes=(1,), B=3, seed=33, vel_scale=0.4, with_source=True)
    outs = []
    for _ in range(2):
        ns = case.native()
        for _ in range(2):
            ok, stats = ns.piso_step(0.03, advection_tol=1e-6, pressure_tol=1e-6)
            assert ok
        outs.append((ns.velocity.clone(), ns.pressure.clone()))
        ns.close()
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()


def test_solver_state_prepared_by_the_assembly_kernels_survives_foreign_calls_in_between(monkeypatch):
    """Round 4: the state of a solve (accumulators, flags, info) is prepared by the kernel launched in front of it -- k_adv_build for the
    BiCGStab solve, k_div for the pressure CG (FgBicgBegin / FgCgBegin) -- and the solver skips its own begin launch when the record of
    who prepared what matches.  The two solvers share flags and info, so a call of the OTHER solver between assembly and solve must
    drop the record: setup_pressure_rhs -> solve_pressure gives bit for bit the same pressure with and without an advection solve
    in between (the right-hand side was built before it), and setup_advection -> solve_advection the same velocity with and without
    a pressure solve in between; repeated solves on one assembly (the ladder's situation) are bit-identical too."""
    import torch

    case = make_case(dims=2, n=(64, 32), fixed_axes=(1,), B=3, seed=21, vel_scale=0.4, nu=0.03)
    dt = 0.05
    shape_v = (case.B, case.dims) + case.shape
    shape_p = (case.B,) + case.shape

    def pressure(with_foreign_call):
        ns = case.native()
        ns.setup_advection(dt)
        assert all(i.converged for i in ns.solve_advection(tol=1e-7))
        ns.setup_pressure_matrix()
        ns.setup_pressure_rhs(dt)
        if with_foreign_call:
            assert all(i.converged for i in ns.solve_advection(tol=1e-7))      # BiCGStab on the same assembly: rewrites flags / info
        info = ns.solve_pressure(tol=1e-7)
        p = ns.buffer(6, shape_p).clone()
        info2 = ns.solve_pressure(tol=1e-7)                                    # a second solve on the same right-hand side
        p2 = ns.buffer(6, shape_p).clone()
        ns.close()
        return p, p2, info, info2

    p_a, p_a2, info_a, info_a2 = pressure(False)
    p_b, p_b2, info_b, _ = pressure(True)
    assert all(i.converged and i.is_finite for i in info_a + info_b + info_a2)
    assert [i.used_iterations for i in info_a] == [i.used_iterations for i in info_b] == [i.used_iterations for i in info_a2]
    assert torch.equal(p_a, p_b) and torch.equal(p_a, p_a2) and torch.equal(p_b, p_b2)
    assert float(p_a.abs().max()) > 0

    def velocity(with_foreign_call):
        ns = case.native()
        ns.setup_advection(dt)
        if with_foreign_call:
            ns.setup_pressure_matrix()
            ns.setup_pressure_rhs(dt)                                          # k_div prepares the CG's state over the BiCGStab's
            ns.solve_pressure(tol=1e-7)
        info = ns.solve_advection(tol=1e-7)
        u = ns.buffer(3, shape_v).clone()
        ns.close()
        return u, info

    u_a, inf_a = velocity(False)
    u_b, inf_b = velocity(True)
    assert all(i.converged for i in inf_a + inf_b)
    assert [i.used_iterations for i in inf_a] == [i.used_iterations for i in inf_b]
    assert torch.equal(u_a, u_b)


def test_subbatched_solve_is_the_whole_batch_solve_bit_for_bit(monkeypatch):
    """Round 4 (FG_BICG_SUB, fg_bicgstab_solve): when the working set of the 2-D two-kernel BiCGStab is far beyond the Infinity Cache
    the envs are solved in cache-sized groups, each group through all its iterations and polls before the next (FgGrid::b0; the
    check kernel judges the group's systems only; the next group's first kernels go out behind the current group's check).  The
    systems are independent and the per-system arithmetic does not change: forced groups of 2 envs (5 envs: 2 + 2 + 1) give the
    whole-batch solve bit for bit, iteration counts included -- with envs that converge at different iterations, from zero and
    from a start vector, velocity and scalar systems."""
    import torch

    case = make_case(dims=2, n=(64, 32), fixed_axes=(1,), B=5, seed=31, vel_scale=0.4, nu=0.03, n_scalars=1)
    scale = torch.tensor([0.2, 1.0, 0.5, 1.5, 0.8]).view(5, 1, 1, 1)

    def solve(sub, from_result, for_scalar):
        monkeypatch.setenv("FG_BICG_SUB", str(sub))           # read at fg_create
        ns = case.native()
        ns.velocity.mul_(scale.to(ns.velocity.device))           # envs of different stiffness: different iteration counts
        ns.set_advection_start(from_result)
        ns.setup_advection(0.08, for_scalar=for_scalar, channel=0)
        if from_result:
            ns.solve_advection(for_scalar=for_scalar, tol=1e-3)   # leaves a start vector behind
        info = ns.solve_advection(for_scalar=for_scalar, tol=1e-7)
        shape = (case.B,) + case.shape if for_scalar else (case.B, case.dims) + case.shape
        x = ns.buffer(7 if for_scalar else 3, shape).clone()
        form = ns.advection_solver_form()
        ns.close()
        return x, [i.used_iterations for i in info], all(i.converged for i in info), form

    for from_result in (False, True):
        for for_scalar in (False, True):
            if for_scalar and from_result:
                continue                                          # (scalar solves always start from zero)
            x0, it0, ok0, form = solve(0, from_result, for_scalar)
            x2, it2, ok2, _ = solve(2, from_result, for_scalar)
            assert form == "two-brick" and ok0 and ok2
            assert it0 == it2, (it0, it2)
            assert torch.equal(x0, x2)
            if not for_scalar:
                assert len(set(it0)) > 1, it0                     # the groups do end at different iterations
