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
