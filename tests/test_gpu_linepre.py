"""The y-line right preconditioner of the advection-diffusion BiCGStab (csrc/fg_linepre.hip) and the preconditioner policy of the
single-block path (fg_set_advection_preconditioner): the reference's preconditionBiCG / BiCG_precondition_fallback
(PISOtorch_diff.py:449-476; ILU(0) there, bicgstab_solver_kernel.cu:191-226).  The preconditioned solves must give the direct
solve's answer like the plain ones, in far fewer iterations on grids refined towards a y wall."""
import numpy as np
import pytest
import torch

from oracle import piso_oracle as O
from tests.helpers import make_case, rel_err

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _wall_refined(case, ratio=60.0):
    """Replace the y widths by a two-sided geometric wall refinement (largest / smallest width = ratio)."""
    ny = len(case.widths[1])
    half = ny // 2
    g = ratio ** (1.0 / max(half - 1, 1))
    w = np.concatenate([g ** np.arange(half), g ** np.arange(ny - half)[::-1]])
    w = (w / w.sum()).astype(np.float32)
    case.widths[1] = w
    case.edges[1] = np.concatenate([[0.0], np.cumsum(w.astype(np.float64))])
    return case


def _solve_all_modes(case, dt, for_scalar, tol=1e-7, modes=(0, 1)):
    out = {}
    for mode in modes:
        ns = case.native()
        ns.set_advection_start(False)
        ns.set_advection_preconditioner(mode)
        ns.setup_advection(dt, for_scalar=for_scalar, channel=0)
        info = ns.solve_advection(for_scalar=for_scalar, tol=tol)
        assert all(i.converged and i.is_finite for i in info), (mode, [i.final_residual for i in info], [i.used_iterations for i in info])
        shape = (case.B,) + case.shape if for_scalar else (case.B, case.dims) + case.shape
        out[mode] = (_np(ns.buffer(7 if for_scalar else 3, shape)), max(i.used_iterations for i in info) + 1)
        ns.close()
    return out


@pytest.mark.parametrize("dims,n,fixed_axes", [(2, (64, 48), (1,)), (2, (30, 40), (0, 1)), (3, (32, 32, 8), (1,)),
                                               (2, (64, 240), (1,))])
def test_preconditioned_velocity_solve_matches_the_direct_solve_in_fewer_iterations(dims, n, fixed_axes):
    """nx % 4 == 0: the LDS kernels; (30, 40) and ny = 240 (> 208 rows of LDS): the streaming kernels."""
    case = _wall_refined(make_case(dims=dims, n=n, fixed_axes=fixed_axes, B=2, seed=4, nu=0.05, vel_scale=0.3), ratio=10.0)
    dt = 0.05
    out = _solve_all_modes(case, dt, for_scalar=False)
    g = case.grid()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)
        C, _, _ = O.build_advection_matrix(dom, dt)
        rhs = O.advection_rhs_velocity(dom, dt)
        for comp in range(dims):
            x_ref = O.solve_direct(C, rhs[comp].ravel()).reshape(case.shape)
            for mode in (0, 1):
                assert rel_err(out[mode][0][b, comp], x_ref) < 3e-5, (mode, b, comp)
    plain, pre = out[0][1], out[1][1]
    assert pre * 2 <= plain and pre <= 40, (plain, pre)


def test_fallback_rung_rescues_a_system_the_plain_recurrence_cannot_solve():
    """Wall refinement 60 : 1 with nu dt / h^2 in the thousands: the plain fp32 recurrence stagnates or diverges (final residuals up
    to 1e7 observed), the reference's answer is its preconditioned rung (BiCG_precondition_fallback, PISOtorch_diff.py:449-476).
    Mode 2 repeats exactly those solves with the line preconditioner; the result is the direct solve's."""
    case = _wall_refined(make_case(dims=2, n=(64, 48), fixed_axes=(1,), B=2, seed=4, nu=0.05, vel_scale=0.3), ratio=60.0)
    dt = 0.05
    ns = case.native()
    ns.set_advection_start(False)
    ns.set_advection_preconditioner(0)
    ns.setup_advection(dt)
    plain = ns.solve_advection(tol=1e-7, max_iterations=400)
    assert not all(i.converged for i in plain)                      # the premise: the first rung fails here
    ns.set_advection_preconditioner(2)
    info = ns.solve_advection(tol=1e-7, max_iterations=400)
    assert all(i.converged and i.is_finite for i in info) and ns.advection_retries() == 1
    x = _np(ns.buffer(3, (case.B, case.dims) + case.shape))
    ns.close()
    g = case.grid()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)
        C, _, _ = O.build_advection_matrix(dom, dt)
        rhs = O.advection_rhs_velocity(dom, dt)
        for comp in range(2):
            assert rel_err(x[b, comp], O.solve_direct(C, rhs[comp].ravel()).reshape(case.shape)) < 3e-5


def test_preconditioned_scalar_solve_and_periodic_y_wrap_is_ignored():
    case = _wall_refined(make_case(dims=2, n=(32, 24), fixed_axes=(1,), B=2, seed=9, n_scalars=1, neumann_faces=(3,)), ratio=10.0)
    out = _solve_all_modes(case, 0.05, for_scalar=True)
    g = case.grid()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)
        Cs, _, _ = O.build_advection_matrix(dom, 0.05, for_scalar=True, channel=0)
        x_ref = O.solve_direct(Cs, O.advection_rhs_scalar(dom, 0.05)[0].ravel()).reshape(case.shape)
        assert rel_err(out[0][0][b], x_ref) < 3e-5 and rel_err(out[1][0][b], x_ref) < 3e-5
    assert out[1][1] < out[0][1]
    # y periodic: the wrap-around coefficients are not part of M (a non-cyclic line solve); still the same answer
    case = make_case(dims=2, n=(32, 24), fixed_axes=(0,), B=2, seed=2, nu=0.2, vel_scale=0.3)
    out = _solve_all_modes(case, 0.1, for_scalar=False)
    assert rel_err(out[1][0], out[0][0]) < 3e-5


def test_fallback_mode_repeats_only_failed_solves_with_the_preconditioner():
    """BiCG_precondition_fallback (mode 2): a solve that runs out of iterations is repeated from zero with the preconditioner."""
    case = _wall_refined(make_case(dims=2, n=(64, 48), fixed_axes=(1,), B=2, seed=4, nu=0.05, vel_scale=0.3), ratio=10.0)
    ns = case.native()
    ns.set_advection_start(False)
    ns.set_advection_preconditioner(2)
    ns.setup_advection(0.05)
    info = ns.solve_advection(tol=1e-7, max_iterations=500)      # plain converges: no retry
    assert all(i.converged for i in info) and ns.advection_retries() == 0
    plain_its = max(i.used_iterations for i in info) + 1
    x_plain = _np(ns.buffer(3, (case.B, case.dims) + case.shape))
    cap = max(plain_its // 2, 2)
    info = ns.solve_advection(tol=1e-7, max_iterations=cap)      # plain cannot: repeated with the line solve, which can
    assert plain_its > cap and ns.advection_retries(reset=True) == 1 and ns.advection_retries() == 0
    assert all(i.converged for i in info)
    assert rel_err(_np(ns.buffer(3, (case.B, case.dims) + case.shape)), x_plain) < 3e-5
    ns.set_advection_preconditioner(0)
    info = ns.solve_advection(tol=1e-7, max_iterations=cap)      # mode 0: the failure is reported
    assert not all(i.converged for i in info) and ns.advection_retries() == 0
    ns.close()


def test_rbc_env_uses_the_line_solve_and_steps_like_the_plain_solver():
    """RBC2D with a strongly wall-refined grid: the policy switch preconditions every solve; one env step with and without it
    agree to the solver tolerance, with fewer iterations.  Without the switch the env runs the reference's rule (mode 2: only
    failed solves are repeated with the preconditioner, BiCG_precondition_fallback)."""
    import fluidgym_amd

    out = {}
    for on in (True, False):
        # (the Helmholtz preconditioner, which the RBC grids get by default, is switched off: this test is about the line rungs)
        old = fluidgym_amd.set_solver_policy(advection_line_preconditioner=on, advection_fd_preconditioner="never")
        try:
            env = fluidgym_amd.make("RBC2D-easy-v0", num_envs=2, n_heaters=4, resolution=8)
            env._non_uniform_grid_base = 1.3       # 20 rows: 1.3^9 = 10.6 (the registered base 1.02 refines by 1.2 only at this size)
            env.reset(seed=3)
            assert env._sim.advection_preconditioner == (1 if on else 2)
            solver = env._domain.solver
            solver.solver_counters(reset=True)
            a = env.sample_action()
            obs, reward, _, _, info = env.step(torch.zeros_like(a))
            c = solver.solver_counters()
            out[on] = (solver.velocity.clone(), reward.clone(), c["velocity"]["mean"], c["scalar"]["mean"])
            env.close()
        finally:
            fluidgym_amd.set_solver_policy(**old)
    u_on, r_on, v_on, s_on = out[True]
    u_off, r_off, v_off, s_off = out[False]
    assert torch.allclose(u_on, u_off, rtol=0, atol=2e-4 * float(u_off.abs().max()))
    assert torch.allclose(r_on, r_off, rtol=1e-3, atol=1e-5)
    assert v_on < v_off and s_on < s_off, (v_on, v_off, s_on, s_off)
