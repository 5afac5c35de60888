"""The six-launch Helmholtz-preconditioned BiCGStab (csrc/fg_fftbicg.hip: k_fbicg_fwd / k_fbicg_inv around the y-line kernel of
fg_linepre.hip) against the oracle's direct solve and against the eleven-launch iteration it replaces (FG_BICG_PFUSED=0), through the C
ABI.  Replaces bicgstabSolveGPU with its preconditioned branch (bicgstab_solver_kernel.cu:63-411, 191-226, 288-293) on the matrix
of PISO_build_matrix (PISO_multiblock_cuda_kernel.cu:3616-3880) -- the systems of the RBC envs (rbc_env_base.py:285-329)."""
import numpy as np
import pytest
import torch

from oracle import piso_oracle as O
from tests.helpers import make_case, rel_err
from tests.test_gpu_linepre import _np, _wall_refined

pytestmark = pytest.mark.gpu


def _solve(case, dt, fused, for_scalar, from_result, tol, monkeypatch, max_iterations=5000, need_converged=True):
    monkeypatch.setenv("FG_BICG_PFUSED", "1" if fused else "0")
    ns = case.native()
    assert ns.has_helmholtz
    ns.set_advection_start(from_result)
    ns.set_advection_preconditioner(3)
    ns.setup_advection(dt, for_scalar=for_scalar, channel=0)
    info = ns.solve_advection(for_scalar=for_scalar, tol=tol, max_iterations=max_iterations)
    torch.cuda.synchronize()
    if need_converged:
        assert all(i.converged and i.is_finite for i in info), (fused, [(i.used_iterations, i.final_residual) for i in info])
    shape = (case.B,) + case.shape if for_scalar else (case.B, case.dims) + case.shape
    x = _np(ns.buffer(7 if for_scalar else 3, shape))
    its = [i.used_iterations for i in info]
    res = [i.final_residual for i in info]
    ns.close()
    return x, its, res


@pytest.mark.parametrize("n", [(64, 20), (128, 32), (512, 24)])
@pytest.mark.parametrize("from_result", [False, True])
def test_velocity_systems(n, from_result, monkeypatch):
    """two velocity components per env, three envs with their own time steps (the Helmholtz factors are per env), zero start
    vector (folded start: no init kernel) and velocityResult start (init kernel); rows not a multiple of eight."""
    case = _wall_refined(make_case(dims=2, n=n, fixed_axes=(1,), B=3, seed=4, nu=0.05, vel_scale=0.3, stretch=0.0), ratio=12.0)
    dt = [0.05, 0.02, 0.08]
    xf, itf, _ = _solve(case, dt, True, False, from_result, 1e-7, monkeypatch)
    xe, ite, _ = _solve(case, dt, False, False, from_result, 1e-7, monkeypatch)
    g = case.grid()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)
        C, _, _ = O.build_advection_matrix(dom, dt[b])
        rhs = O.advection_rhs_velocity(dom, dt[b])
        for comp in range(2):
            x_ref = O.solve_direct(C, rhs[comp].ravel()).reshape(case.shape)
            assert rel_err(xf[b, comp], x_ref) < 1e-4, (b, comp)     # (scale = max of the component: the v component is ~2e-3 here)
            assert rel_err(xe[b, comp], x_ref) < 1e-4, (b, comp)
            assert rel_err(xf[b, comp], xe[b, comp]) < 5e-5
    print(f"FUSED-BICG {n} from_result={from_result}: iterations fused {itf}, eleven launches {ite}")
    assert all(abs(a - c) <= 1 for a, c in zip(itf, ite)), (itf, ite)
    assert max(itf) + 1 <= 8


@pytest.mark.parametrize("neumann", [(), (2, 3)])
def test_scalar_system_and_a_masked_env(neumann, monkeypatch):
    case = _wall_refined(make_case(dims=2, n=(128, 28), fixed_axes=(1,), B=3, seed=9, n_scalars=1, neumann_faces=neumann, stretch=0.0),
                         ratio=10.0)
    dt = [0.05, 0.0, 0.03]      # env 1 masked out
    xf, itf, _ = _solve(case, dt, True, True, False, 1e-7, monkeypatch)
    xe, ite, _ = _solve(case, dt, False, True, False, 1e-7, monkeypatch)
    g = case.grid()
    for b in (0, 2):
        dom = case.oracle_domain(b, g)
        Cs, _, _ = O.build_advection_matrix(dom, dt[b], for_scalar=True, channel=0)
        x_ref = O.solve_direct(Cs, O.advection_rhs_scalar(dom, dt[b])[0].ravel()).reshape(case.shape)
        assert rel_err(xf[b], x_ref) < 3e-5
        assert rel_err(xf[b], xe[b]) < 2e-5
    assert itf[1] == -1 and ite[1] == -1
    assert all(abs(a - c) <= 1 for a, c in zip(itf, ite)), (itf, ite)


def test_iteration_cap_and_reproducible_bits(monkeypatch):
    case = _wall_refined(make_case(dims=2, n=(128, 32), fixed_axes=(1,), B=2, seed=6, nu=0.02, vel_scale=0.8, stretch=0.0), ratio=20.0)
    dt = 0.1
    x1, it1, res1 = _solve(case, dt, True, False, False, 1e-9, monkeypatch, max_iterations=2, need_converged=False)
    x2, it2, res2 = _solve(case, dt, True, False, False, 1e-9, monkeypatch, max_iterations=2, need_converged=False)
    assert np.isfinite(x1).all() and max(it1) <= 2
    assert np.array_equal(x1, x2) and it1 == it2 and res1 == res2
    xe, ite, rese = _solve(case, dt, False, False, False, 1e-9, monkeypatch, max_iterations=2, need_converged=False)
    assert rel_err(x1, xe) < 1e-4      # the same two iterations


def test_rbc_like_piso_step_with_both_fused_solvers(monkeypatch):
    """a whole PISO step with passive scalar + buoyancy on a periodic, wall-refined grid: six-launch BiCGStab + three-launch CG
    against the classic kernels and the oracle."""
    case = _wall_refined(make_case(dims=2, n=(128, 32), fixed_axes=(1,), B=2, seed=3, n_scalars=1, stretch=0.0, nu=0.02,
                                   vel_scale=0.2, wall_motion=0.0), ratio=10.0)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FG_BICG_PFUSED", mode)
        monkeypatch.setenv("FG_CG_FUSED", mode)
        ns = case.native()
        ns.set_advection_preconditioner(3)
        ns.set_velocity_source(torch.zeros_like(ns.velocity))
        ok, stats = ns.piso_step([0.04, 0.025], advection_tol=1e-7, pressure_tol=1e-7, buoyancy_axis=1, buoyancy_factor=1.0)
        torch.cuda.synchronize()
        assert ok, stats
        out[mode] = (_np(ns.velocity), _np(ns.scalar), _np(ns.pressure), stats)
        ns.close()
    u1, t1, p1, s1 = out["1"]
    u0, t0, p0, s0 = out["0"]
    print("RBC-like step iterations fused", s1, "classic", s0)
    # (two preconditioners of the pressure CG -- row-mean against A = 1 factors -- give two iterates inside the same tolerance)
    assert rel_err(t1, t0) < 1e-5 and rel_err(u1, u0) < 2e-4 and rel_err(p1, p0) < 1e-3
    g = case.grid()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)

        def buoy(d, _dt):
            src = np.zeros_like(d.velocity)
            src[1] = d.scalar[0]
            d.velocity_source = src

        O.piso_split_step(dom, [0.04, 0.025][b], prep_fn={"PRE_VELOCITY_SETUP": [buoy]})
        assert rel_err(t1[b], dom.scalar) < 1e-5
        assert rel_err(u1[b], dom.velocity) < 1e-4      # (error scale = the buoyancy forcing the projection cancels, as in test_buoyancy_fused_rbc_like_step)


@pytest.mark.parametrize("from_result", [False, True])
def test_line_sweeps_reach_the_direct_solve_and_hand_over_what_they_cannot(from_result, monkeypatch):
    """Round 6 (opt-in: FG_ADV_LINESWEEP=1): the velocity systems of a wall-refined periodic-x grid by line sweeps
    x <- (D + O_y)^-1 (b - O_x x) (``k_line_sweep_y``, csrc/fg_linepre.hip: the y-line solve with the x stencil folded into its load
    phase) instead of the Helmholtz-preconditioned BiCGStab.  Small time steps (x part of a row well below its diagonal): settled by
    the sweeps, the direct solve's answer to the solver tolerance.  A time step 40 x larger: the sweeps do not contract by 0.7, the
    first check sees it from the residuals of sweeps 3 and 5, BiCGStab solves the system from a cleared start and the kind backs off."""
    case = _wall_refined(make_case(dims=2, n=(128, 32), fixed_axes=(1,), B=3, seed=4, nu=0.05, vel_scale=0.3, stretch=0.0), ratio=12.0)
    monkeypatch.setenv("FG_ADV_LINESWEEP", "1")
    for dt, settled in (([0.002, 0.001, 0.003], True), ([0.08, 0.04, 0.12], False)):
        ns = case.native()
        ns.set_advection_start(from_result)
        ns.set_advection_preconditioner(3)
        ns.setup_advection(dt)
        info = ns.solve_advection(tol=2e-6 / min(dt))      # (a residual fp32 can show: the right-hand side is u / dt ~ 300, its rounding ~4e-4)
        torch.cuda.synchronize()
        assert all(i.converged and i.is_finite for i in info), [(i.used_iterations, i.final_residual) for i in info]
        counts = ns.advection_jacobi_counts()
        x = _np(ns.buffer(3, (case.B, 2) + case.shape))
        ns.close()
        assert counts == ({"settled_by_sweeps": 1, "handed_to_bicgstab": 0} if settled else {"settled_by_sweeps": 0, "handed_to_bicgstab": 1}), (dt, counts)
        print(f"LINE-SWEEPS dt {dt} from_result={from_result}: {counts}, iterations {[i.used_iterations for i in info]}")
        g = case.grid()
        for b in range(case.B):
            dom = case.oracle_domain(b, g)
            C, _, _ = O.build_advection_matrix(dom, dt[b])
            rhs = O.advection_rhs_velocity(dom, dt[b])
            for comp in range(2):
                x_ref = O.solve_direct(C, rhs[comp].ravel()).reshape(case.shape)
                assert rel_err(x[b, comp], x_ref) < 1e-4, (dt, b, comp)
