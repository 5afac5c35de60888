"""The separable Helmholtz preconditioner of the advection-diffusion BiCGStab (mode 3 of fg_set_advection_preconditioner;
csrc/fg_fdprecond.hip fg_fd_helmholtz_apply + csrc/fg_linepre.hip k_helm_coeffs): the exact inverse of the matrix without its
advective part, by fast diagonalisation along the periodic axes and a tridiagonal solve along y.  Same answer as the direct
solve, a handful of iterations where the plain recurrence of the reference (bicgstab_solver_kernel.cu:63-411) needs dozens."""
import numpy as np
import pytest
import torch

from oracle import piso_oracle as O
from tests.helpers import make_case, rel_err
from tests.test_gpu_linepre import _np, _wall_refined

pytestmark = pytest.mark.gpu


def _solve(case, dt, mode, for_scalar=False, tol=1e-7):
    ns = case.native()
    assert ns.has_helmholtz
    ns.set_advection_start(False)
    ns.set_advection_preconditioner(mode)
    ns.setup_advection(dt, for_scalar=for_scalar, channel=0)
    info = ns.solve_advection(for_scalar=for_scalar, tol=tol)
    assert all(i.converged and i.is_finite for i in info), (mode, [i.final_residual for i in info])
    shape = (case.B,) + case.shape if for_scalar else (case.B, case.dims) + case.shape
    x = _np(ns.buffer(7 if for_scalar else 3, shape))
    ns.close()
    return x, max(i.used_iterations for i in info) + 1


@pytest.mark.parametrize("dims,n", [(2, (64, 48)), (2, (128, 32)), (3, (32, 24, 16))])
def test_helmholtz_preconditioned_velocity_solve(dims, n):
    case = _wall_refined(make_case(dims=dims, n=n, fixed_axes=(1,), B=2, seed=4, nu=0.05, vel_scale=0.3, stretch=0.0), ratio=10.0)
    dt = 0.05
    x0, plain = _solve(case, dt, 0)
    x3, pre = _solve(case, dt, 3)
    g = case.grid()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)
        C, _, _ = O.build_advection_matrix(dom, dt)
        rhs = O.advection_rhs_velocity(dom, dt)
        for comp in range(dims):
            x_ref = O.solve_direct(C, rhs[comp].ravel()).reshape(case.shape)
            assert rel_err(x3[b, comp], x_ref) < 3e-5, (b, comp)
            assert rel_err(x0[b, comp], x_ref) < 3e-5
    print(f"HELMHOLTZ dims={dims} n={n}: plain {plain} iterations, preconditioned {pre}")
    assert pre <= 8 and pre * 3 <= plain, (plain, pre)


@pytest.mark.parametrize("neumann", [(), (3,), (2, 3)])
def test_helmholtz_preconditioned_scalar_solve_follows_the_wall_condition(neumann):
    """Dirichlet walls enter the diagonal of the y operator (one-sided coefficient), Neumann walls do not: a wrong wall term would
    still converge (it is only a preconditioner) but in many more iterations -- both the answer and the count are checked."""
    case = _wall_refined(make_case(dims=2, n=(64, 32), fixed_axes=(1,), B=2, seed=9, n_scalars=1, neumann_faces=neumann, stretch=0.0),
                         ratio=10.0)
    dt = 0.05
    x3, pre = _solve(case, dt, 3, for_scalar=True)
    _, plain = _solve(case, dt, 0, for_scalar=True)
    g = case.grid()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)
        Cs, _, _ = O.build_advection_matrix(dom, dt, for_scalar=True, channel=0)
        x_ref = O.solve_direct(Cs, O.advection_rhs_scalar(dom, dt)[0].ravel()).reshape(case.shape)
        assert rel_err(x3[b], x_ref) < 3e-5
    assert pre <= 8 and pre < plain, (plain, pre)


def test_pure_diffusion_is_solved_in_one_iteration():
    """u = 0: the matrix IS the Helmholtz operator, so the preconditioned system is the identity -- BiCGStab converges in its first
    iteration.  This pins the coefficients (dt, nu, the y operator with its wall terms, the eigenvalue sums and the 1/(hx hz)
    scale) against the assembled matrix itself."""
    case = _wall_refined(make_case(dims=2, n=(64, 32), fixed_axes=(1,), B=2, seed=1, nu=0.07, vel_scale=0.0, wall_motion=0.0, stretch=0.0),
                         ratio=6.0)
    case.velocity[:] = 0.0
    ns = case.native()
    ns.set_velocity_source(torch.randn_like(ns.velocity))
    ns.set_advection_start(False)
    ns.set_advection_preconditioner(3)
    ns.setup_advection([0.05, 0.02])
    info = ns.solve_advection(tol=1e-6)
    assert all(i.converged for i in info)
    assert max(i.used_iterations for i in info) <= 1, [i.used_iterations for i in info]     # one iteration (or its first half)
    ns.close()


def test_rbc_env_runs_preconditioned_by_default_and_agrees_with_the_plain_solver():
    import fluidgym_amd

    out = {}
    for pol in ("auto", "never"):
        old = fluidgym_amd.set_solver_policy(advection_fd_preconditioner=pol)
        try:
            env = fluidgym_amd.make("RBC2D-easy-v0", num_envs=2, n_heaters=4, resolution=8)
            env.reset(seed=3)
            assert env._sim.advection_preconditioner == (3 if pol == "auto" else 2)
            solver = env._domain.solver
            solver.solver_counters(reset=True)
            obs, reward, _, _, info = env.step(torch.zeros_like(env.sample_action()))
            c = solver.solver_counters()
            out[pol] = (solver.velocity.clone(), solver.scalar.clone(), reward.clone(), c["velocity"]["mean"], c["scalar"]["mean"])
            env.close()
        finally:
            fluidgym_amd.set_solver_policy(**old)
    u1, t1, r1, v1, s1 = out["auto"]
    u0, t0, r0, v0, s0 = out["never"]
    assert torch.allclose(u1, u0, rtol=0, atol=2e-4 * float(u0.abs().max()))
    assert torch.allclose(t1, t0, rtol=0, atol=2e-4 * float(t0.abs().max()))
    assert torch.allclose(r1, r0, rtol=1e-3, atol=1e-5)
    print(f"RBC2D-easy iterations velocity {v0:.1f} -> {v1:.1f}, scalar {s0:.1f} -> {s1:.1f}")
    assert v1 < v0 and s1 < s0


@pytest.mark.parametrize("dims,n", [(2, (64, 12)), (2, (256, 16)), (2, (512, 6)), (3, (128, 8, 6))])
def test_periodic_axis_is_transformed_by_a_real_fft(dims, n):
    """Uniform PERIODIC x axis of length 64..512: the eigenbasis is the real Fourier basis in FFT order and the device applies it
    as one FFT per row (fg_fdfft.hip, PERIODIC form) instead of the dense GEMM.  Pressure side: one PCG iteration with rA = const
    returns M^-1 r exactly -- compared with the NumPy application of the same factors, and with the GEMM path (FG_FD_NO_FFT=1)."""
    import os

    from fluidgym_amd.simulation.fd_precond import FDPreconditioner

    case = make_case(dims=dims, n=n, fixed_axes=(1,), B=2, seed=8, stretch=0.0)
    fd = FDPreconditioner(case.widths, case.fixed_faces)
    assert fd.x_fourier_width is not None and fd.transform_axes_periodic_uniform
    rng = np.random.default_rng(0)
    r = rng.standard_normal((case.B,) + case.shape).astype(np.float32)
    r -= r.mean(axis=tuple(range(1, r.ndim)), keepdims=True)
    rA = np.full_like(r, 0.5)
    got = {}
    for no_fft in ("0", "1"):
        os.environ["FG_FD_NO_FFT"] = no_fft
        try:
            ns = case.native()
        finally:
            os.environ.pop("FG_FD_NO_FFT", None)
        x = torch.zeros_like(torch.from_numpy(r)).cuda()
        info = ns.poisson_fdcg(torch.from_numpy(rA).cuda(), torch.from_numpy(r).cuda(), x, tol=1e-6)
        torch.cuda.synchronize()
        assert all(i.used_iterations <= 1 for i in info)
        got[no_fft] = _np(x)
        ns.close()
    for b in range(case.B):
        z = fd.apply(r[b].astype(np.float64)) / 0.5
        for k in got:
            assert rel_err(got[k][b] - got[k][b].mean(), z - z.mean()) < 2e-5, k
