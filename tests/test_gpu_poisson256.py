"""north_star's 256^3 Poisson kernels at FULL size (bench.py's ``poisson_micro`` configuration: B = 1, periodic x / z, walls in
y, wall-refined y), held through size-independent properties -- the oracle's sparse matrix at 16.7 M cells is beyond the few
seconds a test has; the same kernels meet the oracle matrix on small grids in test_gpu_parity.py
(test_zmarch_3d_kernels_match_brick_kernels_and_oracle, both ring forms).  At this size the launch geometry is the one the
benchmark times: z-marching kernels, the single-barrier ring for the sweep and the CG kernel (csrc/fg_poisson3d.hip)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _solver(n=256):
    from fluidgym_amd.native import NativeSolver
    from fluidgym_amd.simulation import grids

    hx = np.full(n, 1.0 / n, np.float32)
    hy = np.diff(grids.weights_exp(n, 1.02, "BOTH")).astype(np.float32)
    return NativeSolver([hx, hy, hx.copy()], 1, fixed_faces=(2, 3), device=torch.device("cuda", 0), allocate=False)


def test_operator_sweep_and_solve_at_256_cubed():
    n = 256
    ns = _solver(n)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    shape = (1, n, n, n)
    rA = 1.0 / (100.0 * (1.0 + 0.1 * torch.rand(shape, device=dev, generator=g)))
    u = torch.randn(shape, device=dev, generator=g)
    v = torch.randn(shape, device=dev, generator=g)
    Pu, Pv, P1 = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
    ns.poisson_apply(rA, u, Pu)
    ns.poisson_apply(rA, v, Pv)
    ns.poisson_apply(rA, torch.ones_like(u), P1)
    scale = float(Pu.abs().max())
    # the constant is in the null space (no prescribed-pressure face), the operator is symmetric and negative semi-definite
    assert float(P1.abs().max()) < 1e-5 * scale
    uPv, vPu = float((u.double() * Pv.double()).sum()), float((v.double() * Pu.double()).sum())
    assert abs(uPv - vPu) < 1e-5 * float(Pu.double().norm() * v.double().norm())
    assert float((u.double() * Pu.double()).sum()) < 0.0
    # linearity: P (2 u - 3 v) = 2 P u - 3 P v
    w = torch.empty_like(u)
    ns.poisson_apply(rA, 2.0 * u - 3.0 * v, w)
    assert float((w - (2.0 * Pu - 3.0 * Pv)).abs().max()) < 2e-5 * scale
    # a Jacobi sweep and a red-black sweep leave the solution of their own system where it is: b = P u
    for sweep in (ns.poisson_jacobi, ns.poisson_rbgs):
        x = u.clone()
        sweep(rA, Pu, x, 2, 0.8)
        assert float((x - u).abs().max()) < 2e-5 * float(u.abs().max()), sweep.__name__
    # one Jacobi sweep from zero is x = omega b / diag: with b = diag (diag = -P e summed stencil-wise is not available, so use
    # two right-hand sides) the update is linear in b
    xa, xb, xc = torch.zeros_like(u), torch.zeros_like(u), torch.zeros_like(u)
    ns.poisson_jacobi(rA, u, xa, 1, 0.8)
    ns.poisson_jacobi(rA, v, xb, 1, 0.8)
    ns.poisson_jacobi(rA, u + v, xc, 1, 0.8)
    assert float((xc - (xa + xb)).abs().max()) < 1e-5 * float(xc.abs().max())
    # CG: the solve meets its tolerance on the residual formed with the apply kernel, and every iteration count is plausible
    b = Pu - Pu.mean()                      # a right-hand side in the range of P: the solution is u up to a constant
    x = torch.zeros_like(u)
    tol = 1e-4 * float(b.double().pow(2).mean().sqrt())
    info = ns.poisson_cg(rA, b, x, tol=tol, max_iterations=8000)
    assert info[0].converged and 10 < info[0].used_iterations < 8000, (info[0].converged, info[0].used_iterations)
    r = torch.empty_like(u)
    ns.poisson_apply(rA, x, r)
    res = float((b - r).double().pow(2).mean().sqrt())
    assert res < 2.0 * tol, (res, tol)
    ns.close()
