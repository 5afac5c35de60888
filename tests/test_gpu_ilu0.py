"""ILU(0) of the stencil-form advection-diffusion matrix (csrc/fg_ilu0.hip) -- the reference's own preconditioner of the
BiCGStab rungs (cusparseScsrilu02 + two cusparseSpSV per application, bicgstab_solver_kernel.cu:191-226, 288-293;
preconditionBiCG / BiCG_precondition_fallback, PISOtorch_diff.py:449-476).  The kernel uses the closed form the factorisation takes
on a 5- / 7-point stencil (modified diagonal only) and sweeps hyperplanes; it is held against a GENERIC ILU(0) (IKJ on the
sparsity pattern, NumPy, float64) of the very matrix the GPU assembled."""
import numpy as np
import pytest
import torch

from oracle import piso_oracle as O
from tests.helpers import make_case, rel_err
from tests.test_gpu_linepre import _np, _wall_refined

pytestmark = pytest.mark.gpu


def _dense_from_stencil(A, off, shape, periodic):
    """Dense matrix of one env from the GPU's buffers: A [N], off [2d, N]; cells in natural order (x fastest)."""
    dims = len(shape)                     # shape = (ny, nx) or (nz, ny, nx)
    ext = list(shape[::-1])               # [nx, ny(, nz)]
    N = int(np.prod(ext))
    M = np.zeros((N, N))
    pattern = np.zeros((N, N), bool)
    idx = np.arange(N)
    pos = [idx % ext[0], (idx // ext[0]) % ext[1]] + ([idx // (ext[0] * ext[1])] if dims == 3 else [])
    M[idx, idx] = A
    pattern[idx, idx] = True
    stride = [1, ext[0], ext[0] * ext[1]]
    for f in range(2 * dims):
        ax, up = f >> 1, f & 1
        p = pos[ax] + (1 if up else -1)
        inside = (p >= 0) & (p < ext[ax])
        wrap = ~inside & periodic[ax]
        nb = np.where(inside, idx + (1 if up else -1) * stride[ax], np.where(up, idx - (ext[ax] - 1) * stride[ax], idx + (ext[ax] - 1) * stride[ax]))
        ok = inside | wrap
        M[idx[ok], nb[ok]] += off[f][ok]
        pattern[idx[ok], nb[ok]] = True
    return M, pattern


def _ilu0_generic(M, pattern):
    """IKJ incomplete LU without fill on `pattern` (Saad, Iterative Methods, alg. 10.4); returns unit-lower L and upper U."""
    n = M.shape[0]
    LU = M.copy()
    for i in range(1, n):
        for k in np.nonzero(pattern[i, :i])[0]:
            LU[i, k] /= LU[k, k]
            js = np.nonzero(pattern[i, k + 1:])[0] + k + 1
            LU[i, js] -= LU[i, k] * np.where(pattern[k, js], LU[k, js], 0.0)
    return np.tril(LU, -1) + np.eye(n), np.triu(LU)


@pytest.mark.parametrize("dims,n,fixed_axes", [(2, (12, 10), (1,)), (2, (9, 8), (0, 1)), (2, (8, 12), ()), (3, (6, 5, 4), (1,)),
                                               (3, (5, 4, 6), (0, 1, 2))])
def test_ilu0_application_is_the_generic_incomplete_factorisation(dims, n, fixed_axes):
    case = make_case(dims=dims, n=n, fixed_axes=fixed_axes, B=2, seed=5, nu=0.05, vel_scale=0.5)
    dt = 0.05
    ns = case.native()
    ns.setup_advection(dt)
    A = _np(ns.buffer(0, (case.B, -1)))
    off = _np(ns.buffer(1, (case.B, 2 * dims, -1)))
    g = torch.Generator(device="cpu").manual_seed(3)
    r = torch.randn((case.B, dims) + case.shape, generator=g)
    z = _np(ns.apply_advection_preconditioner(4, r)).reshape(case.B, dims, -1)
    ns.close()
    periodic = [a not in fixed_axes for a in range(dims)]
    for b in range(case.B):
        M, pattern = _dense_from_stencil(A[b], off[b], case.shape, periodic)
        L, U = _ilu0_generic(M, pattern)
        # the closed form the kernel relies on: ILU(0) leaves every off-diagonal of a stencil matrix as it is
        offd = ~np.eye(M.shape[0], dtype=bool)
        assert np.abs((np.triu(U, 1) - np.triu(M, 1)))[offd].max() <= 1e-12 * np.abs(M).max()
        for comp in range(dims):
            z_ref = np.linalg.solve(U, np.linalg.solve(L, _np(r[b, comp]).ravel()))
            assert rel_err(z[b, comp], z_ref) < 2e-5, (b, comp, rel_err(z[b, comp], z_ref))


def test_ilu0_preconditioned_solve_matches_the_direct_solve_in_fewer_iterations():
    case = _wall_refined(make_case(dims=2, n=(64, 48), fixed_axes=(1,), B=2, seed=4, nu=0.05, vel_scale=0.3), ratio=10.0)
    dt = 0.05
    out = {}
    for mode in (0, 4):
        ns = case.native()
        ns.set_advection_start(False)
        ns.set_advection_preconditioner(mode)
        ns.setup_advection(dt)
        info = ns.solve_advection(tol=1e-7)
        assert all(i.converged and i.is_finite for i in info), (mode, [i.final_residual for i in info])
        out[mode] = (_np(ns.buffer(3, (case.B, case.dims) + case.shape)), max(i.used_iterations for i in info) + 1)
        ns.close()
    g = case.grid()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)
        C, _, _ = O.build_advection_matrix(dom, dt)
        rhs = O.advection_rhs_velocity(dom, dt)
        for comp in range(2):
            x_ref = O.solve_direct(C, rhs[comp].ravel()).reshape(case.shape)
            for mode in (0, 4):
                assert rel_err(out[mode][0][b, comp], x_ref) < 3e-5, (mode, b, comp)
    print(f"ILU0 iterations: plain {out[0][1]}, ILU(0) {out[4][1]}")
    assert out[4][1] * 2 <= out[0][1], (out[0][1], out[4][1])


def test_ilu0_fallback_rung_rescues_a_system_the_plain_recurrence_cannot_solve():
    """Mode 5 = the reference's BiCG_precondition_fallback with the reference's preconditioner: the 60 : 1 wall-refined system of
    tests/test_gpu_linepre.py, on which the plain fp32 recurrence fails."""
    case = _wall_refined(make_case(dims=2, n=(64, 48), fixed_axes=(1,), B=2, seed=4, nu=0.05, vel_scale=0.3), ratio=60.0)
    dt = 0.05
    ns = case.native()
    ns.set_advection_start(False)
    ns.set_advection_preconditioner(0)
    ns.setup_advection(dt)
    assert not all(i.converged for i in ns.solve_advection(tol=1e-7, max_iterations=400))
    ns.set_advection_preconditioner(5)
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


def test_ilu0_refuses_axes_shorter_than_four_cells():
    case = make_case(dims=2, n=(3, 8), fixed_axes=(1,), B=1, seed=1)
    ns = case.native()
    from fluidgym_amd import _lib as L
    with pytest.raises(L.NativeLibraryError, match="four cells"):
        ns.set_advection_preconditioner(4)
    ns.close()


def test_policy_switches_the_envs_rungs_to_ilu0():
    """``set_solver_policy(advection_rung_preconditioner="ilu0")``: the env's BiCG_precondition_fallback rung becomes mode 5, and a
    simulation built with preconditionBiCG=True runs every advection-diffusion solve with ILU(0) (mode 4) -- same step as the plain
    recurrence to the solver tolerance, in fewer iterations."""
    import fluidgym_amd

    out = {}
    for pol in ("line", "ilu0"):
        old = fluidgym_amd.set_solver_policy(advection_rung_preconditioner=pol, advection_fd_preconditioner="never")
        try:
            env = fluidgym_amd.make("RBC2D-easy-v0", num_envs=2, n_heaters=4, resolution=8)
            env.reset(seed=3)
            assert env._sim.advection_preconditioner == (5 if pol == "ilu0" else 2)
            solver = env._domain.solver
            if pol == "ilu0":
                solver.set_advection_preconditioner(4)       # what preconditionBiCG=True selects under this policy
            solver.solver_counters(reset=True)
            obs, reward, _, _, info = env.step(torch.zeros_like(env.sample_action()))
            c = solver.solver_counters()
            out[pol] = (solver.velocity.clone(), reward.clone(), c["velocity"]["mean"], c["scalar"]["mean"])
            env.close()
        finally:
            fluidgym_amd.set_solver_policy(**old)
    u_l, r_l, v_l, s_l = out["line"]
    u_i, r_i, v_i, s_i = out["ilu0"]
    assert torch.allclose(u_i, u_l, rtol=0, atol=2e-4 * float(u_l.abs().max())) and torch.allclose(r_i, r_l, rtol=1e-3, atol=1e-5)
    assert v_i < v_l and s_i < s_l, (v_i, v_l, s_i, s_l)


# ---- multi-block path: level-scheduled ILU(0) on the mesh's neighbour table (csrc/fg_mb_step.hip: k_mb_ilu_factor / _solve) --------
def _mb_dense(A, off, nbr):
    N = A.shape[0]
    M = np.zeros((N, N)); pattern = np.eye(N, dtype=bool)
    M[np.arange(N), np.arange(N)] = A
    for f in range(nbr.shape[0]):
        ok = nbr[f] >= 0
        M[np.nonzero(ok)[0], nbr[f][ok]] += off[f][ok]
        pattern[np.nonzero(ok)[0], nbr[f][ok]] = True
    return M, pattern


@pytest.mark.parametrize("mesh", ["skewed_pair", "twisted_ring", "polar_ring", "skewed_pair_3d", "cylinder"])
def test_multi_block_ilu0_is_the_generic_incomplete_factorisation(mesh):
    """The preconditioner of the multi-block BiCG_precondition_fallback rung against a generic ILU(0) of the velocity matrix the
    GPU assembled, on connected blocks with shuffled axes, a ring of three blocks, a 3-D pair and the reference's five-block
    cylinder mesh.  (The kernel implements the general IKJ elimination -- off-diagonal updates included, which occur where two lower
    neighbours of a cell are neighbours of each other, i.e. at an interior vertex shared by three cells; none of these meshes has
    one: every block junction of the cylinder and airfoil meshes is four-valent or lies on a wall, so on them the factorisation
    only modifies the diagonal, which the count below records.)"""
    from fluidgym_amd import _lib as L
    import helpers_mb as H

    if mesh == "cylinder":
        from fluidgym_amd.envs.cylinder_grid import build_domain, make_vortex_street_mesh
        dom = build_domain(make_vortex_street_mesh(4), 0.01, batch=2)
    else:
        dom = getattr(H, mesh)().native(batch=2)
    g = torch.Generator(device="cpu").manual_seed(2)
    dom.velocity.copy_((0.3 * torch.randn(dom.velocity.shape, generator=g)).cuda())
    dom.velocity[:, 0] += 1.0
    dom.piso_step([0.05, 0.02], advection_tol=1e-6, pressure_tol=1e-5, raise_on_failure=False, max_iterations=400, pressure_use_bicgstab=True)
    B, d, N = dom.batch, dom.dims, dom.n_cells
    A = dom.buffer(L.FG_MB_BUF_A).view(B, N).cpu().numpy().astype(np.float64)
    off = dom.buffer(L.FG_MB_BUF_C_OFF).view(B, 2 * d, N).cpu().numpy().astype(np.float64)
    nbr = dom.neighbors()
    r = torch.randn(B, d, N, generator=g)
    z = dom.ilu_apply(r).cpu().numpy().astype(np.float64)
    updated = 0
    for b in range(B):
        M, pattern = _mb_dense(A[b], off[b], nbr)
        Lm, U = _ilu0_generic(M, pattern)
        updated += int((np.abs(np.triu(U, 1) - np.triu(M, 1)) > 1e-12 * np.abs(M).max()).sum())
        for comp in range(d):
            z_ref = np.linalg.solve(U, np.linalg.solve(Lm, r[b, comp].numpy().astype(np.float64)))
            assert rel_err(z[b, comp], z_ref) < 5e-5, (mesh, b, comp, rel_err(z[b, comp], z_ref))
    print(f"MB_ILU0 {mesh}: off-diagonals changed by the factorisation: {updated}")
    dom.close()
