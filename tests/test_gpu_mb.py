"""Parity of the HIP multi-block / non-orthogonal PISO path with the CPU oracle (oracle/mb_oracle.py), through the C ABI."""
import numpy as np
import pytest
import torch

from tests import helpers_mb as H

pytestmark = pytest.mark.gpu


def _state(d, seed, scale=0.2):
    rng = np.random.default_rng(seed)
    u = scale * rng.standard_normal((d.d, d.N))
    p = 0.1 * rng.standard_normal(d.N)
    return u, p - p.mean()


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-30))


@pytest.mark.parametrize("spec_fn", [H.split_rotated_channel, H.skewed_pair, H.twisted_ring, H.polar_ring, H.skewed_pair_3d])
def test_neighbor_table_matches_oracle_walk(spec_fn):
    spec = spec_fn()
    d = spec.oracle()
    dom = spec.native()
    nbr = dom.neighbors()
    for b, pos in d.cells():
        g = d.gidx(b, pos)
        for f in range(2 * d.d):
            if d.at_bound(b, pos, f) and d.is_empty(b, f):
                assert nbr[f, g] < 0
                blk = dom.blocks[b]
                assert -1 - nbr[f, g] == blk.boundary_slot0[f] + d.face_flat(b, f, pos)
            else:
                _, b2, p2, _ = d.resolve_neighbor(b, pos, f)[:4]
                assert nbr[f, g] == d.gidx(b2, p2)
    dom.close()


def _load(dom, states):
    for b, (u, p) in enumerate(states):
        dom.velocity[b] = torch.as_tensor(u, dtype=torch.float32)
        dom.pressure[b] = torch.as_tensor(p, dtype=torch.float32)


def _assembly_parity(dom, d, states, dt, B, check_div):
    from fluidgym_amd import _lib as L

    nd = d.d
    A = dom.buffer(L.FG_MB_BUF_A).view(B, -1).cpu().numpy()
    Coff = dom.buffer(L.FG_MB_BUF_C_OFF).view(B, 2 * nd, -1).cpu().numpy()
    rhs = dom.buffer(L.FG_MB_BUF_RHS).view(B, nd, -1).cpu().numpy()
    Pd = dom.buffer(L.FG_MB_BUF_P_DIAG).view(B, -1).cpu().numpy()
    Po = dom.buffer(L.FG_MB_BUF_P_OFF).view(B, 2 * nd, -1).cpu().numpy()
    h = dom.buffer(L.FG_MB_BUF_H).view(B, nd, -1).cpu().numpy()
    div = dom.buffer(L.FG_MB_BUF_DIV).view(B, -1).cpu().numpy()
    out = []
    for b in range(B):
        trace = {}
        u_ref, p_ref = d.piso_step(states[b][0], states[b][1], dt[b], trace=trace, corrector_steps=1 if check_div else 2)
        assert _rel(A[b], trace["C"][0]) < 2e-5
        assert _rel(Coff[b], trace["C"][1]) < 2e-5
        assert _rel(rhs[b], trace["rhs"]) < 5e-5
        assert _rel(Pd[b], trace["P"][0]) < 5e-5
        assert _rel(Po[b], trace["P"][1]) < 5e-5
        if check_div:  # single corrector: h and the pressure right-hand side are those of corrector 0
            assert _rel(h[b], trace["h"]) < 1e-4
            assert _rel(div[b], trace["prhs"]) < 2e-4
        out.append((u_ref, p_ref))
    return out


@pytest.mark.parametrize("spec_fn,bicg,ptol,project", [(H.split_rotated_channel, False, 2e-6, False), (H.polar_ring, False, 2e-6, False),
                                                      (H.skewed_pair, True, 1e-6, False), (H.skewed_pair_3d, True, 3e-7, False),
                                                      # mean projection is an exact no-op on orthogonal meshes, for CG and BiCGStab
                                                      (H.polar_ring, False, 2e-6, True), (H.polar_ring, True, 2e-6, True),
                                                      (H.split_rotated_channel, True, 2e-6, True),
                                                      (H.odd_channel, False, 2e-6, False), (H.odd_channel, False, 2e-6, True),
                                                      # BiCGStab with fp64 iterative refinement (pressure_use_bicgstab = 2): tighter
                                                      # tolerance than fp32 residuals allow, on meshes where CG stalls
                                                      # (unprojected like the oracle's direct solve: on these meshes the constant is
                                                      # not a null vector of the matrix, so the mean projection the envs use changes
                                                      # the system)
                                                      (H.skewed_pair, 2, 2e-7, False), (H.skewed_pair_3d, 2, 2e-7, False),
                                                      (H.polar_ring, 2, 2e-7, True),
                                                      # the same steps with the solves driven to 2e-8 (fp64-refined iterate): what is left is
                                                      # the fp32 assembly and the fp32 velocity solve -- the north-star's 1e-5 per step
                                                      (H.split_rotated_channel, 2, 2e-8, True), (H.odd_channel, 2, 2e-8, True),
                                                      (H.polar_ring, 2, 2e-8, True), (H.skewed_pair, 2, 2e-8, False)])
def test_piso_step_matches_oracle(spec_fn, bicg, ptol, project):
    """Whole step against the oracle's direct solves.  The pressure solver is CG as in the reference where the mesh is
    orthogonal (symmetric matrix); with strong cross metrics the matrix is not symmetric, CG stalls (there as here, see
    test_cg_on_a_skewed_mesh_returns_its_best_iterate) and the same system is solved with BiCGStab."""
    spec = spec_fn()
    d = spec.oracle()
    B = 2
    dom = spec.native(batch=B)
    dt = [0.05, 0.03]
    states = [_state(d, 10 + b) for b in range(B)]
    _load(dom, states)
    tight = ptol < 1e-7
    its = dom.piso_step(dt, advection_tol=3e-8 if tight else 1e-7, pressure_tol=ptol, pressure_use_bicgstab=bicg, pressure_project_mean=project)
    # (the refined solver verifies convergence on the true fp64 residual: after such a first projection the second corrector's
    # right-hand side can already meet the tolerance -- 0 iterations)
    assert its[0] > 0 and its[1] > 0 and (its[2] > 0 or bicg == 2)
    u_gpu = dom.velocity.cpu().numpy()
    p_gpu = dom.pressure.cpu().numpy()
    refs = _assembly_parity(dom, d, states, dt, B, check_div=False)
    for b in range(B):
        print(f"MB_STEP_ERR {spec_fn.__name__} bicg={bicg} ptol={ptol:g} project={project} env {b}: velocity {_rel(u_gpu[b], refs[b][0]):.2e} "
              f"pressure {_rel(p_gpu[b], refs[b][1]):.2e}")
        # measured (round 3, MB_STEP_ERR lines of the GPU log): 0.9-13e-5 velocity / 0.3-22e-5 pressure at the envs' tolerances --
        # the absolute pressure tolerance of 2e-6 .. 2e-7 is what is left; bounds = 2x the largest seen
        # with the solves driven to 2e-8: 0.6-2.8e-6 velocity / 0.25-4.6e-6 pressure -- inside the north-star's 1e-5 per step
        assert _rel(u_gpu[b], refs[b][0]) < (1e-5 if tight else 2.5e-4), (spec_fn.__name__, b)
        assert _rel(p_gpu[b], refs[b][1]) < (2e-5 if tight else 5e-4), (spec_fn.__name__, b)
    mv = dom.max_velocity()
    for b in range(B):
        assert np.isclose(mv[b], d.max_cfl_velocity(refs[b][0]), rtol=1e-3)
    dom.close()


@pytest.mark.parametrize("spec_fn", [H.skewed_pair, H.twisted_ring, H.cylinder_3d_small, H.airfoil_spec])
def test_assembly_matches_oracle_on_strongly_skewed_meshes(spec_fn):
    """Matrices, right-hand sides, predictor, h and the pressure right-hand side with its lagged corner terms, on meshes
    where every cross-metric branch is active (walls with moving Dirichlet values, connections with shuffled axes, a
    cycle of connections).  Independent of how well the pressure solve converges."""
    from fluidgym_amd import _lib as L

    spec = spec_fn()
    d = spec.oracle()
    B = 2
    dom = spec.native(batch=B)
    # (the airfoil C-mesh -- airfoil/grid.py:629-707, cells down to 1e-4 of the typical area at the nose -- takes a time step
    # of the size its env runs with; PISO_multiblock_cuda_kernel.cu:3616-3880, 4812-4978 are what is being compared)
    dt = [2e-3, 1e-3] if spec_fn is H.airfoil_spec else [0.05, 0.03]
    states = [_state(d, 20 + b) for b in range(B)]
    _load(dom, states)
    dom.piso_step(dt, corrector_steps=1, advection_tol=1e-7, pressure_tol=1e-5, raise_on_failure=False, max_iterations=600)
    _assembly_parity(dom, d, states, dt, B, check_div=True)
    dom.close()


AIRFOIL_U_BOUND, AIRFOIL_P_BOUND = 2e-4, 2e-4     # measured 3.9e-5 / 4.8e-5 (the AIRFOIL_STEP_ERR line of the GPU log)


def _projected_field_errors(d, trace, p_gpu, u_gpu, res_gpu, A, label):
    """Velocity and pressure of a whole step against the oracle's with the ill-determined directions of the oracle's pressure matrix
    removed from the pressure DIFFERENCE (see test_airfoil_mesh_step_matches_the_oracle); prints the raw figures too."""
    P, prhs, u_ref = trace["P"], trace["prhs"], trace["u_new"]
    err_u = _rel(u_gpu, u_ref)
    err_u_l2 = float(np.sqrt(np.mean((u_gpu - u_ref) ** 2)) / np.sqrt(np.mean(u_ref ** 2)))
    U, S, Vt = np.linalg.svd(d.dense(P))
    nz = S > 1e-13 * S[0]                                     # (an exactly singular matrix: its null space takes no part)
    p_star = Vt[nz].T @ ((U[:, nz].T @ prhs) / S[nz])
    sig_cut = max(res_gpu, 1e-9) * np.sqrt(len(S)) / (1e-2 * np.abs(p_star - p_star.mean()).max())
    keep = S >= sig_cut
    proj = lambda x: Vt[keep].T @ (Vt[keep] @ x)
    delta = p_gpu - p_star
    err_p = float(np.abs(proj(delta)).max() / np.abs(proj(p_star)).max())
    u_fix = d.correct_velocity(trace["h"], p_gpu - (delta - proj(delta)), A)
    err_u_fix = _rel(u_fix, u_ref)
    print(f"{label} velocity max-norm {err_u:.2e} rms {err_u_l2:.2e}; with the pressure difference along the {int((~keep).sum())} of {len(S)} "
          f"directions below sigma {sig_cut:.2e} (sigma_max {S[0]:.2e}, sigma_min {S[-1]:.2e}) removed: velocity {err_u_fix:.2e}, pressure {err_p:.2e}; "
          f"max|p*| {np.abs(p_star - p_star.mean()).max():.2e}, GPU residual {res_gpu:.2e}")
    return err_u, err_u_fix, err_p


def test_cylinder_mesh_step_fields_match_the_oracle():
    """The same field comparison on the reference's cylinder mesh at resolution 8 (five blocks, a cycle of connections around the
    cylinder): one corrector, refined BiCGStab at 1e-7 -- here the raw velocity must agree as well (the mesh has no 1e-4-area cells)."""
    spec = H.cylinder_2d(8)
    d = spec.oracle()
    assert abs(sum(H.face_fluxes(d).values())) < 1e-10
    dom = spec.native(batch=1)
    u0, p0 = _state(d, 33, scale=0.05)
    u0[0] += 1.0
    st = [(u0, p0)]
    _load(dom, st)
    its = dom.piso_step([5e-3], corrector_steps=1, advection_tol=1e-7, pressure_tol=1e-7, pressure_use_bicgstab=2,
                        pressure_project_mean=True, max_iterations=3000, raise_on_failure=False)
    assert (dom.env_status() <= 1).all()
    trace = {}
    trace["u_new"], _ = d.piso_step(st[0][0], st[0][1], 5e-3, trace=trace, corrector_steps=1)
    p_gpu = dom.pressure[0].cpu().numpy().astype(np.float64)
    u_gpu = dom.velocity[0].cpu().numpy().astype(np.float64)
    res_gpu = float(np.sqrt(np.mean((d.apply(trace["P"], p_gpu) - trace["prhs"]) ** 2)))
    err_u, err_u_fix, err_p = _projected_field_errors(d, trace, p_gpu, u_gpu, res_gpu, trace["C"][0], f"CYLINDER8_STEP_ERR (iterations {its})")
    assert err_u_fix < 2e-4 and err_p < 2e-4 and err_u < CYLINDER8_RAW_U_BOUND
    dom.close()


CYLINDER8_RAW_U_BOUND = 5e-2     # measured 1.3e-2 (the refined BiCGStab ends at a residual of 3.5e-5 on this mesh after 383 iterations)



def test_airfoil_mesh_step_matches_the_oracle():
    """One whole PISO step (one corrector) on the airfoil C-mesh (resolution_div 4) against the oracle, with the solver the airfoil
    env runs (BiCGStab with fp64 refinement, tolerance 1e-7, airfoil_env_base.py:272).  On this mesh the pressure system is
    singular AND slightly inconsistent (its left null vector is not the constant: cells of 1e-4 of the typical area at the nose),
    so an iterative solver -- the reference's as well as this one -- ends on its best iterate at a residual floor while the
    oracle's least-squares solve returns the minimiser; the two pressures differ along near-null directions.  What must agree:
    the predictor (velocity solve), the pressure right-hand side, the residual the returned pressure leaves in the ORACLE's matrix
    (within a small factor of the least-squares floor), and the velocity correction applied to that pressure."""
    spec = H.airfoil_spec(noise=0.02, balanced=True)
    d = spec.oracle()
    assert abs(sum(H.face_fluxes(d).values())) < 1e-12
    dom = spec.native(batch=1)
    u0, p0 = _state(d, 31, scale=0.05)
    u0[0] += 0.3                                  # the env's inflow speed (airfoil_env_base.py) + a perturbation
    st = [(u0, p0)]
    _load(dom, st)
    its = dom.piso_step([1e-3], corrector_steps=1, advection_tol=1e-7, pressure_tol=1e-7, pressure_use_bicgstab=2,
                        pressure_project_mean=True, max_iterations=3000, raise_on_failure=False)
    status = dom.env_status()
    assert (status <= 1).all()           # (1: a solve ended on its best iterate above 1e-7; 2 would be non-finite)
    trace = {}
    trace["u_new"], _ = d.piso_step(st[0][0], st[0][1], 1e-3, trace=trace, corrector_steps=1)
    p_gpu = dom.pressure[0].cpu().numpy().astype(np.float64)
    u_gpu = dom.velocity[0].cpu().numpy().astype(np.float64)
    P, prhs, A = trace["P"], trace["prhs"], trace["C"][0]
    res_gpu = float(np.sqrt(np.mean((d.apply(P, p_gpu) - prhs) ** 2)))
    res_ls = float(np.sqrt(np.mean((d.apply(P, trace["p0"]) - prhs) ** 2)))
    print(f"airfoil mesh step: iterations {its}, env status {status}, pressure residual in the oracle's matrix {res_gpu:.3e} "
          f"(least-squares floor {res_ls:.3e}, rms of the right-hand side {float(np.sqrt(np.mean(prhs ** 2))):.3e})")
    assert res_gpu < 10.0 * res_ls + 2e-7
    u_chk = d.correct_velocity(trace["h"], p_gpu - p_gpu.mean(), A)
    assert _rel(u_gpu, u_chk) < 2e-4
    # ... and the fields themselves: the velocity against the oracle's, and the pressures with the numerically
    # null directions of the ORACLE's matrix removed from both -- the right singular vectors whose singular value is below
    # 1e-3 of the largest: along those an iterative solve and a least-squares solve are free to differ (a residual difference e
    # moves the pressure by e / sigma), along all others a wrong pressure shows
    u_ref = trace["u_new"]
    err_u = _rel(u_gpu, u_ref)
    err_u_l2 = float(np.sqrt(np.mean((u_gpu - u_ref) ** 2)) / np.sqrt(np.mean(u_ref ** 2)))
    worst = int(np.abs(u_gpu - u_ref).max(axis=0).argmax())
    D = d.dense(P)
    U, S, Vt = np.linalg.svd(D)
    # The matrix is not exactly singular on this mesh (sigma_min / sigma_max ~ 3e-6, the constant is not its null vector), so the
    # system has ONE solution p* = V S^-1 U^T b, and a residual e moves the pressure by (u_i . e) / sigma_i along singular direction
    # i.  The GPU's iterate leaves res_gpu; the directions along which that residual level cannot move the pressure by more than
    # 1 % of max|p*| are the ones a comparison can hold -- the others (the near-constant mode and the modes living on the 1e-4-area
    # cells at the nose) are removed from the pressure DIFFERENCE, in the pressure and in the velocity correction it drives.
    p_star = Vt.T @ ((U.T @ prhs) / S)
    sig_cut = res_gpu * np.sqrt(len(S)) / (1e-2 * np.abs(p_star - p_star.mean()).max())
    keep = S >= sig_cut
    proj = lambda x: Vt[keep].T @ (Vt[keep] @ x)
    delta = p_gpu - p_star
    err_p = float(np.abs(proj(delta)).max() / np.abs(proj(p_star)).max())
    u_fix = d.correct_velocity(trace["h"], p_gpu - (delta - proj(delta)), A)
    err_u_fix = _rel(u_fix, u_ref)
    print(f"AIRFOIL_STEP_ERR velocity max-norm {err_u:.2e} (worst cell {worst}) rms {err_u_l2:.2e}; with the pressure difference along the "
          f"{int((~keep).sum())} of {len(S)} directions below sigma {sig_cut:.2e} (sigma_max {S[0]:.2e}, sigma_min {S[-1]:.2e}) removed: velocity {err_u_fix:.2e}, "
          f"pressure {err_p:.2e}; max|p*| {np.abs(p_star - p_star.mean()).max():.2e}, GPU residual {res_gpu:.2e}")
    assert err_u_fix < AIRFOIL_U_BOUND and err_p < AIRFOIL_P_BOUND
    dom.close()


def test_cg_on_a_skewed_mesh_returns_its_best_iterate():
    """pressure_return_best_result=True (cylinder_env_base.py:318): a CG solve that stalls on the non-symmetric matrix ends
    with the best iterate it saw, not with whatever the recurrence drifted to."""
    spec = H.twisted_ring()
    d = spec.oracle()
    dom = spec.native(batch=1)
    st = [_state(d, 3)]
    _load(dom, st)
    with pytest.raises(Exception, match="status -5"):
        dom.piso_step(0.05, pressure_tol=1e-7, max_iterations=3000)
    assert torch.isfinite(dom.velocity).all() and torch.isfinite(dom.pressure).all()
    u_ref, _ = d.piso_step(st[0][0], st[0][1], 0.05)
    assert float(dom.velocity.abs().max()) < 3.0 * np.abs(u_ref).max()
    dom.close()


def test_first_velocity_solve_starts_from_zero_unless_asked():
    """The reference's non-orthogonal branch hands the first velocity solve of a step x=None (PISOtorch_simulation.py:1735-1742,
    recorded in tests/golden/reference_split_step.json): the default of a handle.  fg_mb_set_advection_start(1) starts it from the
    current velocity: fewer iterations, the same answer within the tolerance."""
    spec = H.skewed_pair()
    d = spec.oracle()
    dom = spec.native(batch=2)
    st = [_state(d, 3 + b) for b in range(2)]
    kw = dict(advection_tol=1e-6, pressure_tol=2e-6, pressure_use_bicgstab=True)
    _load(dom, st)
    cold = dom.piso_step(2e-3, **kw)
    u_cold = dom.velocity.cpu().numpy()
    dom.set_advection_start(True)
    _load(dom, st)
    warm = dom.piso_step(2e-3, **kw)
    u_warm = dom.velocity.cpu().numpy()
    assert 0 < warm[0] < cold[0], (warm, cold)      # |rhs| ~ |u| / dt against |C u - rhs| ~ dt-independent terms
    assert _rel(u_warm, u_cold) < 1e-4
    dom.set_advection_start(False)
    _load(dom, st)
    again = dom.piso_step(2e-3, **kw)
    assert again[0] == cold[0] and again[0] > warm[0]
    assert np.array_equal(dom.velocity.cpu().numpy(), u_cold)        # the same solve twice: the same bits (order-independent reductions)
    dom.close()


def test_inactive_env_is_untouched_and_batch_is_independent():
    spec = H.skewed_pair()
    d = spec.oracle()
    dom = spec.native(batch=3)
    st = [_state(d, 5 + b) for b in range(3)]
    for b in range(3):
        dom.velocity[b] = torch.as_tensor(st[b][0], dtype=torch.float32)
        dom.pressure[b] = torch.as_tensor(st[b][1], dtype=torch.float32)
    before = dom.velocity.clone()
    dom.piso_step([0.04, 0.0, 0.04], advection_tol=1e-7, pressure_tol=2e-6, pressure_use_bicgstab=True)
    assert torch.equal(dom.velocity[1], before[1])
    solo = spec.native(batch=1)
    solo.velocity[0] = torch.as_tensor(st[2][0], dtype=torch.float32)
    solo.pressure[0] = torch.as_tensor(st[2][1], dtype=torch.float32)
    solo.piso_step(0.04, advection_tol=1e-7, pressure_tol=2e-6, pressure_use_bicgstab=True)
    assert _rel(dom.velocity[2].cpu().numpy(), solo.velocity[0].cpu().numpy()) < 1e-5
    dom.close(); solo.close()


def test_reference_quirks_can_be_switched_off():
    """Without the reference's first-layer rule the predictor keeps a uniform stream on a skewed mesh (see
    tests/test_mb_oracle.py::test_flux_balance_of_a_uniform_stream_over_connections)."""
    from fluidgym_amd import _lib as L

    spec = H.skewed_pair(wobble=0.0, stretch=1.0)  # uniform parallelograms: cell and boundary-face metrics coincide
    uc = np.array([0.6, -0.25])
    spec.fixed = []
    o = spec.oracle()
    for b, blk in enumerate(o.blocks):
        for f in range(4):
            if blk.bounds[f].type == "fixed":
                spec.fixed.append((b, f, np.repeat(uc[:, None], blk.bounds[f].velocity.shape[1], axis=1)))
    dom = spec.native(batch=1, reference_quirks=False)
    dom.velocity[0] = torch.as_tensor(np.repeat(uc[:, None], dom.n_cells, axis=1), dtype=torch.float32)
    dom.piso_step(0.05, corrector_steps=0, advection_tol=1e-8)
    ustar = dom.buffer(L.FG_MB_BUF_VELOCITY_RESULT).view(2, -1).cpu().numpy()
    assert np.abs(ustar - uc[:, None]).max() < 2e-5
    dom.close()


@pytest.mark.parametrize("spec_fn", [H.skewed_pair, H.twisted_ring, H.skewed_pair_3d])
def test_all_cross_terms_on_the_right_hand_side_mode(spec_fn):
    """nonOrthoFlags = DIRECT_RHS | DIAGONAL_RHS (10): nothing but the orthogonal Laplacian in the matrices.  The pressure
    matrix is then symmetric with the exact constant null space, and plain CG -- the reference's solver -- converges on the
    strongly skewed meshes where it stalls with the cross terms in the matrix."""
    spec = spec_fn()
    d = spec.oracle()
    B = 2
    dom = spec.native(batch=B, non_ortho_flags=10)
    dt = [0.05, 0.03]
    states = [_state(d, 30 + b) for b in range(B)]
    _load(dom, states)
    # the lagged terms are not a discrete divergence: the right-hand side is compatible only up to its mean, which the
    # solve projects out (what the oracle's least-squares solution does, too)
    # (5e-6: with |p| ~ 10 on these coarse skewed meshes 1e-6 is at the fp32 floor of the residual)
    its = dom.piso_step(dt, advection_tol=1e-7, pressure_tol=5e-6, pressure_project_mean=True)
    assert all(0 < i < 100 for i in its)
    u_gpu, p_gpu = dom.velocity.cpu().numpy(), dom.pressure.cpu().numpy()
    for b in range(B):
        u_ref, p_ref = d.piso_step(states[b][0], states[b][1], dt[b], flags=10)
        assert _rel(u_gpu[b], u_ref) < 1e-3, (spec_fn.__name__, b)
        assert _rel(p_gpu[b], p_ref) < 5e-3, (spec_fn.__name__, b)
    dom.close()


def test_non_finite_solve_drops_only_that_env_and_leaves_its_state_intact():
    """A non-finite velocity solve in ONE env of a batch (here forced through a NaN in its velocity source): the reference
    returns solve_ok=False before CopyVelocityResultToBlocks (PISOtorch_simulation.py:1752-1757) and Simulation.single_step
    returns False with the state intact (simulation.py:259-280).  Batched: that env keeps its pre-step state and is
    reported with status 2, the other env completes exactly as it does alone; nothing raises."""
    from fluidgym_amd.simulation.multiblock import MultiBlockSimulation

    spec = H.split_rotated_channel()
    d = spec.oracle()
    u0, p0 = _state(d, 3)
    u1, p1 = _state(d, 4)
    # env 0 alone
    ref = spec.native(batch=1)
    _load(ref, [(u0, p0)])
    sim_ref = MultiBlockSimulation(ref, dt=0.05, substeps=2, pressure_tol=1e-6, advection_tol=1e-6)
    assert sim_ref.single_step()
    u_ref = ref.velocity[0].cpu().numpy().copy()
    ref.close()
    dom = spec.native(batch=2)
    _load(dom, [(u0, p0), (u1, p1)])
    src = torch.zeros(2, d.d, d.N, device=dom.device)
    src[1, 0, d.N // 2] = float("nan")
    dom.set_velocity_source(src)
    before = dom.velocity[1].cpu().numpy().copy()
    p_before = dom.pressure[1].cpu().numpy().copy()
    sim = MultiBlockSimulation(dom, dt=0.05, substeps=2, pressure_tol=1e-6, advection_tol=1e-6)
    ok = sim.single_step()
    assert ok is False
    assert sim.last_env_status.tolist() == [0, 2]
    after = dom.velocity.cpu().numpy()
    assert np.isfinite(after).all()
    assert np.array_equal(after[1], before)              # not committed
    assert np.array_equal(dom.pressure[1].cpu().numpy(), p_before)   # nor is its pressure (restored from the start-of-step copy)
    assert _rel(after[0], u_ref) < 1e-5                   # the healthy env is not disturbed by its neighbour in the batch
    dom.close()


def test_onchip_cg_matches_the_chunked_cg(monkeypatch):
    """The one-workgroup-per-env CG (k_mbc_onchip) runs the same recurrence as the two-kernel chunked CG; on the reference's
    cylinder mesh (14 232 cells, 16 cells per thread) both reach the tolerance in the same number of iterations (the dot
    products are summed in a different order) and the steps they produce agree to solver tolerance."""
    from fluidgym_amd.envs.cylinder_grid import build_domain, make_vortex_street_mesh

    mesh = make_vortex_street_mesh(24)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("FG_MB_ONCHIP", mode)      # read once per handle at fg_mb_create
        dom = build_domain(mesh, 0.01, batch=3)
        g = torch.Generator(device="cpu").manual_seed(5)
        dom.velocity.copy_((0.3 * torch.randn(dom.velocity.shape, generator=g)).to(dom.device))
        dom.velocity[:, 0] += 1.0
        dom.make_divergence_free(pressure_tol=1e-6, pressure_project_mean=True)
        idle = dom.velocity[2].cpu().numpy().copy()
        its = []
        for _ in range(3):
            its.append(dom.piso_step([0.01, 0.02, 0.0], pressure_tol=1e-6, advection_tol=1e-6, pressure_project_mean=True,
                                     raise_on_failure=False))
        assert np.array_equal(dom.velocity[2].cpu().numpy(), idle)   # dt = 0: that env is left alone
        out[mode] = (dom.velocity.cpu().numpy().copy(), dom.pressure.cpu().numpy().copy(), its)
        dom.close()
    u0, p0, it0 = out["0"]
    u1, p1, it1 = out["1"]
    assert np.isfinite(u1).all() and np.isfinite(p1).all()
    assert _rel(u1[:2], u0[:2]) < 2e-4 and _rel(p1[:2], p0[:2]) < 2e-3
    for a, b in zip(it0, it1):
        for k in (1, 2):                                       # pressure solves: same iteration counts within a few per cent
            assert abs(a[k] - b[k]) <= max(3, 0.1 * a[k]), (it0, it1)
    assert max(max(i) for i in it1) > 20                       # the solves did iterate


@pytest.mark.parametrize("rungs", ["fp64", "preconditioned", "fp64_then_preconditioned", "pressure_fp64"])
def test_retry_ladder_rungs_reproduce_the_plain_solve(rungs):
    """The reference's retry ladder (_linear_solve, PISOtorch_diff.py:410-476): a failed fp32 solve is repeated in fp64
    (solver_double_fallback), a failed BiCGStab solve with a preconditioner (BiCG_precondition_fallback).  The first attempt is
    forced to count as failed (fg_mb_ladder's test mask); each rung, starting from zero, must land on the same step as the
    plain solve and the oracle (skewed two-block mesh, 2 envs)."""
    spec = H.skewed_pair()
    d = spec.oracle()
    states = [_state(d, 11), _state(d, 12)]
    dt = [0.02, 0.03]
    kw = dict(advection_tol=1e-7, pressure_tol=2e-7, pressure_use_bicgstab=True, raise_on_failure=False)
    plain = spec.native(batch=2)
    _load(plain, states)
    plain.piso_step(dt, **kw)
    u_plain = plain.velocity.cpu().numpy().copy()
    assert plain.ladder() == {"velocity_fp64": 0, "velocity_preconditioned": 0, "pressure_fp64": 0, "pressure_cg": 0}
    plain.close()
    dom = spec.native(batch=2)
    _load(dom, states)
    # which rungs the REFERENCE tries for the same scripted outcomes (tests/golden/reference_control.json, produced by its own
    # _linear_solve_wrapper): the advection solve runs without returnBestResult ("unconverged" fails it), the pressure solve
    # with it (only a non-finite residual does)
    from tests.test_control_golden import ladder_attempts
    force, opts, kind, outcomes = {
        "fp64": (1, dict(solver_double_fallback=True), "velocity", ["unconverged", "converged"]),
        "preconditioned": (1, dict(bicg_precondition_fallback=True), "velocity", ["unconverged", "converged"]),
        "fp64_then_preconditioned": (1 | 4, dict(solver_double_fallback=True, bicg_precondition_fallback=True), "velocity",
                                     ["unconverged", "unconverged", "converged"]),
        "pressure_fp64": (2, dict(solver_double_fallback=True), "pressure", ["non_finite", "converged"]),
    }[rungs]
    ref = ladder_attempts(True, kind == "pressure", bool(opts.get("solver_double_fallback")), bool(opts.get("bicg_precondition_fallback")), outcomes)
    expect = tuple(f"{kind}_fp64" if a["dtype"] == "float64" else f"{kind}_preconditioned" for a in ref["attempts"][1:])
    assert expect, ref
    dom.ladder(force_mask=force)
    dom.piso_step(dt, **dict(kw, **opts))
    used = dom.ladder(force_mask=0)
    for k, v in used.items():
        assert (v > 0) == (k in expect), used
    u = dom.velocity.cpu().numpy()
    assert np.isfinite(u).all()
    for b in range(2):
        # (the plain fp32 recurrence declares convergence on its recurrence residual, whose gap to the true residual depends on
        # the trajectory -- 1e-5 .. 3e-4 against the oracle at this tolerance; the rungs are held to the oracle bound)
        assert _rel(u[b], u_plain[b]) < 5e-4, rungs
        u_ref, _ = d.piso_step(states[b][0], states[b][1], dt[b])
        assert _rel(u[b], u_ref) < 2e-4, rungs
    dom.close()


def test_multilevel_preconditioned_onchip_cg_reaches_the_same_answer_in_fewer_iterations():
    """k_mbc_onchip with the additive multilevel preconditioner (fg_mb_set_multilevel) on the reference's cylinder mesh.  The
    pressure matrix of this mesh is nearly singular beyond its constant mode, so at a residual tolerance of 1e-6
    two Krylov trajectories end on velocities that differ by more than rounding; both are measured against the same projection
    solved two orders tighter (plain recurrence): the preconditioned solve must be as close to it as the plain one is, in at
    most 45 % of the iterations (the NumPy replay in tests/test_multilevel_precond.py measures 63 against 193)."""
    from fluidgym_amd.envs.cylinder_grid import build_domain, make_vortex_street_mesh

    mesh = make_vortex_street_mesh(24)
    out = {}
    for mode, tol in (("truth", 1e-8), ("plain", 1e-6), ("multilevel", 1e-6)):
        dom = build_domain(mesh, 0.01, batch=2)
        dom.set_stall_limit(5000)
        if mode == "multilevel":
            assert dom.set_pressure_multilevel() == {"n4": 912, "n8": 228}
        g = torch.Generator(device="cpu").manual_seed(5)
        dom.velocity.copy_((0.3 * torch.randn(dom.velocity.shape, generator=g)).to(dom.device))
        dom.velocity[:, 0] += 1.0
        dom.solver_counters(reset=True)
        dom.make_divergence_free(pressure_tol=tol, max_iterations=5000, pressure_project_mean=True)
        dom.piso_step([0.01, 0.02], pressure_tol=tol, advection_tol=1e-7, pressure_project_mean=True, raise_on_failure=False)
        out[mode] = (dom.velocity.cpu().numpy().copy(), dom.solver_counters())
        dom.close()
    u_t, u_p, u_m = out["truth"][0], out["plain"][0], out["multilevel"][0]
    c_p, c_m = out["plain"][1], out["multilevel"][1]
    assert np.isfinite(u_m).all()
    err_p, err_m = _rel(u_p, u_t), _rel(u_m, u_t)
    assert err_m <= max(2.0 * err_p, 2e-4), (err_p, err_m)
    assert c_m["pressure0"]["mean"] <= 0.45 * c_p["pressure0"]["mean"], (c_p, c_m)
    assert c_m["pressure0"]["mean"] > 3       # it did iterate


def test_aggregate_owned_onchip_cg_is_the_cell_ordered_one(monkeypatch):
    """The aggregate-owned layout of the preconditioned on-chip CG (k_mbc_onchip<AGG>: thread = one 4 x 4 aggregate, restriction and
    prolongation in registers, matrix in slot order, stencil loads batched) against the cell-ordered kernel (FG_MB_OC_AGG=0) on the
    reference's cylinder mesh: the same recurrence with another summation order of the dot products, so the iteration counts of
    every solve may differ by one or two and the projected velocities agree to what two Krylov trajectories at that tolerance
    do.  A mesh the layout does not fit (an aggregate table with more than 256 8 x 8 aggregates) keeps the cell-ordered kernel:
    covered by the other multilevel tests, which run meshes of both kinds."""
    from fluidgym_amd.envs.cylinder_grid import build_domain, make_vortex_street_mesh

    mesh = make_vortex_street_mesh(24)
    out = {}
    for agg in ("0", "1"):
        monkeypatch.setenv("FG_MB_OC_AGG", agg)
        dom = build_domain(mesh, 0.01, batch=3)
        dom.set_stall_limit(5000)
        assert dom.set_pressure_multilevel() == {"n4": 912, "n8": 228}
        g = torch.Generator(device="cpu").manual_seed(11)
        dom.velocity.copy_((0.3 * torch.randn(dom.velocity.shape, generator=g)).to(dom.device))
        dom.velocity[:, 0] += 1.0
        dom.solver_counters(reset=True)
        dom.make_divergence_free(pressure_tol=1e-7, max_iterations=5000, pressure_project_mean=True)
        u0 = dom.velocity.cpu().numpy().copy()
        for _ in range(3):
            dom.piso_step([0.01, 0.02, 0.0], pressure_tol=1e-7, advection_tol=1e-7, pressure_project_mean=True)   # env 2 inactive
        # solves started from an iterate (the second non-orthogonal pass of a corrector, and the warm-start policy): the kernel's
        # residual pass r = rhs - P x0 through the slot-ordered matrix
        dom.piso_step([0.01, 0.02, 0.0], pressure_tol=1e-7, advection_tol=1e-7, pressure_project_mean=True, pressure_non_ortho_steps=2,
                      pressure_warm_start=True)
        out[agg] = (u0, dom.velocity.cpu().numpy().copy(), dom.pressure.cpu().numpy().copy(), dom.solver_counters())
        dom.close()
    (u0_c, u_c, p_c, c_c), (u0_a, u_a, p_a, c_a) = out["0"], out["1"]
    assert np.isfinite(u_a).all() and np.isfinite(p_a).all()
    assert _rel(u0_a, u0_c) < 2e-5 and _rel(u_a, u_c) < 5e-5, (_rel(u0_a, u0_c), _rel(u_a, u_c))
    for k in ("pressure0", "pressure1"):
        assert abs(c_a[k]["mean"] - c_c[k]["mean"]) <= 2.0 and c_a[k]["unconverged"] == 0, (c_c, c_a)
    np.testing.assert_array_equal(u_a[2], u0_a[2])   # the inactive env is untouched


@pytest.mark.parametrize("spec_fn", [H.polar_ring, H.split_rotated_channel, H.odd_channel])
def test_multilevel_preconditioned_step_matches_the_oracle(spec_fn):
    """Whole PISO step with the preconditioned on-chip CG against the oracle's DIRECT solves, on the meshes where CG is a valid
    solver (symmetric pressure matrix): same parity bound as the plain recurrence (test_piso_step_matches_oracle)."""
    spec = spec_fn()
    d = spec.oracle()
    B = 2
    dom = spec.native(batch=B)
    assert dom.set_pressure_multilevel() is not None
    dt = [0.05, 0.03]
    states = [_state(d, 10 + b) for b in range(B)]
    _load(dom, states)
    its = dom.piso_step(dt, advection_tol=1e-7, pressure_tol=2e-6, pressure_use_bicgstab=False, pressure_project_mean=True)
    assert all(i > 0 for i in its)
    u_gpu, p_gpu = dom.velocity.cpu().numpy(), dom.pressure.cpu().numpy()
    refs = _assembly_parity(dom, d, states, dt, B, check_div=False)
    for b in range(B):
        assert _rel(u_gpu[b], refs[b][0]) < 2e-4, (spec_fn.__name__, b)
        assert _rel(p_gpu[b], refs[b][1]) < 2e-3, (spec_fn.__name__, b)
    dom.close()


@pytest.mark.parametrize("spec_fn", [H.polar_ring, H.split_rotated_channel, H.odd_channel])
def test_multilevel_kernel_form_applies_the_tables(spec_fn):
    """mb_ml_apply (k_ml_restrict / k_ml_coarse / k_ml_prolong) against the NumPy formula on the same tables:
    z = r / diag + (1 / 2s) Z4 D4^-1 Z4^T r + (1 / s) Z8 A8^+ Z8^T r with s = trace(P) / trace(S)."""
    spec = spec_fn()
    B = 3
    dom = spec.native(batch=B)
    assert dom.set_pressure_multilevel() is not None
    tab = dom._multilevel_tables
    P = dom.unit_pressure_matrix()           # leaves the A = 1 matrix assembled on the device
    N = dom.n_cells
    rng = np.random.default_rng(3)
    r = rng.standard_normal((B, N)).astype(np.float32)
    z = dom.multilevel_apply(torch.from_numpy(r)).cpu().numpy().astype(np.float64)
    a4, p4 = tab["a4"], tab["parent4"]
    scale_inv = tab["geom_diag_sum"] / P.diagonal().sum()
    for b in range(B):
        rb = r[b].astype(np.float64)
        r4 = np.bincount(a4, weights=rb, minlength=tab["n4"])
        r8 = np.bincount(p4, weights=r4, minlength=tab["n8"])
        ref = rb / P.diagonal() + 0.5 * scale_inv * (r4 / tab["d4"])[a4] + scale_inv * (tab["aci8"] @ r8)[p4[a4]]
        assert _rel(z[b], ref) < 2e-5, (spec_fn.__name__, b)
    dom.close()


@pytest.mark.parametrize("div", [4, 2])
def test_multilevel_kernel_form_on_the_airfoil_mesh(div):
    """The same check on the six-block Airfoil2D mesh (blocks of 19 x 71 ... 95 x 159 cells at full resolution; balanced tiles of
    unequal sizes, the coarse constant deflated): kernel form against the NumPy formula on the tables."""
    from fluidgym_amd.envs.airfoil_grid import make_airfoil_mesh
    from fluidgym_amd.envs.cylinder_grid import build_domain

    B = 2
    dom = build_domain(make_airfoil_mesh(attack_angle_deg=10.0, resolution_div=div), 0.001, batch=B)
    assert dom.set_pressure_multilevel() is not None
    tab = dom._multilevel_tables
    P = dom.unit_pressure_matrix()
    N = dom.n_cells
    rng = np.random.default_rng(5)
    r = rng.standard_normal((B, N)).astype(np.float32)
    z = dom.multilevel_apply(torch.from_numpy(r)).cpu().numpy().astype(np.float64)
    a4, p4 = tab["a4"], tab["parent4"]
    scale_inv = tab["geom_diag_sum"] / P.diagonal().sum()
    for b in range(B):
        rb = r[b].astype(np.float64)
        r4 = np.bincount(a4, weights=rb, minlength=tab["n4"])
        r8 = np.bincount(p4, weights=r4, minlength=tab["n8"])
        ref = rb / P.diagonal() + 0.5 * scale_inv * (r4 / tab["d4"])[a4] + scale_inv * (tab["aci8"] @ r8)[p4[a4]]
        assert _rel(z[b], ref) < 5e-5, (div, b)
    dom.close()


@pytest.mark.parametrize("spec_fn", [H.polar_ring, H.split_rotated_channel, H.odd_channel])
@pytest.mark.parametrize("bicg", [1, 2])
def test_multilevel_preconditioned_bicgstab_step_matches_the_oracle(spec_fn, bicg):
    """The kernel form of the multilevel preconditioner (mb_ml_apply: restrict / coarse / prolong launches) as RIGHT preconditioner
    of the pressure BiCGStab (plain, and with fp64 refinement): whole PISO step against the oracle's direct solves, same parity
    bound as the plain recurrence."""
    spec = spec_fn()
    d = spec.oracle()
    B = 2
    dom = spec.native(batch=B)
    assert dom.set_pressure_multilevel() is not None
    dt = [0.05, 0.03]
    states = [_state(d, 10 + b) for b in range(B)]
    _load(dom, states)
    its = dom.piso_step(dt, advection_tol=1e-7, pressure_tol=2e-6, pressure_use_bicgstab=bicg, pressure_project_mean=True)
    assert all(i > 0 for i in its)
    u_gpu, p_gpu = dom.velocity.cpu().numpy(), dom.pressure.cpu().numpy()
    refs = _assembly_parity(dom, d, states, dt, B, check_div=False)
    for b in range(B):
        assert _rel(u_gpu[b], refs[b][0]) < 2e-4, (spec_fn.__name__, b)
        assert _rel(p_gpu[b], refs[b][1]) < 2e-3, (spec_fn.__name__, b)
    dom.close()


@pytest.mark.parametrize("dump,vec4", [("a", 31), ("a", 0), ("b", 0), ("b", 31), ("c", 0), ("c", 31)])
def test_captured_bicgstab_breakdowns_now_converge(dump, vec4, monkeypatch):
    """The "intermittent non-finite BiCGStab solve" of round 1: three velocity systems of developing Airfoil2D
    batches on which the fp32 recurrence hit an EXACT breakdown (rho = rw.r or rw.v summing to 0.0 at the rounding level) --
    deterministically, a with the four-cell kernels, b and c with the one-cell kernels (21 of 21 repeats each before the guard,
    profiles/bicg_stress.py).  With the breakdown guard (k_mbb_p: restart from the current residual; k_mbb_s: alpha = 0) the
    plain solve -- no retry ladder -- converges on all of them in both kernel forms, and the answer solves the dumped system."""
    import ctypes
    import os
    from fluidgym_amd import _lib as L
    from fluidgym_amd.envs.airfoil_grid import make_airfoil_mesh
    from fluidgym_amd.envs.cylinder_grid import build_domain

    monkeypatch.setenv("FG_MB_BICG_VEC4", str(vec4))      # read once per handle, at fg_mb_create
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"bicg_breakdown_{dump}.npz"))
    B = 2
    dom = build_domain(make_airfoil_mesh(attack_angle_deg=10.0), 0.001, batch=B)
    N, d = dom.n_cells, dom.dims
    assert z["A"].shape == (N,)
    lib, hip = L.load(), ctypes.CDLL("libamdhip64.so")
    for which, host in ((L.FG_MB_BUF_A, np.repeat(z["A"][None], B, 0)), (L.FG_MB_BUF_C_OFF, np.repeat(z["Coff"][None], B, 0)),
                        (L.FG_MB_BUF_RHS, np.repeat(z["rhs"][None], B, 0))):
        ptr, cnt = ctypes.c_void_p(), ctypes.c_int64()
        L.check(lib.fg_mb_get_buffer(dom.handle, which, ctypes.byref(ptr), ctypes.byref(cnt)))
        t = torch.from_numpy(np.ascontiguousarray(host, np.float32)).cuda()
        assert t.numel() == cnt.value
        assert hip.hipMemcpy(ptr, ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(4 * t.numel()), 3) == 0
    torch.cuda.synchronize()
    out = (ctypes.c_int64 * 4)()
    L.check(lib.fg_mb_debug_bicgstab(dom.handle, 1e-6, 5000, 3, out, None, None, None))
    assert (out[0], out[1], out[2]) == (3, 0, 0), list(out)          # three solves, none non-finite, none unconverged
    assert out[3] < 60                                                # (18-22 iterations without the breakdown)
    x = dom.buffer(L.FG_MB_BUF_VELOCITY_RESULT).view(B, d, N).cpu().numpy().astype(np.float64)
    nbr = dom.neighbors()
    A, C = z["A"].astype(np.float64), z["Coff"].astype(np.float64)
    for b in range(B):
        for c in range(d):
            y = A * x[b, c]
            for f in range(2 * d):
                ok = nbr[f] >= 0
                y[ok] += C[f][ok] * x[b, c][nbr[f][ok]]
            res = z["rhs"][c].astype(np.float64) - y
            # criterion: 1e-6 on the fp32 recurrence residual; the true residual of the fp32 answer carries eps |C| |x| ~ 1e-5
            # (the right-hand sides have an RMS of 5-42: 3e-5 is 1e-6 relative)
            assert np.sqrt((res ** 2).mean()) < 3e-5, (dump, vec4, b, c)
    dom.close()


def test_recurrence_words_read_back_exactly():
    """The access pattern of the multi-kernel Krylov recurrences in isolation (fg_coherence_litmus): sums accumulated with
    device-scope atomics, zeroed by a leader workgroup and read by every wave of the next kernel, plus a flag word stored by the
    leader and read by the four kernels that follow, must read back exactly, launch after launch, with the plain loads / stores
    the solvers use (and with agent-scope atomic ones).  The shape is the Airfoil2D batch the round-1 failures were captured on
    (16 envs x 2 components, 46664 cells); they turned out to be breakdowns of the recurrence, not lost updates."""
    import ctypes
    from fluidgym_amd import _lib as L
    for access in (0, 11):
        bad = (ctypes.c_int64 * 12)()
        val = (ctypes.c_double * 12)()
        L.check(L.load().fg_coherence_litmus(access, 32, 46664, 3000, bad, val, None))
        assert sum(bad) == 0, (access, list(bad), list(val))


def test_l2_form_onchip_cg_is_the_register_resident_one(monkeypatch):
    """Round 4: meshes of 16-24 k cells (the cylinder's ``medium`` / ``hard`` ids, resolution 32: 23 424 cells) run the preconditioned
    pressure CG of an env in one workgroup with ONLY the search direction in LDS and x, r, M p / z as per-env vectors in L2
    (``k_mbc_l2``, csrc/fg_mb_onchip.hip) -- against the register-resident cell-ordered kernel with its residual copy in global
    memory (``k_mbc_onchip<..., RTG>``, FG_MB_OC_RTG_NT=1, read at fg_mb_create): the same recurrence, preconditioner, restart and
    best-iterate rules with another summation order of the dot products, so iteration counts may differ by one or two and the
    projected velocities agree to what two Krylov trajectories at that tolerance do.  Covers the start from zero, the start from an
    iterate (second non-orthogonal pass, warm-start policy: the kernel's residual pass) and an inactive env."""
    from fluidgym_amd.envs.cylinder_grid import build_domain, make_vortex_street_mesh

    mesh = make_vortex_street_mesh(32)
    out = {}
    for form in ("1", "0"):
        if form == "1":
            monkeypatch.setenv("FG_MB_OC_RTG_NT", "1")
        else:
            monkeypatch.delenv("FG_MB_OC_RTG_NT", raising=False)
        dom = build_domain(mesh, 0.01, batch=3)
        assert 16384 < dom.n_cells <= 24576
        dom.set_stall_limit(5000)
        ml = dom.set_pressure_multilevel()
        assert ml is not None and ml["n4"] <= 2048 and ml["n8"] <= 512, ml
        g = torch.Generator(device="cpu").manual_seed(11)
        dom.velocity.copy_((0.3 * torch.randn(dom.velocity.shape, generator=g)).to(dom.device))
        dom.velocity[:, 0] += 1.0
        dom.solver_counters(reset=True)
        dom.make_divergence_free(pressure_tol=1e-7, max_iterations=5000, pressure_project_mean=True)
        u0 = dom.velocity.cpu().numpy().copy()
        for _ in range(2):
            dom.piso_step([0.01, 0.02, 0.0], pressure_tol=1e-7, advection_tol=1e-7, pressure_project_mean=True)   # env 2 inactive
        dom.piso_step([0.01, 0.02, 0.0], pressure_tol=1e-7, advection_tol=1e-7, pressure_project_mean=True, pressure_non_ortho_steps=2,
                      pressure_warm_start=True)
        out[form] = (u0, dom.velocity.cpu().numpy().copy(), dom.pressure.cpu().numpy().copy(), dom.solver_counters())
        dom.close()
    (u0_r, u_r, p_r, c_r), (u0_l, u_l, p_l, c_l) = out["1"], out["0"]
    assert np.isfinite(u_l).all() and np.isfinite(p_l).all()
    assert _rel(u0_l, u0_r) < 2e-5 and _rel(u_l, u_r) < 5e-5, (_rel(u0_l, u0_r), _rel(u_l, u_r))
    for k in ("pressure0", "pressure1"):
        assert abs(c_l[k]["mean"] - c_r[k]["mean"]) <= 2.0 and c_l[k]["unconverged"] == 0 and c_l[k]["mean"] < 200, (c_r, c_l)     # (1e-7 on a rough random field: ~90 iterations)
    np.testing.assert_array_equal(u_l[2], u0_l[2])   # the inactive env is untouched
