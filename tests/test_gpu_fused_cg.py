"""The three-launch preconditioned pressure CG (csrc/fg_fftcg.hip: k_fcg_update_fwd, k_fcg_inv_apply + the verdict inside
k_tridiag_y_lds) against the oracle's direct solve and against the five-kernel iteration it replaces (FG_CG_FUSED=0), through the
C ABI.  Replaces cgSolveGPU (cg_solver_kernel.cu:129-471) on the matrix of PISO_build_pressure_matrix
(PISO_multiblock_cuda_kernel.cu:4812-4978); tolerance / criterion as the reference's (cg_solver_kernel.cu:100-106)."""
import numpy as np
import pytest
import torch

from oracle import piso_oracle as O
from tests.helpers import make_case, rel_err

pytestmark = pytest.mark.gpu


def _uniform_x(case):
    """x widths made uniform (the fast transforms need that); y keeps its stretching."""
    nx = len(case.widths[0])
    w = np.full(nx, np.float32(2.0 / nx), np.float32)
    case.widths[0] = w
    case.edges[0] = np.concatenate([[0.0], np.cumsum(w.astype(np.float64))])
    return case


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _oracle_poisson(case, g, rA):
    dom = case.oracle_domain(0, g)
    P, _, _ = O.build_pressure_matrix(dom, 1.0 / rA)
    return P


GRIDS = [((64, 24), (0, 1)), ((128, 40), (0, 1)), ((256, 19), (0, 1)), ((128, 33), (1,)), ((512, 16), (1,)), ((64, 8), (1,))]


@pytest.mark.parametrize("n,fixed_axes", GRIDS)
def test_fused_cg_matches_the_direct_solve_and_the_five_kernels(n, fixed_axes, monkeypatch):
    """cosine basis (FIXED x) and real Fourier basis (periodic x), row counts that are not multiples of the eight rows a workgroup
    transforms, three envs with different coefficient fields."""
    case = _uniform_x(make_case(dims=2, n=n, fixed_axes=fixed_axes, B=3, seed=5, stretch=0.4))
    g = case.grid()
    rng = np.random.default_rng(11)
    rA = (1.0 / (100.0 * rng.uniform(0.8, 1.35, size=(case.B,) + case.shape))).astype(np.float32)
    b_ = rng.standard_normal((case.B,) + case.shape)
    b_ -= b_.mean(axis=(1, 2), keepdims=True)
    b_ = b_.astype(np.float32)
    tol = 2e-6
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FG_CG_FUSED", mode)
        ns = case.native()
        assert ns.has_fd
        x = torch.zeros((case.B,) + case.shape, device="cuda")
        info = ns.poisson_fdcg(torch.from_numpy(rA).cuda(), torch.from_numpy(b_).cuda(), x, tol=tol)
        torch.cuda.synchronize()
        out[mode] = (_np(x), [(i.used_iterations, i.final_residual, i.converged) for i in info])
        ns.close()
    for b in range(case.B):
        P = _oracle_poisson(case, g, rA[b].astype(np.float64))
        ref = O.solve_direct(P, b_[b].astype(np.float64).ravel(), singular=True).reshape(case.shape)
        ref -= ref.mean()
        true_res = {}
        for mode in ("1", "0"):
            got, info = out[mode]
            it, res, conv = info[b]
            assert conv and res < tol, (mode, info[b])
            assert rel_err(got[b] - got[b].mean(), ref) < 1e-4, (mode, b)
            true_res[mode] = np.sqrt(np.mean((P @ got[b].ravel() - b_[b].astype(np.float64).ravel()) ** 2))
        # the residual the solver reports is the residual of what it returns: up to the fp32 round-off of the recurrence, which the
        # five-kernel iteration shows on the same system
        print(f"{n} env {b}: iterations {out['1'][1][b][0]} / {out['0'][1][b][0]}, true residual fused {true_res['1']:.2e}, five kernels {true_res['0']:.2e}")
        assert true_res["1"] < max(3 * tol, 2.0 * true_res["0"]), (b, true_res)
        assert abs(out["1"][1][b][0] - out["0"][1][b][0]) <= 1, (out["1"][1][b], out["0"][1][b])
        assert rel_err(out["1"][0][b] - out["1"][0][b].mean(), out["0"][0][b] - out["0"][0][b].mean()) < 5e-5


def _channel_case(n, B=3, seed=2):
    return _uniform_x(make_case(dims=2, n=n, fixed_axes=(0, 1), B=B, seed=seed, stretch=0.3, with_source=True, vel_scale=0.3,
                                through_flow_axis=0))


@pytest.mark.parametrize("n", [(64, 24), (128, 36)])
def test_piso_step_with_the_fused_cg_is_the_step_of_the_five_kernels_and_of_the_oracle(n, monkeypatch):
    """fg_piso_step end to end: the fused solver, the mean removal folded into the last corrector (the block pressure is mean-free and
    equals the five-kernel path's), one env masked out (dt = 0) left untouched."""
    case = _channel_case(n)
    g = case.grid()
    dt = [0.02, 0.0, 0.03]
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FG_CG_FUSED", mode)
        ns = case.native()
        v0 = ns.velocity.clone()
        ok, stats = ns.piso_step(dt, advection_tol=1e-7, pressure_tol=1e-7)
        torch.cuda.synchronize()
        assert ok, stats
        res[mode] = (_np(ns.velocity), _np(ns.pressure), stats)
        assert torch.equal(ns.velocity[1], v0[1])     # masked env
        ns.close()
    for b in (0, 2):
        dom = case.oracle_domain(b, g)
        O.piso_split_step(dom, dt[b])
        for mode in ("1", "0"):
            u, p, _ = res[mode]
            assert rel_err(u[b], dom.velocity) < 3e-5, (mode, b)
            assert abs(p[b].mean()) < 1e-5 * np.abs(p[b]).max(), (mode, b, p[b].mean())
            assert rel_err(p[b, 0], dom.pressure) < 2e-4, (mode, b)
        assert rel_err(res["1"][0][b], res["0"][0][b]) < 5e-5      # (two preconditioners, two iterates inside the same tolerance)
        assert rel_err(res["1"][1][b], res["0"][1][b]) < 5e-5


def test_residual_restart_inside_the_fused_loop():
    """reset_steps = 3: the recurrence is restarted from r = b - P x in the middle of a solve (cg_solver_kernel.cu:281-302) and still
    lands on the oracle's step; repeated steps are bit-identical (order-independent accumulators)."""
    case = _channel_case((128, 36), B=2, seed=9)
    g = case.grid()
    ns = case.native()
    ns.set_cg_reset_steps(3)
    state = ns.velocity.clone()
    ok, stats = ns.piso_step(0.05, advection_tol=1e-7, pressure_tol=2e-8)
    torch.cuda.synchronize()
    assert stats[2] >= 3, stats       # (0-based index of the last iteration: at least one restart happened)
    u1 = ns.velocity.clone()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)
        O.piso_split_step(dom, 0.05)
        assert rel_err(_np(u1[b]), dom.velocity) < 3e-5
    ns.velocity.copy_(state)
    ns.copy_velocity_result_from_blocks()
    ns.reset_solver_state()
    ns.piso_step(0.05, advection_tol=1e-7, pressure_tol=2e-8)
    torch.cuda.synchronize()
    assert torch.equal(ns.velocity, u1)
    ns.close()


def test_unconverged_fused_solve_hands_back_its_best_iterate():
    """iteration cap below what the tolerance needs: FG_ERR_NOT_CONVERGED is reported in-band, the result is finite and is a CG
    iterate of the system (returnBestResult, cg_solver_kernel.cu:345-361: best-iterate bookkeeping as in the five-kernel path)."""
    case = _uniform_x(make_case(dims=2, n=(128, 40), fixed_axes=(0, 1), B=2, seed=5, stretch=0.4))
    g = case.grid()
    rng = np.random.default_rng(1)
    rA = (1.0 / (100.0 * rng.uniform(0.5, 2.0, size=(case.B,) + case.shape))).astype(np.float32)
    b_ = rng.standard_normal((case.B,) + case.shape)
    b_ -= b_.mean(axis=(1, 2), keepdims=True)
    b_ = b_.astype(np.float32)
    ns = case.native()
    x = torch.zeros((case.B,) + case.shape, device="cuda")
    info = ns.poisson_fdcg(torch.from_numpy(rA).cuda(), torch.from_numpy(b_).cuda(), x, tol=1e-12, max_iterations=3)
    torch.cuda.synchronize()
    got = _np(x)
    assert np.isfinite(got).all()
    for b in range(case.B):
        assert not info[b].converged and info[b].is_finite
        P = _oracle_poisson(case, g, rA[b].astype(np.float64))
        true_res = np.sqrt(np.mean((P @ got[b].ravel() - b_[b].astype(np.float64).ravel()) ** 2))
        assert true_res < 0.2 * np.sqrt(np.mean(b_[b].astype(np.float64) ** 2)), (b, true_res)     # (three iterations on coefficients 0.5 .. 2)
    ns.close()


def test_rowmean_preconditioner_is_exact_for_row_constant_coefficients(monkeypatch):
    """The fused CG is preconditioned by the pressure operator with 1/A averaged along x per row and env (k_fd_rowmean_factor): for a
    coefficient field that only varies across the channel it IS the operator, so the solve ends in its first iteration (or its second, at fp32 round-off), for every
    env with its own profile; with FG_FD_ROWMEAN=0 (the grid's A = 1 factors) the same systems take several."""
    case = _uniform_x(make_case(dims=2, n=(128, 40), fixed_axes=(0, 1), B=3, seed=5, stretch=0.4))
    g = case.grid()
    rng = np.random.default_rng(2)
    prof = 1.0 / (100.0 * rng.uniform(0.6, 1.8, size=(case.B, case.shape[0], 1)))
    rA = np.broadcast_to(prof, (case.B,) + case.shape).astype(np.float32).copy()
    b_ = rng.standard_normal((case.B,) + case.shape)
    b_ -= b_.mean(axis=(1, 2), keepdims=True)
    b_ = b_.astype(np.float32)
    its = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FG_FD_ROWMEAN", mode)
        ns = case.native()
        x = torch.zeros((case.B,) + case.shape, device="cuda")
        info = ns.poisson_fdcg(torch.from_numpy(rA).cuda(), torch.from_numpy(b_).cuda(), x, tol=1e-5)
        torch.cuda.synchronize()
        assert all(i.converged for i in info)
        its[mode] = [i.used_iterations for i in info]
        got = _np(x)
        for b in range(case.B):
            P = _oracle_poisson(case, g, rA[b].astype(np.float64))
            ref = O.solve_direct(P, b_[b].astype(np.float64).ravel(), singular=True).reshape(case.shape)
            assert rel_err(got[b] - got[b].mean(), ref - ref.mean()) < 1e-4, (mode, b)
        ns.close()
    print("row-constant coefficients: iterations row-mean", its["1"], "A = 1 factors", its["0"])
    # (used_iterations is the 0-based index of the last iteration: one or, where the fp32 round-off of the transforms leaves the first
    #  residual just above 1e-5 of a unit right-hand side, two iterations against eight or nine)
    assert max(its["1"]) <= 1 and min(its["0"]) >= 5, its


def test_env_whose_start_vector_meets_the_tolerance_gets_a_zero_pressure():
    """k_fcg_div_fwd starts the solve without storing x_0 = 0 and r_0 = b; the first update kernel writes x and r -- and, for an env
    the verdict on x_0 already stopped (right-hand side below the tolerance: a fluid at rest), the zeros of its result.  Two steps:
    the first leaves a pressure field in pressureResult, then env 1 is put at rest and must come back with exactly zero pressure."""
    case = _uniform_x(make_case(dims=2, n=(128, 32), fixed_axes=(0, 1), B=3, seed=4, stretch=0.2, vel_scale=0.3, wall_motion=0.0))
    ns = case.native()
    ok, stats = ns.piso_step(0.02, advection_tol=1e-7, pressure_tol=1e-6)
    assert ok and float(ns.pressure[1].abs().max()) > 0
    ns.velocity[1].zero_()
    ns.copy_velocity_result_from_blocks()
    ok, stats = ns.piso_step(0.02, advection_tol=1e-7, pressure_tol=1e-6)
    torch.cuda.synchronize()
    assert ok, stats
    c = ns.solver_counters()
    assert float(ns.pressure[1].abs().max()) == 0.0 and float(ns.velocity[1].abs().max()) == 0.0
    assert float(ns.pressure[0].abs().max()) > 0 and float(ns.pressure[2].abs().max()) > 0
    g = case.grid()
    ns.close()


def test_first_iterate_is_judged_before_any_vector_is_updated(monkeypatch):
    """k_fcg_check0: |r_1| from dot products of the first inverse kernel (d.d + 2 (1 - alpha) d.w + (1 - alpha)^2 w.w, d = r - w: no
    cancellation -- env 2's first residual is fp32 noise, 3e-6 of |b|, and must be reported as such).  With a tolerance placed
    BETWEEN the envs' first residuals the batch is mixed -- some envs end on x_1 = alpha z (written by the short path of the first
    update kernel), the others iterate on -- and every env must come out as under FG_FCG_FIRST=0 (the verdict taken on the updated
    residual): same iteration counts, the one-iteration envs bit for bit."""
    import fluidgym_amd._lib as L
    case = _uniform_x(make_case(dims=2, n=(128, 40), fixed_axes=(0, 1), B=4, seed=7, stretch=0.4))
    rng = np.random.default_rng(3)
    rA = (1.0 / (100.0 * rng.uniform(0.6, 1.6, size=(case.B,) + case.shape))).astype(np.float32)
    rA[2] = rA[2].mean()      # a constant-coefficient env: the preconditioner is exact there, one iteration whatever the tolerance
    b_ = rng.standard_normal((case.B,) + case.shape)
    b_ -= b_.mean(axis=(1, 2), keepdims=True)
    b_ = (b_ * np.array([1.0, 3.0, 1.0, 10.0])[:, None, None]).astype(np.float32)

    def solve(tol, first, max_iterations=500):
        monkeypatch.setenv("FG_FCG_FIRST", first)
        ns = case.native()
        x = torch.zeros((case.B,) + case.shape, device="cuda")
        info = ns.poisson_fdcg(torch.from_numpy(rA).cuda(), torch.from_numpy(b_).cuda(), x, tol=tol, max_iterations=max_iterations)
        torch.cuda.synchronize()
        out = (x.clone(), [(i.used_iterations, float(i.final_residual), i.converged) for i in info])
        ns.close()
        return out

    # every env's residual after ONE iteration (from the updated vector), to place the tolerance between them
    _, one = solve(1e-30, "0", max_iterations=1)
    c = np.array([r for _, r, _ in one])
    order = np.sort(c)
    assert order[1] < 0.7 * order[2], c
    tol = float(np.sqrt(order[1] * order[2]))      # two envs end on the first iterate, two go on
    x1, i1 = solve(tol, "1")
    x0, i0 = solve(tol, "0")
    assert [u for u, _, _ in i1] == [u for u, _, _ in i0], (i1, i0)
    assert sorted(u == 0 for u, _, _ in i1) == [False, False, True, True], i1
    assert all(cv for _, _, cv in i1)
    for b in range(case.B):
        if i1[b][0] == 0:
            assert torch.equal(x1[b], x0[b]), b
            noise = 3e-7 * float(np.sqrt((b_[b].astype(np.float64) ** 2).mean()))      # (what fp32 rounding of r_1 itself amounts to)
            assert abs(i1[b][1] - i0[b][1]) <= 2e-3 * i0[b][1] + noise, (b, i1[b], i0[b])      # the two ways to the same |r_1|
            assert abs(i1[b][1] - c[b]) <= 2e-3 * c[b] + noise
        else:
            assert rel_err(_np(x1[b]), _np(x0[b])) < 1e-6, b


def test_piso_step_that_never_stores_its_pressure(monkeypatch):
    """When EVERY env of a pressure solve ends on its first iterate, nothing is written at all: the correctors read alpha z
    (FgLazyRef, k_correct) and the last one stores pressureResult.  Against FG_FCG_FIRST=0 (x stored by the update kernel): the
    velocity and pressureResult bit for bit, the block pressure up to the rounding of its mean (alpha sum(z) against sum(alpha z)),
    a masked env untouched (tests/test_gpu_fused_cg.py::test_env_whose_start_vector_meets_the_tolerance_gets_a_zero_pressure has the
    env that takes no iteration)."""
    import fluidgym_amd._lib as L
    case = _channel_case((128, 36), B=4)
    dt = [0.02, 0.0, 0.03, 0.02]

    def run(first, tol, correctors):
        monkeypatch.setenv("FG_FCG_FIRST", first)
        ns = case.native()
        v0 = ns.velocity.clone()
        for _ in range(3):
            ok, stats = ns.piso_step(dt, corrector_steps=correctors, advection_tol=1e-7, pressure_tol=tol)
            assert ok, stats
        torch.cuda.synchronize()
        cfg = ns.config_dump()
        out = (ns.velocity.clone(), ns.pressure.clone(), ns.buffer(L.FG_BUF_P_RESULT, (case.B,) + case.shape).clone(),
               ns.buffer(L.FG_BUF_DIV, (case.B,) + case.shape).clone(), cfg["unstored_pressure_solves"], cfg["first_iterate_polls"])
        assert torch.equal(ns.velocity[1], v0[1])
        ns.close()
        return out

    for correctors in (1, 2):
        # a tolerance the solves meet after ONE iteration but (the first corrector's at least) not at the start vector
        div = run("1", 1e-7, correctors)[3]
        rms = min(float(div[b].double().pow(2).mean().sqrt()) for b in (0, 2))
        for mult in (0.2, 0.5, 0.9):
            u1, p1, r1, _, unstored, polls = run("1", mult * rms, correctors)
            if unstored >= 2:
                break
        assert unstored >= 2, (correctors, unstored, polls)
        u0, p0, r0, _, unstored0, _ = run("0", mult * rms, correctors)
        assert unstored0 == 0
        assert torch.equal(u1, u0), correctors
        for b in (0, 2, 3):
            assert torch.equal(r1[b], r0[b]), (correctors, b)
        scale = float(p0.abs().max())
        assert scale > 0 and float((p1 - p0).abs().max()) <= 2e-6 * scale
        for b in (0, 2, 3):
            assert abs(float(p1[b].double().mean())) < 1e-5 * float(p1[b].abs().max())


def test_corrector_launched_behind_the_verdict_kernel_changes_no_bit(monkeypatch):
    """FG_FCG_SPEC: the corrector in its unstored-pressure form is launched behind k_fcg_check0, before the host has seen the verdict;
    when an env turns out to iterate on, the corrector that follows the finished solve overwrites what it wrote.  Tolerances on both
    sides of the envs' first residuals (every solve ends on the first iterate / some envs go on / the start vector already does):
    velocity, block pressure and pressureResult bit for bit against FG_FCG_SPEC=0, a masked env untouched."""
    import fluidgym_amd._lib as L
    case = _channel_case((128, 36), B=4)
    dt = [0.02, 0.0, 0.03, 0.02]
    ns = case.native()
    ok, _ = ns.piso_step(dt, corrector_steps=1, advection_tol=1e-7, pressure_tol=1e-7)
    div = ns.buffer(L.FG_BUF_DIV, (case.B,) + case.shape)
    rms = min(float(div[b].double().pow(2).mean().sqrt()) for b in (0, 2, 3))
    ns.close()
    for mult in (0.5, 0.05, 0.005, 5.0):
        res = {}
        for spec in ("1", "0"):
            monkeypatch.setenv("FG_FCG_SPEC", spec)
            ns = case.native()
            v0 = ns.velocity.clone()
            its = []
            for _ in range(3):
                ok, stats = ns.piso_step(dt, advection_tol=1e-7, pressure_tol=mult * rms)
                assert ok, stats
                its.append(tuple(stats))
            torch.cuda.synchronize()
            res[spec] = (ns.velocity.clone(), ns.pressure.clone(), ns.buffer(L.FG_BUF_P_RESULT, (case.B,) + case.shape).clone(), its,
                         ns.config_dump()["unstored_pressure_solves"])
            assert torch.equal(ns.velocity[1], v0[1])
            ns.close()
        a, b = res["1"], res["0"]
        assert a[3] == b[3] and a[4] == b[4], (mult, a[3], b[3], a[4], b[4])
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), mult
        for e in (0, 2, 3):
            assert torch.equal(a[2][e], b[2][e]), (mult, e)
        print("SPEC", mult, "iterations", a[3], "unstored solves", a[4])


@pytest.mark.parametrize("n", [(128, 36), (256, 128), (512, 256)])
def test_factors_made_by_the_first_tridiagonal_solve_are_the_factor_kernels(n, monkeypatch):
    """Round 6: the per-env factors of the row-mean preconditioner are made by the first tridiagonal solve after 1/A changed
    (``k_tridiag_y_lds<.., FAC>``: k_fd_rowmean_factor's arithmetic inside wave 0's forward sweep) instead of by a launch of their
    own (FG_FD_FACFUSE=0).  Same row means, same pivots, the same folded solve: every field bit for bit over three PISO steps,
    tolerances on both sides of the first residuals (an env the verdict stops before its first application still gets its factors for
    the second corrector), a masked env untouched; 128 rows (three LDS arrays) and 256 rows (two: c' written over the forward sweep's
    multipliers)."""
    import fluidgym_amd._lib as L
    case = _channel_case(n, B=4)
    dt = [0.02, 0.0, 0.03, 0.02]
    ns = case.native()
    ok, _ = ns.piso_step(dt, corrector_steps=1, advection_tol=1e-7, pressure_tol=1e-7)
    div = ns.buffer(L.FG_BUF_DIV, (case.B,) + case.shape)
    rms = min(float(div[b].double().pow(2).mean().sqrt()) for b in (0, 2, 3))
    ns.close()
    for mult in (0.05, 5.0, 0.5):
        res = {}
        for fuse in ("1", "0"):
            monkeypatch.setenv("FG_FD_FACFUSE", fuse)
            ns = case.native()
            v0 = ns.velocity.clone()
            its = []
            for _ in range(3):
                ok, stats = ns.piso_step(dt, advection_tol=1e-7, pressure_tol=mult * rms)
                assert ok, stats
                its.append(tuple(stats))
            torch.cuda.synchronize()
            res[fuse] = (ns.velocity.clone(), ns.pressure.clone(), ns.buffer(L.FG_BUF_P_RESULT, (case.B,) + case.shape).clone(), its)
            assert torch.equal(ns.velocity[1], v0[1])
            ns.close()
        a, b = res["1"], res["0"]
        assert a[3] == b[3], (mult, a[3], b[3])
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), mult
