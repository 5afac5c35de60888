"""The reference's retry ladder of a linear solve on the SINGLE-BLOCK path (``_linear_solve_wrapper``, pict/PISOtorch_diff.py:410-476;
csrc/fg_rung64.h, fg_api.hip advection_solve / solve_pressure): a failed fp32 solve is repeated in fp64 on the same system from a
cleared result (``solver_double_fallback``), then with the preconditioner (``BiCG_precondition_fallback``).  As for the multi-block
path (tests/test_gpu_mb.py), the first attempt is forced to count as failed (``fg_ladder``'s test mask), the rungs the REFERENCE
tries for the same scripted outcomes come from tests/golden/reference_control.json (its own wrapper run on 72 outcome sequences),
and every rung must land on the plain solve's step and on the oracle's."""
import numpy as np
import pytest

from oracle import piso_oracle as O
from tests.helpers import make_case, rel_err
from tests.test_control_golden import GOLDEN, ladder_attempts

pytestmark = pytest.mark.gpu

CASES = {"2d": dict(dims=2, n=(32, 24), fixed_axes=(1,), B=2, seed=41), "3d": dict(dims=3, n=(16, 12, 8), fixed_axes=(1,), B=2, seed=42)}
KW = dict(advection_tol=1e-7, pressure_tol=1e-7)
FORCE = {"advection": 1, "pressure": 2}


def _step(case, double_fallback, precond_mode, force, dt=0.05):
    ns = case.native()
    ns.set_double_fallback(double_fallback)
    ns.set_advection_preconditioner(precond_mode)
    ns.ladder(force_mask=force)
    ok, stats = ns.piso_step(dt, **KW)
    used = ns.ladder(force_mask=0)
    u, p = ns.velocity.cpu().numpy().astype(np.float64), ns.pressure.cpu().numpy().astype(np.float64)
    ns.close()
    return ok, used, u, p


@pytest.mark.parametrize("which", list(CASES))
def test_every_recorded_ladder_sequence_is_the_rung_sequence_of_the_single_block_path(which):
    """All 72 recorded sequences: for each (solver kind, returnBestResult, double_fallback, precondition_fallback, scripted outcomes)
    the wrapper's attempts after the first -- fp64, preconditioned -- must be exactly the rungs this path runs when its attempts are
    forced to fail the same way (an attempt can only be forced to FAIL here, so sequences are replayed up to their first success)."""
    case = make_case(vel_scale=0.4, nu=0.03, with_source=True, **CASES[which])
    seen = set()
    n_checked = 0
    for c in GOLDEN["ladder"]:
        use_bicg, rbr, dfb, pfb = c["use_bicg"], c["return_best_result"], c["double_fallback"], c["precondition_fallback"]
        kind = "advection" if (use_bicg and not rbr) else ("pressure" if (not use_bicg and rbr) else None)
        if kind is None:
            continue        # (BiCGStab with returnBestResult / CG without: combinations the step never issues, PISOtorch_simulation.py:1735-1750, 1900-1915)
        rungs = tuple(("fp64" if a["dtype"] == "float64" else "preconditioned") for a in c["attempts"][1:])
        # what can be forced here: the first attempt fails; the fp64 rung fails too iff the reference went on to the preconditioned one
        force = FORCE[kind] | (4 if rungs == ("fp64", "preconditioned") else 0)
        first = c["scripted_outcomes"][0]
        failed_first = (first != "converged") if kind == "advection" else (first == "non_finite")
        if not failed_first:
            force = 0
        key = (kind, dfb, pfb, force)
        if key in seen:
            continue
        seen.add(key)
        ok, used, u, p = _step(case, dfb, 2 if pfb else 0, force)
        expect = {f"{kind}_fp64": int("fp64" in rungs and failed_first), "advection_preconditioned": int("preconditioned" in rungs and failed_first and kind == "advection")}
        got = {k: int(v > 0) for k, v in used.items()}
        for k, v in expect.items():
            assert got[k] == v, (c, used)
        other = "pressure_fp64" if kind == "advection" else "advection_fp64"
        assert got[other] == 0, (c, used)
        assert np.isfinite(u).all() and np.isfinite(p).all()
        n_checked += 1
    assert n_checked >= 8


@pytest.mark.parametrize("which", list(CASES))
@pytest.mark.parametrize("rungs", ["advection_fp64", "advection_fp64_then_preconditioned", "pressure_fp64"])
def test_rungs_reproduce_the_plain_step_and_the_oracle(which, rungs):
    case = make_case(vel_scale=0.4, nu=0.03, with_source=True, **CASES[which])
    ok, used, u_plain, p_plain = _step(case, False, 0, 0)
    assert ok and used == {"advection_fp64": 0, "advection_preconditioned": 0, "pressure_fp64": 0}
    force, dfb, mode, kind, outcomes = {
        "advection_fp64": (1, True, 0, "advection", ["unconverged", "converged"]),
        "advection_fp64_then_preconditioned": (1 | 4, True, 2, "advection", ["unconverged", "unconverged", "converged"]),
        "pressure_fp64": (2, True, 0, "pressure", ["non_finite", "converged"]),
    }[rungs]
    ref = ladder_attempts(kind == "advection", kind == "pressure", True, mode == 2, outcomes)
    expect = [("fp64" if a["dtype"] == "float64" else "preconditioned") for a in ref["attempts"][1:]]
    assert expect == (["fp64", "preconditioned"] if mode == 2 else ["fp64"]), ref
    ok, used, u, p = _step(case, dfb, mode, force)
    assert ok
    assert used[f"{kind}_fp64"] >= 1 and (used["advection_preconditioned"] >= 1) == (mode == 2)
    g = case.grid()
    for b in range(case.B):
        assert rel_err(u[b], u_plain[b]) < 2e-5, (rungs, b)
        dom = case.oracle_domain(b, g)
        O.piso_split_step(dom, 0.05)
        assert rel_err(u[b], dom.velocity) < 3e-5, (rungs, b)
        pr = dom.pressure - dom.pressure.mean()
        assert rel_err(p[b, 0] - p[b, 0].mean(), pr) < 2e-3, (rungs, b)
