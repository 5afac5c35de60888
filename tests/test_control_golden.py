"""Host-side control logic of the hot path against vectors produced by the reference's own Python
(tests/golden/make_golden_control.py -> reference_control.json): the adaptive sub-step schedule (oracle restatement here; the
native stepper is held against the oracle in tests/test_gpu_envs.py) and the retry ladder's attempt sequences (the expectation
the GPU ladder test is built from, tests/test_gpu_mb.py)."""
import json
import os

import numpy as np

from oracle import piso_oracle as O

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_control.json")))


def test_adaptive_substep_schedule_matches_the_reference():
    """``_PISO_adaptive_step`` (PISOtorch_simulation.py:2004-2064) fed scripted max velocities: the time steps it hands to
    ``_PISO_split_step`` (rounded through the float32 velocity dtype) against the oracle's ``adaptive_substeps`` loop."""
    assert len(GOLDEN["adaptive"]) >= 10
    multi = 0
    for c in GOLDEN["adaptive"]:
        t_rem, steps, vels = c["time_step"], [], iter(c["max_velocities"])
        while t_rem > 0 and not np.isclose(t_rem, 0):
            _, ts = O.adaptive_substeps(np.float32(next(vels)), t_rem, c["cfl"])   # getMaxVelocity returns a float32 tensor
            t_rem -= ts
            steps.append(float(np.asarray(ts, dtype=np.float32)))
        assert c["ok"] and steps == c["split_steps"], (c["time_step"], c["cfl"], steps[:4], c["split_steps"][:4])
        multi += len(steps) > 1
    assert multi >= 4          # the script exercises real sub-stepping, not only the single-step branch


def ladder_attempts(use_bicg, return_best_result, double_fallback, precondition_fallback, outcomes):
    """Attempt list of the reference for a scripted outcome sequence (first match in the golden grid)."""
    for c in GOLDEN["ladder"]:
        if (c["use_bicg"], c["return_best_result"], c["double_fallback"], c["precondition_fallback"]) == \
                (use_bicg, return_best_result, double_fallback, precondition_fallback) and c["scripted_outcomes"] == outcomes[:len(c["scripted_outcomes"])] \
                and len(c["scripted_outcomes"]) <= len(outcomes):
            return c
    raise KeyError((use_bicg, return_best_result, double_fallback, precondition_fallback, outcomes))


def test_retry_ladder_rules_of_the_reference():
    """What the golden grid says about ``_linear_solve_wrapper`` (PISOtorch_diff.py:410-476) -- the rules fg_mb_piso_step's ladder
    implements: trigger = any system unconverged (solves without returnBestResult) or non-finite (with it); order
    fp64 then preconditioned; every retry starts from a cleared result; a preconditioned retry only for BiCGStab."""
    for c in GOLDEN["ladder"]:
        att = c["attempts"]
        assert att[0]["dtype"] == "float32" and not att[0]["preconditioned"] and not att[0]["result_is_zero"]   # starts from the caller's guess
        for a in att[1:]:
            assert a["result_is_zero"]
        first = c["scripted_outcomes"][0]
        failed = first == "non_finite" or (first == "unconverged" and not c["return_best_result"])
        if not failed:
            assert len(att) == 1, c
            continue
        kinds = [("fp64" if a["dtype"] == "float64" else "preconditioned" if a["preconditioned"] else "plain") for a in att[1:]]
        expect = (["fp64"] if c["double_fallback"] else [])
        second_failed = (not c["double_fallback"]) or (len(c["scripted_outcomes"]) > 1 and (
            c["scripted_outcomes"][1] == "non_finite" or (c["scripted_outcomes"][1] == "unconverged" and not c["return_best_result"])))
        if c["precondition_fallback"] and c["use_bicg"] and second_failed:
            expect.append("preconditioned")
        assert kinds == expect, (c, kinds, expect)
