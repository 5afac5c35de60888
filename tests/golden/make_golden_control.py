"""Golden vectors for two pieces of host-side control logic of the hot path, produced by the reference's OWN Python running
HERE against recording stand-ins (no reference source is copied):

* the adaptive sub-step schedule of ``Simulation._PISO_adaptive_step`` (``pict/PISOtorch_simulation.py:2004-2064``): for a
  scripted sequence of ``getMaxVelocity`` values, the time steps handed to ``_PISO_split_step``;
* the retry ladder of ``_linear_solve_wrapper`` (``pict/PISOtorch_diff.py:373-488``): for scripted outcomes of
  ``PISOtorch.SolveLinear``, which attempts are made, in which precision, with or without the preconditioner, and whether the
  result tensor is cleared before an attempt.

    python tests/golden/make_golden_control.py        ->  tests/golden/reference_control.json
"""
import importlib.util
import json
import logging
import os
import sys
import types
from contextlib import nullcontext

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_outflow as G  # noqa: E402  (the stand-in loader of the simulation module)

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


# ---- adaptive sub-steps --------------------------------------------------------------------------------------------------
def adaptive_cases(sim_mod):
    fn = sim_mod.Simulation._PISO_adaptive_step
    cases = []
    rng = np.random.default_rng(0)
    specs = [(0.1, 0.8, [3.0] * 40), (0.1, 0.8, [0.0] * 4), (0.05, 0.5, [100.0, 80.0, 60.0, 40.0, 20.0, 10.0] + [5.0] * 200),
             (1.0, 0.8, [0.79999] * 4), (1.0, 0.8, [0.8] * 4), (1.0, 0.8, [0.80001] * 8), (0.02, 0.9, list(40.0 + 30.0 * rng.random(400))),
             (0.25, 0.3, list(np.geomspace(1.0, 50.0, 300))), (1e-3, 0.8, [1e-9] * 4), (0.3, 0.8, list(20.0 * rng.random(600) + 1.0))]
    for time_step, cfl, vels in specs:
        steps, it = [], iter(vels)

        class _Dom:
            def getMaxVelocity(self, *a):
                return torch.tensor(next(it), dtype=torch.float32)

            def getBlock(self, i):
                return types.SimpleNamespace(velocity=torch.zeros(1, dtype=torch.float32))

        me = types.SimpleNamespace(time_step=time_step, adaptive_CFL=cfl, domain=_Dom(), print_adaptive_step_info=False,
                                   _check_domain=lambda: None, _check_stop=lambda: False,
                                   _PISO_split_step=lambda iterations, time_step: (steps.append(float(time_step[0])), True)[1])
        setattr(me, "_Simulation__LOG", logging.getLogger("golden"))
        ok = fn(me)
        used = len(steps)
        cases.append({"time_step": time_step, "cfl": cfl, "max_velocities": [float(np.float32(v)) for v in vels[:used]], "ok": bool(ok),
                      "split_steps": steps})
    return cases


# ---- retry ladder --------------------------------------------------------------------------------------------------------
def load_diff_module(solve_linear):
    for pkg in ["fluidgym", "fluidgym.simulation", "fluidgym.simulation.pict", "fluidgym.simulation.pict.util"]:
        if pkg not in sys.modules:
            G._stub(pkg).__path__ = []

    class _Ext(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return type(name, (), {})

    pt = _Ext("PISOtorch")
    pt.SolveLinear = solve_linear
    G._stub("fluidgym.simulation.extensions", PISOtorch=pt)
    G._stub("fluidgym.simulation.pict.util.profiling", SAMPLE=lambda *a, **k: nullcontext())
    G._stub("fluidgym.simulation.pict.util.logging", get_logger=lambda name="": logging.getLogger(name))
    class StringWriter:       # util/output.py: a line buffer, only used to format the error text
        def __init__(self):
            self.lines = []

        def write_line(self, fmt="", *args):
            self.lines.append(fmt % args if args else fmt)

        def reset(self):
            self.lines = []

        def __str__(self):
            return "\n".join(self.lines)

    G._stub("fluidgym.simulation.pict.util.output", StringWriter=StringWriter)
    return G._load(f"{REF}/fluidgym/simulation/pict/PISOtorch_diff.py", "ref_PISOtorch_diff")


class _Info:
    def __init__(self, converged, finite):
        self.converged, self.isFiniteResidual = converged, finite
        self.finalResidual, self.usedIterations = (1e-9 if converged else (1e-3 if finite else float("nan"))), 7

    def __getattr__(self, name):          # whatever else the reporting code prints
        return 0

    def __str__(self):
        return f"info(converged={self.converged}, finite={self.isFiniteResidual})"


class _Mat:
    def __init__(self, dtype=torch.float32):
        self.dtype = dtype

    def toType(self, dt):
        return _Mat(dt)


def ladder_cases():
    attempts, script = [], []

    def solve_linear(mat, rhs, result, maxit, tol, crit, use_bicg, rank_def, reset, transpose, print_res, best, BiCGwithPreconditioner=True):
        outcome = script.pop(0)
        attempts.append({"dtype": str(rhs.dtype).replace("torch.", ""), "preconditioned": bool(BiCGwithPreconditioner),
                         "result_is_zero": bool(result.eq(0).all()), "use_bicg": bool(use_bicg)})
        result.fill_(float("nan") if outcome == "non_finite" else 1.0)
        return [_Info(outcome == "converged", outcome != "non_finite")]

    diff = load_diff_module(solve_linear)
    cases = []
    grid = []
    for use_bicg, best in ((True, False), (False, True), (True, True)):          # advection solve | pressure CG | pressure BiCGStab
        for dbl in (False, True):
            for pre in (False, True):
                for outcomes in (["converged"], ["unconverged", "converged"], ["unconverged", "unconverged", "converged"],
                                 ["non_finite", "converged"], ["non_finite", "non_finite", "converged"], ["unconverged"] * 3):
                    grid.append((use_bicg, best, dbl, pre, outcomes))
    for use_bicg, best, dbl, pre, outcomes in grid:
        attempts.clear()
        script[:] = list(outcomes) + ["converged"] * 3
        rhs = torch.ones(4)
        result = torch.full((4,), 0.5)          # "previous result" as initial guess
        raised = None
        try:
            diff._linear_solve_wrapper(_Mat(), rhs, result, torch.tensor([100]), torch.tensor([1e-5]), None, use_bicg, False, 0, False, False,
                                       best, is_FWD=True, double_fallback=dbl, BiCG_with_preconditioner=False,
                                       BiCG_precondition_fallback=pre)
        except Exception as e:            # _check_solver_return_infos raises on a failed last attempt
            raised = type(e).__name__
        cases.append({"use_bicg": use_bicg, "return_best_result": best, "double_fallback": dbl, "precondition_fallback": pre,
                      "scripted_outcomes": outcomes[:len(attempts)], "attempts": [dict(a) for a in attempts], "raised": raised})
    return cases


def main():
    sim_mod = G.load_reference_simulation_module()
    out = {"adaptive": adaptive_cases(sim_mod), "ladder": ladder_cases()}
    with open(os.path.join(OUT, "reference_control.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("adaptive cases", len(out["adaptive"]), "ladder cases", len(out["ladder"]))
    for c in out["ladder"][:0]:
        print(c)


if __name__ == "__main__":
    logging.disable(logging.CRITICAL)
    main()
