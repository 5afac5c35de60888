"""Call sequence of the reference's PISO split step, produced by the reference's OWN Python running HERE against recording
stand-ins (no reference source is copied): ``Simulation._PISO_split_step`` (``pict/PISOtorch_simulation.py:1431-2002``) is run
with a backend and a domain that only RECORD what is asked of them -- which compiled operator, with which non-orthogonal flags,
which linear solve (matrix, right-hand side, start vector or none, solver kind, tolerance, best-result flag), which hook, in which
order -- for the solver settings of the reference's four env families (channel / RBC: orthogonal branch; TCF, cylinder 2-D / 3-D,
airfoil: non-orthogonal branch with their numbers of non-orthogonal passes).

    python tests/golden/make_golden_split_step.py        ->  tests/golden/reference_split_step.json

``tests/test_split_step_golden.py`` holds the oracle's step functions (``oracle/piso_oracle.py::piso_split_step``,
``oracle/mb_oracle.py::Domain.piso_step``) against these sequences.
"""
import json
import logging
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_outflow as G  # noqa: E402  (the stand-in loader of the simulation module)

OUT = os.path.dirname(os.path.abspath(__file__))
N = 6        # cells of the stand-in domain (the tensors only have to exist)

HOOKS = ["PRE", "POST_SCALAR_SETUP", "PRE_VELOCITY_SETUP", "POST_VELOCITY_SETUP", "POST_PREDICTION", "POST_PRESSURE_SETUP",
         "POST_PRESSURE_RESULT", "POST_PRESSURE_NON_ORTHO", "POST_VELOCITY_CORRECTION", "POST"]


class Tagged(torch.Tensor):
    """a tensor that remembers which field of the domain it is"""

    @staticmethod
    def make(tag, n=N):
        t = torch.zeros(n).as_subclass(Tagged)
        t.tag = tag
        return t


def tag_of(x):
    return None if x is None else getattr(x, "tag", "tensor")


class RecordingDomain:
    def __init__(self, log, scalar):
        self._log, self._scalar = log, scalar
        for name in ["C", "P", "velocityRHS", "scalarRHS", "pressureRHSdiv", "velocityResult", "pressureResult", "scalarResult"]:
            setattr(self, name, Tagged.make(name))

    # queries of the scalar branch (PISOtorch_simulation.py:1471-1486)
    def hasPassiveScalar(self):
        return self._scalar

    def isPassiveScalarViscosityStatic(self):
        return True

    def isAllFixedBoundariesPassiveScalarTypeStatic(self):
        return True

    def getPassiveScalarChannels(self):
        return 1

    def getTotalSize(self):
        return N

    def hasPassiveScalarViscosity(self):
        return True          # RBC: the temperature has its own diffusivity, the matrix is not shared with the velocity

    def hasBlockViscosity(self):
        return False

    def hasPassiveScalarBlockViscosity(self):
        return False

    def UpdateDomainData(self):
        pass                 # bookkeeping of the compiled domain, not an operator

    def _set(self, name, value):
        centred = bool(value.numel() > 0 and abs(float(torch.as_tensor(value).double().mean())) < 1e-6)
        self._log.append({"op": name, "mean_removed": centred} if name == "setPressureResult" else {"op": name})

    def setScalarResult(self, v):
        self._set("setScalarResult", v)

    def setVelocityResult(self, v):
        self._set("setVelocityResult", v)

    def setPressureResult(self, v):
        self._set("setPressureResult", v)


class RecordingBackend:
    def __init__(self, log):
        self._log = log

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def op(domain, *args, **kw):
            rec = {"op": name}
            # positional layout of the compiled operators as the split step calls them: (domain, time_step, non_ortho_flags, ...)
            if name in ("SetupAdvectionMatrix", "SetupAdvectionScalar", "SetupAdvectionVelocity", "SetupPressureCorrection",
                        "SetupPressureMatrix", "SetupPressureRHS", "SetupPressureRHSdiv"):
                rec["non_ortho_flags"] = int(args[1])
            if name == "SetupAdvectionVelocity":
                rec["apply_pressure_gradient"] = bool(args[2])
            if name in ("SetupPressureCorrection", "SetupPressureMatrix", "SetupPressureRHS", "SetupPressureRHSdiv"):
                rec["use_face_transform"] = bool(args[2])
            if name == "SetupAdvectionMatrix":
                rec["for_scalar"] = bool(kw.get("forPassiveScalar", False))
            if "timeStepNorm" in kw:
                rec["time_step_norm"] = bool(kw["timeStepNorm"])
            if "version" in kw:
                rec["version"] = int(kw["version"])
            self._log.append(rec)

        return op


def run_case(sim_mod, name, scalar, **settings):
    log = []
    dom = RecordingDomain(log, scalar)

    def linear_solve(A, rhs, x=None, **kw):
        log.append({"op": "linear_solve", "matrix": tag_of(A), "rhs": tag_of(rhs), "x0": tag_of(x),
                    "use_BiCG": bool(kw.get("use_BiCG")), "tol": kw.get("tol"),
                    "return_best_result": bool(kw.get("return_best_result", False)),
                    "residual_reset_step": kw.get("residual_reset_step"), "matrix_rank_deficient": kw.get("matrix_rank_deficient")})
        res = torch.arange(1.0, rhs.numel() + 1.0)          # NOT mean-free: setPressureResult shows whether the mean was removed
        return res, True

    me = types.SimpleNamespace(
        domain=dom, differentiable=False, _velocity_corrector_version=0, convergence_tol=None, total_step=0, total_time=0.0,
        advect_passive_scalar=True, exclude_advection_solve_gradients=True, exclude_pressure_solve_gradients=True,
        scipy_solve_advection=False, scipy_solve_pressure=False, pressure_time_step_normalized=False,
        _check_domain=lambda: None, _check_stop=lambda: False, linear_solve=linear_solve,
        _run_prep_fn=lambda hook, **kw: log.append({"op": "hook", "name": hook}),
        end_step=lambda time_step: log.append({"op": "end_step"}), **settings)
    setattr(me, "_Simulation__backend", RecordingBackend(log))
    # the flag word as Simulation.__init__ sets it (:739), from the class's own constant
    mode = getattr(sim_mod.Simulation, "_Simulation__NON_ORTHO_MODE")
    setattr(me, "_Simulation__non_ortho_flags", int(mode) if settings["non_orthogonal"] else 0)
    setattr(me, "_Simulation__LOG", logging.getLogger("golden"))
    ok = sim_mod.Simulation._PISO_split_step(me, 1, time_step=torch.tensor([0.05]))
    return {"name": name, "settings": {k: v for k, v in settings.items()}, "passive_scalar": scalar, "ok": bool(ok), "calls": log}


def main():
    sim_mod = G.load_reference_simulation_module()
    base = dict(corrector_steps=2, advection_use_BiCG=True, pressure_use_BiCG=False, advection_tol=None, pressure_tol=None,
                pressure_return_best_result=True, normalize_pressure_result=True)
    cases = [
        run_case(sim_mod, "channel_orthogonal", False, non_orthogonal=False, advect_non_ortho_steps=1, pressure_non_ortho_steps=1, **base),
        run_case(sim_mod, "rbc_orthogonal_scalar", True, non_orthogonal=False, advect_non_ortho_steps=1, pressure_non_ortho_steps=1, **base),
        run_case(sim_mod, "tcf_cylinder2d_nonorthogonal_1_1", False, non_orthogonal=True, advect_non_ortho_steps=1,
                 pressure_non_ortho_steps=1, **base),
        run_case(sim_mod, "cylinder3d_nonorthogonal_1_4", False, non_orthogonal=True, advect_non_ortho_steps=1,
                 pressure_non_ortho_steps=4, **base),
        run_case(sim_mod, "airfoil_nonorthogonal_2_4_bicg_pressure", False, non_orthogonal=True, advect_non_ortho_steps=2,
                 pressure_non_ortho_steps=4, **{**base, "pressure_use_BiCG": True}),
        run_case(sim_mod, "nonorthogonal_scalar_1_1", True, non_orthogonal=True, advect_non_ortho_steps=1, pressure_non_ortho_steps=1, **base),
    ]
    with open(os.path.join(OUT, "reference_split_step.json"), "w") as f:
        json.dump({"hooks": HOOKS, "cases": cases}, f, indent=1)
    for c in cases:
        print(c["name"], len(c["calls"]), "calls, ok", c["ok"])


if __name__ == "__main__":
    main()
