"""Call sequence and solver parameters of the reference's PISO split step, produced by the reference's OWN Python running HERE
against recording stand-ins (no reference source is copied).  The chain that runs is the reference's: its real
``Simulation.__init__`` (``pict/PISOtorch_simulation.py:489-597``) with the arguments its env families pass (copied here as data),
its ``_PISO_split_step`` (``:1431-2002``), ``linear_solve`` / ``linear_solve_GPU`` (``:1080-1181``) and the real
``_linear_solve_wrapper`` of ``pict/PISOtorch_diff.py`` (``:373-488``).  Only the compiled extension is replaced: a backend and a
domain that RECORD what is asked of them -- which operator with which non-orthogonal flag word, which ``SolveLinear`` call
(matrix, right-hand side, start vector or zeros, iteration cap, tolerance, criterion, solver kind, residual reset, best-result and
preconditioner flags), which hook, in which order.

    python tests/golden/make_golden_split_step.py        ->  tests/golden/reference_split_step.json

``tests/test_split_step_golden.py`` holds the oracles' step functions (``oracle/piso_oracle.py::piso_split_step``,
``oracle/mb_oracle.py::Domain.piso_step``) and the host's ``Simulation`` / ``MultiBlockSimulation`` settings against it.
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
        t = torch.full((n,), 0.5).as_subclass(Tagged)       # non-zero: a given start vector differs from zeros_like(rhs)
        t.tag = tag
        return t


def tag_of(x):
    return None if x is None else getattr(x, "tag", "tensor")


class RecordingDomain:
    def __init__(self, log, scalar):
        self._log, self._scalar = log, scalar
        for name in ["A", "C", "P", "velocityRHS", "scalarRHS", "pressureRHSdiv", "velocityResult", "pressureResult", "scalarResult"]:
            setattr(self, name, Tagged.make(name))

    # make_divergence_free (PISOtorch_simulation.py:1335-1355): A := 1, the velocity itself as the pressure right-hand side field
    def setA(self, v):
        self._log.append({"op": "setA", "all_ones": bool(torch.as_tensor(v).eq(1).all())})

    def setPressureRHS(self, v):
        self._log.append({"op": "setPressureRHS", "field": tag_of(v)})

    # what Simulation.__init__ and its property setters ask of a domain
    def IsInitialized(self):
        return True

    def getNumBlocks(self):
        return 1

    def getBlock(self, i):
        return types.SimpleNamespace(velocity=torch.zeros(1, dtype=torch.float32))

    def getSpatialDims(self):
        return 2

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


class _Info:
    """LinearSolverResultInfo of a converged solve (what the reporting code of the wrapper reads)"""
    converged, isFiniteResidual, finalResidual, usedIterations = True, True, 1e-9, 7

    def __getattr__(self, name):
        return 0


def run_case(sim_mod, diff_mod, name, scalar, ctor, after=None):
    """ctor: keyword arguments of the reference env's Simulation(...) call that reach this class (the env-side wrapper
    fluidgym/simulation/simulation.py passes them through, dt as time_step); after: attributes the env sets afterwards."""
    log = []
    backend = RecordingBackend(log)
    backend.Domain = RecordingDomain                       # isinstance check of the domain setter
    backend.ConvergenceCriterion = types.SimpleNamespace(NORM2_NORMALIZED="NORM2_NORMALIZED")

    def solve_linear(mat, rhs, result, maxit, tol, crit, use_bicg, rank_def, reset, transpose, print_res, best, BiCGwithPreconditioner=True):
        log.append({"op": "SolveLinear", "matrix": tag_of(mat), "rhs": tag_of(rhs),
                    "x0": None if bool(result.eq(0).all()) else tag_of(result), "dtype": str(rhs.dtype).replace("torch.", ""),
                    "max_iterations": int(maxit[0]), "tol": float(tol[0]), "criterion": str(crit), "use_BiCG": bool(use_bicg),
                    "matrix_rank_deficient": bool(rank_def), "residual_reset_step": int(reset), "transpose": bool(transpose),
                    "return_best_result": bool(best), "BiCG_with_preconditioner": bool(BiCGwithPreconditioner)})
        result.copy_(torch.arange(1.0, rhs.numel() + 1.0))    # NOT mean-free: setPressureResult shows whether the mean was removed
        return [_Info()]

    diff_mod.PISOtorch.SolveLinear = solve_linear
    sim_mod.PISOtorch, sim_mod.PISOtorch_diff = backend, diff_mod        # the names Simulation.__init__ / linear_solve_GPU resolve
    dom = RecordingDomain(log, scalar)
    sim = sim_mod.Simulation(domain=dom, **ctor)
    for k, v in (after or {}).items():
        setattr(sim, k, v)
    sim._run_prep_fn = lambda hook, **kw: log.append({"op": "hook", "name": hook})
    sim.end_step = lambda time_step: log.append({"op": "end_step", "time_step": float(torch.as_tensor(time_step).reshape(-1)[0])})
    ok = sim._PISO_split_step(1, time_step=torch.tensor([0.05]))
    n_step = len(log)
    ok_mdf = sim.make_divergence_free()                 # (PISOtorch_simulation.py:1320-1429; the cylinder envs call it after set-up)
    derived = {"linear_solve_max_iterations": sim.linear_solve_max_iterations, "solver_double_fallback": sim.solver_double_fallback,
               "preconditionBiCG": sim.preconditionBiCG, "BiCG_precondition_fallback": sim.BiCG_precondition_fallback,
               "velocity_corrector_version": sim._velocity_corrector_version, "adaptive_CFL": sim.adaptive_CFL,
               "substeps": sim.substeps, "corrector_steps": sim.corrector_steps}
    return {"name": name, "constructor": ctor, "set_after_construction": after or {}, "passive_scalar": scalar, "ok": bool(ok),
            "simulation_attributes": derived, "calls": log[:n_step], "make_divergence_free_ok": bool(ok_mdf),
            "make_divergence_free_calls": log[n_step:]}


def main():
    logging.disable(logging.CRITICAL)
    import make_golden_control as C                    # loader of the reference's real PISOtorch_diff (retry ladder pins)

    diff_mod = C.load_diff_module(lambda *a, **k: [_Info()])
    sim_mod = G.load_reference_simulation_module()
    # the Simulation(...) calls of the reference's env families, as data (cylinder_env_base.py:308-328, rbc_env_base.py:311-327,
    # airfoil_env_base.py:265-284, tcf_env.py:483-503): the arguments that reach this class; dt / rendering arguments left out
    common = dict(time_step=0.1, substeps="ADAPTIVE", corrector_steps=2, pressure_return_best_result=True, velocity_corrector="FD")
    cases = [
        run_case(sim_mod, diff_mod, "rbc", True, dict(common, adaptive_CFL=0.8, pressure_tol=1e-5, advect_non_ortho_steps=1,
                                                      pressure_non_ortho_steps=1, non_orthogonal=False)),
        run_case(sim_mod, diff_mod, "orthogonal_no_scalar", False, dict(common, non_orthogonal=False)),
        run_case(sim_mod, diff_mod, "tcf", False, dict(common, advection_use_BiCG=True, advection_tol=1e-6, pressure_tol=1e-6, adaptive_CFL=0.8,
                                                       advect_non_ortho_steps=1, pressure_non_ortho_steps=1, non_orthogonal=True),
                 after=dict(solver_double_fallback=False, preconditionBiCG=False)),
        run_case(sim_mod, diff_mod, "cylinder2d", False, dict(common, adaptive_CFL=0.8, pressure_tol=1e-5, advect_non_ortho_steps=1,
                                                              pressure_non_ortho_steps=1, non_orthogonal=True),
                 after=dict(solver_double_fallback=True, preconditionBiCG=False, BiCG_precondition_fallback=True)),
        run_case(sim_mod, diff_mod, "cylinder3d", False, dict(common, adaptive_CFL=0.8, pressure_tol=5e-7, advect_non_ortho_steps=1,
                                                              pressure_non_ortho_steps=4, non_orthogonal=True),
                 after=dict(solver_double_fallback=True, preconditionBiCG=False, BiCG_precondition_fallback=True)),
        run_case(sim_mod, diff_mod, "airfoil2d", False, dict(common, advection_tol=1e-6, pressure_tol=1e-7, advect_non_ortho_steps=2,
                                                             pressure_non_ortho_steps=4, non_orthogonal=True),
                 after=dict(solver_double_fallback=True, preconditionBiCG=False)),
        run_case(sim_mod, diff_mod, "defaults_non_orthogonal", False, dict(time_step=0.1)),
    ]
    with open(os.path.join(OUT, "reference_split_step.json"), "w") as f:
        json.dump({"hooks": HOOKS, "cases": cases}, f, indent=1)
    for c in cases:
        print(c["name"], len(c["calls"]), "calls, ok", c["ok"])


if __name__ == "__main__":
    main()
