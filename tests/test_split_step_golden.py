"""The step structure of the oracles and the start-vector policy of the host against the call sequence of the reference's OWN
``Simulation._PISO_split_step`` (``pict/PISOtorch_simulation.py:1431-2002``), recorded here by
``tests/golden/make_golden_split_step.py`` with recording stand-ins for the compiled backend and domain
(``tests/golden/reference_split_step.json``): which operator, which solve (matrix, right-hand side, start vector or none,
solver kind, best-result flag), which hook, in which order, for the solver settings of the reference's env families."""
import json
import os

import numpy as np
import pytest

from oracle import mb_oracle as MB
from oracle import piso_oracle as O
from tests.helpers import make_case
from tests.helpers_mb import split_rotated_channel

with open(os.path.join(os.path.dirname(__file__), "golden", "reference_split_step.json")) as f:
    GOLD = json.load(f)
CASES = {c["name"]: c for c in GOLD["cases"]}


def _project(calls, fields):
    """the records reduced to the fields a comparison is about (a field an oracle does not report is left out on both sides)"""
    out = []
    for r in calls:
        if r["op"] == "end_step":            # time bookkeeping of the driver, outside the oracles' step functions
            continue
        out.append(tuple([r["op"]] + [r.get(k) for k in fields.get(r["op"], ())]))
    return out


def _single_block_domain(with_scalar):
    return make_case(dims=2, n=(8, 6), fixed_axes=(1,), B=1, n_scalars=1 if with_scalar else 0).oracle_domain(0)


@pytest.mark.parametrize("case,scalar,non_orthogonal", [("channel_orthogonal", False, False), ("rbc_orthogonal_scalar", True, False)])
def test_single_block_oracle_steps_like_the_reference_orthogonal_branch(case, scalar, non_orthogonal):
    ref = CASES[case]
    dom = _single_block_domain(scalar)
    calls = []
    opts = O.SolverOptions(corrector_steps=ref["settings"]["corrector_steps"], pressure_return_best_result=ref["settings"]["pressure_return_best_result"],
                           normalize_pressure_result=ref["settings"]["normalize_pressure_result"], non_orthogonal=non_orthogonal)
    O.piso_split_step(dom, 0.05, opts, calls=calls)
    fields = {"hook": ("name",), "SetupAdvectionMatrix": ("for_scalar",), "SetupAdvectionVelocity": ("apply_pressure_gradient",),
              "linear_solve": ("matrix", "rhs", "x0", "use_BiCG", "tol", "return_best_result"), "setPressureResult": ("mean_removed",)}
    assert _project(calls, fields) == _project(ref["calls"], fields)


def test_single_block_oracle_non_orthogonal_flag_starts_the_velocity_solve_from_zero():
    """the reference's TCF env runs the non-orthogonal branch on a rectilinear grid (tcf_env.py:497): there the velocity solve
    starts from zero, everything else of the step sequence that the orthogonal oracle logs stays in the same order"""
    ref = [r for r in CASES["tcf_cylinder2d_nonorthogonal_1_1"]["calls"] if r["op"] == "linear_solve"]
    calls = []
    O.piso_split_step(_single_block_domain(False), 0.05, O.SolverOptions(non_orthogonal=True, pressure_return_best_result=True), calls=calls)
    mine = [r for r in calls if r["op"] == "linear_solve"]
    key = lambda r: (r["matrix"], r["rhs"], r["x0"], r["use_BiCG"], r["return_best_result"])
    assert [key(r) for r in mine] == [key(r) for r in ref]
    assert mine[0]["x0"] is None


@pytest.mark.parametrize("case", ["tcf_cylinder2d_nonorthogonal_1_1", "cylinder3d_nonorthogonal_1_4", "airfoil_nonorthogonal_2_4_bicg_pressure"])
def test_multi_block_oracle_steps_like_the_reference_non_orthogonal_branch(case):
    ref = CASES[case]
    st = ref["settings"]
    dom = split_rotated_channel(nx=6, ny=4, cut=3).oracle()
    n = sum(b.ncells for b in dom.blocks)
    u0 = np.vstack([np.ones(n), np.zeros(n)])
    calls = []
    dom.piso_step(u0, np.zeros(n), 0.05, corrector_steps=st["corrector_steps"], advect_non_ortho_steps=st["advect_non_ortho_steps"],
                  pressure_non_ortho_steps=st["pressure_non_ortho_steps"], calls=calls)
    fields = {"SetupAdvectionMatrix": ("non_ortho_flags", "for_scalar"), "SetupAdvectionVelocity": ("non_ortho_flags", "apply_pressure_gradient"),
              "SetupPressureMatrix": ("non_ortho_flags",), "SetupPressureRHS": ("non_ortho_flags",), "SetupPressureRHSdiv": ("non_ortho_flags",),
              "linear_solve": ("matrix", "rhs", "x0"), "setPressureResult": ("mean_removed",)}
    theirs = [r for r in ref["calls"] if r["op"] != "hook"]       # the multi-block oracle has no hook points
    assert _project(calls, fields) == _project(theirs, fields)
    assert MB.NON_ORTHO_MODE == next(r for r in ref["calls"] if r["op"] == "SetupAdvectionMatrix")["non_ortho_flags"]


def test_host_start_vector_policy_is_the_recorded_one():
    """What the host hands the native stepper: velocity solve from velocityResult in the orthogonal branch, from zero in the
    non-orthogonal one; first pressure solve of a corrector from zero in both -- unless the opt-in policies say otherwise."""
    import torch  # noqa: F401

    from fluidgym_amd.simulation.policy import get_solver_policy, set_solver_policy
    from fluidgym_amd.simulation.simulation import Simulation
    from fluidgym_amd.simulation.domain import Domain
    from tests.stub_solver import StubSolver

    def make_stub_domain():        # a Domain whose native solver is the CPU stand-in of tests/stub_solver.py (nothing touches a GPU)
        dom = Domain.__new__(Domain)
        dom.solver, dom.batch, dom.dims = StubSolver([np.ones(4, np.float32), np.ones(3, np.float32)], 1), 1, 2
        return dom

    def first(case, matrix):
        return next(r for r in CASES[case]["calls"] if r["op"] == "linear_solve" and r["matrix"] == matrix and r["rhs"] != "scalarRHS")

    assert first("channel_orthogonal", "C")["x0"] == "velocityResult" and first("rbc_orthogonal_scalar", "C")["x0"] == "velocityResult"
    assert first("tcf_cylinder2d_nonorthogonal_1_1", "C")["x0"] is None and first("airfoil_nonorthogonal_2_4_bicg_pressure", "C")["x0"] is None
    for case in CASES:
        assert first(case, "P")["x0"] is None
    assert get_solver_policy()["advection_warm_start"] is False and get_solver_policy()["pressure_warm_start"] is False
    for non_orthogonal, expect_from_result in [(False, True), (True, False)]:
        dom = make_stub_domain()
        sim = Simulation(dom, 0.1, non_orthogonal=non_orthogonal)
        assert dom.solver.advection_from_result is expect_from_result and sim.pressure_warm_start is False
    old = set_solver_policy(advection_warm_start=True)
    try:
        dom = make_stub_domain()
        Simulation(dom, 0.1, non_orthogonal=True)
        assert dom.solver.advection_from_result is True
    finally:
        set_solver_policy(**old)
