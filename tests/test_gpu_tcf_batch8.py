"""BASELINE config 4 at full size AND full batch -- "3D turbulent channel flow (TCF), 128x64x64, batch=8 on 1 GPU" -- through the
z-marching two-kernel BiCGStab (csrc/fg_bicgstab3d.hip: the default on this grid) and the z-marching pressure kernels:

* one native PISO step of the whole batch from the env's developing state (eight different envs, the env's body force), envs 0
  and 7 against the oracle run with the REFERENCE's iterative solvers (its BiCGStab / CG recurrences in fp64, non-orthogonal
  branch as tcf_env.py:497: velocity solve from zero) -- direct solves are out of reach at 524 288 cells in 3-D;
* the same step with the five brick kernels (FG_BICG3=0 handle): same answer, same iteration counts;
* one full ``env.step`` of the eight envs at the env's own tolerances (tcf_env.py:491) with the iteration counts reported.
Reference kernels: bicgstab_solver_kernel.cu:63-411 on the matrix of PISO_multiblock_cuda_kernel.cu:3616-3880."""
import numpy as np
import pytest
import torch

import fluidgym_amd
from oracle import piso_oracle as O
from tests.helpers import f64_twin, rel_err

pytestmark = pytest.mark.gpu

B = 8


def _native_step(env, u0, src, dt, monkeypatch, form):
    """A fresh handle of the env's grid (FG_BICG3 is read at fg_create), the env's state, one PISO step at tight tolerances."""
    from fluidgym_amd.native import NativeSolver

    if form == "five":
        monkeypatch.setenv("FG_BICG3", "0")
        monkeypatch.setenv("FG_BICG_FUSED", "0")
    else:
        monkeypatch.delenv("FG_BICG3", raising=False)
        monkeypatch.delenv("FG_BICG_FUSED", raising=False)
    old = env._domain.solver
    ns = NativeSolver([np.diff(np.asarray(e, np.float64)).astype(np.float32) for e in env._block.edges], B, fixed_faces=(2, 3))
    try:
        assert ns.advection_solver_form() == ("five" if form == "five" else "two-zmarch")
        ns.set_viscosity(float(old.viscosity))
        ns.velocity.copy_(u0)
        for f in (2, 3):
            ns.bvel[f].zero_()
        ns.set_velocity_source(src.contiguous())
        ns.copy_velocity_result_from_blocks()
        ns.set_advection_start(False)                         # non-orthogonal branch (tcf_env.py:497): velocity solve from zero
        ok, stats = ns.piso_step(dt, advection_tol=1e-7, pressure_tol=5e-9)
        assert ok, stats
        return ns.velocity.clone(), ns.pressure.clone(), stats
    finally:
        ns.close()


def test_tcf_full_batch_step_matches_the_reference_recurrences(monkeypatch):
    env = fluidgym_amd.make("TCF3D-baseline-v0", num_envs=B, use_marl=False)
    try:
        env.reset(seed=2)
        ns = env._domain.solver
        assert (ns.nx, ns.ny, ns.nz, ns.B) == (128, 64, 64, B)
        assert ns.advection_solver_form() == "two-zmarch"
        for _ in range(2):
            assert env._sim.single_step()
        blk = env._domain.getBlock(0)
        u0 = blk.velocity.clone()
        if ns.velocity_source is not None:
            src = ns.velocity_source.clone()
        else:
            # native wall-stress forcing (policy native_wall_forcing, fg_set_wall_stress_forcing): no source field is bound; the
            # body force of the state is what the reference's PRE hook would write -- uniform G_x = mean of the wall shear stresses
            assert env._native_forcing and env._sim.wall_forcing is not None
            tau_b, tau_t = env._get_wall_stress()
            src = torch.zeros_like(u0)
            src[:, 0] = (0.5 * (tau_b + tau_t)).view(-1, 1, 1, 1)
        assert float(src.abs().max()) > 0                                  # the env's body force is on
        assert float((u0[0] - u0[B - 1]).abs().max()) > 1e-3            # the envs of the batch differ
        dt = 0.25 * float(env._dt)
        vz, pz, st_z = _native_step(env, u0, src, dt, monkeypatch, "zmarch")
        v5, p5, st_5 = _native_step(env, u0, src, dt, monkeypatch, "five")
        print("TCF x 8 one PISO step, iterations [velocity, pressure0, pressure1]: z-march", st_z, "five kernels", st_5)
        assert rel_err(vz.cpu().numpy(), v5.cpu().numpy()) < 2e-5
        # envs 0 and 7 through the fp64 build of the library (brick kernels, plain recurrences) far below the fp32 tolerances
        twin = f64_twin(ns, (0, B - 1), u0)
        for f in (2, 3):
            twin.bvel[f].zero_()
        twin.set_velocity_source(src[[0, B - 1]].double().contiguous())
        twin.set_advection_start(False)
        ok64, st64 = twin.piso_step(dt, advection_tol=1e-13, pressure_tol=1e-13, max_iterations=50000)
        v64, p64 = twin.velocity.cpu().numpy(), twin.pressure.cpu().numpy()
        twin.close()
        # (the oracle's grid = the fp32 widths the library holds, promoted: fp32 path, fp64 twin and oracle see the same metrics)
        g = O.Grid(O.rectilinear_coords([np.concatenate([[0.0], np.cumsum(np.asarray(w, np.float64))]) for w in ns.widths]))
        # (round 5: 1e-13 instead of 1e-10 -- this flow's pressure right-hand side has an rms of 7e-6, and at 1e-10 the ORACLE's own
        #  pressure was only good to 1.3e-3: the fp64 build of the library, driven to 1e-12, showed the same 'error' as the fp32 path)
        opts = O.SolverOptions(direct=False, advection_tol=1e-13, pressure_tol=1e-13, non_orthogonal=True, stats={})
        for b in (0, B - 1):
            bc = {2: O.FixedBC(np.zeros(3)), 3: O.FixedBC(np.zeros(3))}
            ref = O.Domain(g, float(ns.viscosity), u0[b].cpu().numpy().astype(np.float64), np.zeros(g.shape), bc)
            ref.velocity_source = src[b].cpu().numpy().astype(np.float64)
            O.piso_split_step(ref, dt, opts)
            ev = rel_err(vz[b].cpu().numpy().astype(np.float64), ref.velocity)
            pr = ref.pressure - ref.pressure.mean()
            pg = pz[b, 0].cpu().numpy().astype(np.float64)
            ep = rel_err(pg - pg.mean(), pr)
            print(f"TCF_B8_ERR env {b}: velocity {ev:.2e} pressure {ep:.2e}; oracle iterations {opts.stats}")
            k = 0 if b == 0 else 1
            p6 = p64[k, 0] - p64[k, 0].mean()
            ev64, ep64 = rel_err(v64[k], ref.velocity), rel_err(p6, pr)
            print(f"TCF_B8_F64 env {b}: velocity {ev64:.2e} pressure {ep64:.2e} (fp64 build, iterations {st64})")
            # measured (round 5, oracle driven to 1e-13): fp32 velocity 4-5e-7, pressure 6-8e-5 (max norm); fp64 build 7-8e-10 / 2-3e-6.
            # (Rounds 3-4 read a pressure "error" of 1.3-1.7e-3 here: that was the ORACLE's pressure, stopped at 1e-10 on a right-hand
            #  side of rms 7e-6 -- the fp64 twin exposed it by showing the same figure.)  fp32 bounds = 2x measured
            assert ev64 < 1e-7 and ep64 < 1e-5, (b, ev64, ep64)
            assert ev < 3e-6 and ep < 2e-4, (b, ev, ep)
        # the env's own step on the batch (its tolerances, its hooks)
        ns.solver_counters(reset=True)
        obs, reward, term, trunc, info = env.step(env.sample_action())
        c = ns.solver_counters()
        assert torch.isfinite(reward).all() and all(torch.isfinite(v).all() for v in obs.values())
        assert c["velocity"]["mean"] >= 1 and c["velocity"]["max"] < 30 and c["pressure0"]["max"] < 100
        assert sum(v.get("unconverged", 0) for v in c.values() if isinstance(v, dict)) == 0
        print("TCF x 8 env.step iterations per solve:", {k: (v["mean"], v["max"]) for k, v in c.items() if isinstance(v, dict) and v["systems"]})
    finally:
        env.close()
