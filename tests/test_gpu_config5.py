"""BASELINE config 5's per-GPU share -- "2D airfoil ... 512x256, batch=512 across 8 GPUs" = 64 envs of 512 x 256 per GPU -- on the
single-block stand-in ``ChannelJet2D-large-v0`` (SURVEY 8d's mapping; the reference's own airfoil mesh is covered by
test_gpu_airfoil*.py / test_gpu_mb.py).  The only 2-D configuration whose working set (64 x 131 072 cells x ~20 fields = 680 MB)
exceeds the 256 MiB Infinity Cache.

* one native ``single_step`` (flux guard, adaptive CFL, advective outflow, fused PISO step) of the whole batch from a perturbed
  state, envs 0 and 63 against the oracle's direct solves;
* one full ``env.step`` of the 64 envs at the env's own tolerances: finite, per-solve iteration counts reported and > 0."""
import numpy as np
import pytest
import torch

import fluidgym_amd
from fluidgym_amd.simulation import Simulation
from oracle import piso_oracle as O
from tests.helpers import f64_twin, rel_err

pytestmark = pytest.mark.gpu

B = 64


# BASELINE config 5's share (512 x 256) and config 2 at its full size (256 x 128: the headline workload of bench.py), 64 envs each
@pytest.mark.parametrize("env_id,grid", [("ChannelJet2D-large-v0", (512, 256)), ("ChannelJet2D-v0", (256, 128))])
def test_large_channel_batch_step_matches_oracle_and_env_steps(env_id, grid):
    env = fluidgym_amd.make(env_id, num_envs=B)
    try:
        env.reset(seed=3, randomize=False)
        dom = env._domain
        blk = dom.getBlock(0)
        nx, ny = env._x, env._y
        assert (nx, ny) == grid and blk.velocity.shape == (B, 2, ny, nx)
        # perturbed state, different per env; jets blowing at different strengths
        g = torch.Generator(device="cpu").manual_seed(11)
        noise = 0.05 * torch.randn((B, 2, ny, nx), generator=g)
        u0 = (blk.velocity.cpu() + noise).contiguous()
        blk.setVelocity(u0.to(blk.velocity.device))
        act = torch.linspace(-1.0, 1.0, B).reshape(B, 1).to(blk.velocity.device)
        env._apply_action(act)
        dom.solver.reset_solver_state()
        out = blk.getBoundary("+x")
        velm = env._velm_host()
        dt = 0.01
        sim = Simulation(dom, dt=dt, substeps="ADAPTIVE", adaptive_CFL=0.8, outflow=([out], velm, 1e-5), pressure_tol=1e-7,
                         advection_tol=1e-7, pressure_return_best_result=True)
        assert sim._native_ok()
        inflow = blk.getBoundary("-x").velocity.cpu().numpy().astype(np.float64)      # [1 or B, 2, ny, 1]
        outflow0 = out.velocity.cpu().numpy().astype(np.float64)
        lo = blk.getBoundary("-y").velocity.cpu().numpy().astype(np.float64)          # [B, 2, 1, nx]
        hi = blk.getBoundary("+y").velocity.cpu().numpy().astype(np.float64)
        # the twin in the fp64 build BEFORE the fp32 step moves the boundary data: same grid, state and boundary values for envs 0 / 63
        twin = f64_twin(dom.solver, (0, B - 1), u0)
        assert sim.single_step()
        ok64, stats64, sub64 = twin.single_step(dt, 0.8, adaptive=True, outflow_faces=(1,), outflow_velm=[float(v) for v in velm], outflow_tol=1e-5,
                                                advection_tol=1e-13, pressure_tol=1e-13, max_iterations=250000)
        vel64, prs64 = twin.velocity.cpu().numpy(), twin.pressure.cpu().numpy()
        twin.close()
        vel = dom.solver.velocity.cpu().numpy().astype(np.float64)
        prs = dom.solver.pressure.cpu().numpy().astype(np.float64)
        assert np.isfinite(vel).all() and np.isfinite(prs).all()
        # the oracle's grid = the widths the library holds (fp32 values of L / nx, H / ny), promoted: fp32 path, fp64 twin and oracle then
        # see the same metrics to the last bit (exact linspace edges differ from them by 6e-8 relative, which is 1e-6 in the velocity)
        w = [np.asarray(x, np.float64) for x in dom.solver.widths[:2]]
        edges = [np.concatenate([[0.0], np.cumsum(w[0])]), -env.H / 2 + np.concatenate([[0.0], np.cumsum(w[1])])]
        grid = O.Grid(O.rectilinear_coords(edges))
        hooks = {"PRE": [lambda d, ts: O.update_advective_boundaries(d, [1], velm.astype(np.float64), ts, tol=1e-5)]}
        for b in (0, B - 1):
            bc = {0: O.FixedBC(inflow[b % inflow.shape[0]].copy()), 1: O.FixedBC(outflow0[b % outflow0.shape[0]].copy()),
                  2: O.FixedBC(lo[b].copy()), 3: O.FixedBC(hi[b].copy())}
            ref = O.Domain(grid, env._nu, u0[b].numpy().astype(np.float64), np.zeros((ny, nx)), bc)
            O.piso_adaptive_step(ref, dt, 0.8, prep_fn=hooks)
            k = 0 if b == 0 else 1
            e32 = (rel_err(vel[b], ref.velocity), rel_err(prs[b, 0], ref.pressure))
            e64 = (rel_err(vel64[k], ref.velocity), rel_err(prs64[k, 0] - prs64[k, 0].mean(), ref.pressure - ref.pressure.mean()))
            print(f"CHANNEL_{nx}x{ny}_ERR env {b}: fp32 velocity {e32[0]:.2e} pressure {e32[1]:.2e} | fp64 build velocity {e64[0]:.2e} pressure {e64[1]:.2e} (iterations {stats64}, substeps {sub64})")
            # the fp64 build of the same kernels lands on the oracle; the fp32 figures are solver tolerance (1e-7 absolute) x conditioning
            # (plain CG in double at an ABSOLUTE residual of 1e-13: what is left is that tolerance x the conditioning of the grid)
            assert e64[0] < 1e-7 and e64[1] < 1e-7, (b, e64)
            assert e32[0] < 6e-5 and e32[1] < 3e-5, (b, e32)      # measured 6e-6 .. 2.5e-5 / 3e-6 .. 1e-5: bound = 2x
        # the env's own step (25 PISO steps, its tolerances) on the batch
        solver = dom.solver
        solver.solver_counters(reset=True)
        obs, reward, term, trunc, info = env.step(env.sample_action())
        c = solver.solver_counters()
        assert torch.isfinite(reward).all() and all(torch.isfinite(v).all() for v in obs.values())
        assert c["piso_steps"] >= env.n_sim_steps
        assert c["velocity"]["mean"] > 0 and c["velocity"]["max"] < 50
        assert c["pressure0"]["max"] < 200
        print(f"{env_id} x {B} iterations per solve:", {k: (v["mean"], v["max"]) for k, v in c.items() if isinstance(v, dict) and v["systems"]})
    finally:
        env.close()
