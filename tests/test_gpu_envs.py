"""Env-level tests on the GPU: the reset()/step() contract (reference tests/envs/test_all_envs.py:52-100
check shapes / types / metric keys the same way), the native outflow-boundary + adaptive-CFL driver
against the oracle, batching semantics, get_state / set_state."""
import numpy as np
import pytest
import torch

import fluidgym_amd
from fluidgym_amd.simulation import Domain, Simulation, grids, update_advective_boundaries
from oracle import piso_oracle as O
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu

SMALL = {
    "ChannelJet2D-v0": dict(resolution_x=64, resolution_y=32),
    "RBC2D-easy-v0": dict(n_heaters=4, resolution=8),
    "RBC3D-easy-v0": dict(n_heaters=2, resolution=4, use_marl=False),       # (multi-agent by default, like the TCF ids)
    "TCFSmall3D-both-easy-v0": dict(resolution_x_z=16, resolution_y=16, step_length=0.6, use_marl=False),
    "TCFSmall3D-bottom-easy-v0": dict(resolution_x_z=16, resolution_y=16, step_length=0.6, use_marl=False),
}


@pytest.mark.parametrize("env_id", list(SMALL))
@pytest.mark.parametrize("num_envs", [None, 3])
def test_env_contract(env_id, num_envs):
    env = fluidgym_amd.make(env_id, num_envs=num_envs, randomize_initial_state=False, episode_length=3, **SMALL[env_id])
    with pytest.raises(RuntimeError, match="must be reset"):
        env.step(env._zero_action)
    with pytest.raises(ValueError, match="Seed"):
        env.reset()
    obs, info = env.reset(seed=3)
    lead = () if num_envs is None else (num_envs,)
    for k, sp in env.observation_space.items():
        assert tuple(obs[k].shape) == lead + tuple(sp.shape), k
        assert obs[k].dtype == torch.float32 and obs[k].is_cuda
    with pytest.raises(ValueError, match="Action shape"):
        env.step(torch.zeros(lead + (99,), device="cuda"))
    for i in range(3):
        obs, reward, term, trunc, info = env.step(env.sample_action())
        assert tuple(reward.shape) == lead
        assert torch.isfinite(reward).all()
        for m in env._metrics:
            assert m in info and tuple(info[m].shape) == lead
        assert term is False and trunc == (i == 2)
    with pytest.raises(RuntimeError, match="already terminated"):
        env.step(env.sample_action())
    env.close()


def test_reset_is_reproducible_and_state_roundtrip():
    env = fluidgym_amd.make("ChannelJet2D-v0", num_envs=2, resolution_x=64, resolution_y=32)
    o1, _ = env.reset(seed=11)
    a = env.sample_action()
    s0 = env.get_state()
    r1 = env.step(a)
    env.set_state(s0)
    r2 = env.step(a)
    # the replay is EXACT as in the reference (envs/fluid_env.py:1320-1363): order-independent reductions (FgDacc)
    assert torch.equal(r1[1], r2[1])
    assert torch.equal(r1[0]["velocity"], r2[0]["velocity"]) and torch.equal(r1[0]["pressure"], r2[0]["pressure"])
    o2, _ = env.reset(seed=11)
    assert torch.equal(o1["velocity"], o2["velocity"])
    env.close()


def test_native_env_glue_gives_the_torch_expressions_values(monkeypatch):
    """The schedule + observation kernels (csrc/fg_envglue.hip) against the elementwise torch expressions they replace
    (FLUIDGYM_AMD_ENV_GLUE=0): the controls and jets to the bit (so the fields are the same to the bit), the means and the reward
    to fp32 rounding of another summation order."""
    kw = dict(resolution_x=64, resolution_y=32, randomize_initial_state=False, num_envs=3)
    envN = fluidgym_amd.make("ChannelJet2D-v0", **kw)
    envN.reset(seed=5)
    monkeypatch.setenv("FLUIDGYM_AMD_ENV_GLUE", "0")      # (read when the env builds its domain: at its first reset)
    envT = fluidgym_amd.make("ChannelJet2D-v0", **kw)
    envT.reset(seed=5)
    assert envN._native_glue and not envT._native_glue
    gen = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(3):
        a = torch.rand(3, 1, device="cuda", generator=gen) * 2 - 1
        oN, rN, tN, uN, iN = envN.step(a)
        oT, rT, tT, uT, iT = envT.step(a)
        assert torch.equal(envN._block.velocity, envT._block.velocity) and torch.equal(envN._block.pressure, envT._block.pressure)
        assert torch.equal(envN._current_action, envT._current_action) and torch.equal(envN._jets, envT._jets)
        assert torch.equal(oN["velocity"], oT["velocity"]) and torch.equal(oN["pressure"], oT["pressure"])
        assert oN["velocity"].shape == oT["velocity"].shape and rN.shape == rT.shape
        torch.testing.assert_close(rN, rT, rtol=2e-6, atol=1e-9)
        for k in iT:
            assert iN[k].shape == iT[k].shape
            torch.testing.assert_close(iN[k], iT[k], rtol=2e-6, atol=1e-9)
        assert (tN, uN) == (tT, uT)
    envN.close()
    envT.close()


def test_two_lanes_on_two_streams_are_two_independent_shards():
    """``ParallelFluidEnv(lanes=2)``: two 3-env batches of one rank stepped concurrently by two host threads on two HIP streams
    (envs/parallel_env.py, "Lanes") give, bit for bit, what the two batches give stepped alone one after the other (seeds ``seed`` and
    ``seed + 1``: lane l acts as virtual rank l), several steps in a row -- stream order between the caller's stream, the lanes' streams and
    the concatenation included.  The multi-block cluster solvers refuse lanes."""
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    kw = dict(resolution_x=64, resolution_y=32)
    penv = ParallelFluidEnv("ChannelJet2D-v0", num_envs=6, lanes=2, **kw)
    assert [e.num_envs for e in penv.lane_envs] == [3, 3]
    obs, infos = penv.reset(seed=21, randomize=True)
    plain = [fluidgym_amd.make("ChannelJet2D-v0", num_envs=3, **kw) for _ in range(2)]
    ref = [e.reset(seed=21 + r, randomize=True) for r, e in enumerate(plain)]
    for k in obs:
        assert torch.equal(obs[k], torch.cat([r[0][k] for r in ref]))
    assert len(infos) == 6
    gen = torch.Generator(device="cuda").manual_seed(3)
    for step in range(4):
        a = torch.rand(6, 1, device="cuda", generator=gen) * 2 - 1
        o, r, term, trunc, info = penv.step(a)
        outs = [e.step(a[3 * i: 3 * i + 3]) for i, e in enumerate(plain)]
        for k in o:
            assert torch.equal(o[k], torch.cat([x[0][k] for x in outs])), (step, k)
        assert torch.equal(r, torch.cat([x[1] for x in outs])) and r.shape == (6,)
        assert term == [False] * 6 and trunc == [bool(outs[0][3])] * 6 and len(info) == 6
        for i in range(6):
            assert torch.equal(info[i]["wall_shear"], outs[i // 3][4]["wall_shear"][i % 3])
        for e, q in zip(penv.lane_envs, plain):
            assert torch.equal(e._block.velocity, q._block.velocity) and torch.equal(e._block.pressure, q._block.pressure)
    assert tuple(penv.sample_action().shape) == (6, 1)
    penv.close()
    for e in plain:
        e.close()
    with pytest.raises(ValueError, match="single-block solver path"):
        ParallelFluidEnv("CylinderJet2D-easy-v0", num_envs=2, lanes=2)


def test_batched_env_equals_independent_envs():
    """Env b of a batch evolves exactly as if it were alone (no cross-talk through the batched solvers)."""
    kw = dict(resolution_x=64, resolution_y=32, randomize_initial_state=False)
    envB = fluidgym_amd.make("ChannelJet2D-v0", num_envs=2, **kw)
    env1 = fluidgym_amd.make("ChannelJet2D-v0", num_envs=1, **kw)
    envB.reset(seed=0)
    env1.reset(seed=0)
    acts = torch.tensor([[0.7], [-0.4]], device="cuda")
    for _ in range(2):
        oB, rB, *_ = envB.step(acts)
        o1, r1, *_ = env1.step(acts[1:2])
    assert torch.allclose(oB["velocity"][1], o1["velocity"][0], rtol=2e-4, atol=2e-5)
    assert torch.allclose(rB[1], r1[0], rtol=2e-4, atol=1e-6)
    assert not torch.allclose(oB["velocity"][0], oB["velocity"][1])
    envB.close()
    env1.close()


@pytest.mark.parametrize("driver", ["native", "python_hooks"])
def test_channel_driver_with_outflow_matches_oracle(driver):
    """Simulation.single_step (flux guard, adaptive CFL, native advective outflow + flux re-balancing, fused
    PISO step) against the oracle's restatement of the same sequence, 6 steps, per-env different inflow."""
    nx, ny, L, H, nu, dt, B = 48, 24, 6.0, 2.0, 0.02, 0.06, 2
    edges = [np.linspace(0, L, nx + 1), np.linspace(-H / 2, H / 2, ny + 1)]
    dom = Domain(2, torch.tensor([nu]), batch=B)
    blk = dom.CreateBlock(grids.vertex_grid(edges))
    blk.CloseBoundary("-x")
    blk.CloseBoundary("-y")
    dom.PrepareSolve()
    rng = np.random.default_rng(0)
    yc = 0.5 * (edges[1][1:] + edges[1][:-1])
    inflow = np.zeros((B, 2, ny, 1))
    for b in range(B):
        inflow[b, 0, :, 0] = (1.0 + 0.3 * b) * 1.5 * (1 - (2 * yc / H) ** 2)
    u0 = np.broadcast_to(inflow, (B, 2, ny, nx)).copy() + 0.05 * rng.standard_normal((B, 2, ny, nx))
    blk.setVelocity(torch.from_numpy(u0).float())
    blk.getBoundary("-x").setVelocity(torch.from_numpy(inflow).float())
    out = blk.getBoundary("+x")
    out.setVelocity(torch.from_numpy(inflow).float())
    dom.solver.reset_solver_state()
    velm = np.array([1.0, 0.0], dtype=np.float32)

    def pre(domain, time_step, **kw):
        update_advective_boundaries(domain, [out], velm, time_step, tol=1e-5)

    if driver == "native":  # whole single_step inside fg_single_step
        sim = Simulation(dom, dt=dt, substeps="ADAPTIVE", adaptive_CFL=0.5, outflow=([out], velm, 1e-5), pressure_tol=1e-7,
                         advection_tol=1e-7, pressure_return_best_result=True)
        assert sim._native_ok()
    else:  # interpreter-driven: hook closure + fused fg_piso_step per substep
        sim = Simulation(dom, dt=dt, substeps="ADAPTIVE", adaptive_CFL=0.5, prep_fn={"PRE": [pre]}, pressure_tol=1e-7,
                         advection_tol=1e-7, pressure_return_best_result=True)
        assert not sim._native_ok()
    g = O.Grid(O.rectilinear_coords(edges))
    doms = []
    for b in range(B):
        bc = {0: O.FixedBC(inflow[b].copy()), 1: O.FixedBC(inflow[b].copy()), 2: O.FixedBC(np.zeros(2)), 3: O.FixedBC(np.zeros(2))}
        doms.append(O.Domain(g, nu, u0[b].astype(np.float32).astype(np.float64), np.zeros((ny, nx)), bc))
    hooks = {"PRE": [lambda d, ts: O.update_advective_boundaries(d, [1], velm.astype(np.float64), ts, tol=1e-5)]}
    n_sub = []
    for step in range(6):
        assert sim.single_step()
        n_sub.append(sim.substep_count)
        for d in doms:
            O.piso_adaptive_step(d, dt, 0.5, prep_fn=hooks)
    assert max(n_sub) >= 2, "the case should exercise adaptive substepping"
    vel = dom.solver.velocity.cpu().numpy().astype(np.float64)
    bv = out.velocity.cpu().numpy().astype(np.float64)
    for b in range(B):
        assert rel_err(vel[b], doms[b].velocity) < 2e-4
        assert rel_err(bv[b], np.asarray(doms[b].bvel(1))) < 2e-4
        assert abs(O.boundary_flux_balance(doms[b])) < 1e-6
    fb = dom.GetBoundaryFluxBalance().cpu().numpy()
    assert np.abs(fb).max() < 1e-5


def test_unbalanced_boundary_flux_is_rejected():
    env = fluidgym_amd.make("ChannelJet2D-v0", num_envs=1, resolution_x=64, resolution_y=32, randomize_initial_state=False)
    env.reset(seed=0)
    env._block.getBoundary("-x").velocity.mul_(1.5)  # break the inflow/outflow balance behind the env's back
    env._sim.prep_fn = {}
    with pytest.raises(RuntimeError, match="not balanced"):
        env._sim.single_step()
    env.close()


def test_rbc_heating_drives_convection():
    env = fluidgym_amd.make("RBC2D-easy-v0", num_envs=2, n_heaters=4, resolution=8, randomize_initial_state=False,
                            step_length=0.5)
    env.reset(seed=1)
    nus = []
    for _ in range(4):
        _, reward, _, _, info = env.step(torch.zeros(2, 4, 1, device="cuda"))  # (n_heaters, 1) per env, rbc_env_2d.py:112-129
        nus.append(info["nusselt"].cpu().numpy())
    assert np.isfinite(nus).all()
    T = env._block.passiveScalar
    assert float(T.min()) > -0.2 and float(T.max()) < 1.9
    env.close()


@pytest.mark.parametrize("env_id", ["RBC2D-easy-v0", "RBC3D-easy-v0"])
def test_rbc_observations_follow_the_reference_resampling_path(env_id):
    """Observations = fields resampled to the render grid (compiled-kernel corner rule, 16 fill passes) read at the
    integer sensor positions, exactly the indexing of rbc_env_2d.py:175-194 / rbc_env_3d.py:291-330."""
    from oracle import resample_oracle as R

    env = fluidgym_amd.make(env_id, num_envs=2, randomize_initial_state=True, **SMALL[env_id])
    env.reset(seed=5)
    obs, *_ = env.step(env.sample_action())
    d = env._ndims
    edges = env._block.edges
    oshape = env.render_shape[:d]
    sl = env._sensor_locations.cpu().numpy()
    nsx, nsy = env._n_sensors_x, env._n_sensors_y
    for b in range(2):
        T = R.resample_to_uniform(env._block.passiveScalar[b].cpu().numpy(), edges, oshape, 16, corners_3d_quirk=True)[0]
        u = R.resample_to_uniform(env._block.velocity[b].cpu().numpy(), edges, oshape, 16, corners_3d_quirk=True)
        if d == 2:
            Ts = T[sl[1], sl[0]].reshape(nsx, nsy).T
            us = np.transpose(np.transpose(u, (1, 2, 0))[sl[1], sl[0], :].reshape(nsx, nsy, 2), (2, 1, 0))
        else:
            Ts = np.transpose(T[sl[2], sl[1], sl[0]].reshape(nsx, nsy, nsx), (2, 1, 0))
            us = np.transpose(np.transpose(u, (1, 2, 3, 0))[sl[2], sl[1], sl[0], :].reshape(nsx, nsy, nsx, 3), (3, 2, 1, 0))
        assert np.abs(obs["temperature"][b].cpu().numpy() - Ts).max() < 1e-5 * max(1.0, np.abs(Ts).max())
        assert np.abs(obs["velocity"][b].cpu().numpy() - us).max() < 1e-5 * max(1e-3, np.abs(us).max())
    env.close()


def test_tcf_units_actions_and_observations_follow_the_reference():
    """tcf_env.py: unit conversions (:246-265), _action_to_control (:521-547), the y+ = 15 sensing plane (:343-352),
    fluctuation-velocity observations (:646-677) and the both-walls stacking / sign flip (:1143-1180)."""
    env = fluidgym_amd.make("TCFSmall3D-both-easy-v0", num_envs=2, randomize_initial_state=False, resolution_x_z=16,
                            resolution_y=16, step_length=0.6, use_marl=False)
    re_cl = (180 / 0.116) ** (1 / 0.88)
    assert abs(env._nu - 1 / re_cl) < 1e-12 and abs(env._u_wall - 180 / re_cl) < 1e-12
    assert abs(env.step_length - 0.6 * env._nu / env._u_wall ** 2) < 1e-12 and abs(env.dt - env.step_length / 10) < 1e-12
    assert env._grid_refinement_strength == 2 and env._y == 16
    env.reset(seed=2)
    ycen = env._y_centers.cpu().numpy()
    y15 = -1 + 15.0 / (env._u_wall / env._nu)
    assert env._y_obs_bottom_idx == int(np.abs(ycen - y15).argmin()) and env._y_obs_top_idx == 16 - env._y_obs_bottom_idx
    a = env.sample_action()
    assert tuple(a.shape) == (2, env.n_agents, 1) and env.n_agents == 2 * 8 * 8
    obs, reward, _, _, info = env.step(a)
    # control on the walls: zero mean, |v| <= u_tau, top wall = minus the second half of the agents
    ab = a.reshape(2, 2, 8, 8)
    for half, plate, sign in ((0, env._bottom_plate, 1.0), (1, env._top_plate, -1.0)):
        x = ab[:, half] - ab[:, half].mean(dim=(1, 2), keepdim=True)
        x = env._u_wall * x / torch.clamp(x.abs(), min=1.0)
        x = x - x.mean(dim=(1, 2), keepdim=True)
        v = sign * x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).transpose(1, 2)
        bv = plate.velocity
        assert torch.allclose(bv[:, 1, :, 0, :], v, atol=1e-7) and float(bv[:, 0].abs().max()) == 0.0
        assert float(bv[:, 1].sum().abs()) < 1e-4
    # observation: u' = u - <u>_V on the two planes, pressure alongside
    u, p, cs = env._block.velocity, env._block.pressure, env._cell_size
    up = u - (u * cs).sum(dim=(2, 3, 4), keepdim=True) / cs.sum()
    jb, jt = env._y_obs_bottom_idx, min(env._y_obs_top_idx, 15)
    assert tuple(obs["velocity"].shape) == (2, 2, 2, 16, 16) and tuple(obs["pressure"].shape) == (2, 2, 16, 16)
    assert torch.allclose(obs["velocity"][:, 0], up[:, :2, :, jb, :]) and torch.allclose(obs["velocity"][:, 1], up[:, :2, :, jt, :])
    assert torch.allclose(obs["pressure"][:, 1], p[:, 0, :, jt, :])
    assert set(info) >= {"wall_stress", "wall_stress_bottom", "wall_stress_top"}
    assert torch.allclose(reward, 1 - info["wall_stress"])  # tau_ref = 1 without domain statistics
    env.close()


MARL = {
    "RBC2D-easy-v0": dict(n_heaters=4, resolution=8, local_obs_window=3, local_reward_weight=0.2),
    "RBC3D-easy-v0": dict(n_heaters=2, resolution=4, local_obs_window=1, local_reward_weight=0.2),
    "TCFSmall3D-bottom-easy-v0": dict(resolution_x_z=16, resolution_y=16, step_length=0.6, local_obs_window=3, local_reward_weight=0.0),
    "TCFSmall3D-both-easy-v0": dict(resolution_x_z=16, resolution_y=16, step_length=0.6, local_obs_window=1, local_reward_weight=0.0),
}


@pytest.mark.parametrize("env_id", list(MARL))
@pytest.mark.parametrize("num_envs", [None, 2])
def test_multi_agent_contract(env_id, num_envs):
    """use_marl=True: per-agent action (1,), observations [n_agents, *per-agent space], rewards [n_agents]
    (reference fluid_env.py:249-251, 787-788; rbc_env_base.py:613-636; tcf_env.py:994-1010)."""
    env = fluidgym_amd.make(env_id, num_envs=num_envs, use_marl=True, randomize_initial_state=False, episode_length=2,
                            **MARL[env_id])
    lead = () if num_envs is None else (num_envs,)
    assert tuple(env.action_space.shape) == (1,)
    obs, _ = env.reset(seed=1)
    for k, sp in env.observation_space.items():
        assert tuple(obs[k].shape) == lead + (env.n_agents,) + tuple(sp.shape), k
    a = env.sample_action()
    assert tuple(a.shape) == lead + (env.n_agents, 1)
    obs, reward, term, trunc, info = env.step(a)
    assert tuple(reward.shape) == lead + (env.n_agents,) and torch.isfinite(reward).all()
    assert tuple(info["global_reward"].shape) == lead
    if env_id.startswith("TCF"):
        assert torch.allclose(reward, info["global_reward"].unsqueeze(-1).expand_as(reward))
    else:  # RBC: weighted sum of local and global rewards, local = nu_ref - local Nusselt of the agent's window
        w = MARL[env_id]["local_reward_weight"]
        local = env._get_local_rewards()
        g = info["global_reward"].reshape(-1, 1)
        assert torch.allclose(reward.reshape(local.shape), w * local + (1 - w) * g, atol=1e-6)
    env.close()


# ---- the reference's own env tests (tests/env_utils/test_fluid_env.py, tests/envs/test_all_envs.py), restated for the
# ids that are built here; grids shrunk through kwargs so the whole file stays within seconds -----------------------------
REDUCED = {"ChannelJet": dict(resolution_x=64, resolution_y=32), "RBC2D": dict(n_heaters=4, resolution=8, local_obs_window=3),
           # (the wide 3-D ids keep their aspect ratio of 2: four heaters so that the render grid is 13 high like the others)
           "RBC3D-wide": dict(n_heaters=4, resolution=4, local_obs_window=1),
           "RBC3D": dict(n_heaters=2, resolution=4, local_obs_window=1), "TCF": dict(resolution_x_z=16, resolution_y=16, resolution_x=None, resolution_z=None)}


def _built_ids():
    return [i for i in fluidgym_amd.registry.ids if not i.startswith(("Cylinder", "Airfoil", "Toy"))]


def _reduced(env_id):
    for k, v in REDUCED.items():
        if env_id.startswith(k):
            return dict(v, randomize_initial_state=False)
    raise KeyError(env_id)


def test_sampling_before_reset():
    env = fluidgym_amd.make("RBC2D-easy-v0", **_reduced("RBC2D-easy-v0"))
    with pytest.raises(RuntimeError) as excinfo:
        env.step(env.sample_action())
    assert "Environment must be seeded before sampling actions" in str(excinfo.value)


def test_step_before_reset():
    env = fluidgym_amd.make("RBC2D-easy-v0", **_reduced("RBC2D-easy-v0"))
    with pytest.raises(RuntimeError) as excinfo:
        env.step(torch.zeros(env.action_space.shape, device=env.cuda_device))
    assert "Environment must be reset before stepping" in str(excinfo.value)


def _check_obs(env, obs, marl):
    assert isinstance(env.observation_space, fluidgym_amd.spaces.Dict)
    for key, space in env.observation_space.spaces.items():
        assert key in obs, f"Observation missing key: {key}"
        o = obs[key][0] if marl else obs[key]
        assert isinstance(o, torch.Tensor) and tuple(o.shape) == tuple(space.shape), key


def _check_action(env, action, marl):
    a = action[0] if marl else action
    assert isinstance(env.action_space, fluidgym_amd.spaces.Box)
    assert isinstance(a, torch.Tensor) and tuple(a.shape) == tuple(env.action_space.shape)


@pytest.mark.parametrize("env_id", _built_ids())
def test_env_sarl(env_id):
    env = fluidgym_amd.make(env_id, use_marl=False, **_reduced(env_id))
    env.seed(42)
    obs, info = env.reset()
    _check_action(env, env.sample_action(), marl=False)
    obs, reward, terminated, truncated, info = env.step(env.sample_action())
    _check_obs(env, obs, marl=False)
    assert isinstance(reward, torch.Tensor) and isinstance(terminated, bool) and isinstance(truncated, bool)
    assert isinstance(info, dict)
    for metric in env.metrics:
        assert metric in info and isinstance(info[metric], torch.Tensor), metric
    env.close()


@pytest.mark.parametrize("env_id", _built_ids())
def test_env_marl(env_id):
    try:
        env = fluidgym_amd.make(env_id, use_marl=True, **_reduced(env_id))
    except ValueError:
        return  # env does not support MARL
    env.seed(42)
    obs, info = env.reset()
    _check_action(env, env.sample_action(), marl=True)
    obs, reward, terminated, truncated, info = env.step(env.sample_action())
    _check_obs(env, obs, marl=True)
    assert reward.shape[0] == env.n_agents
    assert "global_reward" in info and isinstance(info["global_reward"], torch.Tensor)
    env.close()


def test_forced_3d_channel_adaptive_trajectory_matches_oracle():
    """TCF-like setting end to end: wall-refined 3-D channel, periodic x/z, dynamic forcing in the PRE hook
    (envs/tcf/grid.py:147-176), adaptive CFL substeps; 4 env-level steps against the oracle with the same hook."""
    nx, ny, nz, B, nu, dt, cfl = 16, 12, 8, 2, 0.02, 0.05, 0.3
    yw = grids.tcf_y_weights(N=1, ny_half=ny // 2)
    edges = [np.linspace(-1.5, 1.5, nx + 1), np.asarray(grids.lerp_edges(-1.0, 1.0, yw), np.float64), np.linspace(-0.8, 0.8, nz + 1)]
    edges = [np.concatenate([[e[0]], e[0] + np.cumsum(np.diff(e).astype(np.float32).astype(np.float64))]) for e in edges]
    dom = Domain(3, torch.tensor([nu]), batch=B)
    blk = dom.CreateBlock(grids.vertex_grid(edges))
    blk.CloseBoundary("-y")
    dom.PrepareSolve()
    rng = np.random.default_rng(9)
    yc = 0.5 * (edges[1][1:] + edges[1][:-1])
    u0 = np.zeros((B, 3, nz, ny, nx))
    u0[:, 0] = (1.5 * (1 - yc ** 2))[None, None, :, None] * np.array([1.0, 1.3])[:, None, None, None]
    u0 += 0.05 * rng.standard_normal(u0.shape)
    blk.setVelocity(torch.from_numpy(u0).float())
    blk.setVelocitySource(torch.zeros(1, 3, nz, ny, nx))
    dom.solver.reset_solver_state()
    d_wall = (1.0 + yc[0], 1.0 - yc[-1])

    def forcing(domain, **kw):
        mean_u = blk.velocity[:, 0].mean(dim=(1, 3))
        G = 0.5 * nu * (mean_u[:, 0] / d_wall[0] + mean_u[:, -1] / d_wall[1])
        blk.velocitySource[:, 0] = G.view(-1, 1, 1, 1)

    sim = Simulation(dom, dt=dt, substeps="ADAPTIVE", adaptive_CFL=cfl, prep_fn={"PRE": [forcing]}, pressure_tol=1e-7,
                     advection_tol=1e-7, pressure_return_best_result=True)
    sim.make_divergence_free()
    g = O.Grid(O.rectilinear_coords(edges))
    start = dom.solver.velocity.cpu().numpy().astype(np.float64)
    refs = []
    for b in range(B):
        bc = {2: O.FixedBC(np.zeros(3)), 3: O.FixedBC(np.zeros(3))}
        r = O.Domain(g, nu, start[b], np.zeros((nz, ny, nx)), bc)
        r.velocity_source = np.zeros((3, nz, ny, nx))
        refs.append(r)

    def forcing_ref(d, ts):
        mu = d.velocity[0].mean(axis=(0, 2))
        d.velocity_source[0] = 0.5 * nu * (mu[0] / d_wall[0] + mu[-1] / d_wall[1])

    subs = []
    for step in range(4):
        assert sim.single_step()
        subs.append(sim.substep_count)
        for r in refs:
            O.piso_adaptive_step(r, dt, cfl, prep_fn={"PRE": [forcing_ref]})
    assert max(subs) >= 2
    vel = dom.solver.velocity.cpu().numpy().astype(np.float64)
    for b in range(B):
        assert rel_err(vel[b], refs[b].velocity) < 2e-4
