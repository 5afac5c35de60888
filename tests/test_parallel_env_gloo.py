"""ParallelFluidEnv's distributed plumbing on CPU: world_size 2, gloo backend, a toy env.

The real envs need the GPU; what is covered here is what differs at N > 1: sharding of the env batch,
the action broadcast, the packed observation/reward all_gather, command fan-out (seed / reset / step /
close) in both SPMD mode and the serve() worker loop, and that results equal a single-process run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from fluidgym_amd import spaces


class ToyEnv:
    """Deterministic pure-torch stand-in with the attributes ParallelFluidEnv touches."""

    n_agents = 1
    use_marl = False

    def __init__(self, num_envs=1, cuda_device=None, gain=2.0, **kw):
        self._num_envs = num_envs
        self.gain = gain
        self.action_space = spaces.Box(low=-1, high=1, shape=(3,), dtype=np.float32)
        self.observation_space = spaces.Dict({
            "a": spaces.Box(low=-np.inf, high=np.inf, shape=(2, 2), dtype=np.float32),
            "b": spaces.Box(low=-np.inf, high=np.inf, shape=(5,), dtype=np.float32),
        })
        self._zero_action = torch.zeros(num_envs, 3)
        self.state = torch.zeros(num_envs, 3)
        self.mode = "train"
        self._seed = 0

    def seed(self, s):
        self._seed = s

    def _obs(self):
        return {"a": self.state[:, :2].unsqueeze(-1).expand(-1, 2, 2).contiguous(),
                "b": torch.cat([self.state, self.state[:, :2] * 3], dim=1)}

    def reset(self, seed=None, randomize=None):
        if seed is not None:
            self._seed = seed
        g = torch.Generator().manual_seed(self._seed)
        self.state = torch.randn(self._num_envs, 3, generator=g)
        return self._obs(), {"init_sum": self.state.sum(dim=1), "label": "not numeric: stays on its shard"}

    def step(self, action):
        assert action.shape == (self._num_envs, 3)
        self.state = self.state + self.gain * action
        # env 0 of every shard "terminates" once its state sum exceeds 5: per-env flags must survive the gather
        term = self.state.sum(dim=1) > 5.0
        return self._obs(), self.state.sum(dim=1), term, False, {"m": self.state.mean(dim=1), "v": self.state[:, :2]}

    def sample_action(self):
        g = torch.Generator().manual_seed(self._seed)
        return torch.rand(self._num_envs, 3, generator=g)

    def train(self):
        self.mode = "train"

    def val(self):
        self.mode = "val"

    def test(self):
        self.mode = "test"

    def load_initial_domain(self, idx, mode=None):
        self.loaded = (idx, mode)
        self.reset(seed=1000 + idx)

    def close(self):
        pass


BIG_SEED = 2 ** 32 - 1


def _register():
    import fluidgym_amd

    if "ToyCPU-v0" not in fluidgym_amd.registry.ids:
        fluidgym_amd.register("ToyCPU-v0", ToyEnv, {"gain": 2.0})


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    _register()
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    penv = ParallelFluidEnv("ToyCPU-v0", num_envs=4, backend="gloo")
    assert penv.world == world and penv.num_envs == 4 and penv.local_env._num_envs == 2
    actions = torch.arange(12, dtype=torch.float32).reshape(4, 3) * 0.1
    if mode == "serve" and not penv.is_driver:
        penv.serve()
        q.put((rank, "served", penv.local_env.mode, penv.local_env.loaded))
        return
    from fluidgym_amd.types import EnvMode

    penv.seed(BIG_SEED)    # >= 2**31: must survive the command header (np.random.SeedSequence / getrandbits(32) give such seeds)
    sampled = penv.sample_action()          # a collective: hung in serve() mode before the command existed
    penv.load_initial_domain(3, EnvMode.TEST)
    loaded = penv.local_env.loaded
    obs0, infos0 = penv.reset(seed=11)
    penv.val()
    out = penv.step(actions if penv.is_driver else None)
    out2 = penv.step(actions * 2 if penv.is_driver else None)
    mode_seen = penv.local_env.mode
    penv.close()
    if penv.is_driver:
        extra = {"sampled": sampled.numpy().copy(), "loaded": loaded, "n_infos0": len(infos0),
                 "init_sum": np.array([float(i["init_sum"]) for i in infos0]), "term2": out2[2], "trunc2": out2[3],
                 "info_m": np.array([float(i["m"]) for i in out2[4]]), "info_v": np.stack([i["v"].numpy() for i in out2[4]])}
        q.put((rank, {k: v.numpy().copy() for k, v in obs0.items()}, out[1].numpy().copy(), out2[1].numpy().copy(),
               {k: v.numpy().copy() for k, v in out2[0].items()}, mode_seen, extra))
    else:
        q.put((rank, "spmd", mode_seen, loaded))


def _expected():
    envs = [ToyEnv(num_envs=2), ToyEnv(num_envs=2)]
    actions = torch.arange(12, dtype=torch.float32).reshape(4, 3) * 0.1
    res0 = [e.reset(seed=11 + r) for r, e in enumerate(envs)]
    obs0 = [x[0] for x in res0]
    r1 = [e.step(actions[2 * r: 2 * r + 2])[1] for r, e in enumerate(envs)]
    o2r2 = [e.step(2 * actions[2 * r: 2 * r + 2]) for r, e in enumerate(envs)]
    cat = lambda ds: {k: torch.cat([d[k] for d in ds]) for k in ds[0]}
    extra = {"term2": torch.cat([x[2] for x in o2r2]).tolist(), "info_m": torch.cat([x[4]["m"] for x in o2r2]).numpy(),
             "info_v": torch.cat([x[4]["v"] for x in o2r2]).numpy()}
    extra["init_sum"] = torch.cat([x[1]["init_sum"] for x in res0]).numpy()
    samp = []
    for r in range(2):
        e = ToyEnv(num_envs=2)
        e.seed(BIG_SEED + r)
        samp.append(e.sample_action())
    extra["sampled"] = torch.cat(samp).numpy()
    return cat(obs0), torch.cat(r1), torch.cat([x[1] for x in o2r2]), cat([x[0] for x in o2r2]), extra


@pytest.mark.parametrize("mode", ["spmd", "serve"])
def test_two_rank_gloo_matches_single_process(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    drv = [r for r in results if r[0] == 0][0]
    other = [r for r in results if r[0] == 1][0]
    obs0, r1, r2, o2, exp = _expected()
    for k in obs0:
        assert np.allclose(drv[1][k], obs0[k].numpy())
        assert np.allclose(drv[4][k], o2[k].numpy())
    assert np.allclose(drv[2], r1.numpy()) and np.allclose(drv[3], r2.numpy())
    assert drv[5] == "val" and other[2] == "val"  # mode command reached every shard
    got = drv[6]
    # per-env terminated flags and info entries of EVERY shard (reference parallel_env.py:276-287), not the driver's copy
    assert got["term2"] == exp["term2"] and any(got["term2"]) and not all(got["term2"])
    assert got["trunc2"] == [False] * 4 and got["n_infos0"] == 4
    assert np.allclose(got["init_sum"], exp["init_sum"])          # reset infos per ENV, gathered from every shard (parallel_env.py:222-231)
    assert np.allclose(got["info_m"], exp["info_m"]) and np.allclose(got["info_v"], exp["info_v"])
    assert np.allclose(got["sampled"], exp["sampled"])           # each shard sampled from its own generator (seed + rank)
    from fluidgym_amd.types import EnvMode
    assert got["loaded"] == (3 * 2 + 0, EnvMode.TEST) and other[3] == (3 * 2 + 1, EnvMode.TEST)   # idx and MODE reach the workers


def test_single_process_world_of_one():
    _register()
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    penv = ParallelFluidEnv("ToyCPU-v0", num_envs=3, backend="gloo")
    obs, _ = penv.reset(seed=1)
    assert obs["b"].shape == (3, 5)
    with pytest.raises(ValueError, match="Expected action batch size"):
        penv.step(torch.zeros(2, 3))
    o, r, term, trunc, info = penv.step(torch.ones(3, 3))
    assert r.shape == (3,) and len(term) == 3
    penv.close()


def test_two_lanes_of_one_rank_are_the_two_ranks_of_a_world_of_two():
    """``lanes=2`` at world size 1 (two sub-shards of the rank stepped by two threads, on the GPU on two HIP streams) runs the envs of
    ``world=2``: lane l acts as virtual rank l for seeds, sampled actions and initial-domain indices, and the results come back
    concatenated in lane order."""
    _register()
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv
    from fluidgym_amd.types import EnvMode

    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    penv = ParallelFluidEnv("ToyCPU-v0", num_envs=4, backend="gloo", lanes=2)
    assert penv.world == 1 and penv.num_envs == 4 and [e._num_envs for e in penv.lane_envs] == [2, 2]
    actions = torch.arange(12, dtype=torch.float32).reshape(4, 3) * 0.1
    penv.seed(BIG_SEED)
    sampled = penv.sample_action()
    penv.load_initial_domain(3, EnvMode.TEST)
    assert [e.loaded for e in penv.lane_envs] == [(3 * 2 + 0, EnvMode.TEST), (3 * 2 + 1, EnvMode.TEST)]
    obs0, infos0 = penv.reset(seed=11)
    penv.val()
    assert [e.mode for e in penv.lane_envs] == ["val", "val"]
    out = penv.step(actions)
    out2 = penv.step(actions * 2)
    e_obs0, e_r1, e_r2, e_o2, exp = _expected()
    for k in e_obs0:
        assert torch.equal(obs0[k], e_obs0[k]) and torch.equal(out2[0][k], e_o2[k])
    assert torch.equal(out[1], e_r1) and torch.equal(out2[1], e_r2)
    assert out2[2] == exp["term2"] and out2[3] == [False] * 4 and len(infos0) == 4
    assert np.allclose([float(i["init_sum"]) for i in infos0], exp["init_sum"])
    assert np.allclose([float(i["m"]) for i in out2[4]], exp["info_m"])
    assert np.allclose(np.stack([i["v"].numpy() for i in out2[4]]), exp["info_v"])
    assert np.allclose(sampled.numpy(), exp["sampled"])
    penv.close()
    with pytest.raises(ValueError, match="not divisible by lanes"):
        ParallelFluidEnv("ToyCPU-v0", num_envs=3, backend="gloo", lanes=2)


def _lanes_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    _register()
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    penv = ParallelFluidEnv("ToyCPU-v0", num_envs=8, backend="gloo", lanes=2)
    assert penv.world == 2 and [e._num_envs for e in penv.lane_envs] == [2, 2]
    actions = torch.arange(24, dtype=torch.float32).reshape(8, 3) * 0.05
    penv.seed(7)
    sampled = penv.sample_action()
    obs0, infos0 = penv.reset(seed=11)
    out = penv.step(actions if penv.is_driver else None)
    out2 = penv.step(actions * 2 if penv.is_driver else None)
    penv.close()
    if penv.is_driver:
        q.put({"obs0": {k: v.numpy().copy() for k, v in obs0.items()}, "r1": out[1].numpy().copy(), "r2": out2[1].numpy().copy(),
               "o2": {k: v.numpy().copy() for k, v in out2[0].items()}, "term2": out2[2], "n_infos": len(out2[4]),
               "info_m": np.array([float(i["m"]) for i in out2[4]]), "sampled": sampled.numpy().copy()})


def test_two_ranks_of_two_lanes_are_four_shards():
    """world 2 x lanes 2 (what ``bench.py --gpus N`` runs on every rank): lane l of rank r is virtual rank 2 r + l -- the packed block of
    a rank is its lanes' rows in lane order, the one all_gather puts the ranks in rank order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lanes_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    envs = [ToyEnv(num_envs=2) for _ in range(4)]
    actions = torch.arange(24, dtype=torch.float32).reshape(8, 3) * 0.05
    res0 = [e.reset(seed=11 + v) for v, e in enumerate(envs)]
    r1 = [e.step(actions[2 * v: 2 * v + 2])[1] for v, e in enumerate(envs)]
    o2r2 = [e.step(2 * actions[2 * v: 2 * v + 2]) for v, e in enumerate(envs)]
    for k in res0[0][0]:
        assert np.allclose(got["obs0"][k], torch.cat([x[0][k] for x in res0]).numpy())
        assert np.allclose(got["o2"][k], torch.cat([x[0][k] for x in o2r2]).numpy())
    assert np.allclose(got["r1"], torch.cat(r1).numpy()) and np.allclose(got["r2"], torch.cat([x[1] for x in o2r2]).numpy())
    assert got["term2"] == torch.cat([x[2] for x in o2r2]).tolist() and got["n_infos"] == 8
    assert np.allclose(got["info_m"], torch.cat([x[4]["m"] for x in o2r2]).numpy())
    samp = []
    for v in range(4):
        e = ToyEnv(num_envs=2)
        e.seed(7 + v)
        samp.append(e.sample_action())
    assert np.allclose(got["sampled"], torch.cat(samp).numpy())


def test_forced_collectives_at_world_size_one():
    """The CPU twin of tests/test_gpu_rccl.py: one rank under torch.distributed.run, every command through the group's
    broadcast / all_gather (gloo here, RCCL there), equal to the plain env."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "tests", "rccl_child.py"), "gloo", "ToyCPU-v0", "3"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    rep = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rep["backend"] == "gloo" and rep["world"] == 1 and rep["checks"] >= 8
    # without a process group the switch is refused instead of silently skipping the collectives
    _register()
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    os.environ.pop("RANK", None)
    with pytest.raises(RuntimeError, match="force_collectives needs"):
        ParallelFluidEnv("ToyCPU-v0", num_envs=2, backend="gloo", force_collectives=True)


class ToyShiftingInfoEnv(ToyEnv):
    """Reports another key set at every reset (a randomised reset of a real env may: ADVICE r3)."""

    def reset(self, seed=None, randomize=None):
        obs, info = super().reset(seed, randomize)
        self._n_resets = getattr(self, "_n_resets", 0) + 1
        if self._n_resets > 1:
            info = {"aaa_first": self.state[:, :2], "init_sum": info["init_sum"]}
        return obs, info


def test_reset_info_layout_follows_the_reset():
    import fluidgym_amd
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    if "ToyShift-v0" not in fluidgym_amd.registry.ids:
        fluidgym_amd.register("ToyShift-v0", ToyShiftingInfoEnv, {"gain": 1.0})
    os.environ.pop("RANK", None)
    penv = ParallelFluidEnv("ToyShift-v0", num_envs=3, backend="gloo")
    _, i1 = penv.reset(seed=1)
    assert set(i1[0]) == {"init_sum"}
    _, i2 = penv.reset(seed=2)
    assert set(i2[0]) == {"aaa_first", "init_sum"} and i2[1]["aaa_first"].shape == (2,)
    st = penv.local_env.state
    assert np.allclose([float(i["init_sum"]) for i in i2], st.sum(dim=1).numpy())     # not mis-sliced by the first reset's layout
    assert np.allclose(np.stack([i["aaa_first"].numpy() for i in i2]), st[:, :2].numpy())
    penv.close()


def _dying_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    _register()
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    penv = ParallelFluidEnv("ToyCPU-v0", num_envs=4, backend="gloo", collective_timeout_s=8)
    penv.reset(seed=1)
    if rank == 1:
        os._exit(0)                      # dies between two commands
    import time

    t0 = time.time()
    try:
        penv.step(torch.zeros(4, 3))
        q.put(("hung-or-ok", time.time() - t0))
    except Exception as e:                # noqa: BLE001 -- any backend error is the point
        q.put(("raised", time.time() - t0, type(e).__name__))


def test_dead_rank_raises_instead_of_hanging():
    """A dead rank must not hang the others (the reference's Pipe.recv does, parallel_env.py:233-287): the group is created
    with a timeout (`collective_timeout_s` / FLUIDGYM_COLLECTIVE_TIMEOUT_S)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dying_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=30)
    assert res[0] == "raised" and res[1] < 60, res


def test_num_envs_must_divide_world():
    _register()
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    os.environ.pop("RANK", None)
    penv = ParallelFluidEnv("ToyCPU-v0", num_envs=5, backend="gloo")
    assert penv.num_envs == 5 and penv.world == 1
    penv.close()


class ToyMarlEnv(ToyEnv):
    """Two agents per env: actions [B, 2, 3], observations [B, 2, ...], rewards [B, 2]."""

    n_agents = 2
    use_marl = True

    def __init__(self, num_envs=1, **kw):
        super().__init__(num_envs=num_envs, **kw)
        self._zero_action = torch.zeros(num_envs, 2, 3)
        self.state = torch.zeros(num_envs, 2, 3)

    def _obs(self):
        return {"a": self.state[..., :2].unsqueeze(-1).expand(-1, -1, 2, 2).contiguous(),
                "b": torch.cat([self.state, self.state[..., :2] * 3], dim=-1)}

    def reset(self, seed=None, randomize=None):
        self.state = torch.arange(self._num_envs * 6, dtype=torch.float32).reshape(self._num_envs, 2, 3) + (seed or 0)
        return self._obs(), {}

    def step(self, action):
        assert action.shape == (self._num_envs, 2, 3)
        self.state = self.state + self.gain * action
        return self._obs(), self.state.sum(dim=-1), False, False, {}

    def sample_action(self):
        return torch.ones(self._num_envs, 2, 3)


def test_multi_agent_rows_follow_the_reference_aggregation():
    """Multi-agent observations / sampled actions are concatenated over envs ([num_envs * n_agents, ...], reference
    parallel_env.py:192-200, 356-359), rewards stay [num_envs, n_agents]; both action layouts are accepted."""
    import fluidgym_amd
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    if "ToyMarlCPU-v0" not in fluidgym_amd.registry.ids:
        fluidgym_amd.register("ToyMarlCPU-v0", ToyMarlEnv, {"gain": 1.0})
    os.environ.pop("RANK", None)
    penv = ParallelFluidEnv("ToyMarlCPU-v0", num_envs=3, backend="gloo")
    assert penv.n_agents == 6
    obs, _ = penv.reset(seed=0)
    assert obs["b"].shape == (6, 5) and obs["a"].shape == (6, 2, 2)
    a = penv.sample_action()
    assert a.shape == (6, 3)
    o1, r1, *_ = penv.step(a)                      # rows
    o2, r2, *_ = penv.step(a.reshape(3, 2, 3))     # [num_envs, n_agents, ...]
    assert r1.shape == (3, 2) and o1["b"].shape == (6, 5)
    assert torch.allclose(o2["b"][:, :3] - o1["b"][:, :3], torch.ones(6, 3))
    with pytest.raises(ValueError, match="Expected action batch size"):
        penv.step(torch.zeros(4, 3))
    penv.close()


# ---- a REAL registry env (ChannelJet2D: action smoothing, jets -> boundary data, sensors, reward, Simulation driver) with the
# solver stubbed at its boundary (tests/stub_solver.py), sharded over two gloo ranks and compared with one 4-env process
def _stub_env_patches():
    import fluidgym_amd.simulation.domain as D
    from tests.stub_solver import StubSolver

    D.NativeSolver = StubSolver
    torch.cuda.is_available = lambda: True   # FluidEnv.reset's "FluidGym requires CUDA" guard; nothing here touches a GPU


_STUB_KW = dict(cuda_device="cpu", randomize_initial_state=False, resolution_x=32, resolution_y=16, step_length=0.05)


def _stub_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _stub_env_patches()
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    penv = ParallelFluidEnv("ChannelJet2D-v0", num_envs=4, backend="gloo", **_STUB_KW)
    if not penv.is_driver:
        penv.serve()
        q.put((rank, penv.local_env._sim is None))
        return
    obs0, _ = penv.reset(seed=3)
    g = torch.Generator().manual_seed(0)
    outs = []
    for _ in range(3):
        a = torch.rand(4, 1, generator=g) * 2 - 1
        o, r, term, trunc, info = penv.step(a)
        outs.append(({k: v.numpy().copy() for k, v in o.items()}, r.numpy().copy(), term, trunc,
                     np.array([float(i["wall_shear"]) for i in info])))
    penv.close()
    q.put((rank, {k: v.numpy().copy() for k, v in obs0.items()}, outs))


def test_real_env_with_stubbed_solver_sharded_over_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stub_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    drv = [r for r in results if r[0] == 0][0]
    # single process, the same four envs in one batch
    real_avail = torch.cuda.is_available
    import fluidgym_amd.simulation.domain as D
    real_solver = D.NativeSolver
    try:
        _stub_env_patches()
        import fluidgym_amd

        env = fluidgym_amd.make("ChannelJet2D-v0", num_envs=4, **_STUB_KW)
        obs0, _ = env.reset(seed=3)
        for k in obs0:
            assert np.allclose(drv[1][k], obs0[k].numpy())
        g = torch.Generator().manual_seed(0)
        for step in range(3):
            a = torch.rand(4, 1, generator=g) * 2 - 1
            o, r, term, trunc, info = env.step(a)
            po, pr, pterm, ptrunc, pshear = drv[2][step]
            for k in o:
                assert np.allclose(po[k], o[k].numpy(), atol=1e-6), k
            assert np.allclose(pr, r.numpy(), atol=1e-6)
            assert pterm == [bool(term)] * 4 and ptrunc == [bool(trunc)] * 4
            assert np.allclose(pshear, info["wall_shear"].numpy(), atol=1e-6)
        assert np.abs(drv[2][2][1]).max() > 0   # the actions reached the boundary data and the reward saw them
        env.close()
    finally:
        torch.cuda.is_available = real_avail
        D.NativeSolver = real_solver


# ---- a REAL multi-block env (CylinderJet2D: five-block mesh, jets -> boundary slots, action smoothing, sensor gathers, wall-stress
# forces, reward, the MultiBlockSimulation driver) with the device half of its domain stubbed (tests/stub_mb.py; the mesh tables
# come from the native library's host-only handle), sharded over two gloo ranks and compared with one 4-env process
_MB_KW = dict(cuda_device="cpu", randomize_initial_state=False, initial_domain_steps=2, episode_length=3)


def _mb_patches():
    from tests import stub_mb

    torch.cuda.is_available = lambda: True   # FluidEnv.reset's "FluidGym requires CUDA" guard; nothing here touches a GPU
    return stub_mb.install()


def _mb_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _mb_patches()
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    penv = ParallelFluidEnv("CylinderJet2D-easy-v0", num_envs=4, backend="gloo", **_MB_KW)
    if not penv.is_driver:
        penv.serve()
        from tests import stub_mb
        q.put((rank, stub_mb.CALLS[0]))
        return
    obs0, infos0 = penv.reset(seed=3)
    g = torch.Generator().manual_seed(0)
    outs = []
    for _ in range(2):
        a = torch.rand(4, 1, generator=g) * 2 - 1
        o, r, term, trunc, info = penv.step(a)
        outs.append(({k: v.numpy().copy() for k, v in o.items()}, r.numpy().copy(), term, trunc,
                     np.array([float(i["drag"]) for i in info]), np.array([float(i["lift"]) for i in info])))
    penv.close()
    q.put((rank, {k: v.numpy().copy() for k, v in obs0.items()}, outs, len(infos0)))


def test_multi_block_env_with_stubbed_solve_sharded_over_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mb_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    drv = [r for r in results if r[0] == 0][0]
    worker = [r for r in results if r[0] == 1][0]
    assert worker[1] > 0 and drv[3] == 4            # the worker's shard stepped; one reset info per env
    real_avail = torch.cuda.is_available
    restore = _mb_patches()
    try:
        import fluidgym_amd

        env = fluidgym_amd.make("CylinderJet2D-easy-v0", num_envs=4, **_MB_KW)
        obs0, _ = env.reset(seed=3)
        for k in obs0:
            assert drv[1][k].shape == tuple(obs0[k].shape) and np.allclose(drv[1][k], obs0[k].numpy(), atol=1e-6)
        g = torch.Generator().manual_seed(0)
        for step in range(2):
            a = torch.rand(4, 1, generator=g) * 2 - 1
            o, r, term, trunc, info = env.step(a)
            po, pr, pterm, ptrunc, pdrag, plift = drv[2][step]
            for k in o:
                assert np.allclose(po[k], o[k].numpy(), atol=1e-6), k
            assert np.allclose(pr, r.numpy(), atol=1e-6)
            assert pterm == [bool(term)] * 4 and ptrunc == [bool(trunc)] * 4
            assert np.allclose(pdrag, info["drag"].numpy(), atol=1e-6) and np.allclose(plift, info["lift"].numpy(), atol=1e-6)
        # the jets of env 0 and env 3 differ: so do their forces (the actions reached the boundary slots of the right shard)
        assert abs(drv[2][1][5][0] - drv[2][1][5][3]) > 1e-6
        env.close()
    finally:
        torch.cuda.is_available = real_avail
        restore()
