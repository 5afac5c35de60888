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
        return self._obs(), {}

    def step(self, action):
        assert action.shape == (self._num_envs, 3)
        self.state = self.state + self.gain * action
        return self._obs(), self.state.sum(dim=1), False, False, {"m": self.state.mean(dim=1)}

    def sample_action(self):
        return torch.zeros(self._num_envs, 3)

    def train(self):
        self.mode = "train"

    def val(self):
        self.mode = "val"

    def test(self):
        self.mode = "test"

    def load_initial_domain(self, idx, mode=None):
        self.reset(seed=1000 + idx)

    def close(self):
        pass


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
        q.put((rank, "served", penv.local_env.mode))
        return
    penv.seed(5)
    obs0, _ = penv.reset(seed=11)
    penv.val()
    out = penv.step(actions if penv.is_driver else None)
    out2 = penv.step(actions * 2 if penv.is_driver else None)
    mode_seen = penv.local_env.mode
    penv.close()
    if penv.is_driver:
        q.put((rank, {k: v.numpy().copy() for k, v in obs0.items()}, out[1].numpy().copy(), out2[1].numpy().copy(),
               {k: v.numpy().copy() for k, v in out2[0].items()}, mode_seen))
    else:
        q.put((rank, "spmd", mode_seen))


def _expected():
    envs = [ToyEnv(num_envs=2), ToyEnv(num_envs=2)]
    actions = torch.arange(12, dtype=torch.float32).reshape(4, 3) * 0.1
    obs0 = [e.reset(seed=11 + r)[0] for r, e in enumerate(envs)]
    r1 = [e.step(actions[2 * r: 2 * r + 2])[1] for r, e in enumerate(envs)]
    o2r2 = [e.step(2 * actions[2 * r: 2 * r + 2]) for r, e in enumerate(envs)]
    cat = lambda ds: {k: torch.cat([d[k] for d in ds]) for k in ds[0]}
    return cat(obs0), torch.cat(r1), torch.cat([x[1] for x in o2r2]), cat([x[0] for x in o2r2])


@pytest.mark.parametrize("mode", ["spmd", "serve"])
def test_two_rank_gloo_matches_single_process(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    drv = [r for r in results if r[0] == 0][0]
    other = [r for r in results if r[0] == 1][0]
    obs0, r1, r2, o2 = _expected()
    for k in obs0:
        assert np.allclose(drv[1][k], obs0[k].numpy())
        assert np.allclose(drv[4][k], o2[k].numpy())
    assert np.allclose(drv[2], r1.numpy()) and np.allclose(drv[3], r2.numpy())
    assert drv[5] == "val" and other[2] == "val"  # mode command reached every shard


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
