"""``ParallelFluidEnv``: env batches sharded over the GPUs of one node, RCCL for actions/observations.

Reference: ``envs/parallel_env.py:45-444`` -- one OS process and ONE env per GPU, parent<->child
``multiprocessing.Pipe`` RPC with pickled CUDA tensors, results moved ``.cpu()`` and stacked in the
parent (``:115-200, 233-287``); no collective library anywhere.

MI355X design: one process per GPU in a ``torch.distributed`` group (backend ``nccl`` = RCCL over
xGMI); every rank owns ``num_envs / world`` envs *batched inside one solver handle*.  Per step the
driver rank broadcasts the action block (a few KB) and one ``all_gather`` returns the packed
observation + reward block of every shard; no field data ever leaves a GPU.  Messages are
latency-bound (KBs against 7 x ~153 GB/s links), so each direction is exactly one collective.

Two ways to run it:

* SPMD (``torchrun`` / ``python -m torch.distributed.run``): every rank constructs the object and
  calls ``reset`` / ``step`` collectively; the driver (rank 0) passes the full action tensor and gets
  the aggregated results, the other ranks pass ``None`` -- or call :meth:`serve` and simply follow the
  driver's commands (the reference's Command enum, ``:17-27``);
* reference-style ``ParallelFluidEnv(env_id, cuda_ids=[0, 1, ...])`` from a plain Python process:
  the calling process becomes rank 0 on ``cuda_ids[0]`` and spawns one worker per remaining GPU, each
  running :meth:`serve`.
"""
from __future__ import annotations

import os
from enum import IntEnum
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from ..registry import make


class Command(IntEnum):
    STEP = 0
    RESET = 1
    SEED = 2
    TRAIN = 3
    VAL = 4
    TEST = 5
    LOAD_INITIAL_DOMAIN = 6
    CLOSE = 7
    SAMPLE_ACTION = 8


def _spawn_worker(rank: int, world: int, port: int, env_id: str, cuda_ids: List[int], num_envs: int,
                  env_kwargs: Dict[str, Any], backend: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    env = ParallelFluidEnv(env_id, cuda_ids=cuda_ids, num_envs=num_envs, backend=backend, _spawned=True, **env_kwargs)
    env.serve()


class ParallelFluidEnv:
    def __init__(self, env_id: str, cuda_ids: Optional[Sequence[int]] = None, num_envs: Optional[int] = None,
                 backend: Optional[str] = None, _spawned: bool = False, **env_kwargs: Any):
        if env_kwargs.get("differentiable", False):
            raise ValueError("ParallelFluidEnv does not support differentiable environments.")
        self._env_id = env_id
        self._workers = []
        self._owns_group = False
        launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        self._backend = backend
        if not launched and cuda_ids is not None and len(cuda_ids) > 1:
            # reference-style entry: become rank 0 and spawn the other ranks
            import torch.multiprocessing as mp

            world = len(cuda_ids)
            port = int(os.environ.get("FLUIDGYM_MASTER_PORT", 29000 + (os.getpid() % 2000)))
            n_total = num_envs if num_envs is not None else world
            ctx = mp.get_context("spawn")
            for r in range(1, world):
                p = ctx.Process(target=_spawn_worker,
                                args=(r, world, port, env_id, list(cuda_ids), n_total, env_kwargs, backend), daemon=True)
                p.start()
                self._workers.append(p)
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE=str(world),
                              LOCAL_RANK="0")
            launched = True
        if launched:
            if not dist.is_initialized():
                dist.init_process_group(backend=backend, init_method="env://")
                self._owns_group = True
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        else:
            self.rank, self.world = 0, 1
        local = int(os.environ.get("LOCAL_RANK", self.rank))
        if backend == "nccl":
            dev_index = cuda_ids[local] if cuda_ids is not None else local
            self._device = torch.device("cuda", int(dev_index))
            torch.cuda.set_device(self._device)
        else:
            self._device = torch.device("cpu")
        n_total = num_envs if num_envs is not None else self.world
        if n_total % self.world:
            raise ValueError(f"num_envs={n_total} must be divisible by the number of GPUs ({self.world})")
        self._n_total = int(n_total)
        self._n_local = self._n_total // self.world
        kw = dict(env_kwargs)
        if backend == "nccl":
            kw["cuda_device"] = self._device
        self._env = make(env_id, num_envs=self._n_local, **kw)
        self._obs_keys = sorted(self._env.observation_space.keys())
        self._cmd = torch.zeros(4, dtype=torch.int64, device=self._device)

    # ------------------------------------------------------------------ introspection
    def __getattr__(self, name: str) -> Any:
        return getattr(self._env, name)

    @property
    def is_driver(self) -> bool:
        return self.rank == 0

    @property
    def action_space(self):
        return self._env.action_space

    @property
    def observation_space(self):
        return self._env.observation_space

    @property
    def differentiable(self) -> bool:
        return False

    @property
    def num_envs(self) -> int:
        return self._n_total

    @property
    def n_agents(self) -> int:
        return self._n_total * self._env.n_agents

    @property
    def local_env(self):
        return self._env

    # ------------------------------------------------------------------ collectives
    def _bcast_cmd(self, cmd: Optional[Command] = None, a: int = 0, b: int = 0, c: int = 0) -> List[int]:
        if self.world == 1:
            return [int(cmd), a, b, c]
        if self.is_driver:
            self._cmd.copy_(torch.tensor([int(cmd), int(a), int(b), int(c)], dtype=torch.int64))
        dist.broadcast(self._cmd, src=0)
        return [int(v) for v in self._cmd.tolist()]

    def _all_gather(self, local: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return local
        out = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        try:
            dist.all_gather_into_tensor(out, local.contiguous())
        except (RuntimeError, NotImplementedError):
            parts = [torch.empty_like(local) for _ in range(self.world)]
            dist.all_gather(parts, local.contiguous())
            out = torch.stack(parts)
        return out.reshape((-1,) + tuple(local.shape[1:]))

    def _pack(self, obs: Dict[str, torch.Tensor], reward: Optional[torch.Tensor]) -> torch.Tensor:
        parts = [obs[k].reshape(self._n_local, -1).float() for k in self._obs_keys]
        if reward is not None:
            parts.append(reward.reshape(self._n_local, -1).float())
        return torch.cat(parts, dim=1)

    def _unpack(self, flat: torch.Tensor, obs_like: Dict[str, torch.Tensor], with_reward: bool):
        out, off = {}, 0
        n = flat.shape[0]
        for k in self._obs_keys:
            shp = obs_like[k].shape[1:]
            size = int(np.prod(shp)) if len(shp) else 1
            out[k] = flat[:, off: off + size].reshape((n,) + tuple(shp))
            off += size
        reward = flat[:, off:].reshape(n, *(() if flat.shape[1] - off == 1 else (-1,))) if with_reward else None
        return out, reward

    # ------------------------------------------------------------------ env API (collective)
    # Every public method is a COLLECTIVE call: all ranks that are not inside serve() must call it; the
    # driver's arguments win (they travel in the command broadcast).
    def seed(self, seed: int = 0) -> None:
        _, a, _, _ = self._bcast_cmd(Command.SEED, int(seed))
        self._env.seed(a + self.rank)

    def reset(self, seed: Optional[int] = None, randomize: Optional[bool] = None):
        """All shards reset with seeds ``seed + rank`` (independent envs); returns the observations of
        all ``num_envs`` envs stacked along dim 0 and a list of per-shard info dicts."""
        _, a, b, _ = self._bcast_cmd(Command.RESET, -1 if seed is None else int(seed),
                                     -1 if randomize is None else int(randomize))
        return self._do_reset(None if a < 0 else a, None if b < 0 else bool(b))

    def _do_reset(self, seed, randomize):
        obs, info = self._env.reset(seed=None if seed is None else int(seed) + self.rank, randomize=randomize)
        flat = self._all_gather(self._pack(obs, None))
        obs_all, _ = self._unpack(flat, obs, with_reward=False)
        return self._agents_to_rows(obs_all), [info for _ in range(self.world)]

    def step(self, action: Optional[torch.Tensor] = None):
        """Driver: ``action [num_envs, ...]``.  Other ranks in SPMD mode pass ``None``."""
        if self.is_driver:
            # multi-agent: one row per agent of every env, envs concatenated (what sample_action returns, reference
            # parallel_env.py:356-359), or [num_envs, n_agents, ...]
            ok = (self._n_total, self._n_total * self._env.n_agents) if self._env.use_marl else (self._n_total,)
            if action is None or action.shape[0] not in ok:
                raise ValueError(f"Expected action batch size {ok[-1]}, but got "
                                 f"{None if action is None else action.shape[0]}")
        self._bcast_cmd(Command.STEP)
        return self._do_step(action)

    def _do_step(self, action: Optional[torch.Tensor]):
        a_shape = (self._n_total,) + tuple(self._env._zero_action.shape[1:])
        if self.is_driver:
            full = action.to(self._device, torch.float32).reshape(a_shape).contiguous()
        else:
            full = torch.empty(a_shape, dtype=torch.float32, device=self._device)
        if self.world > 1:
            dist.broadcast(full, src=0)  # actions: one small collective
        mine = full[self.rank * self._n_local: (self.rank + 1) * self._n_local]
        obs, reward, term, trunc, info = self._env.step(mine)
        flat = self._all_gather(self._pack(obs, reward))  # observations + rewards: one collective
        obs_all, reward_all = self._unpack(flat, obs, with_reward=True)
        obs_all = self._agents_to_rows(obs_all)
        infos = [{k: v for k, v in info.items()} for _ in range(1)]
        return obs_all, reward_all, [term] * self._n_total, [trunc] * self._n_total, infos

    def _agents_to_rows(self, obs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Multi-agent observations are concatenated over envs, ``[num_envs * n_agents, ...]`` (reference
        ``__aggregate_obs``, parallel_env.py:192-200); single-agent ones stay stacked ``[num_envs, ...]``."""
        if not self._env.use_marl:
            return obs
        return {k: v.reshape((-1,) + tuple(v.shape[2:])) for k, v in obs.items()}

    def sample_action(self) -> torch.Tensor:
        a = self._all_gather(self._env.sample_action())
        return a.reshape((-1,) + tuple(a.shape[2:])) if self._env.use_marl else a

    def train(self) -> None:
        self._bcast_cmd(Command.TRAIN)
        self._env.train()

    def val(self) -> None:
        self._bcast_cmd(Command.VAL)
        self._env.val()

    def test(self) -> None:
        self._bcast_cmd(Command.TEST)
        self._env.test()

    def load_initial_domain(self, idx: int = 0, mode=None) -> None:
        _, a, _, _ = self._bcast_cmd(Command.LOAD_INITIAL_DOMAIN, int(idx))
        self._env.load_initial_domain(a * self.world + self.rank, mode)

    # ------------------------------------------------------------------ worker loop
    def serve(self) -> None:
        """Follow the driver's commands until CLOSE (the reference's ``_worker``, parallel_env.py:115-160)."""
        assert not self.is_driver
        while True:
            cmd, a, b, _ = self._bcast_cmd()
            if cmd == Command.STEP:
                self._do_step(None)
            elif cmd == Command.RESET:
                self._do_reset(None if a < 0 else a, None if b < 0 else bool(b))
            elif cmd == Command.SEED:
                self._env.seed(a + self.rank)
            elif cmd == Command.TRAIN:
                self._env.train()
            elif cmd == Command.VAL:
                self._env.val()
            elif cmd == Command.TEST:
                self._env.test()
            elif cmd == Command.LOAD_INITIAL_DOMAIN:
                self._env.load_initial_domain(a * self.world + self.rank, None)
            elif cmd == Command.CLOSE:
                break
        self._shutdown()

    def close(self) -> None:
        if self.world > 1:
            self._bcast_cmd(Command.CLOSE)
        self._shutdown()
        for p in self._workers:
            p.join(timeout=30)

    def _shutdown(self) -> None:
        self._env.close()
        if self._owns_group and dist.is_initialized():
            dist.destroy_process_group()
            self._owns_group = False
