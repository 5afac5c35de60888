"""``ParallelFluidEnv``: env batches sharded over the GPUs of one node, RCCL for actions/observations.

Reference: ``envs/parallel_env.py:45-444`` -- one OS process and ONE env per GPU, parent<->child
``multiprocessing.Pipe`` RPC with pickled CUDA tensors, results moved ``.cpu()`` and stacked in the
parent (``:115-200, 233-287``); no collective library anywhere.

MI355X design: one process per GPU in a ``torch.distributed`` group (backend ``nccl`` = RCCL over
xGMI); every rank owns ``num_envs / world`` envs *batched inside one solver handle*.  Per step the
driver rank broadcasts ONE message (command header + action block, a few KB) and ONE ``all_gather`` returns
the packed observation | reward | terminated | truncated | info block of every shard; no field data ever
leaves a GPU.  Messages are latency-bound (KBs against 7 x ~153 GB/s links), so each direction is exactly one
collective; the collectives run on RCCL's own stream, ranks that already know the command (SPMD callers) never
read the header back to the host, and the only host read of a step is the per-env terminated / truncated
flags the reference's return type (Python lists) asks for.

Lanes (``lanes=L``, round 6): a rank's shard may itself be split into L sub-shards ("lanes"), each its own env batch and solver
handle, stepped CONCURRENTLY by L host threads on L HIP streams of the one GPU.  The single-block step is bound by host round trips
(polls of convergence words, launch latency: the GPU idles ~22 % of a 64-env headline step), and a second lane's kernels fill exactly
those gaps: 2 x 32 envs give 1.2x the env-steps/s of 1 x 64 on the same GPU (profiles/headline_two_streams.sh).  It is the in-process
form of what the reference allows as ``cuda_ids=[0, 0]`` (two workers on one GPU, parallel_env.py:115-160).  Lane l of rank r acts as
virtual rank ``r * L + l`` everywhere a rank number enters (seeds, initial-domain indices), so ``world=1, lanes=2`` runs the envs of
``world=2, lanes=1``.  NOT for the multi-block cluster solvers (csrc/fg_mb_cluster.hip): their workgroups spin on each other and need
the whole GPU co-resident, two such kernels at once can starve each other into the time-out path; those envs refuse ``lanes > 1``.

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
from ..types import EnvMode


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


def _free_port() -> int:
    """A port the OS just handed out (29000 + pid % 2000 could collide between two drivers on one host)."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return int(sock.getsockname()[1])


def _spawn_worker(rank: int, world: int, port: int, env_id: str, cuda_ids: List[int], num_envs: int,
                  env_kwargs: Dict[str, Any], backend: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    env = ParallelFluidEnv(env_id, cuda_ids=cuda_ids, num_envs=num_envs, backend=backend, _spawned=True, **env_kwargs)
    env.serve()


class _Lanes:
    """L sub-shards of one rank behind the FluidEnv calls ParallelFluidEnv makes (module docstring, "Lanes").  Every call fans out to
    the lanes -- lane 0 on the calling thread, the others on a thread pool, each under its own HIP stream -- and the per-lane results
    are concatenated along the env axis on the caller's stream.  Stream order: a lane's stream first waits for an event recorded on the
    caller's stream (the actions are produced there, and memory a lane frees is only reused after what the caller enqueued), the
    caller's stream then waits for every lane's end event before it concatenates."""

    def __init__(self, env_id: str, n_local: int, lanes: int, device: torch.device, kw: Dict[str, Any]):
        from concurrent.futures import ThreadPoolExecutor

        if n_local % lanes:
            raise ValueError(f"{n_local} envs per GPU are not divisible by lanes={lanes}")
        self.n_lane, self.n_lanes, self._device = n_local // lanes, lanes, device
        self.envs = [make(env_id, num_envs=self.n_lane, **kw) for _ in range(lanes)]
        self._cuda = device.type == "cuda"
        self._streams = [torch.cuda.Stream(device) for _ in range(lanes)] if self._cuda else [None] * lanes
        self._pool = ThreadPoolExecutor(max_workers=max(lanes - 1, 1), thread_name_prefix="fluidgym-lane")
        z = self.envs[0]._zero_action
        self._zero_action = z.new_zeros((n_local,) + tuple(z.shape[1:]))
        self._check_solvers()

    def __getattr__(self, name: str) -> Any:      # spaces, episode length, counters ...: every lane is the same env class and options
        if name in ("envs", "_pool", "_streams"):      # (not built yet: a failed constructor must not recurse through this hook)
            raise AttributeError(name)
        return getattr(self.envs[0], name)

    def _check_solvers(self) -> None:
        from .cylinder import CylinderEnvBase      # (cylinder and airfoil ids: the multi-block path, fg_mb_*)

        if self._cuda and isinstance(self.envs[0], CylinderEnvBase):
            raise ValueError("lanes > 1 is for the single-block solver path: the multi-block cluster kernels need the whole GPU "
                             "co-resident (fluidgym_amd/envs/parallel_env.py, 'Lanes')")

    def _fan(self, fn):
        """``fn(lane, env)`` on every lane concurrently; the list of results in lane order."""
        if not self._cuda:
            futs = [self._pool.submit(fn, l, self.envs[l]) for l in range(1, self.n_lanes)]
            try:
                first = fn(0, self.envs[0])
            finally:
                from concurrent.futures import wait

                wait(futs)
            return [first] + [f.result() for f in futs]
        main = torch.cuda.current_stream(self._device)
        start = torch.cuda.Event()
        start.record(main)

        def task(l):
            torch.cuda.set_device(self._device)
            st = self._streams[l]
            st.wait_event(start)
            with torch.cuda.stream(st):
                out = fn(l, self.envs[l])
            end = torch.cuda.Event()
            end.record(st)
            return out, end

        futs = [self._pool.submit(task, l) for l in range(1, self.n_lanes)]
        try:
            first = task(0)
        finally:      # (a lane that raised must not leave the others stepping behind the caller's back)
            from concurrent.futures import wait

            wait(futs)
        done = [first] + [f.result() for f in futs]
        for _, end in done:
            main.wait_event(end)
        return [out for out, _ in done]

    def _rows(self, v) -> torch.Tensor:
        """A lane's per-env tensor, or its per-shard value repeated for each of its envs."""
        t = v if isinstance(v, torch.Tensor) else torch.as_tensor(v)
        if t.dim() > 0 and t.shape[0] == self.n_lane:
            return t
        return t.reshape((1,) + tuple(t.shape)).expand((self.n_lane,) + tuple(t.shape))

    def _cat_info(self, infos: List[Dict[str, Any]]) -> Dict[str, Any]:
        out = {}
        for k in infos[0]:
            try:
                parts = [self._rows(i[k]) for i in infos]
                out[k] = torch.cat([q.to(parts[0].device) for q in parts], dim=0)
            except (TypeError, ValueError, RuntimeError):
                out[k] = infos[0][k]      # not numeric: lane 0's (it stays on its shard, like in _layout_of)
        return out

    @staticmethod
    def _cat_obs(obs: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
        return {k: torch.cat([o[k] for o in obs], dim=0) for k in obs[0]}

    def _flag(self, vals):
        if all(not isinstance(v, torch.Tensor) for v in vals) and len({bool(v) for v in vals}) == 1:
            return bool(vals[0])
        return torch.cat([self._rows(v).reshape(self.n_lane, -1)[:, 0].to(self._device) != 0 for v in vals])

    # ---- the FluidEnv calls of ParallelFluidEnv; ``base`` is the virtual rank of lane 0 (rank * lanes)
    def reset(self, seed=None, randomize=None, base: int = 0):
        res = self._fan(lambda l, e: e.reset(seed=None if seed is None else int(seed) + base + l, randomize=randomize))
        return self._cat_obs([r[0] for r in res]), self._cat_info([r[1] for r in res])

    def step(self, action: torch.Tensor):
        n = self.n_lane
        res = self._fan(lambda l, e: e.step(action[l * n: (l + 1) * n]))
        obs = self._cat_obs([r[0] for r in res])
        reward = torch.cat([r[1].reshape((n,) + tuple(r[1].shape[1:])) for r in res], dim=0)
        return obs, reward, self._flag([r[2] for r in res]), self._flag([r[3] for r in res]), self._cat_info([r[4] for r in res])

    def seed(self, seed: int, base: int = 0) -> None:
        for l, e in enumerate(self.envs):
            e.seed(int(seed) + base + l)

    def sample_action(self) -> torch.Tensor:
        return torch.cat([e.sample_action().to(self._device) for e in self.envs], dim=0)

    def load_initial_domain(self, idx: int, mode=None, base: int = 0, stride: int = 1) -> None:
        for l, e in enumerate(self.envs):
            e.load_initial_domain(idx * stride + base + l, mode)

    def train(self) -> None:
        for e in self.envs:
            e.train()

    def val(self) -> None:
        for e in self.envs:
            e.val()

    def test(self) -> None:
        for e in self.envs:
            e.test()

    def close(self) -> None:
        for e in self.__dict__.get("envs", []):
            e.close()
        pool = self.__dict__.get("_pool")
        if pool is not None:
            pool.shutdown(wait=True)


class ParallelFluidEnv:
    def __init__(self, env_id: str, cuda_ids: Optional[Sequence[int]] = None, num_envs: Optional[int] = None,
                 backend: Optional[str] = None, _spawned: bool = False, collective_timeout_s: Optional[float] = None,
                 force_collectives: Optional[bool] = None, lanes: Optional[int] = None, **env_kwargs: Any):
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
            port = int(os.environ.get("FLUIDGYM_MASTER_PORT", 0)) or _free_port()
            n_total = num_envs if num_envs is not None else world
            ctx = mp.get_context("spawn")
            for r in range(1, world):
                p = ctx.Process(target=_spawn_worker,
                                args=(r, world, port, env_id, list(cuda_ids), n_total, dict(env_kwargs, lanes=lanes), backend), daemon=True)
                p.start()
                self._workers.append(p)
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE=str(world),
                              LOCAL_RANK="0")
            launched = True
        # A rank that dies leaves the others inside a collective: with a timeout the survivors raise instead of hanging (the
        # reference's Pipe.recv hangs, parallel_env.py:233-287).  FLUIDGYM_COLLECTIVE_TIMEOUT_S / collective_timeout_s, default 600 s
        # (an env step of the largest 3-D ids takes seconds; first-call RCCL setup tens of seconds).
        if collective_timeout_s is None:
            collective_timeout_s = float(os.environ.get("FLUIDGYM_COLLECTIVE_TIMEOUT_S", 600))
        self._timeout_s = float(collective_timeout_s)
        # force_collectives: issue the broadcast / all_gather of every command at world size 1 too (one-GPU boxes exercise the
        # RCCL branch with it: tests/test_gpu_rccl.py; FLUIDGYM_FORCE_COLLECTIVES=1).  Needs a process group, i.e. a launched rank.
        if force_collectives is None:
            force_collectives = os.environ.get("FLUIDGYM_FORCE_COLLECTIVES", "0") not in ("", "0")
        self._force = bool(force_collectives)
        if launched:
            if not dist.is_initialized():
                from datetime import timedelta

                dist.init_process_group(backend=backend, init_method="env://", timeout=timedelta(seconds=self._timeout_s))
                self._owns_group = True
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        else:
            self.rank, self.world = 0, 1
        local = int(os.environ.get("LOCAL_RANK", self.rank))
        if self._force and not (launched and dist.is_initialized()):
            raise RuntimeError("force_collectives needs a torch.distributed process group (launch the rank with torch.distributed.run)")
        self._collective = self._force or self.world > 1
        if backend == "nccl":
            dev_index = cuda_ids[local] if cuda_ids is not None else local
            self._device = torch.device("cuda", int(dev_index))
            torch.cuda.set_device(self._device)
        elif cuda_ids is not None and torch.cuda.is_available():
            # gloo with GPU envs (host-staged collectives): the shard steps on its GPU, the command / result blocks cross the
            # process group through host memory -- several ranks may share one GPU this way (tests/test_gpu_two_ranks.py)
            self._device = torch.device("cuda", int(cuda_ids[local]))
            torch.cuda.set_device(self._device)
        else:
            self._device = torch.device("cpu")
        self._comm_device = self._device if backend == "nccl" else torch.device("cpu")
        n_total = num_envs if num_envs is not None else self.world
        if n_total % self.world:
            raise ValueError(f"num_envs={n_total} must be divisible by the number of GPUs ({self.world})")
        self._n_total = int(n_total)
        self._n_local = self._n_total // self.world
        kw = dict(env_kwargs)
        if self._device.type == "cuda":
            kw["cuda_device"] = self._device
        # lanes: sub-shards of this rank stepped concurrently on their own HIP streams (module docstring); FLUIDGYM_AMD_LANES
        if lanes is None:
            lanes = int(os.environ.get("FLUIDGYM_AMD_LANES", "1") or 1)
        self._lanes = max(1, int(lanes))
        self._env = make(env_id, num_envs=self._n_local, **kw) if self._lanes == 1 else \
            _Lanes(env_id, self._n_local, self._lanes, self._device, kw)
        self._obs_keys = sorted(self._env.observation_space.keys())
        # ONE message per command: header [cmd, a, b, c] (four int64 fields seen as eight int32 words, so that seeds of 2**31 and
        # more travel: np.random.SeedSequence / getrandbits(32) produce them) followed by the action block bit-cast to int32, so that a
        # step costs one broadcast (header + actions) and one all_gather (obs | reward | terminated | truncated | info)
        self._a_shape = (self._n_total,) + tuple(self._env._zero_action.shape[1:])
        self._a_numel = int(np.prod(self._a_shape))
        self._msg = torch.zeros(self._HDR + self._a_numel, dtype=torch.int32, device=self._comm_device)
        self._info_layout: Optional[List] = None
        self._reset_info_layout: Optional[List] = None
        self.time_shards = False      # accumulate the wall time of this rank's own env.step calls in shard_seconds (a sync per step)
        self.shard_seconds = 0.0

    _HDR = 8   # int32 words: four int64 header fields

    # ------------------------------------------------------------------ introspection
    def __getattr__(self, name: str) -> Any:
        if name == "_env":      # (not built yet: a constructor that raised must not recurse through this hook)
            raise AttributeError(name)
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

    # (parallel_env.py:70-113: answered by the reference from a dummy env; here from this rank's shard)
    @property
    def cuda_device(self) -> torch.device:
        return self._env.cuda_device

    @property
    def episode_length(self) -> int:
        return self._env.episode_length

    @property
    def metrics(self):
        return self._env.metrics

    @property
    def use_marl(self) -> bool:
        return self._env.use_marl

    def render(self, save: bool = False, render_3d: bool = False, filename: Optional[str] = None, output_path: Any = None):
        raise NotImplementedError("Rendering is not implemented for ParallelFluidEnv.")      # parallel_env.py:289-319

    @property
    def num_envs(self) -> int:
        return self._n_total

    @property
    def n_agents(self) -> int:
        return self._n_total * self._env.n_agents

    @property
    def local_env(self):
        return self._env

    @property
    def lane_envs(self) -> List[Any]:
        """The env batches this rank steps: one, or ``lanes`` of them (module docstring)."""
        return [self._env] if self._lanes == 1 else list(self._env.envs)

    # ------------------------------------------------------------------ collectives
    def _send(self, cmd: Optional[Command] = None, a: int = 0, b: int = 0, c: int = 0,
              action: Optional[torch.Tensor] = None, read_header: bool = True) -> List[int]:
        """The one broadcast of a command.  Driver: fills header (+ actions).  ``read_header=False`` (SPMD ranks that
        already know the command because they made the same call) skips the device->host read of the header."""
        if self.is_driver:
            hdr = torch.tensor([int(cmd), int(a), int(b), int(c)], dtype=torch.int64).view(torch.int32)
            self._msg[: self._HDR].copy_(hdr, non_blocking=True)
            if action is not None:
                self._msg[self._HDR:].copy_(action.to(self._comm_device, torch.float32).reshape(-1).view(torch.int32))
            if self._collective:
                dist.broadcast(self._msg, src=0)
            return [int(cmd), int(a), int(b), int(c)]
        dist.broadcast(self._msg, src=0)
        if not read_header:
            return [int(cmd) if cmd is not None else -1, a, b, c]
        return [int(v) for v in self._msg[: self._HDR].cpu().view(torch.int64).tolist()]

    def _actions_from_msg(self) -> torch.Tensor:
        return self._msg[self._HDR:].view(torch.float32).reshape(self._a_shape).to(self._device)

    def _all_gather(self, local: torch.Tensor) -> torch.Tensor:
        if not self._collective:
            return local
        back = local.device
        local = local.to(self._comm_device)      # (host-staged when the group is gloo and the envs live on a GPU)
        out = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        try:
            dist.all_gather_into_tensor(out, local.contiguous())
        except (RuntimeError, NotImplementedError):
            parts = [torch.empty_like(local) for _ in range(self.world)]
            dist.all_gather(parts, local.contiguous())
            out = torch.stack(parts)
        return out.reshape((-1,) + tuple(local.shape[1:])).to(back)

    def _per_env(self, v) -> torch.Tensor:
        """A per-shard scalar / bool or a per-env tensor as one float row per local env."""
        t = torch.as_tensor(v, device=self._device).float()
        if t.dim() == 0 or t.shape[0] != self._n_local:
            t = t.reshape(1, -1).expand(self._n_local, -1)
        return t.reshape(self._n_local, -1)

    def _pack(self, obs: Dict[str, torch.Tensor], reward: Optional[torch.Tensor], term=None, trunc=None,
              info: Optional[Dict[str, Any]] = None) -> torch.Tensor:
        parts = [obs[k].reshape(self._n_local, -1).float() for k in self._obs_keys]
        if reward is None:
            # reset: the reference returns one info dict per worker = per env (parallel_env.py:222-231).  The layout is taken
            # from THIS reset's dict (a cached one would mis-slice the gathered block when a later reset -- another mode, a
            # randomised state -- reports another key set; same env class and call on every rank, so the ranks agree)
            self._reset_info_layout = self._layout_of(info)
            parts += [self._per_env(info[k]) for k, _, _ in self._reset_info_layout]
        if reward is not None:
            parts.append(reward.reshape(self._n_local, -1).float())
            parts.append(self._per_env(term))
            parts.append(self._per_env(trunc))
            # taken from THIS step's dict, like the reset layout (a cached layout mis-slices the gathered block once a
            # step reports another key set -- a mode switch, a failure flag); same env class and call on every rank, so the ranks agree
            self._info_layout = self._layout_of(info)
            parts += [self._per_env(info[k]) for k, _, _ in self._info_layout]
        return torch.cat(parts, dim=1)

    def _layout_of(self, info: Optional[Dict[str, Any]]) -> List:
        """(key, packed row shape, per-env shape or None for a per-shard value) of every numeric entry of an info dict."""
        out = []
        for k in sorted(info or {}):
            try:
                t = torch.as_tensor(info[k])
            except (TypeError, ValueError, RuntimeError):
                continue    # not numeric: stays on its shard
            per_env = tuple(t.shape[1:]) if t.dim() > 0 and t.shape[0] == self._n_local else None
            out.append((k, tuple(self._per_env(info[k]).shape[1:]), per_env))
        return out

    def _unpack_info(self, flat: torch.Tensor, off: int, layout: List) -> Dict[str, torch.Tensor]:
        info, n = {}, flat.shape[0]
        for k, w, shp in layout:
            size = int(np.prod(w))
            v = flat[:, off: off + size]
            info[k] = v.reshape((n,) + shp) if shp is not None else v.reshape(n, -1)
            off += size
        return info

    def _unpack(self, flat: torch.Tensor, obs_like: Dict[str, torch.Tensor], with_reward: bool):
        out, off = {}, 0
        n = flat.shape[0]
        for k in self._obs_keys:
            shp = obs_like[k].shape[1:]
            size = int(np.prod(shp)) if len(shp) else 1
            out[k] = flat[:, off: off + size].reshape((n,) + tuple(shp))
            off += size
        if not with_reward:
            return out, None, None, None, self._unpack_info(flat, off, self._reset_info_layout or [])
        info_w = sum(int(np.prod(w)) for _, w, _ in self._info_layout)
        r_w = flat.shape[1] - off - 2 - info_w
        reward = flat[:, off: off + r_w].reshape(n, *(() if r_w == 1 else (-1,)))
        off += r_w
        term, trunc = flat[:, off] != 0, flat[:, off + 1] != 0
        off += 2
        return out, reward, term, trunc, self._unpack_info(flat, off, self._info_layout)

    # ------------------------------------------------------------------ env API (collective)
    # Every public method is a COLLECTIVE call: all ranks that are not inside serve() must call it; the
    # driver's arguments win (they travel in the command broadcast).
    def seed(self, seed: int = 0) -> None:
        _, a, _, _ = self._send(Command.SEED, int(seed))
        self._seed_shard(a)

    def _seed_shard(self, seed: int) -> None:
        if self._lanes == 1:
            self._env.seed(seed + self.rank)
        else:
            self._env.seed(seed, base=self.rank * self._lanes)

    def _load_shard(self, idx: int, mode) -> None:
        if self._lanes == 1:
            self._env.load_initial_domain(idx * self.world + self.rank, mode)
        else:
            self._env.load_initial_domain(idx, mode, base=self.rank * self._lanes, stride=self.world * self._lanes)

    def reset(self, seed: Optional[int] = None, randomize: Optional[bool] = None):
        """All shards reset with seeds ``seed + rank``; returns the observations of all ``num_envs`` envs stacked along dim 0
        and one info dict per ENV gathered from all shards (the reference: one per worker = per env, parallel_env.py:222-231).
        Departure: the reference hands the SAME seed to every worker (parallel_env.py:133-135, 226-228) -- its envs then differ
        only when their initial domains are drawn differently; here a shard holds a batch whose envs draw from one generator,
        so shards get ``seed + rank`` or every shard would replay the same batch."""
        _, a, b, _ = self._send(Command.RESET, -1 if seed is None else int(seed),
                                -1 if randomize is None else int(randomize))
        return self._do_reset(None if a < 0 else a, None if b < 0 else bool(b))

    def _do_reset(self, seed, randomize):
        if self._lanes == 1:
            obs, info = self._env.reset(seed=None if seed is None else int(seed) + self.rank, randomize=randomize)
        else:
            obs, info = self._env.reset(seed=seed, randomize=randomize, base=self.rank * self._lanes)
        flat = self._all_gather(self._pack(obs, None, info=info))
        obs_all, _, _, _, info_all = self._unpack(flat, obs, with_reward=False)
        host = {k: v.cpu() for k, v in info_all.items()}       # one copy per key, sliced per env on the host
        infos = [{k: v[i] for k, v in host.items()} for i in range(self._n_total)]
        return self._agents_to_rows(obs_all), infos

    def step(self, action: Optional[torch.Tensor] = None):
        """Driver: ``action [num_envs, ...]``.  Other ranks in SPMD mode pass ``None``.  Returns the reference's tuple
        (``parallel_env.py:233-287``): observations of all envs, rewards ``[num_envs, ...]``, and per-env LISTS of
        terminated / truncated flags and info dicts (the flags come back as one host read of the gathered block)."""
        if self.is_driver:
            # multi-agent: one row per agent of every env, envs concatenated (what sample_action returns, reference
            # parallel_env.py:356-359), or [num_envs, n_agents, ...]
            ok = (self._n_total, self._n_total * self._env.n_agents) if self._env.use_marl else (self._n_total,)
            if action is None or action.shape[0] not in ok:
                raise ValueError(f"Expected action batch size {ok[-1]}, but got "
                                 f"{None if action is None else action.shape[0]}")
        if not self._collective:
            return self._step_local(action)
        self._send(Command.STEP, action=action, read_header=False)
        return self._do_step()

    def _step_local(self, action: torch.Tensor):
        """World size 1 without forced collectives: nothing to broadcast or gather, so the step is the shard's own ``env.step`` and
        the reference's return type is built from its results directly -- no message buffer, no packed block, no device read of flags
        the env already holds as Python values (round 6: the wrapper cost 0.36 ms of a 7.2 ms headline step, most of it 128 tensor
        slices for the per-env info dicts and host-to-device copies of two booleans)."""
        n = self._n_total
        obs, reward, term, trunc, info = self._env.step(action.to(self._device).reshape(self._a_shape))

        def flags(v) -> List[bool]:
            if isinstance(v, torch.Tensor):
                vals = v.reshape(-1).cpu().tolist()
                return [bool(x) for x in (vals if len(vals) == n else vals * n)]
            return [bool(v)] * n

        cols = {}
        for k, v in info.items():
            cols[k] = v.unbind(0) if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else (v,) * n
        infos = [{k: c[i] for k, c in cols.items()} for i in range(n)]
        return self._agents_to_rows(obs), reward, flags(term), flags(trunc), infos

    def _do_step(self, lists: bool = True):
        """``lists=False`` (workers inside ``serve()``): nobody reads this rank's return value, so the device -> host read of the
        flags and the per-env dicts are skipped -- only the driver needs the Python lists of the reference's return type."""
        full = self._actions_from_msg()
        mine = full[self.rank * self._n_local: (self.rank + 1) * self._n_local]
        if self.time_shards:       # (bench.py --gpus N: this rank's own step time, so that load imbalance between shards is visible)
            import time

            if self._device.type == "cuda":
                torch.cuda.synchronize(self._device)
            t0 = time.perf_counter()
        obs, reward, term, trunc, info = self._env.step(mine)
        if self.time_shards:
            if self._device.type == "cuda":
                torch.cuda.synchronize(self._device)
            self.shard_seconds += time.perf_counter() - t0
        flat = self._all_gather(self._pack(obs, reward, term, trunc, info))  # everything a step returns: one collective
        obs_all, reward_all, term_all, trunc_all, info_all = self._unpack(flat, obs, with_reward=True)
        if not lists:
            return None
        obs_all = self._agents_to_rows(obs_all)
        flags = torch.stack([term_all, trunc_all]).cpu().tolist()
        infos = [{k: v[i] for k, v in info_all.items()} for i in range(self._n_total)]
        return obs_all, reward_all, [bool(x) for x in flags[0]], [bool(x) for x in flags[1]], infos

    def _agents_to_rows(self, obs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Multi-agent observations are concatenated over envs, ``[num_envs * n_agents, ...]`` (reference
        ``__aggregate_obs``, parallel_env.py:192-200); single-agent ones stay stacked ``[num_envs, ...]``."""
        if not self._env.use_marl:
            return obs
        return {k: v.reshape((-1,) + tuple(v.shape[2:])) for k, v in obs.items()}

    def sample_action(self) -> torch.Tensor:
        """Collective like every other call: each shard samples from its own generator, one all_gather."""
        self._send(Command.SAMPLE_ACTION, read_header=False)
        return self._do_sample()

    def _do_sample(self) -> torch.Tensor:
        a = self._all_gather(self._env.sample_action().to(self._device))
        return a.reshape((-1,) + tuple(a.shape[2:])) if self._env.use_marl else a

    def train(self) -> None:
        self._send(Command.TRAIN, read_header=False)
        self._env.train()

    def val(self) -> None:
        self._send(Command.VAL, read_header=False)
        self._env.val()

    def test(self) -> None:
        self._send(Command.TEST, read_header=False)
        self._env.test()

    _MODES = (None, EnvMode.TRAIN, EnvMode.VAL, EnvMode.TEST)

    def load_initial_domain(self, idx: int = 0, mode=None) -> None:
        _, a, b, _ = self._send(Command.LOAD_INITIAL_DOMAIN, int(idx), self._MODES.index(mode))
        self._load_shard(a, self._MODES[b])

    def get_state(self) -> Any:
        raise NotImplementedError("get_state is not implemented for ParallelFluidEnv.")   # parallel_env.py:370-378

    def set_state(self, state: Any) -> None:
        raise NotImplementedError("set_state is not implemented for ParallelFluidEnv.")

    def save_gif(self, filename: str, output_path=None) -> None:
        raise NotImplementedError("save_gif is not implemented for ParallelFluidEnv.")

    def get_uncontrolled_episode_metrics(self):
        raise NotImplementedError("get_uncontrolled_episode_metrics is not implemented for ParallelFluidEnv.")

    def detach(self) -> None:
        raise NotImplementedError("detach is not implemented for ParallelFluidEnv.")

    # ------------------------------------------------------------------ worker loop
    def serve(self) -> None:
        """Follow the driver's commands until CLOSE (the reference's ``_worker``, parallel_env.py:115-160)."""
        assert not self.is_driver
        while True:
            cmd, a, b, _ = self._send()
            if cmd == Command.STEP:
                self._do_step(lists=False)
            elif cmd == Command.RESET:
                self._do_reset(None if a < 0 else a, None if b < 0 else bool(b))
            elif cmd == Command.SEED:
                self._seed_shard(a)
            elif cmd == Command.TRAIN:
                self._env.train()
            elif cmd == Command.VAL:
                self._env.val()
            elif cmd == Command.TEST:
                self._env.test()
            elif cmd == Command.SAMPLE_ACTION:
                self._do_sample()
            elif cmd == Command.LOAD_INITIAL_DOMAIN:
                self._load_shard(a, self._MODES[b])
            elif cmd == Command.CLOSE:
                break
        self._shutdown()

    def close(self) -> None:
        if self._collective:
            self._send(Command.CLOSE, read_header=False)
        self._shutdown()
        for p in self._workers:
            p.join(timeout=30)

    def _shutdown(self) -> None:
        self._env.close()
        if self._owns_group and dist.is_initialized():
            dist.destroy_process_group()
            self._owns_group = False
