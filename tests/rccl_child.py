"""Child of tests/test_gpu_rccl.py (and of the CPU twin in tests/test_parallel_env_gloo.py): ONE rank started by
``python -m torch.distributed.run --nproc-per-node 1`` that builds ``ParallelFluidEnv(..., force_collectives=True)`` so that
every command goes through the process group's broadcast / all_gather at world size 1 -- on a GPU box backend ``nccl`` = RCCL
(``init_process_group``, the int32 message broadcast, ``all_gather_into_tensor``: reference behaviour to preserve is
``envs/parallel_env.py:115-175, 233-287``) -- and holds reset / step / sample_action against the plain batched env with
``torch.equal``.  Prints one JSON line; exit code 0 = everything equal.

usage: rccl_child.py <backend> <env_id> <num_envs> [key=value ...]      (values are parsed with ``json.loads``)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class _PlainLanes:
    """L plain env batches stepped one after the other, seeds ``seed + l``: what ``ParallelFluidEnv(lanes=L)`` must reproduce."""

    def __init__(self, envs):
        self.envs = envs
        self.action_space = envs[0].action_space
        z = envs[0]._zero_action
        self._zero_action = z.new_zeros((len(envs) * z.shape[0],) + tuple(z.shape[1:]))

    def seed(self, s):
        for l, e in enumerate(self.envs):
            e.seed(s + l)

    def reset(self, seed):
        res = [e.reset(seed=seed + l) for l, e in enumerate(self.envs)]
        return {k: torch.cat([r[0][k] for r in res]) for k in res[0][0]}, [r[1] for r in res]

    def sample_action(self):
        return torch.cat([e.sample_action() for e in self.envs])

    def step(self, a):
        n = a.shape[0] // len(self.envs)
        res = [e.step(a[l * n: (l + 1) * n]) for l, e in enumerate(self.envs)]
        info = {k: torch.cat([torch.as_tensor(r[4][k]) for r in res]) for k in res[0][4]
                if isinstance(res[0][4][k], torch.Tensor) and res[0][4][k].dim() > 0}
        return {k: torch.cat([r[0][k] for r in res]) for k in res[0][0]}, torch.cat([r[1] for r in res]), res[0][2], res[0][3], info

    def close(self):
        for e in self.envs:
            e.close()


def main() -> int:
    backend, env_id, num_envs = sys.argv[1], sys.argv[2], int(sys.argv[3])
    kw = {}
    for item in sys.argv[4:]:
        k, v = item.split("=", 1)
        kw[k] = json.loads(v)
    if backend == "gloo":
        # CPU twin: a toy env registered by the gloo test module stands in for the registry env
        from tests.test_parallel_env_gloo import _register

        _register()
    import fluidgym_amd
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    lanes = int(kw.pop("lanes", 1))      # lanes=L: the rank's shard as L sub-batches on their own HIP streams (ParallelFluidEnv "Lanes")
    penv = ParallelFluidEnv(env_id, num_envs=num_envs, backend=backend, force_collectives=True, lanes=lanes, **kw)
    assert dist.is_initialized() and dist.get_backend() == backend and penv.world == 1 and penv._collective
    plain_kw = dict(kw)
    if backend == "nccl":
        plain_kw["cuda_device"] = penv._device
    if lanes > 1:
        plain = _PlainLanes([fluidgym_amd.make(env_id, num_envs=num_envs // lanes, **plain_kw) for _ in range(lanes)])
    else:
        plain = fluidgym_amd.make(env_id, num_envs=num_envs, **plain_kw)
    report = {"backend": dist.get_backend(), "world": penv.world, "device": str(penv._device), "checks": 0, "lanes": lanes}

    def same(a, b, what):
        assert torch.equal(a.to(b.device), b), what
        report["checks"] += 1

    penv.seed(5)
    plain.seed(5)
    o_p, info_p = penv.reset(seed=7)
    o_s, info_s = plain.reset(seed=7)
    for k in o_s:
        same(o_p[k], o_s[k], f"reset obs {k}")
    assert len(info_p) == num_envs
    same(penv.sample_action(), plain.sample_action().to(penv._device), "sample_action")
    g = torch.Generator().manual_seed(0)
    lo, hi = float(plain.action_space.low.min()), float(plain.action_space.high.max())
    for step in range(3):
        a = (torch.rand(tuple(plain._zero_action.shape), generator=g) * (hi - lo) + lo).to(penv._device)
        r_p = penv.step(a)
        r_s = plain.step(a)
        for k in r_s[0]:
            same(r_p[0][k], r_s[0][k], f"step {step} obs {k}")
        same(r_p[1], r_s[1].float(), f"step {step} reward")
        t_s = torch.as_tensor(r_s[2]).reshape(-1).expand(num_envs) if torch.as_tensor(r_s[2]).numel() == 1 else torch.as_tensor(r_s[2])
        assert r_p[2] == [bool(x) for x in t_s.tolist()], "terminated flags"
        assert len(r_p[4]) == num_envs
        for k, v in r_s[4].items():
            try:
                t = torch.as_tensor(v).float().to(penv._device)
            except (TypeError, ValueError, RuntimeError):
                continue
            if t.dim() > 0 and t.shape[0] == num_envs:
                same(torch.stack([torch.as_tensor(i[k]).reshape(t.shape[1:]) for i in r_p[4]]), t, f"info {k}")
    penv.close()
    plain.close()
    assert not dist.is_initialized()          # the env owned the group and tore it down
    print(json.dumps(report))
    return 0


if __name__ == "__main__":
    sys.exit(main())
