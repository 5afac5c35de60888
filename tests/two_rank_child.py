"""Child of tests/test_gpu_two_ranks.py: one of TWO ranks started by ``python -m torch.distributed.run --nproc-per-node 2`` that share
the box's one GPU.  Backend ``gloo`` with the envs on ``cuda:0`` (host-staged collectives), the REAL library in both processes:
``ParallelFluidEnv(env_id, cuda_ids=[0, 0], num_envs=N)`` in SPMD mode -- every rank makes the same calls, the driver's actions travel in
the command broadcast.  Rank 0 saves everything a user sees (reset / step observations, rewards, flags, per-env infos) for the parent
to compare with two independent single-process runs.  Reference behaviour: ``envs/parallel_env.py:115-175, 233-287``.

usage: two_rank_child.py <env_id> <num_envs> <out.pt> [key=value ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main() -> int:
    env_id, num_envs, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    kw = {}
    for item in sys.argv[4:]:
        k, v = item.split("=", 1)
        kw[k] = json.loads(v)
    from fluidgym_amd.envs.parallel_env import ParallelFluidEnv

    penv = ParallelFluidEnv(env_id, cuda_ids=[0, 0], num_envs=num_envs, backend="gloo", **kw)
    assert dist.is_initialized() and dist.get_backend() == "gloo" and penv.world == 2 and penv._collective
    assert penv._device.type == "cuda" and penv._comm_device.type == "cpu" and penv.local_env.num_envs == num_envs // 2
    rec = {"steps": []}
    penv.seed(5)
    obs, infos = penv.reset(seed=7)
    rec["reset_obs"] = {k: v.cpu() for k, v in obs.items()}
    rec["reset_infos"] = [{k: torch.as_tensor(v).cpu() for k, v in i.items()} for i in infos]
    g = torch.Generator().manual_seed(0)
    shape = (num_envs,) + tuple(penv.local_env._zero_action.shape[1:])
    lo, hi = float(penv.action_space.low.min()), float(penv.action_space.high.max())
    rec["actions"] = []
    for _ in range(3):
        a = torch.rand(shape, generator=g) * (hi - lo) + lo      # (every rank draws the same block; only the driver's travels)
        o, r, term, trunc, info = penv.step(a if penv.is_driver else None)
        rec["actions"].append(a)
        rec["steps"].append({"obs": {k: v.cpu() for k, v in o.items()}, "reward": r.cpu(), "term": term, "trunc": trunc,
                             "infos": [{k: torch.as_tensor(v).cpu() for k, v in i.items()} for i in info]})
    if penv.is_driver:
        torch.save(rec, out)
    penv.close()
    print(json.dumps({"rank": penv.rank, "world": 2, "ok": True}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
