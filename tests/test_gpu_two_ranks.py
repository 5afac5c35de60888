"""Multi-GPU readiness that one GPU can prove: TWO ranks with the real library share ``cuda:0`` under
``torch.distributed.run`` (backend ``gloo``, host-staged collectives: ``tests/two_rank_child.py``), each stepping its shard of
``ParallelFluidEnv(env_id, num_envs=8)``; what the driver returns must equal, bit for bit, two independent 4-env processes' worth of
plain envs seeded ``seed`` and ``seed + 1`` -- sharding, per-shard seeds, the one broadcast and the one all_gather per command, per-env
flags and infos.  Reference behaviour kept: ``envs/parallel_env.py:115-175, 233-287`` (one worker per GPU over pipes)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("env_id,kv,kw", [
    ("CylinderJet2D-easy-v0", ("randomize_initial_state=false",), dict(randomize_initial_state=False)),
    ("ChannelJet2D-v0", ("resolution_x=64", "resolution_y=32"), dict(resolution_x=64, resolution_y=32)),
])
def test_two_ranks_on_one_gpu_reproduce_two_independent_shards(env_id, kv, kw, tmp_path):
    import fluidgym_amd

    n = 8
    out = str(tmp_path / "driver.pt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "two_rank_child.py"), env_id, str(n), out, *kv]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, f"ranks failed:\n{run.stdout[-3000:]}\n{run.stderr[-6000:]}"
    rec = torch.load(out, weights_only=False)
    checks = 0
    for r in range(2):          # shard r alone: what rank r ran
        plain = fluidgym_amd.make(env_id, num_envs=n // 2, **kw)
        try:
            plain.seed(5 + r)
            o, info = plain.reset(seed=7 + r)
            sl = slice(r * n // 2, (r + 1) * n // 2)
            for k in o:
                assert torch.equal(rec["reset_obs"][k][sl], o[k].cpu().float()), ("reset", k, r)
                checks += 1
            for step in range(3):
                a = rec["actions"][step][sl].to(plain.cuda_device)
                o, rew, term, trunc, info = plain.step(a)
                got = rec["steps"][step]
                for k in o:
                    assert torch.equal(got["obs"][k][sl], o[k].cpu().float()), ("obs", k, r, step)
                    checks += 1
                assert torch.equal(got["reward"][sl], rew.cpu().float()), ("reward", r, step)
                t = torch.as_tensor(term).reshape(-1)
                assert got["term"][sl] == [bool(x) for x in (t.expand(n // 2) if t.numel() == 1 else t).tolist()]
                for k, v in info.items():
                    try:
                        tv = torch.as_tensor(v).float().cpu()
                    except (TypeError, ValueError, RuntimeError):
                        continue
                    if tv.dim() > 0 and tv.shape[0] == n // 2:
                        assert torch.equal(torch.stack([got["infos"][r * n // 2 + i][k].reshape(tv.shape[1:]) for i in range(n // 2)]), tv), ("info", k)
                        checks += 1
        finally:
            plain.close()
    assert checks >= 12


def test_bench_world_2_path_runs_on_one_gpu():
    """``bench.py --gpus 2 --share-gpu``: the N > 1 branch of the bench -- barrier, max-over-ranks time, per-rank shard times and
    sub-steps -- as a dry run on the one GPU (gloo, both ranks on cuda:0).  What SCALE will run with RCCL on a real node; the value is
    not a scaling figure and says so."""
    import json

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--envs-per-gpu", "4",
           "--no-micro", "--no-cpu-baseline"]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, f"bench failed:\n{run.stdout[-3000:]}\n{run.stderr[-6000:]}"
    line = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    pr = line["config"]["per_rank"]
    assert len(pr["shard_ms_per_step"]) == 2 and pr["min_ms"] > 0 and pr["imbalance"] >= 1.0
    assert len(pr["mean_substeps_per_sim_step"]) == 2
    assert "dry run" in line["config"]["parallelism"]
