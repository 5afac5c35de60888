"""RCCL runs once on the one GPU the box has: ``ParallelFluidEnv`` with backend ``nccl`` under
``python -m torch.distributed.run --nproc-per-node 1`` and ``force_collectives=True``, so that every command takes the
``world > 1`` branches -- ``init_process_group``, the int32 message broadcast, ``all_gather_into_tensor`` -- and reset / step /
sample_action equal the plain batched env bit for bit (``tests/rccl_child.py``).  Reference behaviour kept:
``envs/parallel_env.py:115-175, 233-287``.  The rank is a CHILD process (nothing here execs from a process that holds the GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_child(backend, env_id, num_envs, *kv, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "rccl_child.py"), backend, env_id, str(num_envs), *kv]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, f"child failed:\n{out.stdout[-3000:]}\n{out.stderr[-6000:]}"
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("env_id,kv", [
    ("ChannelJet2D-v0", ("resolution_x=64", "resolution_y=32", "randomize_initial_state=false")),
    ("CylinderJet2D-easy-v0", ("randomize_initial_state=false",)),
])
def test_rccl_broadcast_and_all_gather_reproduce_the_plain_env(env_id, kv):
    rep = _run_child("nccl", env_id, 4, *kv)
    assert rep["backend"] == "nccl" and rep["world"] == 1 and rep["device"].startswith("cuda") and rep["checks"] >= 10, rep


def test_rccl_collectives_carry_the_lanes_of_a_rank():
    """``lanes=2`` under the RCCL branch (what every rank of ``bench.py --gpus N`` runs): the shard's two sub-batches are stepped on
    their own HIP streams, packed in lane order and carried by the group's broadcast / all_gather -- equal, bit for bit, to the two
    batches stepped alone with seeds ``seed`` and ``seed + 1``."""
    rep = _run_child("nccl", "ChannelJet2D-v0", 4, "resolution_x=64", "resolution_y=32", "randomize_initial_state=false", "lanes=2")
    assert rep["backend"] == "nccl" and rep["world"] == 1 and rep["lanes"] == 2 and rep["checks"] >= 10, rep

