"""``python bench.py --gpus N`` launched plainly (no WORLD_SIZE): the parent must start one rank per GPU as CHILD processes through
the standard launcher and exit with their code -- before anything touches a GPU (it never imports torch on that path)."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_multi_gpu_invocation_relaunches_through_the_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    import subprocess

    seen = {}

    def fake_call(cmd, *a, **k):
        seen["cmd"] = list(cmd)
        seen["torch_cuda_initialised"] = "torch" in sys.modules and sys.modules["torch"].cuda.is_initialized()
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the children's exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["torch_cuda_initialised"] is False


def test_rank_count_must_match_the_gpus_argument(monkeypatch):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(AssertionError, match="WORLD_SIZE=2"):
        bench.main()
