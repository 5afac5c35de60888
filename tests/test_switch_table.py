"""docs/SWITCHES.md is generated from the getenv / os.environ sites of the source (profiles/gen_switch_table.py); this test fails when a
switch was added or removed without regenerating it, and checks that the library reports the switches of a handle (fg_mb_config_dump
works on a host-only handle; fg_config_dump needs a GPU and is exercised by bench.py / tests/test_gpu_fused_cg.py)."""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_switch_table_is_current():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "gen_switch_table.py"), "--check"], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, "docs/SWITCHES.md is stale: run python profiles/gen_switch_table.py\n" + out.stdout + out.stderr


def test_multi_block_handle_reports_its_switches(monkeypatch):
    from fluidgym_amd import _lib as L

    monkeypatch.setenv("FG_MB_COMPACT", "0")
    lib = L.load()
    h = ctypes.c_void_p()
    assert lib.fg_mb_create(2, 1, -1, ctypes.byref(h)) == 0
    try:
        small = ctypes.create_string_buffer(8)
        need = lib.fg_mb_config_dump(h, small, 8)
        assert need > 8                                  # too small: the length needed comes back
        buf = ctypes.create_string_buffer(need)
        assert lib.fg_mb_config_dump(h, buf, need) == 0
        cfg = json.loads(buf.value.decode())
        assert cfg["FG_MB_COMPACT"] == 0 and cfg["FG_MB_BICG_FUSE"] == 2 and cfg["FG_MB_ONCHIP"] == 1
    finally:
        lib.fg_mb_destroy(h)
