"""The one JSON line bench.py prints must stay parseable by the driver, which keeps only a tail of stdout (round 2's 23 KB line
was cut: BENCH_r02.json `parsed: null`).  Built here from a canned full result (the round-2 profile run, regrouped the way
bench.py now groups its legs) without touching a GPU."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEGS = ("quiescent_mode", "warm_start_mode", "rbc_env", "tcf_env", "cylinder_env", "cylinder_env_256", "airfoil_env",
        "airfoil_env_16", "airfoil_env_multilevel_trial_mode")


def _canned():
    with open(os.path.join(ROOT, "profiles", "r02_bench.json")) as f:
        full = json.load(f)
    out = {k: v for k, v in full.items() if k not in LEGS}
    out["legs"] = {k: full[k] for k in LEGS if k in full}
    out["legs"]["large_env"] = dict(full["rbc_env"], env_id="ChannelJet2D-large-v0")
    out["legs"]["broken_leg"] = {"error": "RuntimeError: " + "x" * 400, "leg_seconds": 0.1}
    out["config"]["workload_modified"] = True
    out["config"]["iters_are"] = "iterations per solve (counts; 0 = initial residual met the tolerance)"
    out["config"]["launches_per_piso_step"] = 87.3
    return out


def test_line_is_short_and_round_trips():
    out = _canned()
    text = bench.compact_line(out)
    assert "\n" not in text
    assert len(text) < 4096, len(text)
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == float(f"{out['value']:.6g}")
    assert line["config"]["workload"] and "model" not in line["config"]
    r = line["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    assert isinstance(r["traffic"], (int, float)) and r["traffic"] > 0
    assert set(line["poisson_256"]) == {"jacobi_sweep", "apply", "cg_iteration"}
    cb = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb)
    assert set(line["legs"]) == set(out["legs"])
    assert line["legs"]["cylinder_env"]["value"] > 0 and "iters" in line["legs"]["cylinder_env"]
    assert len(line["legs"]["broken_leg"]["error"]) <= 80


def test_line_survives_oversized_legs():
    out = _canned()
    for i in range(200):
        out["legs"][f"extra_{i}"] = dict(out["legs"]["rbc_env"])
    text = bench.compact_line(out)
    assert len(text) < 4096
    line = json.loads(text)
    assert line["value"] > 0 and line["roofline"]["frac"] > 0 and "dropped" in line["legs"]


def test_detail_file_is_written(tmp_path):
    out = _canned()
    target = tmp_path / "detail.json"
    bench.write_detail(out, str(target))
    assert json.loads(target.read_text())["roofline"]["kernels"]
