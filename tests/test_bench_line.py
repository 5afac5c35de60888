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


def test_lane_profiles_and_iteration_tables_merge_like_one_handle():
    """Two lanes = two solver handles: the live profiles add field by field, the iteration tables add systems / unconverged / PISO steps,
    take the maximum of the maxima and weight the means by systems (bench.merge_profiles / merge_iterations)."""
    a = {"k_x": dict(ms=1.0, samples=2, bytes=10.0, flops=1.0, full_ms=0.5, full_bytes=5.0, full_samples=1, launches=16, all_ms=1.5, all_samples=3)}
    b = {"k_x": dict(ms=3.0, samples=4, bytes=30.0, flops=2.0, full_ms=1.5, full_bytes=15.0, full_samples=3, launches=32, all_ms=3.5, all_samples=5),
         "k_y": dict(ms=1.0, samples=1, bytes=1.0, flops=0.0, full_ms=0.0, full_bytes=0.0, full_samples=0, launches=8, all_ms=1.0, all_samples=1)}
    m = bench.merge_profiles([a, b])
    assert m["k_x"]["ms"] == 4.0 and m["k_x"]["samples"] == 6 and m["k_x"]["launches"] == 48 and m["k_x"]["all_samples"] == 8
    assert m["k_y"] == b["k_y"] and a["k_x"]["ms"] == 1.0      # (the inputs are left alone)
    i0 = {"piso_steps": 10, "velocity": {"mean": 12.0, "max": 12, "unconverged": 0, "systems": 100}, "pressure0": {"mean": 1.0, "max": 2, "unconverged": 1, "systems": 50}}
    i1 = {"piso_steps": 12, "velocity": {"mean": 6.0, "max": 14, "unconverged": 2, "systems": 300}}
    it = bench.merge_iterations([i0, i1])
    assert it["piso_steps"] == 22
    assert it["velocity"] == {"mean": 7.5, "max": 14, "unconverged": 2, "systems": 400}
    assert it["pressure0"] == {"mean": 1.0, "max": 2, "unconverged": 1, "systems": 50}
    assert bench.merge_iterations([i0]) is i0
