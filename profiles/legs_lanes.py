"""RBC / TCF / large-channel legs as 1, 2 and 4 lanes on one GPU (bench.env_leg(lanes=...)).  python profiles/legs_lanes.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for env_id, n, steps, forcing in (("RBC2D-baseline-v0", 32, 8, 0.0), ("TCF3D-baseline-v0", 8, 8, 0.0), ("ChannelJet2D-large-v0", 64, 5, 2.0)):
    for lanes in (1, 2, 4, 1, 2):
        r = bench.env_leg(env_id, n, dev, steps=steps, warmup=1, forcing=forcing, lanes=lanes, **({"seed": 1234} if forcing else {}))
        print(json.dumps({"env": env_id, "lanes": lanes, "value": round(r["value"], 1), "ms": round(r["ms_per_step"], 2)}), flush=True)
