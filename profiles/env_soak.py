"""Soak run of one env id with random actions: finiteness of observations / rewards, unconverged solves, throughput.
python profiles/env_soak.py ENV_ID NUM_ENVS STEPS [key=value ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402

env_id, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kw = {}
for a in sys.argv[4:]:
    k, v = a.split("=")
    kw[k] = json.loads(v)
env = fluidgym_amd.make(env_id, num_envs=B, **kw)
env.reset(seed=5)
finite, n = True, 0
t0 = time.perf_counter()
while n < steps:
    obs, rew, term, trunc, info = env.step(env.sample_action())
    n += 1
    finite = finite and all(torch.isfinite(v).all().item() for v in obs.values()) and torch.isfinite(rew).all().item()
    if trunc if isinstance(trunc, bool) else bool(trunc):
        env.reset(seed=5 + n)
torch.cuda.synchronize()
el = time.perf_counter() - t0
solver = env._domain.solver if hasattr(env._domain, "solver") else env._domain
c = solver.solver_counters()
print(json.dumps({"env": env_id, "envs": B, "env_steps": n, "finite": bool(finite), "env_steps_per_s": round(B * n / el, 1),
                  "iterations": {k: [round(v["mean"], 2) if v["mean"] is not None else None, v["max"]] for k, v in c.items() if isinstance(v, dict)},
                  "unconverged": {k: v["unconverged"] for k, v in c.items() if isinstance(v, dict)}}))
env.close()
