"""Share of an env step that is NOT the native simulation step: wall time of env.step against the same number of bare
`sim.single_step()` calls (jets, forcing, forces, observations, reward = the difference).
python profiles/env_host_share.py ENV_ID NUM_ENVS [steps=2] [key=value ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402

env_id, B = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
kw = {}
for a in sys.argv[4:]:
    k, v = a.split("=")
    kw[k] = json.loads(v)
env = fluidgym_amd.make(env_id, num_envs=B, **kw)
env.reset(seed=0)
env.step(env.sample_action())
sync = torch.cuda.synchronize
sync(); t0 = time.perf_counter()
for _ in range(steps):
    env.step(env.sample_action())
sync(); t_env = (time.perf_counter() - t0) / steps
n = env._n_sim_steps
sync(); t0 = time.perf_counter()
for _ in range(steps * n):
    env._sim.single_step()
sync(); t_sim = (time.perf_counter() - t0) / steps
print(json.dumps({"env": env_id, "envs": B, "sim_steps_per_env_step": n, "env_step_ms": round(1e3 * t_env, 2), "bare_sim_steps_ms": round(1e3 * t_sim, 2),
                  "glue_share": round(1.0 - t_sim / t_env, 3), "env_steps_per_s": round(B / t_env, 1)}))
env.close()
