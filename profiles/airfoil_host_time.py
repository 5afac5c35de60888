"""Host-time split of an Airfoil2D sim step (see cylinder_host_time.py).  python profiles/airfoil_host_time.py [envs=16] [dev_steps=40]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = int(sys.argv[2]) if len(sys.argv) > 2 else 40
env = fluidgym_amd.make("Airfoil2D-easy-v0", num_envs=B, initial_domain_steps=dev, randomize_initial_state=False)
env.reset(seed=0)
a = torch.zeros(B, 3, device="cuda")
env.step(a)
n = env._n_sim_steps
seg = {"apply_action": 0.0, "single_step": 0.0, "drag_lift": 0.0}
sync = torch.cuda.synchronize
sync()
for _ in range(n):
    t0 = time.perf_counter(); env._apply_action(a); sync()
    t1 = time.perf_counter(); env._sim.single_step(); sync()
    t2 = time.perf_counter(); env._get_drag_and_lift(); sync()
    t3 = time.perf_counter()
    seg["apply_action"] += t1 - t0; seg["single_step"] += t2 - t1; seg["drag_lift"] += t3 - t2
out = {k: round(1e6 * v / n, 1) for k, v in seg.items()}
out["substeps"] = env._sim.last_substeps
t0 = time.perf_counter(); obs = env._get_global_obs(); sync(); out["obs_us"] = round(1e6 * (time.perf_counter() - t0), 1)
print(json.dumps(out))
env.close()
