"""Where the host time of a CylinderJet2D sim step goes: wall time of its three parts (jets, PISO step, drag / lift) with a device
synchronisation after each, against the same loop unsynchronised.  python profiles/cylinder_host_time.py [envs=64]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fluidgym_amd.set_solver_policy(pressure_multilevel=True)
env = fluidgym_amd.make("CylinderJet2D-easy-v0", num_envs=B, initial_domain_steps=100, randomize_initial_state=False)
env.reset(seed=0)
gen = torch.Generator(device="cpu").manual_seed(7)
act = lambda: (torch.rand(B, 1, generator=gen) * 2 - 1).cuda()
env.step(act())
n = env._n_sim_steps
seg = {"apply_action": 0.0, "single_step": 0.0, "drag_lift": 0.0}
target = act().reshape(B, 1)
sync = torch.cuda.synchronize
sync()
for _ in range(n):
    t0 = time.perf_counter(); env._apply_action(target); sync()
    t1 = time.perf_counter(); env._sim.single_step(); sync()
    t2 = time.perf_counter(); env._get_drag_and_lift(); sync()
    t3 = time.perf_counter()
    seg["apply_action"] += t1 - t0; seg["single_step"] += t2 - t1; seg["drag_lift"] += t3 - t2
out = {k: round(1e6 * v / n, 1) for k, v in seg.items()}
sync(); t0 = time.perf_counter()
for _ in range(n):
    env._apply_action(target); env._sim.single_step(); env._get_drag_and_lift()
sync()
out["unsynchronised_sim_step_us"] = round(1e6 * (time.perf_counter() - t0) / n, 1)
t0 = time.perf_counter()
for _ in range(n):
    env._sim.single_step()
sync()
out["piso_only_us"] = round(1e6 * (time.perf_counter() - t0) / n, 1)
print(json.dumps(out))
env.close()
