"""Soak run of CylinderJet2D-easy-v0 x B with random jets: every solve converged, every env finite, drag statistics; run once per
layout of the on-chip CG (FG_MB_OC_AGG=1 / 0) the two must tell the same story.  python profiles/cylinder_soak.py [envs=64] [steps=120]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
fluidgym_amd.set_solver_policy(pressure_multilevel=True)
env = fluidgym_amd.make("CylinderJet2D-easy-v0", num_envs=B, initial_domain_steps=100, randomize_initial_state=True, episode_length=steps + 5)
env.reset(seed=3)
gen = torch.Generator(device="cpu").manual_seed(17)
drag, lift, rew = [], [], []
t0 = time.perf_counter()
for k in range(steps):
    a = (torch.rand(B, 1, generator=gen) * 2 - 1).cuda()
    obs, r, term, trunc, info = env.step(a)
    drag.append(info["drag"].cpu().numpy()); lift.append(info["lift"].cpu().numpy()); rew.append(r.cpu().numpy())
torch.cuda.synchronize()
el = time.perf_counter() - t0
drag, lift, rew = np.stack(drag), np.stack(lift), np.stack(rew)
c = env._domain.solver_counters()
print(json.dumps({"agg": os.environ.get("FG_MB_OC_AGG", "1"), "envs": B, "env_steps": steps, "env_steps_per_s": round(B * steps / el, 1),
                  "finite": bool(np.isfinite(drag).all() and np.isfinite(lift).all() and all(torch.isfinite(v).all().item() for v in obs.values())),
                  "drag_mean": round(float(drag.mean()), 4), "drag_min": round(float(drag.min()), 4), "drag_max": round(float(drag.max()), 4),
                  "lift_rms": round(float(np.sqrt((lift ** 2).mean())), 4), "reward_mean": round(float(rew.mean()), 4),
                  "pressure_iterations": [c["pressure0"]["mean"], c["pressure0"]["max"], c["pressure1"]["mean"], c["pressure1"]["max"]],
                  "unconverged": {k: v["unconverged"] for k, v in c.items() if isinstance(v, dict)}, "env_status_max": int(np.max(env._sim.last_env_status))}))
env.close()
