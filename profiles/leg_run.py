"""One bench leg on its own, for rocprofv3 (kernel statistics / PMC passes per leg: profiles/r04_{f_tcf,g_rbc,h_large}_*):
    python profiles/leg_run.py ENV_ID NUM_ENVS [steps=5] [warmup=1] [forcing=0] [key=value ...]
runs `warmup` + `steps` env steps of the env with uniform random actions (the bench legs' policy) and prints env-steps/s and the
iteration counters of the timed steps."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402

env_id, B = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
warmup = int(sys.argv[4]) if len(sys.argv) > 4 else 1
forcing = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
kw = {}
for a in sys.argv[6:]:
    k, v = a.split("=")
    kw[k] = json.loads(v)
env = fluidgym_amd.make(env_id, num_envs=B, **kw)
env.reset(seed=5)
env.seed(5)
dev = env.cuda_device
gen = torch.Generator(device=dev).manual_seed(4321)
blk0 = env._domain.getBlock(0) if forcing > 0 else None
if forcing > 0:
    blk0.setVelocitySource(torch.zeros_like(blk0.velocity))


def one():
    if forcing > 0:
        blk0.velocitySource.normal_(0.0, forcing, generator=gen)
    env.step(env.sample_action())


for _ in range(warmup):
    one()
solver = getattr(env._domain, "solver", env._domain)
solver.solver_counters(reset=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    one()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / steps
c = solver.solver_counters()
its = {k: [round(v["mean"], 2), v["max"]] for k, v in c.items() if isinstance(v, dict) and v["systems"]}
print(json.dumps({"env": env_id, "envs": B, "steps": steps, "ms_per_env_step": round(1e3 * el, 3), "env_steps_per_s": round(B / el, 1),
                  "piso_steps": c["piso_steps"], "iters[mean,max]": its}))
env.close()
