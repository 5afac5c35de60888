"""Airfoil2D on the GPU: development from the projected uniform stream, time per env step, forces.
    python profiles/airfoil_run.py [num_envs] [develop_steps] [env_steps] [pressure_use_BiCG: 2 refined BiCGStab | 0 CG]"""
import sys, time, json
import torch
import fluidgym_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 2
env = fluidgym_amd.make("Airfoil2D-easy-v0", num_envs=B, initial_domain_steps=dev, randomize_initial_state=False, pressure_use_BiCG=mode)
t0 = time.time()
env.reset(seed=0)
torch.cuda.synchronize()
t_dev = time.time() - t0
print(json.dumps({"cells": env._domain.n_cells, "develop_steps": dev, "develop_s": round(t_dev, 2), "s_per_piso_step": round(t_dev / max(dev, 1), 4),
                  "pressure_use_BiCG": mode}), flush=True)
for i in range(n):
    a = torch.zeros(B, 3, device="cuda") if i < n // 2 else env.sample_action()
    t0 = time.time()
    obs, r, term, trunc, info = env.step(a)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(json.dumps({"step": i, "controlled": i >= n // 2, "s": round(dt, 3), "env_steps_per_s": round(B / dt, 2),
                      "drag": [round(x, 4) for x in info["drag"].tolist()], "lift": [round(x, 4) for x in info["lift"].tolist()],
                      "iterations": list(env._sim.last_iterations), "substeps": env._sim.last_substeps}), flush=True)
