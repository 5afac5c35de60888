"""Airfoil3D at the reference's size (96 spanwise layers): time per env step, forces.
    python profiles/airfoil3d_run.py [develop_steps] [env_steps] [res_z]"""
import sys, time, json
import torch
import fluidgym_amd

dev = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
res_z = int(sys.argv[3]) if len(sys.argv) > 3 else 96
env = fluidgym_amd.make("Airfoil3D-easy-v0", num_envs=1, initial_domain_steps=dev, randomize_initial_state=False, res_z=res_z)
t0 = time.time()
env.reset(seed=0)
torch.cuda.synchronize()
print(json.dumps({"cells": env._domain.n_cells, "develop_steps": dev, "reset_s": round(time.time() - t0, 1),
                  "gpu_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}), flush=True)
for i in range(n):
    a = env.sample_action()
    t0 = time.time()
    obs, r, term, trunc, info = env.step(a)
    torch.cuda.synchronize()
    print(json.dumps({"step": i, "s": round(time.time() - t0, 2), "drag": round(float(info["drag"]), 4), "lift": round(float(info["lift"]), 4),
                      "iterations": list(env._sim.last_iterations), "substeps": env._sim.last_substeps,
                      "max_w": round(float(env._domain.velocity[:, 2].abs().max()), 4)}), flush=True)
