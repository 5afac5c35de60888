"""CylinderJet3D on the GPU: development from the projected uniform stream, time per env step, forces.
    python profiles/cylinder3d_run.py [resolution] [num_envs] [develop_steps] [env_steps]"""
import sys, time, json
import torch
import fluidgym_amd

res = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = int(sys.argv[3]) if len(sys.argv) > 3 else 50
n = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n_jets = 8 if res % 8 == 0 else 4
env = fluidgym_amd.make(f"CylinderJet3D-easy-v0", num_envs=B, resolution=res, n_jets=n_jets, initial_domain_steps=dev,
                        randomize_initial_state=False)
t0 = time.time()
env.reset(seed=0)
torch.cuda.synchronize()
t_dev = time.time() - t0
print(json.dumps({"cells": env._domain.n_cells, "develop_steps": dev, "develop_s": round(t_dev, 2),
                  "s_per_piso_step": round(t_dev / max(dev, 1), 4)}), flush=True)
env._domain.velocity.add_(0.025 * torch.randn_like(env._domain.velocity))   # what _randomize_domain adds
env._domain.profile_enable(True)
for i in range(n):
    a = env.sample_action()
    t0 = time.time()
    obs, r, term, trunc, info = env.step(a)
    torch.cuda.synchronize()
    dt = time.time() - t0
    w = env._domain.velocity[:, 2].abs().max().item()
    print(json.dumps({"step": i, "s": round(dt, 2), "env_steps_per_s": round(B / dt, 3), "drag": info["drag"].tolist(),
                      "lift": info["lift"].tolist(), "max_w": round(w, 4), "reward": r.tolist()}), flush=True)
prof = env._domain.profile_read()
for name, r in prof.items():
    if r["samples"] > 0:
        print(json.dumps({"kernel": name, "avg_ms": round(r["ms"] / r["samples"], 5), "samples": r["samples"], "launches": r["launches"],
                          "GBps": round(r["bytes"] / r["ms"] / 1e6, 1), "frac_of_8TBps": round(r["bytes"] / r["ms"] / 1e6 / 8000, 3)}))
print(json.dumps({"substeps": env._sim.last_substeps, "iterations": list(env._sim.last_iterations)}))
