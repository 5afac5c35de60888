"""CylinderJet2D-easy-v0 x B: env-steps/s and CG iterations per solve for the solver modes of the multi-block path.
    python profiles/cylinder_modes.py [envs=64] [steps=3] [modes: comma list of onchip{0,1}-warm{0,1}-multilevel{0,1}]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
modes = (sys.argv[3] if len(sys.argv) > 3 else "0-0-0,1-0-0,1-0-1,0-1-0,1-1-0,1-1-1").split(",")
for mode in modes:
    onchip, warm, ml = (mode.split("-") + ["0"])[:3]
    os.environ["FG_MB_ONCHIP"] = onchip
    fluidgym_amd.set_solver_policy(pressure_warm_start=bool(int(warm)), pressure_stall_accept=1.25 if int(warm) else 0.0,
                                   pressure_multilevel=bool(int(ml)))
    env = fluidgym_amd.make("CylinderJet2D-easy-v0", num_envs=B, initial_domain_steps=100, randomize_initial_state=False)
    env.reset(seed=0)
    gen = torch.Generator(device="cpu").manual_seed(7)
    act = lambda: (torch.rand(B, 1, generator=gen) * 2 - 1).cuda()
    env.step(act())
    dom = env._domain
    dom.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        _, _, _, _, info = env.step(act())
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    prof = dom.profile_read()
    print(json.dumps({"onchip": int(onchip), "warm_start": int(warm), "multilevel": int(ml), "iterations": dom.solver_counters(), "envs": B, "cells": dom.n_cells, "ms_per_env_step": round(1e3 * el, 2),
                      "env_steps_per_s": round(B / el, 1), "last_iterations": list(env._sim.last_iterations),
                      "substeps": env._sim.last_substeps, "drag0": round(float(info["drag"][0]), 4),
                      "profile": {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in prof.items()}}), flush=True)
    env.close()
