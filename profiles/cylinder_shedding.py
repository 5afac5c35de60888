"""Uncontrolled flow around the cylinder on the reference's mesh: drag / lift history and Strouhal number.

The cylinder envs' geometry (channel 22 x 4.1, cylinder diameter 1 offset by 0.05, parabolic inflow with mean 1, Re = 100)
is the Schaefer-Turek benchmark 2D-2 scaled by 10 (published: C_D,max 3.22-3.24, C_L,max 0.99-1.01, St 0.295-0.305),
so the wake that develops here checks the multi-block non-orthogonal path against numbers that do not come from this repo.

    python profiles/cylinder_shedding.py [resolution] [t_end] -> profiles/r01_cylinder_shedding_res<R>.csv
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import fluidgym_amd


def run(resolution=24, t_end=120.0, kick=True, non_ortho_mode="matrix"):
    env = fluidgym_amd.make("CylinderRot2D-easy-v0", num_envs=1, resolution=resolution, initial_domain_steps=0,
                            randomize_initial_state=False, non_ortho_mode=non_ortho_mode)
    env.reset(seed=0)
    dt = env.dt
    n = int(round(t_end / dt))
    hist = np.zeros((n, 3), np.float32)
    t0 = time.time()
    for k in range(n):
        if kick and 100 <= k < 200:  # a short spin breaks the symmetry so that shedding starts early
            env._apply_action(torch.full((1, 1), 0.5, device="cuda"))
        elif kick and k == 200:
            env._apply_action(torch.zeros(1, 1, device="cuda"))
        env._sim.single_step()
        cd, cl = env._get_drag_and_lift()
        hist[k] = ((k + 1) * dt, float(cd[0]), float(cl[0]))
    wall = time.time() - t0
    env.close()
    return hist, wall


def analyse(hist, t_from):
    t, cd, cl = hist[:, 0], hist[:, 1], hist[:, 2]
    m = t >= t_from
    t, cd, cl = t[m], cd[m], cl[m]
    c = cl - cl.mean()
    up = np.nonzero((c[:-1] < 0) & (c[1:] >= 0))[0]
    tz = t[up] + (t[up + 1] - t[up]) * (-c[up]) / (c[up + 1] - c[up])
    period = float(np.diff(tz).mean()) if len(tz) > 2 else float("nan")
    return {"cd_max": float(cd.max()), "cd_mean": float(cd.mean()), "cl_max": float(cl.max()), "cl_min": float(cl.min()),
            "strouhal": 1.0 / period if period == period else float("nan"), "periods": max(len(tz) - 1, 0)}


if __name__ == "__main__":
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    t_end = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    mode = sys.argv[3] if len(sys.argv) > 3 else "matrix"
    hist, wall = run(res, t_end, non_ortho_mode=mode)
    out = os.path.join(ROOT, "gpurun_out", f"r01_cylinder_shedding_res{res}" + ("" if mode == "matrix" else f"_{mode}") + ".csv")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savetxt(out, hist[::5], delimiter=",", header="t,cd,cl", comments="", fmt="%.5f")
    print("mode", mode, "resolution", res, "sim steps", len(hist), "wall s", round(wall, 1), analyse(hist, t_end - 30.0))
