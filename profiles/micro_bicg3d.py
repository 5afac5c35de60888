"""BiCGStab on a TCF-sized synthetic advection-diffusion system (128 x 64 x 64 x 8 envs, walls in y): the z-marching two-kernel
form (fg_bicgstab3d.hip) against the five brick kernels, same box, same systems.  Usage: python profiles/micro_bicg3d.py [reps]
Environment: FG_BICG3 (0 = brick five-kernel path via FG_BICG_FUSED=0), read per handle."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidgym_amd.native import NativeSolver  # noqa: E402
from fluidgym_amd.simulation import grids  # noqa: E402


def build(B=8, n=(128, 64, 64)):
    nx, ny, nz = n
    hx = np.full(nx, 4 * np.pi / nx, np.float32)
    hy = (2.0 * np.diff(grids.weights_exp(ny, 1.05, "BOTH"))).astype(np.float32)
    hz = np.full(nz, 2 * np.pi / nz, np.float32)
    ns = NativeSolver([hx, hy, hz], B, fixed_faces=(2, 3))
    g = torch.Generator(device=ns.device).manual_seed(0)
    ns.set_viscosity(1.0 / 180.0)
    ns.velocity.normal_(0.0, 1.0, generator=g)
    ns.velocity[:, 0] += 15.0
    for f in (2, 3):
        ns.bvel[f].zero_()
    ns.copy_velocity_result_from_blocks()
    return ns


def run(label, env, reps, tol=1e-6, max_iterations=4):
    for k, v in env.items():
        os.environ[k] = v
    ns = build()
    ns.set_advection_start(False)
    ns.setup_advection(0.002)
    info = ns.solve_advection(tol=tol, max_iterations=max_iterations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        info = ns.solve_advection(tol=tol, max_iterations=max_iterations)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    x = ns.buffer(3, (8, 3, 64, 64, 128)).clone()
    its = [i.used_iterations for i in info]
    res = max(i.final_residual for i in info)
    ns.close()
    print(json.dumps({"variant": label, "ms_per_solve": round(1e3 * el, 4), "iterations": [min(its), max(its)], "max_residual": res}), flush=True)
    return x


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    xa = run("five brick kernels", {"FG_BICG_FUSED": "0", "FG_BICG3": "0"}, reps, tol=0.0, max_iterations=cap)
    xb = run("two brick kernels", {"FG_BICG_FUSED": "2", "FG_BICG3": "0"}, reps, tol=0.0, max_iterations=cap)
    os.environ.pop("FG_BICG3")
    xe = run("z-march two kernels + init kernel", {"FG_BICG_FUSED": "1", "FG_BICG3_MIX": "7"}, reps, tol=0.0, max_iterations=cap)
    os.environ.pop("FG_BICG3_MIX")
    xc = run("z-march two kernels", {"FG_BICG_FUSED": "1"}, reps, tol=0.0, max_iterations=cap)
    xd = run("z-march two kernels, 128x8 tiles", {"FG_BICG_FUSED": "1", "FG_BICG3_BXL": "32"}, reps, tol=0.0, max_iterations=cap)
    sc = float(xa.abs().max())
    print(json.dumps({"rel_diff_vs_five": {"brick2": float((xb - xa).abs().max()) / sc, "zmarch": float((xc - xa).abs().max()) / sc,
                                            "zmarch32": float((xd - xa).abs().max()) / sc}}))
