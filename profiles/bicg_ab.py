"""A/B of the two BiCGStab iteration forms of the single-block path (FG_BICG_FUSED = 0 five kernels | 2 two kernels), per
iteration, on the grid sizes of the bench legs.  Fixed iteration count (tolerance 0), time per iteration from HIP events.

    python profiles/bicg_ab.py  ->  one JSON line per (grid, batch, nc, form)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from fluidgym_amd.native import NativeSolver

CASES = [("headline 256x128 x64", (256, 128), 64), ("rbc 512x128 x32", (512, 128), 32), ("large 512x256 x64", (512, 256), 64),
         ("tcf 128x64x64 x8", (128, 64, 64), 8)]


def run(name, n, B, fused, for_scalar, iters=8, reps=8):
    os.environ["FG_BICG_FUSED"] = str(fused)
    dims = len(n)
    widths = [np.full(k, 1.0 / k, np.float32) for k in n]
    ns = NativeSolver(widths, B, fixed_faces=(2, 3), n_scalars=1)
    ns.set_viscosity(0.01)
    ns.set_scalar_viscosity(0, 0.01)
    g = torch.Generator(device=ns.device).manual_seed(0)
    ns.velocity.normal_(0.0, 0.3, generator=g)
    ns.scalar.uniform_(0.0, 1.0, generator=g)
    ns.copy_velocity_result_from_blocks()
    ns.set_advection_start(False)
    ns.setup_advection(0.02, for_scalar=for_scalar, channel=0)
    ns.solve_advection(for_scalar=for_scalar, tol=0.0, max_iterations=4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ns.solve_advection(for_scalar=for_scalar, tol=0.0, max_iterations=iters)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / (reps * iters)
    ns.close()
    nc = 1 if for_scalar else dims
    return {"case": name, "nc": nc, "form": {0: "five kernels", 1: "two kernels (2-D only)", 2: "two kernels"}[fused],
            "us_per_iteration": round(us, 1), "cells_x_systems": int(np.prod(n)) * B * nc}


if __name__ == "__main__":
    for name, n, B in CASES:
        for for_scalar in (True, False):
            for fused in (0, 2):
                print(json.dumps(run(name, n, B, fused, for_scalar)), flush=True)
