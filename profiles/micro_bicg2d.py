"""Fixed number of BiCGStab iterations on the headline grid (256 x 128 x 64 envs, channel boundary conditions) for kernel timing
under rocprofv3 and for knock-out builds (FLUIDGYM_AMD_LIB=<variant .so>).  python profiles/micro_bicg2d.py [reps] [iterations] [nx ny B]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidgym_amd.native import NativeSolver  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nx, ny, B = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (256, 128, 64)
hx = np.full(nx, 8.0 / nx, np.float32)
hy = np.full(ny, 2.0 / ny, np.float32)
ns = NativeSolver([hx, hy], B, fixed_faces=(0, 1, 2, 3))
g = torch.Generator(device=ns.device).manual_seed(0)
ns.set_viscosity(0.01)
ns.velocity.normal_(0.0, 0.3, generator=g)
ns.velocity[:, 0] += 1.0
for f in range(4):
    ns.bvel[f].zero_()
ns.bvel[0][:, 0] = 1.0
ns.bvel[1][:, 0] = 1.0
ns.copy_velocity_result_from_blocks()
ns.set_advection_start(False)
ns.setup_advection(0.01)
info = ns.solve_advection(tol=0.0, max_iterations=cap)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    info = ns.solve_advection(tol=0.0, max_iterations=cap)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / reps
print(json.dumps({"lib": os.environ.get("FLUIDGYM_AMD_LIB", "default"), "grid": [nx, ny, B], "ms_per_solve": round(1e3 * el, 4), "iterations": cap,
                  "us_per_iteration": round(1e6 * el / cap, 2), "residual": max(i.final_residual for i in info)}))
ns.close()
