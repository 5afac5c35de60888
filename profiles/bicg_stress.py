"""Stress test of the multi-block velocity BiCGStab on the systems of the captured failures.

    python profiles/bicg_stress.py [seconds] [dump.npz ...]

Builds the Airfoil2D mesh for 16 envs, loads a dumped failing step (profiles/bicg_vec4_repro.py: diagonal, off-diagonals and the
two velocity right-hand sides of the failing env) into the assembly buffers of every env -- env 0 exactly, the others with 1e-4
relative noise on the right-hand side so that the batch covers a spread of trajectories around the captured one -- and solves
it again and again from zero through fg_mb_debug_bicgstab (the call fg_mb_piso_step makes).  One JSON line per dump: solves,
solves with a non-finite system, rate.  The access mode of the recurrence words is a build switch (profiles/bicg_stress.sh)."""
import ctypes
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidgym_amd import _lib as L  # noqa: E402
from fluidgym_amd.envs.airfoil_grid import make_airfoil_mesh  # noqa: E402
from fluidgym_amd.envs.cylinder_grid import build_domain  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dumps = sys.argv[2:] or sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "bicg_breakdown_*.npz")))
B = int(os.environ.get("STRESS_ENVS", "16"))   # 16 envs x 2 components = the velocity solve of the airfoil leg; 8 envs = its pressure-solve size
lib = L.load()
hip = ctypes.CDLL("libamdhip64.so")
dom = build_domain(make_airfoil_mesh(attack_angle_deg=10.0), 0.001, batch=B)
N, d = dom.n_cells, dom.dims


def upload(which, host):
    ptr, cnt = ctypes.c_void_p(), ctypes.c_int64()
    L.check(lib.fg_mb_get_buffer(dom.handle, which, ctypes.byref(ptr), ctypes.byref(cnt)))
    t = torch.from_numpy(np.ascontiguousarray(host, np.float32)).cuda()
    assert t.numel() == cnt.value, (which, t.numel(), cnt.value)
    rc = hip.hipMemcpy(ptr, ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(4 * t.numel()), 3)
    assert rc == 0, rc
    torch.cuda.synchronize()


for path in dumps:
    z = np.load(path)
    assert z["A"].shape == (N,), (path, z["A"].shape, N)
    rng = np.random.default_rng(0)
    rhs = np.repeat(z["rhs"][None], B, 0).astype(np.float64)
    rhs[1:] *= 1.0 + 1e-4 * rng.standard_normal(rhs[1:].shape)
    upload(L.FG_MB_BUF_A, np.repeat(z["A"][None], B, 0))
    upload(L.FG_MB_BUF_C_OFF, np.repeat(z["Coff"][None], B, 0))
    upload(L.FG_MB_BUF_RHS, rhs)
    out = (ctypes.c_int64 * 4)()
    tot = [0, 0, 0, 0]
    t0 = time.time()
    reps = 200
    while time.time() - t0 < seconds and tot[1] <= 20:
        L.check(lib.fg_mb_debug_bicgstab(dom.handle, 1e-6, 5000, reps, out, None, None, None))
        for k in range(3):
            tot[k] += out[k]
        tot[3] = max(tot[3], out[3])
    dt = time.time() - t0
    print(json.dumps({"dump": os.path.basename(path), "envs": B, "systems_per_solve": B * d, "solves": tot[0], "non_finite_solves": tot[1],
                      "unconverged_solves": tot[2], "max_iterations": tot[3], "seconds": round(dt, 1),
                      "solves_per_s": round(tot[0] / dt, 1), "non_finite_per_million_solves": round(1e6 * tot[1] / max(tot[0], 1), 1)}), flush=True)
dom.close()
