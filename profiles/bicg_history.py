"""Scalar history of the multi-block velocity BiCGStab on a dumped failing step (tests/golden/bicg_breakdown_*.npz), DESIGN.md 4b.

    python profiles/bicg_history.py dump.npz [system] [iterations]

The solve is deterministic, so running it with max_iterations = 1, 2, ... and reading the recurrence words after each run
(fg_mb_debug_bicgstab) gives rho_k, the residual, alpha_(k-1), omega_(k-1), s.s, t.s, t.t of every iteration of the system."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidgym_amd import _lib as L  # noqa: E402
from fluidgym_amd.envs.airfoil_grid import make_airfoil_mesh  # noqa: E402
from fluidgym_amd.envs.cylinder_grid import build_domain  # noqa: E402

path = sys.argv[1]
system = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 24
B = 2
lib = L.load()
hip = ctypes.CDLL("libamdhip64.so")
dom = build_domain(make_airfoil_mesh(attack_angle_deg=10.0), 0.001, batch=B)
N, d = dom.n_cells, dom.dims
z = np.load(path)


def upload(which, host):
    ptr, cnt = ctypes.c_void_p(), ctypes.c_int64()
    L.check(lib.fg_mb_get_buffer(dom.handle, which, ctypes.byref(ptr), ctypes.byref(cnt)))
    t = torch.from_numpy(np.ascontiguousarray(host, np.float32)).cuda()
    assert t.numel() == cnt.value
    assert hip.hipMemcpy(ptr, ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(4 * t.numel()), 3) == 0
    torch.cuda.synchronize()


upload(L.FG_MB_BUF_A, np.repeat(z["A"][None], B, 0))
upload(L.FG_MB_BUF_C_OFF, np.repeat(z["Coff"][None], B, 0))
upload(L.FG_MB_BUF_RHS, np.repeat(z["rhs"][None], B, 0))
out = (ctypes.c_int64 * 4)()
acc = (ctypes.c_double * (12 * B * d))()
sc = (ctypes.c_float * (2 * B * d))()
print(json.dumps({"dump": os.path.basename(path), "system": system, "cells": N}))
for k in range(1, iters + 1):
    L.check(lib.fg_mb_debug_bicgstab(dom.handle, 1e-6, k, 1, out, acc, sc, None))
    a = np.array(acc[:]).reshape(B * d, 12)[system]
    al, om = sc[2 * system], sc[2 * system + 1]
    rho_prev_slot = (k - 1) & 1
    print(json.dumps({"after_iterations": k, "rho_k": a[k & 1], "rho_km1_slot": a[rho_prev_slot], "rms_r": float(np.sqrt(abs(a[6]) / N)) if np.isfinite(a[6]) else None,
                      "alpha": al, "omega": om, "s.s": a[3], "t.s": a[4], "t.t": a[5], "non_finite_solves": int(out[1])}), flush=True)
dom.close()
