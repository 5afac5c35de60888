"""Micro-benchmark of the multi-block pressure CG on the reference's cylinder mesh: fixed iteration count (tolerance 0), time per
iteration per launch from the live profiler (fg_mb_profile_*).  python profiles/onchip_micro.py [envs=64] [iterations=200] [res=24]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidgym_amd.envs.cylinder_grid import build_domain, make_vortex_street_mesh  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
res = int(sys.argv[3]) if len(sys.argv) > 3 else 24
mesh = make_vortex_street_mesh(res)
for onchip, variant in [("0", "0")] + [("1", v) for v in (sys.argv[4] if len(sys.argv) > 4 else "0,16,32,64").split(",")]:
    os.environ["FG_MB_ONCHIP"], os.environ["FG_MB_OC_VARIANT"] = onchip, variant
    dom = build_domain(mesh, 0.01, batch=B)
    dom.set_stall_limit(100000)
    if int(variant) & 128:
        dom.set_pressure_multilevel()
    g = torch.Generator(device="cpu").manual_seed(1)
    dom.velocity.copy_((0.3 * torch.randn(dom.velocity.shape, generator=g)).to(dom.device))
    dom.make_divergence_free(pressure_tol=1e-30, max_iterations=20, pressure_project_mean=True)   # warm-up
    dom.profile_enable(True)
    for _ in range(3):
        dom.make_divergence_free(pressure_tol=1e-30, max_iterations=iters, pressure_project_mean=True)
    torch.cuda.synchronize()
    p = dom.profile_read()
    row = {"onchip": int(onchip), "variant": int(variant), "envs": B, "cells": dom.n_cells}
    if int(onchip):
        k = p["k_mbc_onchip"]
        row.update(us_per_iteration=round(1e3 * k["ms"] / max(k["iterations"] / B, 1), 3), launches=k["launches"], iterations_per_env=k["iterations"] / B / max(k["launches"], 1),
                   streamed_GBps=round(k["bytes"] / k["ms"] / 1e6, 1))
    else:
        a, u = p["k_mbc_ap"], p["k_mbc_update"]
        row.update(us_per_iteration=round(1e3 * (a["ms"] / max(a["samples"], 1) + u["ms"] / max(u["samples"], 1)), 3),
                   ap_us=round(1e3 * a["ms"] / max(a["samples"], 1), 3), update_us=round(1e3 * u["ms"] / max(u["samples"], 1), 3))
    if int(variant) & 256:
        import ctypes
        from fluidgym_amd import _lib as L
        cyc = (ctypes.c_uint64 * 12)()
        L.check(dom.lib.fg_mb_debug_cycles(dom.handle, cyc))
        names = ["checks", "r_to_lds", "sum4", "sum8", "coarse", "table4", "z+rz", "p_update", "stencil", "pAp", "x_r_update"]
        its = max(int(cyc[11]), 1)
        row["cycles_per_iteration"] = {n: round(cyc[k] / its) for k, n in enumerate(names)}
    print(json.dumps(row), flush=True)
    dom.close()
