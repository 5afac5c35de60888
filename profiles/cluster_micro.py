"""Micro-benchmark of the cluster pressure CG (k_mbc_cluster, csrc/fg_mb_cluster.hip) against the one-workgroup kernels on the
reference's cylinder meshes: fixed iteration count (tolerance 0), time per iteration per launch from the live profiler
(fg_mb_profile_*), and -- with a -DFG_CL_CYCLES build (FLUIDGYM_AMD_LIB) -- cycles per phase of workgroup 0 of env 0.
    python profiles/cluster_micro.py [envs=64] [iterations=200] [res=24] [configs: cluster:half:near, ...]"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidgym_amd import _lib as L  # noqa: E402
from fluidgym_amd.envs.cylinder_grid import build_domain, make_vortex_street_mesh  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
res = int(sys.argv[3]) if len(sys.argv) > 3 else 24
configs = (sys.argv[4] if len(sys.argv) > 4 else "0:1:1,1:1:1,1:0:1,1:1:0").split(",")
mesh = make_vortex_street_mesh(res)
for cfg in configs:
    cluster, half, near = cfg.split(":")
    os.environ["FG_MB_CLUSTER"], os.environ["FG_MB_CL_HALF"], os.environ["FG_MB_CL_NEAR"] = cluster, half, near
    dom = build_domain(mesh, 0.01, batch=B)
    dom.set_stall_limit(100000)
    dom.set_pressure_multilevel()
    g = torch.Generator(device="cpu").manual_seed(1)
    dom.velocity.copy_((0.3 * torch.randn(dom.velocity.shape, generator=g)).to(dom.device))
    dom.make_divergence_free(pressure_tol=1e-30, max_iterations=20, pressure_project_mean=True)   # warm-up
    dom.profile_enable(True)
    for _ in range(3):
        dom.make_divergence_free(pressure_tol=1e-30, max_iterations=iters, pressure_project_mean=True)
    torch.cuda.synchronize()
    p = dom.profile_read()
    k = p["k_mbc_onchip"]
    c = dom.config_dump()
    row = {"cluster": int(cluster), "cpt": c["cluster_members_per_thread"] if c["cluster_on"] else 0, "threads": c["cluster_threads"] if c["cluster_on"] else 0,
           "near": int(near), "half": int(half), "envs": B, "cells": dom.n_cells, "halo_max": c["cluster_halo_max"],
           "us_per_iteration": round(1e3 * k["ms"] / max(k["iterations"] / B, 1), 3), "launches": k["launches"],
           "cluster_solves": c["cluster_solves"], "fallbacks": c["cluster_fallbacks"]}
    if int(cluster) and os.environ.get("FLUIDGYM_AMD_LIB", "").endswith("cyc.so"):
        cyc = (ctypes.c_uint64 * 12)()
        L.check(dom.lib.fg_mb_debug_cycles(dom.handle, cyc))
        names = ["loop_top", "coarse_rows", "group_sums_barrier", "z_sums", "xb_barrier2", "boundary_stencil", "p_s_sums", "exchange_C", "update", "other_exchange"]
        names2 = ["xb_barrier1", "xb_publish", "xb_interior_stencil", "xb_polls", "verdict", "-"]
        its = max(int(cyc[11]), 1)
        row["cycles_per_iteration"] = {n: round((cyc[i] & 0xffffffff) / its) for i, n in enumerate(names)}
        row["cycles_per_iteration"].update({n: round((cyc[i] >> 32) / its) for i, n in enumerate(names2) if n != "-"})
        row["near_seen"] = int(cyc[10])
    print(json.dumps(row), flush=True)
    dom.close()
