"""FD-preconditioner kernels (k_gemm_f32, k_tridiag_y) alone, all envs active, for a range of batch sizes.
    python3 profiles/micro_fd.py [nx ny]      -> one JSON line per batch size (live kernel-accurate timing)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fluidgym_amd.native import NativeSolver  # noqa: E402

if __name__ == "__main__":
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ny = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    os.environ["FG_PROF_PERIOD"] = "1"
    dev = torch.device("cuda", 0)
    for B in [int(v) for v in os.environ.get("FG_MICRO_B", "4,8,16,32,64,128").split(",")]:
        hx = np.full(nx, 1.0 / nx, np.float32)
        hy = np.full(ny, 1.0 / ny, np.float32)
        ns = NativeSolver([hx, hy], B, fixed_faces=(0, 1, 2, 3), device=dev, allocate=False)
        g = torch.Generator(device=dev).manual_seed(0)
        rA = 1.0 / (100.0 * (1.0 + 0.5 * torch.rand((B, ny, nx), device=dev, generator=g)))
        b = torch.randn((B, ny, nx), device=dev, generator=g)
        b -= b.mean(dim=(1, 2), keepdim=True)
        x = torch.zeros_like(b)
        ns.poisson_fdcg(rA, b, x, tol=0.0, max_iterations=3)
        ns.profile_enable(True)
        for _ in range(10):
            ns.poisson_fdcg(rA, b, x, tol=0.0, max_iterations=3)
        prof = ns.profile_read()
        ns.profile_enable(False)
        row = {"B": B}
        for k in ("k_gemm_f32", "k_tridiag_y", "k_cg_ap", "k_cg_update"):
            r = prof[k]
            if r["samples"]:
                row[k] = {"us": round(1e3 * r["ms"] / r["samples"], 2), "GBps": round(r["bytes"] / r["ms"] / 1e6),
                          "TFLOPps": round(r["flops"] / r["ms"] / 1e9, 1), "n": r["samples"]}
        print(json.dumps(row))
        ns.close()
