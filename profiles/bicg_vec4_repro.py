"""Repro harness for the intermittent non-finite BiCGStab solve of round 1 (fg_mb_step.hip, four-cells-per-thread kernels).

    python profiles/bicg_vec4_repro.py [reps] [envs] [develop_steps] [masks, comma separated: 0 = one-cell kernels, 31 = four-cell in every solve]

For each kernel mask: `reps` fresh Airfoil2D-easy-v0 batches are developed from the impulsive start (the phase the defect was
seen in).  After every sim step the per-env solver status is read (fg_mb_env_status); the first non-finite solve of a run dumps
the assembled system of the failing env (A, C off-diagonals, RHS, neighbour table) to gpurun_out/bicg_fail_<mask>_<rep>.npz so
that the recurrence can be replayed offline.  One JSON line per run."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = int(sys.argv[3]) if len(sys.argv) > 3 else 60
masks = [int(m) for m in (sys.argv[4] if len(sys.argv) > 4 else "0,31").split(",")]
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)

import fluidgym_amd  # noqa: E402
from fluidgym_amd import _lib as L  # noqa: E402

summary = {}
for mask in masks:
    os.environ["FG_MB_BICG_VEC4"] = str(mask)     # read once per handle at fg_mb_create
    os.environ["FG_MB_TRACE_FAIL"] = "1"          # recurrence scalars of a system that breaks down, on stderr
    fails = 0
    for rep in range(reps):
        env = fluidgym_amd.make("Airfoil2D-easy-v0", num_envs=B, initial_domain_steps=0, randomize_initial_state=False)
        env.reset(seed=rep)
        dom, sim = env._domain, env._sim
        sim.solver_double_fallback = sim.BiCG_precondition_fallback = False   # the retry ladder would hide the failure being hunted
        t0 = time.time()
        first_fail, n_fail_steps, worst_it = None, 0, 0
        for step in range(dev):
            ok = sim.single_step()
            worst_it = max(worst_it, max(sim.last_iterations))
            if not ok:
                n_fail_steps += 1
                if first_fail is None:
                    st = sim.last_env_status
                    bad = int(np.nonzero(st == 2)[0][0])
                    msg = L.load().fg_last_error()
                    first_fail = {"step": step, "envs": np.nonzero(st == 2)[0].tolist(), "iterations": list(sim.last_iterations),
                                  "substeps": sim.last_substeps, "last_error": msg.decode() if msg else ""}
                    N, d = dom.n_cells, dom.dims
                    np.savez_compressed(
                        os.path.join(out_dir, f"bicg_fail_{mask}_{rep}.npz"),
                        A=dom.buffer(L.FG_MB_BUF_A).view(B, N)[bad].cpu().numpy(),
                        Coff=dom.buffer(L.FG_MB_BUF_C_OFF).view(B, 2 * d, N)[bad].cpu().numpy(),
                        rhs=dom.buffer(L.FG_MB_BUF_RHS).view(B, d, N)[bad].cpu().numpy(),
                        x0=dom.velocity[bad].cpu().numpy(), nbr=dom.neighbors(), env=bad, step=step,
                        # Krylov vectors of the failing env's systems as the solve left them (r, rw, p, v, t): the masked env is
                        # skipped by every later kernel of the step, so they are exactly what the failing iteration produced
                        krylov=np.stack([dom.buffer(L.FG_MB_BUF_KRYLOV0 + k).view(B, d, N)[bad].cpu().numpy() for k in range(5)]))
                    kr = np.stack([dom.buffer(L.FG_MB_BUF_KRYLOV0 + k).view(B, d, N)[bad].cpu().numpy() for k in range(5)])
                    bad_cells = {nm: np.nonzero(~np.isfinite(kr[k]).all(0))[0] for k, nm in enumerate(("r", "rw", "p", "v", "t"))}
                    first_fail["nonfinite_cells"] = {nm: {"count": int(len(c)), "first": c[:8].tolist(), "last": c[-4:].tolist()} for nm, c in bad_cells.items()}
        torch.cuda.synchronize()
        finite = bool(torch.isfinite(dom.velocity).all())
        fails += first_fail is not None
        print(json.dumps({"mask": mask, "rep": rep, "seconds": round(time.time() - t0, 2), "failed_steps": n_fail_steps,
                          "first_fail": first_fail, "max_iterations": worst_it, "state_finite": finite}), flush=True)
        env.close()
    summary[mask] = {"runs": reps, "runs_with_non_finite_solve": fails}
print(json.dumps({"summary": summary, "envs": B, "develop_steps": dev}), flush=True)
