"""Generates docs/SWITCHES.md: every environment switch of the package, FROM THE SOURCE (getenv / os.environ sites), with where it is
read, its default, and whether it can change results.  The "results" and "what" columns are annotations kept here; a switch found in
the source without an annotation (or annotated but gone from the source) fails the run, so the table cannot rot:
    python profiles/gen_switch_table.py            # rewrite docs/SWITCHES.md
    python profiles/gen_switch_table.py --check    # exit 1 if the committed table is stale (tests/test_switch_table.py)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name -> (changes results?, what it does).  "bits" = same converged answer, different rounding / iteration trajectory;
# "no" = timing / diagnosis only; "policy" = a documented departure from (or return to) the reference's solver policy
NOTES = {
    "FG_CG_FUSED": ("bits", "1 (default): three-launch preconditioned pressure CG on 2-D fast-transform grids (fg_fftcg.hip); 0: the five kernels"),
    "FG_BICG_PFUSED": ("bits", "1 (default): six-launch Helmholtz-preconditioned BiCGStab (fg_fftbicg.hip); 0: eleven launches per iteration"),
    "FG_BICG_FUSED": ("bits", "1 (default): two-kernel BiCGStab iteration in 2-D; 0: five kernels; 2: two kernels in 3-D bricks too"),
    "FG_BICG_SUB": ("no", "envs per sub-batch of the 2-D two-kernel BiCGStab (unset: by working set; 0: never)"),
    "FG_BICG3": ("bits", "z-marching 3-D BiCGStab: 0 never, N always with z-chunk N (tests), unset by the rule"),
    "FG_BICG3_BXL": ("no", "tile lanes along x of the z-marching BiCGStab (16 / 32)"),
    "FG_BICG3_MIX": ("bits", "debugging: bit 0 kernel a, bit 1 kernel b as z-march, bit 2 keep the init kernel"),
    "FG_REDUCE_WGS": ("no", "workgroups per env of the reduction kernels (0 = rule)"),
    "FG_HELM_ROWFORM": ("bits", "0: Helmholtz factors through k_helm_coeffs + the array-form line kernels (IEEE division instead of v_rcp)"),
    "FG_ADV_LINESWEEP": ("bits", "1: the advection-diffusion systems of the Helmholtz-preconditioned family (wall-refined 2-D grids: RBC) go to line sweeps x <- (D + O_y)^-1 (b - O_x x) first (k_line_sweep_y, fg_linepre.hip), BiCGStab takes what they do not settle; 0 (default): the preconditioned BiCGStab always -- measured on the RBC leg: 16-17 sweeps per velocity solve at the bench state's sub-step, 775 against 793 env-steps/s"),
    "FG_FD_FACFUSE": ("no", "1 (default): the first tridiagonal solve after 1/A changed makes the per-env row-mean factors itself (k_tridiag_y_lds<.., FAC>: same arithmetic, same bits); 0: k_fd_rowmean_factor as a launch of its own (A/B runs)"),
    "FG_FD_ROWMEAN": ("bits", "1 (default): the fused pressure CG is preconditioned by the row-mean operator (per-env factors, one factorisation per PISO step); 0: the grid's A = 1 factors"),
    "FG_ADV_JACOBI": ("bits", "velocity systems of uniform 2-D grids: 1 Jacobi sweeps first (fg_jacobi.hip), 0 BiCGStab always; unset = what fg_set_advection_jacobi says (the Simulation turns it on: policy advection_jacobi)"),
    "FLUIDGYM_AMD_ADVECTION_JACOBI": ("policy", "policy advection_jacobi (default 1): the Simulation asks for the Jacobi sweeps of fg_jacobi.hip on the grids that qualify"),
    "FLUIDGYM_AMD_PRESSURE_REFINEMENT": ("bits", "policy pressure_refinement (default 0 = off): up to N mixed-precision corrections behind every single-block pressure solve (fp64 residual, fp32 corrections: fg_set_pressure_refinement) -- an accuracy mode"),
    "FLUIDGYM_AMD_MULTI_STEP": ("no", "0: the sim steps of an env step are issued one by one from Python instead of through fg_multi_step (A/B runs; same bits)"),
    "FG_FCG_FIRST": ("bits", "1 (default): the fused CG judges its first iterate from I'(0)'s dot products (k_fcg_check0) and stores no pressure when every env ends there; 0: the verdict on the updated residual (A/B runs, tests)"),
    "FG_JAC_WARM": ("bits", "start vector of the Jacobi sweeps of a step's velocity systems: 1 the block velocity, 0 the BiCGStab start vector (zero on the non-orthogonal branch); unset: the block velocity on on-chip grids of 2^17 cells and more (where it saves a pass)"),
    "FG_JAC_PREFACTOR": ("no", "0: the row-mean factors of the pressure preconditioner are made in front of the first tridiagonal solve instead of behind the sweeps' check kernel (A/B runs)"),
    "FG_FCG_SPEC": ("no", "0: the corrector waits for the verdict on the first iterate instead of being launched behind the verdict kernel (A/B runs)"),
    "FG_JAC_SPEC": ("no", "0: k_h and the divergence kernel of the first corrector wait for the verdict on the velocity sweeps instead of being launched behind the check kernel (A/B runs)"),
    "FG_JAC_XCD": ("no", "0: the on-chip Jacobi regions in launch order instead of one XCD per run of an env's regions (A/B runs)"),
    "FG_JAC_SHAPE": ("bits", "A/B runs of the Jacobi sweeps: 1 full-row regions only, 2 bands only (unset: the cheaper of the two for the grid)"),
    "FG_JAC_SWEEPS": ("bits", "A/B runs: sweeps per pass of the Jacobi sweeps (unset: 4 / 6 / 8 for full rows, 8 / 12 for bands, by region depth)"),
    "FG_ROCTX": ("no", "1: named roctx ranges around the phases of a step (fg_single_step, scalar / velocity assembly and solve, pressure correctors; mb_velocity_solve, mb_pressure_solve) for rocprofv3 --marker-trace; the library is dlopen'ed, off by default"),
    "FG_HTRACE": ("no", "1: host time stamps around the polls and the launches behind them (fg_poll.hip), per-pair averages printed at exit"),
    "FG_FORCE_ZMARCH": ("bits", "z-march chunk length of the 3-D Poisson kernels (tests on small grids; -1 = brick kernels)"),
    "FG_ZMARCH_SB": ("no", "single-barrier ring of the z-march kernels: bit per mode, 0 never, unset = rule"),
    "FG_POLL_SPIN": ("no", "0: host polls wait with hipStreamSynchronize instead of spinning on pinned sequence words"),
    "FG_POLL_WORDS": ("no", "0: polled verdicts through the host-pinned mirror + a system-scope release (one L2 write-back per verdict kernel) instead of travelling in the 8-byte result words the host spins on (A/B runs: 8 790 against 9 600 env-steps/s on the headline, one lane)"),
    "FG_DEV_DT": ("no", "0: fg_single_step waits for the CFL maxima and computes the adaptive sub-steps on the host before the PISO step, instead of the CFL kernel's last workgroup taking them on the device (FgDtRule: same doubles, same bits) and the host reading the maxima after the step (A/B runs)"),
    "FLUIDGYM_AMD_F64_FD": ("bits", "0: the fp64 library solves the pressure systems with the reference's plain CG instead of the CG preconditioned by the fast-diagonalisation operator in doubles (csrc/fg_f64_fd.hip; A/B runs: iterations per solve)"),
    "FLUIDGYM_AMD_ENV_GLUE": ("bits", "0: ChannelJet2D's action schedule and reward / observation as elementwise torch launches instead of the two native kernels of fg_envglue.hip (A/B runs; the schedule is bit-identical, the means differ by fp32 rounding of the summation order)"),
    "FLUIDGYM_AMD_LANES": ("no", "default of ParallelFluidEnv(lanes=...): sub-batches of a rank's env batch, each its own solver handle, stepped concurrently by that many host threads on their own HIP streams (1 = one batch; bench.py passes --lanes 2).  Same bits per env as the sub-batches stepped alone"),
    "FG_PROF_PERIOD": ("no", "sampling period of the live kernel timing (64)"),
    "FG_FD_NO_FFT": ("bits", "1: the x basis change of the FD preconditioner as dense GEMM instead of the row FFT (tests)"),
    "FG_MB_BICG_VEC4": ("no", "bit per multi-block BiCGStab kernel: four-cell form (31 = all)"),
    "FG_MB_BICG_FUSE": ("bits", "multi-block BiCGStab: 0 five kernels, 1 s/t fused, 2 (default) also p/v"),
    "FG_MB_ML_FUSE": ("bits", "multilevel BiCGStab forms p / s inside the restriction: 0 never, 1 up to 32 systems, 2 always"),
    "FG_MB_ML_TRY_CAP": ("policy", "iteration cap of an attempt of the multilevel trial (200; tests force 2)"),
    "FG_MB_TRACE": ("no", "residual / verification trace of the pressure BiCGStab on stderr"),
    "FG_MB_TRACE_FAIL": ("no", "recurrence words of a system that ends non-finite"),
    "FG_MB_COMPACT": ("no", "0: every multi-block Krylov launch covers all systems of the batch"),
    "FG_MB_OC_RTG_NT": ("bits", "register-resident on-chip CG with a global residual copy on 16-24 k-cell meshes instead of k_mbc_l2"),
    "FG_MB_ONCHIP": ("bits", "0: no whole-solve on-chip CG (chunked CG kernels)"),
    "FG_MB_OC_AGG": ("bits", "0: cell-ordered on-chip CG instead of the aggregate-owned one"),
    "FG_MB_CLUSTER": ("bits", "pressure CG of an env by a cluster of four workgroups (fg_mb_cluster.hip): 0 never (the one-workgroup kernels), 1 (default) meshes beyond 8 k cells, 2 every mesh its tables fit (tests)"),
    "FG_MB_CL_JACOBI": ("bits", "0: the velocity sweeps of the meshes the cluster CG takes stay one launch per sweep (k_mbj_sweep_env) instead of one launch per solve by the clusters (k_mbj_cluster; A/B runs: 2 552 against 3 045 env-steps/s on the cylinder mesh)"),
    "FG_MB_CL_HALF": ("bits", "0: meshes whose rows of the coarse inverse do not fit LDS as fp32 stream them from L2 instead of holding them as fp16 (A/B runs: 19.7 against 16.8 us per iteration on 23 k cells)"),
    "FG_MB_CL_MAXCL": ("no", "cap on the clusters of one launch of the cluster CG (tests: envs beyond it queue inside the kernel; same bits)"),
    "FG_MB_CL_NEAR": ("no", "0: granule stores of the cluster CG always write through (sc1), also when a cluster's workgroups reported one XCD (A/B runs: 10.0 against 8.6 us per iteration)"),
    "FG_MB_OC_VARIANT": ("no", "on-chip CG variant bits (fences / coefficient layout)"),
    "FG_MB_RUNG_ILU": ("bits", "0: column-scaled preconditioned rung instead of ILU(0)"),
    "FLUIDGYM_AMD_LIB": ("no", "path of libfluidgym_hip.so (sanitizer build: tests/run_sanitizer_suite.sh)"),
    "FLUIDGYM_AMD_LIB_F64": ("no", "path of libfluidgym_hip_f64.so"),
    "FLUIDGYM_AMD_ADVECTION_FD_PRECONDITIONER": ("bits", "policy advection_fd_preconditioner: auto (2-D periodic-x grids) / always / never"),
    "FLUIDGYM_AMD_ADVECTION_LINE_PRECONDITIONER": ("bits", "policy advection_line_preconditioner (off)"),
    "FLUIDGYM_AMD_ADVECTION_RUNG_PRECONDITIONER": ("bits", "preconditioner of the BiCG_precondition_fallback rung: line / ilu"),
    "FLUIDGYM_AMD_ADVECTION_WARM_START": ("policy", "1: velocity solve starts from the current velocity in both branches (reference: branch rule)"),
    "FLUIDGYM_AMD_PRESSURE_WARM_START": ("policy", "1: pressure solves start from the previous pressure (reference: zero)"),
    "FLUIDGYM_AMD_PRESSURE_STALL_ACCEPT": ("policy", "accept a stalled pressure solve within a factor of the tolerance (off)"),
    "FLUIDGYM_AMD_PRESSURE_MULTILEVEL": ("bits", "multilevel preconditioner of the multi-block pressure CG"),
    "FLUIDGYM_AMD_PRESSURE_MULTILEVEL_BICGSTAB": ("bits", "multilevel trial of the airfoil's refined BiCGStab"),
    "FLUIDGYM_AMD_PRESSURE_BICGSTAB_LARGE_MESHES": ("bits", "3-D multi-block ids take the refined BiCGStab"),
    "FLUIDGYM_AMD_NATIVE_WALL_FORCING": ("bits", "policy native_wall_forcing: the TCF forcing hook runs inside the library"),
    "FLUIDGYM_COLLECTIVE_TIMEOUT_S": ("no", "timeout of ParallelFluidEnv's process group (600 s)"),
    "FLUIDGYM_FORCE_COLLECTIVES": ("no", "ParallelFluidEnv issues its collectives at world size 1 too (tests)"),
    "FLUIDGYM_MASTER_PORT": ("no", "rendezvous port of the reference-style ParallelFluidEnv(cuda_ids=...) entry"),
    "FLUIDGYM_DATA_PATH": ("no", "local data path (initial domains), as config.update('local_data_path', ...)"),
}
IGNORE = {"RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY", "PYTHONPATH", "OMP_NUM_THREADS",
          "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "TMPDIR", "GRAFT_REPO_ROOT"}


def scan():
    found = {}
    for base, _, files in os.walk(os.path.join(ROOT, "fluidgym_amd")):
        if "_san" in base or "_f64" in base or "_var" in base or "__pycache__" in base:
            continue
        for fn in files:
            if not fn.endswith((".hip", ".h", ".py")):
                continue
            path = os.path.join(base, fn)
            rel = os.path.relpath(path, ROOT)
            for no, line in enumerate(open(path, errors="replace"), 1):
                for m in re.finditer(r'getenv\("([A-Z][A-Z0-9_]+)"\)', line):
                    found.setdefault(m.group(1), []).append(f"{rel}:{no}")
                for m in re.finditer(r'environ(?:\.get)?[\[(]\s*"([A-Z][A-Z0-9_]+)"', line):
                    found.setdefault(m.group(1), []).append(f"{rel}:{no}")
    return {k: v for k, v in found.items() if k not in IGNORE}


def render(found):
    lines = ["# Environment switches of fluidgym_amd", "",
             "Generated by `python profiles/gen_switch_table.py` from the `getenv` / `os.environ` sites of the source; `tests/test_switch_table.py`",
             "fails when this file is stale.  The native library reads its variables ONCE, at `fg_create` / `fg_mb_create` (never on a step",
             "path); `fg_config_dump` / `fg_mb_config_dump` report the values a handle runs under and `bench.py` stores them with every leg",
             "(`profiles/bench_detail.json`).  None is needed in normal operation.", "",
             "*results*: **no** = timing / diagnosis only; **bits** = the same converged answer through another kernel form (rounding and",
             "iteration trajectories differ); **policy** = a documented departure from (or return to) the reference's solver policy.", "",
             "| variable | results | read at | what |", "|---|---|---|---|"]
    for name in sorted(found):
        res, what = NOTES[name]
        where = ", ".join(f"`{w}`" for w in found[name][:2]) + (" ..." if len(found[name]) > 2 else "")
        lines.append(f"| `{name}` | {res} | {where} | {what} |")
    lines += ["", "Build-time defines (profiling builds only): `-DFG_ACC_ACCESS`, `-DFG_FLAG_ACCESS`, `-DFG_MB_OC_CYCLES`, `-DFG_KNOCK_B`, `-DFG_WAVE_SHFL`,",
              "`-DFG_DCT_KNOCK` -- see the comments where they are tested.", ""]
    return "\n".join(lines)


def main():
    found = scan()
    missing = sorted(set(found) - set(NOTES))
    gone = sorted(set(NOTES) - set(found))
    if missing or gone:
        print(f"switch table out of date: in the source without a note {missing}; noted but not in the source {gone}")
        return 1
    text = render(found)
    path = os.path.join(ROOT, "docs", "SWITCHES.md")
    if "--check" in sys.argv:
        # line numbers move with every edit: compare names and annotations only
        cur = open(path).read() if os.path.exists(path) else ""
        strip = lambda t: re.sub(r"\| `[^|]*:\d+`[^|]*\|", "| |", t)
        return 0 if strip(cur) == strip(text) else 1
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write(text)
    print(f"wrote {path}: {len(found)} switches")
    return 0


if __name__ == "__main__":
    sys.exit(main())
