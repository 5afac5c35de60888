"""CPU baseline for bench.py: the oracle (test infrastructure, a NumPy/SciPy restatement of the reference algorithm) stepping the
bench workload on host cores.  The reference has no CPU path, so this is a "port" baseline.  One env per process -- envs are
independent, which is how a CPU would batch them too.  Imported only by bench.py's ``cpu_baseline`` leg."""
import os
import time


def _run_one(args):
    budget_s, dtype_name, seed = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    import numpy as np

    from fluidgym_amd.envs.channel import CHANNEL_JET_2D_DEFAULT_CONFIG as CFG, inflow_profile
    from oracle import piso_oracle as O

    dt = np.dtype(dtype_name)
    nx, ny, L, H = CFG["resolution_x"], CFG["resolution_y"], 22.0, 4.1
    g = O.Grid(O.rectilinear_coords([np.linspace(0, L, nx + 1), np.linspace(-H / 2, H / 2, ny + 1)], dtype=dt))
    prof = inflow_profile(H, ny).astype(dt)
    u = np.zeros((2, ny, nx), dt)
    u[0] = prof[:, None]
    rng = np.random.default_rng(seed)
    u += (0.05 * rng.standard_normal(u.shape)).astype(dt)
    inflow = np.zeros((2, ny, 1), dt)
    inflow[0, :, 0] = prof
    bc = {0: O.FixedBC(inflow.copy()), 1: O.FixedBC(inflow.copy()), 2: O.FixedBC(np.zeros(2, dt)), 3: O.FixedBC(np.zeros(2, dt))}
    dom = O.Domain(g, dt.type(1.0 / CFG["reynolds_number"]), u, np.zeros((ny, nx), dt), bc)
    # same start vectors as the GPU leg: the channel env runs the reference's non-orthogonal branch (velocity solve from zero)
    opts = O.SolverOptions(direct=False, pressure_tol=1e-5, advection_tol=1e-5, pressure_return_best_result=True, non_orthogonal=True)
    O.make_divergence_free(dom, O.SolverOptions(direct=False, pressure_tol=1e-5))
    velm = np.array([1.0, 0.0], dt)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s:
        O.update_advective_boundaries(dom, [1], velm, CFG["dt"], tol=1e-5)
        O.piso_split_step(dom, CFG["dt"], opts)
        steps += 1
    return steps, time.perf_counter() - t0, str(dom.velocity.dtype)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run(budget_1=8.0, budget_all=12.0, dtype="float32"):
    """(1-thread PISO steps/s, all-cores PISO steps/s, cores, dtype carried by the fields, PISO steps done by all cores).
    Workers are plain child processes of this interpreter (``python -m oracle.cpu_bench <budget> <dtype> <seed>``)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1",
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))

    def launch(budget, seed):
        return subprocess.Popen([sys.executable, "-m", "oracle.cpu_bench", str(budget), dtype, str(seed)], cwd=root, env=env,
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)

    def collect(p):
        out, _ = p.communicate(timeout=600)
        return json.loads(out.strip().splitlines()[-1])

    one = collect(launch(budget_1, 0))
    procs = [launch(budget_all, k) for k in range(cores)]
    res = [collect(p) for p in procs]
    rate_all = sum(r["steps"] / r["seconds"] for r in res)
    return one["steps"] / one["seconds"], rate_all, cores, one["dtype"], sum(r["steps"] for r in res)


if __name__ == "__main__":
    import json
    import sys

    steps, seconds, dt = _run_one((float(sys.argv[1]), sys.argv[2], int(sys.argv[3])))
    print(json.dumps({"steps": steps, "seconds": seconds, "dtype": dt}))
