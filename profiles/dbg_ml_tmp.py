import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers_mb as H
from tests.test_gpu_mb import _state, _load
for fn in (H.polar_ring, H.split_rotated_channel):
    for ml in (0, 1):
        spec = fn(); d = spec.oracle(); B = 2
        dom = spec.native(batch=B)
        if ml: print("tables", dom.set_pressure_multilevel())
        states = [_state(d, 10 + b) for b in range(B)]
        _load(dom, states)
        print("====", fn.__name__, "ml", ml, flush=True)
        its = dom.piso_step([0.05, 0.03], advection_tol=1e-7, pressure_tol=2e-6, pressure_use_bicgstab=1, pressure_project_mean=True, raise_on_failure=False)
        print("its", its, "ladder", dom.ladder(), "counters", dom.solver_counters(), flush=True)
        dom.close()
