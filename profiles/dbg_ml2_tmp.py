import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers_mb as H
from tests.test_gpu_mb import _state, _load
which = sys.argv[1]
if which == "skewed":
    spec = H.skewed_pair(); d = spec.oracle(); B = 2
    dom = spec.native(batch=B)
    _load(dom, [_state(d, 10 + b) for b in range(B)])
    its = dom.piso_step([0.05, 0.03], advection_tol=1e-7, pressure_tol=2e-7, pressure_use_bicgstab=2, pressure_project_mean=False, raise_on_failure=False)
    print("its", its, "status", dom.env_status(), dom.solver_counters())
elif which == "eig":
    from fluidgym_amd.envs.airfoil_grid import make_airfoil_mesh
    from fluidgym_amd.envs.cylinder_grid import build_domain
    from fluidgym_amd.simulation.multiblock import multilevel_tables
    import scipy.sparse as sp
    for div in (2, 1):
        dom = build_domain(make_airfoil_mesh(attack_angle_deg=10.0, resolution_div=div), 0.001, batch=1)
        P = dom.unit_pressure_matrix().astype(np.float64)
        S = (0.5 * (P + P.T)).tocsr(); N = S.shape[0]
        tab = multilevel_tables(P, [(b.size[0], b.size[1], b.cell_offset) for b in dom.blocks], max_n4=65534, max_n8=2048)
        Z4 = sp.csr_matrix((np.ones(N), (np.arange(N), tab["a4"])), shape=(N, tab["n4"]))
        Z8 = sp.csr_matrix((np.ones(tab["n4"]), (np.arange(tab["n4"]), tab["parent4"])), shape=(tab["n4"], tab["n8"]))
        A8 = (Z8.T @ (Z4.T @ S @ Z4) @ Z8).toarray()
        ev = np.linalg.eigvalsh(A8)
        print("div", div, "N", N, "n8", tab["n8"], "eig A8 (largest magnitude)", ev[0], "five closest to zero", ev[-5:], "row sums of S: max |S 1|", np.abs(S @ np.ones(N)).max(), "max |diag|", np.abs(S.diagonal()).max(),
              "asym", abs(P - P.T).sum() / abs(P).sum())
        dom.close()
else:
    import fluidgym_amd
    KW = dict(randomize_initial_state=False, episode_length=3)
    env = fluidgym_amd.make("Airfoil2D-easy-v0", num_envs=2, **dict(KW, initial_domain_steps=12))
    env.reset(seed=3)
    obs, reward, _, _, info = env.step(torch.zeros(2, 3, device="cuda"))
    print("its", env._sim.last_iterations, "drag", info["drag"], "lift", info["lift"], "status", env._domain.env_status(), "ladder", env._domain.ladder())
