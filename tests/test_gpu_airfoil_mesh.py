"""The multi-block path on the reference's AIRFOIL mesh (six blocks, NACA 0012 at 20 deg, ~45 k cells), built from the
coordinates and construction calls recorded from the reference's make_airfoil_domain (tests/golden/make_golden_airfoil.py).
The airfoil env itself is not built yet; this checks that the topology (two blocks wrapped around the section, the
tail blocks joined along a shuffled face, two outflow faces) and the metrics of that mesh run through the solver."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_airfoil_grid.npz"))
FACE = {"-x": 0, "+x": 1, "-y": 2, "+y": 3}


def _build(tag="aoa20", batch=2, nu=0.3 / 1e3):
    from fluidgym_amd.simulation.multiblock import MultiBlockDomain

    calls = [str(c).split() for c in G[f"{tag}_calls"]]
    dom = MultiBlockDomain(2, nu, batch=batch)
    blocks = {}
    for c in calls:
        if c[0] == "block":
            blocks[int(c[1])] = dom.CreateBlock(G[f"{tag}_block{c[1]}"], name=c[2])
    for c in calls:
        if c[0] == "velocity":
            b, face = int(c[1]), c[2]
            v = G[f"{tag}_velocity_{b}_{face}"]
            v = v.reshape(2, -1) if v.size > 2 else v.reshape(2, 1)
            blocks[b].CloseBoundary(face, v)
        elif c[0] == "connect":
            blocks[int(c[1])].ConnectBlock(c[2], blocks[int(c[3])], c[4], c[5])
    dom.PrepareSolve()
    outflow = [(int(x[:-2]), x[-2:]) for x in next(c for c in calls if c[0] == "balance")[1:]]
    return dom, blocks, outflow


def test_topology_and_metrics_of_the_recorded_mesh():
    dom, blocks, outflow = _build(batch=1)
    assert dom.n_cells == sum(b.n_cells for b in dom.blocks) == 45_244 or dom.n_cells > 40_000
    T = dom.cell_transforms()
    assert T[:, -1].min() > 0  # right-handed cells everywhere
    nbr = dom.neighbors()
    # every connection is mutual: the neighbour of my neighbour across the connecting faces is me
    N = dom.n_cells
    for f in range(4):
        ok = nbr[f] >= 0
        back = np.zeros(N, bool)
        for g in range(4):
            back[ok] |= nbr[g][nbr[f][ok]] == np.nonzero(ok)[0]
        assert back[ok].all()
    assert outflow == [(4, "+x"), (5, "+x")]
    dom.close()


def test_uniform_inflow_develops_stably_with_two_outflow_faces():
    dom, blocks, outflow = _build(batch=2)
    dom.velocity[:, 0] = 0.3
    ok = dom.make_divergence_free(outflow=outflow, outflow_velocity=(0.3, 0.0, 0.0))
    assert np.abs(dom.boundary_flux_balance()).max() < 1e-5       # balance_boundary_fluxes over both tail faces
    for _ in range(10):
        n, conv, its = dom.single_step(0.004, cfl=0.8, outflow=outflow, outflow_velocity=(0.3, 0.0, 0.0),
                                       advect_non_ortho_steps=2, pressure_non_ortho_steps=4, advection_tol=1e-6,
                                       pressure_tol=1e-6, pressure_project_mean=True, pressure_warm_start=True,
                                       pressure_stall_accept=1.25)
    assert torch.isfinite(dom.velocity).all() and torch.isfinite(dom.pressure).all()
    assert np.abs(dom.boundary_flux_balance()).max() < 1e-5
    mv = dom.max_velocity()
    assert 20.0 < mv.max() < 5000.0  # |Minv u|: 0.3 / cell size, no blow-up
    # the flow accelerates over the suction side: somewhere faster than the inflow, nowhere absurd
    speed = torch.linalg.vector_norm(dom.velocity, dim=1)
    assert 0.3 < float(speed.max()) < 1.5
    assert torch.allclose(dom.velocity[0], dom.velocity[1], atol=1e-4)   # identical envs stay identical
    dom.close()
