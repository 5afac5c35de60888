"""The advective-outflow boundary update + boundary-flux balancing against vectors produced by the REFERENCE'S OWN Python
(tests/golden/make_golden_outflow.py runs pict/PISOtorch_simulation.py::update_advective_boundaries / balance_boundary_fluxes on
CPU tensors): pins the oracle's restatement (CPU) and the HIP kernels of both paths (GPU, through the C ABI)."""
import os

import numpy as np
import pytest
import torch

from oracle import piso_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_outflow.npz"))
CASES = ["c2d", "c2d_two_free", "c2d_scalar", "c3d"]


def _case(name):
    d = int(G[f"{name}_dims"])
    widths = [G[f"{name}_h{a}"] for a in range(d)]
    tol = float(G[f"{name}_tol"])
    return dict(d=d, widths=widths, free=[int(f) for f in G[f"{name}_free_faces"]], dt=float(G[f"{name}_dt"]),
                tol=None if tol < 0 else tol, velm=G[f"{name}_velm"], velocity=G[f"{name}_velocity"],
                scalar=G[f"{name}_scalar"] if f"{name}_scalar" in G else None,
                bvel_in={f: G[f"{name}_bvel_in_{f}"] for f in range(2 * d)}, bvel_out={f: G[f"{name}_bvel_out_{f}"] for f in range(2 * d)},
                bscal_in={f: G[f"{name}_bscal_in_{f}"] for f in range(2 * d)} if f"{name}_bscal_in_0" in G else None,
                bscal_out={f: G[f"{name}_bscal_out_{f}"] for f in range(2 * d)} if f"{name}_bscal_out_0" in G else None)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_reference_outflow_update(name):
    c = _case(name)
    edges = [np.concatenate([[0.0], np.cumsum(w.astype(np.float64))]) for w in c["widths"]]
    g = O.Grid(O.rectilinear_coords(edges))
    bc = {f: O.FixedBC(velocity=c["bvel_in"][f].astype(np.float64),
                       scalar=None if c["bscal_in"] is None else c["bscal_in"][f].astype(np.float64)) for f in range(2 * c["d"])}
    dom = O.Domain(grid=g, viscosity=0.01, velocity=c["velocity"].astype(np.float64), pressure=np.zeros(g.shape), bc=bc,
                   scalar=None if c["scalar"] is None else c["scalar"].astype(np.float64))
    velm = [v.astype(np.float64) for v in c["velm"]]
    O.update_advective_boundaries(dom, c["free"], velm if len(velm) > 1 else velm[0], c["dt"], tol=c["tol"])
    for f in range(2 * c["d"]):
        assert _rel(dom.bvel(f), c["bvel_out"][f]) < 2e-6, (name, f)     # fp64 restatement vs the reference's fp32 run
        if c["bscal_out"] is not None:
            assert _rel(dom.bscalar(f), c["bscal_out"][f]) < 2e-6
    # the reference did change the free faces (update + rescaling) and nothing else
    assert all(np.array_equal(c["bvel_in"][f], c["bvel_out"][f]) for f in range(2 * c["d"]) if f not in c["free"])
    assert all(not np.array_equal(c["bvel_in"][f], c["bvel_out"][f]) for f in c["free"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_single_block_outflow_matches_the_reference(name):
    """fg_update_advective_boundary + fg_balance_boundary_fluxes (csrc/fg_piso.hip: k_outflow, k_balance_fluxes)."""
    from fluidgym_amd.simulation import grids
    from fluidgym_amd.simulation.domain import Domain
    from fluidgym_amd.simulation.simulation import update_advective_boundaries

    c = _case(name)
    d, B = c["d"], 2
    edges = [np.concatenate([[0.0], np.cumsum(w.astype(np.float64))]) for w in c["widths"]]
    dom = Domain(d, torch.tensor([0.01]), passiveScalarChannels=0 if c["scalar"] is None else 1, batch=B)
    blk = dom.CreateBlock(vertexCoordinates=grids.vertex_grid(edges))
    for a in range(d):
        blk.CloseBoundary(2 * a)
    dom.PrepareSolve()
    blk.setVelocity(torch.as_tensor(c["velocity"]).unsqueeze(0).expand(B, *c["velocity"].shape).contiguous())
    if c["scalar"] is not None:
        blk.setPassiveScalar(torch.as_tensor(c["scalar"]).unsqueeze(0).expand(B, *c["scalar"].shape).contiguous())
    names = ["-x", "+x", "-y", "+y", "-z", "+z"]
    for f in range(2 * d):
        b = blk.getBoundary(names[f])
        b.setVelocity(torch.as_tensor(c["bvel_in"][f]).unsqueeze(0))
        if c["bscal_in"] is not None:
            b.setPassiveScalar(torch.as_tensor(c["bscal_in"][f]).unsqueeze(0))
    bounds = [blk.getBoundary(names[f]) for f in c["free"]]
    velms = [torch.as_tensor(v).reshape(1, d) for v in c["velm"]]
    update_advective_boundaries(dom, bounds, velms if len(velms) > 1 else velms[0], dom.solver.dt_tensor(c["dt"]), tol=c["tol"])
    torch.cuda.synchronize()
    for f in range(2 * d):
        got = blk.getBoundary(names[f]).velocity.cpu().numpy()
        for e in range(B):
            assert _rel(got[e], c["bvel_out"][f]) < 5e-6, (name, f)
        if c["bscal_out"] is not None:
            assert _rel(blk.getBoundary(names[f]).passiveScalar.cpu().numpy()[0], c["bscal_out"][f]) < 5e-6
    dom.solver.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2d", "c3d"])
def test_hip_multi_block_outflow_matches_the_reference(name):
    """fg_mb_update_advective_boundary (csrc/fg_mb_step.hip: k_mb_outflow, k_mb_bflux, k_mb_balance) on the same block given as
    a one-block curvilinear mesh."""
    from fluidgym_amd.simulation import grids
    from fluidgym_amd.simulation.multiblock import MultiBlockDomain

    c = _case(name)
    d, B = c["d"], 2
    edges = [np.concatenate([[0.0], np.cumsum(w.astype(np.float64))]) for w in c["widths"]]
    coords = grids.vertex_grid(edges)
    dom = MultiBlockDomain(d, 0.01, batch=B)
    blk = dom.CreateBlock(np.asarray(coords, np.float32))
    dom.PrepareSolve()
    N = dom.n_cells
    dom.velocity.copy_(torch.as_tensor(c["velocity"]).reshape(1, d, N).expand(B, d, N))
    for f in range(2 * d):
        blk.boundary(f).copy_(torch.as_tensor(c["bvel_in"][f]).reshape(1, d, -1).expand(B, d, -1))
    dom.update_advective_boundary(c["dt"], (blk, c["free"][0]), tuple(float(v) for v in c["velm"][0]) + (0.0,) * (3 - d),
                                  tol=1e-5 if c["tol"] is None else c["tol"])
    torch.cuda.synchronize()
    for f in range(2 * d):
        got = blk.boundary(f).cpu().numpy()
        for e in range(B):
            assert _rel(got[e].reshape(c["bvel_out"][f].shape), c["bvel_out"][f]) < 5e-6, (name, f)
    dom.close()
