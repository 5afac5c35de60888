"""Force integration against vectors from the reference's envs/util/forces.py (tests/golden/make_golden_forces.py)."""
import os

import numpy as np
import torch

from fluidgym_amd.envs.forces import compute_forces_2d, compute_forces_3d, wall_distance_from_vertices

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_forces.npz"))


def test_wall_geometry_matches_reference():
    d, n = wall_distance_from_vertices(torch.as_tensor(G["vc"]), torch.as_tensor(G["centers"]))
    assert np.allclose(d.numpy(), G["dist"], rtol=1e-12)
    assert np.allclose(n.numpy(), G["normals"], rtol=1e-12, atol=1e-14)


def test_forces_match_reference_and_batch():
    t = lambda k: torch.as_tensor(G[k])
    f = compute_forces_2d(t("u_cell"), t("u_b"), t("p"), t("normals"), t("tangent_lengths"), t("dist"), t("face_len"),
                          float(G["nu"][0]))
    assert np.allclose(f.numpy(), G["force"], rtol=1e-10)
    # leading batch dimension: env 1 carries a constant pressure offset, which a closed wall does not feel
    uc = torch.stack([t("u_cell"), t("u_cell")])
    ub = torch.stack([t("u_b"), t("u_b")])
    p = torch.stack([t("p"), t("p") + 3.0])
    fb = compute_forces_2d(uc, ub, p, t("normals"), t("tangent_lengths"), t("dist"), t("face_len"), float(G["nu"][0]))
    assert fb.shape == (2, 2)
    assert np.allclose(fb[0].numpy(), G["force"], rtol=1e-10)
    closure = (t("normals") * t("face_len")).sum(-1)  # sum of n dl over the closed polygon = 0
    assert torch.allclose(fb[1] - fb[0], -3.0 * closure, atol=1e-10) and closure.abs().max() < 1e-12


def test_forces_3d_match_reference_per_layer():
    t = lambda k: torch.as_tensor(G[k])
    f = compute_forces_3d(t("u3"), t("ub3"), t("p3"), t("normals"), t("tangent_lengths"), t("dist"), t("areas"), float(G["nu"][0]))
    assert f.shape == (2, 5)
    assert np.allclose(f.numpy(), G["force3"], rtol=1e-10)
    fb = compute_forces_3d(torch.stack([t("u3")] * 2), torch.stack([t("ub3")] * 2), torch.stack([t("p3"), t("p3") + 1.0]),
                           t("normals"), t("tangent_lengths"), t("dist"), t("areas"), float(G["nu"][0]))
    assert fb.shape == (2, 2, 5) and np.allclose(fb[0].numpy(), G["force3"], rtol=1e-10)
    assert np.allclose(fb[1].numpy(), G["force3"], rtol=1e-9, atol=1e-12)   # closed ring: a pressure offset exerts no force
