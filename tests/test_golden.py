"""Golden vectors produced by the reference's importable Python (tests/golden/make_golden.py):
our own grid generators / profiles must reproduce them, and the oracle's metric computation must be
consistent with grids the reference would hand to its kernels."""
import os

import numpy as np
import pytest

from fluidgym_amd.envs.channel import inflow_profile, jet_profile
from fluidgym_amd.simulation import grids as G
from oracle import piso_oracle as O

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_python.npz"))


@pytest.mark.parametrize("name,fn", [
    ("weights_exp_both_10_1p2", lambda: G.weights_exp(10, 1.2, "BOTH")),
    ("weights_exp_start_7_1p1", lambda: G.weights_exp(7, 1.1, "START")),
    ("weights_exp_end_7_1p1", lambda: G.weights_exp(7, 1.1, "END")),
    ("weights_cos_both_12", lambda: G.weights_cos(12, "BOTH")),
    ("weights_exp_global_16_40", lambda: G.weights_exp_global(16, 40.0, "BOTH")),
    ("tcf_y_weights_N1_h48", lambda: G.tcf_y_weights(1, 48)),
    ("tcf_y_weights_N2_h48", lambda: G.tcf_y_weights(2, 48)),
    ("tcf_y_weights_N1_h16", lambda: G.tcf_y_weights(1, 16)),
    ("tcf_y_weights_N2_h32", lambda: G.tcf_y_weights(2, 32)),
])
def test_weight_laws(name, fn):
    assert np.abs(np.asarray(fn()) - GOLD[name]).max() < 1e-12


@pytest.mark.parametrize("name", ["rbc_96x61", "rbc_16x9", "rbc_uniform_12x8"])
def test_rbc_vertex_grids(name):
    nx, ny, L, base = GOLD[name + ".args"]
    edges = G.wall_refined_edges(int(nx), int(ny), (0, -0.5), (L, 0.5), ["-y", "+y"], base)
    ref = GOLD[name + ".coords"]
    assert np.abs(G.vertex_grid(edges).numpy() - ref).max() < 1e-6
    back = G.edges_from_vertex_grid(ref)
    assert np.abs(back[0] - edges[0]).max() < 1e-6 and np.abs(back[1] - edges[1]).max() < 1e-6
    # the oracle accepts the reference's grid and finds it orthogonal with J = hx*hy
    g = O.Grid(ref.astype(np.float64))
    hx, hy = np.diff(back[0]), np.diff(back[1])
    assert np.allclose(g.det, hy[:, None] * hx[None, :], rtol=1e-5)


def test_extruded_grids():
    e = G.wall_refined_edges(8, 5, (0, -0.5), (2.0, 0.5), ["-y", "+y"], 1.02)
    e.append(G.lerp_edges(0.0, 2.0, G.weights_linear(6)))
    assert np.abs(G.vertex_grid(e).numpy() - GOLD["rbc3d_8x5x6.coords"]).max() < 1e-6
    H, L, D, x, yh, yN, z = GOLD["tcf_8x16x4.args"]
    yw = G.tcf_y_weights(N=int(yN), ny_half=int(yh))
    e = [G.lerp_edges(-L / 2, L / 2, G.weights_linear(int(x))), G.lerp_edges(-H / 2, H / 2, yw),
         G.lerp_edges(-D / 2, D / 2, G.weights_linear(int(z)))]
    assert np.abs(G.vertex_grid(e).numpy() - GOLD["tcf_8x16x4.coords"]).max() < 1e-6
    g = O.Grid(GOLD["tcf_8x16x4.coords"].astype(np.float64))
    assert g.shape == (4, 16, 8)


def test_profiles():
    assert np.abs(jet_profile(7) - GOLD["jet_profile_h7"]).max() < 1e-6
    ref2 = GOLD["inflow_2d_h2_res16"]  # [1,2,16,1]
    assert np.abs(inflow_profile(2.0, 16) - ref2[0, 0, :, 0]).max() < 1e-6
    assert np.abs(ref2[0, 1]).max() == 0
    ref3 = GOLD["inflow_3d_h1_res8_z3"]  # [1,3,3,8,1]
    assert np.abs(inflow_profile(1.0, 8)[None, :] - ref3[0, 0, :, :, 0]).max() < 1e-6


def test_non_rectilinear_grid_is_rejected():
    c = GOLD["rbc_16x9.coords"].copy()
    c[0, 0] += 0.05 * c[0, 1]
    with pytest.raises(ValueError):
        G.edges_from_vertex_grid(c)
