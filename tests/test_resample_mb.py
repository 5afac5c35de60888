"""Multi-block resampling: oracle and the product's folded operator against vectors from the reference's pure-torch
implementation on the recorded cylinder mesh (tests/golden/make_golden_resample_mb.py)."""
import os

import numpy as np
import pytest

from fluidgym_amd.simulation.resample_mb import build_operator
from oracle.resample_oracle import resample_blocks_to_uniform

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "reference_resample_mb.npz"))
M = np.load(os.path.join(HERE, "golden", "reference_cylinder_grid.npz"))
COORDS = [M[f"r8_block{b}"] for b in range(5)]


@pytest.mark.parametrize("case", ["c2_f16", "c1_f0"])
def test_oracle_matches_reference(case):
    fill = int(case.split("_f")[1])
    data = [G[f"{case}/data{b}"][0] for b in range(5)]
    out = resample_blocks_to_uniform(data, COORDS, G["out_shape"], fill)
    exp = G[f"{case}/expected"][0]
    assert out.shape == exp.shape
    assert np.abs(out - exp).max() < 2e-5
    assert ((out == 0) == (exp == 0)).all()


@pytest.mark.parametrize("case", ["c2_f16", "c1_f0"])
def test_folded_operator_matches_reference(case):
    fill = int(case.split("_f")[1])
    W = build_operator(COORDS, G["out_shape"], fill)
    data = np.concatenate([G[f"{case}/data{b}"][0].reshape(G[f"{case}/data{b}"].shape[1], -1) for b in range(5)], axis=1)
    exp = G[f"{case}/expected"][0]
    out = (W @ data.T.astype(np.float64)).T.reshape(exp.shape)
    assert np.abs(out - exp).max() < 2e-5
    # rows are convex combinations of cell values (or empty)
    rs = np.asarray(W.sum(axis=1)).reshape(-1)
    assert np.all((np.abs(rs - 1.0) < 1e-5) | (rs == 0.0))
    assert W.data.min() >= 0.0
