"""Multi-agent observation windows against vectors produced by the reference's own functions
(tests/golden/make_golden_obs.py), single env and with a leading env axis."""
import os

import numpy as np
import pytest
import torch

from fluidgym_amd.envs import obs_extraction as X

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_obs_windows.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})
FN = {"w2d": X.extract_moving_window_2d, "w2dxz": X.extract_moving_window_2d_x_z, "w3d": X.extract_moving_window_3d}


@pytest.mark.parametrize("case", CASES)
def test_windows_equal_reference(case):
    fn = FN[case.split("_")[0]]
    field = torch.from_numpy(GOLD[f"{case}/field"])
    args = [int(v) for v in GOLD[f"{case}/args"]]
    expected = GOLD[f"{case}/expected"]
    out = fn(field, *args)
    assert tuple(out.shape) == expected.shape
    if case.startswith("w2dxz"):   # patch means: summation order may differ
        assert np.allclose(out.numpy(), expected, rtol=1e-6, atol=1e-6)
    else:                          # pure gathers: bit-exact
        assert np.array_equal(out.numpy(), expected)
    # leading env axis: env b of the batched call equals the single-env call on field b
    batch = torch.stack([field, field.flip(-1), 2 * field])
    outb = fn(batch, *args)
    for b in range(3):
        assert torch.equal(outb[b], fn(batch[b], *args))
