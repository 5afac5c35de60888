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


# ---- the reference's own known-answer tests (tests/env_utils/test_obs_extraction.py), same parameters and assertions ----
@pytest.mark.parametrize("n_agents, agent_width, n_agents_per_window", [(8, 12, 1), (8, 12, 3), (8, 12, 5)])
def test_moving_window_2d(n_agents, agent_width, n_agents_per_window):
    torch.manual_seed(0)
    y = 10
    field_2d = torch.rand((y, n_agents * agent_width))
    half = n_agents_per_window // 2
    windows = X.extract_moving_window_2d(field_2d, n_agents=n_agents, agent_width=agent_width,
                                         n_agents_per_window=n_agents_per_window)
    assert windows.shape == (n_agents, y, n_agents_per_window * agent_width)
    total = n_agents * agent_width
    for agent_idx in range(n_agents):
        start = (agent_idx - half) * agent_width
        end = (agent_idx + half + 1) * agent_width
        if start < 0:
            expected = torch.cat((field_2d[:, start % total:], field_2d[:, :end]), dim=1)
        elif end > total:
            expected = torch.cat((field_2d[:, start:], field_2d[:, : end % total]), dim=1)
        else:
            expected = field_2d[:, start:end]
        assert torch.allclose(windows[agent_idx], expected)


@pytest.mark.parametrize("n_agents_x, n_agents_z, agent_width, n_agents_per_window_x, n_agents_per_window_z, pad_x, pad_z",
                         [(10, 20, 2, 1, 1, 0, 0), (20, 40, 2, 5, 3, 4, 1), (10, 20, 4, 5, 5, 4, 4)])
def test_moving_window_2d_x_z(n_agents_x, n_agents_z, agent_width, n_agents_per_window_x, n_agents_per_window_z, pad_x, pad_z):
    torch.manual_seed(0)
    field = torch.rand((n_agents_z * agent_width, n_agents_x * agent_width))
    field[: n_agents_per_window_z * agent_width, : n_agents_per_window_x * agent_width] = 1.0
    result = X.extract_moving_window_2d_x_z(field=field, n_agents_x=n_agents_x, n_agents_z=n_agents_z, agent_width=agent_width,
                                            n_agents_per_window_x=n_agents_per_window_x,
                                            n_agents_per_window_z=n_agents_per_window_z, pad_x=pad_x, pad_z=pad_z)
    assert result.shape == (n_agents_z * n_agents_x, n_agents_per_window_z, n_agents_per_window_x)
    expected = field[: n_agents_per_window_z * agent_width, : n_agents_per_window_x * agent_width]
    expected = expected.view(n_agents_per_window_z, agent_width, n_agents_per_window_x, agent_width).mean(dim=(1, 3))
    assert torch.allclose(result[pad_x * n_agents_z + pad_z], expected)


@pytest.mark.parametrize("n_agents, agent_width, n_agents_per_window", [(10, 1, 1), (10, 2, 3), (20, 4, 1), (20, 4, 3)])
def test_moving_window_3d(n_agents, agent_width, n_agents_per_window):
    torch.manual_seed(0)
    y = 10
    field = torch.rand((n_agents * agent_width, y, n_agents * agent_width))
    field[: n_agents_per_window * agent_width, : n_agents_per_window * agent_width] = 1.0
    result = X.extract_moving_window_3d(field=field, n_agents=n_agents, agent_width=agent_width,
                                        n_agents_per_window=n_agents_per_window)
    assert result.shape == (n_agents * n_agents, agent_width * n_agents_per_window, y, agent_width * n_agents_per_window)
    expected = field[: n_agents_per_window * agent_width, :, : n_agents_per_window * agent_width]
    agent_idx = (n_agents_per_window // 2) * n_agents + (n_agents_per_window // 2)
    assert torch.allclose(result[agent_idx], expected)
