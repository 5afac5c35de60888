"""Observation resampling: oracle pinned on the reference's torch implementation, HIP gather against both.

The golden file holds outputs of ``sample_multi_coords_to_uniform_grid_diff`` (reference ``data/resample.py:361-548``)
made by ``tests/golden/make_golden_resample.py``.  The reference's own test compares its two implementations with
atol = rtol = 1e-3 (``tests/simulation/test_torch_resample.py:76,93,111``); we hold 1e-5 of the field maximum.
"""
import os

import numpy as np
import pytest
import torch

from oracle import resample_oracle as R

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_resample.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files})
TOL = 1e-5


def _case(name):
    d = len(GOLD[f"{name}/out_shape"])
    edges = [GOLD[f"{name}/edges{a}"] for a in range(d)]
    return edges, GOLD[f"{name}/data"], [int(v) for v in GOLD[f"{name}/out_shape"]], int(GOLD[f"{name}/fill"]), GOLD[f"{name}/expected"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_torch_implementation(name):
    edges, data, oshape, fill, expected = _case(name)
    out = R.resample_to_uniform(data[0], edges, oshape, fill)
    assert out.shape == expected[0].shape
    assert np.abs(out - expected[0]).max() <= TOL * np.abs(expected).max()
    # cells nobody wrote stay exactly zero (reference: zeros + index_add)
    assert np.array_equal(out == 0, expected[0] == 0)


def test_host_axis_maps_equal_oracle():
    from fluidgym_amd.simulation.resample import aabb_outer_axis_maps, output_shape

    for name in NAMES:
        edges, _, oshape, _, _ = _case(name)
        for a, b in zip(aabb_outer_axis_maps(edges, oshape), R.aabb_outer_axis_maps(edges, oshape)):
            assert np.array_equal(a, b)
    assert output_shape(7, 3) == [7, 7, 7]
    with pytest.raises(ValueError):
        output_shape([4, 4], 3)


def test_quirk_drops_exactly_the_two_upper_yz_corners():
    """resampling.cu:320 loops idx < (DIMS << 1): in 3-D corners 6 and 7 (y upper and z upper) are never written."""
    rng = np.random.default_rng(1)
    n = (5, 4, 6)
    maps = [np.sort(rng.uniform(0.2, s - 1.2, s)) for s in n]  # continuous indices well inside a grid of the same size
    data = rng.standard_normal((1, n[2], n[1], n[0]))
    full, wfull = R.splat(data, maps, n, corners_3d_quirk=False)
    part, wpart = R.splat(data, maps, n, corners_3d_quirk=True)
    # the difference is the (y upper, z upper) contribution: recompute it directly
    miss = np.zeros_like(wfull)
    for k in range(n[2]):
        for j in range(n[1]):
            for i in range(n[0]):
                bx, by, bz = (int(np.floor(maps[0][i])), int(np.floor(maps[1][j])), int(np.floor(maps[2][k])))
                fx, fy, fz = maps[0][i] - bx, maps[1][j] - by, maps[2][k] - bz
                for ux in (0, 1):
                    miss[bz + 1, by + 1, bx + ux] += (fx if ux else 1 - fx) * fy * fz
    assert np.allclose(wfull - wpart, miss, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_gather_matches_reference_vectors(name):
    from fluidgym_amd.simulation.resample import UniformResampler

    edges, data, oshape, fill, expected = _case(name)
    rs = UniformResampler(edges, oshape, fill_max_steps=fill, compiled_corner_rule=False)
    out = rs(torch.from_numpy(data).cuda()).cpu().numpy()
    assert out.shape == expected.shape
    assert np.abs(out - expected).max() <= TOL * np.abs(expected).max()
    assert np.array_equal(out == 0, expected == 0)
    rs.close()


@pytest.mark.gpu
def test_hip_gather_batched_and_compiled_corner_rule():
    """Batch of envs in one call; 3-D with the compiled kernel's 6-corner rule against the oracle restatement."""
    from fluidgym_amd.simulation.resample import UniformResampler

    rng = np.random.default_rng(5)
    n = (14, 9, 11)
    edges = [np.concatenate([[0.0], np.cumsum(1.0 + 0.3 * rng.uniform(-1, 1, s))]).astype(np.float32) for s in n]
    data = rng.standard_normal((3, 2, n[2], n[1], n[0])).astype(np.float32)
    for oshape, fill in (((7, 6, 5), 0), ((30, 20, 24), 16)):
        rs = UniformResampler(edges, oshape, fill_max_steps=fill)  # default: compiled corner rule
        out = rs(torch.from_numpy(data).cuda()).cpu().numpy()
        for b in range(3):
            ref = R.resample_to_uniform(data[b], edges, oshape, fill, corners_3d_quirk=True)
            assert np.abs(out[b] - ref).max() <= TOL * np.abs(ref).max()
        rs.close()
    # 2-D: large block, many source cells per output cell
    n2 = (256, 128)
    e2 = [np.linspace(0, 22.0, n2[0] + 1).astype(np.float32), np.linspace(-2.05, 2.05, n2[1] + 1).astype(np.float32)]
    d2 = rng.standard_normal((4, 3, n2[1], n2[0])).astype(np.float32)
    rs = UniformResampler(e2, (96, 20), fill_max_steps=0)
    out = rs(torch.from_numpy(d2).cuda()).cpu().numpy()
    for b in range(4):
        ref = R.resample_to_uniform(d2[b], e2, (96, 20), 0)
        assert np.abs(out[b] - ref).max() <= TOL * np.abs(ref).max()
    rs.close()
