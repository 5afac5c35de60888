"""The cylinder mesh builder against vectors recorded from the reference's own make_vortex_street_domain
(tests/golden/make_golden_cylinder.py)."""
import os

import numpy as np
import pytest

from fluidgym_amd.envs.cylinder_grid import make_vortex_street_mesh

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_cylinder_grid.npz"))


@pytest.mark.parametrize("res", [8, 24])
def test_vertex_coordinates_match_reference(res):
    m = make_vortex_street_mesh(res)
    for b in range(5):
        ref = G[f"r{res}_block{b}"]
        assert m.coords[b].shape == ref.shape, (b, m.coords[b].shape, ref.shape)
        assert np.abs(m.coords[b] - ref).max() < 2e-6, b
    # neighbouring blocks share their interface vertices exactly enough for a watertight mesh
    left, top = m.coords[0], m.coords[1]
    assert np.abs(left[:, -1, :] - top[:, ::-1, 0]).max() < 1e-6


@pytest.mark.parametrize("res", [8, 24])
def test_boundaries_and_connections_match_reference_calls(res):
    m = make_vortex_street_mesh(res)
    calls = [str(c) for c in G[f"r{res}_calls"]]
    closed = {(int(c.split()[1]), c.split()[2]) for c in calls if c.startswith("close")}
    assert closed == set(m.fixed.keys())
    conns = [tuple(c.split()[1:]) for c in calls if c.startswith("connect")]
    assert conns == [(str(a), fa, str(b), fb, ax) for a, fa, b, fb, ax in m.connections]
    assert [c.split()[2] for c in calls if c.startswith("block")] == m.names
    assert "varying 4 +x" in calls and m.outflow == (4, "+x")
    inflow = G[f"r{res}_velocity_0_-x"][0, :, :, 0]
    assert np.abs(m.fixed[(0, "-x")] - inflow).max() < 1e-6
    assert np.abs(m.fixed[(4, "+x")] - G[f"r{res}_velocity_4_+x"][0, :, :, 0]).max() < 1e-6


def test_cells_are_right_handed():
    from oracle.piso_oracle import coords_to_transforms

    m = make_vortex_street_mesh(8)
    for c in m.coords:
        _, _, det = coords_to_transforms(c.astype(np.float64))
        assert det.min() > 0


@pytest.mark.parametrize("level,res", [("easy", 24), ("medium", 32)])
def test_sensor_pixels_match_reference(level, res):
    import torch

    import fluidgym_amd

    env = fluidgym_amd.make(f"CylinderJet2D-{level}-v0", cuda_device=torch.device("cpu"))
    ref = G[f"r{res}_sensor_pixels"]
    assert env._sensor_locations.shape == ref.shape == (2, 151)
    assert (env._sensor_locations == ref).all()


def test_extruded_mesh_matches_reference_3d():
    from fluidgym_amd.envs.cylinder_grid import extrude_mesh

    m = extrude_mesh(make_vortex_street_mesh(8), 8)
    for b in range(5):
        ref = G[f"r8_3d_block{b}"]
        assert m.coords[b].shape == ref.shape
        assert np.abs(m.coords[b] - ref).max() < 2e-6
    calls = [str(c) for c in G["r8_3d_calls"]]
    conns = [tuple(c.split()[1:]) for c in calls if c.startswith("connect")]
    assert conns == [(str(c[0]), c[1], str(c[2]), c[3], c[4], c[5]) for c in m.connections]
    assert [tuple(c.split()[1:]) for c in calls if c.startswith("periodic")] == [(str(b), a) for b, a in m.periodic]
    inflow = G["r8_3d_velocity_0_-x"][0, :, :, :, 0].reshape(3, -1)      # [3, z, y] -> y fastest
    assert np.abs(m.fixed[(0, "-x")] - inflow).max() < 1e-6


@pytest.mark.parametrize("res,n_jets", [(8, 4), (24, 8)])
def test_3d_sensor_pixels_match_reference(res, n_jets):
    import torch
    import fluidgym_amd

    env = fluidgym_amd.make("CylinderJet3D-easy-v0", cuda_device=torch.device("cpu"), resolution=res, n_jets=n_jets)
    assert np.array_equal(env._sensor_locations, G[f"r{res}_3d_sensor_pixels"])


class _FakeDomain:
    """Index ramps in place of the fields: with one-hot sensor rows the env's gather returns the flat pixel index of
    every sensor, which is what the golden generator fed the reference's observation code."""

    def __init__(self, rs):
        import torch
        P = rs[0] * rs[1] * rs[2]
        self.velocity = torch.arange(3 * P, dtype=torch.float32).reshape(1, 3, P)
        self.pressure = -torch.arange(P, dtype=torch.float32).reshape(1, P)


def _obs_env(use_marl):
    import torch
    import fluidgym_amd

    env = fluidgym_amd.make("CylinderJet3D-easy-v0", cuda_device=torch.device("cpu"), resolution=8, n_jets=4, use_marl=use_marl)
    rs = env.render_shape
    env._domain = _FakeDomain(rs)
    px = env._sensor_locations.reshape(3, -1)
    from fluidgym_amd.simulation.resample_mb import SensorGather
    env._sensors = SensorGather(torch.as_tensor(px[0] + rs[0] * (px[1] + rs[1] * px[2]))[:, None], torch.ones(px.shape[1], 1))
    return env


def test_3d_global_observation_layout_matches_reference():
    env = _obs_env(False)
    obs = env._get_global_obs()
    assert np.array_equal(obs["velocity"][0].numpy(), G["r8_3d_obs_global_velocity"])
    assert np.array_equal(obs["pressure"][0].numpy(), G["r8_3d_obs_global_pressure"])


def test_3d_local_observation_windows_match_reference():
    env = _obs_env(True)
    obs = env._get_local_obs()
    assert np.array_equal(obs["velocity"][0].numpy(), G["r8_3d_obs_local_velocity"])
    assert np.array_equal(obs["pressure"][0].numpy(), G["r8_3d_obs_local_pressure"])


def test_3d_jet_wall_velocities_match_reference():
    import torch
    import fluidgym_amd
    from fluidgym_amd.envs.cylinder import _face_vertices
    from fluidgym_amd.envs.cylinder_grid import BOTTOM, TOP, extrude_mesh

    env = fluidgym_amd.make("CylinderJet3D-easy-v0", cuda_device=torch.device("cpu"), resolution=8, n_jets=4)
    mesh = extrude_mesh(make_vortex_street_mesh(8), 8)
    top = env._jet_velocities(_face_vertices(mesh, TOP, "-y"), True)        # [2, nx]
    bottom = env._jet_velocities(_face_vertices(mesh, BOTTOM, "+y"), False)
    ref_t, ref_b = G["r8_3d_jet_top"][0, :, :, 0, :], G["r8_3d_jet_bottom"][0, :, :, 0, :]   # [2?, nz, nx]
    assert ref_t.shape[0] == 3 and int(G["r8_3d_nz_per_agent"]) == 2
    assert np.abs(ref_t[2]).max() == 0 and np.abs(ref_b[2]).max() == 0
    assert np.abs(ref_t[:2] - top[:, None, :]).max() < 1e-6 and np.abs(ref_b[:2] - bottom[:, None, :]).max() < 1e-6
    assert np.abs(top).max() > 0.5
