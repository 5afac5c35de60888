"""The airfoil mesh construction against the reference's recorded meshes (tests/golden/make_golden_airfoil.py)."""
import os

import numpy as np
import pytest

from fluidgym_amd.envs.airfoil_grid import make_airfoil_mesh, naca0012_sharp, surface_from_blocks

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_airfoil_grid.npz"))


def _surface(tag, aoa):
    return surface_from_blocks(G[f"{tag}_block1"], G[f"{tag}_block2"], G[f"{tag}_block3"], aoa)


@pytest.mark.parametrize("aoa", [20, 0])
def test_mesh_reproduces_the_recorded_reference_mesh(aoa):
    """Fed with the surface polyline the recorded mesh contains, the construction returns that mesh: split indices,
    block sizes, spacing laws, transfinite patches, inflow profile, connections."""
    tag = f"aoa{aoa}"
    m = make_airfoil_mesh(attack_angle_deg=float(aoa), surface=_surface(tag, float(aoa)))
    for b in range(6):
        ref = G[f"{tag}_block{b}"]
        assert m.coords[b].shape == ref.shape
        assert np.abs(m.coords[b] - ref).max() < 2e-7
    calls = [str(c).split() for c in G[f"{tag}_calls"]]
    assert [c[2] for c in calls if c[0] == "block"] == m.names
    conns = [(int(c[1]), c[2], int(c[3]), c[4], c[5]) for c in calls if c[0] == "connect"]
    assert conns == m.connections
    closed = sorted((int(c[1]), c[2]) for c in calls if c[0] == "close")
    assert closed == sorted(m.fixed)
    inflow = G[f"{tag}_velocity_0_-x"][0, :, :, 0]
    assert np.abs(m.fixed[(0, "-x")] - inflow).max() < 1e-6
    for b in (4, 5):
        assert np.allclose(m.fixed[(b, "+x")], G[f"{tag}_velocity_{b}_+x"].reshape(2, 1))
    assert m.outflows == [(4, "+x"), (5, "+x")]


@pytest.mark.parametrize("div,aoa", [(1, 10.0), (2, 10.0), (4, 10.0), (2, 0.0)])
def test_own_section_gives_a_valid_mesh_at_every_resolution(div, aoa):
    """With the closed-form section (no table): right-handed cells everywhere, blocks meet along shared vertices."""
    m = make_airfoil_mesh(attack_angle_deg=aoa, resolution_div=div)
    for c in m.coords:
        c = c.astype(np.float64)
        ex = c[:, :-1, 1:] - c[:, :-1, :-1]
        ey = c[:, 1:, :-1] - c[:, :-1, :-1]
        assert (ex[0] * ey[1] - ex[1] * ey[0]).min() > 0
    left, front, top, bot, tu, tl = m.coords
    assert np.allclose(left[:, :, -1], front[:, :, 0], atol=1e-6)
    assert np.allclose(front[:, -1, :], top[:, :, 0][:, ::1], atol=1e-6) or np.allclose(front[:, -1, :], top[:, ::-1, 0], atol=1e-6)
    assert np.allclose(top[:, :, -1], tu[:, :, 0], atol=1e-6) and np.allclose(bot[:, :, -1], tl[:, :, 0], atol=1e-6)
    assert np.allclose(tu[:, 0, :], tl[:, -1, :], atol=1e-6)
    assert top.shape[1] == 96 // div


def test_closed_form_section():
    s = naca0012_sharp()
    assert s.shape == (160, 2) and np.allclose(s[0], [1, 0]) and np.allclose(s[-1], [1, 0])
    assert np.allclose(s[1:80, 1], -s[-2:79:-1, 1]) and abs(s[:, 1].max() - 0.06) < 5e-4   # symmetric, 12 % thick
    ref = _surface("aoa0", 0.0)
    assert np.abs(np.interp(ref[:80, 0][::-1], s[:80, 0][::-1], s[:80, 1][::-1]) - ref[:80, 1][::-1]).max() < 1e-3


@pytest.mark.parametrize("aoa", [10, 20, 0])
def test_env_mask_and_sensor_pixels_match_reference(aoa):
    """Section mask on the render grid (the tie rules of the reference's polygon test matter: pixel centres and polygon
    vertices are integers) and the sensors left after masking, from the surface the recorded meshes contain."""
    import torch
    import fluidgym_amd

    tag = f"aoa{aoa}"
    env = fluidgym_amd.make("Airfoil2D-easy-v0", cuda_device=torch.device("cpu"), attack_angle_deg=float(aoa),
                            surface=_surface("aoa0", 0.0))
    mask = np.unpackbits(G[f"{tag}_mask_rows"], axis=1)[:, :600].astype(bool)
    assert np.array_equal(env._airfoil_mask, mask)
    assert np.array_equal(env._sensor_locations, G[f"{tag}_sensor_pixels"])
    assert env.observation_space["velocity"].shape == (G[f"{tag}_sensor_pixels"].shape[1], 2)
    assert env.action_space.shape == (3,) and env._n_sim_steps == 5
    # the closed-form section gives the same sensors (its mask differs by a few edge pixels only)
    own = fluidgym_amd.make("Airfoil2D-easy-v0", cuda_device=torch.device("cpu"), attack_angle_deg=float(aoa))
    assert np.array_equal(own._sensor_locations, env._sensor_locations)
    assert int((own._airfoil_mask != mask).sum()) < 50


@pytest.mark.parametrize("aoa", [20, 0])
def test_jet_cell_ranges_match_reference(aoa):
    from fluidgym_amd.envs.airfoil import jet_locations

    tag = f"aoa{aoa}"
    assert jet_locations(G[f"{tag}_block2"]) == G[f"{tag}_jet_locations"].tolist()
    m = make_airfoil_mesh(attack_angle_deg=float(aoa), surface=_surface(tag, float(aoa)))
    assert jet_locations(m.coords[2]) == G[f"{tag}_jet_locations"].tolist()


def test_airfoil_registry_and_argument_checks():
    import torch
    import fluidgym_amd

    for level, re in (("easy", 1e3), ("medium", 3e3), ("hard", 5e3)):
        env = fluidgym_amd.make(f"Airfoil2D-{level}-v0", cuda_device=torch.device("cpu"))
        assert env._reynolds_number == re and env._attack_angle_deg == 10.0 and env.render_shape == (600, 150, 150)
        assert abs(env._nu - 0.3 / re) < 1e-12
    with pytest.raises(ValueError, match="between 0 and 20"):
        fluidgym_amd.make("Airfoil2D-easy-v0", cuda_device=torch.device("cpu"), attack_angle_deg=25.0)
    e3 = fluidgym_amd.make("Airfoil3D-hard-v0", cuda_device=torch.device("cpu"))
    assert (e3._reynolds_number, e3._ndims, e3._res_z, e3.n_agents) == (5e3, 3, 96, 1)
    assert e3.action_space.shape == (4, 3) and e3.observation_space["velocity"].shape == (4, 1, 3, 209)
    m3 = fluidgym_amd.make("Airfoil3D-easy-v0", cuda_device=torch.device("cpu"), use_marl=True, local_obs_window=3)
    assert m3.n_agents == 4 and m3.action_space.shape == (3,) and m3.observation_space["pressure"].shape == (3, 1, 209)
    with pytest.raises(ValueError, match="evenly divides"):
        fluidgym_amd.make("Airfoil3D-easy-v0", cuda_device=torch.device("cpu"), n_agents=5)


def test_3d_sensor_pixels_match_reference():
    import torch
    import fluidgym_amd

    env = fluidgym_amd.make("Airfoil3D-easy-v0", cuda_device=torch.device("cpu"), surface=_surface("aoa0", 0.0))
    assert np.array_equal(env._sensor_locations, G["aoa10_3d_sensor_pixels"])
    own = fluidgym_amd.make("Airfoil3D-easy-v0", cuda_device=torch.device("cpu"))
    assert np.array_equal(own._sensor_locations, G["aoa10_3d_sensor_pixels"])


def test_extruded_airfoil_mesh_connections_follow_the_reference_calls():
    """grid.py:690-706: x faces connect with (in-plane axis, -z), y faces with (-z, in-plane axis); all blocks z-periodic."""
    from fluidgym_amd.envs.cylinder_grid import extrude_mesh

    m = extrude_mesh(make_airfoil_mesh(resolution_div=4), 4, -0.7, 0.7)
    assert m.connections == [(0, "+x", 1, "-x", "-y", "-z"), (1, "+y", 2, "-x", "-z", "+y"), (1, "-y", 3, "-x", "-z", "-y"),
                             (2, "+x", 4, "-x", "-y", "-z"), (3, "+x", 5, "-x", "-y", "-z"), (4, "-y", 5, "+y", "-z", "-x")]
    assert m.periodic == [(b, "z") for b in range(6)] and m.outflows == [(4, "+x"), (5, "+x")]
    assert m.coords[2].shape[1] == 5 and np.allclose(m.coords[2][2, 0], -0.7) and np.allclose(m.coords[2][2, -1], 0.7)
