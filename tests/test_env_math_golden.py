"""Arithmetic inside the env classes against the reference's OWN methods evaluated on the same inputs
(``tests/golden/make_golden_env_math.py`` takes the method out of the reference's source with ``ast`` and calls it;
``tests/golden/reference_env_math.npz`` holds inputs and outputs): the RBC Nusselt number, the TCF wall actuation and its
time units."""
import os
from types import SimpleNamespace

import numpy as np
import torch

import fluidgym_amd  # noqa: F401
from fluidgym_amd.envs.rbc import RBCEnvBase
from fluidgym_amd.envs.tcf import TCF3DBottomEnv

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_env_math.npz"))


def test_rbc_nusselt_numbers_are_the_reference_s():
    ra, pr = G["nusselt_ra_pr"]
    me = SimpleNamespace(_rayleigh_number=float(ra), _prandtl_number=float(pr))
    # global: one value per env from the fields on the simulation grid (compute_global_nusselt reads them off the block)
    for tag, nd in (("2d_batched", 2), ("3d_batched", 3)):
        T, uy, cs = (torch.as_tensor(G[f"nusselt_{tag}_{k}"]) for k in ("T", "uy", "cell_size"))
        blk = SimpleNamespace(passiveScalar=T.unsqueeze(1), velocity=torch.stack([torch.zeros_like(uy), uy] + [torch.zeros_like(uy)] * (nd - 2), dim=1))
        env = SimpleNamespace(_block=blk, _ndims=nd, _cell_size=cs, **vars(me))
        got = RBCEnvBase.compute_global_nusselt(env)
        assert np.allclose(got.numpy(), G[f"nusselt_{tag}_out"], rtol=2e-6, atol=1e-5), tag
        # local (per-agent windows): the same formula with an agent axis
        got = RBCEnvBase._local_nusselt(env, T.unsqueeze(1), uy.unsqueeze(1), cs)
        assert np.allclose(got[:, 0].numpy(), G[f"nusselt_{tag}_out"], rtol=2e-6, atol=1e-5), tag
    T, uy, cs = (torch.as_tensor(G[f"nusselt_2d_{k}"]) for k in ("T", "uy", "cell_size"))
    blk = SimpleNamespace(passiveScalar=T[None, None], velocity=torch.stack([torch.zeros_like(uy), uy])[None])
    got = RBCEnvBase.compute_global_nusselt(SimpleNamespace(_block=blk, _ndims=2, _cell_size=cs, **vars(me)))
    assert np.allclose(got.numpy(), G["nusselt_2d_out"], rtol=2e-6)


def test_tcf_wall_actuation_is_the_reference_s():
    u_wall, actor = G["tcf_u_wall_actor_size"]
    for tag, scale in (("scaled", True), ("raw", False)):
        a = torch.as_tensor(G[f"tcf_action_{tag}"])
        me = SimpleNamespace(_scale_actions=scale, _u_wall=float(u_wall), _actor_size=int(actor))
        v = TCF3DBottomEnv._action_to_control(me, a[None])            # [B, Z, X]
        want = G[f"tcf_control_{tag}"]                                # [1, 3, Z, 1, X]: only the wall-normal component is set
        assert np.abs(want[0, 0]).max() == 0 and np.abs(want[0, 2]).max() == 0
        assert np.allclose(v[0].numpy(), want[0, 1, :, 0, :], rtol=1e-6, atol=1e-8), tag
        if scale:
            assert abs(float(v.mean())) < 1e-8 and float(v.abs().max()) <= float(u_wall) * 1.5
        # two envs of a batch are scaled independently, each like the reference's single env
        both = TCF3DBottomEnv._action_to_control(me, torch.stack([a, -0.5 * a]))
        assert np.allclose(both[0].numpy(), want[0, 1, :, 0, :], rtol=1e-6, atol=1e-8)


def test_tcf_time_units_are_the_reference_s():
    nu, u_wall, t_wall_of, t_of = G["tcf_time_units"]
    me = SimpleNamespace(_nu=float(nu), _u_wall=float(u_wall))
    assert np.isclose(TCF3DBottomEnv._t_to_t_wall(me, 0.37), t_wall_of, rtol=1e-6)
    assert np.isclose(TCF3DBottomEnv._t_wall_to_t(me, 0.6), t_of, rtol=1e-6)


def test_rbc_heater_profile_is_the_reference_s():
    """actions -> temperature of the bottom plate (rbc_env_2d.py:207-262): mean removed, limited to +-0.75 around T_hot, blended
    between neighbouring heaters over 10 % of a heater width (zones of 1, 2 and 0 cells here)."""
    limit, t_hot = G["rbc_heater_limit_T_hot"]
    for tag, hw in (("w8", 8), ("w20", 20), ("w4", 4)):
        a = torch.as_tensor(G[f"rbc_heater_{tag}_action"])
        x = a.numel() * hw
        idx = torch.arange(x)
        me = SimpleNamespace(_heater_width=hw, _x=x, _seg_id=idx // hw, _x_pos=idx % hw, _num_envs=2, _heater_limit=float(limit), _T_hot=float(t_hot))
        T = RBCEnvBase._action_to_control(me, torch.stack([a, 0.3 * a]))             # two envs, each handled like the reference's one
        prof = RBCEnvBase._smooth_profile(me, T)
        assert np.allclose(prof[0].numpy(), G[f"rbc_heater_{tag}_control"], rtol=1e-6, atol=1e-7), tag
        assert not np.allclose(prof[1].numpy(), prof[0].numpy())


def test_rbc_randomisation_of_the_initial_state_is_the_reference_s():
    """rbc_env_base.py:335-398 on the same fields with the same generators (CPU): x flip, in 3-D a z flip, x shift, in 3-D a z
    shift, noise on T (clamped) and u, then 1-2 time units of simulation -- the same fields and the same number of steps."""
    for tag, ndims in (("2d", 2), ("3d", 3)):
        for seed in (3, 4, 6):
            T = torch.as_tensor(G[f"rbc_rand_{tag}_{seed}_T0"]).clone()
            u = torch.as_tensor(G[f"rbc_rand_{tag}_{seed}_u0"]).clone()
            steps = []
            me = SimpleNamespace(_block=SimpleNamespace(passiveScalar=T, velocity=u), _ndims=ndims, _x=8, _np_rng=np.random.default_rng(seed),
                                 _torch_rng_cuda=torch.Generator().manual_seed(seed), _T_cold=0.0, _T_hot=1.0, _dt=0.05,
                                 _sim=SimpleNamespace(single_step=lambda: steps.append(1)))
            RBCEnvBase._randomize_domain(me)
            assert np.allclose(T.numpy(), G[f"rbc_rand_{tag}_{seed}_T"], atol=1e-7), (tag, seed)
            assert np.allclose(u.numpy(), G[f"rbc_rand_{tag}_{seed}_u"], atol=1e-7), (tag, seed)
            assert len(steps) == int(G[f"rbc_rand_{tag}_{seed}_steps"])


def test_tcf_observations_are_the_reference_s():
    """tcf_env.py:646-678 (global: fluctuation about the volume-weighted mean on the sensing plane) and :918-992 (per-actuator
    windows of the fluctuation about the plane mean; u_x padded by one column less than u_y / p; the top wall's view flipped and
    u_y negated) -- the reference's methods with its own window extraction, on the same fields."""
    u, p, cs = (torch.as_tensor(G[k]) for k in ("tcf_obs_u", "tcf_obs_p", "tcf_obs_cell_size"))
    Z, Y, X = u.shape[2:]
    actor = 2
    blk = SimpleNamespace(velocity=torch.cat([u, 2.0 * u]), pressure=torch.cat([p, 2.0 * p]))     # two envs: the second must not leak
    for W in (1, 3):
        me = SimpleNamespace(_block=blk, _cell_size=cs[0, 0], _y_obs_bottom_idx=1, _n_actors_x=X // actor, _n_actors_z=Z // actor,
                             _actor_size=actor, _local_obs_window=W)
        g = TCF3DBottomEnv._plane_obs(me, 1)
        assert np.allclose(g["velocity"][0].numpy(), G["tcf_global_velocity"], atol=2e-6)
        assert np.allclose(g["pressure"][0].numpy(), G["tcf_global_pressure"], atol=2e-6)
        for flip, y_idx in ((False, 1), (True, Y - 2)):
            l = TCF3DBottomEnv._local_plane_obs(me, y_idx, flip)
            assert np.allclose(l["velocity"][0].numpy(), G[f"tcf_local_w{W}_flip{int(flip)}_velocity"], atol=2e-6), (W, flip)
            assert np.allclose(l["pressure"][0].numpy(), G[f"tcf_local_w{W}_flip{int(flip)}_pressure"], atol=2e-6), (W, flip)
            assert np.allclose(l["velocity"][1].numpy(), 2.0 * G[f"tcf_local_w{W}_flip{int(flip)}_velocity"], atol=4e-6)


def test_rbc_3d_heater_profile_is_the_reference_s():
    """rbc_env_3d.py:201-262: the [n_heaters, n_heaters] actions (z-heater, x-heater) -> bottom-plate temperature [Z, X], blended
    along both axes."""
    limit, t_hot = G["rbc_heater_limit_T_hot"]
    a = torch.as_tensor(G["rbc3d_heater_action"])
    nh, hw = 4, 10
    idx = torch.arange(nh * hw)
    plate = SimpleNamespace(setPassiveScalar=lambda c: setattr(plate, "value", c))
    me = SimpleNamespace(_heater_width=hw, _x=nh * hw, _n_heaters=nh, _seg_id=idx // hw, _x_pos=idx % hw, _num_envs=2, _ndims=3,
                         _heater_limit=float(limit), _T_hot=float(t_hot), _bottom_plate=plate)
    me._action_to_control = lambda action: RBCEnvBase._action_to_control(me, action)
    me._smooth_profile = lambda T: RBCEnvBase._smooth_profile(me, T)
    RBCEnvBase._apply_action(me, torch.stack([a, -a]))
    assert plate.value.shape == (2, 1, nh * hw, 1, nh * hw)
    assert np.allclose(plate.value[0, 0, :, 0, :].numpy(), G["rbc3d_heater_control"], rtol=1e-6, atol=1e-7)


def test_cylinder_jet_velocities_are_the_reference_s():
    """jet_cylinder_env_2d.py:136-183 evaluated by the reference on its own mesh: the velocity vectors of the two synthetic jets on
    the wall faces of the top / bottom block.  Here: the env's own mesh generator (pinned on the same reference mesh in
    tests/test_cylinder_grid.py) and ``_jet_velocities``."""
    from fluidgym_amd.envs.cylinder import BOTTOM, TOP, CylinderJetEnv2D, _face_vertices
    from fluidgym_amd.envs.cylinder_grid import make_vortex_street_mesh

    for res in (8, 24):
        env = fluidgym_amd.make("CylinderJet2D-easy-v0", cuda_device=torch.device("cpu"), resolution=res)
        env._mesh = make_vortex_street_mesh(env._circle_resolution_angular, env.H, env.L, env.cylinder_diameter / 2, env.cylinder_offset_y,
                                            env.cylinder_diameter / 2, env.cylinder_diameter, env._vortex_street_refinement_base)   # as _get_domain (no GPU)
        top = CylinderJetEnv2D._jet_velocities(env, _face_vertices(env._mesh, TOP, "-y"), True)
        bottom = CylinderJetEnv2D._jet_velocities(env, _face_vertices(env._mesh, BOTTOM, "+y"), False)
        assert np.allclose(top, G[f"cyl_jet_r{res}_top"][0, :, 0, :], atol=2e-6), res
        assert np.allclose(bottom, G[f"cyl_jet_r{res}_bottom"][0, :, 0, :], atol=2e-6), res
        assert np.abs(top).max() > 0.9


def test_airfoil_jet_control_is_the_reference_s():
    """airfoil_env_2d.py:168-190: mean removed, scaled to a maximum of 1 only when it exceeds 1, each jet's stretch of the base
    profile multiplied by its action."""
    from fluidgym_amd.envs.airfoil import AirfoilEnv2D

    base = torch.as_tensor(G["airfoil_base_profile"])[0, :, 0, :]            # [2, nx]
    locs = [tuple(int(v) for v in r) for r in G["airfoil_jet_locations"]]
    me = SimpleNamespace(_top_base_profile=base, _jet_locations_top=locs, _num_envs=2)
    for tag in ("small", "large", "equal"):
        a = torch.as_tensor(G[f"airfoil_action_{tag}"])
        got = AirfoilEnv2D._action_to_control(me, torch.stack([a, 0.1 * a]))
        assert np.allclose(got[0].numpy(), G[f"airfoil_control_{tag}"][0, :, 0, :], atol=1e-6), tag


def test_tcf_wall_stress_is_the_reference_s():
    """tcf_env.py:564-584: nu x plane-mean streamwise velocity of the first / last cell row over its distance from the wall."""
    yc = G["tcf_stress_y_centers"]
    vel = torch.as_tensor(G["tcf_stress_velocity"])
    me = SimpleNamespace(_block=SimpleNamespace(velocity=torch.cat([vel, 3.0 * vel])), _nu=3.1e-4, _d_wall=(float(1.0 + yc[0]), float(1.0 - yc[-1])))
    bottom, top = TCF3DBottomEnv._get_wall_stress(me)
    assert np.allclose([float(bottom[0]), float(top[0])], G["tcf_stress_out"], rtol=2e-6)
    assert np.allclose([float(bottom[1]), float(top[1])], 3.0 * G["tcf_stress_out"], rtol=2e-6)


def test_rbc_local_rewards_are_the_reference_s():
    """rbc_env_2d.py:327-358: nu_ref minus the Nusselt number over each agent's window of the simulation grid (the reference's own
    window extraction and Nusselt formula on the same fields), windows of one and three heaters."""
    T, u, cs = (torch.as_tensor(G[k]) for k in ("rbc_local_T", "rbc_local_u", "rbc_local_cell_size"))
    ra, pr = G["nusselt_ra_pr"]
    for W in (1, 3):
        blk = SimpleNamespace(passiveScalar=torch.cat([T, 0.5 * T]), velocity=torch.cat([u, u]))
        me = SimpleNamespace(_block=blk, _cell_size=cs, _n_heaters=6, _heater_width=4, _local_obs_window=W, _ndims=2, nu_ref=2.5,
                             _rayleigh_number=float(ra), _prandtl_number=float(pr))
        me._local_nusselt = lambda lT, lu, lc, me=me: RBCEnvBase._local_nusselt(me, lT, lu, lc)
        got = RBCEnvBase._get_local_rewards(me)
        assert got.shape == (2, 6)
        assert np.allclose(got[0].numpy(), G[f"rbc_local_rewards_w{W}"], rtol=1e-5, atol=1e-4), W


def test_rbc_3d_local_rewards_are_the_reference_s():
    """rbc_env_3d.py:380-411: the same over x-z windows of the n_heaters x n_heaters agents."""
    T, u, cs = (torch.as_tensor(G[k]) for k in ("rbc3d_local_T", "rbc3d_local_u", "rbc3d_local_cell_size"))
    ra, pr = G["nusselt_ra_pr"]
    for W in (1, 3):
        blk = SimpleNamespace(passiveScalar=torch.cat([T, T]), velocity=torch.cat([u, 2.0 * u]))
        me = SimpleNamespace(_block=blk, _cell_size=cs, _n_heaters=3, _heater_width=2, _local_obs_window=W, _ndims=3, nu_ref=2.5,
                             _rayleigh_number=float(ra), _prandtl_number=float(pr))
        me._local_nusselt = lambda lT, lu, lc, me=me: RBCEnvBase._local_nusselt(me, lT, lu, lc)
        got = RBCEnvBase._get_local_rewards(me)
        assert got.shape == (2, 9)
        assert np.allclose(got[0].numpy(), G[f"rbc3d_local_rewards_w{W}"], rtol=1e-5, atol=1e-4), W


def test_rotating_cylinder_wall_velocities_are_the_reference_s():
    """rotating_cylinder_env_2d.py:131-164 on the reference's own mesh: unit tangential velocity on the wall faces of the left, top,
    right and bottom block, in the order of their face cells."""
    from fluidgym_amd.envs.cylinder import BOTTOM, LEFT, RIGHT, TOP, rotating_wall_velocities
    from fluidgym_amd.envs.cylinder_grid import make_vortex_street_mesh

    for res in (8, 24):
        env = fluidgym_amd.make("CylinderRot2D-easy-v0", cuda_device=torch.device("cpu"), resolution=res)
        mesh = make_vortex_street_mesh(env._circle_resolution_angular, env.H, env.L, env.cylinder_diameter / 2, env.cylinder_offset_y,
                                       env.cylinder_diameter / 2, env.cylinder_diameter, env._vortex_street_refinement_base)
        names = {LEFT: "left", TOP: "top", RIGHT: "right", BOTTOM: "bottom"}
        walls = rotating_wall_velocities(mesh)          # what CylinderRotEnv2D._additional_initialization uploads
        assert [b for b, _, _ in walls] == [LEFT, TOP, RIGHT, BOTTOM]
        for b, face, v in walls:
            assert np.allclose(v, G[f"cyl_rot_r{res}_{names[b]}"], atol=3e-6), (res, names[b])


def test_tcf_both_walls_actuation_is_the_reference_s():
    """tcf_env.py:1143-1154: the first half of the agents drives the bottom wall, the second half the top wall with the
    wall-normal velocity reversed; each half scaled on its own."""
    from fluidgym_amd.envs.tcf import TCF3DBothEnv

    a = torch.as_tensor(G["tcf_both_action"])
    u_wall, actor = G["tcf_u_wall_actor_size"]
    walls = {}

    class Plate:
        def __init__(self, name):
            self.velocity = torch.ones(2, 3, 8, 1, 12)
            walls[name] = self

    me = SimpleNamespace(_num_envs=2, _n_actors_x=6, _n_actors_z=4, _scale_actions=True, _u_wall=float(u_wall), _actor_size=int(actor),
                         _bottom_plate=Plate("bottom"), _top_plate=Plate("top"))
    me._action_to_control = lambda x: TCF3DBottomEnv._action_to_control(me, x)
    me._set_wall = lambda plate, v: TCF3DBottomEnv._set_wall(me, plate, v)
    TCF3DBothEnv._apply_action(me, torch.stack([a, -a]))
    assert np.allclose(walls["bottom"].velocity[0].numpy(), G["tcf_both_bottom"][0], atol=1e-7)
    assert np.allclose(walls["top"].velocity[0].numpy(), G["tcf_both_top"][0], atol=1e-7)
    assert np.allclose(walls["bottom"].velocity[1].numpy(), -G["tcf_both_bottom"][0], atol=1e-7)
