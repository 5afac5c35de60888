"""Small pieces of arithmetic inside the reference's env classes, evaluated HERE by the reference's own code: the method is taken out
of its source file with ``ast``, compiled as it stands and called with a stand-in ``self`` that carries the attributes it reads.
Only inputs and outputs are stored (``tests/golden/reference_env_math.npz``); ``tests/test_env_math_golden.py`` holds the env classes
of ``fluidgym_amd`` against them.

* ``RBCEnvBase._compute_nusselt`` (``envs/rbc/rbc_env_base.py:491-513``): volume-weighted convective heat flux -> Nusselt number;
* ``TCF3DBottomEnv._action_to_control`` (``envs/tcf/tcf_env.py:521-547``): actions -> wall-normal velocity of the actuated wall
  (zero net mass flux, clipped at u_tau, actor patches), with and without ``scale_actions``;
* ``RBCEnv2D.__action_to_control`` / ``__smooth_action_profile_1d`` (``envs/rbc/rbc_env_2d.py:207-262``): heater actions ->
  bottom-plate temperature profile (zero mean, limit, cubic blending between neighbouring heaters);
* ``RBCEnvBase._randomize_domain`` (``:335-398``) and ``TCF3DBottomEnv._get_global_obs`` / ``_get_local_obs`` (``:646-678, 918-992``);
* ``TCF3DBottomEnv._t_to_t_wall`` / ``_t_wall_to_t`` (``:323-327``) with the reference's ``TCF_tools.t_star``.

    python tests/golden/make_golden_env_math.py
"""
import ast
import os
import types

import numpy as np
import torch

REF = "/root/reference/src/fluidgym"
OUT = os.path.dirname(os.path.abspath(__file__))


def method(path, cls, name, extra=None):
    tree = ast.parse(open(path).read())
    c = next(n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == cls)
    fn = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.decorator_list = []
    fn.returns = None
    for a in fn.args.args + fn.args.kwonlyargs:
        a.annotation = None
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch, "np": np}
    ns.update(extra or {})
    exec(compile(ast.fix_missing_locations(mod), path, "exec"), ns)
    return ns[name]


def function(path, name, extra=None):
    tree = ast.parse(open(path).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.returns = None
    for a in fn.args.args:
        a.annotation = None
    ns = {"torch": torch, "np": np}
    ns.update(extra or {})
    exec(compile(ast.fix_missing_locations(ast.Module(body=[fn], type_ignores=[])), path, "exec"), ns)
    return ns[name]


def main():
    rng = np.random.default_rng(0)
    out = {}
    # ---- RBC Nusselt number
    nusselt = method(f"{REF}/envs/rbc/rbc_env_base.py", "RBCEnvBase", "_compute_nusselt")
    for tag, ndims, shape, batched in (("2d", 2, (6, 10), False), ("2d_batched", 2, (3, 6, 10), True), ("3d_batched", 3, (2, 4, 5, 6), True)):
        me = types.SimpleNamespace(_ndims=ndims, _rayleigh_number=8e4, _prandtl_number=0.7)
        T = torch.as_tensor(rng.random(shape), dtype=torch.float32)
        uy = torch.as_tensor(rng.standard_normal(shape), dtype=torch.float32)
        cs = torch.as_tensor(0.5 + rng.random(shape[1:] if batched else shape), dtype=torch.float32)
        out[f"nusselt_{tag}_T"], out[f"nusselt_{tag}_uy"], out[f"nusselt_{tag}_cell_size"] = T.numpy(), uy.numpy(), cs.numpy()
        out[f"nusselt_{tag}_out"] = np.asarray(nusselt(me, T, uy, cs))
    out["nusselt_ra_pr"] = np.array([8e4, 0.7])
    # ---- TCF wall actuation
    to_control = method(f"{REF}/envs/tcf/tcf_env.py", "TCF3DBottomEnv", "_action_to_control")
    for tag, scale in (("scaled", True), ("raw", False)):
        me = types.SimpleNamespace(_scale_actions=scale, _u_wall=0.0557, _z=8, _x=12, _actor_size=2, _cuda_device=torch.device("cpu"))
        a = torch.as_tensor(2.5 * rng.standard_normal((6, 4)), dtype=torch.float32)          # [n_actors_x, n_actors_z], some |a| > 1
        out[f"tcf_action_{tag}"] = a.numpy()
        out[f"tcf_control_{tag}"] = to_control(me, a).numpy()                                 # [1, 3, Z, 1, X]
    out["tcf_u_wall_actor_size"] = np.array([0.0557, 2])
    # ---- TCF, both walls actuated: first half of the agents drives the bottom wall, second half the top wall with the sign of the
    # wall-normal velocity reversed (tcf_env.py:1143-1154)
    both = method(f"{REF}/envs/tcf/tcf_env.py", "TCF3DBothEnv", "_apply_action")
    sent = {}
    me = types.SimpleNamespace(n_agents=2 * 6 * 4, _n_actors_x=6, _n_actors_z=4, _scale_actions=True, _u_wall=0.0557, _z=8, _x=12, _actor_size=2,
                               _cuda_device=torch.device("cpu"), _bottom_plate=types.SimpleNamespace(setVelocity=lambda v: sent.__setitem__("bottom", v)),
                               _top_plate=types.SimpleNamespace(setVelocity=lambda v: sent.__setitem__("top", v)))
    me._action_to_control = lambda a: to_control(me, a)
    a_both = torch.as_tensor(2.0 * rng.standard_normal(48), dtype=torch.float32)
    both(me, a_both)
    out["tcf_both_action"], out["tcf_both_bottom"], out["tcf_both_top"] = a_both.numpy(), sent["bottom"].numpy(), sent["top"].numpy()
    # ---- TCF wall shear stress of both walls from the plane-mean streamwise velocity of the first / last cell row (tcf_env.py:564-584)
    stress = method(f"{REF}/envs/tcf/tcf_env.py", "TCF3DBottomEnv", "_get_wall_stress")
    Zs, Ys, Xs = 4, 7, 5
    yc = np.sort(rng.uniform(-0.98, 0.98, Ys)).astype(np.float32)                      # cell-centre heights between the walls at -1 / +1
    cc = torch.as_tensor(np.broadcast_to(yc[None, :, None], (Zs, Ys, Xs)).copy())
    vel = torch.as_tensor(rng.random((1, 3, Zs, Ys, Xs)), dtype=torch.float32)
    blk = types.SimpleNamespace(getCellCoordinates=lambda: torch.stack([torch.zeros_like(cc), cc, torch.zeros_like(cc)])[None], velocity=vel)
    me = types.SimpleNamespace(_domain=types.SimpleNamespace(getBlock=lambda i: blk, viscosity=torch.tensor([3.1e-4]), getDevice=lambda: torch.device("cpu")))
    tb, tt = stress(me)
    out["tcf_stress_y_centers"], out["tcf_stress_velocity"], out["tcf_stress_out"] = yc, vel.numpy(), np.array([float(tb), float(tt)])
    # ---- TCF time units
    t_star = function(f"{REF}/simulation/pict/data/TCF_tools.py", "t_star")
    tools = types.SimpleNamespace(t_star=t_star)
    to_wall = method(f"{REF}/envs/tcf/tcf_env.py", "TCF3DBottomEnv", "_t_to_t_wall", {"TCF_tools": tools})
    from_wall = method(f"{REF}/envs/tcf/tcf_env.py", "TCF3DBottomEnv", "_t_wall_to_t", {"TCF_tools": tools})
    me = types.SimpleNamespace(_viscosity=torch.tensor(3.1e-4), _u_wall=0.0557)
    out["tcf_time_units"] = np.array([3.1e-4, 0.0557, to_wall(me, 0.37), from_wall(me, 0.6)])
    # ---- RBC heaters: actions -> bottom-plate temperature profile (rbc_env_2d.py:207-262; private names are not mangled
    # outside a class body, so the stand-in carries them literally)
    smooth = method(f"{REF}/envs/rbc/rbc_env_2d.py", "RBCEnv2D", "__smooth_action_profile_1d")
    control = method(f"{REF}/envs/rbc/rbc_env_2d.py", "RBCEnv2D", "__action_to_control")
    for tag, n_heaters, hw in (("w8", 12, 8), ("w20", 6, 20), ("w4", 8, 4)):      # blend zones of 1, 2 and 0 cells
        me = types.SimpleNamespace(_heater_width=hw, _x=n_heaters * hw, _heater_limit=0.75, _T_hot=1.0)
        setattr(me, "__smooth_action_profile_1d", lambda T_action, me=me: smooth(me, T_action))
        a = torch.as_tensor(1.5 * rng.standard_normal(n_heaters), dtype=torch.float32)
        out[f"rbc_heater_{tag}_action"] = a.numpy()
        out[f"rbc_heater_{tag}_control"] = control(me, a.clone()).numpy()
    out["rbc_heater_limit_T_hot"] = np.array([0.75, 1.0])
    # ---- cylinder jets: velocity vectors of the two synthetic jets on the wall faces of the top / bottom block
    # (jet_cylinder_env_2d.py:136-183), on the reference's own mesh (tests/golden/reference_cylinder_grid.npz, made by its generator)
    jet_profile = function(f"{REF}/envs/util/profiles.py", "get_jet_profile")
    jets = method(f"{REF}/envs/cylinder/jet_cylinder_env_2d.py", "CylinderJetEnv2D", "_get_boundary_velocities", {"get_jet_profile": jet_profile})
    mesh = np.load(os.path.join(OUT, "reference_cylinder_grid.npz"))
    for res in (8, 24):
        coords = [torch.as_tensor(mesh[f"r{res}_block{b}"], dtype=torch.float32)[None] for b in range(5)]
        me = types.SimpleNamespace(_domain=types.SimpleNamespace(getVertexCoordinates=lambda: coords), _top_block_idx=1, _bottom_block_idx=3,
                                   _jet_angle=10.0, _dtype=torch.float32, _cuda_device=torch.device("cpu"))
        top, bottom = jets(me)
        out[f"cyl_jet_r{res}_top"], out[f"cyl_jet_r{res}_bottom"] = top.numpy(), bottom.numpy()          # [1, 2, 1, n_faces]
    # ---- airfoil jets: actions -> wall velocity of the top face from the base profiles (airfoil_env_2d.py:168-190)
    af = method(f"{REF}/envs/airfoil/airfoil_env_2d.py", "AirfoilEnv2D", "_action_to_control")
    base = torch.as_tensor(rng.standard_normal((1, 2, 1, 30)), dtype=torch.float32)
    locs = [(3, 6), (12, 16), (22, 25)]
    me = types.SimpleNamespace(_jet_locations_top=locs, _n_jets=3, _top_base_profile=base)
    out["airfoil_base_profile"], out["airfoil_jet_locations"] = base.numpy(), np.array(locs)
    for tag, a in (("small", [0.3, -0.2, 0.5]), ("large", [2.0, -1.0, 0.5]), ("equal", [0.7, 0.7, 0.7])):
        act = torch.tensor(a, dtype=torch.float32)
        out[f"airfoil_action_{tag}"], out[f"airfoil_control_{tag}"] = act.numpy(), af(me, act).numpy()
    # ---- RBC local rewards: nu_ref - Nusselt number over each agent's window of the simulation grid (rbc_env_2d.py:327-358)
    import torch.nn.functional as F

    win2d = function(f"{REF}/envs/util/obs_extraction.py", "extract_moving_window_2d", {"F": F})
    local_r = method(f"{REF}/envs/rbc/rbc_env_2d.py", "RBCEnv2D", "_get_local_rewards",
                     {"extract_moving_window_2d": win2d, "get_cell_size": lambda block: block.cell_size})
    nh, hw, ny = 6, 4, 5
    Tl = torch.as_tensor(rng.random((1, 1, ny, nh * hw)), dtype=torch.float32)
    ul = torch.as_tensor(rng.standard_normal((1, 2, ny, nh * hw)), dtype=torch.float32)
    csl = torch.as_tensor(0.5 + rng.random((ny, 1)) * np.ones((1, nh * hw)), dtype=torch.float32)     # cell sizes vary with y only (as on the env's grids)
    out["rbc_local_T"], out["rbc_local_u"], out["rbc_local_cell_size"] = Tl.numpy(), ul.numpy(), csl.numpy()
    for W in (1, 3):
        me = types.SimpleNamespace(_block=types.SimpleNamespace(passiveScalar=Tl, getVelocity=lambda with_bounds: ul, cell_size=csl[None, None]),
                                   _local_obs_window=W, _heater_width=hw, n_agents=nh, nu_ref=2.5, _ndims=2, _rayleigh_number=8e4, _prandtl_number=0.7)
        me._compute_nusselt = lambda T, u_y, cell_size, me=me: nusselt(me, T, u_y, cell_size)
        out[f"rbc_local_rewards_w{W}"] = local_r(me).numpy()
    # ---- the same in 3-D (rbc_env_3d.py:380-411): n_heaters x n_heaters agents, windows in x and z
    win3d = function(f"{REF}/envs/util/obs_extraction.py", "extract_moving_window_3d", {"F": F})
    local_r3 = method(f"{REF}/envs/rbc/rbc_env_3d.py", "RBCEnv3D", "_get_local_rewards",
                      {"extract_moving_window_3d": win3d, "get_cell_size": lambda block: block.cell_size})
    nh3, hw3, ny3 = 3, 2, 4
    T3 = torch.as_tensor(rng.random((1, 1, nh3 * hw3, ny3, nh3 * hw3)), dtype=torch.float32)
    u3 = torch.as_tensor(rng.standard_normal((1, 3, nh3 * hw3, ny3, nh3 * hw3)), dtype=torch.float32)
    cs3 = torch.as_tensor((0.5 + rng.random((1, ny3, 1))) * np.ones((nh3 * hw3, 1, nh3 * hw3)), dtype=torch.float32)
    out["rbc3d_local_T"], out["rbc3d_local_u"], out["rbc3d_local_cell_size"] = T3.numpy(), u3.numpy(), cs3.numpy()
    for W in (1, 3):
        me = types.SimpleNamespace(_block=types.SimpleNamespace(passiveScalar=T3, getVelocity=lambda with_bounds: u3, cell_size=cs3[None, None]),
                                   _local_obs_window=W, _heater_width=hw3, _n_heaters=nh3, nu_ref=2.5, _ndims=3, _rayleigh_number=8e4,
                                   _prandtl_number=0.7)
        me._compute_nusselt = lambda T, u_y, cell_size, me=me: nusselt(me, T, u_y, cell_size)
        out[f"rbc3d_local_rewards_w{W}"] = local_r3(me).numpy()
    # ---- rotating cylinder: unit tangential velocity on the wall faces of the four blocks around it (rotating_cylinder_env_2d.py:131-164)
    rot = method(f"{REF}/envs/cylinder/rotating_cylinder_env_2d.py", "CylinderRotEnv2D", "_get_boundary_velocities")
    for res in (8, 24):
        coords = [torch.as_tensor(mesh[f"r{res}_block{b}"], dtype=torch.float32)[None] for b in range(5)]
        me = types.SimpleNamespace(_domain=types.SimpleNamespace(getVertexCoordinates=lambda: coords), _left_block_idx=0, _top_block_idx=1,
                                   _right_block_idx=2, _bottom_block_idx=3)
        left, top, bottom, right = rot(me)
        for nm, v in (("left", left), ("top", top), ("bottom", bottom), ("right", right)):
            out[f"cyl_rot_r{res}_{nm}"] = v.numpy().reshape(2, -1)
    # ---- RBC 3-D heaters: [n_heaters, n_heaters] actions -> temperature of the bottom plate [Z, X] (rbc_env_3d.py:201-262)
    s1 = method(f"{REF}/envs/rbc/rbc_env_3d.py", "RBCEnv3D", "__smooth_action_profile_1d")
    s2 = method(f"{REF}/envs/rbc/rbc_env_3d.py", "RBCEnv3D", "__smooth_action_profile_2d")
    c3 = method(f"{REF}/envs/rbc/rbc_env_3d.py", "RBCEnv3D", "_action_to_control")
    me = types.SimpleNamespace(_heater_width=10, _x=40, _n_heaters=4, _heater_limit=0.75, _T_hot=1.0)
    setattr(me, "__smooth_action_profile_1d", lambda T_action: s1(me, T_action))
    setattr(me, "__smooth_action_profile_2d", lambda T_action: s2(me, T_action))
    a = torch.as_tensor(1.5 * rng.standard_normal(16), dtype=torch.float32)
    out["rbc3d_heater_action"], out["rbc3d_heater_control"] = a.numpy(), c3(me, a.clone()).numpy()
    # ---- RBC randomisation of the initial state (rbc_env_base.py:335-398): flips, shifts, noise, 1-2 time units of simulation
    randomize = method(f"{REF}/envs/rbc/rbc_env_base.py", "RBCEnvBase", "_randomize_domain")
    for tag, ndims, shape in (("2d", 2, (5, 8)), ("3d", 3, (8, 5, 8))):
        for seed in (3, 4, 6):
            T0 = torch.as_tensor(rng.random((1, 1) + shape), dtype=torch.float32)
            u0 = torch.as_tensor(0.2 * rng.standard_normal((1, ndims) + shape), dtype=torch.float32)
            state, steps = {"T": T0.clone(), "u": u0.clone()}, []
            block = types.SimpleNamespace(passiveScalar=state["T"], getVelocity=lambda with_bounds: state["u"],
                                          setPassiveScalar=lambda v: state.__setitem__("T", v), setVelocity=lambda v: state.__setitem__("u", v))
            me = types.SimpleNamespace(_block=block, _ndims=ndims, _x=8, _np_rng=np.random.default_rng(seed), _cuda_device=torch.device("cpu"),
                                       _torch_rng_cuda=torch.Generator().manual_seed(seed), _dtype=torch.float32, _T_cold=0.0, _T_hot=1.0,
                                       _dt=0.05, _sim=types.SimpleNamespace(single_step=lambda: steps.append(1)))
            randomize(me)
            out[f"rbc_rand_{tag}_{seed}_T0"], out[f"rbc_rand_{tag}_{seed}_u0"] = T0.numpy(), u0.numpy()
            out[f"rbc_rand_{tag}_{seed}_T"], out[f"rbc_rand_{tag}_{seed}_u"] = state["T"].numpy(), state["u"].numpy()
            out[f"rbc_rand_{tag}_{seed}_steps"] = np.array(len(steps))
    # ---- TCF observations (tcf_env.py:646-678 global: volume-mean fluctuation on the sensing plane; :918-992 local: per-actuator
    # windows, plane-mean fluctuation, the top wall's view flipped) with the reference's own window extraction
    window = function(f"{REF}/envs/util/obs_extraction.py", "extract_moving_window_2d_x_z")
    glob = method(f"{REF}/envs/tcf/tcf_env.py", "TCF3DBottomEnv", "_get_global_obs", {"get_cell_size": lambda block: block.cell_size})
    loc = method(f"{REF}/envs/tcf/tcf_env.py", "TCF3DBottomEnv", "_get_local_obs", {"extract_moving_window_2d_x_z": window})
    Z, Y, X, actor = 8, 6, 12, 2
    u = torch.as_tensor(rng.standard_normal((1, 3, Z, Y, X)), dtype=torch.float32)
    pr = torch.as_tensor(rng.standard_normal((1, 1, Z, Y, X)), dtype=torch.float32)
    cs = torch.as_tensor(0.5 + rng.random((1, 1, Z, Y, X)), dtype=torch.float32)
    block = types.SimpleNamespace(getVelocity=lambda with_bounds: u, pressure=pr, cell_size=cs)
    out["tcf_obs_u"], out["tcf_obs_p"], out["tcf_obs_cell_size"] = u.numpy(), pr.numpy(), cs.numpy()
    for W in (1, 3):
        me = types.SimpleNamespace(_domain=types.SimpleNamespace(getBlock=lambda i: block), _y_obs_bottom_idx=1, _n_actors_x=X // actor,
                                   _n_actors_z=Z // actor, _actor_size=actor, _local_obs_window=W)
        g = glob(me)
        out["tcf_global_velocity"], out["tcf_global_pressure"] = g["velocity"].numpy(), g["pressure"].numpy()
        for flip, y_idx in ((False, 1), (True, Y - 2)):
            l = loc(me, y_idx, flip)
            out[f"tcf_local_w{W}_flip{int(flip)}_velocity"], out[f"tcf_local_w{W}_flip{int(flip)}_pressure"] = l["velocity"].numpy(), l["pressure"].numpy()
    np.savez(os.path.join(OUT, "reference_env_math.npz"), **out)
    print({k: np.asarray(v).shape for k, v in out.items()})
    print(out["tcf_time_units"], out["nusselt_2d_out"], out["nusselt_3d_batched_out"])


if __name__ == "__main__":
    main()
