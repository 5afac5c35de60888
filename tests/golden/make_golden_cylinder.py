"""Golden fixture of the reference's cylinder (vortex street) mesh: vertex coordinates of the five blocks, the
connection / boundary calls and the inflow profile, recorded from the reference's own
``envs/cylinder/grid.py::make_vortex_street_domain`` running HERE against a recording stand-in for the CUDA-only
``PISOtorch`` module (the stand-in only logs the construction calls; no reference source is copied).

    python tests/golden/make_golden_cylinder.py

Pins ``fluidgym_amd/envs/cylinder_grid.py`` (tests/test_cylinder_grid.py).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


class _Bound:
    def __init__(self, log, block, face):
        self.log, self.block, self.face = log, block, face

    def setVelocity(self, v):
        self.log.append(("velocity", self.block, self.face, v.detach().cpu().numpy().copy()))

    def makeVelocityVarying(self):
        self.log.append(("varying", self.block, self.face))


class _Block:
    def __init__(self, log, idx, coords, name):
        self.log, self.idx, self.name = log, idx, name
        log.append(("block", idx, name, coords.detach().cpu().numpy().copy()))

    def CloseBoundary(self, face):
        self.log.append(("close", self.idx, face))

    def getBoundary(self, face):
        return _Bound(self.log, self.idx, face)

    def ConnectBlock(self, face, other, other_face, *axes):
        self.log.append(("connect", self.idx, face, other.idx, other_face) + tuple(axes))

    def MakePeriodic(self, axis):
        self.log.append(("periodic", self.idx, axis))


class _Domain:
    def __init__(self, ndims, viscosity, name="", device=None, dtype=None, passiveScalarChannels=0):
        self.log = []
        self.n = 0

    def CreateBlock(self, vertexCoordinates=None, name=""):
        b = _Block(self.log, self.n, vertexCoordinates, name)
        self.n += 1
        return b


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    # package skeleton so that the reference file's absolute imports resolve without running fluidgym/__init__
    for pkg in ["fluidgym", "fluidgym.simulation", "fluidgym.simulation.pict", "fluidgym.simulation.pict.data",
                "fluidgym.simulation.pict.util", "fluidgym.envs", "fluidgym.envs.util"]:
        _stub(pkg).__path__ = []
    _stub("fluidgym.simulation.extensions", PISOtorch=types.SimpleNamespace(Domain=_Domain))
    _stub("fluidgym.simulation.pict.util.output", plot_grids=lambda *a, **k: None)
    shapes = _load(f"{REF}/fluidgym/simulation/pict/data/shapes.py", "fluidgym.simulation.pict.data.shapes")
    sys.modules["fluidgym.simulation.pict.data"].shapes = shapes
    _load(f"{REF}/fluidgym/envs/util/profiles.py", "fluidgym.envs.util.profiles")
    grid = _load(f"{REF}/fluidgym/envs/cylinder/grid.py", "ref_cylinder_grid")

    out = {}
    for res in (8, 24):
        # arguments of CylinderEnvBase._get_domain (envs/cylinder/cylinder_env_base.py:233-252)
        dom = grid.make_vortex_street_domain(
            ndims=2, viscosity=torch.tensor([0.01]), domain_height=4.1, domain_length=22.0, cylinder_radius=0.5,
            cylinder_offset_y=0.05, circle_thickness=0.5, quad_thickness_x=1.0, circle_resolution_angular=res,
            vortex_street_refinement_base=0.95, vortex_street_refinement_axes=["+y", "-y"],
            cuda_device=torch.device("cpu"), dtype=torch.float32)
        calls = []
        for rec in dom.log:
            if rec[0] == "block":
                out[f"r{res}_block{rec[1]}"] = rec[3][0]
                calls.append(f"block {rec[1]} {rec[2]}")
            elif rec[0] == "velocity":
                out[f"r{res}_velocity_{rec[1]}_{rec[2]}"] = rec[3]
                calls.append(f"velocity {rec[1]} {rec[2]}")
            else:
                calls.append(" ".join(str(x) for x in rec))
        out[f"r{res}_calls"] = np.array(calls)
    # the 3-D variant (extruded along z, z-periodic; grid.py:281-294, 330-343, 395-416) at resolution 8
    dom = grid.make_vortex_street_domain(
        ndims=3, viscosity=torch.tensor([0.01]), domain_height=4.1, domain_length=22.0, cylinder_radius=0.5,
        cylinder_offset_y=0.05, circle_thickness=0.5, quad_thickness_x=1.0, circle_resolution_angular=8,
        vortex_street_refinement_base=0.95, vortex_street_refinement_axes=["+y", "-y"],
        cuda_device=torch.device("cpu"), dtype=torch.float32)
    calls = []
    for rec in dom.log:
        if rec[0] == "block":
            out[f"r8_3d_block{rec[1]}"] = rec[3][0]
            calls.append(f"block {rec[1]} {rec[2]}")
        elif rec[0] == "velocity":
            out[f"r8_3d_velocity_{rec[1]}_{rec[2]}"] = rec[3]
            calls.append(f"velocity {rec[1]} {rec[2]}")
        else:
            calls.append(" ".join(str(x) for x in rec))
    out["r8_3d_calls"] = np.array(calls)

    # sensor pixels of CylinderEnvBase (cylinder_env_base.py:430-518): the methods are compiled from the reference file and
    # called on a bare namespace carrying the class attributes they read
    import ast
    with open(f"{REF}/fluidgym/envs/cylinder/cylinder_env_base.py") as fh:
        tree = ast.parse(fh.read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CylinderEnvBase")
    want = {"_get_sensor_locations_2d", "_sensor_locations_to_grid_coords", "_get_sensor_locations"}
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in want]
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(fns, []), "cylinder_env_base.py", "exec"), ns)
    for res in (8, 24, 32):
        z = res * 4
        me = types.SimpleNamespace(H=4.1, L=22.0, cylinder_diameter=1.0, _ndims=2, render_shape=(int(z / 4.1 * 22.0), z, z))
        for k in want:
            setattr(me, k, types.MethodType(ns[k], me))
        out[f"r{res}_sensor_pixels"] = me._get_sensor_locations().numpy()
    # ---- 3-D jet env (jet_cylinder_env_3d.py): sensor pixels, the layout of the global / local observations and the jet
    # wall velocities, from the reference's own methods run on a bare namespace; the resampled fields the observation
    # code reads are replaced by index ramps (value = flat index), so only the expected outputs need storing
    with open(f"{REF}/fluidgym/envs/cylinder/jet_cylinder_env_3d.py") as fh:
        tree3 = ast.parse(fh.read())
    cls3 = next(n for n in tree3.body if isinstance(n, ast.ClassDef) and n.name == "CylinderJetEnv3D")
    want3 = {"_get_sensor_locations", "_get_sensor_locations_3d", "_get_local_obs", "_CylinderJetEnv3D__get_boundary_velocities"}
    fns3 = [n for n in cls3.body if isinstance(n, ast.FunctionDef) and (n.name in want3 or n.name == "__get_boundary_velocities")]
    with open(f"{REF}/fluidgym/envs/util/obs_extraction.py") as fh:
        tree_o = ast.parse(fh.read())
    fn_o = [n for n in tree_o.body if isinstance(n, ast.FunctionDef) and n.name == "extract_global_3d_obs"]
    with open(f"{REF}/fluidgym/envs/util/profiles.py") as fh:
        tree_p = ast.parse(fh.read())
    fn_p = [n for n in tree_p.body if isinstance(n, ast.FunctionDef) and n.name == "get_jet_profile"]
    ns3 = {"torch": torch, "np": np}
    exec(compile(ast.Module(fn_p, []), "profiles.py", "exec"), ns3)
    exec(compile(ast.Module(fns3, []), "jet_cylinder_env_3d.py", "exec"), ns3)
    for res, n_jets in ((8, 4), (24, 8)):
        z = res * 4
        rs = (int(z / 4.1 * 22.0), z, z)
        me = types.SimpleNamespace(H=4.1, L=22.0, cylinder_diameter=1.0, _ndims=3, render_shape=rs, _n_jets=n_jets,
                                   _n_sensors_per_agent=2, _local_obs_window=3, _local_2d_obs=False)
        me._n_sensors_z = n_jets * 2
        for k in want:
            setattr(me, k, types.MethodType(ns[k], me))
        me._get_sensor_locations_2d = types.MethodType(ns["_get_sensor_locations_2d"], me)
        me._sensor_locations_to_grid_coords = types.MethodType(ns["_sensor_locations_to_grid_coords"], me)
        me._get_sensor_locations_3d = types.MethodType(ns3["_get_sensor_locations_3d"], me)
        loc = ns3["_get_sensor_locations"](me)
        out[f"r{res}_3d_sensor_pixels"] = loc.numpy()
        if res != 8:
            continue
        # observation layout
        n_u = 3 * rs[2] * rs[1] * rs[0]
        u_field = torch.arange(n_u, dtype=torch.float32).reshape(1, 3, rs[2], rs[1], rs[0])
        p_field = -torch.arange(rs[2] * rs[1] * rs[0], dtype=torch.float32).reshape(1, 1, rs[2], rs[1], rs[0])
        ns_o = {"torch": torch, "FluidEnv": object,
                "_resample_block_data": lambda data, *a, **k: (u_field if data[0] == "u" else p_field)}
        exec(compile(ast.Module(fn_o, []), "obs_extraction.py", "exec"), ns_o)
        blk = types.SimpleNamespace(velocity="u", pressure="p")
        env = types.SimpleNamespace(_domain=types.SimpleNamespace(getBlocks=lambda: [blk]), _ndims=3, _differentiable=False,
                                    _sim=types.SimpleNamespace(output_resampling_coords=None, output_resampling_shape=None,
                                                               output_resampling_fill_max_steps=16))
        g_obs = ns_o["extract_global_3d_obs"](env=env, sensor_locations=loc, n_agents=n_jets, n_sensors_per_agent=2,
                                              n_sensors_z=n_jets * 2, local_2d_obs=False)
        out["r8_3d_obs_global_velocity"] = g_obs["velocity"].numpy()
        out["r8_3d_obs_global_pressure"] = g_obs["pressure"].numpy()
        me._get_global_obs = lambda: g_obs
        l_obs = ns3["_get_local_obs"](me)
        out["r8_3d_obs_local_velocity"] = l_obs["velocity"].numpy()
        out["r8_3d_obs_local_pressure"] = l_obs["pressure"].numpy()
        # jet wall velocities of the top / bottom block from the recorded 3-D mesh
        coords = [torch.from_numpy(out[f"r8_3d_block{b}"])[None] for b in range(5)]
        me._domain = types.SimpleNamespace(getVertexCoordinates=lambda: coords)
        me._top_block_idx, me._bottom_block_idx = 1, 3
        me._jet_angle, me._dtype, me._cuda_device = 10.0, torch.float32, torch.device("cpu")
        fn = ns3.get("_CylinderJetEnv3D__get_boundary_velocities") or ns3["__get_boundary_velocities"]
        top, bottom, nz_per_agent = fn(me)
        out["r8_3d_jet_top"] = top.numpy()
        out["r8_3d_jet_bottom"] = bottom.numpy()
        out["r8_3d_nz_per_agent"] = np.asarray(nz_per_agent)
    np.savez_compressed(os.path.join(OUT, "reference_cylinder_grid.npz"), **out)
    for k, v in out.items():
        print(k, v.shape if v.dtype.kind != "U" else list(v))


if __name__ == "__main__":
    main()
