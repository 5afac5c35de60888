"""Golden fixture of the reference's AIRFOIL mesh (not built yet in fluidgym_amd: pins the next round's mesh builder).

Vertex coordinates of the six blocks, boundary / connection calls and boundary velocities recorded from the reference's
own ``envs/airfoil/grid.py::make_airfoil_domain`` running HERE against the recording stand-in for the CUDA-only
``PISOtorch`` module used by ``make_golden_cylinder.py`` (the stand-in only logs construction calls; no reference source is
copied).  Arguments as ``AirfoilEnvBase._get_domain`` passes them (airfoil_env_base.py:209-229).

    python tests/golden/make_golden_airfoil.py  ->  tests/golden/reference_airfoil_grid.npz
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_cylinder as M  # noqa: E402  (recording Domain / Block / Boundary, stub helpers)

REF = M.REF
OUT = M.OUT


def main():
    for pkg in ["fluidgym", "fluidgym.simulation", "fluidgym.simulation.pict", "fluidgym.simulation.pict.data",
                "fluidgym.simulation.pict.util", "fluidgym.envs", "fluidgym.envs.util", "fluidgym.envs.airfoil"]:
        M._stub(pkg).__path__ = []

    class _Dom(M._Domain):
        def getBlocks(self):
            return []

    class _Blk(M._Block):
        def CloseBoundary(self, face, *vel):
            self.log.append(("close", self.idx, face))
            if vel and vel[0] is not None:
                self.log.append(("velocity", self.idx, face, vel[0].detach().cpu().numpy().copy()))

    def create(self, vertexCoordinates=None, name=""):
        b = _Blk(self.log, self.n, vertexCoordinates, name)
        self.n += 1
        return b

    _Dom.CreateBlock = create
    M._stub("fluidgym.simulation.extensions", PISOtorch=types.SimpleNamespace(Domain=_Dom))
    M._stub("fluidgym.simulation.pict.util.output", plot_grids=lambda *a, **k: None)
    shapes = M._load(f"{REF}/fluidgym/simulation/pict/data/shapes.py", "fluidgym.simulation.pict.data.shapes")
    sys.modules["fluidgym.simulation.pict.data"].shapes = shapes
    M._load(f"{REF}/fluidgym/envs/util/profiles.py", "fluidgym.envs.util.profiles")
    M._load(f"{REF}/fluidgym/envs/airfoil/coords.py", "fluidgym.envs.airfoil.coords")
    # the generator ends with balance_boundary_fluxes(domain, out_bounds) (grid.py:714), which needs the compiled backend:
    # recorded as a call, the balancing itself is the product's job
    M._stub("fluidgym.simulation.pict.PISOtorch_simulation",
            balance_boundary_fluxes=lambda domain, bounds: domain.log.append(("balance",) + tuple(f"{b.block}{b.face}" for b in bounds)))
    grid = M._load(f"{REF}/fluidgym/envs/airfoil/grid.py", "ref_airfoil_grid")

    out = {}
    for aoa in (20.0, 0.0):
        dom = grid.make_airfoil_domain(n_dims=2, res_z=0, H=1.4, L=4.5, vel_in=0.3, attack_angle_deg=aoa,
                                       viscosity=torch.tensor([0.3 / 1e3]), resolution_div=1, tail_grow_mul=1.01,
                                       cpu_device=torch.device("cpu"), cuda_device=torch.device("cpu"))
        tag = f"aoa{int(aoa)}"
        calls = []
        for rec in dom.log:
            if rec[0] == "block":
                out[f"{tag}_block{rec[1]}"] = rec[3][0] if rec[3].ndim == 4 else rec[3]
                calls.append(f"block {rec[1]} {rec[2]}")
            elif rec[0] == "velocity":
                out[f"{tag}_velocity_{rec[1]}_{rec[2]}"] = rec[3]
                calls.append(f"velocity {rec[1]} {rec[2]}")
            else:
                calls.append(" ".join(str(x) for x in rec))
        out[f"{tag}_calls"] = np.array(calls)
        print(tag, [(k, v.shape) for k, v in out.items() if k.startswith(tag) and v.dtype.kind != "U"])
        print(calls)
    # ---- env-level geometry of AirfoilEnvBase (airfoil_env_base.py:175-214, 559-660): section mask on the render grid,
    # sensor pixels left after masking, jet cell ranges on the top block -- the reference's own methods run on a bare
    # namespace; the default angle of attack (10 degrees) and the recorded ones
    import ast
    with open(f"{REF}/fluidgym/envs/airfoil/airfoil_env_base.py") as fh:
        tree = ast.parse(fh.read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AirfoilEnvBase")
    want = {"_get_airfoil_mask", "_physical_locations_to_grid_coords", "_get_sensor_locations", "_get_sensor_locations_2d"}
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in want]
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(fns, []), "airfoil_env_base.py", "exec"), ns)
    for aoa in (10.0, 20.0, 0.0):
        me = types.SimpleNamespace(H=1.4, L=4.5, D=1.4, airfoil_length=1.0, _ndims=2, render_shape=(600, 150, 150))
        _, c = grid.read_airfoil(attack_angle_deg=aoa, cpu_device=torch.device("cpu"), dtype=torch.float32)
        me._airfoil_coords = c.squeeze()
        for k in want:
            setattr(me, k, types.MethodType(ns[k], me))
        me._airfoil_mask = me._get_airfoil_mask()
        tag = f"aoa{int(aoa)}"
        out[f"{tag}_mask_rows"] = np.packbits(me._airfoil_mask, axis=1)          # [150, 75] bits of the [150, 600] mask
        out[f"{tag}_sensor_pixels"] = me._get_sensor_locations().numpy()
        print(tag, "mask pixels", int(me._airfoil_mask.sum()), "sensors", out[f"{tag}_sensor_pixels"].shape)
    # 3-D env: sensor pixels (x, y, z) on the n_agents * n_sensors_per_agent spanwise planes (airfoil_env_3d.py:303-344)
    with open(f"{REF}/fluidgym/envs/airfoil/airfoil_env_3d.py") as fh:
        tree3 = ast.parse(fh.read())
    cls3 = next(n for n in tree3.body if isinstance(n, ast.ClassDef) and n.name == "AirfoilEnv3D")
    fns3 = [n for n in cls3.body if isinstance(n, ast.FunctionDef) and n.name in {"_get_sensor_locations", "_get_sensor_locations_3d"}]
    ns3 = {"torch": torch, "np": np}
    exec(compile(ast.Module(fns3, []), "airfoil_env_3d.py", "exec"), ns3)
    me = types.SimpleNamespace(H=1.4, L=4.5, D=1.4, airfoil_length=1.0, _ndims=3, render_shape=(600, 150, 150), _n_sensors_z=4)
    _, c = grid.read_airfoil(attack_angle_deg=10.0, cpu_device=torch.device("cpu"), dtype=torch.float32)
    me._airfoil_coords = c.squeeze()
    for k in ("_get_airfoil_mask", "_physical_locations_to_grid_coords", "_get_sensor_locations_2d"):
        setattr(me, k, types.MethodType(ns[k], me))
    me._airfoil_mask = me._get_airfoil_mask()          # [150, 150, 600] in 3-D
    me._get_sensor_locations_3d = types.MethodType(ns3["_get_sensor_locations_3d"], me)
    out["aoa10_3d_sensor_pixels"] = ns3["_get_sensor_locations"](me).numpy()
    print("3d sensors", out["aoa10_3d_sensor_pixels"].shape)
    for aoa in (20.0, 0.0):
        tag = f"aoa{int(aoa)}"
        coords = [torch.from_numpy(out[f"{tag}_block{b}"])[None] for b in range(6)]
        dom = types.SimpleNamespace(getVertexCoordinates=lambda: coords, getSpatialDims=lambda: 2)
        out[f"{tag}_jet_locations"] = np.asarray(grid.get_jet_locations(dom), np.int32)
        print(tag, "jets", out[f"{tag}_jet_locations"].tolist())
    np.savez_compressed(os.path.join(OUT, "reference_airfoil_grid.npz"), **out)


if __name__ == "__main__":
    main()
