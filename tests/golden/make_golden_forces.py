"""Golden vectors of the reference's force integration on the cylinder wall (envs/util/forces.py:
wall_distance_from_vertices, compute_forces_2d), generated HERE from the reference's own functions (the CUDA-only
``PISOtorch`` import is satisfied by an empty stand-in module; nothing of the reference is copied).

    python tests/golden/make_golden_forces.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    for pkg in ["fluidgym", "fluidgym.simulation"]:
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    ext = types.ModuleType("fluidgym.simulation.extensions")
    ext.PISOtorch = types.SimpleNamespace(Domain=object)
    sys.modules["fluidgym.simulation.extensions"] = ext
    spec = importlib.util.spec_from_file_location("ref_forces", f"{REF}/fluidgym/envs/util/forces.py")
    forces = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(forces)

    g = torch.Generator().manual_seed(7)
    n = 32
    # a closed, slightly irregular ring of wall vertices (clockwise like the cylinder blocks' ordering) and cell
    # centres half a cell outside of it
    th = -torch.linspace(0, 2 * np.pi, n + 1, dtype=torch.float64) + 0.02 * torch.randn(n + 1, generator=g, dtype=torch.float64)
    th[-1] = th[0] - 2 * np.pi
    vc = 0.5 * torch.stack([torch.cos(th), torch.sin(th)])
    thc = 0.5 * (th[1:] + th[:-1])
    centers = 0.56 * torch.stack([torch.cos(thc), torch.sin(thc)])
    d, nrm = forces.wall_distance_from_vertices(vc.clone(), centers.clone())
    left, right = torch.roll(centers, -1, -1), torch.roll(centers, 1, -1)
    tangent_lengths = torch.sqrt(torch.sum((left - right) ** 2, dim=0))
    face_len = torch.sqrt((vc[0, 1:] - vc[0, :-1]) ** 2 + (vc[1, 1:] - vc[1, :-1]) ** 2)
    u_cell = torch.randn(2, n, generator=g, dtype=torch.float64)
    u_b = 0.3 * torch.randn(2, n, generator=g, dtype=torch.float64)
    p = torch.randn(n, generator=g, dtype=torch.float64)
    nu = torch.tensor([0.01], dtype=torch.float64)
    f = forces.compute_forces_2d(u_cell=u_cell, u_boundary=u_b, p_cell=p, wall_normals=nrm, tangent_lengths=tangent_lengths,
                                 wall_distances=d, wall_face_lengths=face_len, viscosity=nu)
    # 3-D: the same ring extruded over nz spanwise layers (compute_forces_3d, shapes as CylinderEnvBase hands them over:
    # u_cell / u_boundary [3, NZ, N], p_cell [NZ, N], normals [2, N, 1], distances / tangent lengths [N, 1], areas [N])
    nz = 5
    u3 = torch.randn(3, nz, n, generator=g, dtype=torch.float64)
    ub3 = 0.3 * torch.randn(3, nz, n, generator=g, dtype=torch.float64)
    p3 = torch.randn(nz, n, generator=g, dtype=torch.float64)
    areas = face_len * (4.0 / nz)
    f3 = forces.compute_forces_3d(u_cell=u3, u_boundary=ub3, p_cell=p3, wall_normals=nrm[..., None],
                                  tangent_lengths=tangent_lengths[:, None], wall_distances=d[..., None],
                                  wall_face_areas=areas, viscosity=nu)
    print("force 3d", f3)
    np.savez_compressed(os.path.join(OUT, "reference_forces.npz"), u3=u3.numpy(), ub3=ub3.numpy(), p3=p3.numpy(),
                        areas=areas.numpy(), force3=f3.numpy(), vc=vc.numpy(), centers=centers.numpy(), dist=d.numpy(),
                        normals=nrm.numpy(), tangent_lengths=tangent_lengths.numpy(), face_len=face_len.numpy(),
                        u_cell=u_cell.numpy(), u_b=u_b.numpy(), p=p.numpy(), nu=nu.numpy(), force=f.numpy())
    print("force", f)


if __name__ == "__main__":
    main()
