"""The vortex-street (cylinder) mesh of the reference as five connected curvilinear blocks.

Own construction of what ``envs/cylinder/grid.py::make_vortex_street_domain`` assembles from
``pict/data/shapes.py`` (``make_torus_2D`` :679-766, ``generate_grid_vertices_2D`` :450-507,
``interpolate_vertices_from_borders_2D`` :266-355, ``make_wall_refined_ortho_grid`` :585-638): four blocks wrapped
around the cylinder -- each an O-grid quarter ring of roughly square cells continued by a quadrilateral patch out to
the channel walls / inflow -- and one rectilinear wake block.  Vertex coordinates, boundary calls and connections are
pinned against vectors recorded from the reference function itself (tests/golden/make_golden_cylinder.py ->
tests/test_cylinder_grid.py).

Block order and orientation (every block is right-handed with x to the right and y up in physical space):
0 left (inflow at -x, cylinder at +x), 1 top (cylinder at -y, wall at +y), 2 right (cylinder at -x, wake at +x),
3 bottom (wall at -y, cylinder at +y), 4 vortex street (walls at -y/+y, outflow at +x).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

from ..simulation.grids import wall_refined_edges, weights_exp
from .channel import inflow_profile

LEFT, TOP, RIGHT, BOTTOM, WAKE = range(5)


def _lerp_points(a, b, w):
    w = np.asarray(w, dtype=np.float64)[:, None]
    return np.asarray(a, np.float64)[None, :] * (1.0 - w) + np.asarray(b, np.float64)[None, :] * w


def patch_from_borders(corners, borders, res_y: int, res_x: int, y_weights=None, x_weights=None) -> np.ndarray:
    """Vertices ``[2, res_y, res_x]`` of a quadrilateral patch with prescribed borders.

    corners: (-x-y, +x-y, -x+y, +x+y); borders: [-x, +x, -y, +y] as ``[n, 2]`` arrays or None (straight line between the
    corners, spaced by ``y_weights`` on the -x/+x sides and uniformly on the -y/+y sides).  Every row is the blend of the
    -y and +y borders, shifted to start on the -x border and stretched per coordinate to end on the +x border
    (shapes.py:335-353; a coordinate whose blended extent vanishes is shifted linearly instead, for the whole row).
    """
    wy = np.arange(res_y) / (res_y - 1) if y_weights is None else np.asarray(y_weights, np.float64)
    wx = np.arange(res_x) / (res_x - 1)
    wxb = wx if x_weights is None else np.asarray(x_weights, np.float64)   # spacing of straight -y/+y borders only
    if len(wy) != res_y or len(wxb) != res_x:
        raise ValueError("weights must have one entry per vertex row / column")
    b = list(borders) if borders is not None else [None] * 4
    if b[0] is None:
        b[0] = _lerp_points(corners[0], corners[2], wy)
    if b[1] is None:
        b[1] = _lerp_points(corners[1], corners[3], wy)
    if b[2] is None:
        b[2] = _lerp_points(corners[0], corners[1], wxb)
    if b[3] is None:
        b[3] = _lerp_points(corners[2], corners[3], wxb)
    b = [np.asarray(x, np.float64) for x in b]
    out = np.zeros((2, res_y, res_x))
    for j in range(res_y):
        row = b[2] * (1.0 - wy[j]) + b[3] * wy[j]          # [res_x, 2]
        start, extent = row[0], row[-1] - row[0]
        target = b[1][j] - b[0][j]
        if np.any(np.isclose(extent, 0.0)):
            vals = row - start + (target - extent)[None, :] * wx[:, None] + b[0][j]
        else:
            vals = (row - start) * (target / extent)[None, :] + b[0][j]
        out[:, j, :] = vals.T
    return out


def ring_sector(res: int, r1: float, r2: float, start_deg: float, sweep_deg: float) -> np.ndarray:
    """Quarter ring ``[2, n_r + 1, res + 1]``: x along the angle, y along the radius, radial spacing growing with the
    radius so that the cells stay roughly square (shapes.py:679-740)."""
    nx = res + 1
    start = math.radians(start_deg % 360.0)
    step = math.radians(sweep_deg / (nx - 1))
    ang = start + step * np.arange(nx)
    inner = np.stack([np.cos(ang) * r1, np.sin(ang) * r1], axis=1)
    outer = np.stack([np.cos(ang) * r2, np.sin(ang) * r2], axis=1)
    width_scale = 2.0 * math.pi / nx * (abs(sweep_deg) / 360.0)
    sizes, dpos = [], r1
    while dpos < r2:
        w = dpos * width_scale
        sizes.append(w)
        dpos += w
    sizes = np.asarray(sizes) / ((dpos - r1) / (r2 - r1))
    wr = np.concatenate([[0.0], np.cumsum(sizes) / (r2 - r1)])
    end = start + math.radians(sweep_deg)
    corners = [(math.cos(start) * r1, math.sin(start) * r1), (math.cos(end) * r1, math.sin(end) * r1),
               (math.cos(start) * r2, math.sin(start) * r2), (math.cos(end) * r2, math.sin(end) * r2)]
    return patch_from_borders(corners, [None, None, inner, outer], len(wr), nx, y_weights=wr)


@dataclass
class CylinderMesh:
    coords: List[np.ndarray]                       # per block [d, (nz+1,) ny+1, nx+1] float32
    names: List[str]
    fixed: Dict[Tuple[int, str], np.ndarray]       # (block, face) -> Dirichlet velocity [d, face cells] (zeros = wall)
    connections: List[Tuple]                       # (b1, face1, b2, face2, axis1[, axis2])
    outflow: Tuple[int, str] = (WAKE, "+x")
    cylinder_faces: List[Tuple[int, str]] = field(default_factory=lambda: [(LEFT, "+x"), (TOP, "-y"), (RIGHT, "-x"), (BOTTOM, "+y")])
    dims: int = 2
    periodic: List[Tuple[int, str]] = field(default_factory=list)


def make_vortex_street_mesh(resolution: int, domain_height: float = 4.1, domain_length: float = 22.0,
                            cylinder_radius: float = 0.5, cylinder_offset_y: float = 0.05, circle_thickness: float = 0.5,
                            quad_thickness_x: float = 1.0, refinement_base: float = 0.95) -> CylinderMesh:
    """Defaults are the arguments of ``CylinderEnvBase._get_domain`` (cylinder_env_base.py:233-252)."""
    res = int(resolution)
    quad_y = quad_thickness_x + cylinder_offset_y
    if not math.isclose(domain_height, 2 * cylinder_radius + 2 * circle_thickness + 2 * quad_y):
        raise ValueError("domain_height does not match cylinder_radius, circle_thickness and quad thickness")
    x_min = -(cylinder_radius + circle_thickness + quad_thickness_x)
    x_max = domain_length + x_min
    r1, r2 = cylinder_radius, cylinder_radius + circle_thickness
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # the reference builds every patch in float32

    # ---- quarter rings, re-oriented to x right / y up (grid.py:109-146)
    ring_top = f32(ring_sector(res, r1, r2, 135, -90))
    ring_right = f32(ring_sector(res, r1, r2, 45, -90)).transpose(0, 2, 1)[:, ::-1, :]
    ring_bot = f32(ring_sector(res, r1, r2, -45, -90))[:, ::-1, ::-1]
    ring_left = f32(ring_sector(res, r1, r2, -135, -90)).transpose(0, 2, 1)[:, :, ::-1]
    n_rad = ring_top.shape[1] - 1

    # ---- quadrilateral patches between the ring and the channel (grid.py:148-232)
    ox = r2 + quad_thickness_x
    oy_top, oy_bot = r2 + quad_y + cylinder_offset_y, r2 + quad_y - cylinder_offset_y
    qi = math.sin(math.radians(45)) * r2
    n_ang = res + 1
    n_quad = int(math.ceil(quad_y / circle_thickness * n_rad))
    pts = lambda a: np.asarray(a, np.float64).T  # [2, n] -> [n, 2]
    quad_top = f32(patch_from_borders([(-qi, qi), (qi, qi), (-ox, oy_top), (ox, oy_top)],
                                      [None, None, pts(ring_top[:, -1, :]), None], n_quad, n_ang))
    quad_bot = f32(patch_from_borders([(-ox, -oy_bot), (ox, -oy_bot), (-qi, -qi), (qi, -qi)],
                                      [None, None, None, pts(ring_bot[:, 0, :])], n_quad, n_ang))
    quad_right = f32(patch_from_borders([(qi, -qi), (ox, -oy_bot), (qi, qi), (ox, oy_top)],
                                        [pts(ring_right[:, :, -1]), None, None, None], n_ang, n_quad,
                                        y_weights=weights_exp(n_ang - 1, refinement_base, "BOTH")))
    quad_left = f32(patch_from_borders([(-ox, -oy_bot), (-qi, -qi), (-ox, oy_top), (-qi, qi)],
                                       [None, pts(ring_left[:, :, 0]), None, None], n_ang, n_quad))
    left = np.concatenate([quad_left[:, :, :-1], ring_left], axis=2)
    top = np.concatenate([ring_top[:, :-1, :], quad_top], axis=1)
    right = np.concatenate([ring_right[:, :, :-1], quad_right], axis=2)
    bottom = np.concatenate([quad_bot[:, :-1, :], ring_bot], axis=1)

    # ---- wake block (grid.py:234-243)
    n_wake = int(n_quad / quad_y * 18)
    ex, ey = wall_refined_edges(n_wake, res, (-x_min, -oy_bot), (x_max, oy_top), ("+y", "-y"), refinement_base)
    X, Y = np.meshgrid(ex, ey)
    wake = f32(np.stack([X, Y]))

    prof = inflow_profile(domain_height - 2 * cylinder_offset_y, res).astype(np.float32)
    inflow = np.stack([prof, np.zeros_like(prof)])
    wall = lambda n: np.zeros((2, n), np.float32)
    fixed = {
        (LEFT, "-x"): inflow, (LEFT, "+x"): wall(res),
        (TOP, "+y"): wall(res), (TOP, "-y"): wall(res),
        (RIGHT, "-x"): wall(res),
        (BOTTOM, "-y"): wall(res), (BOTTOM, "+y"): wall(res),
        (WAKE, "+y"): wall(n_wake), (WAKE, "-y"): wall(n_wake), (WAKE, "+x"): inflow.copy(),
    }
    connections = [(LEFT, "+y", TOP, "-x", "+y"), (LEFT, "-y", BOTTOM, "-x", "-y"), (RIGHT, "+y", TOP, "+x", "-y"),
                   (RIGHT, "-y", BOTTOM, "+x", "+y"), (RIGHT, "+x", WAKE, "-x", "-y")]
    names = ["BlockCylinderLeft", "BlockCylinderTop", "BlockCylinderRight", "BlockCylinderBottom", "BlockVortexStreet"]
    return CylinderMesh([f32(left), f32(top), f32(right), f32(bottom), wake], names, fixed, connections)


def extrude_mesh(mesh: CylinderMesh, res_z: int, z0: float = -2.0, z1: float = 2.0) -> CylinderMesh:
    """The 3-D variant of the reference (``extrude_grid_z(g, res_z=res, start_z=-2, end_z=2)``, z-periodic blocks,
    connections with the z axes aligned; envs/cylinder/grid.py:281-294, 330-343, 395-416)."""
    z = (z0 * (1.0 - np.arange(res_z + 1) / res_z) + z1 * (np.arange(res_z + 1) / res_z)).astype(np.float32)
    coords = []
    for c in mesh.coords:
        xy = np.broadcast_to(c[:, None], (2, res_z + 1) + c.shape[1:])
        zz = np.broadcast_to(z[None, :, None, None], (1, res_z + 1) + c.shape[1:])
        coords.append(np.ascontiguousarray(np.concatenate([xy, zz], axis=0), dtype=np.float32))
    fixed = {}
    for key, v in mesh.fixed.items():          # [2, n] -> [3, res_z * n], the face cells run (t, z) with t fastest
        v3 = np.zeros((3, res_z, v.shape[1]), np.float32)
        v3[:2] = v[:, None, :]
        fixed[key] = v3.reshape(3, -1)
    conns = []
    for b1, f1, b2, f2, ax in mesh.connections:
        # 2-D: the one remaining axis; 3-D: the axes after the face axis in cyclic order -- z first if the face is a y face
        conns.append((b1, f1, b2, f2, "-z", ax) if f1[1] == "y" else (b1, f1, b2, f2, ax, "-z"))
    out = CylinderMesh(coords, list(mesh.names), fixed, conns, mesh.outflow, list(mesh.cylinder_faces), dims=3,
                       periodic=[(b, "z") for b in range(len(coords))])
    if hasattr(mesh, "outflows"):
        out.outflows = list(mesh.outflows)
    return out


def build_domain(mesh: CylinderMesh, viscosity: float, batch: int = 1, device=None, reference_quirks: bool = True,
                 non_ortho_flags: int = 25, dtype=None):
    """The mesh as a ``MultiBlockDomain`` on the GPU (``make_vortex_street_domain`` + ``PrepareSolve``); ``dtype``: torch.float32
    (default) or torch.float64 (the fp64 build)."""
    import torch

    from ..simulation.multiblock import MultiBlockDomain

    dom = MultiBlockDomain(mesh.dims, viscosity, batch=batch, device=device, reference_quirks=reference_quirks,
                           non_ortho_flags=non_ortho_flags, dtype=dtype or torch.float32)
    blocks = [dom.CreateBlock(c, name=n) for c, n in zip(mesh.coords, mesh.names)]
    for (b, face), vel in mesh.fixed.items():
        blocks[b].CloseBoundary(face, vel)
    for b, axis in mesh.periodic:
        blocks[b].MakePeriodic(axis)
    for b1, f1, b2, f2, *axes in mesh.connections:
        blocks[b1].ConnectBlock(f1, blocks[b2], f2, *axes)
    dom.PrepareSolve()
    return dom
