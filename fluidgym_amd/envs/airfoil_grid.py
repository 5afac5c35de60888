"""The airfoil mesh of the reference as six connected curvilinear blocks.

Own construction of what ``envs/airfoil/grid.py::make_airfoil_domain`` (:247-716) assembles: the section's surface
polyline is split where its normals point at the upper / lower corner of the inflow rectangle into a TOP, a FRONT and a
BOTTOM part; each part carries a C-grid block out to the channel walls / the inflow rectangle (wall-normal spacing
geometric, base 0.97, refined at the surface), a uniform block leads from the inflow plane to the front block, and two
tail blocks (cells growing by ``tail_grow_mul``) carry the wake to the outflow.  Blocks, in the reference's order:
0 left (inflow at -x), 1 front (section at +x), 2 top (section at -y, wall at +y), 3 bottom (wall at -y, section at +y),
4 tail upper (outflow at +x), 5 tail lower (outflow at +x).

The construction is pinned on the reference's recorded meshes (``tests/golden/reference_airfoil_grid.npz``, angle of attack
0 and 20 degrees) by feeding it the surface polyline those meshes contain.  The reference's own polyline is a 160-point
table of a sharp-trailing-edge NACA 0012 (``envs/airfoil/coords.py``) -- data of the reference that is not reproduced
here; :func:`naca0012_sharp` generates the section from its published closed form on a cosine distribution instead, so a
mesh built without an explicit ``surface`` has the reference's topology, resolution and spacing laws but not its exact
surface points.  (States saved by the reference carry their own vertex coordinates and load unchanged, ``domain_io``.)
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from .cylinder_grid import CylinderMesh, patch_from_borders

LEFT, FRONT, TOP, BOTTOM, TAIL_UPPER, TAIL_LOWER = range(6)


def naca0012_sharp(n_points: int = 160) -> np.ndarray:
    """Closed polyline ``[n, 2]`` of the NACA 0012 with a sharp trailing edge (the 4-digit thickness form with the last
    coefficient -0.1036), chord 1: from the trailing edge over the upper surface to the nose and back underneath.

    Chordwise distribution x(t) = 0.65 t^2 + 2.04 t^3 - 1.69 t^4, t uniform from nose to trailing edge: quadratic at the
    nose (uniform arc length ~1.9e-3 around the leading-edge circle, which also sets the first tail cell), spacing
    ~8e-3 at the trailing edge -- the resolution the reference's tabulated section has at both ends, so that the mesh
    sizes (tail blocks ~160 columns) and the time-step restriction come out as there."""
    half = n_points // 2
    t = (np.arange(half) + 0.5) / (half - 0.5)            # nose straddled by the two middle points, t = 1 at the edge
    x = 0.65 * t ** 2 + 2.04 * t ** 3 - 1.69 * t ** 4
    x[-1] = 1.0
    yt = 0.6 * (0.2969 * np.sqrt(x) - 0.1260 * x - 0.3516 * x ** 2 + 0.2843 * x ** 3 - 0.1036 * x ** 4)
    yt[-1] = 0.0
    upper = np.stack([x[::-1], yt[::-1]], axis=1)           # trailing edge -> nose
    lower = np.stack([x, -yt], axis=1)                      # nose -> trailing edge
    return np.concatenate([upper, lower], axis=0)


def weights_exp(res: int, base: float, refinement: str) -> np.ndarray:
    """``make_weights_exp`` (shapes.py:398-411): res cells growing by ``base`` from the START or towards the END."""
    e = np.arange(res)
    if refinement == "END":
        e = e[::-1]
    sizes = base ** e.astype(np.float64)
    return np.concatenate([[0.0], np.cumsum(sizes) / sizes.sum()])


def _point_line_distance(o, d, p):
    p2 = o + d
    d1, d2 = p2[0] - o[0], p2[1] - o[1]
    a = np.abs(d1 * (o[1] - p[1]) - (o[0] - p[0]) * d2)
    return a / np.sqrt(d1 * d1 + d2 * d2)


def _front_corner_indices(normals, half_height, width_left, attack_angle_deg):
    """Where the normals of the front part point at the upper / lower corner of the inflow rectangle
    (``_ray_rectangle_intersection``, grid.py:149-244; only its index results are used by the mesh)."""
    ang = 180.0 - np.degrees(np.arctan2(normals[1], normals[0]))
    ang = np.where(ang < 180.0, ang, ang - 360.0) - attack_angle_deg
    corner = math.degrees(math.atan2(half_height, width_left))
    upper = ang > 0
    up, low = ang[upper], ang[~upper]
    return int(np.argmin(np.abs(up - corner))), len(up) + int(np.argmin(np.abs(low + corner)))


def make_airfoil_mesh(H: float = 1.4, L: float = 4.5, vel_in: float = 0.3, attack_angle_deg: float = 20.0,
                      resolution_div: int = 1, tail_grow_mul: float = 1.01, surface: Optional[np.ndarray] = None) -> CylinderMesh:
    """Defaults are the arguments of ``AirfoilEnvBase._get_domain`` (airfoil_env_base.py:209-229).  ``surface`` ``[n, 2]``:
    the section at zero angle of attack, trailing edge first, upper side first (default :func:`naca0012_sharp`)."""
    if resolution_div not in (1, 2, 4):
        raise ValueError("resolution_div must be 1, 2, or 4.")
    f32 = np.float32
    offset_left, front_w, hh = 1.5, 0.5, H / 2
    normal_res = 96 // resolution_div
    nw = weights_exp(normal_res - 1, 0.97, "START")
    nwr = weights_exp(normal_res - 1, 0.97, "END")

    pts = np.asarray(naca0012_sharp() if surface is None else surface, f32)        # [n, 2]
    if attack_angle_deg != 0.0:
        a = f32(-attack_angle_deg * math.pi / 180.0)
        c, s = np.cos(a, dtype=f32), np.sin(a, dtype=f32)
        pts = (pts @ np.array([[c, -s], [s, c]], f32).T).astype(f32)
    co = pts.T.copy()                                                               # [2, n]
    len_x = float(co[0].max())
    end = co[:, :1].copy()
    if resolution_div > 1:
        n = (co.shape[1] // resolution_div) * resolution_div
        co = co[:, :n].reshape(2, -1, resolution_div).mean(-1).astype(f32)          # AvgPool2d([1, div])
        co = np.concatenate([end, co, end], axis=1)
    res = co.shape[1]
    end_spacing = f32(np.linalg.norm(co[:, 1] - co[:, 0]))
    end_ext = end + np.array([[end_spacing], [0.0]], f32)
    ext = np.concatenate([end_ext, co, end_ext], axis=1)
    sp2 = ext[:, 2:] - ext[:, :-2]
    normals = np.stack([sp2[1], -sp2[0]])
    normals = normals / np.linalg.norm(normals, axis=0)
    min_size = float(np.linalg.norm(ext[:, 1:] - ext[:, :-1], axis=0).min())
    tail_sizes, tail_dist = [min_size], min_size
    while tail_dist < hh:
        tail_sizes.append(tail_sizes[-1] * tail_grow_mul)
        tail_dist += tail_sizes[-1]
    tail_w = np.concatenate([[0.0], np.cumsum(tail_sizes) / tail_dist])
    tail_res = len(tail_w)

    half = res // 2
    i_top = int(np.argmin(_point_line_distance(co[:, :half], normals[:, :half], np.array([0.0, hh], f32))))
    i_bot = int(np.argmin(_point_line_distance(co[:, half:], normals[:, half:], np.array([0.0, -hh], f32)))) + half
    n_bot_outer = (res - 1 - i_bot) + 1                       # points of the lower outer boundary (grid.py:401-407)
    up_idx, low_idx = _front_corner_indices(normals[:, i_top + 1:i_bot], hh, front_w, attack_angle_deg)
    shift = {1: 7, 2: 2, 4: 0}[resolution_div]
    up_idx, low_idx = up_idx + shift, low_idx + shift
    top_sl = slice(0, n_bot_outer + up_idx + 3)
    front_sl = slice(n_bot_outer + up_idx + 2, n_bot_outer + low_idx + 3)
    bot_sl = slice(n_bot_outer + low_idx + 2, None)
    a_top = co[:, top_sl][:, ::-1].astype(np.float64)        # nose -> trailing edge over the upper side
    a_front = co[:, front_sl][:, ::-1].astype(np.float64)    # bottom -> top around the nose
    a_bot = co[:, bot_sl].astype(np.float64)                 # nose -> trailing edge underneath
    res_top, res_front, res_bot = a_top.shape[1], a_front.shape[1], a_bot.shape[1]
    pst, pet = a_top[:, 0], a_top[:, -1]
    psb, peb = a_bot[:, 0], a_bot[:, -1]

    left = patch_from_borders([(-offset_left, -hh), (-front_w, -hh), (-offset_left, hh), (-front_w, hh)], None,
                              res_front, int(0.75 * normal_res))
    top = patch_from_borders([tuple(pst), tuple(pet), (-front_w, hh), (pet[0], hh)], [None, None, a_top.T, None],
                             normal_res, res_top, y_weights=nwr)
    front = patch_from_borders([(-front_w, -hh), tuple(psb), (-front_w, hh), tuple(pst)], [None, a_front.T, None, None],
                               res_front, normal_res, x_weights=nw)
    bot = patch_from_borders([(-front_w, -hh), (peb[0], -hh), tuple(psb), tuple(peb)], [None, None, None, a_bot.T],
                             normal_res, res_bot, y_weights=nw)
    tail_up = patch_from_borders([tuple(pet), (L, pet[1]), (pet[0], hh), (L, hh)], None, normal_res, tail_res,
                                 y_weights=nwr, x_weights=tail_w)
    tail_low = patch_from_borders([(peb[0], -hh), (L, -hh), tuple(peb), (L, peb[1])], None, normal_res, tail_res,
                                  y_weights=nw, x_weights=tail_w)
    coords = [g.astype(f32) for g in (left, front, top, bot, tail_up, tail_low)]

    # inflow: parabola over the inflow plane's cells, unit mean (profiles.py:35-90), times vel_in
    ny = res_front - 1
    y = np.linspace(-H / 2, H / 2, ny, dtype=f32)
    prof = (6 * (H / 2 - y) * (H / 2 + y) / H ** 2).astype(f32)
    prof = prof / prof.mean() * f32(vel_in)
    zeros = lambda n: np.zeros((2, n), f32)
    const = lambda n: np.stack([np.full(n, vel_in, f32), np.zeros(n, f32)])
    nn = lambda b, axis: coords[b].shape[2 - axis] - 1       # cells of block b along axis (0 = x, 1 = y)
    fixed = {
        (LEFT, "-x"): np.stack([prof, np.zeros(ny, f32)]), (LEFT, "+y"): zeros(nn(LEFT, 0)), (LEFT, "-y"): zeros(nn(LEFT, 0)),
        (TOP, "+y"): zeros(nn(TOP, 0)), (TAIL_UPPER, "+y"): zeros(nn(TAIL_UPPER, 0)), (TAIL_LOWER, "-y"): zeros(nn(TAIL_LOWER, 0)),
        (FRONT, "+x"): zeros(nn(FRONT, 1)), (TOP, "-y"): zeros(nn(TOP, 0)), (BOTTOM, "+y"): zeros(nn(BOTTOM, 0)),
        (TAIL_UPPER, "+x"): const(nn(TAIL_UPPER, 1)), (TAIL_LOWER, "+x"): const(nn(TAIL_LOWER, 1)),
    }
    connections = [(LEFT, "+x", FRONT, "-x", "-y"), (FRONT, "+y", TOP, "-x", "+y"), (FRONT, "-y", BOTTOM, "-x", "-y"),
                   (TOP, "+x", TAIL_UPPER, "-x", "-y"), (BOTTOM, "+x", TAIL_LOWER, "-x", "-y"),
                   (TAIL_UPPER, "-y", TAIL_LOWER, "+y", "-x")]
    names = ["LeftBlock", "AirfoilFront", "AirfoilTop", "AirfoilBot", "TailUpper", "TailLower"]
    mesh = CylinderMesh(coords, names, fixed, connections, outflow=(TAIL_UPPER, "+x"),
                        cylinder_faces=[(BOTTOM, "+y"), (FRONT, "+x"), (TOP, "-y")])
    mesh.outflows = [(TAIL_UPPER, "+x"), (TAIL_LOWER, "+x")]
    return mesh


def surface_from_blocks(front: np.ndarray, top: np.ndarray, bottom: np.ndarray, attack_angle_deg: float) -> np.ndarray:
    """The section polyline ``[n, 2]`` (zero angle of attack, trailing edge first over the upper side) contained in the wall
    faces of a mesh's front / top / bottom blocks -- the inverse of the split above; used to pin the construction on
    recorded meshes."""
    up = top[:, 0, ::-1]                      # trailing edge -> nose side, upper surface
    fr = front[:, ::-1, -1]                   # top -> bottom around the nose
    lo = bottom[:, -1, :]                     # nose side -> trailing edge, lower surface
    co = np.concatenate([up, fr[:, 1:], lo[:, 1:]], axis=1).astype(np.float64)
    a = attack_angle_deg * math.pi / 180.0    # undo the rotation by -aoa
    c, s = math.cos(a), math.sin(a)
    return (np.array([[c, -s], [s, c]]) @ co).T
