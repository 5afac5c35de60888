"""Vertex-grid generators for the rectilinear meshes the RBC / TCF / channel envs use.

Own implementations of the weight laws behind the reference's generators
(``simulation/pict/data/shapes.py:394-447, 585-676`` and ``envs/tcf/grid.py:15-31``); pinned against
vertex grids produced by the reference's Python in ``tests/golden/reference_python.npz``.
All meshes here are axis-aligned boxes, so a grid is fully described by per-axis vertex positions
("edges"); :func:`vertex_grid` expands them to the reference's ``[1, d, (Z+1,) Y+1, X+1]`` layout.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch


def weights_linear(res: int) -> np.ndarray:
    return np.arange(res + 1, dtype=np.float64) / res


def weights_exp(res: int, base: float, refinement: str) -> np.ndarray:
    """Cumulative weights of geometrically growing cells, refined at START / END / BOTH ends
    (shapes.py:398-411)."""
    k = np.arange(res)
    if refinement == "END":
        k = k[::-1]
    elif refinement == "BOTH":
        k = np.concatenate([k[: res // 2], k[::-1][res // 2:]])
    elif refinement != "START":
        raise ValueError(f"unknown refinement {refinement!r}")
    sizes = np.power(float(base), k.astype(np.float64))
    return np.concatenate([[0.0], np.cumsum(sizes) / sizes.sum()])


def weights_exp_global(res: int, global_scale: float, refinement: str) -> np.ndarray:
    """shapes.py:414-421: base chosen so that largest/smallest cell = ``global_scale``."""
    n = res // 2 if refinement == "BOTH" else res
    return weights_exp(res, global_scale ** (1.0 / (n - 1)), refinement)


def weights_cos(res: int, refinement: str) -> np.ndarray:
    """Chebyshev-like clustering (shapes.py:424-447)."""
    t = np.arange(res + 1, dtype=np.float64) / res
    if refinement == "START":
        return 1.0 - np.cos(t * math.pi / 2)
    if refinement == "END":
        return -np.cos(math.pi / 2 + t * math.pi / 2)
    if refinement == "BOTH":
        return 0.5 - 0.5 * np.cos(t * math.pi)
    raise ValueError(f"unknown refinement {refinement!r}")


def tcf_y_weights(N: int = 1, ny_half: int = 48) -> np.ndarray:
    """Wall-normal vertex weights of the turbulent-channel grid (envs/tcf/grid.py:15-31): cells grow
    geometrically (ratio 1.2^(N/2)) from both walls, the two cells next to the centre plane absorb
    the remainder so that the centre vertex sits exactly at 0.5."""
    ny = 2 * (ny_half // N)
    r = 1.2 ** (N / 2)
    h0 = 0.5 * (1 - r) / (1 - r ** (ny / 2))
    n_geo = (ny - 2) // 2
    lower = np.cumsum(h0 * r ** np.arange(n_geo))
    y = np.zeros(ny + 1)
    y[1: 1 + n_geo] = lower
    y[ny // 2] = 0.5
    y[ny - n_geo: ny] = (1.0 - lower)[::-1]
    y[ny] = 1.0
    return y


def lerp_edges(lo: float, hi: float, weights: np.ndarray) -> np.ndarray:
    w = np.asarray(weights, dtype=np.float64)
    return lo * (1.0 - w) + hi * w


def wall_refined_edges(res_x: int, res_y: int, corner_lower, corner_upper, wall_refinement: Sequence[str] = (),
                       base=1.05) -> List[np.ndarray]:
    """Per-axis vertex positions of ``make_wall_refined_ortho_grid`` (shapes.py:585-638)."""
    bases = list(base) if isinstance(base, (list, tuple)) else [base, base]

    def axis_weights(res, lo_tag, hi_tag, b):
        lo, hi = lo_tag in wall_refinement, hi_tag in wall_refinement
        if lo and hi:
            return weights_exp(res, b, "BOTH")
        if lo:
            return weights_exp(res, b, "START")
        if hi:
            return weights_exp(res, b, "END")
        return np.arange(res + 1, dtype=np.float64) / res

    wx = axis_weights(res_x, "-x", "+x", bases[0])
    wy = axis_weights(res_y, "-y", "+y", bases[1])
    return [lerp_edges(corner_lower[0], corner_upper[0], wx), lerp_edges(corner_lower[1], corner_upper[1], wy)]


def vertex_grid(edges: Sequence[np.ndarray], dtype=torch.float32) -> torch.Tensor:
    """``[1, d, (Z+1,) Y+1, X+1]`` vertex coordinates of the tensor-product grid."""
    d = len(edges)
    mesh = np.meshgrid(*[np.asarray(e, dtype=np.float64) for e in reversed(edges)], indexing="ij")
    coords = np.stack([mesh[d - 1 - a] for a in range(d)], axis=0)[None]
    return torch.from_numpy(coords).to(dtype)


def edges_from_vertex_grid(coords, rtol: float = 1e-5):
    """Inverse of :func:`vertex_grid`; raises if the grid is not rectilinear (the HIP path covers
    tensor-product meshes only, see include/fluidgym_hip.h)."""
    c = coords.detach().cpu().double().numpy() if isinstance(coords, torch.Tensor) else np.asarray(coords, np.float64)
    if c.ndim >= 3 and c.shape[0] == 1 and c.shape[1] == c.ndim - 2:
        c = c[0]
    d = c.shape[0]
    edges = []
    scale = max(float(np.abs(c).max()), 1e-30)
    for a in range(d):
        ax = d - a  # axis of c holding spatial axis a (c is [d, (z,) y, x])
        moved = np.moveaxis(c[a], ax - 1, -1)
        e = moved.reshape(-1, moved.shape[-1])
        if np.abs(e - e[0]).max() > rtol * scale:
            raise ValueError("vertex grid is not rectilinear (coordinate varies along another axis)")
        if not np.all(np.diff(e[0]) > 0):
            raise ValueError("vertex coordinates must be strictly increasing")
        edges.append(e[0].copy())
    return edges
