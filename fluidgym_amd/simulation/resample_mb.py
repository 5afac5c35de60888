"""Observation resampling of a multi-block curvilinear mesh onto the uniform render / sensor grid.

Replaces ``sample_multi_coords_to_uniform_grid`` (``pict/data/resample.py:254-358`` -> compiled
``SampleTransformedGridLocalToGlobalMulti`` + ``_FillEmptyCells``, ``extensions/resampling.cu:191-609``) for the meshes
of the cylinder / airfoil envs.  The reference scatters every cell centre with atomics, normalises, then fills the
pixels nothing reached by repeated neighbour averaging -- for every field on every call.  All of it is linear in the
field and the mesh never changes, so here the whole chain (splat, normalisation, fill passes) is folded ONCE on the host
into a sparse operator ``W [pixels, cells]``; resampling a field is then one gather per pixel, and the env's sensors
read only their own rows.  Same AABB_OUTER transform and fp32 cell-centre arithmetic as the single-block resampler
(``simulation/resample.py``); 2-D only (a 3-D mesh extruded along z resamples plane by plane).
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import scipy.sparse as sp
import torch

EPS_F32 = 1e-8  # getEps<float>() (resampling.cu:174)


def _cell_index(coords_list, out_shape):
    d = 2
    c32 = [np.asarray(c, np.float32) for c in coords_list]
    allv = np.concatenate([c.reshape(d, -1) for c in c32], axis=1)
    lower, upper = allv.min(axis=1), allv.max(axis=1)
    size = upper - lower
    center = lower + size * np.float32(0.5)
    n = np.asarray(out_shape, np.float32)
    scale = np.float32(np.max(size / n))
    offs = (scale * (-n * np.float32(0.5) + np.float32(0.5)) + center).astype(np.float32)
    g = []
    for c in c32:
        ctr = ((c[:, :-1, :-1] + c[:, :-1, 1:]) + (c[:, 1:, :-1] + c[:, 1:, 1:])) * np.float32(0.25)
        g.append((ctr.reshape(d, -1).astype(np.float64) - offs[:, None].astype(np.float64)) / float(scale))
    return np.concatenate(g, axis=1), float(scale), offs


def build_operator(coords_list: Sequence[np.ndarray], out_shape: Sequence[int], fill_max_steps: int = 0) -> sp.csr_matrix:
    """``W [oy * ox, N]`` with N = cells of all blocks (block order, x fastest)."""
    ox, oy = int(out_shape[0]), int(out_shape[1])
    g, _, _ = _cell_index(coords_list, out_shape)
    N = g.shape[1]
    bx, by = np.floor(g[0]).astype(np.int64), np.floor(g[1]).astype(np.int64)
    fx, fy = g[0] - bx, g[1] - by
    rows, cols, vals = [], [], []
    cell = np.arange(N)
    for cy in range(2):
        for cx in range(2):
            px, py = bx + cx, by + cy
            w = (fx if cx else 1.0 - fx) * (fy if cy else 1.0 - fy)
            ok = (px >= 0) & (px < ox) & (py >= 0) & (py < oy)
            rows.append((py * ox + px)[ok]); cols.append(cell[ok]); vals.append(w[ok].astype(np.float32))
    S = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(oy * ox, N), dtype=np.float64)
    wsum = np.asarray(S.sum(axis=1)).reshape(-1).astype(np.float32)
    filled = wsum > EPS_F32
    inv = np.zeros(oy * ox)
    inv[filled] = 1.0 / wsum[filled]
    W = sp.diags(inv) @ S
    # _FillEmptyCells as operator algebra: an empty pixel with filled face neighbours takes the mean of their rows
    filled = filled.reshape(oy, ox)
    idx = np.arange(oy * ox).reshape(oy, ox)
    for _ in range(int(fill_max_steps)):
        if filled.all():
            break
        cnt = np.zeros((oy, ox), np.int64)
        pr, pc = [], []
        for dy, dx in ((0, -1), (0, 1), (-1, 0), (1, 0)):
            nb_f = np.zeros_like(filled)
            nb_i = np.zeros_like(idx)
            ys, yd = (slice(0, oy - 1), slice(1, oy)) if dy == -1 else ((slice(1, oy), slice(0, oy - 1)) if dy == 1 else (slice(None), slice(None)))
            xs, xd = (slice(0, ox - 1), slice(1, ox)) if dx == -1 else ((slice(1, ox), slice(0, ox - 1)) if dx == 1 else (slice(None), slice(None)))
            nb_f[yd, xd] = filled[ys, xs]
            nb_i[yd, xd] = idx[ys, xs]
            take = nb_f & ~filled
            cnt += nb_f
            pr.append(idx[take]); pc.append(nb_i[take])
        newly = (cnt > 0) & ~filled
        if not newly.any():
            break
        pr, pc = np.concatenate(pr), np.concatenate(pc)
        A = sp.csr_matrix((1.0 / cnt.reshape(-1)[pr], (pr, pc)), shape=(oy * ox, oy * ox))
        W = W + A @ W
        filled = filled | newly
    return W.tocsr()


class MultiBlockResampler:
    """Static resampling operator on the GPU: full fields via one sparse product, sensors via a small gather."""

    def __init__(self, coords_list: Sequence[np.ndarray], out_shape: Sequence[int], fill_max_steps: int = 0, device=None):
        self.out_shape = (int(out_shape[0]), int(out_shape[1]))
        self.W_host = build_operator(coords_list, out_shape, fill_max_steps)
        self.device = torch.device("cuda") if device is None else torch.device(device)
        W = self.W_host.tocoo()
        idx = torch.as_tensor(np.stack([W.row, W.col]), dtype=torch.int64)
        self._W = torch.sparse_coo_tensor(idx, torch.as_tensor(W.data, dtype=torch.float32), size=W.shape).coalesce().to(self.device)

    def __call__(self, field: torch.Tensor) -> torch.Tensor:
        """field [..., N] -> [..., oy, ox]."""
        lead = field.shape[:-1]
        flat = field.reshape(-1, field.shape[-1]).t().contiguous()  # [N, M]
        out = torch.sparse.mm(self._W, flat)                      # [pixels, M]
        return out.t().reshape(*lead, self.out_shape[1], self.out_shape[0])

    def sensor_gather(self, pixel_xy: np.ndarray):
        """ELL rows of the operator for pixels ``[(x, y), ...]``: (cell index [S, K] long, weight [S, K] float32)."""
        pix = np.asarray(pixel_xy, np.int64)
        rows = pix[:, 1] * self.out_shape[0] + pix[:, 0]
        sub = self.W_host[rows]
        K = max(int(np.diff(sub.indptr).max()), 1)
        idx = np.zeros((len(rows), K), np.int64)
        w = np.zeros((len(rows), K), np.float32)
        for r in range(len(rows)):
            a, b = sub.indptr[r], sub.indptr[r + 1]
            idx[r, : b - a] = sub.indices[a:b]
            w[r, : b - a] = sub.data[a:b]
        return torch.as_tensor(idx, device=self.device), torch.as_tensor(w, device=self.device)
