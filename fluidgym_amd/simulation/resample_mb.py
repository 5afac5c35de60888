"""Observation resampling of a multi-block curvilinear mesh onto the uniform render / sensor grid.

Replaces ``sample_multi_coords_to_uniform_grid`` (``pict/data/resample.py:254-358`` -> compiled
``SampleTransformedGridLocalToGlobalMulti`` + ``_FillEmptyCells``, ``extensions/resampling.cu:191-609``) for the meshes
of the cylinder / airfoil envs.  The reference scatters every cell centre with atomics, normalises, then fills the
pixels nothing reached by repeated neighbour averaging -- for every field on every call.  All of it is linear in the
field and the mesh never changes, so here the whole chain (splat, normalisation, fill passes) is folded ONCE on the host
into a sparse operator ``W [pixels, cells]``; resampling a field is then one gather per pixel, and the env's sensors
read only their own rows.  Same AABB_OUTER transform and fp32 cell-centre arithmetic as the single-block resampler
(``simulation/resample.py``).  2-D meshes use the folded operator; for 3-D meshes (4.7 M output pixels at resolution
24) the operator is kept factored as a :class:`ResamplePlan` -- splat table + the fixed schedule of the fill passes -- which
the GPU replays for whole fields and from which the sensors' rows are expanded on demand.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import scipy.sparse as sp
import torch

EPS_F32 = 1e-8  # getEps<float>() (resampling.cu:174)


def _cell_index(coords_list, out_shape):
    d = int(np.asarray(coords_list[0]).shape[0])
    if d == 3:
        return _cell_index_nd(coords_list, out_shape)
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


def _cell_index_nd(coords_list, out_shape):
    d = int(np.asarray(coords_list[0]).shape[0])
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
        ctr = c
        for a in range(d):                       # mean of the 2^d vertices in fp32 (coords_to_center_coords)
            ax = ctr.ndim - 1 - a
            m = ctr.shape[ax] - 1
            ctr = np.take(ctr, range(0, m), axis=ax) + np.take(ctr, range(1, m + 1), axis=ax)
        ctr = (ctr * np.float32(1.0 / (1 << d))).astype(np.float32)
        g.append((ctr.reshape(d, -1).astype(np.float64) - offs[:, None].astype(np.float64)) / float(scale))
    return np.concatenate(g, axis=1), float(scale), offs


class ResamplePlan:
    """The resampling chain of one mesh in factored form, for 2-D or 3-D (host side, NumPy).

    ``pix [K, N]`` / ``w [K, N]``: the output pixels each cell centre is splatted onto and the multilinear weights
    (pixel -1 = outside); K = 2^d corners, or the compiled kernel's 2 d = 6 of 8 in 3-D with ``corners_3d_quirk``
    (``resampling.cu:320``: the two corners "y upper, z upper" are never written).  ``passes``: the hole-fill schedule --
    pass k fills pixels ``tgt`` with the mean of their face neighbours ``src`` (-1 = none) that were filled before the
    pass.  Which pixels a pass fills depends on the mesh only, never on the field."""

    def __init__(self, coords_list: Sequence[np.ndarray], out_shape: Sequence[int], fill_max_steps: int = 0,
                 corners_3d_quirk: bool = True):
        d = int(np.asarray(coords_list[0]).shape[0])
        self.d = d
        self.out_shape = tuple(int(v) for v in out_shape[:d])          # (x, y(, z))
        self.grid = tuple(self.out_shape[d - 1 - q] for q in range(d))  # array shape ((oz,) oy, ox)
        g, _, _ = _cell_index_nd(coords_list, out_shape)
        N = g.shape[1]
        self.n_cells = N
        base = [np.floor(g[a]).astype(np.int64) for a in range(d)]
        frac = [g[a] - base[a] for a in range(d)]
        K = (2 * d) if (d == 3 and corners_3d_quirk) else (1 << d)
        pix = np.full((K, N), -1, np.int64)
        w = np.zeros((K, N), np.float64)
        for corner in range(K):
            ok = np.ones(N, bool)
            wt = np.ones(N, np.float64)
            flat = np.zeros(N, np.int64)
            stride = 1
            for a in range(d):
                up = (corner >> a) & 1
                pa = base[a] + up
                wt = wt * (frac[a] if up else 1.0 - frac[a])
                ok &= (pa >= 0) & (pa < self.out_shape[a])
                flat += pa * stride
                stride *= self.out_shape[a]
            pix[corner, ok] = flat[ok]
            w[corner, ok] = wt[ok]
        self.pix, self.w = pix, w.astype(np.float32)
        P = int(np.prod(self.out_shape))
        self.n_pixels = P
        valid = pix >= 0
        wsum = np.bincount(pix[valid], weights=self.w[valid].astype(np.float64), minlength=P)
        self.wsum = wsum.astype(np.float32)
        filled = (self.wsum > EPS_F32).reshape(self.grid)
        self.pass_of = np.where(filled, 0, -1).astype(np.int8).reshape(-1)
        self.passes = []
        idx = np.arange(P, dtype=np.int64).reshape(self.grid)
        for k in range(int(fill_max_steps)):
            if filled.all():
                break
            cnt = np.zeros(self.grid, np.int8)
            nbs = []
            for ax in range(d - 1, -1, -1):        # x first
                for sgn in (-1, 1):
                    nb = np.full(self.grid, -1, np.int64)
                    src = [slice(None)] * d
                    dst = [slice(None)] * d
                    if sgn == -1:
                        src[ax], dst[ax] = slice(0, -1), slice(1, None)
                    else:
                        src[ax], dst[ax] = slice(1, None), slice(0, -1)
                    sel = filled[tuple(src)]
                    view = nb[tuple(dst)]
                    view[sel] = idx[tuple(src)][sel]
                    cnt += nb >= 0
                    nbs.append(nb)
            newly = (cnt > 0) & ~filled
            if not newly.any():
                break
            tgt = idx[newly]
            srcs = np.stack([nb[newly] for nb in nbs], axis=1)
            self.passes.append((tgt, srcs, cnt[newly].astype(np.float32)))
            self.pass_of[tgt] = k + 1
            filled = filled | newly
        self._by_pixel = None

    # ---- rows of the folded operator for a few pixels (sensors)
    def _pixel_csr(self):
        if self._by_pixel is None:
            valid = self.pix >= 0
            cells = np.broadcast_to(np.arange(self.n_cells), self.pix.shape)[valid]
            px, wv = self.pix[valid], self.w[valid]
            order = np.argsort(px, kind="stable")
            px, cells, wv = px[order], cells[order], wv[order]
            ptr = np.searchsorted(px, np.arange(self.n_pixels + 1))
            self._by_pixel = (ptr, cells, wv)
            self._fill_pos = [dict(zip(t.tolist(), range(len(t)))) for t, _, _ in self.passes]
        return self._by_pixel

    def row(self, pixel: int, _memo=None) -> dict:
        """{cell: weight} of one output pixel after normalisation and fill."""
        memo = {} if _memo is None else _memo
        if pixel in memo:
            return memo[pixel]
        ptr, cells, wv = self._pixel_csr()
        k = int(self.pass_of[pixel])
        out = {}
        if k == 0:
            a, b = ptr[pixel], ptr[pixel + 1]
            inv = 1.0 / float(self.wsum[pixel])
            for c, x in zip(cells[a:b].tolist(), wv[a:b].tolist()):
                out[c] = out.get(c, 0.0) + x * inv
        elif k > 0:
            tgt, srcs, cnt = self.passes[k - 1]
            j = self._fill_pos[k - 1][pixel]
            inv = 1.0 / float(cnt[j])
            for q in srcs[j].tolist():
                if q >= 0:
                    for c, x in self.row(q, memo).items():
                        out[c] = out.get(c, 0.0) + x * inv
        memo[pixel] = out
        return out

    def rows_ell(self, pixels: np.ndarray):
        """(cell index [S, K], weight [S, K]) of the given flat pixels, zero padded."""
        memo = {}
        rows = [self.row(int(p), memo) for p in np.asarray(pixels).reshape(-1)]
        K = max(max((len(r) for r in rows), default=1), 1)
        idx = np.zeros((len(rows), K), np.int64)
        wt = np.zeros((len(rows), K), np.float32)
        for i, r in enumerate(rows):
            idx[i, : len(r)] = list(r.keys())
            wt[i, : len(r)] = list(r.values())
        return idx, wt

    def apply_numpy(self, field: np.ndarray) -> np.ndarray:
        """field [C, N] -> [C, *grid] on the host (tests)."""
        C = field.shape[0]
        out = np.zeros((C, self.n_pixels), np.float64)
        valid = self.pix >= 0
        for c in range(C):
            vals = (np.broadcast_to(field[c], self.pix.shape)[valid].astype(np.float64)) * self.w[valid]
            out[c] = np.bincount(self.pix[valid], weights=vals, minlength=self.n_pixels)
        ok = self.pass_of == 0
        out[:, ok] /= self.wsum[ok]
        for tgt, srcs, cnt in self.passes:
            padded = np.concatenate([out, np.zeros((C, 1))], axis=1)
            out[:, tgt] = padded[:, srcs].sum(-1) / cnt
        return out.reshape((C,) + self.grid)


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


def _ptr(t: torch.Tensor):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


class SensorGather:
    """Rows of the resampling operator for a set of pixels, applied on the GPU by ``fg_sparse_apply_ell`` (csrc/fg_resample.hip):
    ``gather(field [..., N]) -> [..., S]``.  Indexable like the ``(idx, w)`` pair it replaces."""

    def __init__(self, idx: torch.Tensor, w: torch.Tensor):
        self.idx, self.w = idx, w
        self._idx32 = idx.to(torch.int32).contiguous()
        self._w = w.to(torch.float32).contiguous()

    def __iter__(self):
        return iter((self.idx, self.w))

    def __call__(self, field: torch.Tensor) -> torch.Tensor:
        from .. import _lib as L

        if not field.is_cuda:   # stubbed-solver CPU tests: the same sum written with torch
            return (field[..., self.idx] * self.w).sum(-1)
        lead = field.shape[:-1]
        flat = field.reshape(-1, field.shape[-1])
        if flat.dtype != torch.float32:   # float64 domains: observations are resampled in float32 (the kernels' word size)
            flat = flat.to(torch.float32)
        if not flat.is_contiguous():
            flat = flat.contiguous()
        S, K = self._idx32.shape
        out = torch.empty(flat.shape[0], S, dtype=torch.float32, device=field.device)
        st = torch.cuda.current_stream(field.device).cuda_stream
        L.check(L.load().fg_sparse_apply_ell(_ptr(self._idx32), _ptr(self._w), S, K, _ptr(flat), flat.shape[1], flat.shape[0], _ptr(out),
                                             __import__("ctypes").c_void_p(st)))
        return out.reshape(*lead, S)


class MultiBlockResampler3D:
    """The factored plan replayed on the GPU: one scatter-add for the splat, one gather per fill pass."""

    def __init__(self, coords_list: Sequence[np.ndarray], out_shape: Sequence[int], fill_max_steps: int = 0, device=None,
                 corners_3d_quirk: bool = True):
        self.plan = ResamplePlan(coords_list, out_shape, fill_max_steps, corners_3d_quirk)
        self.out_shape = self.plan.out_shape
        self.device = torch.device("cuda") if device is None else torch.device(device)
        pl = self.plan
        valid = pl.pix >= 0
        cells = np.broadcast_to(np.arange(pl.n_cells), pl.pix.shape)[valid]
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=self.device)
        self._cell, self._pix, self._w = t(cells, torch.long), t(pl.pix[valid], torch.long), t(pl.w[valid], torch.float32)
        inv = np.zeros(pl.n_pixels, np.float32)
        ok = pl.pass_of == 0
        inv[ok] = 1.0 / pl.wsum[ok]
        self._inv = t(inv, torch.float32)
        P = pl.n_pixels                      # fill sources: -1 -> the extra zero slot P
        self._passes = [(t(tgt, torch.long), t(np.where(srcs >= 0, srcs, P), torch.long), t(1.0 / cnt, torch.float32))
                        for tgt, srcs, cnt in pl.passes]

    def __call__(self, field: torch.Tensor) -> torch.Tensor:
        """field [..., N] -> [..., oz, oy, ox]."""
        pl = self.plan
        lead = field.shape[:-1]
        flat = field.reshape(-1, field.shape[-1]).to(torch.float32)
        out = torch.zeros(flat.shape[0], pl.n_pixels + 1, dtype=torch.float32, device=self.device)
        out.index_add_(1, self._pix, flat[:, self._cell] * self._w)
        out[:, :-1] *= self._inv
        for tgt, srcs, inv_cnt in self._passes:
            out[:, tgt] = out[:, srcs].sum(-1) * inv_cnt
        return out[:, :-1].reshape(*lead, *pl.grid)

    def sensor_gather(self, pixel_xyz: np.ndarray):
        """ELL rows for pixels ``[(x, y, z), ...]``: (cell index [S, K] long, weight [S, K] float32)."""
        pix = np.asarray(pixel_xyz, np.int64)
        ox, oy, _ = self.out_shape
        idx, w = self.plan.rows_ell(pix[:, 0] + ox * (pix[:, 1] + oy * pix[:, 2]))
        return SensorGather(torch.as_tensor(idx, device=self.device), torch.as_tensor(w, device=self.device))


class MultiBlockResampler:
    """Static resampling operator on the GPU: full fields via one sparse product, sensors via a small gather."""

    def __init__(self, coords_list: Sequence[np.ndarray], out_shape: Sequence[int], fill_max_steps: int = 0, device=None):
        self.out_shape = (int(out_shape[0]), int(out_shape[1]))
        self.W_host = build_operator(coords_list, out_shape, fill_max_steps)
        self.device = torch.device("cuda") if device is None else torch.device(device)
        W = self.W_host.tocsr()
        W.sort_indices()
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=self.device)
        self._indptr, self._col, self._val = t(W.indptr, torch.int32), t(W.indices, torch.int32), t(W.data, torch.float32)
        self._rows = W.shape[0]

    def __call__(self, field: torch.Tensor) -> torch.Tensor:
        """field [..., N] -> [..., oy, ox]: one native sparse gather (``fg_sparse_apply_csr``, csrc/fg_resample.hip)."""
        import ctypes

        from .. import _lib as L

        lead = field.shape[:-1]
        flat = field.reshape(-1, field.shape[-1]).to(torch.float32)
        if not flat.is_contiguous():
            flat = flat.contiguous()
        out = torch.empty(flat.shape[0], self._rows, dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(L.load().fg_sparse_apply_csr(_ptr(self._indptr), _ptr(self._col), _ptr(self._val), self._rows, _ptr(flat), flat.shape[1],
                                             flat.shape[0], _ptr(out), ctypes.c_void_p(st)))
        return out.reshape(*lead, self.out_shape[1], self.out_shape[0])

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
        return SensorGather(torch.as_tensor(idx, device=self.device), torch.as_tensor(w, device=self.device))
