"""Host side of the observation resampling: block data -> uniform "render" grid.

Mirrors ``sample_multi_coords_to_uniform_grid`` / ``_resample_block_data`` of the reference
(``simulation/pict/data/resample.py:254-358``, ``pict/util/output.py:381-409``) for one rectilinear block.  The
geometry (world -> output-index transform, which source cells touch which output cell, which empty cells get
filled in which pass) is static, so it is computed once here / in ``fg_resampler_create``; ``__call__`` is one
gather kernel (+ one small kernel per fill pass) through the C ABI (``fg_resample``, ``csrc/fg_resample.hip``).
"""
from __future__ import annotations

import ctypes
from typing import Sequence

import numpy as np
import torch

from .. import _lib as L


def aabb_outer_axis_maps(edges: Sequence[np.ndarray], out_shape: Sequence[int]):
    """Continuous output index of every source cell centre along each axis for ``transform_uniform="AABB_OUTER"``
    (``make_uniform_transform_AABB_outer``, resample.py:66-97): isotropic scale ``max_a(size_a / n_a)``, the index
    space is centred on the bounding-box centre, ``lower -> -0.5`` on the axis that sets the scale.  The transform
    entries are formed in fp32 and inverted in fp64 like the reference does (resample.py:436); the cell centres are
    the fp32 average of the vertex coordinates (``coords_to_center_coords``, shapes.py:216-225)."""
    e32 = [np.asarray(e, np.float32) for e in edges]
    lower = np.array([e.min() for e in e32], np.float32)
    upper = np.array([e.max() for e in e32], np.float32)
    size = upper - lower
    center = lower + size * np.float32(0.5)
    n = np.asarray(out_shape, np.float32)
    scale = np.float32(np.max(size / n))
    offs = (scale * (-n * np.float32(0.5) + np.float32(0.5)) + center).astype(np.float32)
    maps = []
    for a, e in enumerate(e32):
        xc = (np.float32(0.5) * (e[:-1] + e[1:])).astype(np.float32)
        maps.append((xc.astype(np.float64) - float(offs[a])) / float(scale))
    return maps


def output_shape(out_shape, dims: int):
    """``get_output_shape`` (resample.py:164-184): int -> [n] * dims; sequences must have ``dims`` ints (x, y[, z])."""
    if isinstance(out_shape, (int, np.integer)):
        return [int(out_shape)] * dims
    out_shape = [int(v) for v in (out_shape.tolist() if hasattr(out_shape, "tolist") else out_shape)]
    if len(out_shape) != dims:
        raise ValueError("Resampling output shape does not match dimensions.")
    return out_shape


class UniformResampler:
    """``data [B, C, (nz,) ny, nx] -> [B, C, (oz,) oy, ox]`` on the uniform grid ``out_shape = (x, y[, z])``.

    ``compiled_corner_rule=True`` (default, what the reference's envs run: ``differentiable=False``) reproduces the
    compiled kernel's 6-of-8 corner loop in 3-D (``resampling.cu:320``); ``False`` is the reference's differentiable
    pure-torch form (all corners).  In 2-D both are the same."""

    def __init__(self, edges: Sequence[np.ndarray], out_shape, fill_max_steps: int = 0, device=None,
                 compiled_corner_rule: bool = True):
        if not torch.cuda.is_available():
            raise L.NativeLibraryError("fluidgym_amd needs a ROCm GPU (MI355X); there is no CPU path")
        self.lib = L.load()
        self.dims = len(edges)
        assert self.dims in (2, 3)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.out_shape = output_shape(out_shape, self.dims)
        self.fill_max_steps = int(fill_max_steps)
        self.n_src = [len(e) - 1 for e in edges]
        maps = aabb_outer_axis_maps(edges, self.out_shape)
        base = np.concatenate([np.floor(g) for g in maps]).astype(np.int32)
        frac = np.concatenate([g - np.floor(g) for g in maps]).astype(np.float32)
        i32 = ctypes.POINTER(ctypes.c_int32)
        n_src = np.asarray(self.n_src, np.int32)
        n_out = np.asarray(self.out_shape, np.int32)
        handle = ctypes.c_void_p()
        L.check(self.lib.fg_resampler_create(self.dims, n_src.ctypes.data_as(i32), n_out.ctypes.data_as(i32),
                                             base.ctypes.data_as(i32), frac.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                             int(compiled_corner_rule), self.device.index or 0, ctypes.byref(handle)))
        self.handle = handle

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        assert data.is_cuda and data.dim() == self.dims + 2
        assert list(data.shape[2:]) == list(reversed(self.n_src)), "data does not match the block resolution"
        in_dtype = data.dtype
        data = data.to(torch.float32).contiguous()      # the gather is an fp32 kernel; fp64 fields are observed through a cast
        B, C = data.shape[:2]
        out = torch.empty((B, C) + tuple(reversed(self.out_shape)), dtype=torch.float32, device=data.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(data.device).cuda_stream)
        L.check(self.lib.fg_resample(self.handle, ctypes.c_void_p(data.data_ptr()), B, C, ctypes.c_void_p(out.data_ptr()),
                                     self.fill_max_steps, stream))
        return out if in_dtype == torch.float32 else out.to(in_dtype)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.fg_resampler_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
