"""Multi-block, non-orthogonal domains on the HIP library.

Host-side mirror of the reference's ``PISOtorch.Domain`` / ``Block`` construction calls for meshes made of several
structured, curvilinear blocks (``Domain.CreateBlock(vertexCoordinates=...)``, ``Block.CloseBoundary``,
``Block.ConnectBlock``, ``Block.MakePeriodic``, ``Domain.PrepareSolve``; ``extensions/PISOtorch.cpp:420-500``,
``domain_structs.cpp:1940-2002``) and of ``Simulation``'s non-orthogonal PISO step
(``pict/PISOtorch_simulation.py:1707-1972``).  What differs by design: every field carries a leading env batch
``B``; cells of all blocks live in ONE flat array per env (block order, x fastest) and all FIXED boundary faces in one
flat slot array, so the kernels of ``csrc/fg_mb_step.hip`` need no per-block launches; the mesh tables are built once
at ``prepare_solve`` on the host (``csrc/fg_mb_topo.hip``).  There is no CPU path: everything below calls the C ABI.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import _lib as L

_FACES = {"-x": 0, "+x": 1, "-y": 2, "+y": 3, "-z": 4, "+z": 5}
_AXES = {"x": 0, "y": 1, "z": 2}


def face_index(face) -> int:
    """``BoundarySideToIndex``: "-x", "+x", ... or the index itself."""
    return _FACES[face] if isinstance(face, str) else int(face)


class MBBlock:
    """One structured block; mirrors the construction-time API of ``PISOtorch.Block``."""

    def __init__(self, domain: "MultiBlockDomain", index: int, coords: np.ndarray, name: str):
        self.domain, self.index, self.name = domain, index, name
        self.coords = coords  # [d, (nz+1,) ny+1, nx+1] float32
        d = coords.shape[0]
        self.size = tuple(coords.shape[-1 - a] - 1 for a in range(d))  # (nx, ny[, nz])
        self.cell_offset = -1
        self.boundary_slot0: List[int] = [-1] * (2 * d)
        self._pending_velocity: Dict[int, np.ndarray] = {}
        self.connections: Dict[int, tuple] = {}   # face -> (other block index, ConnectedBoundary.axes)
        self.periodic_axes: set = set()

    # ---- construction (before prepare_solve)
    def CloseBoundary(self, face, velocity=None):
        """FIXED Dirichlet boundary (``Block::CloseBoundary``); ``velocity`` is [d] (static) or [d, face cells...]."""
        f = face_index(face)
        if velocity is not None:
            self._pending_velocity[f] = np.asarray(
                velocity.detach().cpu().numpy() if isinstance(velocity, torch.Tensor) else velocity, dtype=self.domain._np)

    def ConnectBlock(self, face, other: "MBBlock", other_face, axis1, axis2="-z"):
        lib = self.domain.lib
        d = self.domain.dims
        f1, f2, a1, a2 = face_index(face), face_index(other_face), face_index(axis1), (face_index(axis2) if d == 3 else 0)
        L.check(lib.fg_mb_connect(self.domain.handle, self.index, f1, other.index, f2, a1, a2))
        # the axes vectors both ConnectedBoundary objects get (ConnectBlocks, domain_structs.cpp:1080-1113), kept for domain I/O
        axes1, axes2 = [f2, a1], [f1]
        f1d, f2d = f1 >> 1, f2 >> 1
        if d == 2 or (a1 >> 1) == (f2d + 1) % d:
            axes2.append((((f1d + 1) % d) << 1) | (a1 & 1))
            swapped = False
        else:
            axes2.append((((f1d + 2) % d) << 1) | (a2 & 1))
            swapped = True
        if d == 3:
            axes1.append(a2)
            axes2.append(((((f1d + 2) % d) << 1) | (a2 & 1)) if not swapped else ((((f1d + 1) % d) << 1) | (a1 & 1)))
        self.connections[f1] = (other.index, axes1)
        other.connections[f2] = (self.index, axes2)

    def MakePeriodic(self, axis):
        a = _AXES[axis] if isinstance(axis, str) else int(axis)
        L.check(self.domain.lib.fg_mb_make_periodic(self.domain.handle, self.index, a))
        self.periodic_axes.add(a)

    # ---- views (after prepare_solve)
    @property
    def n_cells(self) -> int:
        return int(np.prod(self.size))

    def face_cells(self, face) -> int:
        f = face_index(face)
        return self.n_cells // self.size[f >> 1]

    def cells(self, field: torch.Tensor) -> torch.Tensor:
        """View of this block's cells of a flat field [B, C, N] as [B, C, (nz,) ny, nx]."""
        v = field[..., self.cell_offset:self.cell_offset + self.n_cells]
        return v.reshape(*field.shape[:-1], *reversed(self.size))

    def boundary(self, face, field: Optional[torch.Tensor] = None) -> torch.Tensor:
        """View of one FIXED face of the boundary-velocity array [B, d, NB] as [B, d, face cells] (lowest remaining axis
        fastest, i.e. the reference's boundary tensor flattened)."""
        f = face_index(face)
        s0 = self.boundary_slot0[f]
        if s0 < 0:
            raise ValueError(f"face {face} of block {self.name} is not a FIXED boundary")
        field = self.domain.boundary_velocity if field is None else field
        return field[..., s0:s0 + self.face_cells(f)]

    def getCellCoordinates(self) -> np.ndarray:
        """Cell centres = mean of the cell's vertices (``Block::getCellCoordinates``)."""
        c = self.coords
        d = c.shape[0]
        out = c
        for a in range(d):
            ax = out.ndim - 1 - a
            n = out.shape[ax] - 1
            out = 0.5 * (np.take(out, range(0, n), axis=ax) + np.take(out, range(1, n + 1), axis=ax))
        return out


def multilevel_tables(P, blocks, max_n4: int = 2048, max_n8: int = 512):
    """Host tables of the multilevel pressure preconditioner (pure NumPy / SciPy, no GPU).  ``P``: pressure matrix for ``A = 1``
    (SciPy sparse, cells of all blocks in flat order); ``blocks``: ``(nx, ny, cell_offset)`` per block.  Aggregates are tiles
    inside the blocks: about 8 x 8 cells (balanced when the size is not a multiple), each split into 2 x 2 sub-tiles of about
    4 x 4.  Returns ``a4`` [N], ``parent4`` [n4], ``d4`` = diag(Z4^T S Z4), ``aci8`` = pinv(Z8^T S Z8), ``geom_diag_sum`` =
    trace(S), with ``S`` the symmetric part of ``P``; None when the aggregate counts exceed the kernel's LDS tables."""
    import scipy.sparse as sp

    S = (0.5 * (P + P.T)).tocsr()
    N = S.shape[0]
    a4 = np.zeros(N, np.int64)
    parent4: List[int] = []
    rects: Dict[int, tuple] = {}
    n4 = n8 = 0
    for nx, ny, offset in blocks:
        g8x, g8y = max(1, -(-nx // 8)), max(1, -(-ny // 8))
        t8x, t8y = (np.arange(nx) * g8x) // nx, (np.arange(ny) * g8y) // ny          # balanced 8-tile index per cell column / row

        def halves(t8):
            h = np.zeros(len(t8), np.int64)
            for tile in np.unique(t8):
                idx = np.nonzero(t8 == tile)[0]
                h[idx[(len(idx) + 1) // 2:]] = 1                                      # second half of the tile
            return h

        t4x, t4y = 2 * t8x + halves(t8x), 2 * t8y + halves(t8y)
        local4 = (t4y[:, None] * (2 * g8x) + t4x[None, :]).reshape(-1)
        a4[offset: offset + nx * ny] = n4 + local4
        for ty in np.unique(t4y):                                                     # every sub-tile as a rectangle of cells
            ys = np.nonzero(t4y == ty)[0]
            for tx in np.unique(t4x):
                xs = np.nonzero(t4x == tx)[0]
                rects[n4 + ty * (2 * g8x) + tx] = (offset + ys[0] * nx + xs[0], len(xs), len(ys), nx)
        par = np.full(4 * g8x * g8y, -1, np.int64)
        par[local4] = (n8 + t8y[:, None] * g8x + t8x[None, :]).reshape(-1)
        parent4.extend(par.tolist())
        n4 += 4 * g8x * g8y
        n8 += g8x * g8y
    parent4 = np.asarray(parent4, np.int64)
    used = np.zeros(n4, bool)
    used[a4] = True                                                                   # sub-tiles no cell fell into (tiles 1 cell wide)
    remap = np.cumsum(used) - 1
    rect4 = np.array([rects[k] for k in np.nonzero(used)[0]], np.int64)                # [n4, 4]: first cell, width, height, row stride
    a4, parent4, n4 = remap[a4], parent4[used], int(used.sum())
    if n4 > max_n4 or n8 > max_n8:
        return None
    Z4 = sp.csr_matrix((np.ones(N), (np.arange(N), a4)), shape=(N, n4))
    Z8 = sp.csr_matrix((np.ones(n4), (np.arange(n4), parent4)), shape=(n4, n8))
    A4 = (Z4.T @ S @ Z4).tocsr()
    A8 = (Z8.T @ A4 @ Z8).toarray()
    # The constant is the null vector of the all-Neumann pressure matrix (P 1 = 0), and the coarse operator inherits it only
    # approximately: to the precision P was assembled in on orthogonal meshes (the GPU's fp32 matrix leaves that eigenvalue at
    # 1e-8..1e-7 of the largest, with either sign), and to 1e-5..1e-6 on non-orthogonal ones, where S = sym(P) has no exact null
    # vector at all (Airfoil2D: -2e-3 against -9e2).  Inverted, that eigenvalue becomes a 1e5..1e7-fold amplification of the coarse
    # constant -- eigenvalues of P M of -2.2 .. +0.03 where 0 belongs -- which wrecks the fp32 BiCGStab.  The coarse problem is
    # therefore solved on the complement of the coarse constant (Z 1_coarse = 1): Q A8^+ Q with Q = I - 1 1^T / n8, and what is
    # left below 1e-6 of the largest eigenvalue is cut (three orders under the smallest genuine one, ~ 1 / n8).
    singular = np.abs(P @ np.ones(N)).max() < 1e-3 * np.abs(P.diagonal()).max()
    if singular and n8 > 1:
        Q = np.eye(n8) - np.full((n8, n8), 1.0 / n8)
        aci8 = Q @ np.linalg.pinv(Q @ A8 @ Q, rcond=1e-6, hermitian=True) @ Q
    else:
        aci8 = np.linalg.pinv(A8, rcond=1e-6, hermitian=True)
    return {"a4": a4, "parent4": parent4, "rect4": rect4, "n4": n4, "n8": n8, "d4": A4.diagonal(), "aci8": aci8, "geom_diag_sum": float(S.diagonal().sum())}


class MultiBlockDomain:
    """``PISOtorch.Domain`` for connected curvilinear blocks, batched over envs."""

    def __init__(self, dims: int, viscosity: float, batch: int = 1, device: Optional[torch.device] = None,
                 reference_quirks: bool = True, non_ortho_flags: int = 25, dtype: torch.dtype = torch.float32):
        if not torch.cuda.is_available():
            raise L.NativeLibraryError("fluidgym_amd needs a GPU: the multi-block path has no CPU fallback")
        self._set_dtype(dtype)
        self.dims, self.batch = int(dims), int(batch)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.viscosity = float(viscosity)
        self.handle = ctypes.c_void_p()
        L.check(self.lib.fg_mb_create(self.dims, self.batch, self.device.index or 0, ctypes.byref(self.handle)))
        if not reference_quirks:
            L.check(self.lib.fg_mb_set_reference_quirks(self.handle, 0, 0))
        if non_ortho_flags != 25:
            L.check(self.lib.fg_mb_set_nonortho_flags(self.handle, int(non_ortho_flags)))
        self.non_ortho_flags = int(non_ortho_flags)
        self.blocks: List[MBBlock] = []
        self.prepared = False
        self.velocity = self.pressure = self.boundary_velocity = self.velocity_source = None
        self.n_cells = self.n_boundary_faces = 0
        self._dt = None
        self.multilevel = None   # aggregate counts once set_pressure_multilevel has installed the preconditioner

    def _set_dtype(self, dtype: torch.dtype) -> None:
        """float32: ``libfluidgym_hip.so``; float64: the fp64 build (``libfluidgym_hip_f64.so``), in which every float of the
        multi-block C ABI is a double (csrc/fg_mb.h) -- the one-cell-per-thread kernels and the plain recurrences only: no
        on-chip CG, no multilevel preconditioner there."""
        if dtype not in (torch.float32, torch.float64):
            raise ValueError("dtype must be torch.float32 or torch.float64")
        f64 = dtype == torch.float64
        self.dtype = dtype
        self.lib = L.load_f64() if f64 else L.load()
        self._np = np.float64 if f64 else np.float32
        self._cf = ctypes.c_double if f64 else ctypes.c_float
        self._step_opt_t = L.FgMbStepOptionsF64 if f64 else L.FgMbStepOptions
        self._sim_opt_t = L.FgMbSimOptionsF64 if f64 else L.FgMbSimOptions

    def CreateBlock(self, vertexCoordinates, name: str = "") -> MBBlock:
        c = vertexCoordinates.detach().cpu().numpy() if isinstance(vertexCoordinates, torch.Tensor) else np.asarray(vertexCoordinates)
        if c.ndim == self.dims + 2:  # reference layout [1, d, ...]
            c = c[0]
        c = np.ascontiguousarray(c, dtype=self._np)
        if c.shape[0] != self.dims or c.ndim != self.dims + 1:
            raise ValueError("vertexCoordinates must be [d, (nz+1,) ny+1, nx+1]")
        size = [c.shape[-1 - a] - 1 for a in range(self.dims)] + [1] * (3 - self.dims)
        bid = ctypes.c_int32(-1)
        L.check(self.lib.fg_mb_add_block(self.handle, c.ctypes.data_as(ctypes.POINTER(self._cf)), size[0], size[1],
                                         size[2], ctypes.byref(bid)))
        blk = MBBlock(self, bid.value, c, name or f"block{bid.value}")
        self.blocks.append(blk)
        return blk

    def PrepareSolve(self):
        """Build the mesh tables and allocate the fields (``Domain::PrepareSolve``, domain_structs.cpp:2570-2693)."""
        L.check(self.lib.fg_mb_finalize(self.handle))
        n, nb = ctypes.c_int32(), ctypes.c_int32()
        L.check(self.lib.fg_mb_sizes(self.handle, ctypes.byref(n), ctypes.byref(nb)))
        self.n_cells, self.n_boundary_faces = n.value, nb.value
        for blk in self.blocks:
            off = ctypes.c_int32()
            slots = (ctypes.c_int32 * 6)()
            L.check(self.lib.fg_mb_block_info(self.handle, blk.index, ctypes.byref(off), slots))
            blk.cell_offset = off.value
            blk.boundary_slot0 = [slots[f] for f in range(2 * self.dims)]
        B, d = self.batch, self.dims
        kw = dict(dtype=self.dtype, device=self.device)
        self.velocity = torch.zeros(B, d, self.n_cells, **kw)
        self.pressure = torch.zeros(B, self.n_cells, **kw)
        self.boundary_velocity = torch.zeros(B, d, max(self.n_boundary_faces, 1), **kw)
        self._dt = torch.zeros(B, **kw)
        for blk in self.blocks:
            for f, v in blk._pending_velocity.items():
                if blk.boundary_slot0[f] < 0:
                    raise ValueError(f"face {f} of {blk.name} was given a velocity but is connected or periodic")
                view = blk.boundary(f)
                t = torch.as_tensor(v, **kw).reshape(d, -1)
                view.copy_(t.expand(d, view.shape[-1]) if t.shape[-1] == 1 else t)
        L.check(self.lib.fg_mb_set_viscosity(self.handle, self.viscosity))
        self._bind()
        self.prepared = True

    def _bind(self):
        src = self.velocity_source
        L.check(self.lib.fg_mb_bind(self.handle, ctypes.c_void_p(self.velocity.data_ptr()),
                                    ctypes.c_void_p(self.pressure.data_ptr()),
                                    ctypes.c_void_p(self.boundary_velocity.data_ptr()),
                                    ctypes.c_void_p(src.data_ptr()) if src is not None else None))

    def set_velocity_source(self, source: Optional[torch.Tensor]):
        self.velocity_source = None if source is None else source.to(self.device, self.dtype).contiguous()
        self._bind()

    def neighbors(self) -> np.ndarray:
        out = np.zeros((2 * self.dims, self.n_cells), dtype=np.int32)
        L.check(self.lib.fg_mb_get_neighbors(self.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        return out

    # ---- stepping
    def piso_step(self, dt, corrector_steps: int = 2, advect_non_ortho_steps: int = 1, pressure_non_ortho_steps: int = 1,
                  advection_tol: float = 1e-5, pressure_tol: float = 1e-5, max_iterations: int = 5000,
                  raise_on_failure: bool = True, pressure_use_bicgstab: bool = False, pressure_warm_start: bool = False,
                  pressure_project_mean: bool = False, solver_double_fallback: bool = False,
                  bicg_precondition_fallback: bool = False):
        """One PISO step of every env (``dt``: scalar or [B]; ``dt <= 0`` leaves an env untouched).  Returns the max
        solver iterations (velocity, pressure corrector 0, pressure corrector 1)."""
        if not self.prepared:
            raise RuntimeError("PrepareSolve() first")
        self._dt.copy_(torch.as_tensor(dt, dtype=self.dtype).expand(self.batch))
        opt = self._step_opt_t(corrector_steps, advect_non_ortho_steps, pressure_non_ortho_steps, max_iterations,
                                advection_tol, pressure_tol, int(pressure_use_bicgstab), int(pressure_warm_start),
                                int(pressure_project_mean), 0.0, int(solver_double_fallback), int(bicg_precondition_fallback))
        stats = (ctypes.c_int32 * 4)()
        st = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.fg_mb_piso_step(self.handle, ctypes.c_void_p(self._dt.data_ptr()), ctypes.byref(opt), stats,
                                      ctypes.c_void_p(st))
        L.check(rc, allow=() if raise_on_failure else (L.FG_ERR_NOT_CONVERGED, L.FG_ERR_NOT_FINITE))
        return stats[1], stats[2], stats[3]

    def max_velocity(self) -> np.ndarray:
        out = (self._cf * self.batch)()
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.fg_mb_max_velocity(self.handle, out, ctypes.c_void_p(st)))
        return np.array(out[:], dtype=self._np)

    def buffer(self, which: int) -> torch.Tensor:
        """Copy of an intermediate buffer of the last step (tests)."""
        ptr, cnt = ctypes.c_void_p(), ctypes.c_int64()
        L.check(self.lib.fg_mb_get_buffer(self.handle, which, ctypes.byref(ptr), ctypes.byref(cnt)))
        out = torch.empty(cnt.value, dtype=self.dtype, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.fg_mb_read_buffer(self.handle, which, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st)))
        return out

    # ---- Simulation.single_step / make_divergence_free
    def boundary_flux_balance(self) -> np.ndarray:
        out = (self._cf * self.batch)()
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.fg_mb_boundary_flux_balance(self.handle, out, ctypes.c_void_p(st)))
        return np.array(out[:], dtype=self._np)

    def set_stall_limit(self, iterations: int) -> None:
        """Iterations a CG solve may go without improving its kept iterate before it ends with it (default 400)."""
        L.check(self.lib.fg_mb_set_stall_limit(self.handle, int(iterations)))

    def set_advection_start(self, from_result: bool = False) -> None:
        """Start vector of the first velocity solve of a step: zero (the reference) or the current velocity (opt-in)."""
        L.check(self.lib.fg_mb_set_advection_start(self.handle, int(from_result)))

    def set_advection_jacobi(self, on: bool = True) -> None:
        """Velocity systems by point-Jacobi sweeps before BiCGStab (``fg_mb_set_advection_jacobi``; policy ``advection_jacobi``)."""
        L.check(self.lib.fg_mb_set_advection_jacobi(self.handle, int(bool(on))))

    def advection_jacobi_counts(self) -> dict:
        out = (ctypes.c_int64 * 2)()
        L.check(self.lib.fg_mb_advection_jacobi_counts(self.handle, out))
        return {"settled_by_sweeps": int(out[0]), "handed_to_bicgstab": int(out[1])}

    def boundary_tables(self):
        """(owner cell [NB], face [NB], Minv|det [NB, d*d+1]) of the boundary slots (host arrays)."""
        nb, tw = self.n_boundary_faces, self.dims * self.dims + 1
        cell = np.zeros(max(nb, 1), np.int32)
        face = np.zeros(max(nb, 1), np.int32)
        T = np.zeros((max(nb, 1), tw), self._np)
        i32, f32 = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(self._cf)
        L.check(self.lib.fg_mb_get_boundary_tables(self.handle, cell.ctypes.data_as(i32), face.ctypes.data_as(i32),
                                                   T.ctypes.data_as(f32)))
        return cell[:nb], face[:nb], T[:nb]

    def cell_transforms(self) -> np.ndarray:
        T = np.zeros((self.n_cells, self.dims * self.dims + 1), self._np)
        L.check(self.lib.fg_mb_get_cell_transforms(self.handle, T.ctypes.data_as(ctypes.POINTER(self._cf))))
        return T

    def _step_options(self, corrector_steps, advect_non_ortho_steps, pressure_non_ortho_steps, advection_tol,
                      pressure_tol, max_iterations, pressure_use_bicgstab, pressure_warm_start=False,
                      pressure_project_mean=False, pressure_stall_accept=0.0, solver_double_fallback=False,
                      bicg_precondition_fallback=False):
        return self._step_opt_t(corrector_steps, advect_non_ortho_steps, pressure_non_ortho_steps, max_iterations,
                                 advection_tol, pressure_tol, int(pressure_use_bicgstab), int(pressure_warm_start),
                                 int(pressure_project_mean), float(pressure_stall_accept), int(solver_double_fallback),
                                 int(bicg_precondition_fallback))

    def _outflow_ranges(self, outflow):
        """``outflow``: one (block, face) or a list of up to two; returns [(slot0, count), (slot0_b, count_b)]."""
        faces = outflow if (isinstance(outflow, (list, tuple)) and isinstance(outflow[0], (list, tuple))) else [outflow]
        if len(faces) > 2:
            raise NotImplementedError("at most two outflow faces")
        r = [self._outflow_slots(f) for f in faces]
        return r + [(0, 0)] * (2 - len(r))

    def _outflow_slots(self, outflow):
        blk, face = outflow
        blk = blk if isinstance(blk, MBBlock) else self.blocks[blk]
        f = face_index(face)
        if blk.boundary_slot0[f] < 0:
            raise ValueError("the outflow face must be a FIXED boundary")
        return blk.boundary_slot0[f], blk.face_cells(f)

    def update_advective_boundary(self, dt: float, outflow, outflow_velocity: Sequence[float] = (1.0, 0.0, 0.0),
                                  tol: float = 5e-6):
        """``update_advective_boundaries`` + ``balance_boundary_fluxes`` for one face (the envs' PRE hook)."""
        (s0, n), (s1, n1) = self._outflow_ranges(outflow)
        v = (self._cf * 3)(*(list(outflow_velocity) + [0.0] * 3)[:3])
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.fg_mb_update_advective_boundary(self.handle, float(dt), s0, n, s1, n1, v, float(tol), ctypes.c_void_p(st)))

    def make_divergence_free(self, pressure_tol: float = 1e-5, max_iterations: int = 1000, pressure_non_ortho_steps: int = 1,
                             pressure_use_bicgstab: bool = False, outflow=None,
                             outflow_velocity: Sequence[float] = (1.0, 0.0, 0.0), outflow_tol: float = 5e-6,
                             pressure_project_mean: bool = False) -> bool:
        if outflow is not None:  # PRE hook with time_step = 1 (PISOtorch_simulation.py:1334-1345)
            self.update_advective_boundary(1.0, outflow, outflow_velocity, outflow_tol)
        opt = self._step_options(1, 1, pressure_non_ortho_steps, 1e-5, pressure_tol, max_iterations, pressure_use_bicgstab,
                                 pressure_project_mean=pressure_project_mean)
        st = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.fg_mb_make_divergence_free(self.handle, ctypes.byref(opt), ctypes.c_void_p(st))
        L.check(rc, allow=(L.FG_ERR_NOT_CONVERGED,))
        return rc == L.FG_OK

    def single_step(self, time_step: float, cfl: float = 0.8, adaptive: bool = True, substeps: int = 1, outflow=None,
                    outflow_velocity: Sequence[float] = (1.0, 0.0, 0.0), outflow_tol: float = 5e-6,
                    flux_balance_tol: float = 1e-5, corrector_steps: int = 2, advect_non_ortho_steps: int = 1,
                    pressure_non_ortho_steps: int = 1, advection_tol: float = 1e-5, pressure_tol: float = 1e-5,
                    max_iterations: int = 5000, pressure_use_bicgstab: bool = False, max_substeps: int = 0,
                    pressure_warm_start: bool = False, pressure_project_mean: bool = False,
                    pressure_stall_accept: float = 0.0, solver_double_fallback: bool = False,
                    bicg_precondition_fallback: bool = False):
        """``Simulation.single_step`` on the native side.  ``outflow``: (block, face) of the FIXED face that follows the
        convective outflow condition.  Returns (substeps, all solves converged, max iterations of the last substep)."""
        o = self._sim_opt_t()
        o.step = self._step_options(corrector_steps, advect_non_ortho_steps, pressure_non_ortho_steps, advection_tol,
                                    pressure_tol, max_iterations, pressure_use_bicgstab, pressure_warm_start,
                                    pressure_project_mean, pressure_stall_accept, solver_double_fallback,
                                    bicg_precondition_fallback)
        o.time_step, o.cfl, o.adaptive, o.substeps = float(time_step), float(cfl), int(adaptive), int(substeps)
        o.flux_balance_tol, o.outflow_tol, o.max_substeps = float(flux_balance_tol), float(outflow_tol), int(max_substeps)
        if outflow is not None:
            (o.outflow_slot0, o.outflow_count), (o.outflow_slot0_b, o.outflow_count_b) = self._outflow_ranges(outflow)
        v = list(outflow_velocity) + [0.0] * 3
        o.outflow_velm[0], o.outflow_velm[1], o.outflow_velm[2] = v[0], v[1], v[2]
        out = (ctypes.c_int32 * 6)()
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.fg_mb_single_step(self.handle, ctypes.byref(o), out, None, ctypes.c_void_p(st)))
        return out[4], bool(out[5]), (out[1], out[2], out[3])

    def ladder(self, force_mask: int = 0) -> dict:
        """How often each rung of the reference's retry ladder ran (``fg_mb_ladder``); ``force_mask`` (tests) makes first attempts
        count as failed: 1 velocity, 2 pressure, 4 also the velocity fp64 rung."""
        out = (ctypes.c_int64 * 4)()
        L.check(self.lib.fg_mb_ladder(self.handle, out, int(force_mask)))
        return {"velocity_fp64": out[0], "velocity_preconditioned": out[1], "pressure_fp64": out[2], "pressure_cg": out[3]}

    def env_status(self) -> np.ndarray:
        """Per-env outcome of the last step: 0 ok, 1 a solve ended on its best iterate, 2 non-finite solve -- that env's step
        was not committed (``fg_mb_env_status``)."""
        out = (ctypes.c_int32 * self.batch)()
        L.check(self.lib.fg_mb_env_status(self.handle, out))
        return np.array(out[:], dtype=np.int32)

    # ---- residual projection of the pressure CG
    def unit_pressure_matrix(self):
        """The pressure matrix for A = 1 (geometry only) of env 0 as a SciPy CSR matrix."""
        import scipy.sparse as sp

        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.fg_mb_unit_pressure_matrix(self.handle, ctypes.c_void_p(st)))
        N, F = self.n_cells, 2 * self.dims
        diag = self.buffer(L.FG_MB_BUF_P_DIAG)[:N].cpu().numpy().astype(np.float64)
        off = self.buffer(L.FG_MB_BUF_P_OFF).view(self.batch, F, N)[0].cpu().numpy().astype(np.float64)
        nbr = self.neighbors()
        rows, cols, vals = [np.arange(N)], [np.arange(N)], [diag]
        for f in range(F):
            ok = nbr[f] >= 0
            rows.append(np.nonzero(ok)[0]); cols.append(nbr[f][ok]); vals.append(off[f][ok])
        return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N))

    def set_pressure_multilevel(self, enable: bool = True) -> Optional[dict]:
        """Build and install the additive multilevel preconditioner of the on-chip pressure CG (``fg_mb_set_multilevel``): Jacobi on
        the cells + half-weighted Jacobi on 4 x 4 aggregates + the exact (pseudo-)inverse on 8 x 8 aggregates, aggregates being
        tiles inside the blocks (:func:`multilevel_tables`).  Everything comes from the symmetric part ``S`` of the pressure matrix
        for ``A = 1`` -- geometry only, so it is built ONCE per mesh and shared by all envs; a per-env scale accounts for
        ``P = S / A`` with ``A`` nearly constant (the same argument as the single-block path's constant-coefficient
        preconditioner).  The reference runs CG without a preconditioner (cg_solver_kernel.cu:129-471);
        converged answers agree to the solver tolerance, iteration counts drop 3-4x (profiles/r02_*).  Two consumers: the on-chip
        CG (meshes up to 16 k cells, 2048 / 512 aggregates: its LDS budget) applies it inside the persistent kernel; the pressure
        BiCGStab of larger 2-D meshes (Airfoil2D: 46.7 k cells) takes it as a right preconditioner in kernel form (three launches
        per application, csrc/fg_mb_step.hip::mb_ml_apply), up to 2048 coarse aggregates.  Returns the aggregate counts, or None
        when the mesh does not qualify (3-D, or too many aggregates; nothing is installed)."""
        if self.dims != 2 or self.dtype != torch.float32:   # (the fp64 build runs the plain recurrences)
            return None
        if not enable:
            L.check(self.lib.fg_mb_set_multilevel(self.handle, 0, 0, None, None, None, None, None, 0.0, 0))
            return None
        P = self.unit_pressure_matrix().astype(np.float64)
        tab = multilevel_tables(P, [(b.size[0], b.size[1], b.cell_offset) for b in self.blocks], max_n4=65534, max_n8=2048)
        if tab is None:
            return None
        i32, f32 = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float)
        a4_32, p4_32 = np.ascontiguousarray(tab["a4"], np.int32), np.ascontiguousarray(tab["parent4"], np.int32)
        d4_32, aci_32 = np.ascontiguousarray(tab["d4"], np.float32), np.ascontiguousarray(tab["aci8"], np.float32)
        rect_32 = np.ascontiguousarray(tab["rect4"], np.int32)
        L.check(self.lib.fg_mb_set_multilevel(self.handle, tab["n4"], tab["n8"], a4_32.ctypes.data_as(i32), p4_32.ctypes.data_as(i32),
                                              rect_32.ctypes.data_as(i32), d4_32.ctypes.data_as(f32), aci_32.ctypes.data_as(f32),
                                              float(tab["geom_diag_sum"]), 1))
        self.multilevel = {"n4": tab["n4"], "n8": tab["n8"]}
        self._multilevel_tables = tab
        return self.multilevel

    def wall_forces(self, cell_index: torch.Tensor, slot_index: torch.Tensor, geom: torch.Tensor, area_scale: float,
                    viscosity: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``[B, 2, layers]`` force on a closed wall from the current fields (``fg_mb_wall_forces``): ``cell_index`` /
        ``slot_index`` int32 ``[layers, n]`` in ring order, ``geom`` float32 ``[5, n]`` (normal x, y, tangential spacing, wall
        distance, face length).  Asynchronous on the current stream."""
        layers, n = cell_index.shape
        if out is None:
            out = torch.empty(self.batch, 2, layers, dtype=self.dtype, device=self.device)
        elif tuple(out.shape) != (self.batch, 2, layers) or out.dtype != self.dtype or not out.is_contiguous() or out.device != self.device:
            raise ValueError("wall_forces: out must be a contiguous [B, 2, layers] tensor of the domain's dtype on its device")
        geom = geom if geom.dtype == self.dtype else geom.to(self.dtype)
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.fg_mb_wall_forces(self.handle, ctypes.c_void_p(cell_index.data_ptr()), ctypes.c_void_p(slot_index.data_ptr()),
                                           ctypes.c_void_p(geom.data_ptr()), int(n), int(layers), float(area_scale), float(viscosity),
                                           ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st)))
        return out

    def multilevel_status(self) -> dict:
        """The pressure BiCGStab's trial of the multilevel right preconditioner: current back-off (solves run plain after a failed
        attempt; 4 = every attempt converges, 0 = no tables), attempts, failed attempts."""
        out = (ctypes.c_int32 * 3)()
        L.check(self.lib.fg_mb_multilevel_status(self.handle, out))
        return {"backoff": int(out[0]), "attempts": int(out[1]), "failed_attempts": int(out[2])}

    def ilu_apply(self, r: torch.Tensor) -> torch.Tensor:
        """``z = U^-1 L^-1 r`` [B, d, N] with ILU(0) of the velocity matrix the last step assembled (``fg_mb_debug_ilu_apply``): the
        preconditioner of the ``BiCG_precondition_fallback`` rung (tests)."""
        r = r.to(self.device, self.dtype).contiguous()
        z = torch.empty_like(r)
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.fg_mb_debug_ilu_apply(self.handle, ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(z.data_ptr()), ctypes.c_void_p(st)), lib=self.lib)
        return z

    def multilevel_apply(self, r: torch.Tensor) -> torch.Tensor:
        """``z = M r`` [B, N] with the kernel form of the multilevel preconditioner on the pressure matrix currently assembled."""
        r = r.to(self.device, self.dtype).contiguous()
        z = torch.empty_like(r)
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.fg_mb_multilevel_apply(self.handle, ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(z.data_ptr()), ctypes.c_void_p(st)))
        return z

    def set_pressure_deflation(self) -> float:
        """Keep the pressure-CG residuals orthogonal to the LEFT near-null vector of the pressure matrix instead of the
        constant (``pressure_project_mean``).  With cross-metric terms the matrix is not symmetric and that vector is not
        constant, so a flux-balanced right-hand side retains a component along it that no iteration removes -- the residual
        floor that sits at the envs' tolerance on the reference's cylinder mesh.  The vector depends on the geometry almost
        only (cos > 0.999999 between A = 1 and a developed flow's A), so it is computed once, by shift-invert Arnoldi on the
        A = 1 matrix.  Returns the cosine between that vector and the constant (1 on orthogonal meshes)."""
        import scipy.sparse.linalg as spla

        P = self.unit_pressure_matrix()
        N = P.shape[0]
        try:
            w, V = spla.eigs(P.T.tocsc(), k=1, sigma=0.0, which="LM", tol=1e-10)
            y = np.real(V[:, 0])
        except Exception:  # singular factorisation on an exactly singular (orthogonal) matrix: the constant is the answer
            y = np.ones(N)
        y = y / np.linalg.norm(y)
        if y.sum() < 0:
            y = -y
        res = np.linalg.norm(P.T @ y) / max(abs(P).max(), 1e-30)
        if not np.isfinite(res) or res > 1e-3:
            y = np.ones(N) / np.sqrt(N)
        y32 = np.ascontiguousarray(y, dtype=self._np)
        L.check(self.lib.fg_mb_set_residual_projection(self.handle, y32.ctypes.data_as(ctypes.POINTER(self._cf))))
        return float(y.sum() / np.sqrt(N))

    def solver_counters(self, reset: bool = False) -> dict:
        """Iterations of the linear solves since the last reset: per kind (scalar, velocity, pressure corrector 0 / 1) the
        mean and max per system (env x component) and the number of PISO steps (``fg_mb_solver_counters``)."""
        unconv = (ctypes.c_int64 * 4)()
        L.check(self.lib.fg_mb_solver_unconverged(self.handle, unconv))      # (read before the counters are cleared)
        out = (ctypes.c_int64 * 13)()
        L.check(self.lib.fg_mb_solver_counters(self.handle, out, int(reset)))
        names = ("scalar", "velocity", "pressure0", "pressure1")
        res = {n: {"mean": (out[k] / out[4 + k]) if out[4 + k] else None, "max": int(out[8 + k]), "systems": int(out[4 + k]),
                    "unconverged": int(unconv[k])}
               for k, n in enumerate(names)}
        res["piso_steps"] = int(out[12])
        return res

    def config_dump(self) -> dict:
        """The switches this handle runs under (``fg_mb_config_dump``) plus the process environment's FG_* / FLUIDGYM_* variables."""
        import json
        import os

        buf = ctypes.create_string_buffer(4096)
        L.check(self.lib.fg_mb_config_dump(self.handle, buf, 4096))
        out = json.loads(buf.value.decode())
        out["env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("FG_", "FLUIDGYM_"))}
        return out

    # ---- live kernel timing (bench.py)
    def profile_enable(self, on: bool = True) -> None:
        L.check(self.lib.fg_mb_profile_enable(self.handle, int(on)))

    def profile_read(self) -> dict:
        out = {}
        for kind in range(3):
            ms, nb = ctypes.c_double(), ctypes.c_double()
            n, launches = ctypes.c_int64(), ctypes.c_int64()
            L.check(self.lib.fg_mb_profile_read(self.handle, kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(nb),
                                                ctypes.byref(launches)))
            out[self.lib.fg_mb_profile_kind_name(kind).decode()] = {"ms": ms.value, "samples": n.value, "bytes": nb.value,
                                                                    "launches": launches.value}
        its = ctypes.c_int64()
        L.check(self.lib.fg_mb_profile_iterations(self.handle, ctypes.byref(its)))
        out["k_mbc_onchip"]["iterations"] = its.value
        return out

    # ---- what FluidEnv needs of a Domain
    def solver_hints(self, values=None) -> torch.Tensor:
        """The 48 words a handle remembers between solves (``fg_mb_solver_hints``): read, or written from ``values`` (a 36-word
        snapshot of rounds 3-5 is padded: the sweeps' back-off starts cleared)."""
        vals = [0] * 48 if values is None else ([int(v) for v in values] + [0] * 48)[:48]
        buf = (ctypes.c_int32 * 48)(*vals)
        L.check(self.lib.fg_mb_solver_hints(self.handle, buf, 0 if values is None else 1))
        return torch.tensor(list(buf), dtype=torch.int32)

    def Clone(self) -> dict:
        """State snapshot of ``FluidEnv.get_state`` (reference ``Domain.Clone()``): the bound fields and the solver's hints, so
        that ``set_state`` + ``step`` replays bit for bit (envs/fluid_env.py:1320-1363)."""
        return {"velocity": self.velocity.clone(), "pressure": self.pressure.clone(),
                "boundary_velocity": self.boundary_velocity.clone(), "solver_hints": self.solver_hints()}

    def Restore(self, snap: dict) -> None:
        self.velocity.copy_(snap["velocity"])
        self.pressure.copy_(snap["pressure"])
        self.boundary_velocity.copy_(snap["boundary_velocity"])
        if "solver_hints" in snap:
            self.solver_hints(snap["solver_hints"].tolist())

    @property
    def solver(self):
        return self

    def reset_solver_state(self) -> None:
        """Nothing is carried between steps besides the bound fields (the pressure result IS the pressure field)."""

    def getBlock(self, i: int) -> MBBlock:
        return self.blocks[i]

    def close(self):
        if self.handle:
            self.lib.fg_mb_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiBlockSimulation:
    """``Simulation`` (simulation/simulation.py:124-280) for a :class:`MultiBlockDomain`: the settings of the reference's
    constructor that change results, ``single_step`` and ``make_divergence_free``; the loop itself runs natively
    (``fg_mb_single_step``)."""

    def __init__(self, domain: MultiBlockDomain, dt: float, adaptive_CFL: float = 0.8, substeps="ADAPTIVE",
                 corrector_steps: int = 2, advection_tol: Optional[float] = None, pressure_tol: Optional[float] = None,
                 advect_non_ortho_steps: int = 1, pressure_non_ortho_steps: int = 1, max_iterations: int = 5000,
                 pressure_use_BiCG: bool = False, outflow=None, outflow_velocity: Sequence[float] = (1.0, 0.0, 0.0),
                 outflow_tol: float = 5e-6, flux_balance_tol: float = 1e-5, pressure_warm_start: Optional[bool] = None,
                 pressure_project_mean: bool = True, pressure_stall_accept: Optional[float] = None,
                 solver_double_fallback: bool = False, BiCG_precondition_fallback: bool = True,
                 advection_warm_start: Optional[bool] = None):
        # the reference's retry ladder (Simulation attributes set by the envs, e.g. cylinder_env_base.py:326-328)
        self.solver_double_fallback, self.BiCG_precondition_fallback = bool(solver_double_fallback), bool(BiCG_precondition_fallback)
        from .policy import get_solver_policy

        pol = get_solver_policy()   # reference behaviour unless asked otherwise: cold start, no stall acceptance
        self.pressure_warm_start = bool(pol["pressure_warm_start"] if pressure_warm_start is None else pressure_warm_start)
        self.pressure_stall_accept = float(pol["pressure_stall_accept"] if pressure_stall_accept is None else pressure_stall_accept)
        # first velocity solve of a step from zero as the reference's non-orthogonal branch (policy.py), or from the current velocity
        self.advection_warm_start = bool(pol["advection_warm_start"] if advection_warm_start is None else advection_warm_start)
        domain.set_advection_start(self.advection_warm_start)
        # policy advection_jacobi (default on): the velocity systems go to the Jacobi sweeps first (mb_jacobi); meshes they do not
        # contract on (the airfoil's) are handed to BiCGStab after the first check and the handle backs off
        self.advection_jacobi = bool(pol["advection_jacobi"])
        if hasattr(domain, "set_advection_jacobi"):
            domain.set_advection_jacobi(self.advection_jacobi)
        self.pressure_project_mean = pressure_project_mean
        self.domain, self.time_step, self.adaptive_CFL, self.substeps = domain, float(dt), float(adaptive_CFL), substeps
        self.corrector_steps = corrector_steps
        self.advection_tol = 1e-5 if advection_tol is None else advection_tol   # _get_solver_tolerance (PISOtorch_diff.py:247-253)
        self.pressure_tol = 1e-5 if pressure_tol is None else pressure_tol
        self.advect_non_ortho_steps, self.pressure_non_ortho_steps = advect_non_ortho_steps, pressure_non_ortho_steps
        self.max_iterations, self.pressure_use_BiCG = max_iterations, pressure_use_BiCG
        self.outflow, self.outflow_velocity, self.outflow_tol = outflow, tuple(outflow_velocity), outflow_tol
        self.flux_balance_tol = flux_balance_tol
        self.total_time, self.total_step, self.last_substeps, self.last_iterations = 0.0, 0, 0, (0, 0, 0)

    def make_divergence_free(self) -> bool:
        # the reference ends the call with end_step(time_step = 1): the counters advance (PISOtorch_simulation.py:1334, 1427)
        self.total_time += 1.0
        self.total_step += 1
        return self.domain.make_divergence_free(pressure_tol=self.pressure_tol, pressure_non_ortho_steps=self.pressure_non_ortho_steps,
                                                pressure_use_bicgstab=self.pressure_use_BiCG, outflow=self.outflow,
                                                outflow_velocity=self.outflow_velocity, outflow_tol=self.outflow_tol,
                                                pressure_project_mean=self.pressure_use_BiCG and self.pressure_project_mean)

    def single_step(self) -> bool:
        adaptive = self.substeps in ("ADAPTIVE", -1)
        n, ok, its = self.domain.single_step(
            self.time_step, cfl=self.adaptive_CFL, adaptive=adaptive, substeps=1 if adaptive else int(self.substeps),
            outflow=self.outflow, outflow_velocity=self.outflow_velocity, outflow_tol=self.outflow_tol,
            flux_balance_tol=self.flux_balance_tol, corrector_steps=self.corrector_steps,
            advect_non_ortho_steps=self.advect_non_ortho_steps, pressure_non_ortho_steps=self.pressure_non_ortho_steps,
            advection_tol=self.advection_tol, pressure_tol=self.pressure_tol, max_iterations=self.max_iterations,
            pressure_use_bicgstab=self.pressure_use_BiCG, pressure_warm_start=self.pressure_warm_start,
            pressure_project_mean=self.pressure_project_mean, pressure_stall_accept=self.pressure_stall_accept,
            solver_double_fallback=self.solver_double_fallback, bicg_precondition_fallback=self.BiCG_precondition_fallback)
        self.total_time += self.time_step
        self.total_step += 1
        self.last_substeps, self.last_iterations = n, its
        # unconverged solves hand back their best iterate (pressure_return_best_result=True), as the envs ask; an env whose
        # solve was NON-FINITE keeps its pre-step state while the rest of the batch completes, and the step reports False
        # like the reference's (simulation.py:259-280) -- the reference's envs carry on regardless (cylinder_env_base.py:755)
        self.last_env_status = self.domain.env_status() if not ok else np.zeros(self.domain.batch, np.int32)
        if (self.last_env_status == 2).any():
            import logging
            logging.getLogger("PISOsim").error("Simulation failed in step (total step %d): non-finite linear solve in envs %s; "
                                               "their state was left unchanged", self.total_step,
                                               np.nonzero(self.last_env_status == 2)[0].tolist())
            return False
        return True
