"""CPU oracle: NumPy/SciPy restatement of the reference's PISO stepper.

TEST INFRASTRUCTURE ONLY.  Nothing in ``fluidgym_amd`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and only as the checker.  The product path is the HIP library (``fluidgym_amd/csrc``).

PARITY UNPINNED: the reference (safe-autonomous-systems/fluidgym v0.1.2) has no CPU
path (every native entry point does CHECK_INPUT_CUDA), cannot be built here (CUDA +
cuSPARSE/cuBLAS, not vendored, no nvcc) and its own tests hold no golden vectors for the
solver.  This file therefore restates the algorithm from the
reference sources line by line; what *is* pinned against importable reference Python
(grid generators, inflow profiles) is pinned in ``tests/golden`` (see
``tests/golden/make_golden.py``).

Scope: single block, per-cell *orthogonal* (diagonal) transforms, boundaries
PERIODIC or FIXED (velocity Dirichlet; passive scalar Dirichlet / Neumann), 2-D and 3-D.
On such grids every non-orthogonal term of the reference is multiplied by a zero cross
metric (``PISO_multiblock_cuda_kernel.cu:3772, 4904``) and vanishes.

All file:line citations are relative to
``/root/reference/src/fluidgym/simulation``; ``K.cu`` abbreviates
``extensions/PISO_multiblock_cuda_kernel.cu`` and ``SIM.py`` abbreviates
``pict/PISOtorch_simulation.py``.

Array conventions: cell fields are C-ordered ``[(nz,) ny, nx]``; vector fields carry the
component as the leading axis (component 0 = x).  Spatial axis ``a`` (0=x,1=y,2=z) is the
NumPy axis ``-1-a``.  Faces are numbered ``-x,+x,-y,+y,-z,+z`` = ``0..2d-1``
(``K.cu:211-236``): axis = face>>1, upper = face&1.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

DIRICHLET = 0
NEUMANN = 1


# --------------------------------------------------------------------------------------
# grid metrics
# --------------------------------------------------------------------------------------
def _ax(a: int) -> int:
    """NumPy axis of spatial axis ``a`` in a cell/vertex array without leading channels."""
    return -1 - a


def coords_to_transforms(coords: np.ndarray):
    """Per-cell transform ``M | Minv | det`` from vertex coordinates.

    Follows ``extensions/grid_gen.cu:298-354`` (``k_CoordsToTransforms``):
    ``M[i][k] = x_i(centre of face +k) - x_i(centre of face -k)`` where a face centre is
    the mean of its ``2^(d-1)`` vertices.

    coords: ``[d, (nz+1,) ny+1, nx+1]``.  Returns ``M, Minv`` with shape
    ``[(nz,) ny, nx, d, d]`` and ``det`` with shape ``[(nz,) ny, nx]``.
    """
    d = coords.shape[0]
    assert coords.ndim == d + 1
    cell_shape = tuple(s - 1 for s in coords.shape[1:])
    M = np.zeros(cell_shape + (d, d), dtype=coords.dtype)
    for k in range(d):
        # mean over the vertices of face (+k) minus mean over those of face (-k)
        hi = np.zeros((d,) + cell_shape, dtype=coords.dtype)
        lo = np.zeros((d,) + cell_shape, dtype=coords.dtype)
        for vert in range(1 << d):
            sl = [slice(None)]
            for a in reversed(range(d)):  # numpy order z,y,x
                off = (vert >> a) & 1
                sl.append(slice(off, off + cell_shape[_ax(a)]))
            v = coords[tuple(sl)]
            if (vert >> k) & 1:
                hi += v
            else:
                lo += v
        norm = 1.0 / (1 << (d - 1))
        diff = (hi - lo) * norm  # [d(i), cells]
        for i in range(d):
            M[..., i, k] = diff[i]
    det = np.linalg.det(M)
    Minv = np.linalg.inv(M)
    return M, Minv, det


def boundary_face_transform(coords: np.ndarray, face: int):
    """Face transform sliced at boundary ``face`` (what ``FixedBoundary.transform`` holds).

    Follows ``extensions/grid_gen.cu:398-494`` (``k_CoordsToFaceTransforms``) restricted to
    the boundary plane, as sliced by ``domain_structs.cpp:1825-1851``
    (``GetFaceTransformBoundarySlice``): the face-normal column is the one-sided
    difference between the two vertex planes of the first cell (``normWeight = 1`` at a
    bound, ``:427-434``), tangential columns are edge differences averaged over the
    face's edges (``:456-477``).

    Returns ``M, Minv [slab, d, d]`` and ``det [slab]`` where slab is the cell shape with
    extent 1 along the face axis.
    """
    d = coords.shape[0]
    axis = face >> 1
    upper = face & 1
    nvert = coords.shape[1:]
    n_axis = nvert[_ax(axis)] - 1  # cells along axis
    # vertex planes bounding the first / last cell
    p_lo = 0 if not upper else n_axis - 1
    p_hi = p_lo + 1
    # the face itself sits on plane p_face
    p_face = 0 if not upper else n_axis
    slab_shape = [s - 1 for s in nvert]
    slab_shape[_ax(axis) % d] = 1
    slab_shape = tuple(slab_shape)
    M = np.zeros(slab_shape + (d, d), dtype=coords.dtype)
    tang = [(axis + i) % d for i in range(1, d)]
    n_face_vert = 1 << (d - 1)

    def plane_vertices(plane: int, bits: int):
        sl = [slice(None)] * (d + 1)
        sl[_ax(axis)] = slice(plane, plane + 1)
        for j, t in enumerate(tang):
            off = (bits >> j) & 1
            sl[_ax(t)] = slice(off, off + nvert[_ax(t)] - 1)
        return coords[tuple(sl)]

    # normal column
    col = np.zeros((d,) + slab_shape, dtype=coords.dtype)
    for bits in range(n_face_vert):
        col += plane_vertices(p_hi, bits) - plane_vertices(p_lo, bits)
    col *= 1.0 / n_face_vert
    for i in range(d):
        M[..., i, axis] = col[i]
    # tangential columns: edges lying in the boundary plane
    n_edge_vert = max(1, 1 << (d - 2))
    for j, t in enumerate(tang):
        col = np.zeros((d,) + slab_shape, dtype=coords.dtype)
        for bits in range(n_face_vert):
            sign = 1.0 if (bits >> j) & 1 else -1.0
            col += sign * plane_vertices(p_face, bits)
        col *= 1.0 / n_edge_vert
        for i in range(d):
            M[..., i, t] = col[i]
    det = np.linalg.det(M)
    Minv = np.linalg.inv(M)
    return M, Minv, det


def rectilinear_coords(edges: Sequence[np.ndarray], dtype=np.float64) -> np.ndarray:
    """Vertex coordinates ``[d, (nz+1,) ny+1, nx+1]`` of a tensor-product grid from per-axis
    vertex positions ``edges = [x_edges, y_edges(, z_edges)]``."""
    d = len(edges)
    mesh = np.meshgrid(*[np.asarray(e, dtype=dtype) for e in reversed(edges)], indexing="ij")
    # mesh[0] varies along z (or y in 2-D) ... mesh[-1] along x
    return np.stack([mesh[d - 1 - a] for a in range(d)], axis=0)


@dataclass
class FixedBC:
    """A FIXED boundary (``domain_structs.h`` ``FixedBoundary``): Dirichlet velocity, and per
    passive-scalar channel a Dirichlet value or a Neumann gradient.

    ``velocity``: ``[d]`` (static) or ``[d, slab]`` (varying), slab = cell shape with extent 1
    on the face axis.  ``scalar``: ``[C]`` or ``[C, slab]``.
    """

    velocity: np.ndarray
    scalar: Optional[np.ndarray] = None
    scalar_types: Optional[Sequence[int]] = None  # DIRICHLET / NEUMANN per channel


class Grid:
    """Single block with per-cell orthogonal metrics and its boundary transforms."""

    def __init__(self, coords: np.ndarray, ortho_tol: float = 1e-9):
        coords = np.asarray(coords)
        if coords.ndim >= 2 and coords.shape[0] == 1 and coords.shape[1] == coords.ndim - 2:
            coords = coords[0]  # strip reference batch dim [1,d,...]
        self.coords = coords
        self.dims = d = coords.shape[0]
        self.shape = tuple(s - 1 for s in coords.shape[1:])  # numpy order
        self.n = int(np.prod(self.shape))
        self.M, self.Minv, self.det = coords_to_transforms(coords)
        off = self.Minv.copy()
        for a in range(d):
            off[..., a, a] = 0
        scale = np.abs(self.Minv).max()
        if np.abs(off).max() > ortho_tol * scale:
            raise ValueError("oracle supports orthogonal (diagonal) transforms only")
        self.b_M, self.b_Minv, self.b_det = {}, {}, {}
        for f in range(2 * d):
            self.b_M[f], self.b_Minv[f], self.b_det[f] = boundary_face_transform(coords, f)

    def size(self, a: int) -> int:
        return self.shape[_ax(a)]

    # Laplace coefficient alpha_a = det * |Minv_row_a|^2  (K.cu:1224-1239)
    def alpha(self, a: int) -> np.ndarray:
        row = self.Minv[..., a, :]
        return self.det * np.sum(row * row, axis=-1)

    def alpha_b(self, face: int) -> np.ndarray:  # K.cu:1468-1481
        a = face >> 1
        row = self.b_Minv[face][..., a, :]
        return self.b_det[face] * np.sum(row * row, axis=-1)

    def contravariant(self, u: np.ndarray, a: int) -> np.ndarray:
        """``U_a = det * (Minv_row_a . u)`` (K.cu:495-510)."""
        acc = np.zeros(self.shape, dtype=u.dtype)
        for c in range(self.dims):
            acc = acc + self.Minv[..., a, c] * u[c]
        return self.det * acc

    def contravariant_b(self, ub: np.ndarray, face: int) -> np.ndarray:
        """Same on a boundary slab with the boundary transform (K.cu:525-537)."""
        a = face >> 1
        Mi = self.b_Minv[face]
        acc = np.zeros(Mi.shape[:-2], dtype=ub.dtype)
        for c in range(self.dims):
            acc = acc + Mi[..., a, c] * ub[c]
        return self.b_det[face] * acc

    def slab_shape(self, face: int):
        s = list(self.shape)
        s[_ax(face >> 1) % self.dims] = 1
        return tuple(s)

    def cell_centers(self) -> np.ndarray:
        d = self.dims
        acc = np.zeros((d,) + self.shape, dtype=self.coords.dtype)
        for vert in range(1 << d):
            sl = [slice(None)]
            for a in reversed(range(d)):
                off = (vert >> a) & 1
                sl.append(slice(off, off + self.shape[_ax(a)]))
            acc += self.coords[tuple(sl)]
        return acc / (1 << d)


# --------------------------------------------------------------------------------------
# state
# --------------------------------------------------------------------------------------
@dataclass
class Domain:
    """Mirror of the parts of ``Domain``/``Block`` the path reads (Appendix B of SURVEY.md)."""

    grid: Grid
    viscosity: float
    velocity: np.ndarray  # [d, cells]
    pressure: np.ndarray  # [cells]
    bc: Dict[int, Optional[FixedBC]] = field(default_factory=dict)  # face -> FixedBC | None (periodic)
    scalar: Optional[np.ndarray] = None  # [C, cells]
    scalar_viscosity: Optional[Sequence[float]] = None  # per channel (or length 1)
    velocity_source: Optional[np.ndarray] = None  # [d] or [d, cells]
    # per-cell viscosity of the velocity system (Block.setViscosity; getViscosityBlock, K.cu:1816-1837): [cells] or None = `viscosity`
    viscosity_field: Optional[np.ndarray] = None
    # solver vectors kept between steps (domain.velocityResult / pressureResult)
    velocity_result: Optional[np.ndarray] = None
    pressure_result: Optional[np.ndarray] = None

    def __post_init__(self):
        d = self.grid.dims
        for f in range(2 * d):
            self.bc.setdefault(f, None)
        # CloseBoundary also closes the periodic partner (domain_structs.cpp:1981-2002)
        for a in range(d):
            lo, hi = self.bc[2 * a], self.bc[2 * a + 1]
            if (lo is None) != (hi is None):
                raise ValueError("a FIXED face needs a FIXED partner on the opposite side")
        if self.velocity_result is None:
            self.velocity_result = self.velocity.copy()
        if self.pressure_result is None:
            self.pressure_result = self.pressure.copy()

    @property
    def dims(self):
        return self.grid.dims

    def is_fixed(self, face: int) -> bool:
        return self.bc[face] is not None

    def bvel(self, face: int) -> np.ndarray:
        """Boundary velocity broadcast to ``[d, slab]`` (static -> ``data[c]``,
        varying -> ``data[flatten(pos with pos[axis]=0)]``; K.cu:608-626)."""
        g = self.grid
        v = np.asarray(self.bc[face].velocity)
        if v.ndim == 1:
            v = v.reshape((g.dims,) + (1,) * g.dims)
        return np.broadcast_to(v, (g.dims,) + g.slab_shape(face))

    def bscalar(self, face: int) -> np.ndarray:
        g = self.grid
        C = self.scalar.shape[0]
        s = self.bc[face].scalar
        if s is None:
            s = np.zeros((C,), dtype=self.scalar.dtype)
        s = np.asarray(s)
        if s.ndim == 1:
            s = s.reshape((C,) + (1,) * g.dims)
        return np.broadcast_to(s, (C,) + g.slab_shape(face))

    def bscalar_type(self, face: int, ch: int) -> int:
        t = self.bc[face].scalar_types
        return DIRICHLET if t is None else int(t[ch])

    def kappa(self, ch: int) -> float:
        """``getViscosity(domain, forPassiveScalar=True, ch)`` (K.cu:1803-1815)."""
        if self.scalar_viscosity is None:
            return self.viscosity
        sv = np.atleast_1d(self.scalar_viscosity)
        return float(sv[0] if sv.size == 1 else sv[ch])

    def source(self) -> np.ndarray:
        g = self.grid
        if self.velocity_source is None:
            return np.zeros((g.dims,) + g.shape, dtype=self.velocity.dtype)
        s = np.asarray(self.velocity_source)
        if s.ndim == 1:
            s = s.reshape((g.dims,) + (1,) * g.dims)
        return np.broadcast_to(s, (g.dims,) + g.shape)

    def copy(self) -> "Domain":
        import copy

        return copy.deepcopy(self)


# --------------------------------------------------------------------------------------
# helpers: neighbours, masks, boundary scatter
# --------------------------------------------------------------------------------------
def _nbr(q: np.ndarray, a: int, s: int) -> np.ndarray:
    """Value at the neighbour across face (a, s) with periodic wrap (K.cu:1625-1641)."""
    return np.roll(q, -s, axis=_ax(a))


def _at_bound(g: Grid, face: int) -> np.ndarray:
    """``isAtBound`` (K.cu:241-246) as a boolean cell mask."""
    a, upper = face >> 1, face & 1
    idx = np.arange(g.size(a))
    m = (idx == g.size(a) - 1) if upper else (idx == 0)
    shp = [1] * g.dims
    shp[_ax(a) % g.dims] = g.size(a)
    return np.broadcast_to(m.reshape(shp), g.shape)


def _prescribed(dom: Domain, face: int) -> np.ndarray:
    """at bound && isEmptyBound (FIXED) (K.cu:203-209, 3713-3714)."""
    if dom.is_fixed(face):
        return _at_bound(dom.grid, face)
    return np.zeros(dom.grid.shape, dtype=bool)


def _slab_to_cells(g: Grid, face: int, slab: np.ndarray) -> np.ndarray:
    """Scatter a boundary-slab array onto a zero cell array at the boundary layer."""
    a, upper = face >> 1, face & 1
    out = np.zeros(slab.shape[: slab.ndim - g.dims] + g.shape, dtype=slab.dtype)
    sl = [slice(None)] * out.ndim
    sl[_ax(a)] = slice(g.size(a) - 1, g.size(a)) if upper else slice(0, 1)
    out[tuple(sl)] = slab
    return out


def _cells_slab(g: Grid, face: int, q: np.ndarray) -> np.ndarray:
    """Boundary-adjacent cell layer of ``q`` as a slab."""
    a, upper = face >> 1, face & 1
    sl = [slice(None)] * q.ndim
    sl[_ax(a)] = slice(g.size(a) - 1, g.size(a)) if upper else slice(0, 1)
    return q[tuple(sl)]


def _global_index(g: Grid) -> np.ndarray:
    return np.arange(g.n).reshape(g.shape)


# --------------------------------------------------------------------------------------
# fluxes (K.cu:1567-1645 computeFluxesNDLoop)
# --------------------------------------------------------------------------------------
def compute_fluxes(dom: Domain, vel: np.ndarray) -> List[np.ndarray]:
    """Face fluxes ``F_f`` for every cell, NOT multiplied by the face sign.

    Interior / periodic faces: ``0.5*(U_a(P)+U_a(N))`` with contravariant ``U``
    (``:1625-1642``); FIXED faces: contravariant flux of the *boundary* velocity with the
    *boundary* transform (``:1593-1599``).
    """
    g = dom.grid
    out = []
    for f in range(2 * g.dims):
        a, s = f >> 1, (f & 1) * 2 - 1
        U = g.contravariant(vel, a)
        F = 0.5 * (U + _nbr(U, a, s))
        if dom.is_fixed(f):
            Ub = g.contravariant_b(dom.bvel(f), f)
            F = np.where(_at_bound(g, f), _slab_to_cells(g, f, Ub), F)
        out.append(F)
    return out


# --------------------------------------------------------------------------------------
# advection-diffusion matrix (K.cu:3616-3880 PISO_build_matrix)
# --------------------------------------------------------------------------------------
def _assemble_csr(g: Grid, diag: np.ndarray, offs: List[np.ndarray], valid: List[np.ndarray]) -> sp.csr_matrix:
    """Rows sorted by ascending global column index (K.cu:3861-3875); one entry per
    non-prescribed face + diagonal: nnz = (2d+1)N - #prescribed face cells
    (``domain_structs.cpp:2167-2177``)."""
    idx = _global_index(g)
    rows = [idx.ravel()]
    cols = [idx.ravel()]
    vals = [diag.ravel()]
    for f in range(2 * g.dims):
        a, s = f >> 1, (f & 1) * 2 - 1
        m = valid[f].ravel()
        rows.append(idx.ravel()[m])
        cols.append(_nbr(idx, a, s).ravel()[m])
        vals.append(offs[f].ravel()[m])
    A = sp.csr_matrix(
        (np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(g.n, g.n)
    )
    A.sort_indices()
    return A


def build_advection_matrix(dom: Domain, dt: float, for_scalar: bool = False, channel: int = 0):
    """Returns ``(C csr, A diag [cells], offs list[2d] of [cells])``; every row is divided by
    the cell determinant (``:3864``) and ``A = diag/det`` (``:3877``).

    diag = det/dt + sum_f [ s_f*F_f/2 + (alpha_P nu_P + alpha_N nu_N)/2 ]   (non-prescribed)
                  + sum_f (1-slip) * 2 * nu_P * alpha_P                      (prescribed, :3846)
    off_f = s_f*F_f/2 - (alpha_P nu_P + alpha_N nu_N)/2                      (:3717-3749)
    Fluxes are taken from the *block* velocity u^n (``velocityGlobal == nullptr``, :3661).
    """
    g = dom.grid
    nu = dom.kappa(channel) if for_scalar else dom.viscosity
    if not for_scalar and dom.viscosity_field is not None:   # per-cell viscosity of the velocity system (:3698, 3742)
        nu = np.asarray(dom.viscosity_field, dtype=g.det.dtype).reshape(g.shape)
    F = compute_fluxes(dom, dom.velocity)
    diag = g.det / dt
    offs, valid = [], []
    for f in range(2 * g.dims):
        a, s = f >> 1, (f & 1) * 2 - 1
        al_p = g.alpha(a)
        al_n = _nbr(al_p, a, s)
        presc = _prescribed(dom, f)
        visc = 0.5 * (al_p * nu + al_n * (_nbr(nu, a, s) if isinstance(nu, np.ndarray) else nu))   # (:3745)
        ff = s * 0.5 * F[f]
        if dom.is_fixed(f):
            if for_scalar:
                slip = 0.0 if dom.bscalar_type(f, channel) == DIRICHLET else 1.0
            else:
                slip = 0.0  # velocity boundaryType is DIRICHLET (:3839-3841)
            bterm = (1.0 - slip) * 2.0 * nu * al_p
        else:
            bterm = 0.0
        diag = diag + np.where(presc, bterm, ff + visc)
        offs.append(np.where(presc, 0.0, ff - visc) / g.det)
        valid.append(~presc)
    A = diag / g.det
    C = _assemble_csr(g, A, offs, valid)
    return C, A, offs


def advection_rhs_velocity(dom: Domain, dt: float) -> np.ndarray:
    """``kPISO_build_advection_RHS`` (K.cu:4296-4400), ``applyPressureGradient=False``
    (SIM.py:1648):

    rhs_c = [ det*u_c/dt - sum_FIXED u_b,c*(s_f*U_b) + sum_FIXED u_b,c*(1-slip)*nu*2*alpha_b ]/det + S_c
    """
    g = dom.grid
    rhs = g.det * dom.velocity / dt
    rhs = rhs + _boundary_source_velocity(dom)
    rhs = rhs / g.det + dom.source()
    return rhs


def _boundary_source_velocity(dom: Domain) -> np.ndarray:
    """Boundary advection+diffusion source shared by the velocity RHS (K.cu:4321-4380) and the
    H operator (K.cu:5183-5243); NOT yet divided by det."""
    g = dom.grid
    acc = np.zeros((g.dims,) + g.shape, dtype=dom.velocity.dtype)
    for f in range(2 * g.dims):
        if not dom.is_fixed(f):
            continue
        s = (f & 1) * 2 - 1
        ub = dom.bvel(f)
        flux = g.contravariant_b(ub, f) * s
        alpha = g.alpha_b(f)
        slip = 0.0
        # getViscosityFixedBoundary = the adjacent cell's viscosity (K.cu:1840-1843, 4342)
        nu_b = dom.viscosity if dom.viscosity_field is None else _cells_slab(g, f, np.asarray(dom.viscosity_field).reshape(g.shape))
        term = -ub * flux + ub * (1.0 - slip) * nu_b * 2.0 * alpha
        acc = acc + _slab_to_cells(g, f, term)
    return acc


def advection_rhs_scalar(dom: Domain, dt: float) -> np.ndarray:
    """``kPISO_build_scalar_advection_RHS`` (K.cu:4094-4198).  Dirichlet wall:
    ``+T_b*kappa*2*alpha_b``; Neumann wall: ``+g_b*kappa`` (no alpha, :4146); the advective
    term uses the stored boundary datum whatever its type (:4140)."""
    g = dom.grid
    C = dom.scalar.shape[0]
    rhs = g.det * dom.scalar / dt
    for f in range(2 * g.dims):
        if not dom.is_fixed(f):
            continue
        s = (f & 1) * 2 - 1
        flux = g.contravariant_b(dom.bvel(f), f) * s
        alpha = g.alpha_b(f)
        sb = dom.bscalar(f)
        term = np.zeros((C,) + g.slab_shape(f), dtype=rhs.dtype)
        for ch in range(C):
            k = dom.kappa(ch)
            t = -sb[ch] * flux
            if dom.bscalar_type(f, ch) == DIRICHLET:
                t = t + sb[ch] * k * 2.0 * alpha
            else:
                t = t + sb[ch] * k
            term[ch] = t
        rhs = rhs + _slab_to_cells(g, f, term)
    return rhs / g.det


# --------------------------------------------------------------------------------------
# pressure system
# --------------------------------------------------------------------------------------
def build_pressure_matrix(dom: Domain, A: np.ndarray):
    """``PISO_build_pressure_matrix`` (K.cu:4812-4978): ``off_f = (alpha_P/A_P + alpha_N/A_N)/2``
    for non-prescribed faces, ``diag = -sum off`` (:4883-4887); prescribed faces get no entry
    (homogeneous Neumann)."""
    g = dom.grid
    rA = 1.0 / A
    diag = np.zeros(g.shape, dtype=A.dtype)
    offs, valid = [], []
    for f in range(2 * g.dims):
        a, s = f >> 1, (f & 1) * 2 - 1
        al_p = g.alpha(a)
        coef = 0.5 * (al_p * rA + _nbr(al_p * rA, a, s))
        presc = _prescribed(dom, f)
        coef = np.where(presc, 0.0, coef)
        diag = diag - coef
        offs.append(coef)
        valid.append(~presc)
    return _assemble_csr(g, diag, offs, valid), diag, offs


def pressure_rhs(dom: Domain, dt: float, A: np.ndarray, offs: List[np.ndarray], vel_result: np.ndarray) -> np.ndarray:
    """``PISO_build_pressure_rhs`` (K.cu:5136-5255):
    ``h_c = (1/A)[u_c^n/dt - sum_{j!=P} C_Pj * u~_j,c + S_bnd,c/det + S_c]``."""
    g = dom.grid
    H = np.zeros_like(vel_result)
    for f in range(2 * g.dims):
        a, s = f >> 1, (f & 1) * 2 - 1
        for c in range(g.dims):
            H[c] += offs[f] * _nbr(vel_result[c], a, s)
    S = _boundary_source_velocity(dom) / g.det + dom.source()
    return (dom.velocity / dt - H + S) / A


def divergence(dom: Domain, h: np.ndarray) -> np.ndarray:
    """``k_computePressureRHSdivergenceFromFlux`` (K.cu:5389-5434): ``sum_d F_+d - F_-d`` of the
    fluxes of ``h`` (prescribed faces keep the boundary-velocity flux); no det, no dt
    (``timeStepNorm=False``)."""
    F = compute_fluxes(dom, h)
    div = np.zeros(dom.grid.shape, dtype=h.dtype)
    for a in range(dom.dims):
        div = div + F[2 * a + 1] - F[2 * a]
    return div


def pressure_gradient(dom: Domain, p: np.ndarray) -> np.ndarray:
    """``getPressureGradient`` (K.cu:816-849): central difference with factor 1/2, one-sided with
    factor 1 at a prescribed face, then ``matmul(grad, Minv)``."""
    g = dom.grid
    comp = []
    for a in range(g.dims):
        lo_presc = _prescribed(dom, 2 * a)
        hi_presc = _prescribed(dom, 2 * a + 1)
        valN = np.where(lo_presc, p, _nbr(p, a, -1))
        valP = np.where(hi_presc, p, _nbr(p, a, +1))
        fac = np.where(lo_presc | hi_presc, 1.0, 0.5)
        comp.append((valP - valN) * fac)
    out = np.zeros((g.dims,) + g.shape, dtype=p.dtype)
    for c in range(g.dims):
        for a in range(g.dims):
            out[c] += comp[a] * g.Minv[..., a, c]
    return out


def correct_velocity(dom: Domain, h: np.ndarray, p: np.ndarray, A: np.ndarray) -> np.ndarray:
    """``PISO_update_velocity`` (K.cu:5962-5995): ``u = h - (1/A) grad p``."""
    return h - pressure_gradient(dom, p) / A


# --------------------------------------------------------------------------------------
# linear solvers
# --------------------------------------------------------------------------------------
@dataclass
class SolveInfo:
    final_residual: float = 0.0
    used_iterations: int = -1
    converged: bool = False
    finite: bool = True


def _rms(r: np.ndarray) -> float:
    """NORM2_NORMALIZED criterion: ``||r||_2 / sqrt(n)`` (cg_solver_kernel.cu:100-106)."""
    return float(np.linalg.norm(r) / math.sqrt(r.size))


def cg_reference(A: sp.csr_matrix, f: np.ndarray, x0: Optional[np.ndarray], tol: float, maxit: int = 5000,
                 residual_reset_steps: int = 0, return_best: bool = False):
    """Plain CG exactly as ``cgSolveGPU`` iterates (cg_solver_kernel.cu:250-442), including the
    residual restart every ``residualResetSteps`` (:281-302), best-iterate tracking (:345-361)
    and the rising-residual cut-off of 100 iterations (:187, 414-427)."""
    x = np.zeros_like(f) if x0 is None else x0.copy()
    info = SolveInfo()
    r = f - A @ x
    p = r.copy()
    rho = float(r @ r)
    best_x, best_crit, best_it = None, 0.0, -1
    last_crit, rising = 0.0, 0
    for i in range(maxit):
        if residual_reset_steps > 0 and (i + 1) % residual_reset_steps == 0:
            r = f - A @ x
            p = r.copy()
            rho = float(r @ r)
        Ap = A @ p
        alpha = rho / float(p @ Ap)
        x = x + alpha * p
        r = r - alpha * Ap
        crit = _rms(r)
        if not math.isfinite(crit):
            info = SolveInfo(crit, i, False, False)
            break
        if return_best:
            if i == 0 or crit < best_crit:
                best_crit, best_it, best_x = crit, i, x.copy()
            rising = rising + 1 if (i > 0 and crit >= last_crit) else 0
            last_crit = crit
        info.used_iterations, info.final_residual = i, crit
        if crit < tol:
            info.converged = True
            break
        if return_best and (i == maxit - 1 or rising >= 100):
            x = best_x
            info = SolveInfo(best_crit, best_it, False, True)
            break
        rhop = rho
        rho = float(r @ r)
        p = r + (rho / rhop) * p
    return x, info


def bicgstab_reference(A: sp.csr_matrix, f: np.ndarray, x0: Optional[np.ndarray], tol: float, maxit: int = 5000):
    """Un-preconditioned BiCGStab as ``bicgstabSolveGPU`` iterates
    (bicgstab_solver_kernel.cu:259-372)."""
    x = np.zeros_like(f) if x0 is None else x0.copy()
    r = f - A @ x
    rw = r.copy()
    p = r.copy()
    v = np.zeros_like(f)
    info = SolveInfo()
    nrm0 = _rms(r)
    if nrm0 < tol:
        return x, SolveInfo(nrm0, -1, True, True)
    rho = alpha = omega = 1.0
    for i in range(maxit):
        rhop = rho
        rho = float(rw @ r)
        if i > 0:
            beta = (rho / rhop) * (alpha / omega)
            p = r + beta * (p - omega * v)
        v = A @ p
        alpha = rho / float(rw @ v)
        r = r - alpha * v
        x = x + alpha * p
        nrm = _rms(r)
        if not math.isfinite(nrm):
            return x, SolveInfo(nrm, i, False, False)
        info.used_iterations, info.final_residual = i, nrm
        if nrm < tol:
            info.converged = True
            break
        s = r
        t = A @ s
        omega = float(t @ r) / float(t @ t)
        x = x + omega * s
        r = r - omega * t
        nrm = _rms(r)
        info.final_residual = nrm
        if nrm < tol:
            info.converged = True
            info.used_iterations = i + 1
            break
    return x, info


def solve_direct(A: sp.csr_matrix, f: np.ndarray, singular: bool = False) -> np.ndarray:
    """Solver-independent ground truth (the reference offers the same route:
    ``linear_solve_scipy``, SIM.py:1068-1078).  For the rank-deficient all-Neumann/periodic
    pressure matrix the constant null space is removed with a Lagrange row, which also
    projects a slightly incompatible RHS."""
    A = A.astype(np.float64)
    f = f.astype(np.float64)
    if not singular:
        return spla.spsolve(A.tocsc(), f)
    n = A.shape[0]
    ones = sp.csr_matrix(np.ones((1, n)))
    K = sp.bmat([[A, ones.T], [ones, None]], format="csc")
    sol = spla.spsolve(K, np.concatenate([f, [0.0]]))
    return sol[:n]


def get_solver_tolerance(tol, dtype) -> float:
    """``_get_solver_tolerance`` (pict/PISOtorch_diff.py:247-253)."""
    if tol is None:
        return 1e-8 if np.dtype(dtype) == np.float64 else 1e-5
    return tol


# --------------------------------------------------------------------------------------
# stepping
# --------------------------------------------------------------------------------------
@dataclass
class SolverOptions:
    """Subset of ``Simulation.__init__`` arguments that change results
    (simulation.py:124-156; SIM.py:520-580)."""

    corrector_steps: int = 2
    advection_tol: Optional[float] = None
    pressure_tol: Optional[float] = None
    max_iterations: int = 5000
    pressure_return_best_result: bool = False
    normalize_pressure_result: bool = True
    advect_passive_scalar: bool = True
    direct: bool = True  # spsolve ground truth instead of Krylov iterations
    # the reference's non-orthogonal branch run on this (rectilinear) grid, as its TCF env does (tcf_env.py:497): same
    # discretisation here, but the velocity solve starts from zero instead of velocityResult (SIM.py:1735-1742 vs 1689-1693)
    non_orthogonal: bool = False
    stats: Optional[dict] = None  # filled with iteration counts when not None


def _lin_solve(A, rhs, x0, tol, opts: SolverOptions, kind: str, singular=False, **kw):
    """``_linear_solve_wrapper`` policy (pict/PISOtorch_diff.py:373-491): an all-zero RHS
    short-circuits to a zero result (:392, 489-490)."""
    flat = rhs.reshape(-1)
    if not np.any(flat != 0):
        return np.zeros_like(flat)
    if opts.direct:
        return solve_direct(A, flat, singular=singular).astype(rhs.dtype)
    tol = get_solver_tolerance(tol, rhs.dtype)
    x0f = None if x0 is None else x0.reshape(-1)
    if kind == "bicg":
        x, info = bicgstab_reference(A, flat, x0f, tol, opts.max_iterations)
    else:
        x, info = cg_reference(A, flat, x0f, tol, opts.max_iterations, **kw)
    if opts.stats is not None:
        opts.stats.setdefault(kind, []).append(info.used_iterations)
    return x


def piso_split_step(dom: Domain, dt: float, opts: SolverOptions = SolverOptions(),
                    prep_fn: Optional[Dict[str, List[Callable]]] = None, calls: Optional[list] = None) -> dict:
    """One ``_PISO_split_step(iterations=1, time_step=dt)`` in its orthogonal branch
    (SIM.py:1431-2002; ``non_orthogonal=False`` blocks :1515-1563, 1667-1705, 1779-1831).
    Hooks are called as ``fn(dom, dt)`` at PRE / PRE_VELOCITY_SETUP / POST.

    Returns a dict of intermediates (A, rhs, h, div, p ...) for fixture generation.  ``calls`` (a list) receives the step's
    sequence of operators, solves and hooks in the vocabulary of the reference's backend; ``tests/test_split_step_golden.py``
    holds it against the sequence recorded from the reference's own ``_PISO_split_step``.
    """
    g = dom.grid
    out = {}

    def log(op, **kw):
        if calls is not None:
            calls.append(dict(op=op, **kw))

    def run(name):
        log("hook", name=name)
        if prep_fn and name in prep_fn:
            for fn in prep_fn[name]:
                fn(dom, dt)

    def solve(A, rhs, x0, tol, kind, names, **kw):
        log("SolveLinear", matrix=names[0], rhs=names[1], x0=None if x0 is None else names[2], use_BiCG=kind == "bicg", tol=tol,
            return_best_result=bool(kw.get("return_best", False)))
        return _lin_solve(A, rhs, x0, tol, opts, kind, **kw)

    run("PRE")
    # ---- passive scalar (SIM.py:1471-1644): matrix with scalar diffusivity from u^n fluxes
    if opts.advect_passive_scalar and dom.scalar is not None:
        Cn = dom.scalar.shape[0]
        mats = []
        for ch in range(Cn):   # (one matrix per channel: the reference shares channel 0's when the diffusivities allow, :1493-1513)
            log("SetupAdvectionMatrix", for_scalar=True)
            mats.append(build_advection_matrix(dom, dt, for_scalar=True, channel=ch)[0])
        log("SetupAdvectionScalar")
        rhs_s = advection_rhs_scalar(dom, dt)
        run("POST_SCALAR_SETUP")
        res = np.empty_like(dom.scalar)
        for ch in range(Cn):
            res[ch] = solve(mats[ch], rhs_s[ch], None, opts.advection_tol, "bicg", ("C", "scalarRHS", None)).reshape(g.shape)
        out["scalar_rhs"] = rhs_s
        log("setScalarResult")
        dom.scalar = res  # CopyScalarResultToBlocks (:1644)
        log("CopyScalarResultToBlocks")
    run("PRE_VELOCITY_SETUP")
    # ---- velocity predictor (SIM.py:1646-1762)
    log("SetupAdvectionMatrix", for_scalar=False)
    C, A, offs = build_advection_matrix(dom, dt)
    log("SetupAdvectionVelocity", apply_pressure_gradient=False)
    rhs = advection_rhs_velocity(dom, dt)
    run("POST_VELOCITY_SETUP")
    vel = np.empty_like(dom.velocity)
    # advect_use_prev_result in the orthogonal branch (:1436, 1689-1693); the non-orthogonal branch starts its first pass from zero
    # (:1735-1742).  One solve in the reference (all components in one block right-hand side), one per component here
    log("SolveLinear", matrix="C", rhs="velocityRHS", x0=None if opts.non_orthogonal else "velocityResult", use_BiCG=True,
        tol=opts.advection_tol, return_best_result=False)
    for c in range(g.dims):
        x0 = None if opts.non_orthogonal else dom.velocity_result[c]
        vel[c] = _lin_solve(C, rhs[c], x0, opts.advection_tol, opts, "bicg").reshape(g.shape)
    dom.velocity_result = vel
    log("setVelocityResult")
    out.update(A=A, C=C, C_offs=offs, velocity_rhs=rhs, velocity_pred=vel.copy())
    run("POST_PREDICTION")
    # ---- correctors (SIM.py:1777-1972).  SetupPressureCorrection assembles matrix and right-hand side in every corrector; the
    # matrix only depends on A, so it is built once here
    P, Pdiag, Poffs = build_pressure_matrix(dom, A)
    out.update(P=P, P_diag=Pdiag, P_offs=Poffs)
    for cstep in range(opts.corrector_steps):
        log("SetupPressureCorrection")
        h = pressure_rhs(dom, dt, A, offs, dom.velocity_result)
        div = divergence(dom, h)
        run("POST_PRESSURE_SETUP")
        p = solve(P, div, None, opts.pressure_tol, "cg", ("P", "pressureRHSdiv", None), singular=True,
                  return_best=opts.pressure_return_best_result).reshape(g.shape)
        if opts.normalize_pressure_result:
            p = p - p.mean()  # SIM.py:1817-1820
        log("setPressureResult", mean_removed=bool(opts.normalize_pressure_result))
        dom.pressure_result = p
        run("POST_PRESSURE_RESULT")
        run("POST_PRESSURE_NON_ORTHO")
        dom.pressure = p.copy()  # CopyPressureResultToBlocks (:1953)
        log("CopyPressureResultToBlocks")
        log("CorrectVelocity")
        dom.velocity_result = correct_velocity(dom, h, p, A)
        run("POST_VELOCITY_CORRECTION")
        out[f"h{cstep}"], out[f"div{cstep}"], out[f"p{cstep}"] = h, div, p.copy()
        out[f"u{cstep}"] = dom.velocity_result.copy()
    dom.velocity = dom.velocity_result.copy()  # CopyVelocityResultToBlocks (:1974)
    log("CopyVelocityResultToBlocks")
    run("POST")
    return out


def velocity_gradient(dom: Domain) -> np.ndarray:
    """``getBlockDataGradient`` for the velocity components (K.cu:2997-3040): ``grad[i, a]`` = d u_i / d x_a per cell.  Along
    every axis (value above - value below) / distance in index space -- a neighbour cell counts 1, a FIXED (Dirichlet) face its
    boundary value at 0.5 -- then ``dataGrad @ Minv`` (rectilinear: / h_a)."""
    g = dom.grid
    d = g.dims
    grad = np.zeros((d, d) + g.shape, dtype=dom.velocity.dtype)
    u = dom.velocity.reshape((d,) + g.shape)
    for a in range(d):
        diff = np.zeros_like(u)
        dist = np.full(g.shape, 2.0)
        for up in (0, 1):
            f, s = 2 * a + up, (1 if up else -1)
            val = _nbr(u, a, s)                                   # [d, cells]: the neighbour cell (periodic wrap)
            if dom.is_fixed(f):
                at = _at_bound(g, f)
                val = np.where(at, _slab_to_cells(g, f, dom.bvel(f)), val)
                dist = dist - 0.5 * at
            diff = diff + s * val
        grad[:, a] = diff / dist * g.Minv[..., a, a]      # dataGrad @ Minv with a diagonal Minv
    return grad


def sgs_smagorinsky(dom: Domain, coefficient: float) -> np.ndarray:
    """``SGSviscosityIncompressibleSmagorinsky`` (K.cu:6913-6966): ``C * Delta^2 * sqrt(2 S:S)``, ``Delta^2`` the largest squared
    column length of the cell's M (its extents on a rectilinear grid; the kernel never takes the root)."""
    g = dom.grid
    d = g.dims
    grad = velocity_gradient(dom)
    acc = np.zeros(g.shape, dtype=dom.velocity.dtype)
    for i in range(d):
        for j in range(i, d):
            sij = (0.5 * (grad[i, j] + grad[j, i])) ** 2
            acc = acc + (2.0 * sij if i != j else sij)
    delta = np.max(np.stack([np.sum(g.M[..., :, a] ** 2, axis=-1) for a in range(d)]), axis=0)   # squared column lengths of M
    return coefficient * delta * np.sqrt(2.0 * acc)


def max_velocity(dom: Domain) -> float:
    """``Domain.getMaxVelocity(withBounds=True, computational=True)``
    (domain_structs.cpp:1360-1366, 1580-1611): max over components of ``|Minv . u|`` over
    cells and FIXED boundaries (with their own transforms)."""
    g = dom.grid
    m = 0.0
    for a in range(g.dims):
        acc = np.zeros(g.shape, dtype=dom.velocity.dtype)
        for c in range(g.dims):
            acc = acc + g.Minv[..., a, c] * dom.velocity[c]
        m = max(m, float(np.abs(acc).max()))
    for f in range(2 * g.dims):
        if dom.is_fixed(f):
            ub = dom.bvel(f)
            Mi = g.b_Minv[f]
            for a in range(g.dims):
                acc = np.zeros(Mi.shape[:-2], dtype=ub.dtype)
                for c in range(g.dims):
                    acc = acc + Mi[..., a, c] * ub[c]
                m = max(m, float(np.abs(acc).max()))
    return m


def adaptive_substeps(max_vel: float, t_remaining: float, cfl: float):
    """Time-step choice of ``_PISO_adaptive_step`` (SIM.py:2013-2028)."""
    if np.isclose(max_vel, 0):
        max_ts = t_remaining
    else:
        max_ts = cfl / max_vel
    if max_ts >= t_remaining:
        return 1, t_remaining
    n = int(np.ceil(t_remaining / max_ts))
    return n, t_remaining / n


def piso_adaptive_step(dom: Domain, time_step: float, cfl: float, opts: SolverOptions = SolverOptions(),
                       prep_fn=None, dtype=np.float32) -> int:
    """``_PISO_adaptive_step`` (SIM.py:2004-2064): recompute ``max_vel`` before every substep and
    take ONE split step of ``ts``; ``ts`` is rounded through the domain dtype (:2029-2031)."""
    t_rem = time_step
    n_sub = 0
    while t_rem > 0 and not np.isclose(t_rem, 0):
        _, ts = adaptive_substeps(max_velocity(dom), t_rem, cfl)
        t_rem -= ts
        ts = float(np.asarray(ts, dtype=dtype))
        piso_split_step(dom, ts, opts, prep_fn)
        n_sub += 1
    return n_sub


def boundary_flux_balance(dom: Domain) -> float:
    """``Domain.GetGlobalFluxBalance`` (domain_structs.cpp:2476-2509): sum of FIXED-boundary
    contravariant fluxes, lower faces negated."""
    g = dom.grid
    tot = 0.0
    for f in range(2 * g.dims):
        if dom.is_fixed(f):
            fl = float(g.contravariant_b(dom.bvel(f), f).sum())
            tot += fl if (f & 1) else -fl
    return tot


def make_divergence_free(dom: Domain, opts: SolverOptions = SolverOptions()):
    """``make_divergence_free`` (SIM.py:1320-1429): with ``A := 1`` and ``dt := 1`` treat the
    current velocity as ``h``, solve one pressure system and correct."""
    g = dom.grid
    A = np.ones(g.shape, dtype=dom.velocity.dtype)
    P, _, _ = build_pressure_matrix(dom, A)
    div = divergence(dom, dom.velocity)
    p = _lin_solve(P, div, None, opts.pressure_tol, opts, "cg", singular=True).reshape(g.shape)
    p = p - p.mean()
    dom.velocity = correct_velocity(dom, dom.velocity, p, A)
    dom.velocity_result = dom.velocity.copy()
    return p


# --------------------------------------------------------------------------------------
# advective outflow (SIM.py:228-393) and flux balancing (SIM.py:188-224)
# --------------------------------------------------------------------------------------
def update_advective_boundaries(dom: Domain, faces: Sequence[int], velm: np.ndarray, dt: float, tol=None):
    """``phi_b <- phi_b - t (phi_b - phi_cell)``, ``t = 1 - 1/(1 + 2 dt (Minv_row_n . u_m))``
    for each free FIXED face, then ``balance_boundary_fluxes``."""
    g = dom.grid
    for k, f in enumerate(faces):
        a = f >> 1
        bc = dom.bc[f]
        vb = np.array(dom.bvel(f))
        Mi = g.b_Minv[f]
        adv = np.zeros(Mi.shape[:-2], dtype=vb.dtype)
        vm = velm[k] if isinstance(velm, (list, tuple)) else velm   # one characteristic velocity per face, or a global one (:146-147)
        for c in range(g.dims):
            adv = adv + Mi[..., a, c] * vm[c]
        t = 1.0 - 1.0 / (1.0 + dt * 2.0 * adv)
        bc.velocity = vb - t * (vb - _cells_slab(g, f, dom.velocity))
        if dom.scalar is not None and bc.scalar is not None:
            sb = np.array(dom.bscalar(f))
            bc.scalar = sb - t * (sb - _cells_slab(g, f, dom.scalar))
    balance_boundary_fluxes(dom, faces, tol)


def balance_boundary_fluxes(dom: Domain, free_faces: Sequence[int], tol=None):
    g = dom.grid
    fixed = var = 0.0
    for f in range(2 * g.dims):
        if not dom.is_fixed(f):
            continue
        fl = float(g.contravariant_b(dom.bvel(f), f).sum())
        fl = fl if (f & 1) else -fl
        if f in free_faces:
            var += fl
        else:
            fixed += fl
    atol = get_solver_tolerance(tol, dom.velocity.dtype) * 0.01
    if not np.isclose(fixed + var, 0.0, rtol=1e-5, atol=atol):
        scale = -fixed / var
        for f in free_faces:
            dom.bc[f].velocity = np.array(dom.bvel(f)) * scale
