"""CPU oracle for the multi-block, non-orthogonal PISO step.

TEST INFRASTRUCTURE ONLY.  Nothing in ``fluidgym_amd`` may import this module; only ``tests/`` use it, as the
checker of the HIP multi-block path (``fluidgym_amd/csrc/fg_mb*.hip``).

PARITY UNPINNED: as for ``piso_oracle.py`` the reference has no CPU path and its tests hold no vectors for the solver;
this file restates the per-cell device functions of the reference literally (one Python call per cell and face, so it
is only usable on meshes of a few thousand cells).  What pins it: (i) a rectilinear channel cut into blocks with
shuffled / inverted axes must reproduce the single-block oracle (``tests/test_mb_oracle.py``), (ii) flows that are
exact for the discretisation on skewed meshes (uniform flow, Couette), (iii) discrete invariants (flux balance,
constant null space of the pressure operator including its lagged corner terms).

Citations: ``K.cu`` = ``/root/reference/src/fluidgym/simulation/extensions/PISO_multiblock_cuda_kernel.cu``,
``DS.cpp`` = ``extensions/domain_structs.cpp``, ``GG.cu`` = ``extensions/grid_gen.cu``, ``SIM.py`` =
``pict/PISOtorch_simulation.py``.

Conventions: ``pos = [x, y(, z)]``; faces ``-x,+x,-y,+y,-z,+z = 0..2d-1`` (axis = face >> 1, upper = face & 1);
cell arrays are C-ordered ``[(nz,) ny, nx]``; global cell index = block offset + x + nx (y + ny z)
(``flattenIndexGlobal``, ``K.cu:183-190``); global vectors are component-major ``[d, N]``.

Reference behaviour reproduced on purpose: every walk that crosses a block connection while collecting *diagonal*
neighbours lands one layer inside the connected block (``computeConnectedPos(..., borderOffset=1)``,
``K.cu:2152, 2658, 2825``) -- ``CONNECTED_DIAGONAL_OFFSET`` below; set it to 0 for the geometrically adjacent cell.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .piso_oracle import boundary_face_transform, coords_to_transforms

FIXED, CONNECTED, PERIODIC = "fixed", "connected", "periodic"
CONNECTED_DIAGONAL_OFFSET = 1
# K.cu:1952 treats the inner face of a first-layer cell as prescribed when the FAR side of the block is a wall (matrices only)
FIRST_LAYER_QUIRK = True

# nonOrthoFlags (SIM.py:479-487); the simulation always runs CENTER_MATRIX | DIRECT_MATRIX | DIAGONAL_RHS
NON_ORTHO_DIRECT_MATRIX, NON_ORTHO_DIRECT_RHS, NON_ORTHO_DIAGONAL_RHS, NON_ORTHO_CENTER_MATRIX = 1, 2, 8, 16
NON_ORTHO_MODE = NON_ORTHO_CENTER_MATRIX | NON_ORTHO_DIRECT_MATRIX | NON_ORTHO_DIAGONAL_RHS


def face_transform_boundary(coords: np.ndarray, face: int):
    """Transforms of the boundary faces of one block side (``k_CoordsToFaceTransforms`` GG.cu:398-470 sliced by
    ``Block::GetFaceTransformBoundarySlice`` DS.cpp:1825-1851), flattened over the face cells (x fastest)."""
    d = coords.shape[0]
    M, Minv, det = boundary_face_transform(coords, face)
    return M.reshape(-1, d, d), Minv.reshape(-1, d, d), det.reshape(-1)


@dataclass
class Bound:
    type: str
    # FIXED: Dirichlet velocity on the face cells, [d, cells on the face in C order of the remaining axes]
    velocity: Optional[np.ndarray] = None
    # CONNECTED: index of the other block and the `axes` vector of ConnectedBoundary (DS.cpp:1080-1113)
    other: int = -1
    axes: Tuple[int, ...] = ()
    Minv: Optional[np.ndarray] = None
    det: Optional[np.ndarray] = None


@dataclass
class Block:
    coords: np.ndarray  # [d, (nz+1,) ny+1, nx+1]
    bounds: List[Bound] = field(default_factory=list)
    size: Tuple[int, ...] = ()
    offset: int = 0
    Minv: np.ndarray = None
    det: np.ndarray = None

    @property
    def ncells(self):
        return int(np.prod(self.size))


class Domain:
    """Plain description of a multi-block domain + the reference's per-cell functions on it."""

    def __init__(self, dims: int, viscosity: float):
        self.d = dims
        self.nu = float(viscosity)
        self.blocks: List[Block] = []

    # ------------------------------------------------------------------ construction
    def add_block(self, coords) -> int:
        coords = np.asarray(coords, dtype=np.float64)
        assert coords.shape[0] == self.d
        blk = Block(coords=coords)
        blk.size = tuple(coords.shape[-1 - a] - 1 for a in range(self.d))
        _, Minv, det = coords_to_transforms(coords)
        blk.Minv, blk.det = Minv, det
        blk.bounds = [None] * (2 * self.d)
        self.blocks.append(blk)
        for f in range(2 * self.d):
            self.close(len(self.blocks) - 1, f)
        return len(self.blocks) - 1

    def _face_shape(self, b, face):
        axis = face >> 1
        return tuple(self.blocks[b].size[a] for a in reversed(range(self.d)) if a != axis)

    def close(self, b: int, face: int, velocity=None):
        blk = self.blocks[b]
        _, Minv, det = face_transform_boundary(blk.coords, face)
        n = int(np.prod(self._face_shape(b, face)))
        vel = np.zeros((self.d, n)) if velocity is None else np.asarray(velocity, np.float64).reshape(self.d, n).copy()
        blk.bounds[face] = Bound(FIXED, velocity=vel, Minv=Minv, det=det)

    def make_periodic(self, b: int, axis: int):
        self.blocks[b].bounds[2 * axis] = Bound(PERIODIC)
        self.blocks[b].bounds[2 * axis + 1] = Bound(PERIODIC)

    def connect(self, b1: int, face1: int, b2: int, face2: int, axis1: int, axis2: int = 0):
        """``ConnectBlocks`` (DS.cpp:1080-1113)."""
        d = self.d
        axes1, axes2 = [face2], [face1]
        if d > 1:
            axes1.append(axis1)
            f1d, f2d = face1 >> 1, face2 >> 1
            if d == 2 or (axis1 >> 1) == (f2d + 1) % d:
                axes2.append((((f1d + 1) % d) << 1) | (axis1 & 1))
                swapped = False
            else:
                assert (axis2 >> 1) == (f2d + 1) % d
                axes2.append((((f1d + 2) % d) << 1) | (axis2 & 1))
                swapped = True
            if d > 2:
                axes1.append(axis2)
                axes2.append(((((f1d + 2) % d) << 1) | (axis2 & 1)) if not swapped else ((((f1d + 1) % d) << 1) | (axis1 & 1)))
        self.blocks[b1].bounds[face1] = Bound(CONNECTED, other=b2, axes=tuple(axes1))
        self.blocks[b2].bounds[face2] = Bound(CONNECTED, other=b1, axes=tuple(axes2))

    def finalize(self):
        off = 0
        for blk in self.blocks:
            blk.offset = off
            off += blk.ncells
        self.N = off

    # ------------------------------------------------------------------ indexing helpers
    def flat(self, b, pos):
        s = self.blocks[b].size
        i = pos[0]
        if self.d > 1:
            i += s[0] * pos[1]
        if self.d > 2:
            i += s[0] * s[1] * pos[2]
        return i

    def gidx(self, b, pos):
        return self.blocks[b].offset + self.flat(b, pos)

    def face_flat(self, b, face, pos):
        """index of the boundary-face cell of `pos` on `face` (position with the face axis dropped)."""
        axis = face >> 1
        s = self.blocks[b].size
        idx, stride = 0, 1
        for a in range(self.d):
            if a == axis:
                continue
            idx += stride * pos[a]
            stride *= s[a]
        return idx

    def cells(self):
        for b, blk in enumerate(self.blocks):
            for flat in range(blk.ncells):
                pos, r = [], flat
                for a in range(self.d):
                    pos.append(r % blk.size[a])
                    r //= blk.size[a]
                yield b, pos

    def at_bound(self, b, pos, face):
        a = face >> 1
        return pos[a] == (self.blocks[b].size[a] - 1 if face & 1 else 0)

    def is_empty(self, b, face):  # isEmptyBound (K.cu:203-209): a prescribed boundary
        return self.blocks[b].bounds[face].type == FIXED

    def Tcell(self, b, pos):
        blk = self.blocks[b]
        ix = tuple(pos[a] for a in reversed(range(self.d)))
        return blk.Minv[ix], blk.det[ix]

    def Tbound(self, b, face, pos):
        bd = self.blocks[b].bounds[face]
        k = self.face_flat(b, face, pos)
        return bd.Minv[k], bd.det[k]

    # ------------------------------------------------------------------ connections (K.cu:329-375, 446-492)
    def connected_pos(self, b, pos, bdim, cb: Bound, border_offset=0):
        o = self.blocks[cb.other]
        cp = list(pos)
        ca = cb.axes[0] >> 1
        cp[ca] = (o.size[ca] - 1 - border_offset) if (cb.axes[0] & 1) else border_offset
        for k in range(1, self.d):
            axis = (bdim + k) % self.d
            ca = cb.axes[k] >> 1
            cp[ca] = (o.size[ca] - 1 - pos[axis]) if (cb.axes[k] & 1) else pos[axis]
        return cp

    def connected_dir(self, direction, bdim, cb: Bound):
        rel = ((direction >> 1) - bdim) % self.d
        return cb.axes[rel] ^ (direction & 1)

    def connected_channel(self, comp, bdim, cb: Bound):  # computeConnectedPosWithChannel's .w
        return cb.axes[(comp - bdim) % self.d] >> 1

    def resolve_neighbor(self, b, pos, face):
        """resolveNeighborCell: ('cell', b2, pos2, axis_mapping) or ('bound', b, face, pos)."""
        mapping = [2 * a for a in range(self.d)]
        axis = face >> 1
        if self.at_bound(b, pos, face):
            bd = self.blocks[b].bounds[face]
            if bd.type == FIXED:
                return ("bound", b, face, list(pos), mapping)
            if bd.type == CONNECTED:
                p2 = self.connected_pos(b, pos, axis, bd)
                mapping = [self.connected_dir(m, axis, bd) for m in mapping]
                return ("cell", bd.other, p2, mapping)
            p2 = list(pos)
            p2[axis] = 0 if face & 1 else self.blocks[b].size[axis] - 1
            return ("cell", b, p2, mapping)
        p2 = list(pos)
        p2[axis] += 1 if face & 1 else -1
        return ("cell", b, p2, mapping)

    # ------------------------------------------------------------------ metrics
    def alpha(self, b, pos, c1, c2):  # getLaplaceCoefficient (K.cu:1259-1273)
        Minv, det = self.Tcell(b, pos)
        return det * float(Minv[c1] @ Minv[c2])

    def alpha_bound(self, b, face, pos, c1, c2):  # getLaplaceCoefficientsNeighbourBoundary (K.cu:1484-1506)
        Minv, det = self.Tbound(b, face, pos)
        return det * float(Minv[c1] @ Minv[c2])

    def contra(self, b, pos, comp, u):  # getContravariantComponent (K.cu:658-669): det * Minv[comp] . u
        Minv, det = self.Tcell(b, pos)
        g = self.gidx(b, pos)
        return det * float(Minv[comp] @ u[:, g])

    def contra_bound(self, b, face, pos, comp):
        Minv, det = self.Tbound(b, face, pos)
        v = self.blocks[b].bounds[face].velocity[:, self.face_flat(b, face, pos)]
        return det * float(Minv[comp] @ v)

    def bound_value(self, b, face, pos, comp):
        return self.blocks[b].bounds[face].velocity[comp, self.face_flat(b, face, pos)]

    # ------------------------------------------------------------------ fluxes (computeFluxesNDLoop, K.cu:1568-1645)
    def fluxes(self, b, pos, u):
        out = [0.0] * (2 * self.d)
        for face in range(2 * self.d):
            dim, upper = face >> 1, face & 1
            velC = self.contra(b, pos, dim, u)
            if self.at_bound(b, pos, face):
                bd = self.blocks[b].bounds[face]
                if bd.type == FIXED:
                    out[face] = self.contra_bound(b, face, pos, dim)
                elif bd.type == CONNECTED:
                    p2 = self.connected_pos(b, pos, dim, bd)
                    velN = self.contra(bd.other, p2, bd.axes[0] >> 1, u)
                    if (bd.axes[0] & 1) == upper:
                        velN = -velN
                    out[face] = 0.5 * (velN + velC)
                else:
                    p2 = list(pos)
                    p2[dim] = 0 if upper else self.blocks[b].size[dim] - 1
                    out[face] = 0.5 * (self.contra(b, p2, dim, u) + velC)
            else:
                p2 = list(pos)
                p2[dim] += 1 if upper else -1
                out[face] = 0.5 * (self.contra(b, p2, dim, u) + velC)
        return out

    # ------------------------------------------------------------------ corner values (getCornerValue, K.cu:2757-2874)
    def corner_value(self, b, pos, dir1, dir2, inc0, inc1, max_depth, cell_get, bound_get):
        """Returns (data, num_cells); num_cells == 0: the value comes from a (Dirichlet) boundary.
        ``cell_get(g)`` reads a cell value by global index, ``bound_get(b, face, pos)`` a boundary value."""
        data, num = 0.0, 0
        if inc0:
            data += cell_get(self.gidx(b, pos))
        num += 1
        cyc = [dict(d1=dir1, d2=dir2, b=b, pos=list(pos)), dict(d1=dir2, d2=dir1, b=b, pos=list(pos))]
        for depth in range(1, max_depth + 1):
            for k in range(2):
                c = cyc[k]
                axis, sign = c["d1"] >> 1, (1 if c["d1"] & 1 else -1)
                if self.at_bound(c["b"], c["pos"], c["d1"]):
                    bd = self.blocks[c["b"]].bounds[c["d1"]]
                    if bd.type == FIXED:
                        if self.at_bound(c["b"], c["pos"], c["d2"]):
                            return bound_get(c["b"], c["d1"], c["pos"]), 0
                        npos = list(c["pos"])
                        npos[c["d2"] >> 1] += 1 if c["d2"] & 1 else -1
                        return 0.5 * (bound_get(c["b"], c["d1"], c["pos"]) + bound_get(c["b"], c["d1"], npos)), 0
                    if bd.type == CONNECTED:
                        p2 = self.connected_pos(c["b"], c["pos"], axis, bd, CONNECTED_DIAGONAL_OFFSET)
                        d1 = c["d1"]
                        c["d1"] = self.connected_dir(c["d2"], axis, bd)
                        c["d2"] = self.connected_dir(d1, axis, bd) ^ 1
                        c["b"], c["pos"] = bd.other, p2
                    else:
                        c["pos"][axis] = 0 if c["d1"] & 1 else self.blocks[c["b"]].size[axis] - 1
                        c["d1"], c["d2"] = c["d2"], c["d1"] ^ 1
                else:
                    c["pos"][axis] += sign
                    c["d1"], c["d2"] = c["d2"], c["d1"] ^ 1
                o = cyc[k ^ 1]
                if c["b"] == o["b"] and c["pos"] == o["pos"]:
                    return data / num, num
                if depth > 1 or inc1:
                    data += cell_get(self.gidx(c["b"], c["pos"]))
                num += 1
        return data / num, num

    def neighbor_data(self, b, pos, direction, cell_get, bound_get):
        """getBlockDataNeighbor (K.cu:2534-2578); over a connection it lands borderOffset = 1 inside the other block."""
        dim = direction >> 1
        if self.at_bound(b, pos, direction):
            bd = self.blocks[b].bounds[direction]
            if bd.type == FIXED:
                return bound_get(b, direction, pos)
            if bd.type == CONNECTED:
                return cell_get(self.gidx(bd.other, self.connected_pos(b, pos, dim, bd, CONNECTED_DIAGONAL_OFFSET)))
            q = list(pos)
            q[dim] = 0 if direction & 1 else self.blocks[b].size[dim] - 1
            return cell_get(self.gidx(b, q))
        q = list(pos)
        q[dim] += 1 if direction & 1 else -1
        return cell_get(self.gidx(b, q))

    def neighbor_diagonal(self, b, pos, dir1, dir2, cell_get, bound_get):
        """getBlockDataNeighborDiagonal (K.cu:2630-2678); directions are NOT remapped across a connection there."""
        first_empty = self.is_empty(b, dir1)
        dirs = (dir2, dir1) if first_empty else (dir1, dir2)
        cb, cp = b, list(pos)
        for face in dirs:
            dim = face >> 1
            if self.at_bound(cb, cp, face):
                bd = self.blocks[cb].bounds[face]
                if bd.type == FIXED:
                    return bound_get(cb, face, cp)
                if bd.type == CONNECTED:
                    cp = self.connected_pos(cb, cp, dim, bd, CONNECTED_DIAGONAL_OFFSET)
                    cb = bd.other
                else:
                    cp[dim] = 0 if face & 1 else self.blocks[cb].size[dim] - 1
            else:
                cp[dim] += 1 if face & 1 else -1
        return cell_get(self.gidx(cb, cp))

    # ------------------------------------------------------------------ cross metrics on faces
    def _ra(self, g, rA, with_visc):
        return (1.0 if rA is None else rA[g]) * (self.nu if with_visc else 1.0)

    def alpha_interp_matrix(self, b, pos, face, taxis, rA, with_visc):
        """interpolateNonOrthoLaplaceComponents (K.cu:1927-2001), what the matrices use.  Reproduced as written: the face
        counts as prescribed unless the cell is interior along the axis or THIS side of the block is not prescribed
        (:1952) -- a cell in the first layer therefore gets 0 on its inner face when the far side is a wall -- and the
        neighbour's metric axes are not mapped through a connection (:1957)."""
        axis = face >> 1
        if FIRST_LAYER_QUIRK:
            if not ((0 < pos[axis] < self.blocks[b].size[axis] - 1) or not self.is_empty(b, face)):
                return 0.0
        elif self.at_bound(b, pos, face) and self.is_empty(b, face):
            return 0.0
        raP = self._ra(self.gidx(b, pos), rA, with_visc)
        aP = self.alpha(b, pos, axis, taxis)
        _, b2, p2, _ = self.resolve_neighbor(b, pos, face)[:4]
        aN = self.alpha(b2, p2, taxis, axis)
        return 0.5 * (aP * raP + aN * self._ra(self.gidx(b2, p2), rA, with_visc))

    def alpha_interp_rhs(self, b, pos, face, taxis, rA, with_visc):
        """face coefficient of getNonOrthoLaplaceRHS_v2 (K.cu:3140-3160): here the neighbour's axes ARE mapped through
        the connection (getLaplaceCoefficientsSingleAxisNeighbor, K.cu:1431-1466; the inversion bit is dropped)."""
        axis = face >> 1
        raP = self._ra(self.gidx(b, pos), rA, with_visc)
        aP = self.alpha(b, pos, axis, taxis)
        res = self.resolve_neighbor(b, pos, face)
        b2, p2, mapping = res[1], res[2], res[4] if res[0] == "bound" else res[3]
        Minv, det = self.Tcell(b2, p2)
        aN = det * float(Minv[mapping[taxis] >> 1] @ Minv[mapping[axis] >> 1])
        return 0.5 * (aP * raP + aN * self._ra(self.gidx(b2, p2), rA, with_visc))

    # ------------------------------------------------------------------ advection-diffusion matrix (K.cu:3616-3880)
    def build_matrix(self, u, dt, flags=NON_ORTHO_MODE):
        """Returns diag [N], off [2d, N], nbr [2d, N] (global index or -1); rows already divided by det; A = diag."""
        d, N = self.d, self.N
        diag = np.zeros(N)
        off = np.zeros((2 * d, N))
        nbr = -np.ones((2 * d, N), dtype=np.int64)
        for b, pos in self.cells():
            g = self.gidx(b, pos)
            det = self.Tcell(b, pos)[1]
            fl = self.fluxes(b, pos, u)
            dg = det / dt
            row = [0.0] * (2 * d)
            alpha_p = [self.alpha(b, pos, a, a) for a in range(d)]
            for face in range(2 * d):
                dim, upper = face >> 1, face & 1
                fs = 1.0 if upper else -1.0
                atb = self.at_bound(b, pos, face)
                if not atb or not self.is_empty(b, face):
                    ff = fs * 0.5 * fl[face]
                    dg += ff
                    row[face] += ff
                    _, b2, p2, mapping = self.resolve_neighbor(b, pos, face)[:4]
                    comp = dim
                    if atb and self.blocks[b].bounds[face].type == CONNECTED:
                        comp = self.connected_channel(dim, dim, self.blocks[b].bounds[face])
                    aN = self.alpha(b2, p2, comp, comp)
                    nbr[face, g] = self.gidx(b2, p2)
                    vc = 0.5 * (alpha_p[dim] * self.nu + aN * self.nu)
                    dg += vc
                    row[face] -= vc
                    inc_n, inc_d = flags & NON_ORTHO_DIRECT_MATRIX, flags & NON_ORTHO_CENTER_MATRIX
                    if d > 1 and (inc_n or inc_d):
                        for i in range(1, d):
                            t = (dim + i) % d
                            a = self.alpha_interp_matrix(b, pos, face, t, None, True)
                            if a == 0.0:
                                continue
                            for tu in range(2):
                                tf = 2 * t + tu
                                tfs = 1.0 if tu else -1.0
                                _, num = self.corner_value(b, pos, face, tf, False, False, 2, lambda q: 0.0,
                                                           lambda *q: 0.0)
                                if num < 1:
                                    continue  # Dirichlet corner: value goes to the right-hand side
                                c = fs * tfs * a / num
                                if inc_d:
                                    dg -= c
                                if inc_n:
                                    row[face] -= c
                                    row[tf] -= c
                else:
                    dg += 2.0 * self.nu * alpha_p[dim]  # no-slip Dirichlet wall
            diag[g] = dg / det
            for face in range(2 * d):
                off[face, g] = row[face] / det if nbr[face, g] >= 0 else 0.0
        return diag, off, nbr

    # ------------------------------------------------------------------ lagged corner terms (K.cu:3048-3202)
    def nonortho_rhs(self, b, pos, comp, field_get, bound_get, flags, pressure, rA=None):
        """getNonOrthoLaplaceRHS_v2 for one cell and channel; velocity: with viscosity, pressure: with 1/A."""
        if not (flags & (NON_ORTHO_DIRECT_RHS | NON_ORTHO_DIAGONAL_RHS)):
            return 0.0
        d = self.d
        S = 0.0
        for face in range(2 * d):
            axis, upper = face >> 1, face & 1
            fs = 1.0 if upper else -1.0
            atb = self.at_bound(b, pos, face)
            if atb and self.is_empty(b, face):
                if pressure:
                    continue
                for k in range(1, d):
                    t = (axis + k) % d
                    size_t = self.blocks[b].size[t]
                    lo, hi = list(pos), list(pos)
                    dist = 0.5
                    if pos[t] != 0:
                        lo[t] -= 1
                    if pos[t] != size_t - 1:
                        hi[t] += 1
                    if pos[t] == 0 or pos[t] == size_t - 1:
                        dist = 1.0
                    grad = dist * (bound_get(b, face, hi) - bound_get(b, face, lo))
                    S -= fs * self.alpha_bound(b, face, pos, t, axis) * grad * self.nu
                continue
            for k in range(1, d):
                t = (axis + k) % d
                fa = self.alpha_interp_rhs(b, pos, face, t, rA, not pressure)
                grad = 0.0
                for tu in range(2):
                    tf = 2 * t + tu
                    tfs = 1.0 if tu else -1.0
                    val, num = self.corner_value(b, pos, face, tf, False, bool(flags & NON_ORTHO_DIRECT_RHS), 2,
                                                 field_get, bound_get)
                    if num == 0 and pressure:
                        to, tos = tf ^ 1, -tfs
                        if flags & NON_ORTHO_DIRECT_RHS:
                            grad += tfs * self.neighbor_data(b, pos, face, field_get, bound_get) * 0.75
                            grad += tos * self.neighbor_data(b, pos, to, field_get, bound_get) * 0.25
                        if flags & NON_ORTHO_DIAGONAL_RHS:
                            grad += tos * self.neighbor_diagonal(b, pos, face, to, field_get, bound_get) * 0.25
                    else:
                        grad += tfs * val
                S -= fs * fa * grad
        return S

    # ------------------------------------------------------------------ velocity right-hand side (K.cu:4296-4400)
    def velocity_rhs(self, u_old, u_result, dt, source=None, flags=NON_ORTHO_MODE):
        d, N = self.d, self.N
        rhs = np.zeros((d, N))
        for b, pos in self.cells():
            g = self.gidx(b, pos)
            det = self.Tcell(b, pos)[1]
            for comp in range(d):
                r = det * u_old[comp, g] / dt
                for face in range(2 * d):
                    if self.at_bound(b, pos, face) and self.is_empty(b, face):
                        fn = 1.0 if face & 1 else -1.0
                        vel = self.bound_value(b, face, pos, comp)
                        a = self.alpha_bound(b, face, pos, face >> 1, face >> 1)
                        flux = self.contra_bound(b, face, pos, face >> 1) * fn
                        r -= vel * flux
                        r += vel * self.nu * 2.0 * a
                r -= self.nonortho_rhs(b, pos, comp, lambda q, c=comp: u_result[c, q],
                                       lambda bb, ff, pp, c=comp: self.bound_value(bb, ff, pp, c), flags, False)
                r /= det
                if source is not None:
                    r += source[comp, g]
                rhs[comp, g] = r
        return rhs

    # ------------------------------------------------------------------ pressure matrix (K.cu:4812-4978)
    def build_pressure_matrix(self, A, flags=NON_ORTHO_MODE):
        d, N = self.d, self.N
        rA = 1.0 / A
        diag = np.zeros(N)
        off = np.zeros((2 * d, N))
        nbr = -np.ones((2 * d, N), dtype=np.int64)
        for b, pos in self.cells():
            g = self.gidx(b, pos)
            for face in range(2 * d):
                dim, upper = face >> 1, face & 1
                fs = 1.0 if upper else -1.0
                atb = self.at_bound(b, pos, face)
                if atb and self.is_empty(b, face):
                    continue
                _, b2, p2, _ = self.resolve_neighbor(b, pos, face)[:4]
                comp = dim
                if atb and self.blocks[b].bounds[face].type == CONNECTED:
                    comp = self.connected_channel(dim, dim, self.blocks[b].bounds[face])
                gN = self.gidx(b2, p2)
                c = 0.5 * (self.alpha(b, pos, dim, dim) * rA[g] + self.alpha(b2, p2, comp, comp) * rA[gN])
                diag[g] -= c
                off[face, g] += c
                nbr[face, g] = gN
        # second pass: corner terms need the neighbour slots of the whole row
        for b, pos in self.cells():
            g = self.gidx(b, pos)
            inc_n, inc_d = flags & NON_ORTHO_DIRECT_MATRIX, flags & NON_ORTHO_CENTER_MATRIX
            if d < 2 or not (inc_n or inc_d):
                continue
            for face in range(2 * d):
                dim, upper = face >> 1, face & 1
                fs = 1.0 if upper else -1.0
                if self.at_bound(b, pos, face) and self.is_empty(b, face):
                    continue
                for i in range(1, d):
                    t = (dim + i) % d
                    a = self.alpha_interp_matrix(b, pos, face, t, rA, False)
                    if a == 0.0:
                        continue
                    for tu in range(2):
                        tf = 2 * t + tu
                        tfs = 1.0 if tu else -1.0
                        _, num = self.corner_value(b, pos, face, tf, False, False, 2, lambda q: 0.0, lambda *q: 0.0)
                        if num < 1:
                            c = fs * tfs * a * 0.25
                            if inc_d:
                                diag[g] += 3 * c
                            if inc_n:
                                off[face, g] += 3 * c
                                if nbr[tf ^ 1, g] >= 0:
                                    off[tf ^ 1, g] -= c
                        else:
                            c = fs * tfs * a / num
                            if inc_d:
                                diag[g] += c
                            if inc_n:
                                off[face, g] += c
                                if nbr[tf, g] >= 0:
                                    off[tf, g] += c
        return diag, off, nbr

    # ------------------------------------------------------------------ pressure right-hand side (K.cu:5136-5255, 5389-5434, 5471-5493)
    def pressure_rhs_h(self, C, u_old, u_star, dt, source=None):
        """h = (u_old/dt - H u* + S) / A with H the off-diagonal part of C; S as in the velocity RHS but without the
        lagged corner terms (PRESSURE_RHS_WITH_BOUNDARY_SOURCES, K.cu:5134)."""
        diag, off, nbr = C
        d, N = self.d, self.N
        h = np.zeros((d, N))
        for b, pos in self.cells():
            g = self.gidx(b, pos)
            det = self.Tcell(b, pos)[1]
            for comp in range(d):
                H = sum(off[f, g] * u_star[comp, nbr[f, g]] for f in range(2 * d) if nbr[f, g] >= 0)
                S = 0.0
                for face in range(2 * d):
                    if self.at_bound(b, pos, face) and self.is_empty(b, face):
                        fn = 1.0 if face & 1 else -1.0
                        vel = self.bound_value(b, face, pos, comp)
                        a = self.alpha_bound(b, face, pos, face >> 1, face >> 1)
                        S -= vel * self.contra_bound(b, face, pos, face >> 1) * fn
                        S += vel * self.nu * 2.0 * a
                S /= det
                if source is not None:
                    S += source[comp, g]
                h[comp, g] = (u_old[comp, g] / dt - H + S) / diag[g]
        return h

    def divergence(self, h):
        div = np.zeros(self.N)
        for b, pos in self.cells():
            fl = self.fluxes(b, pos, h)
            div[self.gidx(b, pos)] = sum(fl[2 * a + 1] - fl[2 * a] for a in range(self.d))
        return div

    def pressure_nonortho(self, p, A, flags=NON_ORTHO_MODE):
        rA = 1.0 / A
        out = np.zeros(self.N)
        for b, pos in self.cells():
            out[self.gidx(b, pos)] = self.nonortho_rhs(b, pos, 0, lambda q: p[q], lambda *q: 0.0, flags, True, rA)
        return out

    # ------------------------------------------------------------------ corrector (K.cu:816-849, 5962-5995, 2176-2255)
    def pressure_gradient(self, b, pos, p):
        grad = np.zeros(self.d)
        for dim in range(self.d):
            fac = 0.5
            lo_b = pos[dim] == 0 and self.is_empty(b, 2 * dim)
            hi_b = pos[dim] == self.blocks[b].size[dim] - 1 and self.is_empty(b, 2 * dim + 1)
            g0 = self.gidx(b, pos)
            if lo_b:
                vn, fac = p[g0], 1.0
            else:
                _, b2, p2, _ = self.resolve_neighbor(b, pos, 2 * dim)[:4]
                vn = p[self.gidx(b2, p2)]
            if hi_b:
                vp, fac = p[g0], 1.0
            else:
                _, b2, p2, _ = self.resolve_neighbor(b, pos, 2 * dim + 1)[:4]
                vp = p[self.gidx(b2, p2)]
            grad[dim] = (vp - vn) * fac
        Minv, _ = self.Tcell(b, pos)
        return grad @ Minv

    def correct_velocity(self, h, p, A):
        u = np.zeros_like(h)
        for b, pos in self.cells():
            g = self.gidx(b, pos)
            u[:, g] = h[:, g] - self.pressure_gradient(b, pos, p) / A[g]
        return u

    # ------------------------------------------------------------------ linear algebra
    def apply(self, M, x):
        diag, off, nbr = M
        y = diag * x
        for f in range(off.shape[0]):
            ok = nbr[f] >= 0
            y[..., ok] += off[f, ok] * x[..., nbr[f, ok]]
        return y

    def dense(self, M):
        diag, off, nbr = M
        D = np.diag(diag)
        for f in range(off.shape[0]):
            for g in np.nonzero(nbr[f] >= 0)[0]:
                D[g, nbr[f, g]] += off[f, g]
        return D

    def solve(self, M, rhs, singular=False):
        D = self.dense(M)
        if singular:
            x = np.linalg.lstsq(D, rhs, rcond=None)[0]
            return x - x.mean()
        return np.linalg.solve(D, rhs.T).T if rhs.ndim == 2 else np.linalg.solve(D, rhs)

    # ------------------------------------------------------------------ one PISO step (SIM.py:1431-2002, non-orthogonal branch)
    def piso_step(self, u, p_result, dt, source=None, corrector_steps=2, advect_non_ortho_steps=1,
                  pressure_non_ortho_steps=1, flags=NON_ORTHO_MODE, trace: Optional[dict] = None, calls: Optional[list] = None):
        """u [d, N], p_result [N] (pressure of the previous solve: the lagged corner terms of the first pressure
        right-hand side read it, SIM.py:1841-1858).  Returns u_new, p_new.  ``calls`` (a list) receives the step's sequence of
        operators and solves in the vocabulary of the reference's backend (the solves here are direct, so the start vector the
        reference would pass is recorded, not used); ``tests/test_split_step_golden.py`` holds it against the sequence recorded
        from the reference's own ``_PISO_split_step``."""
        def log(op, **kw):
            if calls is not None:
                calls.append(dict(op=op, **kw))

        log("SetupAdvectionMatrix", non_ortho_flags=flags, for_scalar=False)
        C = self.build_matrix(u, dt, flags)
        A = C[0]
        log("CopyVelocityResultFromBlocks")
        u_res = u.copy()  # CopyVelocityResultFromBlocks (SIM.py:1707)
        for no in range(advect_non_ortho_steps):
            log("SetupAdvectionVelocity", non_ortho_flags=flags, apply_pressure_gradient=False)
            rhs = self.velocity_rhs(u, u_res, dt, source, flags)
            # x = None if no_step == 0 or not advect_non_ortho_reuse_result else velocityResult (SIM.py:1735-1742)
            log("SolveLinear", matrix="C", rhs="velocityRHS", x0=None if no == 0 else "velocityResult")
            u_res = self.solve(C, rhs)
            log("setVelocityResult")
        if trace is not None:
            trace.update(C=C, rhs=rhs, u_star=u_res.copy())
        p = p_result.copy()
        for c in range(corrector_steps):
            log("SetupPressureMatrix", non_ortho_flags=flags)
            P = self.build_pressure_matrix(A, flags)
            for ps in range(pressure_non_ortho_steps):
                if ps == 0:
                    log("SetupPressureRHS", non_ortho_flags=flags)
                    h = self.pressure_rhs_h(C, u, u_res, dt, source)
                    div = self.divergence(h)
                else:
                    log("SetupPressureRHSdiv", non_ortho_flags=flags)
                b_rhs = div + self.pressure_nonortho(p, A, flags)
                # x = None if pstep == 0 or not pressure_reuse_result else pressureResult (SIM.py:1877-1881)
                log("SolveLinear", matrix="P", rhs="pressureRHSdiv", x0=None if ps == 0 else "pressureResult")
                p = self.solve(P, b_rhs, singular=True)
                log("setPressureResult", mean_removed=True)   # solve(singular=True) returns the mean-free solution (SIM.py:1929-1932)
            if trace is not None and c == 0:
                trace.update(P=P, h=h.copy(), div=div.copy(), prhs=b_rhs.copy(), p0=p.copy())
            log("CopyPressureResultToBlocks")
            log("CorrectVelocity")
            u_res = self.correct_velocity(h, p, A)
        log("CopyVelocityResultToBlocks")
        return u_res, p

    def max_cfl_velocity(self, u):
        """max |Minv u| over cells and boundary faces (domain_structs.cpp:1580-1611)."""
        m = 0.0
        for b, pos in self.cells():
            Minv, _ = self.Tcell(b, pos)
            m = max(m, float(np.abs(Minv @ u[:, self.gidx(b, pos)]).max()))
        for b, blk in enumerate(self.blocks):
            for face, bd in enumerate(blk.bounds):
                if bd.type == FIXED:
                    for k in range(bd.velocity.shape[1]):
                        m = max(m, float(np.abs(bd.Minv[k] @ bd.velocity[:, k]).max()))
        return m
