"""Drag and lift on the cylinder wall, batched over envs.

Follows ``envs/util/forces.py`` of the reference (``wall_distance_from_vertices`` :7-37, ``collect_boundary_coords``
:40-106, ``collect_boundary_fields`` :109-190, ``compute_forces_2d`` :193-276) and its use in
``CylinderEnvBase.__prepare_drag_and_lift_computation`` / ``_get_drag_and_lift`` (cylinder_env_base.py:616-700):
traction = (2 nu S - p I) n on every wall face, S from the one-sided normal derivative (cell - wall) / distance and a
central tangential derivative over the ring of wall-adjacent cells, summed with the face lengths.  The ring is gathered
from the flat multi-block fields with index tables built once; the arithmetic is a handful of tiny tensor ops per step
(reward plumbing, not part of the solver path).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch


def wall_distance_from_vertices(vc: torch.Tensor, centers: torch.Tensor):
    """Distance of each cell centre to its wall edge and the edge normal pointing into the fluid."""
    v0, v1 = vc[:, :-1], vc[:, 1:]
    e = v1 - v0
    eps = 1e-20
    t = e / (torch.linalg.norm(e, dim=0, keepdim=True) + eps)
    n = torch.stack([t[1], -t[0]], dim=0)
    mid = 0.5 * (v0 + v1)
    d = ((centers - mid) * n).sum(dim=0).abs().clamp(min=eps)
    return d, -n


def compute_forces_2d(u_cell, u_boundary, p_cell, wall_normals, tangent_lengths, wall_distances, wall_face_lengths,
                      viscosity) -> torch.Tensor:
    """Total force [..., 2] on the wall; leading batch dimensions of u_cell [..., 2, N], u_boundary, p_cell [..., N]."""
    nx, ny = wall_normals[0], wall_normals[1]
    tx, ty = ny, -nx
    u_left = torch.roll(u_cell, shifts=-1, dims=-1)
    u_right = torch.roll(u_cell, shifts=1, dims=-1)
    dn = (u_cell - u_boundary) / wall_distances          # [..., 2, N]: d(u,v)/dn
    dt = (u_right - u_left) / (2 * tangent_lengths)       # d(u,v)/dt
    du_dx = dn[..., 0, :] * nx + dt[..., 0, :] * tx
    du_dy = dn[..., 0, :] * ny + dt[..., 0, :] * ty
    dv_dx = dn[..., 1, :] * nx + dt[..., 1, :] * tx
    dv_dy = dn[..., 1, :] * ny + dt[..., 1, :] * ty
    sxy = 0.5 * (du_dy + dv_dx)
    two_nu = 2 * viscosity
    fx = (two_nu * du_dx - p_cell) * nx + two_nu * sxy * ny
    fy = two_nu * sxy * nx + (two_nu * dv_dy - p_cell) * ny
    return torch.stack([(fx * wall_face_lengths).sum(-1), (fy * wall_face_lengths).sum(-1)], dim=-1)


def compute_forces_3d(u_cell, u_boundary, p_cell, wall_normals, tangent_lengths, wall_distances, wall_face_areas,
                      viscosity) -> torch.Tensor:
    """Force [..., 2, NZ] per spanwise layer on a wall extruded along z (``compute_forces_3d``, forces.py:278-377): the
    2-D traction of every layer from its in-plane velocity components, times the face area (edge length x layer
    height).  u_cell, u_boundary [..., 3, NZ, N]; p_cell [..., NZ, N]; the ring geometry as in the 2-D function."""
    uc = u_cell[..., :2, :, :].transpose(-3, -2)      # [..., NZ, 2, N]
    ub = u_boundary[..., :2, :, :].transpose(-3, -2)
    f = compute_forces_2d(uc, ub, p_cell, wall_normals, tangent_lengths, wall_distances, wall_face_areas, viscosity)
    return f.transpose(-1, -2)                         # [..., NZ, 2] -> [..., 2, NZ]


class WallRing:
    """Index tables of a closed wall made of one face of several blocks, in the reference's traversal order.

    ``segments``: (block, face, reverse) in ring order; ``reverse`` flips the running index of that face
    (``flip_dims`` of ``collect_boundary_coords``).
    """

    def __init__(self, domain, segments: Sequence[Tuple[int, str, bool]]):
        from ..simulation.multiblock import face_index

        cells, slots, verts, centers = [], [], [], []
        self.dims = domain.dims
        self.nz = domain.blocks[segments[0][0]].size[2] if self.dims == 3 else 1
        layer_cells, layer_slots = [], []
        for k, (b, face, reverse) in enumerate(segments):
            blk = domain.blocks[b]
            f = face_index(face)
            axis, upper = f >> 1, f & 1
            nx, ny = blk.size[0], blk.size[1]
            n_t = ny if axis == 0 else nx
            t = np.arange(n_t)
            fixed = (blk.size[axis] - 1) if upper else 0
            x = np.full(n_t, fixed) if axis == 0 else t
            y = t if axis == 0 else np.full(n_t, fixed)
            cell = blk.cell_offset + x + nx * y
            slot = blk.boundary_slot0[f] + t
            vfix = blk.size[axis] if upper else 0
            vt = np.arange(n_t + 1)
            vx = np.full(n_t + 1, vfix) if axis == 0 else vt
            vy = vt if axis == 0 else np.full(n_t + 1, vfix)
            if self.dims == 3:   # the ring geometry is that of the first layer (cylinder_env_base.py:633-638)
                v = blk.coords[:2, 0, vy, vx].astype(np.float64)
                cc = blk.getCellCoordinates()[:2, 0, y, x].astype(np.float64)
            else:
                v = blk.coords[:, vy, vx].astype(np.float64)
                cc = blk.getCellCoordinates()[:, y, x].astype(np.float64)
            layer_cells += [nx * ny] * n_t      # one spanwise layer further: cells by nx*ny, face slots by the face width
            layer_slots += [n_t] * n_t
            if reverse:
                cell, slot, v, cc = cell[::-1], slot[::-1], v[:, ::-1], cc[:, ::-1]
            if k != len(segments) - 1:
                v = v[:, :-1]  # shared vertex with the next segment
            cells.append(cell); slots.append(slot); verts.append(v); centers.append(cc)
        dev = domain.device
        cells, slots = np.concatenate(cells), np.concatenate(slots)
        if self.dims == 3:
            z = np.arange(self.nz)[:, None]
            cells = cells[None, :] + z * np.asarray(layer_cells)[None, :]      # [NZ, N]
            slots = slots[None, :] + z * np.asarray(layer_slots)[None, :]
        self.cell_index = torch.as_tensor(cells, dtype=torch.long, device=dev)
        self.slot_index = torch.as_tensor(slots, dtype=torch.long, device=dev)
        vc = torch.as_tensor(np.concatenate(verts, axis=1))
        ctr = torch.as_tensor(np.concatenate(centers, axis=1))
        dist, normals = wall_distance_from_vertices(vc, ctr)
        tl = torch.sqrt(((torch.roll(ctr, -1, -1) - torch.roll(ctr, 1, -1)) ** 2).sum(0))
        fl = torch.sqrt(((vc[:, 1:] - vc[:, :-1]) ** 2).sum(0))
        f32 = dict(dtype=torch.float32, device=dev)
        self.vertices, self.centers = vc, ctr
        self.wall_distances, self.wall_normals = dist.to(**f32), normals.to(**f32)
        self.tangent_lengths, self.face_lengths = tl.to(**f32), fl.to(**f32)
        self._native = None   # int32 index tables + packed geometry of fg_mb_wall_forces, built on first use

    def forces(self, domain, viscosity: float, layer_height: float = 1.0, out: torch.Tensor = None) -> torch.Tensor:
        """[B, 2] force on the wall from the domain's current fields; [B, 2, NZ] per spanwise layer in 3-D (``out``: a contiguous
        float32 ``[B, 2, NZ]`` tensor to write into).  One launch of
        ``fg_mb_wall_forces`` on the domain's bound fields (the tensor form below -- :func:`compute_forces_2d`, pinned on the
        reference's own function in tests/test_env_math_golden.py -- costs ~40 launches per sim step and is what the kernel is
        tested against)."""
        if self._native is None:
            i32 = dict(dtype=torch.int32, device=self.cell_index.device)
            geom = torch.stack([self.wall_normals[0], self.wall_normals[1], self.tangent_lengths, self.wall_distances,
                                self.face_lengths]).contiguous()
            self._native = (self.cell_index.reshape(self.nz, -1).to(**i32).contiguous(),
                            self.slot_index.reshape(self.nz, -1).to(**i32).contiguous(), geom)
        out = domain.wall_forces(*self._native, float(layer_height) if self.dims == 3 else 1.0, float(viscosity), out=out)   # [B, 2, NZ]
        return out if self.dims == 3 else out[:, :, 0]

    def forces_tensor_form(self, domain, viscosity: float, layer_height: float = 1.0) -> torch.Tensor:
        """The same through :func:`compute_forces_2d` / :func:`compute_forces_3d` (the reference's arithmetic, op by op)."""
        u_cell = domain.velocity[:, :, self.cell_index]
        u_b = domain.boundary_velocity[:, :, self.slot_index]
        p = domain.pressure[:, self.cell_index]
        if self.dims == 3:
            return compute_forces_3d(u_cell, u_b, p, self.wall_normals, self.tangent_lengths, self.wall_distances,
                                     self.face_lengths * layer_height, viscosity)
        return compute_forces_2d(u_cell, u_b, p, self.wall_normals, self.tangent_lengths, self.wall_distances,
                                 self.face_lengths, viscosity)
