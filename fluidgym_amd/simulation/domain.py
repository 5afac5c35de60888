"""Batched ``Domain`` / ``Block`` / ``FixedBoundary`` facade over the HIP solver handle.

Names and call order follow the reference's PISOtorch object model so that env code reads like the
reference's (``extensions/domain_structs.h:73-802``, bindings ``extensions/PISOtorch.cpp:98-505``;
conventions in SURVEY.md Appendix B):

* faces are ``-x,+x,-y,+y,-z,+z`` = 0..2d-1, string forms accepted (``domain_structs.cpp:143-158``);
* a new block is fully periodic (``domain_structs.cpp:1161-1165``); ``CloseBoundary`` installs a
  FIXED Dirichlet boundary on the face *and* its partner (``PISOtorch.cpp:343-350``);
* every field tensor has the env batch as leading axis ``[B, C, (Z,) Y, X]`` where the reference has
  ``[1, C, ...]``; setters accept ``[1, ...]`` (broadcast over the batch) or ``[B, ...]``;
* setters copy *into* the bound device buffers, so ``UpdateDomainData()`` (which re-packs and
  re-uploads the reference's atlas, ``domain_structs.cpp:3047-3283``) is a no-op kept for
  call-compatibility.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Union

import numpy as np
import torch

from .. import _lib as L
from ..native import NativeSolver, coords_to_transforms
from . import grids

_FACE_NAMES = {"-x": 0, "+x": 1, "-y": 2, "+y": 3, "-z": 4, "+z": 5}


def face_index(face: Union[int, str]) -> int:
    if isinstance(face, str):
        return _FACE_NAMES[face]
    return int(face)


class BoundaryConditionType:
    DIRICHLET = L.FG_DIRICHLET
    NEUMANN = L.FG_NEUMANN


def _bcast_into(dst: torch.Tensor, src: torch.Tensor):
    """Copy ``src`` into ``dst`` allowing a leading batch of 1 and static ``[N, C]`` data."""
    src = src.to(device=dst.device, dtype=dst.dtype)
    if src.dim() == 2:  # static [N, C] -> broadcast over space
        src = src.reshape(src.shape[0], src.shape[1], *([1] * (dst.dim() - 2)))
    dst.copy_(src.expand_as(dst))


class FixedBoundary:
    """FIXED boundary of one face (``domain_structs.h`` FixedBoundary): Dirichlet velocity and a
    per-channel Dirichlet/Neumann passive scalar."""

    def __init__(self, block: "Block", face: int):
        self._block = block
        self.face = face
        self.passiveScalarTypes: List[int] = [BoundaryConditionType.DIRICHLET] * block.domain.n_scalars
        self._pending_velocity = None
        self._pending_scalar = None

    # tensors live in the solver once the domain is prepared
    @property
    def velocity(self) -> torch.Tensor:
        return self._block.domain.solver.bvel[self.face]

    @property
    def passiveScalar(self) -> torch.Tensor:
        return self._block.domain.solver.bscal[self.face]

    def setVelocity(self, v: torch.Tensor):
        if self._block.domain.solver is None:
            self._pending_velocity = v
        else:
            _bcast_into(self.velocity, v)

    def setPassiveScalar(self, s: torch.Tensor):
        if self._block.domain.solver is None:
            self._pending_scalar = s
        else:
            _bcast_into(self.passiveScalar, s)

    def setPassiveScalarType(self, types):
        assert self._block.domain.solver is None, "scalar boundary types are fixed at PrepareSolve()"
        self.passiveScalarTypes = [int(t) for t in types]

    # reference API parity helpers
    isVelocityStatic = False

    def makeVelocityVarying(self):  # always varying here
        return None

    def isPassiveScalarStatic(self) -> bool:
        return False

    def hasPassiveScalar(self) -> bool:
        return self._block.domain.n_scalars > 0

    def hasTransform(self) -> bool:
        return True

    def getSizes(self):
        return self._block.domain.solver.slab(self.face)

    @property
    def transform(self) -> torch.Tensor:
        """Boundary transform ``[1, slab, 2d^2+1]``; on a rectilinear grid it equals the adjacent cell
        layer's transform (``grid_gen.cu:423-452``: one-sided normal extent, tangential edge lengths)."""
        t = self._block.transform
        axis = self.face >> 1
        dim = t.dim() - 2 - axis  # spatial axis position in [1,(Z,)Y,X,T]
        idx = t.shape[dim] - 1 if (self.face & 1) else 0
        return t.narrow(dim, idx, 1)

    def GetFluxes(self) -> torch.Tensor:
        """Contravariant boundary flux ``det_b * (Minv_row_axis . u_b)`` per boundary cell ``[B, slab]``."""
        d = self._block.domain.dims
        axis = self.face >> 1
        tr = self.transform[0]
        det = tr[..., 2 * d * d]
        minv_aa = tr[..., d * d + axis * d + axis]
        return self.velocity[:, axis] * (det * minv_aa).unsqueeze(0)


class Block:
    def __init__(self, domain: "Domain", coords: torch.Tensor, name: str = "Block"):
        self.domain = domain
        self.name = name
        self.vertexCoordinates = coords
        self.edges = grids.edges_from_vertex_grid(coords)
        self.widths = [np.diff(e).astype(np.float64 if domain.dtype == torch.float64 else np.float32) for e in self.edges]
        d = domain.dims
        assert len(self.edges) == d
        self._fixed: Dict[int, FixedBoundary] = {}
        self._transform = None
        self._pending: Dict[str, torch.Tensor] = {}

    # ---- topology -------------------------------------------------------------------------
    def CloseBoundary(self, face, velocity: Optional[torch.Tensor] = None, passiveScalar: Optional[torch.Tensor] = None):
        assert self.domain.solver is None, "boundaries must be closed before PrepareSolve()"
        f = face_index(face)
        for ff in (f, f ^ 1):  # the periodic partner closes too (domain_structs.cpp:1981-2002)
            if ff not in self._fixed:
                self._fixed[ff] = FixedBoundary(self, ff)
        if velocity is not None:
            self._fixed[f].setVelocity(velocity)
        if passiveScalar is not None:
            self._fixed[f].setPassiveScalar(passiveScalar)
        return self._fixed[f]

    def getBoundary(self, face) -> FixedBoundary:
        f = face_index(face)
        if f not in self._fixed:
            raise KeyError(f"face {face} is periodic; only FIXED faces expose a boundary object")
        return self._fixed[f]

    def getFixedBoundaries(self):
        return sorted(self._fixed.items())

    def isFixed(self, face) -> bool:
        return face_index(face) in self._fixed

    # ---- fields ---------------------------------------------------------------------------
    def _field(self, name):
        return getattr(self.domain.solver, name)

    @property
    def velocity(self) -> torch.Tensor:
        return self._field("velocity")

    @property
    def pressure(self) -> torch.Tensor:
        return self._field("pressure")

    @property
    def passiveScalar(self) -> torch.Tensor:
        return self._field("scalar")

    @property
    def velocitySource(self) -> Optional[torch.Tensor]:
        return self._field("velocity_source")

    def getVelocity(self, computational: bool = False) -> torch.Tensor:
        if not computational:
            return self.velocity
        rh = self.domain.inv_widths_tensors()
        return self.velocity * rh

    def _set(self, name, t):
        if self.domain.solver is None:
            self._pending[name] = t
        else:
            _bcast_into(getattr(self.domain.solver, name), t)

    def setVelocity(self, v):
        self._set("velocity", v)

    def setPressure(self, p):
        self._set("pressure", p)

    def setPassiveScalar(self, s):
        self._set("scalar", s)

    def hasPassiveScalar(self) -> bool:
        return self.domain.n_scalars > 0

    def setVelocitySource(self, s: Optional[torch.Tensor]):
        sol = self.domain.solver
        if sol is None:
            self._pending["velocity_source"] = s
            return
        if s is None:
            sol.set_velocity_source(None)
            return
        if sol.velocity_source is None:
            sol.set_velocity_source(torch.zeros_like(sol.velocity))
        _bcast_into(sol.velocity_source, s)

    def setViscosity(self, visc: Optional[torch.Tensor]):
        """``Block.setViscosity``: a per-cell viscosity of the velocity system, ``[1 | B, 1, (Z,) Y, X]`` (the reference's layout) or
        ``[B, (Z,) Y, X]``; None returns to the domain's viscosity.  Used by the Smagorinsky hook of the TCF env
        (tcf_env.py:441-474; the kernels read it through getViscosityBlock, PISO_multiblock_cuda_kernel.cu:1816-1837)."""
        sol = self.domain.solver
        if sol is None:
            raise RuntimeError("setViscosity: PrepareSolve() first")
        if visc is None:
            sol.set_viscosity_field(None)
            return
        if getattr(sol, "viscosity_field", None) is None:
            sol.set_viscosity_field(torch.empty((sol.B,) + tuple(sol.spatial), dtype=sol.dtype, device=sol.device))
        v = visc.to(sol.device, sol.dtype)
        if v.dim() == len(sol.spatial) + 2:      # NCDHW with C = 1
            v = v[:, 0]
        sol.viscosity_field.copy_(v.expand_as(sol.viscosity_field))

    @property
    def viscosity(self) -> Optional[torch.Tensor]:
        return getattr(self.domain.solver, "viscosity_field", None)

    # ---- geometry -------------------------------------------------------------------------
    def getSizes(self):
        s = self.domain.solver
        return (s.nx, s.ny, s.nz)

    @property
    def transform(self) -> torch.Tensor:
        """``[1,(Z,)Y,X,2d^2+1]`` = M | Minv | det per cell (``domain_structs_gpu.h:160-168``), computed
        by the native ``fg_coords_to_transforms`` kernel."""
        if self._transform is None:
            dev = self.domain.device
            self._transform = coords_to_transforms(self.vertexCoordinates.to(dev, torch.float32).contiguous())
        return self._transform

    def hasTransform(self) -> bool:
        return True

    def getCellCoordinates(self) -> torch.Tensor:
        """Cell-centre coordinates ``[1, d, (Z,) Y, X]`` (mean of the cell's vertices)."""
        centers = [0.5 * (e[1:] + e[:-1]) for e in self.edges]
        d = len(centers)
        mesh = np.meshgrid(*[c for c in reversed(centers)], indexing="ij")
        c = np.stack([mesh[d - 1 - a] for a in range(d)], axis=0)[None]
        return torch.from_numpy(c).to(self.domain.device, self.domain.dtype)

    def getCellSizes(self) -> torch.Tensor:
        """Cell volumes ``[1, 1, (Z,) Y, X]`` (the determinant of the cell transform)."""
        d = self.domain.dims
        return self.transform[..., 2 * d * d].unsqueeze(1)


class Domain:
    """Batched stand-in for ``PISOtorch.Domain`` (one block, rectilinear grid)."""

    def __init__(self, spatialDims: int, viscosity, passiveScalarChannels: int = 0, name: str = "Domain",
                 device=None, dtype=torch.float32, batch: int = 1):
        if dtype not in (torch.float32, torch.float64):
            raise NotImplementedError("field dtype must be torch.float32 (the reference's default, fluid_env.py:146) or torch.float64")
        # torch.float64: the fields, metrics and every kernel of the step run in fp64 on the fp64 build of the library
        # (libfluidgym_hip_f64.so); the pressure CG is preconditioned by the fast-diagonalisation operator there as well (plain
        # kernels in doubles, csrc/fg_f64_fd.hip), the fp32-tuned kernel families are not part of that build
        self.dims = int(spatialDims)
        self.name = name
        self.dtype = dtype
        self.device = torch.device("cuda") if device is None else torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self.batch = int(batch)
        self.n_scalars = int(passiveScalarChannels)
        self._viscosity = float(torch.as_tensor(viscosity).reshape(-1)[0])
        self._scalar_viscosity: Optional[List[float]] = None
        self.blocks: List[Block] = []
        self.solver: Optional[NativeSolver] = None
        self._rh = None

    # ---- construction ---------------------------------------------------------------------
    def CreateBlock(self, vertexCoordinates: torch.Tensor, name: str = "Block") -> Block:
        if self.blocks:
            raise NotImplementedError("this class is the single-block (rectilinear) domain; connected curvilinear blocks are built "
                                      "with fluidgym_amd.simulation.multiblock.MultiBlockDomain (SURVEY 8f-3)")
        b = Block(self, vertexCoordinates, name)
        self.blocks.append(b)
        return b

    def setScalarViscosity(self, v):
        vals = [float(x) for x in torch.as_tensor(v).reshape(-1)]
        self._scalar_viscosity = vals
        if self.solver is not None:
            for ch in range(self.n_scalars):
                self.solver.set_scalar_viscosity(ch, vals[0] if len(vals) == 1 else vals[ch])

    @property
    def viscosity(self) -> torch.Tensor:
        return torch.tensor([self._viscosity], dtype=self.dtype)

    def setViscosity(self, v):
        self._viscosity = float(torch.as_tensor(v).reshape(-1)[0])
        if self.solver is not None:
            self.solver.set_viscosity(self._viscosity)

    def PrepareSolve(self):
        """Allocate the solver workspace and bind all fields (``Domain::PrepareSolve``,
        ``domain_structs.cpp:2570-2693``)."""
        assert self.blocks, "domain has no block"
        blk = self.blocks[0]
        fixed = sorted(blk._fixed.keys())
        scalar_bc = {f: blk._fixed[f].passiveScalarTypes for f in fixed} if self.n_scalars else None
        self.solver = NativeSolver(blk.widths, self.batch, fixed_faces=fixed, n_scalars=self.n_scalars,
                                   scalar_bc=scalar_bc, device=self.device, dtype=self.dtype)
        self.solver.set_viscosity(self._viscosity)
        if self._scalar_viscosity is not None:
            self.setScalarViscosity(self._scalar_viscosity)
        for name, t in blk._pending.items():
            if name == "velocity_source":
                blk.setVelocitySource(t)
            elif t is not None:
                blk._set(name, t)
        blk._pending.clear()
        for f, bnd in blk._fixed.items():
            if bnd._pending_velocity is not None:
                bnd.setVelocity(bnd._pending_velocity)
            if bnd._pending_scalar is not None and self.n_scalars:
                bnd.setPassiveScalar(bnd._pending_scalar)
            bnd._pending_velocity = bnd._pending_scalar = None
        self.solver.copy_velocity_result_from_blocks()

    def IsInitialized(self) -> bool:
        return self.solver is not None

    def UpdateDomainData(self):
        return None

    # ---- queries --------------------------------------------------------------------------
    def getBlock(self, i: int) -> Block:
        return self.blocks[i]

    def getBlocks(self):
        return list(self.blocks)

    def getNumBlocks(self) -> int:
        return len(self.blocks)

    def getSpatialDims(self) -> int:
        return self.dims

    def getDtype(self):
        return self.dtype

    def hasPassiveScalar(self) -> bool:
        return self.n_scalars > 0

    def getPassiveScalarChannels(self) -> int:
        return self.n_scalars

    def inv_widths_tensors(self) -> torch.Tensor:
        """``1/h`` per component broadcastable against ``[B, d, (Z,) Y, X]``."""
        if self._rh is None:
            blk = self.blocks[0]
            d = self.dims
            comps = []
            for a in range(d):
                shape = [1] * d
                shape[d - 1 - a] = len(blk.widths[a])
                comps.append(torch.from_numpy(1.0 / blk.widths[a]).reshape(shape).expand(*self.solver.spatial))
            self._rh = torch.stack(comps, 0).unsqueeze(0).to(self.device).contiguous()
        return self._rh

    def getMaxVelocity(self, withBounds: bool = True, computational: bool = True) -> torch.Tensor:
        """Per-env ``[B]`` (reference: scalar; ``domain_structs.cpp:1580-1611``)."""
        assert withBounds and computational, "only the variant the stepper uses is implemented natively"
        return self.solver.max_velocity()

    def GetBoundaryFluxBalance(self) -> torch.Tensor:
        return self.solver.boundary_flux_balance()

    def Clone(self) -> dict:
        """State snapshot used by ``FluidEnv.get_state`` (reference: ``Domain.Clone()``)."""
        s = self.solver
        snap = {"velocity": s.velocity.clone(), "pressure": s.pressure.clone()}
        if s.scalar is not None:
            snap["scalar"] = s.scalar.clone()
        if s.velocity_source is not None:
            snap["velocity_source"] = s.velocity_source.clone()
        snap["bvel"] = {f: t.clone() for f, t in s.bvel.items()}
        snap["bscal"] = {f: t.clone() for f, t in s.bscal.items()}
        if hasattr(s, "solver_hints"):     # which iteration the next solves run (the sweeps' back-off): part of a bit-exact replay
            snap["solver_hints"] = s.solver_hints()
        return snap

    def Restore(self, snap: dict):
        s = self.solver
        s.velocity.copy_(snap["velocity"])
        s.pressure.copy_(snap["pressure"])
        if "scalar" in snap:
            s.scalar.copy_(snap["scalar"])
        if "velocity_source" in snap and s.velocity_source is not None:
            s.velocity_source.copy_(snap["velocity_source"])
        for f, t in snap["bvel"].items():
            s.bvel[f].copy_(t)
        for f, t in snap["bscal"].items():
            s.bscal[f].copy_(t)
        if "solver_hints" in snap and hasattr(s, "solver_hints"):
            s.solver_hints(snap["solver_hints"])
        s.copy_velocity_result_from_blocks()
