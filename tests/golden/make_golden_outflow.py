"""Golden vectors for the advective-outflow boundary update and the boundary-flux balancing, produced by the reference's OWN
Python (``pict/PISOtorch_simulation.py``: ``update_advective_boundaries`` :228-393, ``get_advective_velocity`` :146-185,
``balance_boundary_fluxes`` :188-224) running HERE on CPU tensors against stand-in domain objects.

    python tests/golden/make_golden_outflow.py        ->  tests/golden/reference_outflow.npz

The stand-ins replace only what lives in the CUDA extension: the ``PISOtorch`` classes the functions test with ``isinstance``,
the container methods they call (``getBlock``, ``getBoundary``, ``getFixedBoundaries``, ``getSizes`` ...), the boundary
transform of a RECTILINEAR block (M = diag(h), Minv = diag(1/h), det = prod h, layout [M | Minv | det] as
``domain_structs_gpu.h:160-168``) and ``FixedBoundary::GetFluxes`` (det * Minv_row_axis . u_b per face cell,
``domain_structs.cpp:1825-1881`` / ``PISO_multiblock_cuda_kernel.cu:495-510``).  Everything the fixture pins -- the interpolation
weight, which slab of cells is read, the scaling rule and its threshold, scale-all-components -- is executed from the reference
file itself; no reference source is copied.
"""
import importlib.util
import os
import sys
import types
from contextlib import nullcontext
from enum import Enum

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


class BoundaryConditionType(Enum):
    DIRICHLET = 0
    NEUMANN = 1


class FixedBoundary:
    """Stand-in for PISOtorch.FixedBoundary on a rectilinear block."""

    def __init__(self, domain, block, face, velocity, scalar=None):
        self.domain, self.block, self.face = domain, block, face
        self.velocity = velocity                      # [1, d, (Z,) Y, X] with extent 1 along the face axis
        self.passiveScalar = scalar
        self.isVelocityStatic = False
        self.velocityType = BoundaryConditionType.DIRICHLET
        self.passiveScalarTypes = [BoundaryConditionType.DIRICHLET] * (0 if scalar is None else scalar.shape[1])

    def getSpatialDims(self):
        return self.domain.dims

    def getParentDomain(self):
        return self.domain

    def getSizes(self):                               # x, y(, z)
        return [self.velocity.shape[-1 - a] for a in range(self.domain.dims)]

    def hasTransform(self):
        return True

    def hasPassiveScalar(self):
        return self.passiveScalar is not None

    def isPassiveScalarStatic(self):
        return False

    @property
    def transform(self):
        """[1, (Z,) Y, X, 2 d^2 + 1] of the boundary face: the adjacent cell layer's widths (grid_gen.cu:423-452)."""
        d = self.domain.dims
        axis = self.face >> 1
        h = []
        for a in range(d):
            w = torch.as_tensor(self.block.widths[a], dtype=torch.float32)
            if a == axis:
                w = w[-1:] if (self.face & 1) else w[:1]
            shape = [1] * d
            shape[d - 1 - a] = w.numel()
            h.append(w.reshape(shape))
        sp = [self.velocity.shape[2 + k] for k in range(d)]
        h = [x.expand(*sp) for x in h]
        t = torch.zeros(*sp, 2 * d * d + 1)
        for a in range(d):
            t[..., a * d + a] = h[a]
            t[..., d * d + a * d + a] = 1.0 / h[a]
        det = h[0].clone()
        for a in range(1, d):
            det = det * h[a]
        t[..., 2 * d * d] = det
        return t.unsqueeze(0)

    def GetFluxes(self):
        d = self.domain.dims
        axis = self.face >> 1
        t = self.transform[0]
        return self.velocity[0, axis] * t[..., 2 * d * d] * t[..., d * d + axis * d + axis]

    def setVelocity(self, v):
        self.velocity = v

    def setPassiveScalar(self, s):
        self.passiveScalar = s


class _Dummy:
    pass


class Block:
    def __init__(self, domain, widths, velocity, scalar=None):
        self.domain, self.widths, self.velocity, self.passiveScalar = domain, widths, velocity, scalar
        self.bounds = {}

    def hasPassiveScalar(self):
        return self.passiveScalar is not None

    def getBoundary(self, idx):
        return self.bounds.get(idx, _Dummy())

    def getFixedBoundaries(self):
        return sorted(self.bounds.items())


class Domain:
    def __init__(self, dims):
        self.dims, self.blocks = dims, []

    def getSpatialDims(self):
        return self.dims

    def getNumBlocks(self):
        return len(self.blocks)

    def getBlock(self, i):
        return self.blocks[i]

    def getBlocks(self):
        return self.blocks

    def getDtype(self):
        return torch.float32

    def getDevice(self):
        return torch.device("cpu")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_simulation_module():
    for pkg in ["fluidgym", "fluidgym.simulation", "fluidgym.simulation.pict", "fluidgym.simulation.pict.util",
                "fluidgym.simulation.pict.data"]:
        _stub(pkg).__path__ = []
    class _Ext(types.ModuleType):
        """every other name of the extension (only used in type annotations at import time) resolves to a dummy class"""

        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return type(name, (), {})

    pt = _Ext("PISOtorch")
    pt.__dict__.update(FixedBoundary=FixedBoundary, VaryingDirichletBoundary=type("VaryingDirichletBoundary", (), {}),
                       StaticDirichletBoundary=type("StaticDirichletBoundary", (), {}),
                       BoundaryConditionType=BoundaryConditionType, Domain=Domain)
    _stub("fluidgym.simulation.extensions", PISOtorch=pt)
    _stub("fluidgym.simulation.pict.util.profiling", SAMPLE=lambda *a, **k: nullcontext())
    _stub("fluidgym.simulation.pict.util.output")
    _stub("fluidgym.simulation.pict.util.outputVtk", save_vtk=lambda *a, **k: None)
    _stub("fluidgym.simulation.pict.util.domain_io", save_domain=lambda *a, **k: None)
    import logging
    _stub("fluidgym.simulation.pict.util.logging", get_logger=lambda name="": logging.getLogger(name))

    def _get_solver_tolerance(tol, dtype=torch.float32):     # PISOtorch_diff.py:247-253 (the module needs the extension to import)
        if tol is None:
            return 1e-8 if dtype == torch.float64 else 1e-5
        return tol

    diff = _stub("fluidgym.simulation.pict.PISOtorch_diff", _get_solver_tolerance=_get_solver_tolerance)
    sys.modules["fluidgym.simulation.pict"].PISOtorch_diff = diff
    return _load(f"{REF}/fluidgym/simulation/pict/PISOtorch_simulation.py", "ref_PISOtorch_simulation")


def make_case(dims, n, seed, free_faces, with_scalar=False, stretch=1.0):
    rng = np.random.default_rng(seed)
    widths = []
    for a in range(dims):
        w = (0.5 + rng.random(n[a])) * (stretch if a == 0 else 1.0) / n[a]
        widths.append(w.astype(np.float32))
    sp = [n[dims - 1 - k] for k in range(dims)]                  # (Z,) Y, X
    vel = torch.as_tensor((0.4 * rng.standard_normal([1, dims] + sp)).astype(np.float32))
    vel[:, 0] += 1.0
    scal = torch.as_tensor(rng.random([1, 1] + sp).astype(np.float32)) if with_scalar else None
    dom = Domain(dims)
    blk = Block(dom, widths, vel, scal)
    dom.blocks.append(blk)
    for f in range(2 * dims):
        slab = list(sp)
        slab[dims - 1 - (f >> 1)] = 1
        bv = torch.as_tensor((0.3 * rng.standard_normal([1, dims] + slab)).astype(np.float32))
        bv[:, f >> 1] += 0.8 if f < 2 else 0.0                    # a through-flow along x
        bs = torch.as_tensor(rng.random([1, 1] + slab).astype(np.float32)) if with_scalar else None
        blk.bounds[f] = FixedBoundary(dom, blk, f, bv, bs)
    return dom, blk, [blk.bounds[f] for f in free_faces]


def main():
    sim = load_reference_simulation_module()
    out = {}
    cases = [
        ("c2d", dict(dims=2, n=(12, 8), seed=1, free_faces=[1]), torch.tensor([[1.0, 0.0]]), 0.02, None),
        ("c2d_two_free", dict(dims=2, n=(10, 6), seed=2, free_faces=[1, 3]), [torch.tensor([[1.0, 0.1]]), torch.tensor([[0.2, 0.7]])], 0.05, None),
        ("c2d_scalar", dict(dims=2, n=(9, 7), seed=3, free_faces=[1], with_scalar=True, stretch=2.0), torch.tensor([[0.8, 0.0]]), 0.03, 1e-5),
        ("c3d", dict(dims=3, n=(8, 6, 5), seed=4, free_faces=[1]), torch.tensor([[1.0, 0.0, 0.0]]), 0.02, None),
    ]
    for name, kw, velm, dt, tol in cases:
        dom, blk, free = make_case(**kw)
        d = dom.dims
        out[f"{name}_dims"] = np.array(d)
        out[f"{name}_free_faces"] = np.array(kw["free_faces"])
        out[f"{name}_dt"] = np.array(dt, np.float32)
        out[f"{name}_tol"] = np.array(-1.0 if tol is None else tol)
        out[f"{name}_velm"] = np.stack([v.numpy()[0] for v in (velm if isinstance(velm, list) else [velm])])
        for a in range(d):
            out[f"{name}_h{a}"] = blk.widths[a]
        out[f"{name}_velocity"] = blk.velocity.numpy()[0].copy()
        if blk.passiveScalar is not None:
            out[f"{name}_scalar"] = blk.passiveScalar.numpy()[0].copy()
        for f, b in blk.bounds.items():
            out[f"{name}_bvel_in_{f}"] = b.velocity.numpy()[0].copy()
            if b.passiveScalar is not None:
                out[f"{name}_bscal_in_{f}"] = b.passiveScalar.numpy()[0].copy()
        sim.update_advective_boundaries(dom, free, velm, dt, tol=tol)          # includes balance_boundary_fluxes
        for f, b in blk.bounds.items():
            out[f"{name}_bvel_out_{f}"] = b.velocity.numpy()[0].copy()
            if b.passiveScalar is not None:
                out[f"{name}_bscal_out_{f}"] = b.passiveScalar.numpy()[0].copy()
        # the balancing alone on the updated state must be a no-op now (flux already balanced)
        before = {f: b.velocity.clone() for f, b in blk.bounds.items()}
        sim.balance_boundary_fluxes(dom, free, tol=tol)
        out[f"{name}_rebalance_is_noop"] = np.array(all(torch.equal(before[f], b.velocity) for f, b in blk.bounds.items()))
    np.savez_compressed(os.path.join(OUT, "reference_outflow.npz"), **out)
    print("wrote reference_outflow.npz:", len(out), "arrays;", {k: bool(v) for k, v in out.items() if k.endswith("noop")})


if __name__ == "__main__":
    main()
