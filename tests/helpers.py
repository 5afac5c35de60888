"""Shared builders for parity tests: the same seeded inputs go to the CPU oracle
(``oracle.piso_oracle``, fp64) and to the HIP path (``fluidgym_amd.native.NativeSolver``, fp32)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from oracle import piso_oracle as O


def stretched_edges(n: int, length: float, rng: np.random.Generator, strength: float) -> np.ndarray:
    """Monotone vertex positions with random (smooth-ish) cell-size variation."""
    w = 1.0 + strength * rng.uniform(-0.5, 0.5, size=n)
    if strength > 0:
        w = w * (1.0 + strength * np.cos(np.linspace(0, 2 * np.pi, n)))
    e = np.concatenate([[0.0], np.cumsum(w)])
    return e / e[-1] * length


@dataclass
class Case:
    dims: int
    shape: tuple  # numpy order
    edges: List[np.ndarray]
    widths: List[np.ndarray]
    fixed_faces: List[int]
    B: int
    nu: float
    velocity: np.ndarray  # [B,d,...]
    bvel: Dict[int, np.ndarray]  # face -> [B,d,slab]
    scalar: Optional[np.ndarray] = None  # [B,C,...]
    bscal: Optional[Dict[int, np.ndarray]] = None
    scalar_bc: Optional[Dict[int, List[int]]] = None
    kappa: Optional[List[float]] = None
    source: Optional[np.ndarray] = None  # [B,d,...]

    def grid(self) -> O.Grid:
        return O.Grid(O.rectilinear_coords(self.edges))

    def oracle_domain(self, b: int, grid: Optional[O.Grid] = None) -> O.Domain:
        g = grid or self.grid()
        bc = {}
        for f in self.fixed_faces:
            bc[f] = O.FixedBC(
                velocity=self.bvel[f][b].astype(np.float64),
                scalar=None if self.bscal is None else self.bscal[f][b].astype(np.float64),
                scalar_types=None if self.scalar_bc is None else self.scalar_bc.get(f),
            )
        return O.Domain(
            grid=g,
            viscosity=self.nu,
            velocity=self.velocity[b].astype(np.float64),
            pressure=np.zeros(g.shape),
            bc=bc,
            scalar=None if self.scalar is None else self.scalar[b].astype(np.float64),
            scalar_viscosity=self.kappa,
            velocity_source=None if self.source is None else self.source[b].astype(np.float64),
        )

    def native(self, device=None, dtype=None):
        """``dtype=torch.float64``: the fp64 build of the library; the fields then carry the oracle's own fp64 values."""
        import torch

        from fluidgym_amd.native import NativeSolver

        dtype = torch.float32 if dtype is None else dtype
        npt = np.float64 if dtype == torch.float64 else np.float32
        ns = NativeSolver(self.widths, self.B, fixed_faces=self.fixed_faces,
                          n_scalars=0 if self.scalar is None else self.scalar.shape[1],
                          scalar_bc=self.scalar_bc, device=device, dtype=dtype)
        dev = ns.device
        ns.set_viscosity(self.nu)
        ns.velocity.copy_(torch.from_numpy(self.velocity.astype(npt)).to(dev))
        for f in self.fixed_faces:
            ns.bvel[f].copy_(torch.from_numpy(self.bvel[f].astype(npt)).to(dev))
            if self.bscal is not None:
                ns.bscal[f].copy_(torch.from_numpy(self.bscal[f].astype(npt)).to(dev))
        if self.scalar is not None:
            ns.scalar.copy_(torch.from_numpy(self.scalar.astype(npt)).to(dev))
            for ch, k in enumerate(self.kappa or []):
                ns.set_scalar_viscosity(ch, k)
        if self.source is not None:
            ns.set_velocity_source(torch.from_numpy(self.source.astype(npt)).to(dev).contiguous())
        ns.copy_velocity_result_from_blocks()  # velocityResult starts as the block velocity
        return ns


def make_case(dims=2, n=(16, 12), fixed_axes: Sequence[int] = (), B=2, seed=0, stretch=0.3, nu=0.05,
              n_scalars=0, neumann_faces: Sequence[int] = (), with_source=False, wall_motion=0.3,
              through_flow_axis: Optional[int] = None, vel_scale=0.5) -> Case:
    """n = (nx, ny[, nz]).  Faces of ``fixed_axes`` are FIXED; boundary velocities are tangential
    wall motion (zero normal flux) unless ``through_flow_axis`` is set, in which case that axis
    carries a balanced inflow/outflow profile."""
    rng = np.random.default_rng(seed)
    lengths = [2.0, 1.0, 1.5][:dims]
    edges = [stretched_edges(n[a], lengths[a], rng, stretch) for a in range(dims)]
    # round-trip the widths through fp32 so both sides see identical metrics
    widths = [np.diff(e).astype(np.float32) for e in edges]
    edges = [np.concatenate([[0.0], np.cumsum(w.astype(np.float64))]) for w in widths]
    shape = tuple(reversed(n[:dims]))
    velocity = vel_scale * rng.standard_normal((B, dims) + shape)
    fixed_faces = sorted([2 * a for a in fixed_axes] + [2 * a + 1 for a in fixed_axes])
    bvel = {}
    for f in fixed_faces:
        a = f >> 1
        slab = list(shape)
        slab[dims - 1 - a] = 1
        v = wall_motion * rng.standard_normal((B, dims) + tuple(slab))
        v[:, a] = 0.0  # no normal flux through walls
        bvel[f] = v
    if through_flow_axis is not None:
        a = through_flow_axis
        assert a in fixed_axes
        for b in range(B):
            prof = 0.5 + 0.3 * rng.uniform(size=bvel[2 * a][b, a].shape)
            bvel[2 * a][b, a] = prof
            bvel[2 * a + 1][b, a] = prof  # same profile on a rectilinear grid => fluxes balance
    scalar = bscal = scalar_bc = kappa = None
    if n_scalars:
        scalar = rng.uniform(0, 1, size=(B, n_scalars) + shape)
        kappa = [0.03 + 0.02 * ch for ch in range(n_scalars)]
        bscal, scalar_bc = {}, {}
        for f in fixed_faces:
            slab = list(shape)
            slab[dims - 1 - (f >> 1)] = 1
            bscal[f] = rng.uniform(0, 1, size=(B, n_scalars) + tuple(slab))
            scalar_bc[f] = [O.NEUMANN if f in neumann_faces else O.DIRICHLET] * n_scalars
    source = 0.2 * rng.standard_normal((B, dims) + shape) if with_source else None
    return Case(dims, shape, edges, widths, fixed_faces, B, nu, velocity, bvel, scalar, bscal, scalar_bc, kappa, source)


def rel_err(a: np.ndarray, b: np.ndarray) -> float:
    """max |a-b| / max |b|  (the gate of SURVEY.md section 8d)."""
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / scale


def f64_twin(ns, envs, velocity, scalar=None, with_source=False):
    """The fp64 build of the library (``libfluidgym_hip_f64.so``) set up as the twin of the fp32 ``NativeSolver`` ``ns`` for its envs
    ``envs``: the same grid (the fp32 widths promoted, which is what the oracle sees), boundary data, viscosities, and the given
    start state ``velocity [B, d, ...]`` / ``scalar [B, C, ...]`` (CPU or GPU tensors of the WHOLE batch).  Used by the full-size
    tests to show that a loose fp32 bound is fp32 round-off and solver tolerance, not an error of the kernels: the same state through
    the same kernels in double lands on the fp64 oracle to ~1e-9."""
    import torch

    from fluidgym_amd.native import NativeSolver

    idx = list(envs)
    fixed = [f for f in range(2 * ns.dims) if ns.fixed[f]]
    t = NativeSolver([np.asarray(w, np.float64) for w in ns.widths], len(idx), fixed_faces=fixed, n_scalars=ns.n_scalars,
                     scalar_bc=ns.scalar_bc or None, device=ns.device, dtype=torch.float64)
    t.set_viscosity(ns.viscosity)
    for ch, k in ns.scalar_viscosities.items():
        t.set_scalar_viscosity(ch, k)
    dev = t.device
    pick = lambda a: torch.as_tensor(a)[idx].to(dev).double().contiguous()
    t.velocity.copy_(pick(velocity))
    if scalar is not None:
        t.scalar.copy_(pick(scalar))
    for f in fixed:
        bv = ns.bvel[f]
        t.bvel[f].copy_(bv[idx].double() if bv.shape[0] == ns.B else bv.double().expand_as(t.bvel[f]))
        if ns.n_scalars:
            bs = ns.bscal[f]
            t.bscal[f].copy_(bs[idx].double() if bs.shape[0] == ns.B else bs.double().expand_as(t.bscal[f]))
    if with_source:
        t.set_velocity_source(torch.zeros_like(t.velocity))
    t.copy_velocity_result_from_blocks()
    return t
