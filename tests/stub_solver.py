"""CPU stand-in for ``fluidgym_amd.native.NativeSolver`` -- TEST INFRASTRUCTURE ONLY.

It lets the real registry envs (``ChannelJet2D-*``: action -> boundary mapping, action smoothing, sensors, reward, the
``Simulation`` driver) run without a GPU so that the multi-process plumbing of ``ParallelFluidEnv`` can be covered with
world_size 2 on gloo.  The "physics" is a deterministic relaxation of the velocity towards the boundary data -- nothing
here resembles the solver, and nothing in ``fluidgym_amd/`` imports this module.
"""
from types import SimpleNamespace

import numpy as np
import torch


class StubSolver:
    def __init__(self, widths, batch, fixed_faces=(), n_scalars=0, scalar_bc=None, device=None, allocate=True, dtype=torch.float32):
        self.device = torch.device("cpu")
        self.dims = len(widths)
        self.widths = [np.ascontiguousarray(w, dtype=np.float32) for w in widths]
        self.nx, self.ny = len(self.widths[0]), len(self.widths[1])
        self.nz = len(self.widths[2]) if self.dims == 3 else 1
        self.B, self.n_scalars = int(batch), int(n_scalars)
        self.fixed = [f in fixed_faces for f in range(6)]
        self.velocity = self.pressure = self.scalar = self.velocity_source = None
        self.bvel, self.bscal = {}, {}
        self.viscosity = 0.0
        self.calls = 0
        if allocate:
            z = lambda *s: torch.zeros(s)
            self.velocity = z(self.B, self.dims, *self.spatial)
            self.pressure = z(self.B, 1, *self.spatial)
            if self.n_scalars:
                self.scalar = z(self.B, self.n_scalars, *self.spatial)
            for f in range(2 * self.dims):
                if self.fixed[f]:
                    self.bvel[f] = z(self.B, self.dims, *self.slab(f))
                    if self.n_scalars:
                        self.bscal[f] = z(self.B, self.n_scalars, *self.slab(f))

    @property
    def spatial(self):
        return (self.ny, self.nx) if self.dims == 2 else (self.nz, self.ny, self.nx)

    def slab(self, face):
        s = list(self.spatial)
        s[len(s) - 1 - (face >> 1)] = 1
        return tuple(s)

    def set_velocity(self, t):
        self.velocity = t

    def set_pressure(self, t):
        self.pressure = t

    def set_scalar(self, t):
        self.scalar = t

    def set_velocity_source(self, t):
        self.velocity_source = t

    def set_boundary_velocity(self, face, t):
        self.bvel[face] = t

    def set_boundary_scalar(self, face, t):
        self.bscal[face] = t

    def set_viscosity(self, nu):
        self.viscosity = float(nu)

    def set_scalar_viscosity(self, ch, k):
        pass

    def set_advection_start(self, from_result=True):
        self.advection_from_result = bool(from_result)

    def set_return_best(self, on=True):
        pass

    def copy_velocity_result_from_blocks(self):
        pass

    def reset_solver_state(self):
        pass

    def make_divergence_free(self, tol=1e-5, max_iterations=1000):
        return [SimpleNamespace(converged=1, is_finite=1, used_iterations=0, final_residual=0.0) for _ in range(self.B)]

    def single_step(self, time_step, cfl, adaptive=True, substeps=1, **kw):
        """Relax every env's velocity towards the mean of its own boundary data (env-local, deterministic)."""
        self.calls += 1
        target = sum(t.mean(dim=tuple(range(2, t.dim())), keepdim=True) for t in self.bvel.values())
        self.velocity.mul_(0.9).add_(0.1 * target)
        self.pressure.copy_(self.velocity[:, :1] * self.velocity[:, 1:2])
        return True, [-1, 1, 1, 1], 1

    def close(self):
        pass
