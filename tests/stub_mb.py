"""CPU stand-in for the device half of ``fluidgym_amd.simulation.multiblock.MultiBlockDomain`` -- TEST INFRASTRUCTURE ONLY.

The mesh tables of the multi-block path are built on the host by the native library (a handle created with device < 0 serves
them without a GPU), so the REAL cylinder env -- mesh, boundary slots, jets -> boundary data, action smoothing, sensor gathers,
wall-stress forces, reward, the ``MultiBlockSimulation`` driver -- runs on the CPU once the solve itself is replaced.  That is
what this module does, so that ``ParallelFluidEnv`` can shard a multi-block env over two gloo ranks (tests/test_parallel_env_gloo.py).
The "physics" is a deterministic relaxation of every env's velocity towards its own boundary data; nothing in
``fluidgym_amd/`` imports this module."""
import ctypes

import numpy as np
import torch

from fluidgym_amd import _lib as L


CALLS = [0]   # single_step calls of every stubbed domain of this process


def install():
    """Patch MultiBlockDomain in place; returns a function that restores it."""
    import fluidgym_amd.simulation.multiblock as M

    D = M.MultiBlockDomain
    saved = {k: getattr(D, k) for k in ("__init__", "_bind", "single_step", "make_divergence_free", "set_pressure_multilevel",
                                          "set_advection_start", "env_status", "solver_hints", "solver_counters", "boundary_flux_balance", "wall_forces")}

    def __init__(self, dims, viscosity, batch=1, device=None, reference_quirks=True, non_ortho_flags=25, dtype=torch.float32):
        assert dtype == torch.float32
        self.dtype, self._np, self._cf = torch.float32, np.float32, ctypes.c_float
        self._step_opt_t, self._sim_opt_t = L.FgMbStepOptions, L.FgMbSimOptions
        self.lib = L.load()
        self.dims, self.batch = int(dims), int(batch)
        self.device = torch.device("cpu")
        self.viscosity = float(viscosity)
        self.handle = ctypes.c_void_p()
        L.check(self.lib.fg_mb_create(self.dims, self.batch, -1, ctypes.byref(self.handle)))    # host-only: tables, no compute
        self.non_ortho_flags = int(non_ortho_flags)
        self.blocks = []
        self.prepared = False
        self.velocity = self.pressure = self.boundary_velocity = self.velocity_source = None
        self.n_cells = self.n_boundary_faces = 0
        self._dt = None
        self.multilevel = None
        self.calls = 0

    def single_step(self, time_step, **kw):
        self.calls += 1
        CALLS[0] += 1
        target = self.boundary_velocity.mean(dim=2, keepdim=True)            # [B, d, 1]: env-local, deterministic
        self.velocity.mul_(0.9).add_(0.1 * target)
        self.velocity[:, 0] += 0.01 * torch.sin(torch.arange(self.n_cells, dtype=torch.float32) * 0.01)[None]
        self.pressure.copy_(self.velocity[:, 0] * self.velocity[:, 1])
        return 1, True, (1, 1, 1)

    def wall_forces(self, cell_index, slot_index, geom, area_scale, viscosity, out=None):
        """The tensor form (the reference's arithmetic) on the CPU fields, in the kernel's argument and result layout."""
        from fluidgym_amd.envs.forces import compute_forces_2d

        ci, si = cell_index.long(), slot_index.long()                                   # [layers, n]
        u, ub, p = self.velocity[:, :2][:, :, ci], self.boundary_velocity[:, :2][:, :, si], self.pressure[:, ci]   # [B, 2, L, n], [B, L, n]
        f = compute_forces_2d(u.transpose(1, 2), ub.transpose(1, 2), p, geom[:2], geom[2], geom[3], geom[4] * area_scale, viscosity)
        f = f.transpose(1, 2).contiguous()                                              # [B, L, 2] -> [B, 2, L]
        return f if out is None else out.copy_(f)

    D.__init__ = __init__
    D.wall_forces = wall_forces
    D._bind = lambda self: None
    D.single_step = single_step
    D.make_divergence_free = lambda self, **kw: True
    D.set_pressure_multilevel = lambda self, enable=True: None
    D.set_advection_start = lambda self, from_result=False: None
    D.env_status = lambda self: np.zeros(self.batch, np.int32)
    D.solver_hints = lambda self, values=None: torch.zeros(48, dtype=torch.int32)
    D.solver_counters = lambda self, reset=False: {"piso_steps": self.calls}
    D.boundary_flux_balance = lambda self: np.zeros(self.batch, np.float32)

    def restore():
        for k, v in saved.items():
            setattr(D, k, v)

    return restore
