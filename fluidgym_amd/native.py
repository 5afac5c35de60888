"""Thin object wrapper over the C ABI: one :class:`NativeSolver` per (grid, batch) on one GPU.

PyTorch-ROCm tensors are used only as device-memory owners (zero-copy ``data_ptr()``) and for the
current HIP stream; all arithmetic happens in ``libfluidgym_hip.so``.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class LinsolveError(RuntimeError):
    """A linear solve failed (reference: ``PISOtorch_diff.LinsolveError``, PISOtorch_diff.py:262)."""


class NativeSolver:
    """Owns an ``fg_handle`` and the field tensors bound to it.

    Parameters mirror ``fg_config``: ``faces`` maps face index (0..2d-1 = -x,+x,-y,+y,-z,+z) to
    ``"periodic"`` / ``"fixed"``; ``scalar_bc[face][channel]`` is ``FG_DIRICHLET``/``FG_NEUMANN``.
    ``widths`` = per-axis cell widths ``[hx, hy(, hz)]`` (rectilinear grid).
    """

    def __init__(self, widths: Sequence[np.ndarray], batch: int, fixed_faces: Sequence[int] = (),
                 n_scalars: int = 0, scalar_bc: Optional[Dict[int, Sequence[int]]] = None,
                 device: Optional[torch.device] = None, allocate: bool = True, dtype: torch.dtype = torch.float32):
        if not torch.cuda.is_available():
            raise L.NativeLibraryError("fluidgym_amd needs a ROCm GPU (MI355X); there is no CPU path")
        if dtype not in (torch.float32, torch.float64):
            raise ValueError("dtype must be torch.float32 or torch.float64")
        # fp64 fields (the reference's FluidEnv(dtype=torch.float64), envs/fluid_env.py:146) run on the fp64 build of the same
        # sources (libfluidgym_hip_f64.so: fg_real = double); its entry points take doubles wherever this one takes floats
        self.dtype = dtype
        self.f64 = dtype == torch.float64
        self.lib = L.load_f64() if self.f64 else L.load()
        self._np_real = np.float64 if self.f64 else np.float32
        self._c_real = ctypes.c_double if self.f64 else ctypes.c_float
        self._StepOptions = L.FgStepOptionsF64 if self.f64 else L.FgStepOptions
        self._SimOptions = L.FgSimOptionsF64 if self.f64 else L.FgSimOptions
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dims = len(widths)
        assert self.dims in (2, 3)
        self.widths = [np.ascontiguousarray(w, dtype=self._np_real) for w in widths]
        self.nx = len(self.widths[0])
        self.ny = len(self.widths[1])
        self.nz = len(self.widths[2]) if self.dims == 3 else 1
        self.n = self.nx * self.ny * self.nz
        self.B = int(batch)
        self.n_scalars = int(n_scalars)
        self.scalar_bc = {int(f): [int(t) for t in v] for f, v in (scalar_bc or {}).items()}
        self.scalar_viscosities: Dict[int, float] = {}
        self.fixed = [f in fixed_faces for f in range(6)]
        cfg = L.FgConfig()
        cfg.dims, cfg.nx, cfg.ny, cfg.nz = self.dims, self.nx, self.ny, self.nz
        cfg.batch, cfg.n_scalars = self.B, self.n_scalars
        cfg.device = self.device.index or 0
        for f in range(6):
            cfg.face_type[f] = L.FG_FIXED if self.fixed[f] else L.FG_PERIODIC
            for ch in range(L.FG_MAX_SCALARS):
                cfg.scalar_bc[f][ch] = L.FG_DIRICHLET
            if scalar_bc and f in scalar_bc:
                for ch, t in enumerate(scalar_bc[f]):
                    cfg.scalar_bc[f][ch] = int(t)
        fp = ctypes.POINTER(self._c_real)
        hz = self.widths[2].ctypes.data_as(fp) if self.dims == 3 else None
        handle = ctypes.c_void_p()
        L.check(self.lib.fg_create(ctypes.byref(cfg), self.widths[0].ctypes.data_as(fp),
                                   self.widths[1].ctypes.data_as(fp), hz, ctypes.byref(handle)))
        self.handle = handle
        self._bound: Dict[int, torch.Tensor] = {}
        self.viscosity = 0.0
        # fields
        self.velocity = self.pressure = self.scalar = self.velocity_source = None
        self.viscosity_field = None
        self.bvel: Dict[int, torch.Tensor] = {}
        self.bscal: Dict[int, torch.Tensor] = {}
        self._dt = torch.zeros(self.B, dtype=dtype, device=self.device)
        self._dt_host = torch.zeros(self.B, dtype=dtype).pin_memory()
        self._out_B = torch.zeros(self.B, dtype=dtype, device=self.device)
        # fast-diagonalisation preconditioner (needs FIXED y faces); default pressure solver when available.  The fp32 library applies it
        # with MFMA basis changes / LDS FFTs; the fp64 library applies the same operator with plain kernels in doubles (round 6,
        # csrc/fg_f64_fd.hip: until then it ran the reference's plain CG -- 250 000 iterations per solve on the refined 512 x 256 grid)
        # (FLUIDGYM_AMD_F64_FD=0: the fp64 library's plain CG, as before -- A/B runs)
        self.has_fd = bool(self.fixed[2] and self.fixed[3]) and not (self.f64 and os.environ.get("FLUIDGYM_AMD_F64_FD", "1") == "0")
        self.has_helmholtz = False
        if self.has_fd:
            fp = ctypes.POINTER(ctypes.c_float)
            from .simulation.fd_precond import FDPreconditioner

            fd = FDPreconditioner(self.widths, [f for f in range(6) if self.fixed[f]])
            fpp = lambda a: a.ctypes.data_as(fp)
            qz, qzt = (fpp(fd.Qz), fpp(fd.QzT)) if self.dims == 3 else (None, None)
            L.check(self.lib.fg_set_fd_preconditioner(self.handle, fpp(fd.Qx), fpp(fd.QxT), qz, qzt, fpp(fd.lower),
                                                      fpp(fd.inv), fpp(fd.cp)))
            no_fft = self.f64 or os.environ.get("FG_FD_NO_FFT", "0") != "0"      # (the row FFTs and the Helmholtz operator are fp32 kernels)
            if fd.x_cosine_width is not None and not no_fft:
                # the x basis is the DCT-II basis: apply it as a fast cosine transform instead of the dense GEMM
                L.check(self.lib.fg_set_fd_fast_transform(self.handle, 0, fd.x_cosine_width), lib=self.lib)
            # Helmholtz preconditioner of the advection-diffusion solves (mode 3 of set_advection_preconditioner): available when
            # the transform axes are periodic and uniform (RBC, TCF)
            if fd.x_fourier_width is not None and not no_fft:
                # periodic uniform x: the real Fourier basis, applied as one FFT per row
                L.check(self.lib.fg_set_fd_fast_transform(self.handle, 0, fd.x_fourier_width), lib=self.lib)
            self.has_helmholtz = bool(fd.transform_axes_periodic_uniform) and not self.f64
            if self.has_helmholtz:
                L.check(self.lib.fg_set_fd_helmholtz(self.handle, fpp(fd.lam)), lib=self.lib)
        self.default_method = L.FG_SOLVER_FDCG if self.has_fd else L.FG_SOLVER_CG
        if allocate:
            self.allocate_fields()

    # ------------------------------------------------------------------ shapes
    @property
    def spatial(self):
        return (self.ny, self.nx) if self.dims == 2 else (self.nz, self.ny, self.nx)

    def slab(self, face: int):
        s = list(self.spatial)
        s[len(s) - 1 - (face >> 1)] = 1
        return tuple(s)

    # ------------------------------------------------------------------ binding
    def bind(self, field: int, t: Optional[torch.Tensor]):
        if t is not None:
            assert t.is_cuda and t.dtype == self.dtype and t.is_contiguous(), f"fields must be contiguous {self.dtype} CUDA tensors"
            self._bound[field] = t  # keep alive
        else:
            self._bound.pop(field, None)
        L.check(self.lib.fg_bind(self.handle, field, _ptr(t)), lib=self.lib)

    def allocate_fields(self):
        z = lambda *shape: torch.zeros(shape, dtype=self.dtype, device=self.device)
        self.set_velocity(z(self.B, self.dims, *self.spatial))
        self.set_pressure(z(self.B, 1, *self.spatial))
        if self.n_scalars:
            self.set_scalar(z(self.B, self.n_scalars, *self.spatial))
        for f in range(2 * self.dims):
            if self.fixed[f]:
                self.set_boundary_velocity(f, z(self.B, self.dims, *self.slab(f)))
                if self.n_scalars:
                    self.set_boundary_scalar(f, z(self.B, self.n_scalars, *self.slab(f)))

    def set_velocity(self, t):
        self.velocity = t
        self.bind(L.FG_VELOCITY, t)

    def set_pressure(self, t):
        self.pressure = t
        self.bind(L.FG_PRESSURE, t)

    def set_scalar(self, t):
        self.scalar = t
        self.bind(L.FG_SCALAR, t)

    def set_velocity_source(self, t):
        self.velocity_source = t
        self.bind(L.FG_VELOCITY_SOURCE, t)

    def set_viscosity_field(self, t: Optional[torch.Tensor]):
        """Per-cell viscosity ``[B, *grid]`` of the velocity system (``Block.setViscosity``: the SGS hook of the TCF env), or None
        for the global one.  The tensor is bound, not copied: writing into it changes the next step's matrix."""
        self.viscosity_field = t
        self.bind(L.FG_VISCOSITY_FIELD, t)

    def sgs_smagorinsky(self, coefficient: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``SGSviscosityIncompressibleSmagorinsky`` of the bound velocity: ``[B, *grid]`` (``fg_sgs_smagorinsky``)."""
        if out is None:
            out = torch.empty((self.B,) + tuple(self.spatial), dtype=self.dtype, device=self.device)
        L.check(self.lib.fg_sgs_smagorinsky(self.handle, float(coefficient), _ptr(out), _stream(self.device)), lib=self.lib)
        return out

    def set_boundary_velocity(self, face, t):
        self.bvel[face] = t
        self.bind(L.FG_BOUND_VELOCITY + face, t)

    def set_boundary_scalar(self, face, t):
        self.bscal[face] = t
        self.bind(L.FG_BOUND_SCALAR + face, t)

    def set_viscosity(self, nu: float):
        self.viscosity = float(nu)
        L.check(self.lib.fg_set_viscosity(self.handle, float(nu)), lib=self.lib)

    def set_scalar_viscosity(self, ch: int, k: float):
        self.scalar_viscosities[int(ch)] = float(k)
        L.check(self.lib.fg_set_scalar_viscosity(self.handle, ch, float(k)), lib=self.lib)

    # ------------------------------------------------------------------ helpers
    def dt_tensor(self, dt) -> torch.Tensor:
        """Per-env time steps on the device; ``dt`` may be a float or a length-B sequence
        (``<= 0`` = env inactive for the call)."""
        if isinstance(dt, torch.Tensor) and dt.is_cuda:
            return dt
        self._dt_host.copy_(torch.as_tensor(np.broadcast_to(np.asarray(dt, dtype=self._np_real), (self.B,)).copy()))
        self._dt.copy_(self._dt_host, non_blocking=True)
        return self._dt

    def buffer(self, which: int, shape) -> torch.Tensor:
        """Copy of an internal solver vector (tests only)."""
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        L.check(self.lib.fg_get_buffer(self.handle, which, ctypes.byref(p), ctypes.byref(n)), lib=self.lib)
        out = torch.empty(int(n.value), dtype=self.dtype, device=self.device)
        L.check(self.lib.fg_read_buffer(self.handle, which, _ptr(out), _stream(self.device)), lib=self.lib)
        return out.view(*shape)

    def _infos(self, n):
        return (L.FgSolveInfo * n)()

    # ------------------------------------------------------------------ reductions
    def max_velocity(self) -> torch.Tensor:
        L.check(self.lib.fg_max_velocity(self.handle, _ptr(self._out_B), _stream(self.device)), lib=self.lib)
        return self._out_B.clone()

    def boundary_flux_balance(self) -> torch.Tensor:
        L.check(self.lib.fg_boundary_flux_balance(self.handle, _ptr(self._out_B), _stream(self.device)), lib=self.lib)
        return self._out_B.clone()

    def step_diagnostics(self):
        """(flux_balance[B], max_velocity[B]) as NumPy arrays with a single device->host sync."""
        buf = np.empty(2 * self.B, dtype=self._np_real)
        L.check(self.lib.fg_step_diagnostics(self.handle, buf.ctypes.data_as(ctypes.POINTER(self._c_real)),
                                             _stream(self.device)))
        return buf[: self.B], buf[self.B:]

    def update_advective_boundary(self, face: int, velm, dt):
        v = np.ascontiguousarray(velm, dtype=self._np_real).reshape(-1)
        L.check(self.lib.fg_update_advective_boundary(self.handle, face, v.ctypes.data_as(ctypes.POINTER(self._c_real)),
                                                      _ptr(self.dt_tensor(dt)), _stream(self.device)))

    def balance_boundary_fluxes(self, free_faces, atol: float, dt):
        mask = 0
        for f in free_faces:
            mask |= 1 << int(f)
        L.check(self.lib.fg_balance_boundary_fluxes(self.handle, mask, float(atol), _ptr(self.dt_tensor(dt)),
                                                    _stream(self.device)))

    # ------------------------------------------------------------------ PISO pieces
    def setup_advection(self, dt, for_scalar=False, channel=0):
        L.check(self.lib.fg_setup_advection(self.handle, _ptr(self.dt_tensor(dt)), int(for_scalar), channel,
                                            _stream(self.device)))

    def solve_advection(self, for_scalar=False, channel=0, tol=1e-5, max_iterations=5000) -> List[L.FgSolveInfo]:
        n = self.B * (1 if for_scalar else self.dims)
        info = self._infos(n)
        rc = self.lib.fg_solve_advection(self.handle, int(for_scalar), channel, tol, max_iterations, info,
                                         _stream(self.device))
        L.check(rc, allow=(L.FG_ERR_NOT_CONVERGED, L.FG_ERR_NOT_FINITE))
        return list(info)

    def copy_scalar_result_to_blocks(self, channel=0):
        L.check(self.lib.fg_copy_scalar_result_to_blocks(self.handle, channel, _stream(self.device)), lib=self.lib)

    def setup_pressure_matrix(self):
        L.check(self.lib.fg_setup_pressure_matrix(self.handle, _stream(self.device)), lib=self.lib)

    def setup_pressure_rhs(self, dt):
        L.check(self.lib.fg_setup_pressure_rhs(self.handle, _ptr(self.dt_tensor(dt)), _stream(self.device)), lib=self.lib)

    def solve_pressure(self, tol=1e-5, max_iterations=5000, method=None, use_previous=False):
        method = self.default_method if method is None else method
        info = self._infos(self.B)
        rc = self.lib.fg_solve_pressure(self.handle, method, tol, max_iterations, int(use_previous), info,
                                        _stream(self.device))
        L.check(rc, allow=(L.FG_ERR_NOT_CONVERGED, L.FG_ERR_NOT_FINITE))
        return list(info)

    def correct_velocity(self):
        L.check(self.lib.fg_correct_velocity(self.handle, _stream(self.device)), lib=self.lib)

    def copy_velocity_result_to_blocks(self):
        L.check(self.lib.fg_copy_velocity_result_to_blocks(self.handle, _stream(self.device)), lib=self.lib)

    def copy_velocity_result_from_blocks(self):
        L.check(self.lib.fg_copy_velocity_result_from_blocks(self.handle, _stream(self.device)), lib=self.lib)

    def piso_step(self, dt, corrector_steps=2, advect_scalar=True, advection_tol=1e-5, pressure_tol=1e-5,
                  max_iterations=5000, buoyancy_axis=-1, buoyancy_factor=0.0, method=None,
                  pressure_warm_start=False):
        method = self.default_method if method is None else method
        opt = self._StepOptions(corrector_steps, int(advect_scalar), method, max_iterations, advection_tol,
                              pressure_tol, buoyancy_axis, buoyancy_factor, int(pressure_warm_start))
        stats = (ctypes.c_int32 * 4)()
        rc = self.lib.fg_piso_step(self.handle, _ptr(self.dt_tensor(dt)), ctypes.byref(opt), stats,
                                   _stream(self.device))
        if rc == L.FG_ERR_NOT_FINITE:
            raise LinsolveError("linear solve produced a non-finite residual")
        L.check(rc, allow=(L.FG_ERR_NOT_CONVERGED,))
        return rc == L.FG_OK, list(stats)

    def _sim_options(self, time_step, cfl, adaptive=True, substeps=1, flux_balance_tol=1e-5, outflow_faces=(),
                     outflow_velm=(0.0, 0.0, 0.0), outflow_tol=1e-5, corrector_steps=2, advect_scalar=True,
                     advection_tol=1e-5, pressure_tol=1e-5, max_iterations=5000, buoyancy_axis=-1, buoyancy_factor=0.0,
                     method=None, pressure_warm_start=False, max_substeps=1000):
        method = self.default_method if method is None else method
        o = self._SimOptions()
        o.step = self._StepOptions(corrector_steps, int(advect_scalar), method, max_iterations, advection_tol, pressure_tol,
                                 buoyancy_axis, buoyancy_factor, int(pressure_warm_start))
        o.time_step, o.cfl, o.adaptive, o.substeps = float(time_step), float(cfl), int(adaptive), int(substeps)
        o.flux_balance_tol = float(flux_balance_tol)
        mask = 0
        for f in outflow_faces:
            mask |= 1 << int(f)
        o.outflow_mask = mask
        for i in range(3):
            o.outflow_velm[i] = float(outflow_velm[i]) if i < len(outflow_velm) else 0.0
        o.outflow_tol = float(outflow_tol)
        o.max_substeps = int(max_substeps)
        return o

    def single_step(self, time_step, cfl, adaptive=True, substeps=1, flux_balance_tol=1e-5, outflow_faces=(),
                    outflow_velm=(0.0, 0.0, 0.0), outflow_tol=1e-5, corrector_steps=2, advect_scalar=True,
                    advection_tol=1e-5, pressure_tol=1e-5, max_iterations=5000, buoyancy_axis=-1, buoyancy_factor=0.0,
                    method=None, pressure_warm_start=False, max_substeps=1000):
        """Native ``Simulation.single_step``; returns (all_converged, solver_stats[4], substeps)."""
        o = self._sim_options(time_step, cfl, adaptive, substeps, flux_balance_tol, outflow_faces, outflow_velm, outflow_tol, corrector_steps,
                              advect_scalar, advection_tol, pressure_tol, max_iterations, buoyancy_axis, buoyancy_factor, method,
                              pressure_warm_start, max_substeps)
        out = (ctypes.c_int32 * 6)()
        flux = (self._c_real * self.B)()
        rc = self.lib.fg_single_step(self.handle, ctypes.byref(o), out, flux, _stream(self.device))
        if rc == L.FG_ERR_FLUX_BALANCE:
            raise RuntimeError(
                "Domain boundary fluxes not balanced, cannot proceed with simulation step. "
                f"Flux balance: {list(flux)}, flux_balance_tol: {flux_balance_tol}")
        if rc == L.FG_ERR_NOT_FINITE:
            raise LinsolveError("linear solve produced a non-finite residual")
        L.check(rc)
        return bool(out[5]), [int(out[i]) for i in range(4)], int(out[4])

    def multi_step(self, n, time_step, cfl, boundary_schedule=None, **kw):
        """``n`` native ``single_step`` calls in one C call (``fg_multi_step``): the sim steps of an env step without a return to the
        interpreter in between.  ``boundary_schedule``: {face: tensor [n, B, d, ...]} -- slice k is bound to the face before sim step
        k (what ``set_boundary_velocity(face, t[k])`` in front of every step does).  Keyword arguments as :meth:`single_step`.
        Returns a list of (all_converged, solver_stats[4], substeps), one per step."""
        o = self._sim_options(time_step, cfl, **kw)
        sched = None
        if boundary_schedule:
            sched = (ctypes.c_void_p * (6 * n))()
            for face, t in boundary_schedule.items():
                # (t may be a strided view along the step axis -- the channel keeps [n, wall, ...] -- as long as every slice is dense)
                assert t.shape[0] == n and t[0].is_contiguous() and t.dtype == self.dtype and t.device == self.device
                base, step_bytes = t.data_ptr(), t.stride(0) * t.element_size()
                for k in range(n):
                    sched[6 * k + int(face)] = base + k * step_bytes
        out = (ctypes.c_int32 * (6 * n))()
        flux = (self._c_real * self.B)()
        done = ctypes.c_int32(0)
        rc = self.lib.fg_multi_step(self.handle, ctypes.byref(o), int(n), sched, out, flux, ctypes.byref(done), _stream(self.device))
        if boundary_schedule:
            for face, t in boundary_schedule.items():      # what the handle is bound to now
                self.bvel[int(face)] = t[max(min(done.value, n - 1), 0)]
        # the steps that completed (all of them, or up to a failure): kept on the handle so that a caller that catches the exception
        # below can still account for them (Simulation.multi_step advances its step / time counters from this)
        self.last_multi_results = [(bool(out[6 * k + 5]), [int(out[6 * k + i]) for i in range(4)], int(out[6 * k + 4])) for k in range(min(done.value, n))]
        if rc == L.FG_ERR_FLUX_BALANCE:
            raise RuntimeError(
                "Domain boundary fluxes not balanced, cannot proceed with simulation step. "
                f"Flux balance: {list(flux)}, flux_balance_tol: {kw.get('flux_balance_tol', 1e-5)}")
        if rc == L.FG_ERR_NOT_FINITE:
            raise LinsolveError("linear solve produced a non-finite residual")
        L.check(rc)
        return self.last_multi_results

    def set_pressure_refinement(self, max_corrections: int = 3, target_tol: float = 1e-10, inner_relative_tol: float = 1e-4):
        """Opt-in accuracy mode (``fg_set_pressure_refinement``): fp64 residual, fp32 corrections, at most ``max_corrections`` per
        pressure solve or until the fp64 residual's RMS is below ``target_tol``; 0 switches it off."""
        L.check(self.lib.fg_set_pressure_refinement(self.handle, int(max_corrections), self._c_real(target_tol), self._c_real(inner_relative_tol)), lib=self.lib)

    def solver_hints(self, values=None):
        """The 12 words the handle remembers between solves and that decide which iteration runs (``fg_solver_hints``): read as a
        list, or written from ``values``."""
        buf = (ctypes.c_int32 * 12)(*([0] * 12 if values is None else [int(v) for v in values]))
        L.check(self.lib.fg_solver_hints(self.handle, buf, 0 if values is None else 1), lib=self.lib)
        return list(buf)

    def reset_solver_state(self):
        L.check(self.lib.fg_reset_solver_state(self.handle, _stream(self.device)), lib=self.lib)

    def make_divergence_free(self, tol=1e-5, max_iterations=1000):
        info = self._infos(self.B)
        rc = self.lib.fg_make_divergence_free(self.handle, tol, max_iterations, info, _stream(self.device))
        L.check(rc, allow=(L.FG_ERR_NOT_CONVERGED,))
        return list(info)

    # ------------------------------------------------------------------ standalone Poisson
    def poisson_apply(self, rA, x, y=None):
        y = torch.empty_like(x) if y is None else y
        L.check(self.lib.fg_poisson_apply(self.handle, _ptr(rA), _ptr(x), _ptr(y), _stream(self.device)), lib=self.lib)
        return y

    def poisson_jacobi(self, rA, b, x, sweeps, omega=1.0):
        L.check(self.lib.fg_poisson_jacobi(self.handle, _ptr(rA), _ptr(b), _ptr(x), sweeps, omega, _stream(self.device)), lib=self.lib)
        return x

    def poisson_rbgs(self, rA, b, x, sweeps, omega=1.0):
        L.check(self.lib.fg_poisson_rbgs(self.handle, _ptr(rA), _ptr(b), _ptr(x), sweeps, omega, _stream(self.device)), lib=self.lib)
        return x

    def poisson_cg(self, rA, b, x, tol=1e-5, max_iterations=5000, use_x0=False):
        info = self._infos(self.B)
        rc = self.lib.fg_poisson_cg(self.handle, _ptr(rA), _ptr(b), _ptr(x), tol, max_iterations, int(use_x0), info,
                                    _stream(self.device))
        L.check(rc, allow=(L.FG_ERR_NOT_CONVERGED, L.FG_ERR_NOT_FINITE))
        return list(info)


    def solver_counters(self, reset: bool = False) -> dict:
        """Iterations of the linear solves since the last reset: per kind (scalar, velocity, pressure corrector 0 / 1) the
        mean and max per system (env x component) and the number of PISO steps (``fg_solver_counters``)."""
        unconv = (ctypes.c_int64 * 4)()
        L.check(self.lib.fg_solver_unconverged(self.handle, unconv))      # (read before the counters are cleared, lib=self.lib)
        out = (ctypes.c_int64 * 13)()
        L.check(self.lib.fg_solver_counters(self.handle, out, int(reset)), lib=self.lib)
        names = ("scalar", "velocity", "pressure0", "pressure1")
        res = {n: {"mean": (out[k] / out[4 + k]) if out[4 + k] else None, "max": int(out[8 + k]), "systems": int(out[4 + k]),
                    "unconverged": int(unconv[k])}
               for k, n in enumerate(names)}
        res["piso_steps"] = int(out[12])
        return res

    def config_dump(self) -> dict:
        """The switches this handle runs under (``fg_config_dump``: the FG_* variables read at create time + the solver policies set
        through the API) plus the process environment's FG_* / FLUIDGYM_AMD_* variables."""
        import json

        buf = ctypes.create_string_buffer(4096)
        L.check(self.lib.fg_config_dump(self.handle, buf, 4096), lib=self.lib)
        out = json.loads(buf.value.decode())
        out["env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("FG_", "FLUIDGYM_"))}
        return out

    def set_return_best(self, on: bool = True):
        """``pressure_return_best_result`` of the reference's Simulation: keep / hand back the best CG iterate."""
        L.check(self.lib.fg_set_return_best(self.handle, int(on)), lib=self.lib)

    def set_cg_reset_steps(self, steps: int = 100):
        """``residualResetSteps`` of the pressure CG (``cg_solver_kernel.cu:281-302``); 0 = never."""
        L.check(self.lib.fg_set_cg_reset_steps(self.handle, int(steps)), lib=self.lib)

    def set_advection_start(self, from_result: bool = True):
        """Start vector of the velocity solve: ``velocityResult`` (the reference's orthogonal branch) or zero (its non-orthogonal
        branch, first pass) -- ``fg_set_advection_start``."""
        L.check(self.lib.fg_set_advection_start(self.handle, int(from_result)), lib=self.lib)

    def set_wall_stress_forcing(self, axis: int, coef_lo: float = 0.0, coef_hi: float = 0.0) -> None:
        """Native form of the turbulent-channel env's PRE hook (``fg_set_wall_stress_forcing``): before every PISO step the uniform
        body force ``1/2 (coef_lo <u_axis>_{-y layer} + coef_hi <u_axis>_{+y layer})`` per env along ``axis``; ``axis < 0`` = off."""
        L.check(self.lib.fg_set_wall_stress_forcing(self.handle, int(axis), float(coef_lo), float(coef_hi)), lib=self.lib)

    def set_advection_preconditioner(self, mode: int = 0):
        """Preconditioner policy of the advection-diffusion BiCGStab (``fg_set_advection_preconditioner``): 0 plain (the
        reference's first rung), 1 every solve right-preconditioned by the y-line solve (its ``preconditionBiCG``), 2 only
        to repeat a failed solve (its ``BiCG_precondition_fallback``), 3 every solve right-preconditioned by the separable
        Helmholtz operator (fast diagonalisation; needs ``has_helmholtz``), 4 / 5 like 1 / 2 with the reference's own preconditioner,
        ILU(0) of the matrix (``csrc/fg_ilu0.hip``; hyperplane sweeps: correct and slow, ~1 ms per application)."""
        if self.f64:
            return    # the preconditioners are fp32 kernel families: the fp64 build keeps the plain recurrence (mode 0)
        L.check(self.lib.fg_set_advection_preconditioner(self.handle, int(mode)), lib=self.lib)

    def apply_advection_preconditioner(self, mode: int, r: torch.Tensor) -> torch.Tensor:
        """``z = M^-1 r`` for ``r [B, nc, *grid]`` with the preconditioner of ``mode`` (1 y-line, 4 ILU(0)) built from the matrix
        ``setup_advection`` assembled last (``fg_debug_apply_preconditioner``; tests)."""
        r = r.to(self.device, torch.float32).contiguous()
        z = torch.empty_like(r)
        L.check(self.lib.fg_debug_apply_preconditioner(self.handle, int(mode), int(r.shape[1]), ctypes.c_void_p(r.data_ptr()),
                                                       ctypes.c_void_p(z.data_ptr()), _stream(self.device)), lib=self.lib)
        return z

    def advection_retries(self, reset: bool = False) -> int:
        out = ctypes.c_int64()
        L.check(self.lib.fg_advection_retries(self.handle, ctypes.byref(out), int(reset)), lib=self.lib)
        return int(out.value)

    def set_double_fallback(self, on: bool = True) -> None:
        """``solver_double_fallback`` of the reference's linear-solve ladder (PISOtorch_diff.py:418-445): see ``fg_set_double_fallback``."""
        L.check(self.lib.fg_set_double_fallback(self.handle, int(bool(on))), lib=self.lib)

    def set_advection_jacobi(self, on: bool = True) -> None:
        """Velocity systems of uniform 2-D grids with walls in y: point-Jacobi sweeps, several per pass with the tile on chip
        (``csrc/fg_jacobi.hip``), before the reference's BiCGStab -- see ``fg_set_advection_jacobi`` and the policy ``advection_jacobi``."""
        L.check(self.lib.fg_set_advection_jacobi(self.handle, int(bool(on))), lib=self.lib)

    def advection_jacobi_counts(self) -> Dict[str, int]:
        out = (ctypes.c_int64 * 2)()
        L.check(self.lib.fg_advection_jacobi_counts(self.handle, out), lib=self.lib)
        return {"settled_by_sweeps": int(out[0]), "handed_to_bicgstab": int(out[1])}

    def ladder(self, force_mask: int = -1) -> Dict[str, int]:
        """How often each rung of the retry ladder ran (``fg_ladder``); ``force_mask`` >= 0 (tests) makes first attempts count as
        failed: 1 advection, 2 pressure, 4 also the advection fp64 rung."""
        out = (ctypes.c_int64 * 4)()
        L.check(self.lib.fg_ladder(self.handle, out, int(force_mask)), lib=self.lib)
        return {"advection_fp64": int(out[0]), "advection_preconditioned": int(out[1]), "pressure_fp64": int(out[2])}

    def advection_solver_form(self, nc: Optional[int] = None) -> str:
        """Kernels of the next un-preconditioned advection-diffusion solve: 'five' | 'two-brick' | 'two-zmarch'."""
        out = ctypes.c_int32()
        L.check(self.lib.fg_advection_solver_form(self.handle, int(self.dims if nc is None else nc), ctypes.byref(out)), lib=self.lib)
        return ("five", "two-brick", "two-zmarch")[out.value]

    def profile_enable(self, on: bool = True):
        L.check(self.lib.fg_profile_enable(self.handle, int(on)), lib=self.lib)

    def profile_read(self):
        """{kernel name: dict(ms, samples, bytes, flops, full_ms, full_bytes, full_samples, launches)} of the sampled
        solver-kernel launches since profile_enable (see fg_profile_read in include/fluidgym_hip.h)."""
        out = {}
        for k in range(self.lib.fg_profile_kinds()):
            ms, by, fl, fms, fby, ams = (ctypes.c_double() for _ in range(6))
            n, fn, nl, an = (ctypes.c_int64() for _ in range(4))
            L.check(self.lib.fg_profile_read(self.handle, k, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by),
                                             ctypes.byref(fl), ctypes.byref(fms), ctypes.byref(fby), ctypes.byref(fn),
                                             ctypes.byref(nl), ctypes.byref(ams), ctypes.byref(an)))
            out[self.lib.fg_profile_kind_name(k).decode()] = dict(
                ms=ms.value, samples=n.value, bytes=by.value, flops=fl.value, full_ms=fms.value,
                full_bytes=fby.value, full_samples=fn.value, launches=nl.value, all_ms=ams.value,
                all_samples=an.value)
        return out

    def poisson_fdcg(self, rA, b, x, tol=1e-5, max_iterations=500, use_x0=False):
        info = self._infos(self.B)
        rc = self.lib.fg_poisson_fdcg(self.handle, _ptr(rA), _ptr(b), _ptr(x), tol, max_iterations, int(use_x0), info,
                                      _stream(self.device))
        L.check(rc, allow=(L.FG_ERR_NOT_CONVERGED, L.FG_ERR_NOT_FINITE))
        return list(info)

    def close(self):
        if getattr(self, "handle", None):
            torch.cuda.synchronize(self.device)
            self.lib.fg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def coords_to_transforms(coords: torch.Tensor) -> torch.Tensor:
    """``PISOtorch.CoordsToTransforms`` (grid_gen.cu:356-390): coords ``[1,d,(Z+1,)Y+1,X+1]`` ->
    transforms ``[1,(Z,)Y,X,2d^2+1]``."""
    lib = L.load()
    assert coords.is_cuda and coords.dtype == torch.float32
    coords = coords.contiguous()
    d = coords.shape[1]
    sp = [s - 1 for s in coords.shape[2:]]
    nx, ny = sp[-1], sp[-2]
    nz = sp[0] if d == 3 else 1
    out = torch.empty([1] + sp + [2 * d * d + 1], dtype=torch.float32, device=coords.device)
    L.check(lib.fg_coords_to_transforms(_ptr(coords), _ptr(out), d, nx, ny, nz, _stream(coords.device)))
    return out
