"""PISO driver: the batched counterpart of FluidGym's ``Simulation``.

Mirrors, for the orthogonal single-block case,

* ``fluidgym/simulation/simulation.py:124-280``  (``Simulation.__init__``, ``single_step``: flux-balance
  guard + substep policy),
* ``pict/PISOtorch_simulation.py:2004-2064``      (``_PISO_adaptive_step``),
* ``pict/PISOtorch_simulation.py:1431-2002``      (``_PISO_split_step``, hooks PRE /
  PRE_VELOCITY_SETUP / POST_VELOCITY_SETUP / POST_PREDICTION / POST_PRESSURE_SETUP /
  POST_PRESSURE_RESULT / POST_VELOCITY_CORRECTION / POST),
* ``pict/PISOtorch_simulation.py:1320-1429``      (``make_divergence_free``),
* ``pict/PISOtorch_simulation.py:188-393``        (advective outflow + boundary-flux balancing).

Everything numerical runs in ``libfluidgym_hip.so``.  What stays in Python is control flow and the
per-substep CFL decision, which -- as in the reference -- needs one device->host read of the
per-env maximum velocity.  Because envs are batched, every env gets its own ``dt``; envs that have
already covered the requested time span are masked out with ``dt = 0``.
"""
from __future__ import annotations

import os

import logging
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from .. import _lib as L
from ..native import LinsolveError
from .domain import Domain, FixedBoundary
from .policy import get_solver_policy

_LOG = logging.getLogger("PISOsim")

HOOK_NAMES = (
    "PRE", "POST_SCALAR_SETUP", "PRE_VELOCITY_SETUP", "POST_VELOCITY_SETUP", "POST_PREDICTION",
    "POST_PRESSURE_SETUP", "POST_PRESSURE_RESULT", "POST_PRESSURE_NON_ORTHO", "POST_VELOCITY_CORRECTION", "POST",
)
# hooks that can run around the fused native step (everything between them has no hook point)
_FUSED_OK = {"PRE", "POST"}


def get_solver_tolerance(tol: Optional[float], dtype=torch.float32) -> float:
    """``_get_solver_tolerance`` (pict/PISOtorch_diff.py:247-253)."""
    if tol is None:
        return 1e-8 if dtype == torch.float64 else 1e-5
    return float(tol)


class Simulation:
    def __init__(
        self,
        domain: Domain,
        dt: float,
        verbose: bool = False,
        substeps: Union[int, str] = 1,
        corrector_steps: int = 2,
        density_viscosity=None,
        adaptive_CFL: float = 0.8,
        prep_fn: Optional[Dict[str, List[Callable]]] = None,
        advection_use_BiCG: bool = True,
        pressure_use_BiCG: bool = False,
        scipy_solve_advection: bool = False,
        scipy_solve_pressure: bool = False,
        preconditionBiCG: bool = False,
        BiCG_precondition_fallback: bool = True,
        advection_tol: Optional[float] = None,
        pressure_tol: Optional[float] = None,
        flux_balance_tol: float = 1e-5,
        convergence_tol: Optional[float] = None,
        solver_double_fallback: bool = False,
        advect_non_ortho_steps: int = 1,
        pressure_non_ortho_steps: int = 1,
        normalize_pressure_result: bool = True,
        pressure_return_best_result: bool = False,
        advect_passive_scalar: bool = True,
        pressure_time_step_normalized: bool = False,
        velocity_corrector: str = "FD",
        non_orthogonal: bool = True,
        differentiable: bool = False,
        output_resampling_shape=None,
        output_resampling_fill_max_steps: int = 0,
        buoyancy: Optional[tuple] = None,
        wall_forcing: Optional[tuple] = None,
        pressure_warm_start: Optional[bool] = None,
        advection_warm_start: Optional[bool] = None,
        outflow: Optional[tuple] = None,
        **_ignored: Any,
    ):
        if not isinstance(domain, Domain):
            raise TypeError("domain must be a fluidgym_amd Domain object.")
        if not domain.IsInitialized():
            raise RuntimeError("domain must be initilized. Call domain.PrepareSolve() before assignment.")
        if differentiable:
            raise NotImplementedError("differentiable mode (autograd kernels) is out of scope (SURVEY 2.2)")
        if scipy_solve_advection or scipy_solve_pressure or pressure_use_BiCG or not advection_use_BiCG:
            raise NotImplementedError("only the env configuration (BiCGStab advection, CG pressure) is built")
        if velocity_corrector != "FD" or pressure_time_step_normalized or not normalize_pressure_result:
            raise NotImplementedError("only velocity_corrector='FD', un-normalised time step, mean-free pressure")
        if isinstance(substeps, str):
            if substeps.upper() != "ADAPTIVE":
                raise ValueError("Invalid substeps")
            substeps = -1
        self.domain = domain
        self.time_step = float(dt)
        self.substeps = int(substeps)
        self.corrector_steps = int(corrector_steps)
        self.adaptive_CFL = float(adaptive_CFL)
        self.prep_fn: Dict[str, List[Callable]] = dict(prep_fn or {})
        for k in self.prep_fn:
            if k not in HOOK_NAMES:
                raise KeyError(f"unknown prep_fn hook {k!r}")
        self.advection_tol = advection_tol
        self.pressure_tol = pressure_tol
        self.flux_balance_tol = float(flux_balance_tol)
        self.linear_solve_max_iterations = 5000  # PISOtorch_simulation.py:564
        self.pressure_return_best_result = pressure_return_best_result
        # returnBestResult of the pressure solve (PISOtorch_simulation.py:1913): the native CG keeps / restores it
        domain.solver.set_return_best(bool(pressure_return_best_result))
        self.advect_passive_scalar = advect_passive_scalar
        self.non_orthogonal = non_orthogonal  # identical results on orthogonal grids
        self.buoyancy = buoyancy  # (axis, factor): native form of the RBC PRE_VELOCITY_SETUP hook
        # (axis, coef_lo, coef_hi): native form of the turbulent-channel env's PRE hook -- a uniform body force along `axis` equal to
        # the mean of the two wall shear stresses, recomputed before every PISO step (fg_set_wall_stress_forcing)
        self.wall_forcing = wall_forcing
        if wall_forcing is not None:
            domain.solver.set_wall_stress_forcing(*wall_forcing)
        # the reference starts every pressure solve of this path from zero (orthogonal branch x=None,
        # PISOtorch_simulation.py:1804-1807; non-orthogonal branch x=None at pstep 0, :1877-1881, and this path runs
        # pressure_non_ortho_steps == 1): that is the default.  Starting from the previous pressure is the opt-in
        # performance mode of simulation/policy.py
        self.pressure_warm_start = bool(get_solver_policy()["pressure_warm_start"] if pressure_warm_start is None
                                        else pressure_warm_start)
        # start vector of the velocity solve: the reference's orthogonal branch starts from velocityResult
        # (PISOtorch_simulation.py:1689-1693), its non-orthogonal branch -- which the TCF env runs on this kind of grid
        # (tcf_env.py:497) -- from zero on its first (here only) non-orthogonal pass (:1735-1742).  The policy switch
        # advection_warm_start (policy.py) starts from velocityResult in both
        self.advection_warm_start = bool(get_solver_policy()["advection_warm_start"] if advection_warm_start is None
                                         else advection_warm_start)
        domain.solver.set_advection_start((not non_orthogonal) or self.advection_warm_start)
        # The reference's retry chain of the advection-diffusion solves (_linear_solve, PISOtorch_diff.py:449-476) on this path:
        # preconditionBiCG preconditions every solve, BiCG_precondition_fallback repeats a failed one with the preconditioner
        # (cuSPARSE ILU(0) there, the y-line solve of csrc/fg_linepre.hip here).  On top of that the policy switch
        # advection_line_preconditioner (policy.py, default OFF) preconditions every solve on grids refined towards a y wall, where
        # the plain recurrence needs 20-35 iterations (RBC 512 x 128): same system, same tolerance, another Krylov trajectory.
        # solver_double_fallback (PISOtorch_diff.py:418-445): a solve that failed in fp32 is repeated in fp64 on the same matrix and
        # right-hand side from a cleared result, BEFORE the preconditioned rung (csrc/fg_rung64.h; round 4 -- until then the kwarg
        # was accepted and ignored on this path).
        self.preconditionBiCG = bool(preconditionBiCG)
        self.BiCG_precondition_fallback = bool(BiCG_precondition_fallback)
        self.solver_double_fallback = bool(solver_double_fallback)
        solver = domain.solver
        if hasattr(solver, "set_double_fallback"):
            solver.set_double_fallback(self.solver_double_fallback)
        if hasattr(solver, "set_advection_preconditioner"):
            hy = np.asarray(solver.widths[1], dtype=np.float64)
            refined = bool(get_solver_policy()["advection_line_preconditioner"]) and float(hy.max() / hy.min()) >= 3.0
            self.advection_preconditioner = 1 if (self.preconditionBiCG or refined) else (2 if self.BiCG_precondition_fallback else 0)
            # policy advection_rung_preconditioner = "ilu0": the rungs use the reference's own preconditioner (ILU(0), modes 4 / 5)
            if get_solver_policy()["advection_rung_preconditioner"] == "ilu0" and min(len(w) for w in solver.widths[:solver.dims]) >= 4 and not refined:
                self.advection_preconditioner = {1: 4, 2: 5}.get(self.advection_preconditioner, self.advection_preconditioner)
            # policy advection_fd_preconditioner (default on): where the grid allows it -- periodic, uniform x (and z), walls in y:
            # the RBC and TCF families -- every advection-diffusion solve is right-preconditioned by the exact inverse of its
            # diffusion part (separable Helmholtz operator, fast diagonalisation).  2-D only by default: in 3-D the four basis
            # changes per application cost what the 2-3 saved iterations of a 4-iteration solve give back
            fd_pol = get_solver_policy()["advection_fd_preconditioner"]
            if getattr(solver, "has_helmholtz", False) and not self.preconditionBiCG and (fd_pol == "always" or (fd_pol == "auto" and solver.dims == 2)):
                self.advection_preconditioner = 3
            solver.set_advection_preconditioner(self.advection_preconditioner)
        # policy advection_jacobi (default on): un-preconditioned velocity solves of uniform 2-D grids go to the on-chip Jacobi
        # sweeps first (csrc/fg_jacobi.hip); the library checks per solve whether the grid qualifies and whether they contract
        self.advection_jacobi = bool(get_solver_policy()["advection_jacobi"])
        if hasattr(solver, "set_advection_jacobi"):
            solver.set_advection_jacobi(self.advection_jacobi)
        # policy pressure_refinement (default 0 = off): the opt-in accuracy mode of the single-block pressure solves -- fp64 residual,
        # fp32 corrections (fg_set_pressure_refinement); fp32 library only
        self.pressure_refinement = int(get_solver_policy()["pressure_refinement"])
        if self.pressure_refinement > 0 and hasattr(solver, "set_pressure_refinement") and not getattr(solver, "f64", False):
            solver.set_pressure_refinement(self.pressure_refinement, target_tol=1e-3 * get_solver_tolerance(pressure_tol), inner_relative_tol=1e-4)
        # (bounds, velm, tol): advective-outflow PRE hook of the cylinder/airfoil envs (PISOtorch_simulation.py:
        # 228-393, wired in cylinder_env_base.py:280-300), kept as data so the native driver can run it
        self.outflow = outflow
        if outflow is not None:
            bounds, velm, otol = outflow

            def _outflow_hook(domain, time_step, **kw):
                update_advective_boundaries(domain, list(bounds), velm, time_step, tol=otol)

            self.prep_fn = {**self.prep_fn, "PRE": [_outflow_hook] + list(self.prep_fn.get("PRE", []))}
            self._native_pre = _outflow_hook
        else:
            self._native_pre = None
        self.output_resampling_shape = output_resampling_shape
        self.output_resampling_fill_max_steps = output_resampling_fill_max_steps
        self.total_step = 0
        self.total_time = np.zeros(domain.batch, dtype=np.float64)
        self.last_stats: List[int] = []
        self.substep_count = 0
        self._max_vel_hint = None
        if not verbose:
            _LOG.setLevel("ERROR")

    # ------------------------------------------------------------------------------------------
    @property
    def _solver(self):
        return self.domain.solver

    def _run_prep_fn(self, name: str, **kw):
        for fn in self.prep_fn.get(name, ()):
            fn(domain=self.domain, **kw)

    def _fused_ok(self) -> bool:
        return all((k in _FUSED_OK) or not v for k, v in self.prep_fn.items())

    # ------------------------------------------------------------------------------------------
    def single_step(self, static: bool = False) -> bool:
        """``Simulation.single_step`` (simulation.py:206-280)."""
        if static:
            raise NotImplementedError("advect_static is not on the env path")
        if self._native_ok():
            return self._single_step_native()
        # one device->host transfer for both per-step scalars (flux balance guard, simulation.py:223-231, and the
        # CFL velocity of the first substep, PISOtorch_simulation.py:2013-2014)
        balance, max_vel = self._solver.step_diagnostics()
        worst = float(np.abs(balance).max())
        if not (worst <= self.flux_balance_tol):
            raise RuntimeError(
                f"Domain boundary fluxes not balanced, cannot proceed with simulation step. "
                f"Flux balance: {balance.tolist()}, flux_balance_tol: {self.flux_balance_tol}"
            )
        self._max_vel_hint = max_vel
        try:
            if self.substeps > 0:
                ok = self._PISO_split_step(self.substeps, None)
            elif self.substeps == -1:
                ok = self._PISO_adaptive_step()
            else:
                raise ValueError("Invalid substeps")
        except LinsolveError:
            _LOG.exception("Simulation failed in step (total step %d):", self.total_step)
            return False
        return ok

    def _native_ok(self) -> bool:
        """True when every registered hook has a native equivalent, i.e. the whole ``single_step`` (guard,
        adaptive substeps, outflow hook, PISO step) can run inside ``fg_single_step`` without the interpreter."""
        for k, fns in self.prep_fn.items():
            for fn in fns:
                if fn is not self._native_pre:
                    return False
        return self.substeps == -1 or self.substeps > 0

    def _single_step_native(self) -> bool:
        s = self._solver
        bax, bfac = self.buoyancy if self.buoyancy is not None else (-1, 0.0)
        faces, velm, otol = (), (0.0, 0.0, 0.0), 1e-5
        if self.outflow is not None:
            bounds, velm_t, otol = self.outflow
            faces = [b.face for b in bounds]
            velm = np.asarray(velm_t.detach().cpu() if isinstance(velm_t, torch.Tensor) else velm_t, dtype=np.float64).reshape(-1)
        try:
            ok, stats, n_sub = s.single_step(
                self.time_step, self.adaptive_CFL, adaptive=(self.substeps == -1), substeps=max(self.substeps, 1),
                flux_balance_tol=self.flux_balance_tol, outflow_faces=faces, outflow_velm=velm,
                outflow_tol=get_solver_tolerance(otol), corrector_steps=self.corrector_steps,
                advect_scalar=self.advect_passive_scalar and self.domain.hasPassiveScalar(),
                advection_tol=get_solver_tolerance(self.advection_tol), pressure_tol=get_solver_tolerance(self.pressure_tol),
                max_iterations=self.linear_solve_max_iterations, buoyancy_axis=bax, buoyancy_factor=bfac,
                pressure_warm_start=self.pressure_warm_start)
        except LinsolveError:
            _LOG.exception("Simulation failed in step (total step %d):", self.total_step)
            return False
        self.last_stats, self.substep_count = stats, n_sub
        self.total_step += n_sub
        self.total_time += self.time_step if self.substeps == -1 else self.time_step * max(self.substeps, 1)
        if not ok and not self.pressure_return_best_result:
            _LOG.error("linear solve did not converge (iterations %s)", stats)
            return False
        return True

    def multi_step(self, n: int, boundary_schedule=None) -> bool:
        """``n`` calls of :meth:`single_step` -- in ONE native call when every hook has a native equivalent (``fg_multi_step``): the
        sim steps of an env step (``FluidEnv.step``, fluid_env.py:788-800) then run without the interpreter in between.
        ``boundary_schedule``: {face: tensor [n, ...]}, slice k bound to the face's boundary velocity before sim step k (the
        smoothed jet control changes every sim step).  Same arithmetic, same order, same results as the loop."""
        s = self._solver
        if not (self._native_ok() and hasattr(s, "multi_step")) or os.environ.get("FLUIDGYM_AMD_MULTI_STEP", "1") == "0":
            for k in range(n):
                for face, t in (boundary_schedule or {}).items():
                    s.set_boundary_velocity(face, t[k])
                if not self.single_step():
                    return False
            return True
        bax, bfac = self.buoyancy if self.buoyancy is not None else (-1, 0.0)
        faces, velm, otol = (), (0.0, 0.0, 0.0), 1e-5
        if self.outflow is not None:
            bounds, velm_t, otol = self.outflow
            faces = [b.face for b in bounds]
            velm = np.asarray(velm_t.detach().cpu() if isinstance(velm_t, torch.Tensor) else velm_t, dtype=np.float64).reshape(-1)
        def account(results):
            good_ = True
            for ok, stats, n_sub in results:
                self.last_stats, self.substep_count = stats, n_sub
                self.total_step += n_sub
                self.total_time += self.time_step if self.substeps == -1 else self.time_step * max(self.substeps, 1)
                if not ok and not self.pressure_return_best_result:
                    _LOG.error("linear solve did not converge (iterations %s)", stats)
                    good_ = False
            return good_

        s.last_multi_results = []
        try:
            res = s.multi_step(
                n, self.time_step, self.adaptive_CFL, boundary_schedule, adaptive=(self.substeps == -1), substeps=max(self.substeps, 1),
                flux_balance_tol=self.flux_balance_tol, outflow_faces=faces, outflow_velm=velm,
                outflow_tol=get_solver_tolerance(otol), corrector_steps=self.corrector_steps,
                advect_scalar=self.advect_passive_scalar and self.domain.hasPassiveScalar(),
                advection_tol=get_solver_tolerance(self.advection_tol), pressure_tol=get_solver_tolerance(self.pressure_tol),
                max_iterations=self.linear_solve_max_iterations, buoyancy_axis=bax, buoyancy_factor=bfac,
                pressure_warm_start=self.pressure_warm_start)
        except LinsolveError:
            account(getattr(s, "last_multi_results", []))      # the steps that did complete: fields and counters stay in step
            _LOG.exception("Simulation failed in step (total step %d):", self.total_step)
            return False
        except RuntimeError:                                   # the flux-balance guard: the same accounting, then the caller's problem
            account(getattr(s, "last_multi_results", []))
            raise
        return account(res)

    def _PISO_adaptive_step(self, CFL_cond: Optional[float] = None, max_substeps: int = 1000) -> bool:
        """Per-env version of ``_PISO_adaptive_step`` (PISOtorch_simulation.py:2004-2064): before every
        substep read ``max_vel`` and choose ``ts = t_rem / ceil(t_rem / (CFL / max_vel))``."""
        cfl = self.adaptive_CFL if CFL_cond is None else CFL_cond
        B = self.domain.batch
        t_rem = np.full(B, self.time_step, dtype=np.float64)
        substep = 0
        while True:
            active = (t_rem > 0) & ~np.isclose(t_rem, 0)
            if not active.any():
                break
            if self._max_vel_hint is not None:
                max_vel, self._max_vel_hint = self._max_vel_hint.astype(np.float64), None
            else:
                max_vel = self._solver.step_diagnostics()[1].astype(np.float64)
            ts = np.zeros(B, dtype=np.float64)
            for b in np.nonzero(active)[0]:
                mv = max_vel[b]
                max_ts = t_rem[b] if np.isclose(mv, 0) else cfl / mv
                if max_ts >= t_rem[b]:
                    ts[b] = t_rem[b]
                else:
                    ts[b] = t_rem[b] / int(np.ceil(t_rem[b] / max_ts))
            t_rem = np.where(active, t_rem - ts, t_rem)
            ok = self._PISO_split_step(1, ts.astype(np.float32))
            substep += 1
            if not ok:
                return False
            if substep > max_substeps:
                _LOG.warning("adaptive step (CFL=%.02f) needs more than %d substeps", cfl, max_substeps)
        self.substep_count = substep
        return True

    def _PISO_split_step(self, iterations: int, time_step) -> bool:
        """``iterations`` PISO steps of ``time_step`` (scalar, per-env array, or None = ``self.time_step``)."""
        s = self._solver
        if time_step is None:
            time_step = self.time_step
        dt = s.dt_tensor(time_step)
        dt_host = np.broadcast_to(np.asarray(time_step, dtype=np.float64), (self.domain.batch,))
        adv_tol = get_solver_tolerance(self.advection_tol)
        p_tol = get_solver_tolerance(self.pressure_tol)
        hook_kw = dict(time_step=dt, total_step=self.total_step, total_time=self.total_time)
        for step in range(int(iterations)):
            self._run_prep_fn("PRE", local_step=step, **hook_kw)
            if self._fused_ok():
                bax, bfac = self.buoyancy if self.buoyancy is not None else (-1, 0.0)
                ok, stats = s.piso_step(dt, corrector_steps=self.corrector_steps,
                                        advect_scalar=self.advect_passive_scalar and self.domain.hasPassiveScalar(),
                                        advection_tol=adv_tol, pressure_tol=p_tol,
                                        max_iterations=self.linear_solve_max_iterations,
                                        buoyancy_axis=bax, buoyancy_factor=bfac,
                                        pressure_warm_start=self.pressure_warm_start)
                self.last_stats = stats
                if not ok and not self.pressure_return_best_result:
                    raise LinsolveError(f"linear solve did not converge (iterations {stats})")
            else:
                self._split_step_hooked(dt, step, adv_tol, p_tol, hook_kw)
            self._run_prep_fn("POST", local_step=step, **hook_kw)
            self.total_step += 1
            self.total_time += dt_host
        return True

    def _split_step_hooked(self, dt, step, adv_tol, p_tol, hook_kw):
        """Same sequence as the fused driver, one C-ABI call per reference backend call, with every
        hook point of the reference in between (PISOtorch_simulation.py:1453-2000)."""
        s = self._solver
        dom = self.domain
        maxit = self.linear_solve_max_iterations
        if self.advect_passive_scalar and dom.hasPassiveScalar():
            for ch in range(dom.n_scalars):
                s.setup_advection(dt, for_scalar=True, channel=ch)
                self._run_prep_fn("POST_SCALAR_SETUP", no_step=0, local_step=step, **hook_kw)
                self._check(s.solve_advection(for_scalar=True, channel=ch, tol=adv_tol, max_iterations=maxit), False)
                s.copy_scalar_result_to_blocks(ch)
        if self.buoyancy is not None:
            ax, fac = self.buoyancy
            src = torch.zeros_like(s.velocity)
            src[:, ax] = s.scalar[:, 0] * fac
            dom.getBlock(0).setVelocitySource(src)
        self._run_prep_fn("PRE_VELOCITY_SETUP", local_step=step, **hook_kw)
        s.setup_advection(dt)
        self._run_prep_fn("POST_VELOCITY_SETUP", no_step=0, local_step=step, **hook_kw)
        self._check(s.solve_advection(tol=adv_tol, max_iterations=maxit), False)
        self._run_prep_fn("POST_PREDICTION", local_step=step, **hook_kw)
        s.setup_pressure_matrix()
        for _ in range(self.corrector_steps):
            s.setup_pressure_rhs(dt)
            self._run_prep_fn("POST_PRESSURE_SETUP", local_step=step, **hook_kw)
            self._check(s.solve_pressure(tol=p_tol, max_iterations=maxit, use_previous=self.pressure_warm_start),
                        self.pressure_return_best_result)
            self._run_prep_fn("POST_PRESSURE_RESULT", local_step=step, **hook_kw)
            self._run_prep_fn("POST_PRESSURE_NON_ORTHO", local_step=step, **hook_kw)
            s.correct_velocity()
            self._run_prep_fn("POST_VELOCITY_CORRECTION", local_step=step, **hook_kw)
        s.copy_velocity_result_to_blocks()

    @staticmethod
    def _check(infos, allow_unconverged: bool):
        """``_check_solver_return_infos`` policy (pict/PISOtorch_diff.py:266-351)."""
        if any(not i.is_finite for i in infos):
            raise LinsolveError(f"Linear solve reported non-finite residual: {infos}")
        if any(not i.converged for i in infos):
            if allow_unconverged:
                _LOG.warning("linear solve did not converge, using last result: %s", infos)
            else:
                raise LinsolveError(f"Linear solve did not converge: {infos}")

    def make_divergence_free(self, tol: Optional[float] = None, max_iterations: int = 1000):
        """``make_divergence_free`` (PISOtorch_simulation.py:1320-1429)."""
        infos = self._solver.make_divergence_free(get_solver_tolerance(tol if tol is not None else self.pressure_tol),
                                                  max_iterations)
        # the reference ends the call with end_step(time_step = 1): the counters advance (PISOtorch_simulation.py:1334, 1427)
        self.total_step += 1
        self.total_time += 1.0
        return all(i.converged for i in infos)


# ----------------------------------------------------------------------------------------------
# advective outflow boundary + flux balancing (PISOtorch_simulation.py:188-393), batched torch ops
# on boundary slabs only (a few KB); no field data leaves the solver.
# ----------------------------------------------------------------------------------------------
def _cell_slab(t: torch.Tensor, face: int) -> torch.Tensor:
    axis = face >> 1
    dim = t.dim() - 1 - axis
    idx = t.shape[dim] - 1 if (face & 1) else 0
    return t.narrow(dim, idx, 1)


def balance_boundary_fluxes(domain: Domain, free_bounds: Sequence[FixedBoundary], tol: Optional[float] = None):
    """Scale the free boundaries' velocity by ``-flux_fixed / flux_free`` per env where the total
    boundary flux exceeds ``0.01 * tol`` (PISOtorch_simulation.py:188-224).  The reference branches on
    a host-side ``torch.allclose``; here the decision is a per-env ``torch.where`` on the device."""
    blk = domain.getBlock(0)
    fixed_flux = torch.zeros(domain.batch, device=domain.device)
    free_flux = torch.zeros(domain.batch, device=domain.device)
    for f, b in blk.getFixedBoundaries():
        fl = b.GetFluxes().reshape(domain.batch, -1).sum(dim=1)
        fl = fl if (f & 1) else -fl
        if any(b is fb for fb in free_bounds):
            free_flux = free_flux + fl
        else:
            fixed_flux = fixed_flux + fl
    atol = get_solver_tolerance(tol) * 0.01
    need = (fixed_flux + free_flux).abs() > atol
    scale = torch.where(need, -fixed_flux / free_flux, torch.ones_like(free_flux))
    for b in free_bounds:
        b.velocity.mul_(scale.view(-1, *([1] * (b.velocity.dim() - 1))))


def update_advective_boundaries(domain: Domain, bounds: Sequence[FixedBoundary], velms, dt, tol: Optional[float] = None):
    """Convective outflow: ``phi_b <- phi_b - t (phi_b - phi_cell)``, ``t = 1 - 1/(1 + 2 dt (Minv_n . u_m))``
    (PISOtorch_simulation.py:282-389), then flux balancing (:393) -- two small native kernels per call.
    ``dt``: per-env ``[B]`` device tensor (or anything ``NativeSolver.dt_tensor`` accepts); ``velms``: one
    static characteristic velocity ``[1, d]`` / ``[d]`` on the HOST (or a list, one per boundary)."""
    s = domain.solver
    for i, b in enumerate(bounds):
        velm = velms[i] if isinstance(velms, (list, tuple)) else velms
        velm = velm.detach().cpu().numpy() if isinstance(velm, torch.Tensor) else np.asarray(velm)
        s.update_advective_boundary(b.face, velm.reshape(-1)[: domain.dims], dt)
    s.balance_boundary_fluxes([b.face for b in bounds], get_solver_tolerance(tol) * 0.01, dt)
