"""Solver policy switches that change HOW a converged answer is reached, never WHICH equations are solved.

The defaults are the reference's behaviour.  Everything that departs from it is opt-in and named here so that a
benchmark line can say which mode it ran in:

``pressure_warm_start``  (default False)
    The reference starts the first pressure solve of every corrector from zero (``x=None``: orthogonal branch
    ``PISOtorch_simulation.py:1804-1807``; non-orthogonal branch ``x = None if (pstep == 0 or not pressure_reuse_result)``,
    ``:1877-1881``) and re-uses the previous result only for the later non-orthogonal pressure iterations of the same
    corrector.  ``True`` starts from the pressure field of the previous solve instead: same tolerance, same criterion,
    fewer iterations -- a performance mode, reported separately by ``bench.py``.
``advection_warm_start`` (default False)
    The reference's non-orthogonal branch -- cylinder and airfoil envs, and TCF / channel on their rectilinear grids -- starts the
    velocity solve of the first non-orthogonal pass from zero (``x = None if (no_step == 0 or not advect_non_ortho_reuse_result)``,
    ``PISOtorch_simulation.py:1735-1742``); its orthogonal branch (RBC) starts from ``velocityResult`` (``:1689-1693``).  Both
    rules are recorded from the reference's own Python in ``tests/golden/reference_split_step.json``.  ``True`` starts every
    velocity solve from the current velocity: same tolerance, fewer iterations -- a performance mode like ``pressure_warm_start``.
``pressure_stall_accept`` (default 0 = off)
    Multi-block CG only: a solve whose kept iterate is within this factor of the tolerance and has not improved for 20
    iterations ends with that iterate.  This loosens the effective tolerance by the factor, so it
    is off unless asked for.

``pressure_multilevel`` (default True)
    Multi-block 2-D envs whose pressure CG runs on-chip (the cylinder family): the additive multilevel preconditioner of
    ``MultiBlockDomain.set_pressure_multilevel``.  It changes the Krylov trajectory, not the system or its tolerance (the
    single-block path is preconditioned in the same spirit); ``False`` gives the reference's plain CG.
``pressure_multilevel_bicgstab`` (default True since round 3)
    The same preconditioner, in kernel form, as right preconditioner of the pressure BiCGStab of the Airfoil2D envs -- a trial
    (capped attempts verified on the true residual, plain fallback, exponential back-off) that cuts the pressure iterations
    3x and the env step by a third.  It was opt-in in round 2 because identical envs of a batch came apart by 1-30 % in one run
    out of six with it although every solve met its tolerance on the true residual.  The cause was the summation order of the
    dot products (fp64 atomics in arrival order): with the order-independent accumulators of round 3 (``csrc/fg_internal.h``
    FgDacc) identical envs stay BIT-identical and runs reproduce bit for bit with and without the trial
    (``profiles/airfoil_trial_study.py``, ``profiles/r03_airfoil_trial_study.jsonl``: 4 runs x 4 envs x 3 env steps after 40
    development steps), so it is the default.  ``False`` gives the plain refined recurrence.

``advection_line_preconditioner`` (default False)
    Single-block path: on grids refined towards a y wall (largest / smallest y width >= 3: the RBC and TCF families) every
    advection-diffusion BiCGStab is right-preconditioned by the tridiagonal part of its matrix along y (``csrc/fg_linepre.hip``)
    instead of only the repeated ones (the reference's rule, ``BiCG_precondition_fallback``).  Off by default because it does
    not pay on the registered grids: on RBC 512 x 128 the x and y diffusion numbers are comparable (5.7 and 1-14), the line
    solve halves the iterations (44 -> 21 on a synthetic RBC matrix, 11 with an alternating x / y line solve) and costs as
    much per iteration as it saves.  The ``preconditionBiCG`` / ``BiCG_precondition_fallback`` kwargs work either way.

``pressure_bicgstab_large_meshes`` (default True)
    Cylinder envs on meshes beyond the preconditioned on-chip CG (every 3-D id, and 2-D meshes of more than 24 576 cells): the
    pressure systems are solved by the fp64-refined BiCGStab (the airfoil envs' solver, ``pressure_use_BiCG = 2``) instead of the
    reference's plain CG -- same systems, same tolerance; plain CG needs 100-220 iterations per solve there and BiCGStab 37-44
    (``CylinderJet3D-easy-v0`` x 4: 3.2 -> 4.4).  ``False``, or ``pressure_use_BiCG`` given explicitly to the env, keeps the choice.
    Rounds 2-3 also sent the ``medium`` / ``hard`` 2-D ids (resolution 32, 23 k cells) this way (103 -> 245 env-steps/s at 64 envs);
    since round 4 the multilevel-preconditioned on-chip CG reaches them (``k_mbc_l2``: 16 iterations per solve, 527 env-steps/s).

``native_wall_forcing`` (default True)
    Turbulent-channel envs without a sub-grid-scale hook: the dynamic forcing of the ``PRE`` hook (``G_x`` = mean of the two wall
    shear stresses) is computed natively before every PISO step (``fg_set_wall_stress_forcing``) and added to the velocity
    right-hand side as a uniform body force, so the whole ``single_step`` runs in ``fg_single_step`` and no source field is
    streamed.  ``False``: the Python hook writes the block's velocity source, as the reference's does.

``advection_rung_preconditioner`` (default ``"line"``)
    Single-block path: which preconditioner the reference's rungs use -- ``preconditionBiCG`` (every solve) and
    ``BiCG_precondition_fallback`` (a failed solve is repeated with it).  ``"line"``: the tridiagonal part of the matrix along y
    (``csrc/fg_linepre.hip``, ~15 us per application).  ``"ilu0"``: the reference's own preconditioner, ILU(0) of the matrix
    (``csrc/fg_ilu0.hip``: the closed form on the stencil, swept hyperplane by hyperplane by one workgroup per system -- what
    cuSPARSE's level-scheduled triangular solves do for this matrix; 0.5-1 ms per application on the bench grids).  Same
    systems, same tolerances either way; the default is the fast one.

``advection_fd_preconditioner`` (default ``"auto"``)
    Single-block path, grids with periodic uniform x (and z) and walls in y (the RBC and TCF families): every advection-diffusion
    BiCGStab is right-preconditioned by the separable Helmholtz operator ``I/dt - nu Laplacian`` -- its matrix without the advective
    part -- inverted exactly by fast diagonalisation (the pressure preconditioner's eigenvectors + one tridiagonal solve along y per
    mode and env, ``fg_fd_helmholtz_apply``).  What is left for the Krylov method is the advective part: RBC 512 x 128 goes from
    36 / 21 iterations (scalar / velocity) to a handful.  ``"auto"`` = in 2-D only (in 3-D the basis changes cost what they save on
    4-iteration solves), ``"always"`` / ``"never"`` force it.  Same systems, same tolerances, another Krylov trajectory -- like
    ``pressure_multilevel``; the reference's own preconditioner for these solves is ILU(0), off by default.

``advection_jacobi`` (default True)
    Single-block path, velocity systems of 2-D grids with walls in y whose rows are 64 / 128 / 256 / 512 cells (the channel family:
    the headline workload): point-Jacobi sweeps x <- D^-1 (b - O x), 2-8 per pass over the field with the tile kept on chip
    (``csrc/fg_jacobi.hip``), instead of the reference's BiCGStab.  On these grids at the envs' time steps the rows are strongly
    diagonally dominant (sum |O| / D = 0.34 at 256 x 128, 0.63 at 512 x 256) and the sweeps reach the reference's criterion -- RMS
    residual below ``advection_tol``, measured as D (x_new - x) = b - A x -- in 11 / 24 sweeps = 2-6 passes over the matrix, where
    BiCGStab takes 4-8 iterations of two matrix applications and six vector passes each.  A solve the sweeps do not settle (no
    contraction by 0.7 per pass: refined grids, large time steps) is handed to BiCGStab from a cleared start vector and the solver
    backs off from trying.  Same systems, same tolerance, same criterion, another iteration -- like the preconditioners above;
    ``False`` = BiCGStab always (``bench.py`` reports that mode as the ``krylov_mode`` leg).
    Grids without an on-chip region shape (3-D: the turbulent channel) get the same sweeps in streaming form, one launch per sweep
    (TCF: 8 sweeps, contraction 0.03 per sweep at CFL 0.1).
    Multi-block path: the same sweeps over the neighbour table, one launch per sweep (``mb_jacobi``, ``csrc/fg_mb_krylov.hip``): the
    cylinder meshes take 12-16 sweeps where BiCGStab took 5-6 iterations of three launches; the airfoil meshes contract by 0.8 per
    sweep, which the first check sees -- BiCGStab takes over (from the sweeps' iterate while that is well above the tolerance, from
    zero otherwise) and the handle backs off.
    Start vector of the sweeps: zero, like the BiCGStab they stand in for, except on on-chip grids of 2^17 cells and more (512 x 256),
    where the first pass reads the block velocity -- 16.6 sweeps instead of 24 (``FG_JAC_WARM``, docs/SWITCHES.md).

``pressure_refinement`` (default 0 = off)
    Single-block path, an ACCURACY mode: every pressure solve is followed by up to this many corrections of a mixed-precision
    iterative refinement -- the iterate is kept in fp64, ``r = b - P x`` is formed in fp64 with the fp32 matrix entries promoted (as
    the reference's fp64 fallback promotes its CSR values, ``PISOtorch_diff.py:418-445``), each correction is solved by the fp32
    FD-preconditioned CG to a relative tolerance of 1e-4 (``fg_set_pressure_refinement``, ``csrc/fg_poisson.hip``), until the fp64
    residual is 1000 x below the env's pressure tolerance.  What it buys: the fp32 path's distance from the fp64 answer on refined
    grids is the ABSOLUTE residual tolerance of its pressure solves, not kernel error -- RBC 512 x 128: velocity 2.4e-5 of the forcing
    scale without it, 9e-7 with two corrections (``tests/test_gpu_config3.py``), i.e. inside ``north_star``'s 1e-5.  Per env a few
    launches and two host reads per correction: not a fast path.

Set with :func:`set_solver_policy` or the environment variables ``FLUIDGYM_AMD_PRESSURE_WARM_START`` / ``FLUIDGYM_AMD_ADVECTION_WARM_START`` /
``FLUIDGYM_AMD_PRESSURE_STALL_ACCEPT`` / ``FLUIDGYM_AMD_PRESSURE_MULTILEVEL`` / ``FLUIDGYM_AMD_ADVECTION_LINE_PRECONDITIONER`` / ``FLUIDGYM_AMD_PRESSURE_MULTILEVEL_BICGSTAB`` (read once
at import).
"""
from __future__ import annotations

import os
from typing import Any, Dict

_POLICY: Dict[str, Any] = {
    "pressure_warm_start": os.environ.get("FLUIDGYM_AMD_PRESSURE_WARM_START", "0") not in ("0", "", "false", "False"),
    "advection_warm_start": os.environ.get("FLUIDGYM_AMD_ADVECTION_WARM_START", "0") not in ("0", "", "false", "False"),
    "pressure_stall_accept": float(os.environ.get("FLUIDGYM_AMD_PRESSURE_STALL_ACCEPT", "0") or 0.0),
    "pressure_multilevel": os.environ.get("FLUIDGYM_AMD_PRESSURE_MULTILEVEL", "1") not in ("0", "", "false", "False"),
    "advection_line_preconditioner": os.environ.get("FLUIDGYM_AMD_ADVECTION_LINE_PRECONDITIONER", "0") not in ("0", "", "false", "False"),
    "advection_fd_preconditioner": os.environ.get("FLUIDGYM_AMD_ADVECTION_FD_PRECONDITIONER", "auto"),
    "advection_rung_preconditioner": os.environ.get("FLUIDGYM_AMD_ADVECTION_RUNG_PRECONDITIONER", "line"),
    "pressure_bicgstab_large_meshes": os.environ.get("FLUIDGYM_AMD_PRESSURE_BICGSTAB_LARGE_MESHES", "1") not in ("0", "", "false", "False"),
    "native_wall_forcing": os.environ.get("FLUIDGYM_AMD_NATIVE_WALL_FORCING", "1") not in ("0", "", "false", "False"),
    "advection_jacobi": os.environ.get("FLUIDGYM_AMD_ADVECTION_JACOBI", "1") not in ("0", "", "false", "False"),
    "pressure_multilevel_bicgstab": os.environ.get("FLUIDGYM_AMD_PRESSURE_MULTILEVEL_BICGSTAB", "1") not in ("0", "", "false", "False"),
    "pressure_refinement": int(os.environ.get("FLUIDGYM_AMD_PRESSURE_REFINEMENT", "0") or 0),
}


def get_solver_policy() -> Dict[str, Any]:
    return dict(_POLICY)


def set_solver_policy(**kw: Any) -> Dict[str, Any]:
    """Change policy switches for simulations created AFTER the call; returns the previous values."""
    old = dict(_POLICY)
    for k, v in kw.items():
        if k not in _POLICY:
            raise KeyError(f"unknown solver policy {k!r} (known: {sorted(_POLICY)})")
        _POLICY[k] = type(_POLICY[k])(v)
    return old
