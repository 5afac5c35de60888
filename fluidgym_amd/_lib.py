"""ctypes binding of ``libfluidgym_hip.so`` (C ABI in ``include/fluidgym_hip.h``).

This is the only door from Python into the solver: there is NO CPU or PyTorch fallback.  If the
shared library is missing, or a call is made without a GPU, the error is raised loudly.
The role this module plays in the reference is the pybind11 module
``fluidgym.simulation.extensions.PISOtorch`` (``extensions/PISOtorch.cpp:40-670``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLUIDGYM_AMD_LIB: another build of the same library (the host-sanitizer build of tests/run_sanitizer_suite.sh); a path that does
# not exist is an error like a missing default build
LIB_PATH = os.environ.get("FLUIDGYM_AMD_LIB") or os.path.join(_HERE, "libfluidgym_hip.so")
# the fp64 build of the single-block entry points (fg_real = double, include/fluidgym_hip.h): FluidEnv(dtype=torch.float64)
LIB_F64_PATH = os.environ.get("FLUIDGYM_AMD_LIB_F64") or os.path.join(_HERE, "libfluidgym_hip_f64.so")

FG_MAX_SCALARS = 4
FG_OK = 0
FG_ERR_NOT_CONVERGED = -5
FG_ERR_NOT_FINITE = -6
FG_ERR_FLUX_BALANCE = -7
FG_PERIODIC, FG_FIXED = 0, 1
FG_DIRICHLET, FG_NEUMANN = 0, 1
FG_VELOCITY, FG_PRESSURE, FG_SCALAR, FG_VELOCITY_SOURCE, FG_VISCOSITY_FIELD = 0, 1, 2, 3, 4
FG_BOUND_VELOCITY, FG_BOUND_SCALAR = 8, 16
FG_SOLVER_CG, FG_SOLVER_JACOBI, FG_SOLVER_RBGS, FG_SOLVER_MGCG, FG_SOLVER_FDCG = 0, 1, 2, 3, 4
(FG_BUF_A, FG_BUF_C_OFF, FG_BUF_ADV_RHS, FG_BUF_VEL_RESULT, FG_BUF_H, FG_BUF_DIV, FG_BUF_P_RESULT,
 FG_BUF_SCALAR_RESULT) = range(8)


class FgConfig(Structure):
    _fields_ = [
        ("dims", c_int32),
        ("nx", c_int32),
        ("ny", c_int32),
        ("nz", c_int32),
        ("batch", c_int32),
        ("n_scalars", c_int32),
        ("face_type", c_int32 * 6),
        ("scalar_bc", (c_int32 * FG_MAX_SCALARS) * 6),
        ("device", c_int32),
    ]


class FgSolveInfo(Structure):
    _fields_ = [
        ("final_residual", c_float),
        ("used_iterations", c_int32),
        ("converged", c_int32),
        ("is_finite", c_int32),
    ]

    def __repr__(self):
        return (f"FgSolveInfo(residual={self.final_residual:.3e}, iterations={self.used_iterations}, "
                f"converged={bool(self.converged)}, finite={bool(self.is_finite)})")


class FgStepOptions(Structure):
    _fields_ = [
        ("corrector_steps", c_int32),
        ("advect_scalar", c_int32),
        ("pressure_method", c_int32),
        ("max_iterations", c_int32),
        ("advection_tol", c_float),
        ("pressure_tol", c_float),
        ("buoyancy_axis", c_int32),
        ("buoyancy_factor", c_float),
        ("pressure_warm_start", c_int32),
    ]


class FgMbStepOptions(Structure):
    _fields_ = [
        ("corrector_steps", c_int32),
        ("advect_non_ortho_steps", c_int32),
        ("pressure_non_ortho_steps", c_int32),
        ("max_iterations", c_int32),
        ("advection_tol", c_float),
        ("pressure_tol", c_float),
        ("pressure_use_bicgstab", c_int32),
        ("pressure_warm_start", c_int32),
        ("pressure_project_mean", c_int32),
        ("pressure_stall_accept", c_float),
        ("solver_double_fallback", c_int32),
        ("bicg_precondition_fallback", c_int32),
    ]


class FgMbSimOptions(Structure):
    _fields_ = [
        ("step", FgMbStepOptions),
        ("time_step", c_float),
        ("cfl", c_float),
        ("adaptive", c_int32),
        ("substeps", c_int32),
        ("flux_balance_tol", c_float),
        ("outflow_slot0", c_int32),
        ("outflow_count", c_int32),
        ("outflow_velm", c_float * 3),
        ("outflow_tol", c_float),
        ("max_substeps", c_int32),
        ("outflow_slot0_b", c_int32),
        ("outflow_count_b", c_int32),
    ]


(FG_MB_BUF_A, FG_MB_BUF_C_OFF, FG_MB_BUF_RHS, FG_MB_BUF_H, FG_MB_BUF_DIV, FG_MB_BUF_P_DIAG, FG_MB_BUF_P_OFF,
 FG_MB_BUF_VELOCITY_RESULT) = range(8)
FG_MB_BUF_KRYLOV0 = 8


class FgSimOptions(Structure):
    _fields_ = [
        ("step", FgStepOptions),
        ("time_step", c_float),
        ("cfl", c_float),
        ("adaptive", c_int32),
        ("substeps", c_int32),
        ("flux_balance_tol", c_float),
        ("outflow_mask", c_int32),
        ("outflow_velm", c_float * 3),
        ("outflow_tol", c_float),
        ("max_substeps", c_int32),
    ]


# name -> (restype, argtypes); must list every symbol declared in include/fluidgym_hip.h
SIGNATURES = {
    "fg_abi_version": (c_int, []),
    "fg_last_error": (c_char_p, []),
    "fg_create": (c_int, [POINTER(FgConfig), POINTER(c_float), POINTER(c_float), POINTER(c_float), POINTER(c_void_p)]),
    "fg_destroy": (c_int, [c_void_p]),
    "fg_bind": (c_int, [c_void_p, c_int, c_void_p]),
    "fg_set_viscosity": (c_int, [c_void_p, c_float]),
    "fg_set_scalar_viscosity": (c_int, [c_void_p, c_int, c_float]),
    "fg_set_fd_preconditioner": (c_int, [c_void_p] + [POINTER(c_float)] * 7),
    "fg_max_velocity": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fg_set_fd_fast_transform": (c_int, [c_void_p, c_int, c_float]),
    "fg_set_return_best": (c_int, [c_void_p, c_int]),
    "fg_set_cg_reset_steps": (c_int, [c_void_p, c_int]),
    "fg_set_advection_start": (c_int, [c_void_p, c_int]),
    "fg_set_wall_stress_forcing": (c_int, [c_void_p, c_int, c_float, c_float]),
    "fg_boundary_flux_balance": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fg_step_diagnostics": (c_int, [c_void_p, POINTER(c_float), c_void_p]),
    "fg_update_advective_boundary": (c_int, [c_void_p, c_int, POINTER(c_float), c_void_p, c_void_p]),
    "fg_balance_boundary_fluxes": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "fg_setup_advection": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fg_solve_advection": (c_int, [c_void_p, c_int, c_int, c_float, c_int, POINTER(FgSolveInfo), c_void_p]),
    "fg_copy_scalar_result_to_blocks": (c_int, [c_void_p, c_int, c_void_p]),
    "fg_setup_pressure_matrix": (c_int, [c_void_p, c_void_p]),
    "fg_setup_pressure_rhs": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fg_solve_pressure": (c_int, [c_void_p, c_int, c_float, c_int, c_int, POINTER(FgSolveInfo), c_void_p]),
    "fg_correct_velocity": (c_int, [c_void_p, c_void_p]),
    "fg_copy_velocity_result_to_blocks": (c_int, [c_void_p, c_void_p]),
    "fg_copy_velocity_result_from_blocks": (c_int, [c_void_p, c_void_p]),
    "fg_piso_step": (c_int, [c_void_p, c_void_p, POINTER(FgStepOptions), POINTER(c_int32), c_void_p]),
    "fg_single_step": (c_int, [c_void_p, POINTER(FgSimOptions), POINTER(c_int32), POINTER(c_float), c_void_p]),
    "fg_multi_step": (c_int, [c_void_p, POINTER(FgSimOptions), c_int32, POINTER(c_void_p), POINTER(c_int32), POINTER(c_float), POINTER(c_int32), c_void_p]),
    "fg_make_divergence_free": (c_int, [c_void_p, c_float, c_int, POINTER(FgSolveInfo), c_void_p]),
    "fg_reset_solver_state": (c_int, [c_void_p, c_void_p]),
    "fg_solver_hints": (c_int, [c_void_p, POINTER(c_int32), c_int32]),
    "fg_set_pressure_refinement": (c_int, [c_void_p, c_int32, c_float, c_float]),
    "fg_get_buffer": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_int64)]),
    "fg_read_buffer": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "fg_poisson_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fg_poisson_jacobi": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "fg_poisson_rbgs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "fg_poisson_cg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, POINTER(FgSolveInfo),
                              c_void_p]),
    "fg_poisson_fdcg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, POINTER(FgSolveInfo),
                                c_void_p]),
    "fg_sparse_apply_ell": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "fg_sparse_apply_csr": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "fg_stream_triad": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int32, POINTER(c_float), c_void_p]),
    "fg_mb_multilevel_status": (c_int, [c_void_p, POINTER(c_int32)]),
    "fg_mb_multilevel_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fg_mb_debug_ilu_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fg_mb_wall_forces": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_float, c_void_p, c_void_p]),
    "fg_mb_debug_bicgstab": (c_int, [c_void_p, c_float, c_int32, c_int32, POINTER(c_int64), POINTER(ctypes.c_double), POINTER(c_float), c_void_p]),
    "fg_dacc_host_sum": (c_int, [POINTER(ctypes.c_double), c_int64, ctypes.c_double, POINTER(ctypes.c_double)]),
    "fg_dacc_device_sum": (c_int, [POINTER(ctypes.c_double), c_int64, ctypes.c_double, c_int32, POINTER(ctypes.c_double), c_void_p]),
    "fg_coherence_litmus": (c_int, [c_int32, c_int32, c_int32, c_int32, POINTER(c_int64), POINTER(ctypes.c_double), c_void_p]),
    "fg_profile_enable": (c_int, [c_void_p, c_int]),
    "fg_profile_kinds": (c_int, []),
    "fg_profile_kind_name": (ctypes.c_char_p, [c_int]),
    "fg_profile_read": (c_int, [c_void_p, c_int, POINTER(ctypes.c_double), POINTER(c_int64), POINTER(ctypes.c_double),
                                POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                                POINTER(c_int64), POINTER(c_int64), POINTER(ctypes.c_double), POINTER(c_int64)]),
    "fg_coords_to_transforms": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "fg_resampler_create": (c_int, [c_int, POINTER(ctypes.c_int32), POINTER(ctypes.c_int32), POINTER(ctypes.c_int32),
                                    POINTER(ctypes.c_float), c_int, c_int, POINTER(c_void_p)]),
    "fg_resampler_destroy": (c_int, [c_void_p]),
    "fg_resample": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "fg_envglue_jet_schedule": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "fg_envglue_channel_observe": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fg_mb_create": (c_int, [c_int32, c_int32, c_int32, POINTER(c_void_p)]),
    "fg_mb_destroy": (c_int, [c_void_p]),
    "fg_config_dump": (c_int, [c_void_p, ctypes.c_char_p, c_int]),
    "fg_mb_config_dump": (c_int, [c_void_p, ctypes.c_char_p, c_int]),
    "fg_mb_add_block": (c_int, [c_void_p, POINTER(c_float), c_int32, c_int32, c_int32, POINTER(c_int32)]),
    "fg_mb_connect": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "fg_mb_make_periodic": (c_int, [c_void_p, c_int32, c_int32]),
    "fg_mb_set_reference_quirks": (c_int, [c_void_p, c_int32, c_int32]),
    "fg_mb_set_nonortho_flags": (c_int, [c_void_p, c_int32]),
    "fg_mb_finalize": (c_int, [c_void_p]),
    "fg_mb_sizes": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32)]),
    "fg_mb_block_info": (c_int, [c_void_p, c_int32, POINTER(c_int32), POINTER(c_int32)]),
    "fg_mb_get_host_table": (c_int, [c_void_p, c_int32, c_void_p, POINTER(c_int64)]),
    "fg_mb_get_neighbors": (c_int, [c_void_p, POINTER(c_int32)]),
    "fg_mb_bind": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fg_mb_set_viscosity": (c_int, [c_void_p, c_float]),
    "fg_mb_piso_step": (c_int, [c_void_p, c_void_p, POINTER(FgMbStepOptions), POINTER(c_int32), c_void_p]),
    "fg_mb_max_velocity": (c_int, [c_void_p, POINTER(c_float), c_void_p]),
    "fg_mb_set_residual_projection": (c_int, [c_void_p, POINTER(c_float)]),
    "fg_mb_set_stall_limit": (c_int, [c_void_p, c_int32]),
    "fg_mb_set_advection_start": (c_int, [c_void_p, c_int]),
    "fg_mb_set_advection_jacobi": (c_int, [c_void_p, c_int]),
    "fg_mb_advection_jacobi_counts": (c_int, [c_void_p, POINTER(c_int64)]),
    "fg_mb_set_multilevel": (c_int, [c_void_p, c_int32, c_int32, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_float),
                                     POINTER(c_float), c_float, c_int32]),
    "fg_mb_env_status": (c_int, [c_void_p, POINTER(c_int32)]),
    "fg_mb_ladder": (c_int, [c_void_p, POINTER(c_int64), c_int32]),
    "fg_mb_debug_cycles": (c_int, [c_void_p, POINTER(ctypes.c_uint64)]),
    "fg_mb_solver_hints": (c_int, [c_void_p, POINTER(c_int32), c_int32]),
    "fg_mb_solver_unconverged": (c_int, [c_void_p, POINTER(c_int64)]),
    "fg_solver_unconverged": (c_int, [c_void_p, POINTER(c_int64)]),
    "fg_mb_solver_counters": (c_int, [c_void_p, POINTER(c_int64), c_int32]),
    "fg_set_advection_preconditioner": (c_int, [c_void_p, c_int]),
    "fg_debug_apply_preconditioner": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fg_sgs_smagorinsky": (c_int, [c_void_p, c_float, c_void_p, c_void_p]),
    "fg_set_fd_helmholtz": (c_int, [c_void_p, POINTER(c_float)]),
    "fg_advection_retries": (c_int, [c_void_p, POINTER(c_int64), c_int32]),
    "fg_advection_solver_form": (c_int, [c_void_p, c_int, POINTER(c_int32)]),
    "fg_set_double_fallback": (c_int, [c_void_p, c_int]),
    "fg_set_advection_jacobi": (c_int, [c_void_p, c_int]),
    "fg_advection_jacobi_counts": (c_int, [c_void_p, POINTER(c_int64)]),
    "fg_ladder": (c_int, [c_void_p, POINTER(c_int64), c_int32]),
    "fg_solver_counters": (c_int, [c_void_p, POINTER(c_int64), c_int32]),
    "fg_mb_profile_iterations": (c_int, [c_void_p, POINTER(c_int64)]),
    "fg_mb_unit_pressure_matrix": (c_int, [c_void_p, c_void_p]),
    "fg_mb_profile_enable": (c_int, [c_void_p, c_int32]),
    "fg_mb_profile_kind_name": (c_char_p, [c_int32]),
    "fg_mb_profile_read": (c_int, [c_void_p, c_int32, POINTER(ctypes.c_double), POINTER(c_int64), POINTER(ctypes.c_double),
                                   POINTER(c_int64)]),
    "fg_mb_get_buffer": (c_int, [c_void_p, c_int32, POINTER(c_void_p), POINTER(c_int64)]),
    "fg_mb_read_buffer": (c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    "fg_mb_single_step": (c_int, [c_void_p, POINTER(FgMbSimOptions), POINTER(c_int32), POINTER(c_float), c_void_p]),
    "fg_mb_update_advective_boundary": (c_int, [c_void_p, c_float, c_int32, c_int32, c_int32, c_int32, POINTER(c_float), c_float, c_void_p]),
    "fg_mb_make_divergence_free": (c_int, [c_void_p, POINTER(FgMbStepOptions), c_void_p]),
    "fg_mb_boundary_flux_balance": (c_int, [c_void_p, POINTER(c_float), c_void_p]),
    "fg_mb_get_boundary_tables": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_float)]),
    "fg_mb_get_cell_transforms": (c_int, [c_void_p, POINTER(c_float)]),
}

# ---- the fp64 build: the single-block AND (since round 3) the multi-block entry points with fg_real = double.  Same names, every
# float of a signature (by value, by pointer, inside the option structures) becomes a double; the fp32-only entry points keep
# their types (they answer FG_ERR_UNSUPPORTED there) and the resampling symbols are not part of that library.
def _f64_struct(cls):
    fields = []
    for name, tp in cls._fields_:
        if tp is c_float:
            tp = ctypes.c_double
        elif isinstance(tp, type) and issubclass(tp, ctypes.Array) and tp._type_ is c_float:
            tp = ctypes.c_double * tp._length_
        elif isinstance(tp, type) and issubclass(tp, Structure) and tp in _F64_STRUCTS:
            tp = _F64_STRUCTS[tp]
        fields.append((name, tp))
    return type(cls.__name__ + "F64", (Structure,), {"_fields_": fields})


_F64_STRUCTS: dict = {}
_F64_STRUCTS[FgStepOptions] = _f64_struct(FgStepOptions)
_F64_STRUCTS[FgSimOptions] = _f64_struct(FgSimOptions)
_F64_STRUCTS[FgMbStepOptions] = _f64_struct(FgMbStepOptions)
_F64_STRUCTS[FgMbSimOptions] = _f64_struct(FgMbSimOptions)
FgStepOptionsF64, FgSimOptionsF64 = _F64_STRUCTS[FgStepOptions], _F64_STRUCTS[FgSimOptions]
FgMbStepOptionsF64, FgMbSimOptionsF64 = _F64_STRUCTS[FgMbStepOptions], _F64_STRUCTS[FgMbSimOptions]
_F64_KEEP_FLOAT = ("fg_set_fd_preconditioner", "fg_set_fd_fast_transform", "fg_set_fd_helmholtz", "fg_coords_to_transforms", "fg_stream_triad")
_F64_ABSENT_PREFIXES = ("fg_resampl", "fg_sparse_", "fg_envglue_")


def _f64_type(tp):
    if tp is c_float:
        return ctypes.c_double
    if tp is POINTER(c_float):
        return POINTER(ctypes.c_double)
    for k, v in _F64_STRUCTS.items():
        if tp is POINTER(k):
            return POINTER(v)
    return tp


SIGNATURES_F64 = {name: ((res, args) if name in _F64_KEEP_FLOAT else (res, [_f64_type(a) for a in args]))
                  for name, (res, args) in SIGNATURES.items() if not name.startswith(_F64_ABSENT_PREFIXES)}

_lib = None
_lib_f64 = None


class NativeLibraryError(RuntimeError):
    """The HIP extension is missing or failed -- there is no fallback path."""


def load() -> ctypes.CDLL:
    """Load the shared library (once) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C fluidgym_amd/csrc`. fluidgym_amd has no CPU / PyTorch fallback."
        )
    # PyTorch-ROCm bundles its own libamdhip64; import it FIRST so that this library's
    # libamdhip64.so.7 dependency resolves to the runtime already in the process (two HIP runtimes in
    # one process do not see each other's devices/streams).
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    if lib.fg_abi_version() != 1:
        raise NativeLibraryError("libfluidgym_hip.so ABI version mismatch")
    _lib = lib
    return lib


def load_f64() -> ctypes.CDLL:
    """The fp64 build (``libfluidgym_hip_f64.so``), typed with ``SIGNATURES_F64``."""
    global _lib_f64
    if _lib_f64 is not None:
        return _lib_f64
    if not os.path.exists(LIB_F64_PATH):
        raise NativeLibraryError(f"{LIB_F64_PATH} not found (dtype=torch.float64 needs the fp64 build: `make -C fluidgym_amd/csrc`). "
                                 "fluidgym_amd has no CPU / PyTorch fallback.")
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_F64_PATH)
    for name, (res, args) in SIGNATURES_F64.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.fg_abi_version() != 1:
        raise NativeLibraryError("libfluidgym_hip_f64.so ABI version mismatch")
    _lib_f64 = lib
    return lib


def check(rc: int, allow=(), lib=None):
    """Raise on a negative status (except those in ``allow``), with the message of the library that reported it."""
    if rc == FG_OK or rc in allow:
        return rc
    msg = (lib or load()).fg_last_error()
    if not msg and lib is None and _lib_f64 is not None:
        msg = _lib_f64.fg_last_error()
    raise NativeLibraryError(f"libfluidgym_hip call failed with status {rc}: {msg.decode() if msg else ''}")
