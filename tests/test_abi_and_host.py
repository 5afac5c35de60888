"""CPU-side checks of the boundary: the shared library loads and exports every symbol that
include/fluidgym_hip.h declares (no compute calls without a GPU), error paths return codes instead
of aborting, the registry has the reference's semantics, and the product refuses to run without the
HIP extension / GPU (no fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import fluidgym_amd
from fluidgym_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fluidgym_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in fluidgym_hip.h but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature in fluidgym_amd/_lib.py"
    assert set(L.SIGNATURES) <= set(declared)
    assert lib.fg_abi_version() == 1


def test_fp64_build_exports_the_single_block_entry_points_with_double_signatures():
    """libfluidgym_hip_f64.so (fg_real = double, include/fluidgym_hip.h): every single-block and multi-block symbol of the header,
    typed with doubles where the fp32 library takes floats; the resampling and env-glue symbols are not part of it."""
    lib = L.load_f64()
    declared = [n for n in _declared_symbols() if not n.startswith(L._F64_ABSENT_PREFIXES)]
    assert set(L.SIGNATURES_F64) == set(declared)
    for name in declared:
        assert hasattr(lib, name), name
    assert not hasattr(lib, "fg_resample")
    assert L.SIGNATURES_F64["fg_mb_set_viscosity"][1][1] is ctypes.c_double
    assert dict(L.FgMbSimOptionsF64._fields_)["cfl"] is ctypes.c_double
    # a host-only handle of the fp64 build builds its tables from double coordinates
    h2 = ctypes.c_void_p()
    assert lib.fg_mb_create(2, 1, -1, ctypes.byref(h2)) == 0
    c = np.stack(np.meshgrid(np.linspace(0, 1, 4), np.linspace(0, 1, 5), indexing="xy")).astype(np.float64)   # [2, ny+1, nx+1]
    bid = ctypes.c_int32(-1)
    assert lib.fg_mb_add_block(h2, np.ascontiguousarray(c).ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 3, 4, 1, ctypes.byref(bid)) == 0
    assert lib.fg_mb_finalize(h2) == 0
    T = np.zeros((12, 5), np.float64)
    assert lib.fg_mb_get_cell_transforms(h2, T.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
    assert np.allclose(T[:, 4], (1 / 3) * (1 / 4), rtol=1e-14)     # det = cell area, to double precision
    assert lib.fg_mb_destroy(h2) == 0
    assert lib.fg_abi_version() == 1
    assert L.SIGNATURES_F64["fg_set_viscosity"][1][1] is ctypes.c_double
    assert L.SIGNATURES_F64["fg_create"][1][1] is ctypes.POINTER(ctypes.c_double)
    assert dict(L.FgSimOptionsF64._fields_)["time_step"] is ctypes.c_double
    assert dict(L.FgStepOptionsF64._fields_)["advection_tol"] is ctypes.c_double
    # error paths answer with codes there too
    cfg = L.FgConfig()
    cfg.dims, cfg.nx, cfg.ny, cfg.nz, cfg.batch = 4, 8, 8, 1, 1
    h = ctypes.c_void_p()
    w = np.ones(8, np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    assert lib.fg_create(ctypes.byref(cfg), w.ctypes.data_as(dp), w.ctypes.data_as(dp), None, ctypes.byref(h)) == -1
    assert b"dims" in lib.fg_last_error()


def test_invalid_arguments_return_status_codes():
    lib = L.load()
    cfg = L.FgConfig()
    cfg.dims, cfg.nx, cfg.ny, cfg.nz, cfg.batch = 4, 8, 8, 1, 1
    h = ctypes.c_void_p()
    fp = ctypes.POINTER(ctypes.c_float)
    w = np.ones(8, np.float32)
    rc = lib.fg_create(ctypes.byref(cfg), w.ctypes.data_as(fp), w.ctypes.data_as(fp), None, ctypes.byref(h))
    assert rc == -1 and b"dims" in lib.fg_last_error()
    cfg.dims, cfg.nx = 2, 2  # fewer than 3 cells (domain_structs.cpp:1193-1195)
    assert lib.fg_create(ctypes.byref(cfg), w.ctypes.data_as(fp), w.ctypes.data_as(fp), None, ctypes.byref(h)) == -1
    cfg.nx = 8
    cfg.face_type[2] = L.FG_FIXED  # -y fixed but +y periodic
    assert lib.fg_create(ctypes.byref(cfg), w.ctypes.data_as(fp), w.ctypes.data_as(fp), None, ctypes.byref(h)) == -1
    assert lib.fg_bind(None, 0, None) == -1
    assert lib.fg_destroy(None) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_silent_cpu_fallback():
    from fluidgym_amd.native import NativeSolver

    with pytest.raises(L.NativeLibraryError):
        NativeSolver([np.ones(8, np.float32), np.ones(8, np.float32)], 1)
    env = fluidgym_amd.make("ChannelJet2D-gate-v0", cuda_device=torch.device("cuda", 0))
    with pytest.raises(RuntimeError, match="CUDA is not available"):
        env.reset(seed=0)


def test_missing_library_is_an_error(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libfluidgym_hip.so")
    with pytest.raises(L.NativeLibraryError):
        L.load()


def test_registry_semantics():
    from fluidgym_amd.registry import EnvRegistry

    r = EnvRegistry()
    r.register("A-v0", lambda **kw: kw, {"a": 1, "b": 2}, b=3)
    assert r.make("A-v0", a=5) == {"a": 5, "b": 3}  # kwargs override defaults (registry.py:72)
    with pytest.raises(ValueError):
        r.register("A-v0", dict, {})
    with pytest.raises(ValueError):
        r.make("missing")
    ids = fluidgym_amd.registry.ids
    for must in ("RBC2D-easy-v0", "TCFSmall3D-both-easy-v0", "ChannelJet2D-v0"):
        assert must in ids
    # every env id of the reference is registered (fluidgym/__init__.py:28-352)
    for fam, n in (("CylinderJet2D", 3), ("CylinderRot2D", 3), ("CylinderJet3D", 3), ("Airfoil2D", 3), ("Airfoil3D", 3), ("TCF", 12)):
        assert sum(i.startswith(fam) and "baseline" not in i for i in ids) == n, fam
    a3 = fluidgym_amd.make("Airfoil3D-easy-v0", cuda_device=torch.device("cpu"))
    assert a3._ndims == 3 and a3.action_space.shape == (4, 3)
    cyl = fluidgym_amd.make("CylinderJet2D-medium-v0", cuda_device=torch.device("cpu"))  # construction touches no GPU
    assert (cyl._reynolds_number, cyl._circle_resolution_angular) == (250, 32)
    assert cyl.render_shape == (686, 128, 128) and cyl._n_sim_steps == 25
    assert cyl._sensor_locations.shape == (2, 151)
    c3 = fluidgym_amd.make("CylinderJet3D-hard-v0", cuda_device=torch.device("cpu"))
    assert (c3._reynolds_number, c3._circle_resolution_angular, c3._ndims, c3.n_agents) == (500, 48, 3, 1)
    assert c3.action_space.shape == (8, 1) and c3.observation_space["velocity"].shape == (8, 2, 3, 151)
    m3 = fluidgym_amd.make("CylinderJet3D-easy-v0", cuda_device=torch.device("cpu"), use_marl=True)
    assert m3.n_agents == 8 and m3.action_space.shape == (1,) and m3.observation_space["pressure"].shape == (3, 2, 151)
    with pytest.raises(ValueError, match="evenly divides"):
        fluidgym_amd.make("CylinderJet3D-easy-v0", cuda_device=torch.device("cpu"), n_jets=5)
    with pytest.raises(ValueError, match="multi-agent"):
        fluidgym_amd.make("CylinderJet3D-easy-v0", cuda_device=torch.device("cpu"), local_2d_obs=True)


def test_env_contract_errors_without_gpu():
    env = fluidgym_amd.make("ChannelJet2D-gate-v0", cuda_device=torch.device("cpu"))
    with pytest.raises(RuntimeError, match="must be reset"):
        env.step(torch.zeros(1))
    with pytest.raises(ValueError, match="Seed"):
        env.seed(None)
    assert env._n_sim_steps == 25  # max(1, int(step_length/dt)) (fluid_env.py:840-842)
    assert env.action_space.shape == (1,)
    assert set(env.observation_space.keys()) == {"velocity", "pressure"}
    rbc = fluidgym_amd.make("RBC2D-easy-v0", cuda_device=torch.device("cpu"))
    assert (rbc._x, rbc._y) == (96, 61)  # rbc_env_base.py:177-178
    assert rbc._n_sim_steps == 20


def test_oracle_is_not_imported_by_the_product():
    import subprocess
    import sys

    code = ("import sys, fluidgym_amd, fluidgym_amd.native, fluidgym_amd.simulation, fluidgym_amd.envs, "
            "fluidgym_amd.envs.channel, fluidgym_amd.envs.rbc, fluidgym_amd.envs.tcf; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle leaked'")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_ctypes_structs_match_the_header_layout():
    """Field-by-field comparison of every ctypes Structure with the typedef of the same role in include/fluidgym_hip.h
    (a struct that drifts shifts every later field silently)."""
    import re

    from fluidgym_amd import _lib as L

    text = open(os.path.join(os.path.dirname(__file__), "..", "include", "fluidgym_hip.h")).read()
    pairs = {"fg_step_options": L.FgStepOptions, "fg_sim_options": L.FgSimOptions, "fg_mb_step_options": L.FgMbStepOptions,
             "fg_mb_sim_options": L.FgMbSimOptions, "fg_solve_info": L.FgSolveInfo, "fg_config": L.FgConfig}
    for cname, cls in pairs.items():
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname + ";", text, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            if decl.strip():
                names += [re.match(r"\s*(\w+)", x).group(1) for x in decl.strip().split(None, 1)[1].split(",")]
        assert names == [f[0] for f in cls._fields_], cname
