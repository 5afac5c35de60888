"""fluidgym_amd -- MI355X-native (gfx950, HIP) simulation hot path of FluidGym.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + C ABI), ``_lib`` (ctypes
binding), ``native`` (handle wrapper), ``simulation`` (PISO driver mirroring the reference's
``Simulation``), ``envs`` (FluidEnv / ParallelFluidEnv surface) and the registry
(``fluidgym.make``-compatible).  Importing the package never touches the GPU; the shared library is
loaded on first use and its absence is an error (there is no CPU / PyTorch fallback).

Registered ids: the reference registers 39 ids (``fluidgym/__init__.py:28-352``).  The RBC and TCF
families (single block, orthogonal) are registered under the reference's ids with the reference's
defaults; the cylinder (2-D, 3-D) and airfoil (2-D, 3-D) families run on the multi-block curvilinear path
(``envs/cylinder.py``, ``envs/airfoil.py``) on the reference's own meshes.  ``ChannelJet2D-*`` are
single-block stand-ins for the BASELINE configs named on grid sizes the reference's meshes do not have.
"""
from __future__ import annotations

import numpy as np

from .registry import make, register, registry  # noqa: F401
from .config import config  # noqa: F401
from .simulation.policy import get_solver_policy, set_solver_policy  # noqa: F401

__version__ = "0.1.0"


def _lazy(module: str, cls: str):
    def ctor(**kw):
        import importlib

        return getattr(importlib.import_module(module, __name__), cls)(**kw)

    ctor.__name__ = ctor.__qualname__ = cls      # the registry shows the class the id stands for (fluidgym/__init__.py)
    return ctor


def _register_all():
    from .envs.channel import CHANNEL_JET_2D_DEFAULT_CONFIG as CH
    from .envs.rbc import RBC_2D_DEFAULT_CONFIG as R2, RBC_3D_DEFAULT_CONFIG as R3
    from .envs.tcf import LARGE_TCF_3D_DEFAULT_CONFIG as TL, SMALL_TCF_3D_DEFAULT_CONFIG as TS

    ch = _lazy(".envs.channel", "ChannelJetEnv2D")
    register("ChannelJet2D-v0", ch, CH)                                         # BASELINE config 2 stand-in
    register("ChannelJet2D-gate-v0", ch, CH, resolution_x=128, resolution_y=64)  # BASELINE config 1
    register("ChannelJet2D-large-v0", ch, CH, resolution_x=512, resolution_y=256)  # BASELINE config 5
    r2, r3 = _lazy(".envs.rbc", "RBCEnv2D"), _lazy(".envs.rbc", "RBCEnv3D")
    # reference ids (fluidgym/__init__.py): easy/medium/hard = Ra 8e4 / 4e5 / 8e5, CFL 0.8 / 0.5 / 0.5
    register("RBC2D-easy-v0", r2, R2)
    register("RBC2D-medium-v0", r2, R2, rayleigh_number=4e5, adaptive_cfl=0.5)
    register("RBC2D-hard-v0", r2, R2, rayleigh_number=8e5, adaptive_cfl=0.5)
    register("RBC2D-wide-easy-v0", r2, R2, n_heaters=24, aspect_ratio=2)
    register("RBC2D-wide-medium-v0", r2, R2, rayleigh_number=4e5, adaptive_cfl=0.5, n_heaters=24, aspect_ratio=2)
    register("RBC2D-wide-hard-v0", r2, R2, rayleigh_number=8e5, adaptive_cfl=0.5, n_heaters=24, aspect_ratio=2)
    register("RBC2D-baseline-v0", r2, R2, n_heaters=64, resolution=8, aspect_ratio=2.55)  # 512x128 (BASELINE config 3)
    # 3-D: Ra 6e3 / 8e3 / 1e4, multi-agent by default (fluidgym/__init__.py; rbc_env_3d.py default config)
    for tag, extra in (("", {}), ("wide-", dict(n_heaters=16, aspect_ratio=2))):
        register(f"RBC3D-{tag}easy-v0", r3, R3, **extra)
        register(f"RBC3D-{tag}medium-v0", r3, R3, rayleigh_number=8e3, **extra)
        register(f"RBC3D-{tag}hard-v0", r3, R3, rayleigh_number=1e4, **extra)
    # reference ids (fluidgym/__init__.py:215-300): Small/Large x bottom/both x easy/medium/hard = Re_tau 180 / 330 / 550
    tcf = {"bottom": _lazy(".envs.tcf", "TCF3DBottomEnv"), "both": _lazy(".envs.tcf", "TCF3DBothEnv")}
    for size, cfg in (("Small", TS), ("Large", TL)):
        for act, ctor in tcf.items():
            for level, re_tau in (("easy", 180), ("medium", 330), ("hard", 550)):
                register(f"TCF{size}3D-{act}-{level}-v0", ctor, cfg, reynolds_number_wall=re_tau)
    register("TCF3D-baseline-v0", tcf["both"], TS, resolution_x=128, resolution_z=64, resolution_y=64, L=2 * np.pi,
             D=np.pi)  # 128 x 64 x 64 (BASELINE config 4)
    # reference ids (fluidgym/__init__.py:28-75): multi-block curvilinear mesh, non-orthogonal PISO
    from .envs.cylinder import CYLINDER_JET_2D_DEFAULT_CONFIG as CJ, CYLINDER_ROT_2D_DEFAULT_CONFIG as CR
    cj, cr = _lazy(".envs.cylinder", "CylinderJetEnv2D"), _lazy(".envs.cylinder", "CylinderRotEnv2D")
    for level, re, res in (("easy", 100, 24), ("medium", 250, 32), ("hard", 500, 32)):
        register(f"CylinderJet2D-{level}-v0", cj, CJ, reynolds_number=re, resolution=res)
        register(f"CylinderRot2D-{level}-v0", cr, CR, reynolds_number=re, resolution=res)
    from .envs.cylinder import CYLINDER_JET_3D_DEFAULT_CONFIG as CJ3
    cj3 = _lazy(".envs.cylinder", "CylinderJetEnv3D")
    for level, re, res in (("easy", 100, 24), ("medium", 250, 32), ("hard", 500, 48)):   # fluidgym/__init__.py:79-102
        register(f"CylinderJet3D-{level}-v0", cj3, CJ3, reynolds_number=re, resolution=res)
    from .envs.airfoil import AIRFOIL_2D_DEFAULT_CONFIG as AF2
    af2 = _lazy(".envs.airfoil", "AirfoilEnv2D")
    for level, re in (("easy", 1e3), ("medium", 3e3), ("hard", 5e3)):                     # fluidgym/__init__.py:307-328
        register(f"Airfoil2D-{level}-v0", af2, AF2, reynolds_number=re)
    from .envs.airfoil import AIRFOIL_3D_DEFAULT_CONFIG as AF3
    af3 = _lazy(".envs.airfoil", "AirfoilEnv3D")
    for level, re in (("easy", 1e3), ("medium", 3e3), ("hard", 5e3)):                     # fluidgym/__init__.py:333-352
        register(f"Airfoil3D-{level}-v0", af3, AF3, reynolds_number=re)


_register_all()
