"""fluidgym_amd -- MI355X-native (gfx950, HIP) simulation hot path of FluidGym.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + C ABI), ``_lib`` (ctypes
binding), ``native`` (handle wrapper), ``simulation`` (PISO driver mirroring the reference's
``Simulation``), ``envs`` (FluidEnv / ParallelFluidEnv surface).  Importing the package never
touches the GPU; the shared library is loaded on first use and its absence is an error.
"""
__version__ = "0.1.0"
