from .fluid_env import FluidEnv  # noqa: F401
from .parallel_env import ParallelFluidEnv  # noqa: F401
