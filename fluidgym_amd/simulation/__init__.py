from .domain import Block, BoundaryConditionType, Domain, FixedBoundary  # noqa: F401
from .simulation import Simulation, balance_boundary_fluxes, update_advective_boundaries  # noqa: F401
from .policy import get_solver_policy, set_solver_policy  # noqa: F401
