"""3-D turbulent channel flow with wall blowing/suction, batched.

Follows ``envs/tcf/tcf_env.py`` and ``envs/tcf/grid.py``:

* grid ``x`` x ``2*(y_half//N)`` x ``z`` with geometric wall refinement (``grid.py:15-81``), channel
  half-height ``delta = 1``, ``L`` x ``2`` x ``D``; periodic in x and z, FIXED no-slip walls at ``+-y``
  (``grid.py:222-227``);
* ``nu = 1 / Re_tau`` with ``u_tau = 1`` (``tcf_env.py:268-275``); initial velocity = Reichardt profile
  (``grid.py:91-112``) + noise;
* dynamic forcing in the ``PRE`` hook: ``G_x = mean wall shear`` of both walls (``grid.py:147-163``);
* action: wall-normal velocity on the ``-y`` wall per actuator patch, zero-mean over the wall so the
  boundary fluxes stay balanced (``tcf_env.py:521-555``); reward = relative wall-shear reduction
  (``tcf_env.py:788-824``);
* solver: adaptive CFL 0.1, pressure tol 1e-6, 2 correctors (``tcf_env.py:478-500``).

Deviations: the reference's curl-simplex-noise initial perturbation (a separate CUDA extension,
``extensions/noise``) is replaced by Gaussian noise made discretely divergence-free; MARL windows
are a "next" item.
"""
from __future__ import annotations

from typing import Any, Dict

import numpy as np
import torch

from .. import spaces
from ..simulation import grids
from ..simulation.domain import Domain
from ..simulation.simulation import Simulation
from .fluid_env import FluidEnv

SMALL_TCF_3D_DEFAULT_CONFIG = {
    "resolution_y": 65,
    "resolution_x_z": 64,
    "actor_size": 2,
    "L": np.pi,
    "D": np.pi / 2,
    "reynolds_number_wall": 180,
    "adaptive_cfl": 0.1,
    "dt": 0.06,
    "step_length": 0.6,
    "episode_length": 1000,
    "use_marl": False,
    "init_with_noise": True,
    "dtype": torch.float32,
    "load_initial_domain": False,
    "load_domain_statistics": False,
    "randomize_initial_state": True,
    "enable_actions": True,
    "differentiable": False,
}
LARGE_TCF_3D_DEFAULT_CONFIG = {**SMALL_TCF_3D_DEFAULT_CONFIG, "resolution_x_z": 128, "L": 2 * np.pi, "D": np.pi}


def reichardt_profile(y_plus: torch.Tensor) -> torch.Tensor:
    """Reichardt's law of the wall (grid.py:94-100)."""
    k = 0.41
    y11 = y_plus / 11.0
    return (1 / k) * torch.log(1 + k * y_plus) + 7.8 * (1 - torch.exp(-y11) - y11 * torch.exp(-y_plus / 3))


class TCF3DBottomEnv(FluidEnv):
    _supports_marl = False
    _metrics = ["wall_stress"]
    _max_action_velocity: float = 1.0  # in units of u_tau

    def __init__(self, resolution_y, resolution_x_z, actor_size, L, D, reynolds_number_wall, adaptive_cfl, step_length,
                 episode_length, dt=0.06, init_with_noise=True, resolution_x=None, resolution_z=None,
                 refinement_strength: int = 1, **kw):
        self._x = int(resolution_x if resolution_x is not None else resolution_x_z)
        self._z = int(resolution_z if resolution_z is not None else resolution_x_z)
        self._y_half = int(resolution_y) // 2
        self._N = int(refinement_strength)
        self._y = 2 * (self._y_half // self._N)
        self._actor = int(actor_size)
        self._L, self._D = float(L), float(D)
        self._re_tau = float(reynolds_number_wall)
        self._nu = 1.0 / self._re_tau
        self._init_with_noise = init_with_noise
        assert self._x % self._actor == 0 and self._z % self._actor == 0
        super().__init__(dt=dt, adaptive_cfl=adaptive_cfl, step_length=step_length, episode_length=episode_length,
                         ndims=3, **kw)

    def _get_action_space(self):
        return spaces.Box(low=-1.0, high=1.0, shape=(self._z // self._actor, self._x // self._actor), dtype=np.float32)

    def _get_observation_space(self):
        shape = (self._z // self._actor, self._x // self._actor)
        return spaces.Dict({"velocity": spaces.Box(low=-np.inf, high=np.inf, shape=(3,) + shape, dtype=np.float32)})

    def _get_domain(self) -> Domain:
        yw = grids.tcf_y_weights(N=self._N, ny_half=self._y_half)
        edges = [grids.lerp_edges(-self._L / 2, self._L / 2, grids.weights_linear(self._x)),
                 grids.lerp_edges(-1.0, 1.0, yw),
                 grids.lerp_edges(-self._D / 2, self._D / 2, grids.weights_linear(self._z))]
        dom = Domain(3, torch.tensor([self._nu]), passiveScalarChannels=0, name="ChannelDomain",
                     device=self._cuda_device, dtype=self._dtype, batch=self._num_envs)
        blk = dom.CreateBlock(vertexCoordinates=grids.vertex_grid(edges), name="ChannelBlock")
        blk.CloseBoundary("-y")
        dom.PrepareSolve()
        blk.setVelocitySource(torch.zeros(1, 3, *dom.solver.spatial))
        return dom

    def _additional_initialization(self) -> None:
        self._block = self._domain.getBlock(0)
        e = self._block.edges[1]
        self._d_wall = (float(0.5 * (e[0] + e[1]) + 1.0), float(1.0 - 0.5 * (e[-1] + e[-2])))
        ycen = torch.from_numpy(0.5 * (e[1:] + e[:-1])).float().to(self._cuda_device)
        self._y_plus = (1 - ycen.abs()) * self._re_tau
        # observation plane: first cell layer with y+ >= 15 (tcf_env.py sensing plane)
        self._obs_j = int(torch.nonzero(self._y_plus >= 15.0)[0]) if bool((self._y_plus >= 15.0).any()) else 1
        self._tau_ref = None

    def _get_prep_fn(self, domain: Domain) -> Dict[str, Any]:
        def forcing(domain, **kw):
            # dynamic forcing G_x = mean of the two wall shear stresses (grid.py:147-163)
            u = self._block.velocity[:, 0]  # [B,Z,Y,X]
            mean_u = u.mean(dim=(1, 3))  # [B,Y]
            G = 0.5 * self._nu * (mean_u[:, 0] / self._d_wall[0] + mean_u[:, -1] / self._d_wall[1])
            src = self._block.velocitySource
            src[:, 0] = G.view(-1, 1, 1, 1)

        return {"PRE": [forcing]}

    def _get_simulation(self, domain, prep_fn) -> Simulation:
        return Simulation(domain=domain, prep_fn=prep_fn, substeps="ADAPTIVE", adaptive_CFL=self._adaptive_cfl,
                          dt=self._dt, corrector_steps=2, pressure_tol=1e-6, advect_non_ortho_steps=1,
                          pressure_non_ortho_steps=1, pressure_return_best_result=True, velocity_corrector="FD",
                          non_orthogonal=True, solver_double_fallback=False)

    def _fill_initial_fields(self) -> None:
        B = self._num_envs
        u_prof = reichardt_profile(self._y_plus)  # u_tau = 1
        u = torch.zeros(B, 3, self._z, self._y, self._x, device=self._cuda_device)
        u[:, 0] = u_prof.view(1, 1, -1, 1)
        if self._init_with_noise:
            u += 0.1 * u_prof.view(1, 1, 1, -1, 1) * torch.randn(u.shape, device=u.device, generator=self._torch_rng_cuda)
        self._block.setVelocity(u)
        self._block.pressure.zero_()
        self._block.getBoundary("-y").velocity.zero_()
        self._block.getBoundary("+y").velocity.zero_()
        self._domain.solver.reset_solver_state()
        self._sim.make_divergence_free()

    def _wall_stress(self) -> torch.Tensor:
        u = self._block.velocity[:, 0]
        return self._nu * u[:, :, 0, :].mean(dim=(1, 2)) / self._d_wall[0]

    def _apply_action(self, action: torch.Tensor) -> None:
        a = action.reshape(self._num_envs, self._z // self._actor, self._x // self._actor)
        a = a - a.mean(dim=(1, 2), keepdim=True)  # zero net mass flux (tcf_env.py:538-545)
        v = a.repeat_interleave(self._actor, dim=1).repeat_interleave(self._actor, dim=2) * self._max_action_velocity
        bv = self._block.getBoundary("-y").velocity  # [B,3,Z,1,X]
        bv.zero_()
        bv[:, 1, :, 0, :] = v

    def _get_global_obs(self):
        u = self._block.velocity[:, :, :, self._obs_j, :]  # [B,3,Z,X]
        u = u.reshape(self._num_envs, 3, self._z // self._actor, self._actor, self._x // self._actor, self._actor)
        return {"velocity": u.mean(dim=(3, 5))}

    def _step_impl(self, action: torch.Tensor):
        if self._tau_ref is None:
            self._tau_ref = self._wall_stress().clone()
        if self._enable_actions:
            self._apply_action(action)
        for _ in range(self._n_sim_steps):
            if not self._sim.single_step():
                raise RuntimeError("simulation step failed")
        tau = self._wall_stress()
        reward = 1.0 - tau / self._tau_ref
        return self._get_global_obs(), reward, False, {"wall_stress": tau}

    @property
    def id(self) -> str:
        return f"TCF3D_Re{self._re_tau}_{self._x}x{self._y}x{self._z}"
