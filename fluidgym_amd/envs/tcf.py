"""3-D turbulent channel flow with wall blowing/suction, batched.

Mirrors ``envs/tcf/tcf_env.py`` (``TCF3DBottomEnv`` :94-1063, ``TCF3DBothEnv`` :1065-1200) and ``envs/tcf/grid.py``:

* units as in the reference: ``Re_cl = (Re_tau / 0.116)^(1/0.88)`` (``TCF_tools.py:36-41``), ``nu = delta / Re_cl``,
  ``u_tau = Re_tau / Re_cl`` (``tcf_env.py:246-249``); ``step_length`` is given in wall units and converted with
  ``t* = nu / u_tau^2`` (``:262``), ``dt = step_length / 10`` (``:265``);
* grid ``x`` x ``2 * y_half`` x ``z`` with the geometric wall refinement of ``_make_y_weights`` (``grid.py:15-31``,
  strength 2 below 64 cells in x, ``tcf_env.py:253``), ``L x 2 x D`` centred on the origin; periodic in x and z, no-slip
  FIXED walls at ``+-y`` (``grid.py:222-227``); initial velocity = Reichardt profile in wall units times ``u_tau``
  (``grid.py:84-100``);
* dynamic forcing in the ``PRE`` hook: ``G_x`` = mean of the two wall shear stresses (``grid.py:147-176``);
* action ``[n_agents, 1]``: wall-normal velocity per ``actor_size`` x ``actor_size`` patch, made zero-mean and clamped
  to ``u_tau`` (``_action_to_control`` :521-547); "both" drives the top wall with the second half of the agents,
  sign flipped (``:1143-1154``);
* observation: fluctuation velocity ``u - <u>_V`` (components x, y) and pressure on the plane ``y+ = 15``
  (``_get_global_obs`` :646-677), stacked bottom/top for "both" (``:1166-1180``);
* reward ``1 - tau / tau_ref`` with the wall stress averaged over the sim steps of the env step (``:788-824``),
  ``tau_ref`` = 1 without domain statistics (``:556-562``); multi-agent mode: per-actuator windows of patch means
  (``_get_local_obs`` :918-992, ``:1182-1195``) and the global reward for every agent (``:994-1010``);
* solver: adaptive CFL 0.1, advection and pressure tol 1e-6, 2 correctors (``:478-500``).

Deviations: the reference's curl-simplex-noise initial perturbation (a separate CUDA extension, ``extensions/noise``)
is replaced by Gaussian noise made discretely divergence-free; no published initial domains / statistics (no network).
The Smagorinsky sub-grid viscosity (``C_smag != 0``, ``use_van_driest``: off in every registered id) runs as the reference's ``PRE``
hook on ``fg_sgs_smagorinsky`` and a per-cell viscosity field (``Block.setViscosity``).
"""
from __future__ import annotations

from typing import Any, Dict

import numpy as np
import torch

from .. import spaces
from ..simulation import grids
from ..simulation.domain import Domain
from ..simulation.simulation import Simulation
from . import obs_extraction as X
from .fluid_env import FluidEnv

SMALL_TCF_3D_DEFAULT_CONFIG = {
    "resolution_y": 65,
    "resolution_x_z": 64,
    "actor_size": 2,
    "L": np.pi,
    "D": np.pi / 2,
    "reynolds_number_wall": 180,
    "adaptive_cfl": 0.1,
    "step_length": 0.6,   # wall units (tcf_env.py:69); dt = step_length / 10 in physical units
    "episode_length": 1000,
    "local_obs_window": 1,
    "local_reward_weight": 0.0,
    "use_marl": True,   # tcf_env.py:73
    "init_with_noise": True,
    "C_smag": 0.0,            # Smagorinsky coefficient (tcf_env.py:441-474); 0 = no sub-grid model (every registered id)
    "use_van_driest": False,
    "dtype": torch.float32,
    "load_initial_domain": True,
    "load_domain_statistics": True,
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
    _supports_marl = True
    _actuation = "bottom"
    _scale_actions = True
    _delta: float = 1.0
    _H: float = 2.0
    _y_obs_wall: float = 15.0
    _metrics = ["wall_stress", "wall_stress_bottom", "wall_stress_top"]

    def __init__(self, resolution_y, resolution_x_z, actor_size, L, D, reynolds_number_wall, adaptive_cfl, step_length,
                 episode_length, init_with_noise=True, resolution_x=None, resolution_z=None, C_smag: float = 0.0,
                 use_van_driest: bool = False, local_obs_window: int = 1, local_reward_weight: float = 0.0, dt=None, **kw):
        # Smagorinsky sub-grid viscosity (tcf_env.py:441-474): off in every registered id (C_smag = 0); built since round 3
        self._C_smag, self._use_van_driest = float(C_smag), bool(use_van_driest)
        self._L, self._D = float(L), float(D)
        self._re_wall = float(reynolds_number_wall)
        self._re_center = (self._re_wall / 0.116) ** (1 / 0.88)   # TCF_tools.Re_wall_to_cl
        self._nu = self._delta / self._re_center
        self._u_wall = self._re_wall / self._re_center
        self._x = int(resolution_x if resolution_x is not None else resolution_x_z)
        self._z = int(resolution_z if resolution_z is not None else resolution_x_z)
        self._y_half = int(resolution_y) // 2
        self._y = 2 * self._y_half
        self._grid_refinement_strength = 2 if self._x < 64 else 1
        self._init_with_noise = init_with_noise
        self._actor_size = int(actor_size)
        self._local_obs_window = int(local_obs_window)
        self._local_reward_weight = local_reward_weight
        assert self._x % 4 == 0 and self._x % self._actor_size == 0 and self._z % self._actor_size == 0
        step_length = self._t_wall_to_t(step_length)  # wall units -> physical (tcf_env.py:262)
        super().__init__(dt=(step_length / 10 if dt is None else dt), adaptive_cfl=adaptive_cfl, step_length=step_length,
                         episode_length=episode_length, ndims=3, **kw)

    @property
    def render_shape(self):
        """(x, y, z) of the rendered domain (tcf_env.py:295-301)."""
        x = 2 * self._x
        return (x, int(x / self._L * self._H), int(x / self._L * self._D))

    @property
    def scale_actions(self) -> bool:
        """Whether actions are scaled by u_wall (tcf_env.py:429-436)."""
        return self._scale_actions

    @scale_actions.setter
    def scale_actions(self, value: bool) -> None:
        self._scale_actions = value

    # opposition-control baseline episodes next to the initial domains (tcf_env.py:1017-1062)
    def save_opposition_control_episode(self, idx: int, mode, df) -> None:
        path = self._get_domain_dir(idx=idx)
        df.to_csv(path / f"{mode.value}_opposition_control_{self._actuation}_episode.csv", index=False)

    def load_opposition_control_episode(self, idx: int, mode):
        import pandas as pd

        path = self._get_domain_dir(idx=idx)
        return pd.read_csv(path / f"{mode.value}_opposition_control_{self._actuation}_episode.csv")

    # ---- unit conversions (tcf_env.py:323-341) ---------------------------------------------
    def _t_wall_to_t(self, t_wall: float) -> float:
        return t_wall * self._nu / self._u_wall ** 2

    def _t_to_t_wall(self, t: float) -> float:
        return t / (self._nu / self._u_wall ** 2)

    def _y_wall_to_y(self, pos_wall: float) -> float:
        return -self._delta + pos_wall / (self._u_wall / self._nu)

    def _y_to_y_wall(self, pos: float) -> float:
        return (pos + self._delta) * (self._u_wall / self._nu)

    # ---- spaces -----------------------------------------------------------------------------
    @property
    def _n_actors_x(self) -> int:
        return self._x // self._actor_size

    @property
    def _n_actors_z(self) -> int:
        return self._z // self._actor_size

    @property
    def n_agents(self) -> int:
        return self._n_actors_x * self._n_actors_z

    def _get_action_space(self):
        """Per-agent action space (tcf_env.py:367-384)."""
        return spaces.Box(low=-1.0, high=1.0, shape=(1,) if self._use_marl else (self.n_agents, 1), dtype=np.float32)

    def _obs_space(self, velocity_shape, pressure_shape):
        return spaces.Dict({
            "velocity": spaces.Box(low=-np.inf, high=np.inf, shape=velocity_shape, dtype=np.float32),
            "pressure": spaces.Box(low=-np.inf, high=np.inf, shape=pressure_shape, dtype=np.float32),
        })

    def _get_observation_space(self):
        """tcf_env.py:386-428."""
        W = self._local_obs_window
        if self._use_marl:
            return self._obs_space((W, W, 2), (W, W))
        return self._obs_space((2, self._z, self._x), (self._z, self._x))

    # ---- domain / simulation ----------------------------------------------------------------
    def _get_domain(self) -> Domain:
        N = self._grid_refinement_strength
        yw = grids.tcf_y_weights(N=N, ny_half=self._y_half * N)  # _make_grid passes ny_half = y_half * yN (grid.py:56)
        edges = [grids.lerp_edges(-self._L / 2, self._L / 2, grids.weights_linear(self._x)),
                 grids.lerp_edges(-self._delta, self._delta, yw),
                 grids.lerp_edges(-self._D / 2, self._D / 2, grids.weights_linear(self._z))]
        assert len(edges[1]) - 1 == self._y
        dom = Domain(3, torch.tensor([self._nu]), passiveScalarChannels=0, name="ChannelDomain",
                     device=self._cuda_device, dtype=self._dtype, batch=self._num_envs)
        blk = dom.CreateBlock(vertexCoordinates=grids.vertex_grid(edges), name="ChannelBlock")
        blk.CloseBoundary("-y")
        dom.PrepareSolve()
        # dynamic forcing: natively (a uniform body force per env, fg_set_wall_stress_forcing) unless a sub-grid-scale hook keeps the
        # step in the interpreter anyway or the policy says no -- then through the block's velocity source, as in the reference
        from ..simulation.policy import get_solver_policy
        self._native_forcing = bool(get_solver_policy()["native_wall_forcing"]) and self._C_smag == 0.0
        if not self._native_forcing:
            blk.setVelocitySource(torch.zeros(1, 3, *dom.solver.spatial))
        return dom

    def _additional_initialization(self) -> None:
        self._block = self._domain.getBlock(0)
        self._bottom_plate = self._block.getBoundary("-y")
        self._top_plate = self._block.getBoundary("+y")
        e = self._block.edges[1]
        ycen = 0.5 * (e[1:] + e[:-1])
        self._d_wall = (float(1.0 + ycen[0]), float(1.0 - ycen[-1]))   # grid.py:149-150
        self._y_centers = torch.from_numpy(ycen).float().to(self._cuda_device)
        # sensing plane: cell layer closest to y+ = 15 (tcf_env.py:343-352)
        self._y_obs_bottom_idx = int(np.abs(ycen - self._y_wall_to_y(self._y_obs_wall)).argmin())
        self._cell_size = self._block.getCellSizes()[0, 0]   # [Z, Y, X]

    def _get_prep_fn(self, domain: Domain) -> Dict[str, Any]:
        def forcing(domain, **kw):
            tau_b, tau_t = self._get_wall_stress()   # [B] each
            self._block.velocitySource[:, 0] = (0.5 * (tau_b + tau_t)).view(-1, 1, 1, 1)

        hooks = [] if self._native_forcing else [forcing]
        if self._C_smag != 0.0:
            # tcf_env.py:441-474: every (sub)step the block's viscosity = nu + C Delta^2 |S| (x the squared van Driest damping
            # (1 - exp(-y+ / 25))^2 of the wall distance, util.py:75-125), computed from the current velocity
            cache = {}

            def add_block_sgs_viscosity(domain, **kw):
                visc = domain.solver.sgs_smagorinsky(self._C_smag)
                if self._use_van_driest:
                    if "vd2" not in cache:      # (the cell centres exist once _additional_initialization has run)
                        y_plus = (1.0 - self._y_centers.abs()) * (self._u_wall / self._nu)                      # [Y]
                        cache["vd2"] = ((1.0 - torch.exp(-y_plus / 25.0)) ** 2).to(visc.dtype).view(1, 1, -1, 1)   # over [B, Z, Y, X]
                    visc = visc * cache["vd2"]
                self._block.setViscosity(visc + self._nu)

            hooks.append(add_block_sgs_viscosity)
        return {"PRE": hooks}

    def _get_simulation(self, domain, prep_fn) -> Simulation:
        wall_forcing = None
        if self._native_forcing:
            e = domain.getBlock(0).edges[1]
            ycen = 0.5 * (e[1:] + e[:-1])
            wall_forcing = (0, self._nu / float(1.0 + ycen[0]), self._nu / float(1.0 - ycen[-1]))   # grid.py:147-176
        return Simulation(domain=domain, prep_fn=prep_fn, wall_forcing=wall_forcing, substeps="ADAPTIVE", adaptive_CFL=self._adaptive_cfl,
                          dt=self._dt, corrector_steps=2, advection_tol=1e-6, pressure_tol=1e-6, advect_non_ortho_steps=1,
                          pressure_non_ortho_steps=1, pressure_return_best_result=True, velocity_corrector="FD",
                          non_orthogonal=True, solver_double_fallback=False)

    def _fill_initial_fields(self) -> None:
        B = self._num_envs
        y_plus = (1 - self._y_centers.abs()) * self._u_wall / self._nu
        u_prof = reichardt_profile(y_plus) * self._u_wall   # grid.py:84-100
        u = torch.zeros(B, 3, self._z, self._y, self._x, device=self._cuda_device)
        u[:, 0] = u_prof.view(1, 1, -1, 1)
        if self._init_with_noise:
            u += 0.1 * u_prof.view(1, 1, 1, -1, 1) * torch.randn(u.shape, device=u.device, generator=self._torch_rng_cuda)
        self._block.setVelocity(u)
        self._block.pressure.zero_()
        self._bottom_plate.velocity.zero_()
        self._top_plate.velocity.zero_()
        self._domain.solver.reset_solver_state()
        self._sim.make_divergence_free()

    def _randomize_domain(self) -> None:
        """tcf_env.py:879-916: 1 % Gaussian noise on u and p, then a random number of sim steps."""
        max_n = int(0.01 * self._episode_length)
        n_steps = int(self._np_rng.integers(int(0.5 * max_n), max(max_n, int(0.5 * max_n) + 1))) + 1
        u, p = self._block.velocity, self._block.pressure
        u += 0.01 * torch.randn(u.shape, device=u.device, generator=self._torch_rng_cuda)
        p += 0.01 * torch.randn(p.shape, device=p.device, generator=self._torch_rng_cuda)
        self._domain.solver.reset_solver_state()
        for _ in range(n_steps):
            self._sim.single_step()

    # ---- actions ----------------------------------------------------------------------------
    def _action_to_control(self, action: torch.Tensor) -> torch.Tensor:
        """``action [B, nax, naz]`` -> wall-normal velocity ``[B, Z, X]`` (tcf_env.py:521-547)."""
        a = action
        if self._scale_actions:
            a = a - a.mean(dim=(1, 2), keepdim=True)                 # zero net mass flux
            a = self._u_wall * a / torch.clamp(a.abs(), min=1.0)     # |v| <= u_tau
            a = a - a.mean(dim=(1, 2), keepdim=True)
        v = a.repeat_interleave(self._actor_size, dim=1).repeat_interleave(self._actor_size, dim=2)   # [B, X, Z]
        return v.transpose(1, 2)

    def _set_wall(self, plate, v: torch.Tensor) -> None:
        bv = plate.velocity   # [B, 3, Z, 1, X]
        bv.zero_()
        bv[:, 1, :, 0, :] = v

    def _apply_action(self, action: torch.Tensor) -> None:
        a = action.reshape(self._num_envs, self._n_actors_x, self._n_actors_z)
        self._set_wall(self._bottom_plate, self._action_to_control(a))

    # ---- metrics / observations -----------------------------------------------------------
    @property
    def tau_ref(self) -> float:
        return float(self._metrics_stats.get("wall_stress_bottom", 1.0))

    def _get_wall_stress(self):
        """(bottom, top) wall shear stress per env ``[B]`` (tcf_env.py:564-584)."""
        # (only the two wall-adjacent cell layers are needed: the mean over the whole field read 17 MB twice per sim step at
        #  128 x 64 x 64 x 8 envs -- 2 % of the TCF leg -- for two of its 64 rows)
        ux = self._block.velocity[:, 0]                          # [B, Z, Y, X]
        mean_lo, mean_hi = ux[:, :, 0, :].mean(dim=(1, 2)), ux[:, :, -1, :].mean(dim=(1, 2))
        return self._nu * mean_lo / self._d_wall[0], self._nu * mean_hi / self._d_wall[1]

    def _plane_obs(self, y_idx: int):
        u = self._block.velocity                                   # [B, 3, Z, Y, X]
        cs = self._cell_size
        # volume-weighted mean per env and component as ONE pass over the field (a matrix-vector product), subtracted from the plane
        # only: (u * cs).sum() / cs.sum() and (u - mean)[plane] made three full-field passes (a 50 MB product, its sum, a 50 MB
        # difference) for a [B, 2, Z, X] slice
        B = u.shape[0]
        w = cs.reshape(-1).to(u.dtype)
        mean_u = (u[:, :2].reshape(B * 2, -1) @ w).reshape(B, 2, 1, 1) / w.sum()
        return {"velocity": u[:, :2, :, y_idx, :] - mean_u, "pressure": self._block.pressure[:, 0, :, y_idx, :]}

    def _get_global_obs(self):
        return self._plane_obs(self._y_obs_bottom_idx)

    def _local_plane_obs(self, y_idx: int, flip_obs: bool):
        """Per-actuator windows of patch-mean fluctuations on one plane (tcf_env.py:918-992; the plane mean is taken
        over the slice here, and the x padding of u_x is one less than that of u_y / p, both as in the reference)."""
        u = self._block.velocity[:, :2, :, y_idx, :]                     # [B, 2, Z, X]
        p = self._block.pressure[:, 0, :, y_idx, :]
        up = u - u.mean(dim=(2, 3), keepdim=True)
        W, nax, naz, aw = self._local_obs_window, self._n_actors_x, self._n_actors_z, self._actor_size
        win = lambda f, pad_x: X.extract_moving_window_2d_x_z(f, nax, naz, aw, W, W, pad_x, W // 2)   # [B, n, Wz, Wx]
        ux, uy, pw = win(up[:, 0], W - 1), win(up[:, 1], W), win(p, W)
        if flip_obs:   # top wall: consistent orientation (tcf_env.py:957-985)
            ux = torch.flip(ux, dims=[3])
            uy = -torch.flip(uy, dims=[3])
            pw = torch.flip(pw, dims=[2])
        return {"velocity": torch.stack((ux, uy), dim=-1), "pressure": pw}

    def _get_local_obs(self):
        return self._local_plane_obs(self._y_obs_bottom_idx, False)

    def _step_marl_impl(self, action: torch.Tensor):
        """tcf_env.py:994-1010: every agent receives the global reward."""
        if self._local_reward_weight is None:
            raise ValueError("local_reward_weight must be set for multi-agent step.")
        _, global_reward, terminated, info = self._step_impl(action, want_obs=False)      # (the agents' windows are taken below)
        agent_rewards = global_reward.unsqueeze(1).expand(-1, self.n_agents).contiguous()
        info["global_reward"] = global_reward
        return self._get_local_obs(), agent_rewards, terminated, info

    def _get_reward(self, tau_total, tau_bottom):
        return 1 - tau_bottom / self.tau_ref

    def _step_impl(self, action: torch.Tensor, want_obs: bool = True):
        if self._enable_actions:
            self._apply_action(action)
        tb, tt = [], []
        for _ in range(self._n_sim_steps):
            if not self._sim.single_step():
                raise RuntimeError("simulation step failed")
            b, t = self._get_wall_stress()
            tb.append(b)
            tt.append(t)
        tau_bottom, tau_top = torch.stack(tb).mean(dim=0), torch.stack(tt).mean(dim=0)
        tau_total = 0.5 * (tau_bottom + tau_top)
        info = {"wall_stress": tau_total, "wall_stress_bottom": tau_bottom, "wall_stress_top": tau_top}
        return (self._get_global_obs() if want_obs else None), self._get_reward(tau_total, tau_bottom), False, info

    @property
    def id(self) -> str:
        return f"ChannelFlow3D_Re{int(self._re_wall)}_L{self._L:.2f}"   # tcf_env.py:874-877

    @property
    def initial_domain_id(self) -> str:
        """tcf_env.py:866-872."""
        return f"channel_flow3D_L{self._L:.2f}_Re{int(self._re_wall)}_Res{self._x}_Ref{self._grid_refinement_strength}"


class TCF3DBothEnv(TCF3DBottomEnv):
    """Both walls actuated: the first half of the agents drives the bottom wall, the second half the top wall
    (tcf_env.py:1065-1200)."""

    _actuation = "both"

    @property
    def n_agents(self) -> int:
        return 2 * self._n_actors_x * self._n_actors_z

    def _get_observation_space(self):
        """tcf_env.py:1081-1124."""
        W = self._local_obs_window
        if self._use_marl:
            return self._obs_space((W, W, 2), (W, W))
        return self._obs_space((2, 2, self._z, self._x), (2, self._z, self._x))

    def _get_local_obs(self):
        """Bottom-wall agents first, then the top-wall agents with flipped orientation (tcf_env.py:1182-1195)."""
        lo = self._local_plane_obs(self._y_obs_bottom_idx, False)
        hi = self._local_plane_obs(min(self._y_obs_top_idx, self._y - 1), True)
        return {k: torch.cat((lo[k], hi[k]), dim=1) for k in lo}

    @property
    def tau_ref(self) -> float:
        return float(self._metrics_stats.get("wall_stress", 1.0))

    def _additional_initialization(self) -> None:
        super()._additional_initialization()
        self._y_obs_top_idx = self._y - self._y_obs_bottom_idx   # tcf_env.py:1141 (sic: not y - 1 - idx)

    def _apply_action(self, action: torch.Tensor) -> None:
        a = action.reshape(self._num_envs, 2, self._n_actors_x, self._n_actors_z)
        self._set_wall(self._bottom_plate, self._action_to_control(a[:, 0]))
        self._set_wall(self._top_plate, -self._action_to_control(a[:, 1]))

    def _get_reward(self, tau_total, tau_bottom):
        return 1 - tau_total / self.tau_ref

    def _get_global_obs(self):
        lo, hi = self._plane_obs(self._y_obs_bottom_idx), self._plane_obs(min(self._y_obs_top_idx, self._y - 1))
        return {"velocity": torch.stack((lo["velocity"], hi["velocity"]), dim=1),
                "pressure": torch.stack((lo["pressure"], hi["pressure"]), dim=1)}
