"""2-D channel with wall jets: the single-block stand-in for BASELINE.json's "cylinder" and
"airfoil" configs.

The reference's cylinder / airfoil envs are 5- and 6-block body-fitted curvilinear meshes
(``envs/cylinder/grid.py:283-418``, ``envs/airfoil/grid.py:629-707``) and it has no Cartesian or
immersed-boundary formulation at all, so the synthetic BASELINE configs
"2D cylinder 256x128, batch 64" and "2D airfoil 512x256, batch 512" are mapped -- as SURVEY.md
section 8d prescribes -- onto a single block carrying the cylinder env's boundary-condition set:

* ``-x`` FIXED Dirichlet parabolic inflow with unit mean (``envs/util/profiles.py:35-90``),
* ``+x`` FIXED varying Dirichlet, updated every substep by the advective outflow rule and re-balanced
  (``pict/PISOtorch_simulation.py:228-393``; cylinder wiring ``cylinder_env_base.py:280-300``),
* ``+-y`` FIXED no-slip walls carrying a zero-net-flux pair of jets (the cylinder's two jets,
  ``jet_cylinder_env_2d.py:134-188``, moved to the channel walls),
* channel ``L x H = 22 x 4.1``, ``U_mean = 1``, ``nu = 1/Re`` (``cylinder_env_base.py:120-126``),
  ``dt = 0.01``, ``step_length = 0.25`` -> 25 PISO steps per env step, adaptive CFL 0.8, action
  smoothing ``alpha = 0.1`` per sim step (``cylinder_env_base.py:118, 748-753``), pressure tol 1e-5.

Everything the solver does per step is therefore exactly what the reference's cylinder env asks of it
on a single block; what is missing is the body (multi-block connections).
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import numpy as np
import torch

from .. import spaces
from ..simulation import grids
from ..simulation.domain import Domain
from ..simulation.simulation import Simulation, update_advective_boundaries
from .fluid_env import FluidEnv

CHANNEL_JET_2D_DEFAULT_CONFIG = {
    "reynolds_number": 1e2,
    "resolution_x": 256,
    "resolution_y": 128,
    "dt": 1e-2,
    "adaptive_cfl": 0.8,
    "step_length": 0.25,
    "episode_length": 80,
    "lift_penalty": 1.0,
    "use_marl": False,
    "dtype": torch.float32,
    "load_initial_domain": False,
    "load_domain_statistics": False,
    "randomize_initial_state": True,
    "enable_actions": True,
    "differentiable": False,
}


def inflow_profile(h: float, res_y: int) -> np.ndarray:
    """Parabolic profile with unit mean sampled like ``get_inflow_profile`` (profiles.py:72-77)."""
    y = np.linspace(-h / 2, h / 2, res_y)
    p = 6 * (h / 2 - y) * (h / 2 + y) / h**2
    return p / p.mean()


def jet_profile(h: int) -> np.ndarray:
    """``get_jet_profile`` (profiles.py:6-32): parabola over ``h`` cells with unit maximum."""
    y = np.linspace(-h / 2, h / 2, h)
    p = 6 * (h / 2 - y) * (h / 2 + y) / h**2
    return p / p.max()


class ChannelJetEnv2D(FluidEnv):
    _supports_marl = False
    _action_smoothing_alpha: float = 0.1
    H: float = 4.1
    L: float = 22.0
    _U_mean: float = 1.0
    _jet_max: float = 1.0  # jet centre-line velocity at |action| = 1 (in units of U_mean)
    _n_sensors_x: int = 16
    _n_sensors_y: int = 8
    _metrics = ["cross_flow_energy", "wall_shear"]

    def __init__(self, reynolds_number: float, resolution_x: int, resolution_y: int, dt: float, adaptive_cfl: float,
                 step_length: float, episode_length: int, lift_penalty: float = 1.0, **kw):
        self._reynolds_number = reynolds_number
        self._x, self._y = int(resolution_x), int(resolution_y)
        self._lift_penalty = lift_penalty
        self._nu = self._U_mean * 1.0 / reynolds_number
        self._jet_cells = max(4, self._x // 32)
        self._jet_start = self._x // 5
        super().__init__(dt=dt, adaptive_cfl=adaptive_cfl, step_length=step_length, episode_length=episode_length,
                         ndims=2, **kw)
        self._current_action = None

    # ---- spaces ---------------------------------------------------------------------------
    def _get_action_space(self):
        return spaces.Box(low=-1.0, high=1.0, shape=(1,), dtype=np.float32)

    def _get_observation_space(self):
        n = self._n_sensors_x * self._n_sensors_y
        return spaces.Dict({
            "velocity": spaces.Box(low=-np.inf, high=np.inf, shape=(n, 2), dtype=np.float32),
            "pressure": spaces.Box(low=-np.inf, high=np.inf, shape=(n,), dtype=np.float32),
        })

    # ---- domain ---------------------------------------------------------------------------
    def _get_domain(self) -> Domain:
        edges = [np.linspace(0.0, self.L, self._x + 1), np.linspace(-self.H / 2, self.H / 2, self._y + 1)]
        coords = grids.vertex_grid(edges)
        dom = Domain(2, torch.tensor([self._nu]), passiveScalarChannels=0, name="ChannelDomain",
                     device=self._cuda_device, dtype=self._dtype, batch=self._num_envs)
        blk = dom.CreateBlock(vertexCoordinates=coords, name="ChannelBlock")
        blk.CloseBoundary("-x")
        blk.CloseBoundary("-y")
        dom.PrepareSolve()
        return dom

    def _additional_initialization(self) -> None:
        dev = self._cuda_device
        self._block = self._domain.getBlock(0)
        prof = torch.from_numpy(inflow_profile(self.H, self._y).astype(np.float32)).to(dev) * self._U_mean
        self._inflow = torch.zeros(1, 2, self._y, 1, device=dev)
        self._inflow[0, 0, :, 0] = prof
        self._block.getBoundary("-x").setVelocity(self._inflow)
        self._outflow = self._block.getBoundary("+x")
        jp = torch.from_numpy(jet_profile(self._jet_cells).astype(np.float32)).to(dev) * self._jet_max
        self._jet_shape = torch.zeros(1, 2, 1, self._x, device=dev)
        self._jet_shape[0, 1, 0, self._jet_start: self._jet_start + self._jet_cells] = jp
        self._velm = np.array([self._U_mean, 0.0], dtype=np.float32)  # host: characteristic outflow velocity
        # sensor probes: nearest cell centre of a regular lattice in the downstream 3/4 of the channel
        ix = np.linspace(self._x // 4, self._x - 1, self._n_sensors_x).round().astype(np.int64)
        iy = np.linspace(0, self._y - 1, self._n_sensors_y + 2).round().astype(np.int64)[1:-1]
        flat = (iy[None, :] * self._x + ix[:, None]).reshape(-1)
        self._sensor_idx = torch.from_numpy(flat).to(dev)
        self._hy = float(self.H / self._y)
        # the glue either side of the sim steps as two native launches (csrc/fg_envglue.hip) instead of ~25 elementwise torch launches;
        # FLUIDGYM_AMD_ENV_GLUE=0 keeps the torch expressions below (the A/B switch; they are also what a CPU test stub runs)
        self._native_glue = (dev.type == "cuda" and self._dtype == torch.float32 and self._x % 4 == 0
                             and os.environ.get("FLUIDGYM_AMD_ENV_GLUE", "1") != "0")
        self._decay = None

    def _get_simulation(self, domain: Domain, prep_fn: Dict[str, Any]) -> Simulation:
        return Simulation(
            domain=domain, prep_fn=prep_fn, substeps="ADAPTIVE", adaptive_CFL=self._adaptive_cfl, dt=self._dt,
            corrector_steps=2, pressure_tol=1e-5, advect_non_ortho_steps=1, pressure_non_ortho_steps=1,
            pressure_return_best_result=True, velocity_corrector="FD", non_orthogonal=True,
            solver_double_fallback=True,
            # advective outflow + flux re-balancing every substep (cylinder_env_base.py:280-300)
            outflow=([self._block_outflow(domain)], self._velm_host(), 1e-5),
        )

    @staticmethod
    def _block_outflow(domain):
        return domain.getBlock(0).getBoundary("+x")

    def _velm_host(self):
        return np.array([self._U_mean, 0.0], dtype=np.float32)

    def _fill_initial_fields(self) -> None:
        """Developed-profile initial state: inflow profile everywhere, zero pressure, outflow = inflow."""
        u0 = self._inflow.expand(self._num_envs, 2, self._y, self._x)
        self._block.setVelocity(u0)
        self._block.pressure.zero_()
        self._outflow.setVelocity(self._inflow)
        self._domain.solver.reset_solver_state()
        self._current_action = torch.zeros(self._num_envs, 1, device=self._cuda_device)
        self._apply_action(self._current_action)

    def _randomize_domain(self) -> None:
        """Divergence-free-projected noise on top of the profile + a random number of warm-up steps
        (the cylinder env adds noise and 0..n warm-up steps, cylinder_env_base.py:364-404)."""
        u = self._block.velocity
        noise = torch.randn(u.shape, device=u.device, generator=self._torch_rng_cuda) * 0.05
        u.add_(noise)
        self._sim.make_divergence_free()
        for _ in range(int(self._np_rng.integers(0, 5))):
            self._sim.single_step()

    # ---- control --------------------------------------------------------------------------
    def _apply_action(self, action: torch.Tensor) -> None:
        a = action.reshape(self._num_envs, 1, 1, 1)
        jets = self._jet_shape * a  # [B,2,1,X]; same wall-normal velocity on both walls => zero net flux
        self._block.getBoundary("-y").setVelocity(jets)
        self._block.getBoundary("+y").setVelocity(jets)

    def _get_global_obs(self):
        B = self._num_envs
        u = self._block.velocity.reshape(B, 2, -1).index_select(2, self._sensor_idx).permute(0, 2, 1).contiguous()
        p = self._block.pressure.reshape(B, -1).index_select(1, self._sensor_idx)
        return {"velocity": u, "pressure": p}

    def _metrics_now(self):
        u = self._block.velocity
        cross = (u[:, 1] ** 2).mean(dim=(1, 2))
        shear = self._nu * (u[:, 0, 0, :].mean(dim=1) + u[:, 0, -1, :].mean(dim=1)) / (0.5 * self._hy)
        return cross, shear

    def _step_native(self, action: torch.Tensor):
        """``_step_impl`` with the schedule and the observation by ``fg_envglue_*`` (same values: the schedule to the bit, the means to
        fp32 rounding of a different summation order)."""
        from .. import _lib as L
        B, n, X, S = self._num_envs, self._n_sim_steps, self._x, int(self._sensor_idx.numel())
        dev, lib = self._cuda_device, L.load()
        stream = torch.cuda.current_stream(dev).cuda_stream
        if self._enable_actions:
            if self._decay is None or self._decay.numel() != n:
                self._decay = ((1.0 - self._action_smoothing_alpha) ** torch.arange(1, n + 1, device=dev, dtype=torch.float32)).contiguous()
                self._jet_rows = self._jet_shape.reshape(2, X).contiguous()
            target = action.reshape(B).contiguous()
            jets = torch.empty(n, 2, B, 2, 1, X, device=dev)
            last = torch.empty(B, 1, device=dev)
            L.check(lib.fg_envglue_jet_schedule(target.data_ptr(), self._current_action.contiguous().data_ptr(), self._decay.data_ptr(),
                                                self._jet_rows.data_ptr(), n, B, X, jets.data_ptr(), last.data_ptr(), stream))
            self._jets = jets
        if not self._sim.multi_step(n, {2: jets[:, 0], 3: jets[:, 1]} if self._enable_actions else None):
            raise RuntimeError("simulation step failed")
        if self._enable_actions:
            self._current_action = last
        u, p = self._block.velocity, self._block.pressure
        obs_u, obs_p = torch.empty(B, S, 2, device=dev), torch.empty(B, S, device=dev)
        out = torch.empty(3, B, device=dev)
        L.check(lib.fg_envglue_channel_observe(u.data_ptr(), p.data_ptr(), self._sensor_idx.data_ptr(), S, B, self._y, X,
                                               float(self._nu / (0.5 * self._hy)), float(self._lift_penalty), obs_u.data_ptr(),
                                               obs_p.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), stream))
        return {"velocity": obs_u, "pressure": obs_p}, out[2], False, {"cross_flow_energy": out[0], "wall_shear": out[1]}

    def _step_impl(self, action: torch.Tensor):
        if self._native_glue and self._block.velocity.is_contiguous() and self._block.pressure.is_contiguous():
            return self._step_native(action)
        target = action.reshape(self._num_envs, 1)
        n = self._n_sim_steps
        if self._enable_actions:
            # action smoothing per sim step (cylinder_env_base.py:748-753): a_k = a_{k-1} + alpha (target - a_{k-1}), i.e.
            # a_k = target + (a_0 - target) (1 - alpha)^k -- all n controls and jet slabs of this env step in three launches
            # instead of six per sim step (same values to rounding; the per-step loop only copies them in)
            decay = (1.0 - self._action_smoothing_alpha) ** torch.arange(1, n + 1, device=target.device, dtype=target.dtype)
            controls = target[None] + (self._current_action - target)[None] * decay.view(n, 1, 1)       # [n, B, 1]
            # [n, wall, B, 2, 1, X]: each wall gets its OWN slice (the reference keeps two independent boundary tensors; an in-place
            # write to one wall -- setVelocity, a loaded state -- must not reach the other)
            jets = (self._jet_shape[None] * controls.reshape(n, self._num_envs, 1, 1, 1)).unsqueeze(1).repeat(1, 2, 1, 1, 1, 1)
            self._jets = jets          # (the walls stay bound to their last slices after the step)
        # the n sim steps of this env step in one native call (fg_multi_step): before sim step k the walls are BOUND to that step's
        # slices (the same wall-normal velocity on both: zero net flux) -- a pointer update, no copy launch
        if not self._sim.multi_step(n, {2: jets[:, 0], 3: jets[:, 1]} if self._enable_actions else None):
            raise RuntimeError("simulation step failed")
        if self._enable_actions:
            self._current_action = controls[n - 1]
        cross, shear = self._metrics_now()
        obs = self._get_global_obs()
        reward = -(shear + self._lift_penalty * cross)
        return obs, reward, False, {"cross_flow_energy": cross, "wall_shear": shear}

    def _get_extra_state(self):
        return None if self._current_action is None else self._current_action.clone()

    def _set_extra_state(self, extra):
        if extra is not None:
            self._current_action = extra.clone()

    @property
    def id(self) -> str:
        return f"ChannelJet2D_Re{self._reynolds_number}_{self._x}x{self._y}"
