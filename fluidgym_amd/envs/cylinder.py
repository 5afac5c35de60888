"""Cylinder (von Karman vortex street) environments on the multi-block HIP path.

Mirrors ``envs/cylinder/cylinder_env_base.py`` (CylinderEnvBase), ``jet_cylinder_env_2d.py`` (CylinderJetEnv2D) and
``rotating_cylinder_env_2d.py`` (CylinderRotEnv2D) of the reference: the same five-block mesh (``cylinder_grid.py``,
pinned on the reference's vertex coordinates), solver settings (:303-329), convective outflow hook (:276-300), 151
velocity / pressure sensors read from the uniformly resampled fields (:430-518), drag / lift by wall-stress integration
(:616-700, ``forces.py``), action smoothing and reward (:741-776).  Batched over ``num_envs`` like every env here.

Not carried over: the published initial domains (HuggingFace ``fluidgym-data``; no network) -- ``reset`` starts from a
projected uniform stream and runs ``initial_domain_steps`` developed-flow steps (the reference generates its initial
domains with 400, :138) -- and the domain statistics (``_cd_ref`` is 0 unless ``drag_reference`` is given).

``CylinderJetEnv3D`` (``jet_cylinder_env_3d.py``) runs the same mesh extruded over the span (z-periodic): ``n_jets``
spanwise jet segments, 151 x ``n_jets * 2`` sensors read from the 3-D resampled fields, drag / lift per spanwise layer,
single-agent and multi-agent (one agent per segment, windows of neighbouring segments) interfaces.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import torch

from ..simulation.multiblock import MultiBlockDomain, MultiBlockSimulation
from ..simulation.resample_mb import MultiBlockResampler, MultiBlockResampler3D
from .. import spaces
from .channel import jet_profile
from .cylinder_grid import BOTTOM, LEFT, RIGHT, TOP, build_domain, extrude_mesh, make_vortex_street_mesh
from .fluid_env import FluidEnv
from .forces import WallRing

CYLINDER_JET_2D_DEFAULT_CONFIG = {
    "reynolds_number": 1e2, "resolution": 24, "dt": 1e-2, "adaptive_cfl": 0.8, "step_length": 0.25,
    "episode_length": 80, "lift_penalty": 1.0, "use_marl": False, "dtype": torch.float32,
    "load_initial_domain": True, "load_domain_statistics": True, "randomize_initial_state": True,
    "enable_actions": True, "differentiable": False,
}
CYLINDER_ROT_2D_DEFAULT_CONFIG = dict(CYLINDER_JET_2D_DEFAULT_CONFIG)


ONCHIP_PCG_MAX_CELLS = 24 * 1024   # the multilevel-preconditioned on-chip CG (2-D): fg_mb_onchip.hip OC_L2_CELLS


class CylinderEnvBase(FluidEnv):
    _supports_marl = False
    _action_smoothing_alpha: float = 0.1
    H: float = 4.1
    D: float = 4.0  # span of the 3-D variants
    L: float = 22.0
    cylinder_diameter: float = 1.0
    _U_mean: float = 1.0
    cylinder_offset_y: float = 0.05
    _n_sensors_x_y: int = 151
    _vortex_street_refinement_base: float = 0.95
    _metrics = ["drag", "lift"]
    _initial_domain_steps = 400

    def __init__(self, reynolds_number: float, resolution: int, dt: float, adaptive_cfl: float, step_length: float,
                 episode_length: int, lift_penalty: float = 1.0, ndims: int = 2, initial_domain_steps: Optional[int] = None,
                 drag_reference: float = 0.0, pressure_use_BiCG=None, pressure_deflation: bool = False,
                 non_ortho_mode: str = "matrix", **kw):
        if ndims not in (2, 3):
            raise ValueError("ndims must be 2 or 3")
        self._reynolds_number = reynolds_number
        self._circle_resolution_angular = int(resolution)
        self._lift_penalty = lift_penalty
        self._nu = self._U_mean / reynolds_number
        self._cd_ref = float(drag_reference)
        self._pressure_use_bicg = pressure_use_BiCG
        self._pressure_deflation = pressure_deflation
        if non_ortho_mode not in ("matrix", "rhs"):
            raise ValueError("non_ortho_mode: 'matrix' (the reference's nonOrthoFlags) or 'rhs' (every cross-metric term lagged)")
        self._non_ortho_flags = 25 if non_ortho_mode == "matrix" else 10
        if initial_domain_steps is not None:
            self._initial_domain_steps = int(initial_domain_steps)
        super().__init__(dt=dt, adaptive_cfl=adaptive_cfl, step_length=step_length, episode_length=episode_length,
                         ndims=ndims, **kw)
        self._last_control = None
        self._sensor_locations = self._get_sensor_locations()

    # ---- spaces (cylinder_env_base.py:183-213)
    def _get_action_space(self):
        return spaces.Box(low=-1.0, high=1.0, shape=(1,), dtype=np.float32)

    def _get_observation_space(self):
        n = self._n_sensors_x_y
        return spaces.Dict({
            "velocity": spaces.Box(low=-np.inf, high=np.inf, shape=(n, 2), dtype=np.float32),
            "pressure": spaces.Box(low=-np.inf, high=np.inf, shape=(n,), dtype=np.float32),
        })

    @property
    def render_shape(self):
        z_res = self._circle_resolution_angular * 4
        return (int(z_res / self.H * self.L), z_res, z_res)

    @property
    def id(self) -> str:
        return f"{type(self).__name__}_Re{self._reynolds_number}"

    @property
    def initial_domain_id(self) -> str:
        return f"cylinder_{self._ndims}D_Re{int(self._reynolds_number)}_Res{self._circle_resolution_angular}"

    # ---- sensors (cylinder_env_base.py:430-518)
    def _get_sensor_locations_2d(self) -> np.ndarray:
        xs, ys = np.arange(1.0, 5.0, 0.5), np.arange(-1.5, 1.75, 0.5)
        gx, gy = np.meshgrid(xs, ys, indexing="ij")
        main = np.stack([gx.ravel(), gy.ravel()])
        x1 = np.arange(-0.25, 1, 0.25)
        x2 = np.concatenate([[-0.25], np.arange(0.25, 1.25, 0.25)])
        x3 = np.array([0.75] * 3)
        extra = np.stack([np.concatenate([x1, x1, x2, x2, x3]),
                          np.concatenate([np.full_like(x1, -1.5), np.full_like(x1, 1.5), np.full_like(x2, self.cylinder_diameter),
                                          np.full_like(x2, -self.cylinder_diameter), np.array([-0.5, 0, 0.5])])])
        ang = np.linspace(0, 2 * np.pi, 36).astype(np.float32)  # float32 like the reference's torch.linspace: the sensor at
        c1 = np.float32(1.0) * np.stack([np.cos(ang), np.sin(ang)])   # angle 2 pi sits on a rounding tie of the pixel grid
        c2 = np.float32(0.625) * np.stack([np.cos(ang), np.sin(ang)])
        return np.concatenate([main.astype(np.float32), c1, c2, extra.astype(np.float32)], axis=1).astype(np.float32)

    def _get_sensor_locations(self) -> np.ndarray:
        """Pixel (x, y) of every sensor in the resampled field (``_sensor_locations_to_grid_coords``)."""
        p = self._get_sensor_locations_2d().copy()
        p[0] = (p[0] + np.float32(2.0)) * np.float32((self.render_shape[0] - 1) / (self.L - 2.0))
        p[1] = (p[1] + np.float32(self.H / 2)) * np.float32((self.render_shape[1] - 1) / self.H)
        return np.round(p).astype(np.int64)

    # ---- domain and simulation (cylinder_env_base.py:233-329)
    def _get_domain(self) -> MultiBlockDomain:
        self._mesh = make_vortex_street_mesh(self._circle_resolution_angular, self.H, self.L, self.cylinder_diameter / 2,
                                             self.cylinder_offset_y, self.cylinder_diameter / 2, self.cylinder_diameter,
                                             self._vortex_street_refinement_base)
        if self._ndims == 3:   # grid.py:281-294: res_z = resolution, z in [-2, 2]
            self._mesh = extrude_mesh(self._mesh, self._circle_resolution_angular, -self.D / 2, self.D / 2)
        # dtype=torch.float64: the fp64 build of the multi-block path (plain recurrences, one cell per thread; observations are
        # resampled in float32 and handed back in the env's dtype)
        return build_domain(self._mesh, self._nu, batch=self._num_envs, device=self._cuda_device,
                            non_ortho_flags=self._non_ortho_flags, dtype=self._dtype)

    def _get_simulation(self, domain, prep_fn):
        # solver policy pressure_bicgstab_large_meshes: beyond the preconditioned on-chip CG's reach (2-D meshes of up to 24 576 cells:
        # every 2-D id; csrc/fg_mb_onchip.hip k_mbc_onchip / k_mbc_l2) the pressure systems go to the fp64-refined BiCGStab unless
        # the caller chose (policy.py)
        from ..simulation.policy import get_solver_policy
        if self._pressure_use_bicg is None:     # (None = not chosen by the caller; False = the reference's CG, 1 / 2 = BiCGStab / refined)
            self._pressure_use_bicg = 2 if (get_solver_policy()["pressure_bicgstab_large_meshes"]
                                            and (self._ndims == 3 or domain.n_cells > ONCHIP_PCG_MAX_CELLS)) else False
        sim = MultiBlockSimulation(domain, dt=self._dt, adaptive_CFL=self._adaptive_cfl, substeps="ADAPTIVE", corrector_steps=2,
                                   pressure_tol=1e-5 if self._ndims == 2 else 5e-7, advect_non_ortho_steps=1,
                                   pressure_non_ortho_steps=1 if self._ndims == 2 else 4,
                                   pressure_use_BiCG=self._pressure_use_bicg, outflow=self._mesh.outflow,
                                   outflow_velocity=(self._U_mean, 0.0, 0.0), outflow_tol=5e-6,
                                   solver_double_fallback=True, BiCG_precondition_fallback=True)   # cylinder_env_base.py:326-328
        return sim

    def _additional_initialization(self) -> None:
        dom = self._domain
        self._ring = WallRing(dom, [(LEFT, "+x", False), (TOP, "-y", False), (RIGHT, "-x", True), (BOTTOM, "+y", True)])
        if self._ndims == 3:
            self._resampler = MultiBlockResampler3D(self._mesh.coords, self.render_shape, fill_max_steps=16, device=dom.device)
            self._sensors = self._resampler.sensor_gather(self._sensor_locations.reshape(3, -1).T)
        else:
            self._resampler = MultiBlockResampler(self._mesh.coords, self.render_shape[:2], fill_max_steps=16, device=dom.device)
            self._sensors = self._resampler.sensor_gather(self._sensor_locations.T)
        self._deflation_cos = dom.set_pressure_deflation() if self._pressure_deflation else 1.0
        from ..simulation.policy import get_solver_policy
        self._multilevel = dom.set_pressure_multilevel() if (get_solver_policy()["pressure_multilevel"] and not self._pressure_deflation
                                                              and not self._pressure_use_bicg) else None
        self._initial_boundary = dom.boundary_velocity.clone()  # inflow / outflow profile, walls at rest
        self._last_control = torch.zeros(self._num_envs, self._n_controls, device=dom.device, dtype=self._dtype)

    _n_controls = 1

    def _fill_initial_fields(self) -> None:
        """Uniform stream projected onto the mesh, then ``initial_domain_steps`` uncontrolled steps (the reference ships
        states generated with 400 steps, cylinder_env_base.py:138; they are cached per env instance here)."""
        dom = self._domain
        if getattr(self, "_developed", None) is None:
            dom.boundary_velocity.copy_(self._initial_boundary)
            dom.velocity.zero_()
            dom.velocity[:, 0] = self._U_mean
            dom.pressure.zero_()
            self._sim.make_divergence_free()
            for _ in range(self._initial_domain_steps):
                self._sim.single_step()
            self._developed = dom.Clone()
        dom.Restore(self._developed)
        self._last_control = torch.zeros(self._num_envs, self._n_controls, device=dom.device, dtype=self._dtype)

    def _randomize_domain(self) -> None:
        """cylinder_env_base.py:364-404."""
        period = 1.0 / (0.3 * self._U_mean / self.cylinder_diameter)
        max_n = 2 * int(period / self._step_length) - 1
        n_steps = int(self._np_rng.integers(int(0.5 * max_n), max_n)) + 1
        dom = self._domain
        g = self._torch_rng_cuda
        dom.velocity.add_(torch.randn(dom.velocity.shape, device=dom.device, generator=g) * 0.025)
        dom.pressure.add_(torch.randn(dom.pressure.shape, device=dom.device, generator=g) * 0.025)
        for _ in range(n_steps):
            self._sim.single_step()

    # ---- observations, forces, step (cylinder_env_base.py:541-776)
    def _get_global_obs(self) -> Dict[str, torch.Tensor]:
        dom = self._domain
        u = self._sensors(dom.velocity).to(self._dtype)   # [B, 2, S]
        p = self._sensors(dom.pressure).to(self._dtype)      # [B, S]
        return {"velocity": u.permute(0, 2, 1).contiguous(), "pressure": p}

    def get_velocity(self) -> torch.Tensor:
        """Resampled velocity [B, 2, y, x] (``FluidEnv.get_velocity``)."""
        return self._resampler(self._domain.velocity)

    def get_pressure(self) -> torch.Tensor:
        return self._resampler(self._domain.pressure)

    def _get_drag_and_lift(self):
        """[B] in 2-D; [B, NZ] per spanwise layer in 3-D (face area = edge length x D / resolution, :676-689)."""
        f = self._ring.forces(self._domain, self._nu, layer_height=self.D / self._circle_resolution_angular)
        norm = 0.5 * self._U_mean ** 2 * self.cylinder_diameter
        return f[:, 0] / norm, f[:, 1] / norm

    def _apply_action(self, action: torch.Tensor) -> None:
        raise NotImplementedError

    def _step_impl(self, action: torch.Tensor):
        target = action.reshape(self._num_envs, self._n_controls)
        n = self._n_sim_steps
        # The smoothed control of every sim step (cylinder_env_base.py:560-566: c <- c + alpha (target - c)) is a recurrence on a
        # [B, n_controls] tensor: evaluated once on the host in the same fp32 operations (separately rounded multiply and add,
        # as the per-step tensor expression) and uploaded in one copy, instead of three tiny launches per sim step.
        # The host keeps its own copy of the last control (valid while `_last_control` is the tensor this method left there), so the
        # only read-back of a step is the action itself -- none when the caller hands over a host tensor.
        mirror = getattr(self, "_last_control_mirror", None)
        c = mirror[1] if mirror is not None and mirror[0] is self._last_control else self._last_control.cpu()
        t_host, alpha, controls = target.cpu(), self._action_smoothing_alpha, []
        for _ in range(n):
            c = c + alpha * (t_host - c)
            controls.append(c)
        c_last = c
        controls = torch.stack(controls).to(self._last_control.device, non_blocking=False)      # [n, B, n_controls]
        # raw wall forces of every sim step land in one buffer; normalised and averaged once (elementwise division, then the
        # mean over the stack: what torch.stack of the per-step coefficients gave)
        raw = torch.empty(n, self._num_envs, 2, self._ring.nz, dtype=self._dtype, device=self._last_control.device)
        for k in range(n):
            self._last_control = controls[k]
            if self._enable_actions:
                self._apply_action(controls[k])
            self._sim.single_step()
            self._ring.forces(self._domain, self._nu, layer_height=self.D / self._circle_resolution_angular, out=raw[k])
        self._last_control_mirror = (self._last_control, c_last)
        obs = self._get_global_obs()
        coeff = raw / (0.5 * self._U_mean ** 2 * self.cylinder_diameter)
        if self._ndims == 2:
            coeff = coeff[..., 0]
        # mean over the sim steps as a loop of adds in step order: `mean(0)` picks its summation order per column, and identical
        # envs of a batch then report drag / lift (hence rewards) that differ in the last bit
        acc = coeff[0].clone()
        for k in range(1, n):
            acc += coeff[k]
        acc /= n
        cd, cl = acc[:, 0], acc[:, 1]
        if self._ndims == 3:   # summed over the span here; CylinderJetEnv3D divides by D (:765-768)
            reward = self._cd_ref - cd.sum(-1) - self._lift_penalty * cl.sum(-1).abs()
        else:
            reward = self._cd_ref - cd - self._lift_penalty * cl.abs()
        return obs, reward, False, {"drag": cd, "lift": cl}

    # ---- on-disk initial domains in the reference's format (fluid_env.py:1044-1112)
    def _save_initial_domain(self, mode, idx: int, env: int = 0) -> None:
        from ..simulation.domain_io import save_multiblock_domain

        out_dir = self._get_domain_dir(idx)
        out_dir.mkdir(parents=True, exist_ok=True)
        save_multiblock_domain(self._domain, str(out_dir / mode.value), env=env, name="CylinderDomain")

    def load_initial_domain(self, idx: int, mode=None) -> None:
        from pathlib import Path

        from ..simulation.domain_io import load_multiblock_domain

        mode = self._mode if mode is None else mode
        path = self._get_domain_dir(idx) / mode.value
        if not Path(str(path) + ".json").exists():
            return super().load_initial_domain(idx, mode)
        self._pinned_initial = True      # (see FluidEnv.load_initial_domain)
        if self._domain is None:
            self._set_initial_state(randomize=False)
        loaded = load_multiblock_domain(str(path), device=self._cuda_device, batch=self._num_envs)
        try:
            if loaded.n_cells != self._domain.n_cells or [b.size for b in loaded.blocks] != [b.size for b in self._domain.blocks]:
                raise ValueError(f"initial domain {path} does not match this environment's mesh")
            snap = loaded.Clone()
        finally:
            loaded.close()
        self._loaded_initial = snap
        self._domain.Restore(snap)

    def _get_extra_state(self):
        return {"last_control": self._last_control.clone()}

    def _set_extra_state(self, extra) -> None:
        if extra is not None:
            self._last_control = extra["last_control"].clone()

    def render(self, *a, **kw) -> np.ndarray:
        speed = torch.linalg.vector_norm(self.get_velocity()[0], dim=0)
        if self._ndims == 3:
            speed = speed[speed.shape[0] // 2]      # mid-span slice
        return speed.detach().cpu().numpy()


def _face_vertices(mesh, block: int, face: str) -> np.ndarray:
    c = mesh.coords[block].astype(np.float64)
    if c.shape[0] == 3:
        c = c[:2, 0]          # the first spanwise layer ("we set z = 0", jet_cylinder_env_3d.py:378-381)
    return {"+x": c[:, :, -1], "-x": c[:, :, 0], "-y": c[:, 0, :], "+y": c[:, -1, :]}[face]


class CylinderJetEnv2D(CylinderEnvBase):
    """Two synthetic jets at the poles of the cylinder, blowing / sucking with zero net mass flux
    (jet_cylinder_env_2d.py:124-186)."""

    _jet_angle: float = 10.0  # degrees

    def _jet_velocities(self, boundary_vertices: np.ndarray, top: bool) -> np.ndarray:
        centers = 0.5 * (boundary_vertices[:, :-1] + boundary_vertices[:, 1:])
        base = np.pi / 2 if top else -np.pi / 2
        deg = np.rad2deg(base - np.arctan2(centers[1], centers[0]))
        mag = np.abs(deg)
        mag[mag > self._jet_angle] = 0.0
        nz = np.nonzero(mag > 0.0)[0]
        lo, hi = nz[0] - 1, nz[-1] + 1
        prof = jet_profile(int(hi - lo + 1))
        vel = np.zeros_like(centers)
        for i, u in zip(range(lo, hi + 1), prof):
            vel[0, i] = u * np.sin(np.deg2rad(deg[i]))
            vel[1, i] = u * np.cos(np.deg2rad(deg[i]))
        return vel.astype(np.float32)

    def _additional_initialization(self) -> None:
        super()._additional_initialization()
        dev = self._domain.device
        self._top_velocity = torch.as_tensor(self._jet_velocities(_face_vertices(self._mesh, TOP, "-y"), True), device=dev).to(self._dtype)
        self._bottom_velocity = torch.as_tensor(self._jet_velocities(_face_vertices(self._mesh, BOTTOM, "+y"), False), device=dev).to(self._dtype)

    def _apply_action(self, action: torch.Tensor) -> None:
        a = action.reshape(self._num_envs, 1, 1)
        dom = self._domain
        # written straight into the boundary slots (one launch per jet instead of a product and a copy)
        torch.mul(self._top_velocity[None], a, out=dom.blocks[TOP].boundary("-y"))
        torch.mul(self._bottom_velocity[None], a, out=dom.blocks[BOTTOM].boundary("+y"))


def rotating_wall_velocities(mesh):
    """Unit tangential velocity on the cylinder-wall faces of the four blocks around it, ``[(block, face, [2, n_faces])]``
    (rotating_cylinder_env_2d.py:131-164)."""
    out = []
    for b, face in ((LEFT, "+x"), (TOP, "-y"), (RIGHT, "-x"), (BOTTOM, "+y")):
        v = _face_vertices(mesh, b, face)
        ctr = 0.5 * (v[:, :-1] + v[:, 1:])
        th = np.arctan2(ctr[1], ctr[0])
        out.append((b, face, np.stack([np.sin(th), -np.cos(th)]).astype(np.float32)))
    return out


class CylinderRotEnv2D(CylinderEnvBase):
    """The cylinder wall rotates with the commanded speed (rotating_cylinder_env_2d.py:121-176)."""

    def _additional_initialization(self) -> None:
        super()._additional_initialization()
        dev = self._domain.device
        self._wall = [(b, face, torch.as_tensor(v, device=dev).to(self._dtype)) for b, face, v in rotating_wall_velocities(self._mesh)]

    def _apply_action(self, action: torch.Tensor) -> None:
        a = action.reshape(self._num_envs, 1, 1)
        for b, face, vel in self._wall:
            torch.mul(vel[None], a, out=self._domain.blocks[b].boundary(face))


CYLINDER_JET_3D_DEFAULT_CONFIG = {
    "n_jets": 8, "reynolds_number": 1e2, "resolution": 24, "dt": 1e-2, "adaptive_cfl": 0.8, "step_length": 0.25,
    "lift_penalty": 1.0, "episode_length": 80, "local_obs_window": 3, "local_reward_weight": 0.8, "local_2d_obs": False,
    "use_marl": False, "dtype": torch.float32, "load_initial_domain": True, "load_domain_statistics": True,
    "randomize_initial_state": True, "enable_actions": True, "differentiable": False,
}


class CylinderJetEnv3D(CylinderJetEnv2D):
    """``CylinderJetEnv3D`` (jet_cylinder_env_3d.py:44-480): the cylinder spans z in [-2, 2] (periodic), the two jet slots
    are cut into ``n_jets`` spanwise segments, each driven by one action (single agent: all of them; multi-agent: one
    agent per segment, observing ``local_obs_window`` neighbouring segments).

    Observation layout: the reference gathers the sensors as ``[z, sensor, component]`` and then *views* that memory as
    ``[z, component, sensor]`` (obs_extraction.py:127-135, ``view`` where a ``permute`` was meant), so the velocity
    observation interleaves sensors and components.  Policies trained on the reference see that layout; it is kept
    (pinned on the recorded reference output, tests/test_cylinder_grid.py)."""

    _supports_marl = True
    _n_sensors_per_agent: int = 2

    def __init__(self, n_jets: int, reynolds_number: float, resolution: int, dt: float, adaptive_cfl: float,
                 step_length: float, episode_length: int, lift_penalty: float, local_obs_window: int, use_marl: bool,
                 local_reward_weight: Optional[float], local_2d_obs: bool = False, **kw):
        if n_jets < 1 or resolution % n_jets != 0:
            raise ValueError("n_agents must be a positive integer that evenly divides circle_resolution_angular.")
        if local_2d_obs and not use_marl:
            raise ValueError("Local 2D observations are only supported in multi-agent mode.")
        self._local_2d_obs = bool(local_2d_obs)
        self._n_jets = int(n_jets)
        self._n_controls = self._n_jets
        self._local_obs_window = int(local_obs_window)
        self._local_reward_weight = local_reward_weight
        if local_2d_obs:
            self._n_sensors_per_agent = 1
            self._local_obs_window = 1
        kw.pop("ndims", None)
        super().__init__(reynolds_number=reynolds_number, resolution=resolution, dt=dt, adaptive_cfl=adaptive_cfl,
                         step_length=step_length, episode_length=episode_length, lift_penalty=lift_penalty, ndims=3,
                         use_marl=use_marl, **kw)

    # ---- spaces (jet_cylinder_env_3d.py:188-258)
    def _get_action_space(self):
        shape = (1,) if self._use_marl else (self._n_jets, 1)
        return spaces.Box(low=-1.0, high=1.0, shape=shape, dtype=np.float32)

    def _get_observation_space(self):
        n, spa = self._n_sensors_x_y, self._n_sensors_per_agent
        if self._use_marl and self._local_2d_obs:
            v_shape, p_shape = (n, 2), (n,)
        else:
            lead = self._local_obs_window if self._use_marl else self._n_jets
            v_shape, p_shape = (lead, spa, 3, n), (lead, spa, n)
        return spaces.Dict({
            "velocity": spaces.Box(low=-np.inf, high=np.inf, shape=v_shape, dtype=np.float32),
            "pressure": spaces.Box(low=-np.inf, high=np.inf, shape=p_shape, dtype=np.float32),
        })

    @property
    def n_agents(self) -> int:
        return self._n_jets if self._use_marl else 1

    @property
    def _n_sensors_z(self) -> int:
        return self._n_jets * self._n_sensors_per_agent

    @property
    def id(self) -> str:
        return f"JetCylinder3D_Re{self._reynolds_number}"

    # ---- sensors (jet_cylinder_env_3d.py:277-304; cylinder_env_base.py:436-449)
    def _get_sensor_locations(self) -> np.ndarray:
        """Pixel (x, y, z) of every sensor, [3, n_sensors_z, 151].  The spanwise positions are spread over H (not D) and
        scaled with the y resolution, as in the reference."""
        xy = super()._get_sensor_locations()                                     # [2, 151]
        nz = self._n_sensors_z
        z = np.linspace(-self.H / 2, self.H / 2, nz + 1, dtype=np.float32)[:-1] + np.float32(self.H / (2 * nz))
        pz = (z.astype(np.float32) + np.float32(self.H / 2)) * np.float32((self.render_shape[1] - 1) / self.H)
        pz = np.round(pz).astype(np.int64)
        out = np.empty((3, nz, xy.shape[1]), np.int64)
        out[0], out[1], out[2] = xy[0][None, :], xy[1][None, :], pz[:, None]
        return out

    # ---- observations (obs_extraction.py:60-151; jet_cylinder_env_3d.py:306-339)
    def _get_global_obs(self) -> Dict[str, torch.Tensor]:
        dom = self._domain
        B, nz, n = self._num_envs, self._n_sensors_z, self._n_sensors_x_y
        u = self._sensors(dom.velocity).to(self._dtype)       # [B, 3, nz * 151]
        p = self._sensors(dom.pressure).to(self._dtype)          # [B, nz * 151]
        if self._local_2d_obs:
            u = u[:, :2]
        vd = u.shape[1]
        u = u.permute(0, 2, 1).contiguous()                                       # [B, nz * 151, vd], as gathered there
        u = u.reshape(B, nz, vd, n).reshape(B, self._n_jets, self._n_sensors_per_agent, vd, n)   # the reference's view
        if self._local_2d_obs:
            u = u.permute(0, 1, 2, 4, 3)
        return {"velocity": u, "pressure": p.reshape(B, self._n_jets, self._n_sensors_per_agent, n)}

    def _get_local_obs(self) -> Dict[str, torch.Tensor]:
        """Agent a sees the segments a - w//2 ... a + w//2 (periodic): [B, n_agents, window, ...]."""
        g = self._get_global_obs()
        W, nj = self._local_obs_window, self._n_jets
        seg = (torch.arange(nj)[:, None] + torch.arange(W)[None, :] - W // 2) % nj    # [agent, window]
        out = {}
        for k, v in g.items():
            win = v[:, seg.to(v.device)]                                           # [B, nj, W, ...]
            if self._local_2d_obs:
                win = win.reshape(self._num_envs, nj, *win.shape[4:])            # window 1, one sensor layer: squeeze
            out[k] = win
        return out

    # ---- actuation (jet_cylinder_env_3d.py:341-421)
    def _additional_initialization(self) -> None:
        super()._additional_initialization()
        nz = self._circle_resolution_angular
        self._nz_per_agent = nz // self._n_jets
        z3 = lambda v: torch.cat([v, torch.zeros_like(v[:1])], dim=0)[:, None, :].expand(3, nz, v.shape[1])
        self._top_velocity3 = z3(self._top_velocity).contiguous()                 # [3, nz, nx]
        self._bottom_velocity3 = z3(self._bottom_velocity).contiguous()

    def _apply_action(self, action: torch.Tensor) -> None:
        """One amplitude per spanwise segment, repeated over the segment's layers.  (The reference then rebalances the
        boundary fluxes over both jet faces and the outflow with tol 1e-7; the jets are flux-neutral by construction and
        the outflow is rebalanced by every step's PRE hook here, so that call is not repeated.)"""
        B = self._num_envs
        a = action.reshape(B, self._n_jets).repeat_interleave(self._nz_per_agent, dim=1)[:, None, :, None]   # [B, 1, nz, 1]
        dom = self._domain
        dom.blocks[TOP].boundary("-y").copy_((self._top_velocity3[None] * a).reshape(B, 3, -1))
        dom.blocks[BOTTOM].boundary("+y").copy_((self._bottom_velocity3[None] * a).reshape(B, 3, -1))

    # ---- step (jet_cylinder_env_3d.py:427-480)
    def _step_impl(self, action: torch.Tensor):
        obs, _, term, info = super()._step_impl(action)
        all_cds, all_cls = info.pop("drag"), info.pop("lift")                     # [B, NZ]
        cd, cl = all_cds.sum(-1) / self.D, all_cls.sum(-1) / self.D
        reward = self._cd_ref - cd - self._lift_penalty * cl.abs()
        return obs, reward, term, {"drag": cd, "lift": cl, "all_cds": all_cds, "all_cls": all_cls}

    def _step_marl_impl(self, actions: torch.Tensor):
        if self._local_reward_weight is None:
            raise ValueError("local_reward_weight must be set for multi-agent step.")
        _, global_reward, terminated, info = self._step_impl(actions)
        local_obs = self._get_local_obs()
        all_cds, all_cls = info.pop("all_cds"), info.pop("all_cls")
        B, nj = self._num_envs, self._n_jets
        local_cd = all_cds.reshape(B, nj, -1).sum(-1) / (self.D / nj)
        local_cl = all_cls.reshape(B, nj, -1).sum(-1) / (self.D / nj)
        local_rewards = self._cd_ref - local_cd - self._lift_penalty * local_cl.abs()
        w = self._local_reward_weight
        agent_rewards = w * local_rewards + (1 - w) * global_reward[:, None]
        info["global_reward"] = global_reward
        return local_obs, agent_rewards, terminated, info

    def get_velocity(self) -> torch.Tensor:
        """Resampled velocity [B, 3, z, y, x]."""
        return self._resampler(self._domain.velocity)
