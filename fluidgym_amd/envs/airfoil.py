"""Airfoil (separation control) environment on the multi-block HIP path.

Mirrors ``envs/airfoil/airfoil_env_base.py`` (AirfoilEnvBase) and ``airfoil_env_2d.py`` (AirfoilEnv2D) of the reference:
NACA 0012 at an angle of attack in a channel (six-block C-mesh, ``airfoil_grid.py``), three synthetic jets on the suction
side with zero net mass flux (:463-513, airfoil_env_2d.py:173-190), the solver settings of :283-311 (two non-orthogonal
advection and four pressure iterations, tolerances 1e-6 / 1e-7), convective outflow over both tail faces (:258-281),
sensors in the wake and under the section with those inside the section dropped (:559-660), drag / lift by wall-stress
integration over the front / top / bottom faces (:334-461), reward = lift / drag - reference (:744-776).

Pressure solver: the reference asks for CG and relies on its fallback chain (fp64, then preconditioned BiCGStab,
``solver_double_fallback`` / ``BiCG_precondition_fallback``, :313-315) -- on this mesh the pressure matrix is 3.7 % non-symmetric
and CG stagnates at 2-4e-5 against the tolerance of 1e-7.  The default here is the path's BiCGStab with fp64 iterative
refinement (``pressure_use_BiCG=2``), which reaches the tolerance in ~30 iterations per warm-started solve;
``pressure_use_BiCG=False`` selects the stagnating CG (with ``stall_limit`` deciding where its solves are cut).

Batched over ``num_envs`` like every env here.  Not carried over: the published initial domains / statistics (no network:
``reset`` develops the flow from a projected uniform stream).  The section is the closed-form NACA 0012
(``airfoil_grid.naca0012_sharp``) unless a ``surface`` polyline is given; states saved by the reference carry their mesh.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .. import spaces
from ..simulation.multiblock import MultiBlockSimulation
from ..simulation.policy import get_solver_policy
from ..simulation.resample_mb import MultiBlockResampler, MultiBlockResampler3D
from .airfoil_grid import BOTTOM, FRONT, TAIL_LOWER, TAIL_UPPER, TOP, make_airfoil_mesh, naca0012_sharp
from .channel import jet_profile
from .cylinder import CylinderEnvBase
from .cylinder_grid import build_domain, extrude_mesh
from .fluid_env import FluidEnv
from .forces import WallRing

AIRFOIL_2D_DEFAULT_CONFIG = {
    "reynolds_number": 3e3, "dt": 0.05, "step_length": 0.25, "adaptive_cfl": 0.8, "episode_length": 300,
    "attack_angle_deg": 10.0, "use_marl": False, "dtype": torch.float32, "load_initial_domain": True,
    "load_domain_statistics": True, "randomize_initial_state": True, "enable_actions": True, "differentiable": False,
}
JET_CENTERS = (0.2, 0.4, 0.6)   # grid.py:14-15
JET_WIDTH = 0.08


def points_in_polygon(poly: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """Crossing test with the tie rules of the "crossings multiply" algorithm (Haines, Graphics Gems IV) that
    ``matplotlib.path.Path.contains_points`` applies -- the reference builds its mask with that call on integer pixel
    centres and an integer-rounded polygon, where points ON edges are common and the tie rule decides the mask.
    poly [n, 2] (closed implicitly), pts [m, 2]."""
    tx, ty = pts[:, 0], pts[:, 1]
    inside = np.zeros(len(pts), bool)
    x0, y0 = poly[-1]
    for x1, y1 in poly:
        f0, f1 = y0 >= ty, y1 >= ty
        hit = (f0 != f1) & (((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == f1)
        inside ^= hit
        x0, y0 = x1, y1
    return inside


def jet_locations(top_coords: np.ndarray):
    """Cell ranges [start, end] (inclusive) of the jets on the top block's wall face (grid.py:18-48): the surface vertices
    whose x is closest to centre -/+ half the jet width."""
    x = np.asarray(top_coords)[0, 0, :]
    return [[int(np.argmin(np.abs(x - np.float32(c - JET_WIDTH / 2)))), int(np.argmin(np.abs(x - np.float32(c + JET_WIDTH / 2))))]
            for c in JET_CENTERS]


class AirfoilEnvBase(CylinderEnvBase):
    _supports_marl = False
    _action_smoothing_alpha: float = 0.1
    _n_jets: int = 3
    U_mean: float = 0.3
    airfoil_length: float = 1.0
    H: float = 1.4
    L: float = 4.5
    D: float = 1.4
    _metrics = ["drag", "lift"]
    _initial_domain_steps = 400

    def __init__(self, ndims: int, reynolds_number: float, adaptive_cfl: float, step_length: float, episode_length: int,
                 dt: float, attack_angle_deg: float, initial_domain_steps: Optional[int] = None,
                 lift_drag_reference: float = 0.0, resolution_div: int = 1, surface: Optional[np.ndarray] = None,
                 pressure_use_BiCG=2, non_ortho_mode: str = "matrix", stall_limit: int = 400,
                 pressure_deflation: bool = False, debug: bool = False, **kw):
        if attack_angle_deg < 0.0 or attack_angle_deg > 20.0:
            raise ValueError("Attack angle must be between 0 and 20 degrees.")
        if ndims not in (2, 3):
            raise ValueError("ndims must be 2 or 3")
        self._ndims = ndims
        self._reynolds_number = reynolds_number
        self._nu = self.U_mean * self.airfoil_length / reynolds_number
        self._attack_angle_deg = float(attack_angle_deg)
        self._resolution_div = int(resolution_div)
        self._surface = np.asarray(naca0012_sharp() if surface is None else surface, np.float64)
        self._cl_cd_ref = float(lift_drag_reference)
        self._pressure_use_bicg = pressure_use_BiCG
        self._pressure_deflation = bool(pressure_deflation)
        if non_ortho_mode not in ("matrix", "rhs"):
            raise ValueError("non_ortho_mode: 'matrix' (the reference's nonOrthoFlags) or 'rhs' (every cross-metric term lagged)")
        self._non_ortho_flags = 25 if non_ortho_mode == "matrix" else 10
        self._stall_limit = int(stall_limit)
        self._U_mean = self.U_mean
        self._n_controls = self._n_jets
        if initial_domain_steps is not None:
            self._initial_domain_steps = int(initial_domain_steps)
        a = -self._attack_angle_deg * np.pi / 180.0
        rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        self._airfoil_coords = (self._surface @ rot.T).T.astype(np.float32)          # [2, n] as read_airfoil returns it
        self._airfoil_mask = self._get_airfoil_mask()
        self._sensor_locations = self._get_sensor_locations()
        FluidEnv.__init__(self, dt=dt, adaptive_cfl=adaptive_cfl, step_length=step_length, episode_length=episode_length,
                          ndims=ndims, **kw)
        self._last_control = None

    # ---- geometry on the render grid (airfoil_env_base.py:175-214, 559-660)
    @property
    def render_shape(self):
        return (600, 150, 150)

    @property
    def id(self) -> str:
        return f"Airfoil{self._ndims}D_Re{self._reynolds_number}_AoA{self._attack_angle_deg}"

    @property
    def initial_domain_id(self) -> str:
        return f"airfoil_{self._ndims}D_Re{int(self._reynolds_number)}_AoA{int(self._attack_angle_deg)}"

    def _physical_locations_to_grid_coords(self, p: np.ndarray) -> np.ndarray:
        p = np.array(p, np.float32, copy=True)
        p[0] = (p[0] + np.float32(1.5)) * np.float32(self.render_shape[0] / (self.L + 1.5))
        p[1] = (p[1] + np.float32(self.H / 2)) * np.float32(self.render_shape[1] / self.H)
        return np.round(p).astype(np.int64)

    def _get_airfoil_mask(self) -> np.ndarray:
        """Pixels [ny, nx] whose centre lies inside the section's polygon (in rounded pixel coordinates, as there)."""
        poly = self._physical_locations_to_grid_coords(self._airfoil_coords).T.astype(np.float64)
        nx, ny = self.render_shape[0], self.render_shape[1]
        xx, yy = np.meshgrid(np.linspace(0, nx - 1, nx), np.linspace(0, ny - 1, ny))
        return points_in_polygon(poly, np.stack([xx.ravel(), yy.ravel()], axis=1)).reshape(ny, nx)

    def _get_sensor_locations_2d(self) -> np.ndarray:
        def grid(xs, ys):
            gx, gy = np.meshgrid(xs, ys, indexing="ij")
            return np.stack([gx.ravel(), gy.ravel()])

        f32 = np.float32
        ys = np.linspace(-self.H / 2, self.H / 2, 10, dtype=f32)[1:-1]
        coarse = grid(np.arange(1.5, 2.6, 0.125, dtype=f32), ys)
        fine = grid(np.arange(1.05, 1.5 - 0.05, 0.05, dtype=f32), ys)
        near = grid(np.linspace(-0.125, self.airfoil_length, 10, dtype=f32), np.linspace(-0.5, 0.125, 8, dtype=f32))
        return np.concatenate([coarse, fine, near], axis=1).astype(f32)

    def _get_sensor_locations(self) -> np.ndarray:
        px = self._physical_locations_to_grid_coords(self._get_sensor_locations_2d())
        keep = ~self._airfoil_mask[px[1], px[0]]
        return px[:, keep]

    # ---- spaces (airfoil_env_2d.py:128-160)
    def _get_action_space(self):
        return spaces.Box(low=-1.0, high=1.0, shape=(self._n_jets,), dtype=np.float32)

    def _get_observation_space(self):
        n = self._sensor_locations.shape[-1]
        return spaces.Dict({
            "velocity": spaces.Box(low=-np.inf, high=np.inf, shape=(n, self._ndims), dtype=np.float32),
            "pressure": spaces.Box(low=-np.inf, high=np.inf, shape=(n,), dtype=np.float32),
        })

    # ---- domain and simulation (airfoil_env_base.py:216-311)
    def _get_domain(self):
        grow = 1.001 if (self._ndims == 3 and self._reynolds_number >= 5000) else 1.01      # airfoil_env_base.py:217-220
        self._mesh = self._mesh2d = make_airfoil_mesh(self.H, self.L, self.U_mean, self._attack_angle_deg, self._resolution_div,
                                                      grow, surface=self._surface)
        if self._ndims == 3:   # grid.py:601-617: res_z layers over z in [-H/2, H/2]
            self._mesh = extrude_mesh(self._mesh2d, self._res_z, -self.H / 2, self.H / 2)
        dom = build_domain(self._mesh, self._nu, batch=self._num_envs, device=self._cuda_device,
                           non_ortho_flags=self._non_ortho_flags, dtype=self._dtype)   # float64: the fp64 build (plain recurrences)
        # the pressure system of this mesh has a residual floor (2-4e-5) far above the reference's tolerance (1e-7): every
        # solve ends on its best iterate after ``stall_limit`` iterations without improvement ("Airfoil")
        dom.set_stall_limit(self._stall_limit)
        # 2-D: the multilevel preconditioner as a TRIAL of the pressure BiCGStab (solver policy; geometry-only tables, built once):
        # attempts are capped and verified on the true residual, one that fails is repeated with the plain recurrence and makes
        # the domain back off exponentially (3x fewer pressure iterations; the stiff start-up solves are the ones that fail)
        # policy pressure_multilevel_bicgstab (default on since the reductions are order-independent: policy.py)
        self._multilevel = dom.set_pressure_multilevel() if (self._ndims == 2 and get_solver_policy()["pressure_multilevel_bicgstab"]) else None
        return dom

    def _get_simulation(self, domain, prep_fn):
        return MultiBlockSimulation(domain, dt=self._dt, adaptive_CFL=self._adaptive_cfl, substeps="ADAPTIVE", corrector_steps=2,
                                    advection_tol=1e-6, pressure_tol=1e-7 if self._ndims == 2 else 1e-8, advect_non_ortho_steps=2,
                                    pressure_non_ortho_steps=4, pressure_use_BiCG=self._pressure_use_bicg,
                                    outflow=list(self._mesh.outflows), outflow_velocity=(self.U_mean, 0.0, 0.0),
                                    solver_double_fallback=True, BiCG_precondition_fallback=True)   # airfoil_env_base.py:283-285

    def _additional_initialization(self) -> None:
        dom = self._domain
        self._ring = WallRing(dom, [(FRONT, "+x", False), (TOP, "-y", False), (BOTTOM, "+y", True)])
        if self._ndims == 3:
            self._resampler = MultiBlockResampler3D(self._mesh.coords, self.render_shape, fill_max_steps=128, device=dom.device)
            self._sensors = self._resampler.sensor_gather(self._sensor_locations.reshape(3, -1).T)
        else:
            self._resampler = MultiBlockResampler(self._mesh.coords, self.render_shape[:2], fill_max_steps=128, device=dom.device)
            self._sensors = self._resampler.sensor_gather(self._sensor_locations.T)
        self._deflation_cos = dom.set_pressure_deflation() if self._pressure_deflation else 1.0
        self._initial_boundary = dom.boundary_velocity.clone()
        self._last_control = torch.zeros(self._num_envs, self._n_controls, device=dom.device, dtype=self._dtype)
        self._jet_locations_top = self._get_jet_locations()
        self._top_base_profile = torch.as_tensor(self._get_base_jet_profiles(), device=dom.device).to(self._dtype)      # [2, nx_top]
        # outward flux of every boundary slot per unit velocity component: s_f det Minv[axis, :]
        _, face, T = dom.boundary_tables()
        d = dom.dims
        axis, sign = face >> 1, np.where(face & 1, 1.0, -1.0)
        Minv = T[:, : d * d].reshape(-1, d, d)
        w = sign[:, None] * T[:, d * d][:, None] * Minv[np.arange(len(face)), axis, :]
        self._flux_w = torch.as_tensor(w.T.astype(np.float32), device=dom.device).to(self._dtype)         # [d, NB]
        free = np.zeros(len(face), bool)
        for b, f in list(self._mesh.outflows) + [(TOP, "-y")]:
            blk = dom.blocks[b]
            from ..simulation.multiblock import face_index
            s0 = blk.boundary_slot0[face_index(f)]
            free[s0:s0 + blk.face_cells(face_index(f))] = True
        self._free_slots = torch.as_tensor(free, device=dom.device)

    def _get_jet_locations(self):
        return jet_locations(self._mesh2d.coords[TOP])

    def _get_base_jet_profiles(self) -> np.ndarray:
        """Unit-flux-sum jet profiles along the wall normal (airfoil_env_base.py:463-513).  The normals are taken from the
        wall ring at ``vertices of the front face + cell index`` -- one entry past the cell's own normal, as there."""
        normals = self._ring.wall_normals.cpu().numpy()                  # [2, ring]
        n_front = self._mesh2d.coords[FRONT].shape[1]
        nx = self._mesh2d.coords[TOP].shape[2] - 1
        out = np.zeros((2, nx), np.float32)
        for i0, i1 in self._jet_locations_top:
            prof = jet_profile(i1 - i0 + 3)[1:-1]
            prof = prof / prof.sum()
            out[:, i0:i1 + 1] = prof[None, :] * normals[:, n_front + i0:n_front + i1 + 1]
        return out

    def _fill_initial_fields(self) -> None:
        self._developed = getattr(self, "_developed", None)
        dom = self._domain
        if self._developed is None:
            dom.boundary_velocity.copy_(self._initial_boundary)
            dom.velocity.zero_()
            dom.velocity[:, 0] = self.U_mean
            dom.pressure.zero_()
            self._sim.make_divergence_free()
            for _ in range(self._initial_domain_steps):
                self._sim.single_step()
            self._developed = dom.Clone()
        dom.Restore(self._developed)
        self._last_control = torch.zeros(self._num_envs, self._n_controls, device=dom.device, dtype=self._dtype)

    def _randomize_domain(self) -> None:
        """airfoil_env_base.py:327-332."""
        max_n = int(0.05 * self._episode_length)
        n_steps = int(self._np_rng.integers(int(0.5 * max_n), max_n)) + 1
        dom, g = self._domain, self._torch_rng_cuda
        dom.velocity.add_(torch.randn(dom.velocity.shape, device=dom.device, generator=g) * 0.01)
        dom.pressure.add_(torch.randn(dom.pressure.shape, device=dom.device, generator=g) * 0.01)
        for _ in range(n_steps):
            self._sim.single_step()

    # ---- forces, actuation, step
    def _get_drag_and_lift(self):
        """[B] in 2-D; [B, NZ] per spanwise layer in 3-D (face area = edge length x D / res_z, :448-449)."""
        f = self._ring.forces(self._domain, self._nu, layer_height=self.D / getattr(self, "_res_z", 1))
        norm = 0.5 * self.U_mean ** 2 * self.airfoil_length
        return f[:, 0] / norm, f[:, 1] / norm

    def _action_to_control(self, action: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def _balance_boundary_fluxes(self) -> None:
        """``balance_boundary_fluxes(domain, [tail lower, tail upper, airfoil top])`` (PISOtorch_simulation.py:188-224):
        the velocities of the free faces are scaled so that the fluxes of all prescribed faces sum to zero."""
        ub = self._domain.boundary_velocity
        flux = (ub * self._flux_w[None]).sum(1)                                   # [B, NB]
        var = (flux * self._free_slots).sum(-1)
        fixed = flux.sum(-1) - var
        scale = torch.where((fixed + var).abs() > 1e-7, -fixed / var, torch.ones_like(var))
        ub.mul_(torch.where(self._free_slots[None, None, :], scale[:, None, None], torch.ones_like(scale)[:, None, None]))

    def _apply_action(self, action: torch.Tensor) -> None:
        self._domain.blocks[TOP].boundary("-y").copy_(self._action_to_control(action.reshape(self._num_envs, self._n_jets)))
        self._balance_boundary_fluxes()

    def _step_impl(self, action: torch.Tensor):
        target = action.reshape(self._num_envs, self._n_controls)
        cds, cls = [], []
        for _ in range(self._n_sim_steps):
            control = self._last_control + self._action_smoothing_alpha * (target - self._last_control)
            self._last_control = control
            if self._enable_actions:
                self._apply_action(control)
            self._sim.single_step()
            cd, cl = self._get_drag_and_lift()
            cds.append(cd); cls.append(cl)
        obs = self._get_global_obs()
        # mean over the sim steps as adds in step order (mean(0) of the stack picks its summation order per column: identical envs
        # then differ in the last bit of drag / lift)
        cd, cl = cds[0].clone(), cls[0].clone()
        for k in range(1, len(cds)):
            cd += cds[k]; cl += cls[k]
        cd /= len(cds); cl /= len(cls)
        return obs, cl / cd - self._cl_cd_ref, False, {"drag": cd, "lift": cl}

    def _save_initial_domain(self, mode, idx: int, env: int = 0) -> None:
        from ..simulation.domain_io import save_multiblock_domain

        out_dir = self._get_domain_dir(idx)
        out_dir.mkdir(parents=True, exist_ok=True)
        save_multiblock_domain(self._domain, str(out_dir / mode.value), env=env, name="AirfoilDomain")

    def get_velocity(self) -> torch.Tensor:
        u = self._resampler(self._domain.velocity)                   # [B, d, (z,) y, x]
        u[..., torch.as_tensor(self._airfoil_mask, device=u.device)] = 0.0
        return u


class AirfoilEnv2D(AirfoilEnvBase):
    """``AirfoilEnv2D`` (airfoil_env_2d.py:27-190): one agent drives the three jets; the mean of the action is removed so
    that the jets together blow as much as they suck."""

    def __init__(self, reynolds_number: float, adaptive_cfl: float, step_length: float, episode_length: int, dt: float,
                 attack_angle_deg: float, **kw):
        kw.pop("ndims", None)
        super().__init__(ndims=2, reynolds_number=reynolds_number, adaptive_cfl=adaptive_cfl, step_length=step_length,
                         episode_length=episode_length, dt=dt, attack_angle_deg=attack_angle_deg, **kw)

    @property
    def n_agents(self) -> int:
        return self._n_jets if self._use_marl else 1

    def _action_to_control(self, action: torch.Tensor) -> torch.Tensor:
        """[B, n_jets] -> wall velocity of the top face [B, 2, nx] (airfoil_env_2d.py:173-190)."""
        v = action - action.mean(dim=1, keepdim=True)
        mx = v.abs().amax(dim=1, keepdim=True)
        v = torch.where(mx > 1.0, v / mx, v)
        prof = self._top_base_profile[None].repeat(self._num_envs, 1, 1)
        for i, (i0, i1) in enumerate(self._jet_locations_top):
            prof[:, :, i0:i1 + 1] *= v[:, i, None, None]
        return prof


AIRFOIL_3D_DEFAULT_CONFIG = {
    "n_agents": 4, "reynolds_number": 3e3, "dt": 0.05, "adaptive_cfl": 0.8, "step_length": 0.25, "episode_length": 200,
    "attack_angle_deg": 10.0, "local_obs_window": 1, "use_marl": False, "local_reward_weight": 0.5, "local_2d_obs": False,
    "init_from_2d": True, "dtype": torch.float32, "load_initial_domain": True, "load_domain_statistics": True,
    "randomize_initial_state": True, "enable_actions": True, "differentiable": False,
}


class AirfoilEnv3D(AirfoilEnvBase):
    """``AirfoilEnv3D`` (airfoil_env_3d.py:50-593): the wing spans z in [-H/2, H/2] (periodic, ``res_z`` = 96 layers), the
    three jets are cut into ``n_agents`` spanwise segments with one action triple each.  Observation layout, local windows
    and the per-layer force bookkeeping are those of the 3-D cylinder env (same reference functions,
    ``extract_global_3d_obs`` / ``transform_global_to_local_obs_3d``).

    ``init_from_2d``: the reference starts the 3-D flow from a published 2-D initial domain extruded over the span
    (:524-563); here the first half of the development steps is run on the 2-D mesh and extruded the same way."""

    _supports_marl = True
    _n_sensors_per_agent: int = 1
    _res_z: int = 96

    def __init__(self, n_agents: int, reynolds_number: float, adaptive_cfl: float, step_length: float, episode_length: int,
                 dt: float, attack_angle_deg: float, local_obs_window: int, use_marl: bool,
                 local_reward_weight: Optional[float], local_2d_obs: bool = False, init_from_2d: bool = True,
                 res_z: Optional[int] = None, **kw):
        if res_z is not None:
            self._res_z = int(res_z)
        if n_agents < 1 or self._res_z % n_agents != 0:
            raise ValueError("n_agents must be a positive integer that evenly divides circle_resolution_angular.")
        if local_2d_obs and not use_marl:
            raise ValueError("Local 2D observations are only supported in multi-agent mode.")
        self._local_2d_obs = bool(local_2d_obs)
        self._n_agents = int(n_agents)
        self._local_obs_window = int(local_obs_window)
        self._local_reward_weight = local_reward_weight
        self._init_from_2d = bool(init_from_2d)
        if local_2d_obs:
            self._n_sensors_per_agent = 1
            self._local_obs_window = 1
        kw.pop("ndims", None)
        super().__init__(ndims=3, reynolds_number=reynolds_number, adaptive_cfl=adaptive_cfl, step_length=step_length,
                         episode_length=episode_length, dt=dt, attack_angle_deg=attack_angle_deg, use_marl=use_marl, **kw)
        self._n_controls = self._n_agents * self._n_jets

    # ---- spaces (airfoil_env_3d.py:205-275)
    def _get_action_space(self):
        shape = (self._n_jets,) if self._use_marl else (self._n_agents, self._n_jets)
        return spaces.Box(low=-1.0, high=1.0, shape=shape, dtype=np.float32)

    def _get_observation_space(self):
        n, spa = self._sensor_locations.shape[-1], self._n_sensors_per_agent
        if self._use_marl and self._local_2d_obs:
            v_shape, p_shape = (n, 2), (n,)
        else:
            lead = self._local_obs_window if self._use_marl else self._n_agents
            v_shape, p_shape = (lead, spa, 3, n), (lead, spa, n)
        return spaces.Dict({
            "velocity": spaces.Box(low=-np.inf, high=np.inf, shape=v_shape, dtype=np.float32),
            "pressure": spaces.Box(low=-np.inf, high=np.inf, shape=p_shape, dtype=np.float32),
        })

    @property
    def n_agents(self) -> int:
        return self._n_agents if self._use_marl else 1

    @property
    def _n_sensors_z(self) -> int:
        return self._n_agents * self._n_sensors_per_agent

    @property
    def _nz_per_agent(self) -> int:
        return self._res_z // self._n_agents

    # ---- sensors (airfoil_env_3d.py:303-344): the kept 2-D pixels on n_sensors_z spanwise planes
    def _get_sensor_locations(self) -> np.ndarray:
        xy = super()._get_sensor_locations()                                    # [2, n] after masking
        nz = self._n_sensors_z
        z = np.linspace(-self.H / 2, self.H / 2, nz + 1, dtype=np.float32)[:-1] + np.float32(self.H / (2 * nz))
        pz = np.round((z + np.float32(self.D / 2)) * np.float32(self.render_shape[1] / self.D)).astype(np.int64)
        out = np.empty((3, nz, xy.shape[1]), np.int64)
        out[0], out[1], out[2] = xy[0][None, :], xy[1][None, :], pz[:, None]
        return out

    # ---- observations (obs_extraction.py:60-205)
    def _get_global_obs(self) -> Dict[str, torch.Tensor]:
        dom = self._domain
        B, nz, n = self._num_envs, self._n_sensors_z, self._sensor_locations.shape[-1]
        u = self._sensors(dom.velocity).to(self._dtype)      # [B, 3, nz * n]
        p = self._sensors(dom.pressure).to(self._dtype)
        if self._local_2d_obs:
            u = u[:, :2]
        vd = u.shape[1]
        u = u.permute(0, 2, 1).contiguous().reshape(B, nz, vd, n)              # the reference's view of [z, sensor, comp]
        u = u.reshape(B, self._n_agents, self._n_sensors_per_agent, vd, n)
        if self._local_2d_obs:
            u = u.permute(0, 1, 2, 4, 3)
        return {"velocity": u, "pressure": p.reshape(B, self._n_agents, self._n_sensors_per_agent, n)}

    def _get_local_obs(self) -> Dict[str, torch.Tensor]:
        g = self._get_global_obs()
        W, na = self._local_obs_window, self._n_agents
        seg = (torch.arange(na)[:, None] + torch.arange(W)[None, :] - W // 2) % na
        out = {}
        for k, v in g.items():
            win = v[:, seg.to(v.device)]                                          # [B, agents, W, ...]
            if self._local_2d_obs:
                win = win.reshape(self._num_envs, na, *win.shape[4:])
            out[k] = win
        return out

    # ---- initial state
    def _fill_initial_fields(self) -> None:
        if getattr(self, "_developed", None) is None and self._init_from_2d:
            self._develop_from_2d()
        super()._fill_initial_fields()

    def _develop_from_2d(self) -> None:
        """Half of the development steps on the 2-D mesh, extruded over the span, as the start of the 3-D development."""
        from ..simulation.multiblock import MultiBlockSimulation as Sim

        n2d = self._initial_domain_steps // 2
        dom2 = build_domain(self._mesh2d, self._nu, batch=1, device=self._cuda_device)
        try:
            sim2 = Sim(dom2, dt=self._dt, adaptive_CFL=self._adaptive_cfl, substeps="ADAPTIVE", corrector_steps=2,
                       advection_tol=1e-6, pressure_tol=1e-7, advect_non_ortho_steps=2, pressure_non_ortho_steps=4,
                       pressure_use_BiCG=self._pressure_use_bicg, outflow=list(self._mesh2d.outflows),
                       outflow_velocity=(self.U_mean, 0.0, 0.0))
            dom2.velocity[:, 0] = self.U_mean
            sim2.make_divergence_free()
            for _ in range(n2d):
                sim2.single_step()
            dom3, nz = self._domain, self._res_z
            dom3.boundary_velocity.copy_(self._initial_boundary)
            dom3.velocity.zero_()
            for b2, b3 in zip(dom2.blocks, dom3.blocks):
                n2 = b2.n_cells
                u2 = dom2.velocity[0, :, b2.cell_offset:b2.cell_offset + n2]     # [2, n2]
                p2 = dom2.pressure[0, b2.cell_offset:b2.cell_offset + n2]
                dom3.velocity[:, :2, b3.cell_offset:b3.cell_offset + nz * n2] = u2.repeat(1, nz)[None]
                dom3.pressure[:, b3.cell_offset:b3.cell_offset + nz * n2] = p2.repeat(nz)[None]
                for f, s0 in enumerate(b2.boundary_slot0):                       # outflow faces carry the convected profile
                    if s0 >= 0 and f < 4:
                        w = b2.face_cells(f)
                        dom3.boundary_velocity[:, :2, b3.boundary_slot0[f]:b3.boundary_slot0[f] + nz * w] = \
                            dom2.boundary_velocity[0, :, s0:s0 + w].repeat(1, nz)[None]
        finally:
            dom2.close()
        self._sim.make_divergence_free()
        for _ in range(self._initial_domain_steps - n2d):
            self._sim.single_step()
        self._developed = self._domain.Clone()

    # ---- actuation (airfoil_env_3d.py:366-407)
    def _additional_initialization(self) -> None:
        super()._additional_initialization()
        nz = self._res_z
        base = self._top_base_profile                                            # [2, nx]
        self._top_base_profile3 = torch.cat([base, torch.zeros_like(base[:1])], 0)[:, None, :].expand(3, nz, base.shape[1]).contiguous()

    def _action_to_control(self, action: torch.Tensor) -> torch.Tensor:
        """[B, n_agents * n_jets] -> wall velocity of the top face [B, 3, nz * nx]: per agent the mean over its jets is
        removed, amplitudes above one are scaled back, every agent drives its own spanwise layers."""
        B = self._num_envs
        v = action.reshape(B, self._n_agents, self._n_jets)
        v = v - v.mean(dim=2, keepdim=True)
        mx = v.abs().amax(dim=2, keepdim=True)
        v = torch.where(mx > 1.0, v / mx, v).repeat_interleave(self._nz_per_agent, dim=1)      # [B, nz, n_jets]
        prof = self._top_base_profile3[None].repeat(B, 1, 1, 1)                              # [B, 3, nz, nx]
        for i, (i0, i1) in enumerate(self._jet_locations_top):
            prof[:, :, :, i0:i1 + 1] *= v[:, None, :, i, None]
        return prof.reshape(B, 3, -1)

    def _apply_action(self, action: torch.Tensor) -> None:
        self._domain.blocks[TOP].boundary("-y").copy_(self._action_to_control(action))
        self._balance_boundary_fluxes()

    # ---- step (airfoil_env_3d.py:409-458)
    def _step_impl(self, action: torch.Tensor):
        obs, _, term, info = super()._step_impl(action)
        all_cds, all_cls = info.pop("drag"), info.pop("lift")                    # [B, NZ]
        cd, cl = all_cds.sum(-1) / self.D, all_cls.sum(-1) / self.D
        return obs, cl / cd - self._cl_cd_ref, term, {"drag": cd, "lift": cl, "all_cds": all_cds, "all_cls": all_cls}

    def _step_marl_impl(self, actions: torch.Tensor):
        if self._local_reward_weight is None:
            raise ValueError("local_reward_weight must be set for multi-agent step.")
        _, global_reward, terminated, info = self._step_impl(actions)
        local_obs = self._get_local_obs()
        all_cds, all_cls = info.pop("all_cds"), info.pop("all_cls")
        B, na = self._num_envs, self._n_agents
        local_cd = all_cds.reshape(B, na, -1).sum(-1) / (self.D / na)
        local_cl = all_cls.reshape(B, na, -1).sum(-1) / (self.D / na)
        w = self._local_reward_weight
        agent_rewards = w * (local_cl / local_cd - self._cl_cd_ref) + (1 - w) * global_reward[:, None]
        info["global_reward"] = global_reward
        return local_obs, agent_rewards, terminated, info

    def render(self, *a, **kw) -> np.ndarray:
        speed = torch.linalg.vector_norm(self.get_velocity()[0], dim=0)
        return speed[speed.shape[0] // 2].detach().cpu().numpy()
