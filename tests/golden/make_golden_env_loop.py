"""What the reference's ``FluidEnv`` base class DOES around the simulation -- the generator of the on-disk initial domains
(``init``, ``envs/fluid_env.py:1114-1190``), the reset-time choice of an initial domain (``_set_initial_state``, ``:507-551``) and
the step / truncation bookkeeping (``:749-800``) -- recorded HERE by running the reference's own class with a toy subclass: the
simulation, the domain I/O, the data download helpers and the packages the image lacks (gymnasium, seaborn) are recording
stand-ins, nothing of the reference's source is stored.

    python tests/golden/make_golden_env_loop.py        ->  tests/golden/reference_env_loop.json

``tests/test_env_loop_golden.py`` drives ``fluidgym_amd``'s ``FluidEnv`` with the same toy subclass and compares the event sequences.
"""
import importlib.util
import json
import logging
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))
EVENTS = []          # the log every stand-in writes to
ON_DISK = set()      # paths "saved" so far (load_domain raises FileNotFoundError for anything else, like a missing file)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _Box:          # gymnasium.spaces.Box as far as the base class needs it
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = np.full(shape, low, dtype), np.full(shape, high, dtype), tuple(shape), dtype


class _Dict(dict):
    @property
    def spaces(self):
        return self


class _Domain:
    def PrepareSolve(self):
        pass


def load_reference_fluid_env(data_dir):
    for pkg in ["fluidgym", "fluidgym.simulation", "fluidgym.simulation.pict", "fluidgym.simulation.pict.util", "fluidgym.util", "fluidgym.envs"]:
        _stub(pkg).__path__ = []
    _stub("seaborn")
    gym = _stub("gymnasium", spaces=types.SimpleNamespace(Box=_Box, Dict=_Dict, Space=object))
    sys.modules["gymnasium.spaces"] = gym.spaces

    class _Ext(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return type(name, (), {})

    _stub("fluidgym.simulation.extensions", PISOtorch=_Ext("PISOtorch"))

    def save_domain(domain, path):
        ON_DISK.add(path)
        EVENTS.append({"op": "save_domain", "path": os.path.relpath(path, data_dir)})

    def load_domain(path, **kw):
        if path not in ON_DISK:
            raise FileNotFoundError(path)
        EVENTS.append({"op": "load_domain", "path": os.path.relpath(path, data_dir)})
        return _Domain()

    _stub("fluidgym.simulation.pict.util.domain_io", load_domain=load_domain, save_domain=save_domain)
    _stub("fluidgym.simulation.pict.util.output", _resample_block_data=lambda *a, **k: None, plot_grids=lambda *a, **k: None)
    _stub("fluidgym.simulation.simulation", Simulation=type("Simulation", (), {}))

    def _missing(*a, **k):
        raise FileNotFoundError("not downloaded")

    _stub("fluidgym.util.data_utils", load_statistics=_missing, load_uncontrolled_episode=_missing,
          prepare_initial_domains=lambda **k: EVENTS.append({"op": "prepare_initial_domains"}), save_statistics=lambda *a, **k: None,
          save_uncontrolled_episode=lambda *a, **k: None)
    cfg = _load(f"{REF}/fluidgym/config.py", "fluidgym.config")            # the reference's own config and types modules
    cfg.config.update("local_data_path", data_dir)
    _load(f"{REF}/fluidgym/types.py", "fluidgym.types")
    return _load(f"{REF}/fluidgym/envs/fluid_env.py", "fluidgym.envs.fluid_env")


def make_toy(mod, restart, initial_domain_steps):
    class Toy(mod.FluidEnv):
        _supports_marl = False
        _initial_domain_restart = restart
        _initial_domain_steps = initial_domain_steps
        _metrics = ["m"]

        def _get_action_space(self):
            return _Box(-1.0, 1.0, (2,))

        def _get_observation_space(self):
            return _Dict(o=_Box(-1.0, 1.0, (3,)))

        @property
        def n_agents(self):
            return 1

        @property
        def render_shape(self):
            return (4, 4)

        @property
        def id(self):
            return "toy"

        @property
        def initial_domain_id(self):
            return "toy_domain"

        def _get_domain(self):
            EVENTS.append({"op": "get_domain"})
            return _Domain()

        def _get_prep_fn(self, domain):
            return {}

        def _get_simulation(self, domain, prep_fn):
            return object()

        def _randomize_domain(self):
            EVENTS.append({"op": "randomize_domain", "draw": int(self._np_rng.integers(0, 1000))})

        def _apply_action(self, action):
            pass

        def _get_global_obs(self):
            return {"o": torch.zeros(3)}

        def _get_local_obs(self):
            return {"o": torch.zeros(1, 3)}

        def _step_impl(self, action):
            EVENTS.append({"op": "step_impl", "actions_enabled": bool(self._enable_actions)})
            return self._get_global_obs(), torch.zeros(()), False, {"m": torch.zeros(())}

        def _step_marl_impl(self, action):
            raise NotImplementedError

        def _get_render_data(self, render_3d, output_path=None):
            return {}

        def render(self, *a, **k):
            return np.zeros((1, 1))

        def plot(self, output_path=None):
            pass

        def _additional_initialization(self):
            pass

    return Toy


def _compress(events):
    """runs of step_impl as one record with a count"""
    out = []
    for e in events:
        if e["op"] == "step_impl" and out and out[-1]["op"] == "step_impl" and out[-1]["actions_enabled"] == e["actions_enabled"]:
            out[-1]["count"] += 1
        else:
            out.append(dict(e, count=1) if e["op"] == "step_impl" else dict(e))
    return out


def main():
    logging.disable(logging.CRITICAL)
    torch.cuda.is_available = lambda: True          # the base class's "FluidGym requires CUDA" guard; nothing here touches a GPU
    data_dir = tempfile.mkdtemp()
    mod = load_reference_fluid_env(data_dir)
    kw = dict(adaptive_cfl=0.8, dt=0.1, step_length=0.2, episode_length=1000, ndims=2, use_marl=False, cuda_device=torch.device("cpu"),
              load_initial_domain=False, load_domain_statistics=False, randomize_initial_state=True)
    cases = []
    for restart in (False, True):
        ON_DISK.clear()
        EVENTS.clear()
        env = make_toy(mod, restart, 20)(**kw)
        EVENTS.clear()
        env.init(domain_idxs=[0, 1])
        first = _compress(EVENTS)
        EVENTS.clear()
        env.init(domain_idxs=[1, 2])                 # index 1 exists now: kept, only loaded
        second = _compress(EVENTS)
        EVENTS.clear()
        env.init()                                   # all N_INITIAL_DOMAINS indices (0-2 exist already)
        third = _compress(EVENTS)
        # resets after init: the initial domain is chosen by the env's generator when randomising, index 0 otherwise
        draws = []
        for seed, randomize in ((5, True), (6, True), (7, True), (8, True), (11, False), (12, None)):
            EVENTS.clear()
            try:
                env.reset(seed=seed, randomize=randomize)
                draws.append({"seed": seed, "randomize": randomize, "events": _compress(EVENTS)})
            except RuntimeError as e:
                draws.append({"seed": seed, "randomize": randomize, "raised": str(e)})
        cases.append({"initial_domain_restart": restart, "initial_domain_steps": 20, "n_initial_domains": mod.N_INITIAL_DOMAINS,
                      "mode_seeds": list(mod.MODE_SEEDS), "init_0_1": first, "init_1_2": second, "init_all_saves": [e["path"] for e in third if e["op"] == "save_domain"],
                      "init_all_step_runs": [e["count"] for e in third if e["op"] == "step_impl"], "resets_after_init": draws,
                      "flags_after_init": {"enable_actions": bool(env._enable_actions)}})
    # step / truncation bookkeeping
    ON_DISK.clear()
    env = make_toy(mod, False, 0)(**dict(kw, episode_length=3, randomize_initial_state=False))
    errors = {}
    try:
        env.step(torch.zeros(2))
    except RuntimeError as e:
        errors["step_before_reset"] = str(e)
    try:
        env.reset()
    except ValueError as e:
        errors["reset_without_seed"] = str(e)
    env.reset(seed=1)
    try:
        env.step(torch.zeros(3))
    except ValueError as e:
        errors["wrong_action_shape"] = str(e)
    flags = [list(env.step(torch.zeros(2))[2:4]) for _ in range(3)]
    try:
        env.step(torch.zeros(2))
    except RuntimeError as e:
        errors["step_after_truncation"] = str(e)
    # sample_action: uniform in the action box from the env's device generator (fluid_env.py:360-381); CPU device here
    env.seed(42)
    samples = [env.sample_action().tolist() for _ in range(3)]
    env2 = make_toy(mod, False, 0)(**dict(kw, randomize_initial_state=False))
    try:
        env2.sample_action()
        unseeded = None
    except RuntimeError as e:
        unseeded = str(e)
    out = {"sample_action": {"seed": 42, "samples": samples, "unseeded_error": unseeded}, "init": cases, "step": {"episode_length": 3, "terminated_truncated": flags, "errors": errors, "n_sim_steps": env.n_sim_steps,
                                   "time_passed_after_3": env.time_passed}}
    with open(os.path.join(OUT, "reference_env_loop.json"), "w") as f:
        json.dump(out, f, indent=1)
    for c in cases:
        print("restart", c["initial_domain_restart"], [(e["op"], e.get("count", e.get("path", e.get("draw", "")))) for e in c["init_0_1"]][:14])
    print(out["step"])


if __name__ == "__main__":
    main()
