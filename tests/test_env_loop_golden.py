"""``fluidgym_amd``'s ``FluidEnv`` base class against what the reference's base class DOES (``tests/golden/reference_env_loop.json``,
recorded by ``tests/golden/make_golden_env_loop.py`` from the reference's own class driving a toy subclass): the generator of the
on-disk initial domains (``init``: seeds, draws of the env's NumPy generator, numbers of uncontrolled steps, order and paths of
the files), the initial domain a reset picks once files exist, and the step / truncation bookkeeping with its error messages.
The same toy subclass runs here; the domain I/O is a recording stand-in (nothing touches a GPU)."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import fluidgym_amd  # noqa: F401
from fluidgym_amd import spaces
from fluidgym_amd.envs import fluid_env as FE
from fluidgym_amd.simulation import domain_io

with open(os.path.join(os.path.dirname(__file__), "golden", "reference_env_loop.json")) as f:
    GOLD = json.load(f)


class _FakeDomain:
    solver = SimpleNamespace(nx=1, ny=1, nz=1, fixed=[], close=lambda: None, reset_solver_state=lambda: None)

    def Clone(self):
        return {}

    def Restore(self, snap):
        pass


def _make_toy(events, restart, initial_domain_steps):
    class Toy(FE.FluidEnv):
        _supports_marl = False
        _initial_domain_restart = restart
        _initial_domain_steps = initial_domain_steps
        _metrics = ["m"]

        def _get_action_space(self):
            return spaces.Box(low=-1.0, high=1.0, shape=(2,), dtype=np.float32)

        def _get_observation_space(self):
            return spaces.Dict({"o": spaces.Box(low=-1.0, high=1.0, shape=(3,), dtype=np.float32)})

        @property
        def id(self):
            return "toy"

        @property
        def initial_domain_id(self):
            return "toy_domain"

        def _get_domain(self):
            return _FakeDomain()

        def _fill_initial_fields(self):          # a generated (not loaded) initial state: the reference builds a new domain here
            events.append({"op": "get_domain"})

        def _get_prep_fn(self, domain):
            return {}

        def _get_simulation(self, domain, prep_fn):
            return object()

        def _additional_initialization(self):
            pass

        def _randomize_domain(self):
            events.append({"op": "randomize_domain", "draw": int(self._np_rng.integers(0, 1000))})

        def _apply_action(self, action):
            pass

        def _get_global_obs(self):
            return {"o": torch.zeros(1, 3)}

        def _step_impl(self, action):
            events.append({"op": "step_impl", "actions_enabled": bool(self._enable_actions)})
            return self._get_global_obs(), torch.zeros(1), False, {"m": torch.zeros(1)}

        def close(self):
            pass

    return Toy


def _compress(events):
    out = []
    for e in events:
        if e["op"] == "step_impl" and out and out[-1]["op"] == "step_impl" and out[-1]["actions_enabled"] == e["actions_enabled"]:
            out[-1]["count"] += 1
        else:
            out.append(dict(e, count=1) if e["op"] == "step_impl" else dict(e))
    return out


@pytest.fixture
def recording_io(tmp_path, monkeypatch):
    events = []
    monkeypatch.setenv("FLUIDGYM_DATA_PATH", str(tmp_path))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)      # the "FluidGym requires CUDA" guard of reset()

    def save_domain(domain, path, env=0):
        open(path + ".json", "w").close()
        events.append({"op": "save_domain", "path": os.path.relpath(path, tmp_path)})

    def load_domain(path, device=None, batch=1):
        events.append({"op": "load_domain", "path": os.path.relpath(path, tmp_path)})
        return _FakeDomain()

    monkeypatch.setattr(domain_io, "save_domain", save_domain)
    monkeypatch.setattr(domain_io, "load_domain", load_domain)
    return events


KW = dict(adaptive_cfl=0.8, dt=0.1, step_length=0.2, episode_length=1000, ndims=2, use_marl=False, cuda_device=torch.device("cpu"),
          load_initial_domain=False, load_domain_statistics=False, randomize_initial_state=True)


def _generating(events):
    """the reference loads an existing initial domain twice per mode while skipping it (existence check + load); here it is
    skipped without a read: compare what GENERATES state"""
    return [e for e in events if e["op"] != "load_domain"]


@pytest.mark.parametrize("case", GOLD["init"], ids=lambda c: f"restart_{c['initial_domain_restart']}")
def test_init_follows_the_reference_generator(case, recording_io):
    events = recording_io
    assert (FE.N_INITIAL_DOMAINS, FE.MODE_SEEDS) == (case["n_initial_domains"], case["mode_seeds"])
    env = _make_toy(events, case["initial_domain_restart"], case["initial_domain_steps"])(**KW)
    events.clear()
    env.init(domain_idxs=[0, 1])
    assert _compress(events) == _generating(case["init_0_1"])
    events.clear()
    env.init(domain_idxs=[1, 2])
    assert _compress(events) == _generating(case["init_1_2"])
    events.clear()
    env.init()
    third = _compress(events)
    assert [e["path"] for e in third if e["op"] == "save_domain"] == case["init_all_saves"]
    assert [e["count"] for e in third if e["op"] == "step_impl"] == case["init_all_step_runs"]
    assert env._enable_actions is case["flags_after_init"]["enable_actions"] and env._load_domain_on_reset is True
    # resets once the files exist: index drawn from the env's generator when randomising (before the randomisation itself), else 0
    for want in case["resets_after_init"]:
        events.clear()
        env.reset(seed=want["seed"], randomize=want["randomize"])
        assert _compress(events) == want["events"], want


def test_reset_reports_a_missing_initial_domain_like_the_reference(recording_io):
    events = recording_io
    case = GOLD["init"][0]
    env = _make_toy(events, False, 20)(**KW)
    env.init(domain_idxs=[0, 1])
    env.init(domain_idxs=[1, 2])
    # (the rule is what is held here:
    # a drawn index without files raises the reference's message, index 0 is used without randomisation)
    drawn = int(np.random.default_rng(5).integers(0, FE.N_INITIAL_DOMAINS))
    assert drawn not in (0, 1, 2)
    with pytest.raises(RuntimeError, match="Initial domain not found. Please ensure it was downloaded."):
        env.reset(seed=5, randomize=True)
    events.clear()
    env.reset(seed=11, randomize=False)
    assert _compress(events) == [e for r in case["resets_after_init"] if r["seed"] == 11 for e in r["events"]]


def test_step_bookkeeping_and_messages_are_the_reference_s(recording_io):
    ref = GOLD["step"]
    env = _make_toy(recording_io, False, 0)(**dict(KW, episode_length=ref["episode_length"], randomize_initial_state=False))
    with pytest.raises(RuntimeError) as e:
        env.step(torch.zeros(2))
    assert str(e.value) == ref["errors"]["step_before_reset"]
    with pytest.raises(ValueError) as e:
        env.reset()
    assert str(e.value) == ref["errors"]["reset_without_seed"]
    env.reset(seed=1)
    with pytest.raises(ValueError) as e:
        env.step(torch.zeros(3))
    assert str(e.value) == ref["errors"]["wrong_action_shape"]
    flags = [list(env.step(torch.zeros(2))[2:4]) for _ in range(ref["episode_length"])]
    assert flags == ref["terminated_truncated"]
    with pytest.raises(RuntimeError) as e:
        env.step(torch.zeros(2))
    assert str(e.value) == ref["errors"]["step_after_truncation"]
    assert env.n_sim_steps == ref["n_sim_steps"] and env.time_passed == pytest.approx(ref["time_passed_after_3"])


def test_sample_action_draws_like_the_reference(recording_io):
    """fluid_env.py:360-381: ``low + (high - low) * torch.rand(shape, generator=<the env's device generator seeded by seed()>)``
    -- the same numbers for the same seed on the same (CPU) device."""
    ref = GOLD["sample_action"]
    env = _make_toy(recording_io, False, 0)(**dict(KW, randomize_initial_state=False))
    with pytest.raises(RuntimeError) as e:
        env.sample_action()
    assert str(e.value) == ref["unseeded_error"]
    env.seed(ref["seed"])
    for want in ref["samples"]:
        assert env.sample_action().tolist() == want
