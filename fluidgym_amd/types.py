"""``EnvMode`` and the ``FluidEnvLike`` protocol (reference ``fluidgym/types.py:15-243``).

``FluidEnvLike`` is the surface RL code programs against: wrappers, the SB3 / PettingZoo / TorchRL adapters and
``ParallelFluidEnv`` of the reference only ever touch these members, so anything that satisfies the protocol is a
drop-in.  Here it is a ``runtime_checkable`` structural protocol (``isinstance(env, FluidEnvLike)`` checks member
presence) that both :class:`fluidgym_amd.envs.fluid_env.FluidEnv` and :class:`fluidgym_amd.envs.parallel_env.ParallelFluidEnv`
satisfy; ``tests/test_abi_and_host.py`` holds the member list against the reference's.  Rendering members
(``render`` / ``save_gif``) exist on the envs but raise: rendering is outside the hot path.
"""
from __future__ import annotations

from enum import Enum
from pathlib import Path
from typing import Any, Dict, Optional, Protocol, Tuple, Union, runtime_checkable

import torch


class EnvMode(Enum):
    TRAIN = "train"
    VAL = "val"
    TEST = "test"


@runtime_checkable
class FluidEnvLike(Protocol):
    """Members of the reference protocol, same names and argument meaning (``types.py:24-243``)."""

    @property
    def action_space(self) -> Any: ...

    @property
    def observation_space(self) -> Any: ...

    @property
    def differentiable(self) -> bool: ...

    @property
    def use_marl(self) -> bool: ...

    @property
    def metrics(self) -> list: ...

    @property
    def episode_length(self) -> int: ...

    @property
    def n_agents(self) -> int: ...

    @property
    def cuda_device(self) -> torch.device: ...

    def step(self, action: torch.Tensor) -> Tuple[Union[torch.Tensor, Dict[str, torch.Tensor]], torch.Tensor, Any, Any, Any]: ...

    def reset(self, seed: Optional[int] = None, randomize: Optional[bool] = None) -> Tuple[Any, Any]: ...

    def seed(self, seed: int) -> None: ...

    def render(self, *args: Any, **kwargs: Any) -> Any: ...

    def sample_action(self) -> torch.Tensor: ...

    def get_state(self) -> Any: ...

    def set_state(self, state: Any) -> None: ...

    def train(self) -> None: ...

    def val(self) -> None: ...

    def test(self) -> None: ...

    def save_gif(self, filename: str, output_path: Optional[Path] = None) -> None: ...

    def load_initial_domain(self, idx: int, mode: Optional[EnvMode] = None) -> None: ...

    def get_uncontrolled_episode_metrics(self) -> Any: ...

    def detach(self) -> None: ...


# the member names of the reference protocol, for the conformance test
FLUID_ENV_LIKE_MEMBERS = (
    "action_space", "observation_space", "differentiable", "use_marl", "metrics", "episode_length", "n_agents", "cuda_device",
    "step", "reset", "seed", "render", "sample_action", "get_state", "set_state", "train", "val", "test", "save_gif",
    "load_initial_domain", "get_uncontrolled_episode_metrics", "detach",
)
