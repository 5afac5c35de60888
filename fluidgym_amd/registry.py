"""Environment registry with the reference's semantics (``fluidgym/registry.py:21-116``):
``register(id, entry_point, defaults, **overrides)``; ``make(id, **kwargs)`` merges kwargs over the
registered defaults (``:72``); duplicate ids and unknown ids raise ``ValueError``."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict, List


@dataclass
class EnvSpec:
    entry_point: Callable
    kwargs: Dict[str, Any]


class EnvRegistry:
    def __init__(self) -> None:
        self.env_specs: Dict[str, EnvSpec] = {}

    def register(self, id: str, entry_point: Callable, defaults: Dict[str, Any], **kwargs: Any) -> None:
        if id in self.env_specs:
            raise ValueError(f"Environment {id} is already registered.")
        self.env_specs[id] = EnvSpec(entry_point=entry_point, kwargs={**defaults, **kwargs})

    def make(self, id: str, **kwargs: Any):
        if id not in self.env_specs:
            raise ValueError(f"Environment {id} not found. Did you register it?")
        spec = self.env_specs[id]
        return spec.entry_point(**{**spec.kwargs, **kwargs})

    @property
    def ids(self) -> List[str]:
        return list(self.env_specs.keys())


registry = EnvRegistry()


def register(id: str, entry_point: Callable, defaults: Dict[str, Any], **kwargs: Any) -> None:
    registry.register(id, entry_point, defaults, **kwargs)


def make(id: str, **kwargs: Any):
    return registry.make(id, **kwargs)
