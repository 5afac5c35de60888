"""``gymnasium.spaces`` when available, otherwise the two classes the env surface needs
(gymnasium is not part of the MI355X image; the reference imports it at ``envs/fluid_env.py:14``)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - exercised only where gymnasium is installed
    from gymnasium.spaces import Box, Dict  # type: ignore
except Exception:  # noqa: BLE001

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.shape(low)
            self.shape = tuple(int(s) for s in shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Dict(dict):
        def __init__(self, spaces=None, **kw):
            super().__init__(spaces or {}, **kw)

        @property
        def spaces(self):
            return self

        def sample(self):
            return {k: v.sample() for k, v in self.items()}

        def seed(self, seed=None):
            for i, v in enumerate(self.values()):
                v.seed(None if seed is None else seed + i)
