"""Global settings object with the surface of the reference's ``fluidgym.config`` (``config.py:43-157``): where the on-disk data
(initial domains) lives and the default dtype.  ``config.update("local_data_path", path)`` / ``config["local_data_path"] = path``
is how the reference's users point the envs at their data directory; ``$FLUIDGYM_DATA_PATH`` gives the start value here, the
reference's per-user data directory (``platformdirs.user_data_dir("FluidGym", "fluidgym")``) otherwise.  The HuggingFace repository
id is kept as information only: nothing in this package downloads."""
from __future__ import annotations

import os
from pathlib import Path
from typing import Any, Dict, List, Optional

import torch

_KEYS = ("hf_intial_domains_repo_id", "local_data_path", "dtype")        # (the reference's spelling of the first key)
_DTYPES = {"FP32": torch.float32, "FP64": torch.float64}
_PALETTE = ["#003a7d", "#008dff", "#ff73b6", "#ff9d3a", "#4ecb8d", "#f9e858", "#d83034", "#c701ff"]


def _default_data_path() -> Path:
    root = os.environ.get("FLUIDGYM_DATA_PATH")
    return Path(root) if root else Path.home() / ".local" / "share" / "FluidGym"


class Config:
    def __init__(self) -> None:
        self.settings: Dict[str, Any] = {"hf_intial_domains_repo_id": "safe-autonomous-systems/fluidgym-data",
                                         "local_data_path": None, "dtype": torch.float32}

    @staticmethod
    def _key(key: str) -> str:
        if key not in _KEYS:
            raise ValueError(f"Key '{key}' is not a valid configuration key. Allowed keys are: {list(_KEYS)}")
        return key

    def update(self, key: str, value: str) -> None:
        key = self._key(key)
        if key == "dtype":
            if value not in _DTYPES:
                raise ValueError(f"Value '{value}' is not a valid data type. Allowed values are:{list(_DTYPES)}")
            self.settings[key] = _DTYPES[value]
        elif key == "local_data_path":
            self.settings[key] = Path(value).resolve()
        else:
            raise ValueError(f"Unhandled configuration key: {key}")

    def get(self, key: str) -> Optional[Any]:
        key = self._key(key)
        return self.local_data_path if key == "local_data_path" else self.settings.get(key)

    def __getitem__(self, key: str) -> Any:
        return self.get(key)

    def __setitem__(self, key: str, value: str) -> None:
        self.update(key, value)

    @property
    def hf_intial_domains_repo_id(self) -> str:
        return self.settings["hf_intial_domains_repo_id"]

    @property
    def local_data_path(self) -> Path:
        # not set explicitly: follows $FLUIDGYM_DATA_PATH at the time of the call (tests and launch scripts set it late)
        p = self.settings["local_data_path"]
        return _default_data_path() if p is None else p

    @property
    def initial_domains_path(self) -> Path:
        return self.local_data_path / "initial_domains"

    @property
    def dtype(self) -> torch.dtype:
        return self.settings["dtype"]

    @property
    def palette(self) -> List[str]:
        return list(_PALETTE)


config = Config()
