"""Multi-agent observation windows, batched over envs.

Same results as the reference's ``envs/util/obs_extraction.py`` (``extract_moving_window_2d`` :194-238,
``extract_moving_window_2d_x_z`` :241-343, ``extract_moving_window_3d`` :346-411) -- pinned by
``tests/golden/reference_obs_windows.npz`` -- but written as one gather per call instead of a Python loop of
``torch.roll`` per agent, and with any number of leading (env) axes.
"""
from __future__ import annotations

import torch


_INDEX_CACHE: dict = {}


def _window_index(n_agents: int, window: int, pad: int, device) -> torch.Tensor:
    """``idx[i, k] = (i - pad + k) mod n``: the agents seen by agent ``i`` (circular).  Static per layout: made once per
    (layout, device) -- an env step of the turbulent-channel ids asked for it 24 times (four launches each)."""
    key = (int(n_agents), int(window), int(pad), str(device))
    idx = _INDEX_CACHE.get(key)
    if idx is None:
        i = torch.arange(n_agents, device=device).view(-1, 1)
        k = torch.arange(window, device=device).view(1, -1)
        idx = _INDEX_CACHE[key] = (i - pad + k) % n_agents
    return idx


def extract_moving_window_2d(field: torch.Tensor, n_agents: int, agent_width: int, n_agents_per_window: int) -> torch.Tensor:
    """``field [..., Y, X]`` (agents in a row along X) -> ``[..., n_agents, Y, window * agent_width]``."""
    *lead, Y, X = field.shape
    if X != n_agents * agent_width:
        raise ValueError("X must equal n_agents * agent_width")
    fa = field.reshape(*lead, Y, n_agents, agent_width)
    idx = _window_index(n_agents, n_agents_per_window, n_agents_per_window // 2, field.device)   # [n, W]
    win = fa[..., idx, :]                                        # [..., Y, n, W, aw]
    win = win.movedim(-3, -4)                                    # [..., n, Y, W, aw]
    return win.reshape(*lead, n_agents, Y, n_agents_per_window * agent_width)


def extract_moving_window_2d_x_z(field: torch.Tensor, n_agents_x: int, n_agents_z: int, agent_width: int,
                                 n_agents_per_window_x: int, n_agents_per_window_z: int, pad_x: int, pad_z: int) -> torch.Tensor:
    """``field [..., Z, X]`` -> per-agent patch means in circular windows, ``[..., n_agents_x * n_agents_z, Wz, Wx]``
    (x-agent outer, z-agent inner, like the reference)."""
    *lead, Z, X = field.shape
    if X != n_agents_x * agent_width or Z != n_agents_z * agent_width:
        raise ValueError("field must be [n_agents_z * agent_width, n_agents_x * agent_width]")
    if pad_x < 0 or pad_x > n_agents_per_window_x:
        raise ValueError("pad_x must be in range [0, n_agents_per_window_x]")
    if pad_z < 0 or pad_z > n_agents_per_window_z:
        raise ValueError("pad_z must be in range [0, n_agents_per_window_z]")
    means = field.reshape(*lead, n_agents_z, agent_width, n_agents_x, agent_width).mean(dim=(-3, -1))   # [..., nz, nx]
    iz = _window_index(n_agents_z, n_agents_per_window_z, pad_z, field.device)   # [nz, Wz]
    ix = _window_index(n_agents_x, n_agents_per_window_x, pad_x, field.device)   # [nx, Wx]
    win = means[..., iz[:, None, :, None], ix[None, :, None, :]]                 # [..., nz, nx, Wz, Wx]
    win = win.movedim(-3, -4)                                                    # [..., nx, nz, Wz, Wx]
    return win.reshape(*lead, n_agents_x * n_agents_z, n_agents_per_window_z, n_agents_per_window_x)


def extract_moving_window_3d(field: torch.Tensor, n_agents: int, agent_width: int, n_agents_per_window: int) -> torch.Tensor:
    """``field [..., Z, Y, X]`` (agents on an n x n lattice in Z, X) -> ``[..., n * n, W * aw, Y, W * aw]``
    (z-agent outer, x-agent inner)."""
    *lead, Z, Y, X = field.shape
    if X != n_agents * agent_width or Z != n_agents * agent_width:
        raise ValueError("X and Z must equal n_agents * agent_width")
    W, aw, n = n_agents_per_window, agent_width, n_agents
    fa = field.reshape(*lead, n, aw, Y, n, aw)                   # [..., nz, awz, Y, nx, awx]
    idx = _window_index(n, W, W // 2, field.device)              # [n, W]
    win = fa[..., idx, :, :, :, :]                               # [..., nz, Wz, awz, Y, nx, awx]
    win = win[..., idx, :]                                       # [..., nz, Wz, awz, Y, nx, Wx, awx]
    win = win.movedim(-3, -6)                                    # [..., nz, nx, Wz, awz, Y, Wx, awx]
    return win.reshape(*lead, n * n, W * aw, Y, W * aw)
