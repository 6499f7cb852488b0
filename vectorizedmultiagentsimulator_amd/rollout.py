"""Rollout collection on one shard + the end-of-rollout gather across GPUs.

``collect(env, policy, T)`` steps a (sharded) environment T times with ``policy(obs) -> actions``
and stacks observations / rewards / dones into ``[T, b, ...]`` buffers on the device;
``gather_rollout`` then all-gathers them over RCCL/xGMI so that every rank holds the global
``[T, B, ...]`` rollout (the ONLY collective of the whole pipeline).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
from torch import Tensor

from .shard import EnvShard, RolloutGather


def collect(env, policy: Callable[[List[Tensor]], List[Tensor]], n_steps: int,
            obs: Optional[List[Tensor]] = None, auto_reset: bool = False) -> Dict[str, Tensor]:
    """Returns {"obs": [T, b, A, D], "rew": [T, b, A], "done": [T, b]} for this shard.  ``auto_reset``:
    environments that report done are restarted in place (``Environment.reset_where``, no host sync) and
    the policy sees their fresh observation at the next step; obs[t] is what step t returned."""
    if obs is None:
        obs = env.reset()
    A = len(env.agents)
    b = env.num_envs
    D = obs[0].shape[-1]
    dev = obs[0].device
    out = {
        "obs": torch.empty(n_steps, b, A, D, device=dev, dtype=torch.float32),
        "rew": torch.empty(n_steps, b, A, device=dev, dtype=torch.float32),
        "done": torch.empty(n_steps, b, device=dev, dtype=torch.bool),
    }
    for t in range(n_steps):
        actions = policy(obs)
        obs, rews, dones, _ = env.step(actions)
        out["obs"][t] = torch.stack(obs, dim=1)
        out["rew"][t] = torch.stack(rews, dim=1)
        out["done"][t] = dones
        if auto_reset:
            obs = env.reset_where(dones)
    return out


def gather_rollout(buffers: Dict[str, Tensor], shard: EnvShard) -> Dict[str, Tensor]:
    """Global rollout on every rank (env axis = dim 1, global environment order)."""
    return RolloutGather(shard).gather(buffers, env_dim=1)
