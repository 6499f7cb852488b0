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

from .shard import EnvShard, NativeRollout, PackedRollout, RolloutGather


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


def collect_packed(env, policy: Callable[[List[Tensor]], List[Tensor]], n_steps: int, shard: EnvShard,
                   obs: Optional[List[Tensor]] = None, auto_reset: bool = False) -> PackedRollout:
    """``collect`` straight into the gather's layout (``PackedRollout``: one ``[b, T, W]`` buffer, environment axis first):
    ``.views()`` are this shard's ``obs [b, T, A, D]`` / ``rew [b, T, A]`` / ``done [b, T]``, ``.gather()`` the global rollout
    on every rank with ONE collective and no copies around it."""
    if obs is None:
        obs = env.reset()
    pr = PackedRollout(shard, n_steps, len(env.agents), obs[0].shape[-1], obs[0].device)
    for t in range(n_steps):
        obs, rews, dones, _ = env.step(policy(obs))
        pr.write(t, obs, rews, dones)
        if auto_reset:
            obs = env.reset_where(dones)
    return pr


def collect_native(env, actions: List[Tensor], shard: EnvShard, into: Optional[NativeRollout] = None) -> NativeRollout:
    """K steps with PRE-COMPUTED actions (``actions[i]``: agent i's ``[K, b, action_size]``) as ONE kernel launch
    (``Environment.rollout``) whose per-step outputs are stored straight into the buffer the end-of-rollout gather sends:
    ``.fields`` are this shard's ``obs [K, A, b, D]`` / ``rew [K, A, b]`` / ``done [K, b]`` (+ info terms), ``.gather()`` the
    rollouts of all ranks with ONE collective and no copy on either side (SURVEY.md 8e + 8f-3).  For the scenarios whose
    step is one launch; ``into``: a buffer of an earlier call (same K) to write again."""
    K = int(actions[0].shape[0])
    nr = into if into is not None else NativeRollout.for_env(shard, env, K)
    env.rollout(actions, out=nr.fields)
    return nr
