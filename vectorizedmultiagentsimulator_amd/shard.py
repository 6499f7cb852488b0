"""Multi-GPU: shard the environment batch across ranks, gather rollouts at the end.

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in
the CPU tests).  Environments are independent, so every rank owns a contiguous block of
``num_envs`` and steps it with its own ``HipWorld`` - NO collective on the step path.  The only
exchange is the end-of-rollout gather of observation / reward / done buffers: an
all-gather over the fully connected xGMI mesh puts one peer shard on each link, i.e. its time is
about shard_bytes / 153 GB/s, independent of the number of GPUs.

Broad-phase semantics across shards: each shard behaves exactly like a reference environment of
its own size (the reference's batch-global ``.any()`` of core.py:2797-2801 is evaluated per
shard when ``exact_broad_phase`` is on, and not at all otherwise); no per-substep collective.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(num_envs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the global batch owned by ``rank``; sizes differ by at most 1."""
    assert 0 <= rank < world_size and num_envs >= 0
    base, rem = divmod(num_envs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@dataclass
class EnvShard:
    num_envs: int  # global batch
    rank: int
    world_size: int

    @property
    def lo(self) -> int:
        return shard_range(self.num_envs, self.rank, self.world_size)[0]

    @property
    def hi(self) -> int:
        return shard_range(self.num_envs, self.rank, self.world_size)[1]

    @property
    def local_envs(self) -> int:
        return self.hi - self.lo

    @property
    def max_local_envs(self) -> int:
        return -(-self.num_envs // self.world_size)

    @staticmethod
    def from_env(num_envs: int) -> "EnvShard":
        if dist.is_available() and dist.is_initialized():
            return EnvShard(num_envs, dist.get_rank(), dist.get_world_size())
        return EnvShard(num_envs, 0, 1)

    def seed(self, base_seed: int) -> int:
        """Per-shard seed: the global env index of the shard's first environment offsets it, so a
        1-GPU and an N-GPU run of the same global batch draw different but reproducible streams."""
        return base_seed * 1_000_003 + self.lo


class RolloutGather:
    """All-gather of per-shard rollout buffers whose environment axis is ``env_dim``.

    ``gather({"obs": [T, b, A, D], "rew": [T, b, A], "done": [T, b]}, env_dim=1)`` returns the
    same dict with the env axis of size ``num_envs`` (global order) on every rank.  One collective
    per tensor on the caller's stream; shards of unequal size are padded to the largest."""

    def __init__(self, shard: EnvShard, group: Optional[dist.ProcessGroup] = None):
        self.shard, self.group = shard, group

    def gather(self, buffers: Dict[str, torch.Tensor], env_dim: int = 1) -> Dict[str, torch.Tensor]:
        sh = self.shard
        if sh.world_size == 1:
            return dict(buffers)
        out = {}
        for name, t in buffers.items():
            assert t.shape[env_dim] == sh.local_envs, f"{name}: env axis {t.shape[env_dim]} != shard size {sh.local_envs}"
            # env axis first, contiguous: one flat block per rank
            x = t.movedim(env_dim, 0)
            was_bool = x.dtype == torch.bool
            if was_bool:
                x = x.to(torch.uint8)
            pad = sh.max_local_envs - sh.local_envs
            if pad:
                x = torch.cat([x, x.new_zeros((pad,) + tuple(x.shape[1:]))], dim=0)
            x = x.contiguous()
            full = x.new_empty((sh.world_size * sh.max_local_envs,) + tuple(x.shape[1:]))
            dist.all_gather_into_tensor(full, x, group=self.group)
            parts = []
            for r in range(sh.world_size):
                lo, hi = shard_range(sh.num_envs, r, sh.world_size)
                parts.append(full[r * sh.max_local_envs : r * sh.max_local_envs + (hi - lo)])
            g = torch.cat(parts, dim=0)
            if was_bool:
                g = g.to(torch.bool)
            out[name] = g.movedim(0, env_dim)
        return out


def max_over_ranks(value: float, device) -> float:
    """bench.py's timing reduction: the slowest rank defines the step time."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
