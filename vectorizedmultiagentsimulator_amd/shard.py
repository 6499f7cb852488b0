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


def _valid_rows(sh: "EnvShard", device) -> torch.Tensor:
    """Rows of a gathered ``[world_size * max_local_envs, ...]`` buffer that hold environments (unequal shards pad)."""
    rows = []
    for r in range(sh.world_size):
        lo, hi = shard_range(sh.num_envs, r, sh.world_size)
        rows.append(torch.arange(r * sh.max_local_envs, r * sh.max_local_envs + (hi - lo), device=device))
    return torch.cat(rows)


class PackedRollout:
    """One shard's rollout in ONE buffer laid out for the gather: ``[b, T, W]`` float32, environment axis first, where a
    step's record is ``W = A * D + A + 1`` numbers - the agents' observations, their rewards, done (0 / 1).  Every rank's
    buffer is one flat block, so the end-of-rollout exchange is ONE ``all_gather_into_tensor`` straight
    into the global ``[B, T, W]`` buffer: ranks own contiguous blocks of environments, i.e. rank order IS environment order
    - no pad / cat / movedim copies on either side (shards of unequal size: the buffer has ``max_local_envs`` rows and the
    holes are dropped by one index_select after the collective).  ``obs`` / ``rew`` / ``done`` are views."""

    def __init__(self, shard: EnvShard, n_steps: int, n_agents: int, obs_dim: int, device, group: Optional[dist.ProcessGroup] = None):
        self.shard, self.group = shard, group
        self.T, self.A, self.D = int(n_steps), int(n_agents), int(obs_dim)
        self.W = self.A * self.D + self.A + 1
        self.local = torch.zeros(shard.max_local_envs, self.T, self.W, device=device, dtype=torch.float32)
        self._full: Optional[torch.Tensor] = None

    def _views(self, buf: torch.Tensor, n_envs: int) -> Dict[str, torch.Tensor]:
        A, D = self.A, self.D
        x = buf[:n_envs]
        return {"obs": x[..., : A * D].unflatten(-1, (A, D)), "rew": x[..., A * D : A * D + A], "done": x[..., -1]}

    def views(self) -> Dict[str, torch.Tensor]:
        """This shard's ``obs [b, T, A, D]``, ``rew [b, T, A]``, ``done [b, T]`` (float 0 / 1): views of the buffer."""
        return self._views(self.local, self.shard.local_envs)

    def write(self, t: int, obs, rews, dones) -> None:
        """Step ``t`` of the rollout from what ``Environment.step`` returned (per-agent lists)."""
        v = self.views()
        v["obs"][:, t] = torch.stack(list(obs), dim=1)
        v["rew"][:, t] = torch.stack(list(rews), dim=1)
        v["done"][:, t] = dones.to(torch.float32)

    def gather(self) -> Dict[str, torch.Tensor]:
        """The global rollout on every rank: ``obs [B, T, A, D]``, ``rew [B, T, A]``, ``done [B, T]`` in global environment
        order - ONE collective on the caller's stream, its output buffer re-used by the next call."""
        sh = self.shard
        if sh.world_size == 1:
            return self.views()
        if self._full is None:
            self._full = torch.empty((sh.world_size * sh.max_local_envs, self.T, self.W), device=self.local.device, dtype=torch.float32)
        dist.all_gather_into_tensor(self._full, self.local, group=self.group)
        full = self._full
        if sh.num_envs % sh.world_size:  # unequal shards: drop the padding rows of the smaller ones
            full = full.index_select(0, _valid_rows(sh, full.device))
        return self._views(full, sh.num_envs)


class RolloutGather:
    """All-gather of per-shard rollout buffers whose environment axis is ``env_dim`` (any set of tensors).

    ``gather({"obs": [T, b, A, D], "rew": [T, b, A], "done": [T, b]}, env_dim=1)`` returns the same dict with the env axis
    of size ``num_envs`` (global order) on every rank.  ONE collective for the whole dict: every tensor is packed, env axis
    first, into its byte range of one ``[max_local_envs, bytes per environment]`` buffer (the only copy on the send side),
    and the results are views of the gathered buffer.  (``PackedRollout`` collects straight into that layout: no copy.)"""

    def __init__(self, shard: EnvShard, group: Optional[dist.ProcessGroup] = None):
        self.shard, self.group = shard, group

    def gather(self, buffers: Dict[str, torch.Tensor], env_dim: int = 1) -> Dict[str, torch.Tensor]:
        sh = self.shard
        if sh.world_size == 1:
            return dict(buffers)
        layout, off = [], 0
        dev = next(iter(buffers.values())).device
        for name, t in buffers.items():
            assert t.shape[env_dim] == sh.local_envs, f"{name}: env axis {t.shape[env_dim]} != shard size {sh.local_envs}"
            rest = tuple(t.movedim(env_dim, 0).shape[1:])
            n = 1
            for d in rest:
                n *= d
            nbytes = n * t.element_size()
            off = (off + 15) // 16 * 16  # 16-byte aligned ranges: the views can be re-typed
            layout.append((name, off, nbytes, rest, t.dtype))
            off += nbytes
        row = (off + 15) // 16 * 16
        send = torch.zeros(sh.max_local_envs, row, dtype=torch.uint8, device=dev)
        for (name, o, nbytes, rest, dtype), t in zip(layout, buffers.values()):
            x = t.movedim(env_dim, 0)
            send[: sh.local_envs, o : o + nbytes].view(torch.uint8).copy_(
                (x.to(torch.uint8) if dtype == torch.bool else x).contiguous().view(sh.local_envs, -1).view(torch.uint8))
        full = torch.empty(sh.world_size * sh.max_local_envs, row, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(full, send, group=self.group)
        if sh.num_envs % sh.world_size:
            full = full.index_select(0, _valid_rows(sh, dev))
        out = {}
        for name, o, nbytes, rest, dtype in layout:
            raw = full[:, o : o + nbytes]
            if dtype == torch.bool:
                g = raw.view(sh.num_envs, *rest).to(torch.bool)
            else:  # (16-byte aligned rows and ranges: the strided byte slice re-types in place)
                g = raw.view(dtype).view(sh.num_envs, *rest)
            out[name] = g.movedim(0, env_dim)
        return out


def max_over_ranks(value: float, device) -> float:
    """bench.py's timing reduction: the slowest rank defines the step time."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
