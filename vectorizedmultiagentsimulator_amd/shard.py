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


class NativeRollout:
    """One shard's K-step rollout in ONE byte buffer laid out as the step kernel writes it: the per-step outputs of
    ``Environment.rollout`` (``env.rollout_fields(K)``: obs ``[K, A, b, D]``, rew ``[K, A, b]``, done ``[K, b]`` + the scenario's
    info terms) side by side at 256-byte aligned offsets.  ``fields`` are typed views of the buffer: passed to
    ``env.rollout(actions, out=nr.fields)`` the kernel's stores land in it directly - no copy between the fast rollout and
    the collective - and ``gather()`` is ONE ``all_gather_into_tensor`` of the buffer (the pipeline's only
    exchange).  Ranks own contiguous blocks of environments, so rank order IS environment order: the gathered tensors
    carry it as a leading axis, ``obs [R, K, A, b, D]`` with global environment ``r * b + e`` - views of the gathered
    buffer, nothing is moved (``env_major`` makes the flat ``[K, A, B, D]`` copy for consumers that insist on it).
    Shards of unequal size: the offsets are those of the largest shard on every rank, a rank's tensors cover its own
    ``local_envs`` and ``gather()`` returns a list of per-rank views instead of one stacked view."""

    def __init__(self, shard: EnvShard, fields, device, env_axis: Optional[Dict[str, int]] = None,
                 group: Optional[dist.ProcessGroup] = None):
        """``fields``: (name, shape, dtype) for THIS rank's ``shard.local_envs`` environments (``env.rollout_fields(K)``);
        ``env_axis[name]``: which axis of ``shape`` is the environment axis (default: axis 2 of a four-dimensional output -
        ``[K, A, b, D]`` -, else the last - ``[K, b]``, ``[K, A, b]``: what the shipped scenarios write)."""
        self.shard, self.group = shard, group
        b, bmax = shard.local_envs, shard.max_local_envs
        self.layout = []
        off = 0
        for name, shape, dtype in fields:
            shape = tuple(int(x) for x in shape)
            ax = (env_axis or {}).get(name)
            if ax is None:
                ax = 2 if len(shape) == 4 else len(shape) - 1
            assert shape[ax] == b, f"{name}: axis {ax} of {shape} is not the shard's {b} environments"
            item = torch.empty((), dtype=dtype).element_size()
            full = shape[:ax] + (bmax,) + shape[ax + 1:]
            n_full = item
            for x in full:
                n_full *= x
            self.layout.append((name, shape, dtype, ax, off, item))
            off += (n_full + 255) // 256 * 256
        self.nbytes = max(off, 256)
        self.local = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.fields: Dict[str, torch.Tensor] = {}
        for name, shape, dtype, ax, o, item in self.layout:
            n = item
            for x in shape:
                n *= x
            self.fields[name] = self.local[o:o + n].view(dtype).view(shape)
        self._full: Optional[torch.Tensor] = None
        #: a single-rank group normally skips the collective (its result IS the local buffer); True makes the call anyway -
        #: the one-GPU test boxes then run backend `nccl` (= RCCL) on the very buffer the step kernel stored into
        self.force_collective = False

    @classmethod
    def for_env(cls, shard: EnvShard, env, n_steps: int, group: Optional[dist.ProcessGroup] = None) -> "NativeRollout":
        assert env.num_envs == shard.local_envs, f"the environment has {env.num_envs} envs, the shard {shard.local_envs}"
        return cls(shard, env.rollout_fields(n_steps), env.device, group=group)

    def _rank_fields(self, buf: torch.Tensor, r: int) -> Dict[str, torch.Tensor]:
        sh = self.shard
        lo, hi = shard_range(sh.num_envs, r, sh.world_size)
        out = {}
        for name, shape, dtype, ax, o, item in self.layout:
            shp = shape[:ax] + (hi - lo,) + shape[ax + 1:]
            n = item
            for x in shp:
                n *= x
            out[name] = buf[o:o + n].view(dtype).view(shp)
        return out

    def gather(self):
        """Every rank's rollout on every rank - ONE collective on the caller's stream, its output buffer re-used by the next
        call.  Equal shards: ``{name: tensor [R, *shape]}`` (views of the gathered buffer); unequal: ``{name: [R tensors]}``."""
        sh = self.shard
        R = sh.world_size
        if R == 1 and not self.force_collective:
            return {k: v.unsqueeze(0) for k, v in self.fields.items()}
        if self._full is None:
            self._full = torch.empty(R * self.nbytes, dtype=torch.uint8, device=self.local.device)
        dist.all_gather_into_tensor(self._full, self.local, group=self.group)
        rows = self._full.view(R, self.nbytes)
        if sh.num_envs % R == 0:
            out = {}
            for name, shape, dtype, ax, o, item in self.layout:
                n = item
                for x in shape:
                    n *= x
                out[name] = rows[:, o:o + n].view(dtype).view((R,) + shape)  # (256-byte aligned ranges: re-typed in place)
            return out
        per_rank = [self._rank_fields(rows[r], r) for r in range(R)]
        return {name: [pr[name] for pr in per_rank] for name in self.fields}

    def env_major(self, gathered, name: str) -> torch.Tensor:
        """``gathered[name]`` with the global environment axis in the field's own position (``obs [K, A, B, D]``): the one
        copy, for consumers that want the reference's flat batch."""
        ax = next(l[3] for l in self.layout if l[0] == name)
        g = gathered[name]
        if isinstance(g, list):
            return torch.cat(g, dim=ax)
        x = g.movedim(0, ax)  # [..., R, b, ...]
        return x.reshape(x.shape[:ax] + (x.shape[ax] * x.shape[ax + 1],) + x.shape[ax + 2:])


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
