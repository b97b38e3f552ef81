"""Multi-GPU driver: groups are partitioned across ranks (one process per GPU), every rank fits its own
groups with no data-path communication, and one all-gather of the fixed-size per-group records assembles
the full result on every rank.  `torch.distributed` backend "nccl" is RCCL over xGMI on
ROCm; "gloo" is used by the CPU tests of the partition / gather plumbing.

Partitioning: rank r owns the contiguous range [r*ceil(G/W), (r+1)*ceil(G/W)) of the *sorted distinct
keys*; a DuckDB shim would route rows with `hash(key) % W` instead — any assignment that keeps all rows of
a key on one rank works, because no arithmetic crosses groups (ols_aggregate.cpp:257-337 loops groups
independently).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_groups: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous group range of `rank`; ranges are equal-sized (the last ones may be short or empty)."""
    per = (n_groups + world - 1) // world
    lo = min(n_groups, rank * per)
    hi = min(n_groups, lo + per)
    return lo, hi


def padded_shard_len(n_groups: int, world: int) -> int:
    return (n_groups + world - 1) // world


def gather_records(local: torch.Tensor, n_groups: int, group: Optional[dist.ProcessGroup] = None,
                   out: Optional[torch.Tensor] = None, async_op: bool = False):
    """All-gather the per-rank record blocks [G_local, L] into [n_groups, L] on every rank.

    Shards are padded to ceil(G/W) rows so that one `all_gather_into_tensor` moves everything."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    L = local.shape[1]
    per = padded_shard_len(n_groups, world)
    lo, hi = shard_range(n_groups, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} records, expected {hi - lo}")
    if local.shape[0] != per:
        pad = torch.full((per, L), float("nan"), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        local = pad
    if out is None:
        out = torch.empty((per * world, L), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
    res = out[:n_groups]
    return (res, work) if async_op else res


class ShardedBatchFit:
    """One rank's part of a sharded GROUP BY fit on device-resident grouped columns."""

    def __init__(self, ctx, n_groups_total: int, group: Optional[dist.ProcessGroup] = None):
        self.ctx = ctx
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_groups_total = n_groups_total
        self.lo, self.hi = shard_range(n_groups_total, self.rank, self.world)
        self._out = None
        self._out_inf = None

    def fit(self, row_offsets, y, x_cols, w, options):
        """Inputs hold this rank's groups only.  Returns (core_all[G, p+6], inf_all or None)."""
        core, inf = self.ctx.fit_batch_device(row_offsets, y, x_cols, w, options)
        if self.world == 1:
            return core, inf
        per = padded_shard_len(self.n_groups_total, self.world)
        if self._out is None or self._out.shape != (per * self.world, core.shape[1]):
            self._out = torch.empty((per * self.world, core.shape[1]), dtype=core.dtype, device=core.device)
        all_core = gather_records(core, self.n_groups_total, self.group, out=self._out)
        all_inf = None
        if inf is not None:
            if self._out_inf is None or self._out_inf.shape != (per * self.world, inf.shape[1]):
                self._out_inf = torch.empty((per * self.world, inf.shape[1]), dtype=inf.dtype, device=inf.device)
            all_inf = gather_records(inf, self.n_groups_total, self.group, out=self._out_inf)
        return all_core, all_inf
