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
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal path (several ranks sharing one GPU, CPU tests): gloo moves host memory
        host = torch.empty((per * world, L), dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=group)
        out.copy_(host)
        work = None
    else:
        work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
    res = out[:n_groups]
    return (res, work) if async_op else res


class ShardedBatchFit:
    """One rank's part of a sharded GROUP BY fit on device-resident grouped columns.

    The record buffers are double buffered and the all-gather of step k is only waited for when its buffers are
    reused (step k+2) or in `finish()`, so the gather (RCCL's own stream, xGMI) overlaps the next step's
    accumulate kernel (HBM-bound) instead of serialising behind it."""

    def __init__(self, ctx, n_groups_total: int, group: Optional[dist.ProcessGroup] = None, depth: int = 2):
        self.ctx = ctx
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_groups_total = n_groups_total
        self.lo, self.hi = shard_range(n_groups_total, self.rank, self.world)
        self.depth = depth
        self._slot = 0
        self._bufs = [dict(core=None, inf=None, out=None, out_inf=None, work=[]) for _ in range(depth)]

    def _wait(self, b):
        for w in b["work"]:
            if w is not None:
                w.wait()
        b["work"] = []

    def fit(self, row_offsets, y, x_cols, w, options):
        """Inputs hold this rank's groups only.  Returns (core_all[G, p+6], inf_all or None); with several
        ranks the returned tensors are complete after `finish()` (or once their slot is reused)."""
        if self.world == 1:
            return self.ctx.fit_batch_device(row_offsets, y, x_cols, w, options)
        b = self._bufs[self._slot]
        self._slot = (self._slot + 1) % self.depth
        self._wait(b)  # the gather that last read these buffers
        p = len(x_cols)
        G_local = int(row_offsets.numel()) - 1
        per = padded_shard_len(self.n_groups_total, self.world)
        dev = y.device
        if b["core"] is None:
            b["core"] = torch.empty((G_local, p + 6), dtype=torch.float64, device=dev)
            b["out"] = torch.empty((per * self.world, p + 6), dtype=torch.float64, device=dev)
            if options.compute_inference:
                b["inf"] = torch.empty((G_local, 5 * p + 2), dtype=torch.float64, device=dev)
                b["out_inf"] = torch.empty((per * self.world, 5 * p + 2), dtype=torch.float64, device=dev)
        core, inf = self.ctx.fit_batch_device(row_offsets, y, x_cols, w, options, core=b["core"], inference=b["inf"])
        all_core, wk = gather_records(core, self.n_groups_total, self.group, out=b["out"], async_op=True)
        b["work"].append(wk)
        all_inf = None
        if inf is not None:
            all_inf, wk2 = gather_records(inf, self.n_groups_total, self.group, out=b["out_inf"], async_op=True)
            b["work"].append(wk2)
        return all_core, all_inf

    def finish(self):
        """Wait for every outstanding gather (makes the current stream wait; call before reading results)."""
        for b in self._bufs:
            self._wait(b)
