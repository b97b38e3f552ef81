"""Multi-GPU driver: groups are partitioned across ranks (one process per GPU), every rank fits its own
groups with no data-path communication, and one all-gather of the fixed-size per-group records assembles
the full result on every rank.  `torch.distributed` backend "nccl" is RCCL over xGMI on
ROCm; "gloo" is used by the CPU tests of the partition / gather plumbing.

Partitioning: the synthetic bench gives rank r the contiguous range [r*ceil(G/W), (r+1)*ceil(G/W)) of the *sorted
distinct keys*; arriving rows are routed with `hash64(key) % W` (`hash_partition`, mirrored by the C++ ingest of
duckdb_shim/sharded_arena.hpp: W device states, per-shard page-locked buffers).  Any assignment that keeps all rows of a
key on one shard works, because no arithmetic crosses groups (ols_aggregate.cpp:257-337 loops groups independently).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def hash64(keys):
    """splitmix64 finaliser over uint64 keys (numpy array or int) — the routing hash of the hash-partitioned ingest,
    bit-identical to anofox_shim::hash64 (duckdb_shim/sharded_arena.hpp)."""
    import numpy as np
    z = np.asarray(keys).astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z += np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def hash_partition(keys, world: int):
    """shard[i] = hash64(keys[i]) % world: which rank / device owns the group of row i.  All rows of a key land on one
    shard, so no arithmetic crosses shards (BASELINE north_star: groups hash-partitioned across the GPUs)."""
    import numpy as np
    return (hash64(keys) % np.uint64(world)).astype(np.int64)


def shard_range(n_groups: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous group range of `rank`; ranges are equal-sized (the last ones may be short or empty)."""
    per = (n_groups + world - 1) // world
    lo = min(n_groups, rank * per)
    hi = min(n_groups, lo + per)
    return lo, hi


def padded_shard_len(n_groups: int, world: int) -> int:
    return (n_groups + world - 1) // world


def gather_records(local: torch.Tensor, n_groups: int, group: Optional[dist.ProcessGroup] = None,
                   out: Optional[torch.Tensor] = None, async_op: bool = False, padded: Optional[torch.Tensor] = None):
    """All-gather the per-rank record blocks [G_local, L] into [n_groups, L] on every rank.

    Shards are padded to ceil(G/W) rows so that one `all_gather_into_tensor` moves everything.  `padded`
    (optional): a [ceil(G/W), L] buffer whose first G_local rows ARE `local` (same storage) and whose other rows
    are already NaN — then nothing is allocated or copied here (ShardedBatchFit keeps such buffers per slot)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    L = local.shape[1]
    per = padded_shard_len(n_groups, world)
    lo, hi = shard_range(n_groups, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} records, expected {hi - lo}")
    if padded is not None:
        if padded.shape != (per, L) or padded.data_ptr() != local.data_ptr():
            raise ValueError("`padded` must be the [ceil(G/W), L] buffer that `local` is the head of")
        local = padded
    elif local.shape[0] != per:
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

    Two things overlap with the HBM-bound accumulate kernel of the NEXT step instead of serialising behind it:
      * the all-gather of a step's records (RCCL's own stream, xGMI): the record buffers are multi-buffered and a
        gather is only waited for when its buffers are reused or in `finish()`;
      * the small latency-bound kernels of a step (solve, refinement): consecutive steps alternate between `depth`
        contexts, each with its own workspace and its own torch stream, so step k+1's accumulate kernel starts
        while step k's solve is still running.
    Every step's work is enqueued when `fit()` returns; `finish()` makes the calling stream wait for all of it."""

    def __init__(self, ctx, n_groups_total: int, group: Optional[dist.ProcessGroup] = None, depth: int = 2):
        self.ctx = ctx
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_groups_total = n_groups_total
        self.lo, self.hi = shard_range(n_groups_total, self.rank, self.world)
        self.depth = depth
        self._slot = 0
        self._bufs = [dict(core=None, inf=None, core_pad=None, inf_pad=None, out=None, out_inf=None, work=[], ctx=None,
                           stream=None, acc_done=None)
                      for _ in range(depth)]
        self._last_acc = None  # event recorded after the most recent accumulate kernel

    def _wait(self, b):
        for w in b["work"]:
            if w is not None:
                w.wait()
        b["work"] = []

    def last_stream(self):
        """The stream the most recent `fit` was enqueued on (None before the first one): an event recorded there marks the
        end of that step's kernels."""
        b = self._bufs[(self._slot - 1) % self.depth]
        return b["stream"]

    def close(self):
        """Finish and give the pipeline slots' own contexts (streams, workspaces) back; the caller's context stays.  (An idle
        context is not free: with the slots' streams alive, host-to-device copies on another stream of the process ran at 47.6
        instead of 55 GB/s — scripts/e2e_bisect.py.)"""
        self.finish()
        torch.cuda.synchronize()
        for b in self._bufs:
            c = b.get("ctx")
            if c is not None and c is not self.ctx:
                c.close()
            b["ctx"] = None
            b["stream"] = None
            b["acc_done"] = None

    def contexts(self):
        """The contexts in use (for timing collection)."""
        return [b["ctx"] for b in self._bufs if b["ctx"] is not None]

    def fit(self, row_offsets, y, x_cols, w, options):
        """Inputs hold this rank's groups only.  Returns (core_all[G, p+6], inf_all or None); the returned tensors
        are complete after `finish()` (or once their slot is reused)."""
        b = self._bufs[self._slot]
        self._slot = (self._slot + 1) % self.depth
        p = len(x_cols)
        G_local = int(row_offsets.numel()) - 1
        per = padded_shard_len(self.n_groups_total, self.world)
        dev = y.device
        caller = torch.cuda.current_stream(dev)
        if b["ctx"] is None:
            first = all(o["ctx"] is None for o in self._bufs)
            b["ctx"] = self.ctx if first else type(self.ctx)(dev.index)
            b["stream"] = torch.cuda.Stream(device=dev)
            b["acc_done"] = torch.cuda.Event()
        shape = (G_local, p, bool(options.compute_inference))
        if b.get("shape") != shape:
            self._wait(b)
            b["shape"] = shape
            # record buffers padded to the all-gather's shard length once, here: the fit writes the first G_local
            # rows, the padding rows stay NaN, and no per-step allocation or copy sits between fit and gather
            rows = per if self.world > 1 else G_local
            b["core_pad"] = torch.full((rows, p + 6), float("nan"), dtype=torch.float64, device=dev)
            b["core"] = b["core_pad"][:G_local]
            b["inf_pad"] = b["inf"] = b["out"] = b["out_inf"] = None
            if options.compute_inference:
                b["inf_pad"] = torch.full((rows, 5 * p + 2), float("nan"), dtype=torch.float64, device=dev)
                b["inf"] = b["inf_pad"][:G_local]
            if self.world > 1:
                b["out"] = torch.empty((per * self.world, p + 6), dtype=torch.float64, device=dev)
                if options.compute_inference:
                    b["out_inf"] = torch.empty((per * self.world, 5 * p + 2), dtype=torch.float64, device=dev)
        s = b["stream"]
        s.wait_stream(caller)  # the inputs were produced on the caller's stream
        with torch.cuda.stream(s):
            self._wait(b)      # the gather that last read these buffers
            # the accumulate kernels of consecutive steps run one after the other (both are HBM-bound); only the
            # solve / refinement tail of step k overlaps the accumulate kernel of step k + 1
            b["acc_done"].record(s)  # materialise the event handle before the library records into it
            # (the gate is one-shot: the library drops both handles inside the fit call below, so neither a destroyed
            # event nor a stale one is ever waited for or recorded by a later direct use of the context)
            b["ctx"].set_accumulate_gate(self._last_acc, b["acc_done"])
            self._last_acc = b["acc_done"]
            core, inf = b["ctx"].fit_batch_device(row_offsets, y, x_cols, w, options, core=b["core"], inference=b["inf"])
            if self.world == 1:
                return core, inf
            all_core, wk = gather_records(core, self.n_groups_total, self.group, out=b["out"], async_op=True,
                                          padded=b["core_pad"])
            b["work"].append(wk)
            all_inf = None
            if inf is not None:
                all_inf, wk2 = gather_records(inf, self.n_groups_total, self.group, out=b["out_inf"], async_op=True,
                                              padded=b["inf_pad"])
                b["work"].append(wk2)
        return all_core, all_inf

    def prepare(self, row_offsets, y, x_cols, w, options):
        """Create every slot's context, stream, buffers and workspace now (one fit per slot), so that no allocation
        happens inside a timed or latency-sensitive region later."""
        for _ in range(self.depth):
            self.fit(row_offsets, y, x_cols, w, options)
        self.finish()
        torch.cuda.synchronize(y.device)

    def finish(self):
        """Make the calling stream wait for every outstanding step and gather (call before reading results)."""
        for b in self._bufs:
            if b["stream"] is None:
                continue
            with torch.cuda.stream(b["stream"]):
                self._wait(b)
            torch.cuda.current_stream(b["stream"].device).wait_stream(b["stream"])
