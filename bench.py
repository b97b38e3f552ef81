#!/usr/bin/env python3
"""bench.py — group fits/s of the grouped least-squares hot path on MI355X.

A "step" is one pass of the hot path over the whole synthetic GROUP BY: accumulate + solve on every
rank's groups and (N > 1) the all-gather of the per-group records.  Workload = BASELINE.json's metric
config: OLS, 1M groups x n = 1000 x p = 8, inputs resident in HBM before the timed region; with N ranks
the same 1M groups are partitioned across the ranks ("strong" scaling, BASELINE config 4).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (metric / roofline / cpu_baseline), see DESIGN.md "Measurement".
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

PKG = "anofox-statistics_amd"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
# FP64 matrix (v_mfma_f64_16x16x4_f64) dense peak.  The guide lists the FP32 matrix peak (157.3 TFLOP/s = the
# vector rate); FP64 MFMA runs at half of it on MI355X (AMD datasheet: 78.6 TFLOP/s FP64 matrix = FP64 vector).
FP64_MFMA_PEAK_TFLOPS = 78.6
# Measured on the MI355X boxes of this pool (not datasheet): a plain 16-B/lane streaming read reaches 6.05-6.39 TB/s
# (csrc/tools/hbm_read_rate, profiles/r01_hbm_read_rate.txt) and a pure v_mfma_f64_16x16x4_f64 loop whose instructions
# take DIFFERENT operand registers 68-72 TFLOP/s (csrc/tools/mfma_f64_rate, profiles/r02_mfma_f64_rate.txt; round 1's
# 47 TFLOP/s came from a loop that fed every instruction the same two registers).  Reported next to `peak`; `frac`
# stays achieved / peak.
MEASURED_STREAM_READ_GBS = 6390.0
MEASURED_MFMA_F64_TFLOPS = 72.0


def algorithmic_bytes_per_fit(n: int, p: int, weighted: bool, inference: bool) -> int:
    """8 n (p+1) [+ 8 n weights] input + 8 (p+6) core record [+ 8 (5p+2) inference]."""
    return 8 * n * (p + 1) + (8 * n if weighted else 0) + 8 * (p + 6) + (8 * (5 * p + 2) if inference else 0)


def algorithmic_flops_per_fit(n: int, p: int) -> float:
    """2 n (p'(p'+1)/2 + p' + 1) for the moment accumulation, p' = p + 1 (the O(p'^3) solve is
    not counted)."""
    pp = p + 1
    return 2.0 * n * (pp * (pp + 1) / 2 + pp + 1)


def sample_windows(G: int, sample: int):
    """Three windows of groups — head, middle and tail of the batch — `sample` groups in total (all of them when
    the batch is smaller): the tail window lies in the LAST slab of a multi-slab launch."""
    if G <= sample:
        return [(0, G)]
    k = max(1, sample // 3)
    mid = (G - k) // 2
    return [(0, k), (mid, mid + k), (G - (sample - 2 * k), G)]


def parity_gate(pkg, core, inf, offs, y, x_cols, w, model, kw, p, sample):
    """Re-check a sample of groups (head, middle and tail of this rank's batch) against the CPU oracle; returns
    (ok, max coef err, max diag err)."""
    import oracle  # checker only
    G = core.shape[0]
    ok_all, cerr, derr = True, 0.0, 0.0
    for g0, g1 in sample_windows(G, sample):
        so = offs[g0:g1 + 1].cpu().numpy()
        r0, r1 = int(so[0]), int(so[-1])
        ys = y[r0:r1].cpu().numpy()
        xs = [c[r0:r1].cpu().numpy() for c in x_cols]
        ws = w[r0:r1].cpu().numpy() if w is not None else None
        rcore, rinf = oracle.fit_groups(ys, xs, so - r0, w=ws, model=model,
                                        n_threads=len(os.sched_getaffinity(0)), **kw)
        c = core[g0:g1].cpu().numpy()
        if not np.array_equal(c[:, p + 5], rcore[:, p + 5]):
            return False, float("inf"), float("inf")
        scale = np.max(np.abs(rcore[:, :p + 1]), axis=1, keepdims=True)
        cerr = max(cerr, float(np.max(np.abs(c[:, :p + 1] - rcore[:, :p + 1]) / np.maximum(np.abs(rcore[:, :p + 1]), 1e-3 * scale))))
        derr = max(derr, float(np.max(np.abs(c[:, p + 1:p + 4] / rcore[:, p + 1:p + 4] - 1.0))))
        if rinf is not None:
            gi = inf[g0:g1].cpu().numpy()
            derr = max(derr, float(np.max(np.abs(gi - rinf) / np.maximum(np.abs(rinf), 1e-300))))
        ok_all = ok_all and (cerr < 1e-9 and derr < 1e-6)
    return ok_all, cerr, derr


def cpu_baseline(offs, y, x_cols, w, model, kw, n, p, budget_s=8.0):
    """Time the oracle's two restatements of the reference's algorithm classes on the host cores, on a bounded sample
    of the same workload (passes over the first S groups until ~budget_s of wall time each): dense Householder QR per
    group (`solver = qr`), and QR + SVD of the triangular factor (`solver = svd`, the aggregates' DEFAULT:
    ols_aggregate.cpp:51).  `value` is the SVD figure — what the reference's SQL surface runs unless told otherwise."""
    import oracle
    cores = len(os.sched_getaffinity(0))
    G = offs.numel() - 1
    # sample size: about one second of work per pass at ~1 GFLOP/s per core of dense QR (2 n p'^2 flops per fit)
    t_fit = 2.0 * n * (p + 1) ** 2 / 1e9
    S = int(max(min(G, cores), min(G, 65536, cores / max(t_fit, 1e-9))))
    n_rows = int(offs[S].item())
    ys = y[:n_rows].cpu().numpy()
    xs = [c[:n_rows].cpu().numpy() for c in x_cols]
    ws = w[:n_rows].cpu().numpy() if w is not None else None
    so = offs[:S + 1].cpu().numpy()

    def timed(**solver):   # the reference's algorithm class as it is: no refinement pass in the timed baseline
        k2 = dict(kw, **solver)
        oracle.fit_groups(ys, xs, so, w=ws, model=model, n_threads=cores, **k2)       # warm the pages
        passes, t0 = 0, time.perf_counter()
        while True:
            oracle.fit_groups(ys, xs, so, w=ws, model=model, n_threads=cores, **k2)
            passes += 1
            t = time.perf_counter() - t0
            if t >= budget_s or passes >= 200:
                return S * passes / t, passes, t

    qr, qr_passes, qr_t = timed(plain_qr=True)
    svd, svd_passes, svd_t = timed(plain_svd=True)
    what = f"the first {S} groups x {n} rows x p={p} ({model}) with oracle.fit_groups, {cores} threads"
    return {"value": svd, "unit": "fits/s", "cores": cores, "kind": "port",
            "sample": f"{svd_passes} passes over {what}: QR of the design + one-sided Jacobi SVD of R per group (solver = svd, "
                      f"the aggregates' default), {svd_t:.1f} s",
            "qr": {"value": qr, "unit": "fits/s",
                   "sample": f"{qr_passes} passes over {what}: dense Householder QR per group (solver = qr), {qr_t:.1f} s"}}


def end_to_end_leg(pkg, ctx, offs, y, x_cols, w, opts, model, kw, G, n, p, batch_groups=65536, chunk_rows=1 << 22, with_row_log=False):
    """SURVEY.md 8(d) "also report end-to-end including H2D separately": the path as the DuckDB aggregate drives it.  Rows
    lie in page-locked HOST memory in shuffled group order (row-major x, a slot number per row — the arena's chunk layout),
    stream over PCIe into the GPU-resident aggregate state (anofox_hip_agg_state_update_host, `chunk_rows` rows per call),
    and Finalize copies one record per group back to the host.  All G slots x n rows are streamed: every batch of
    `batch_groups` slots re-sends the same host block (the first `batch_groups` groups of the bench data, shuffled once) under
    new slot numbers, so the host needs 5 GB, not 72.  Timed: first Update -> records on the host.  Never part of `value`."""
    import ctypes as C
    import oracle  # checker only
    weighted = w is not None
    n_batches = (G + batch_groups - 1) // batch_groups
    B = min((G + n_batches - 1) // n_batches, offs.numel() - 1)      # equal batches where G allows (1M: 16 x 62 500)
    nr = int(offs[B].item())
    dev = y.device
    perm = torch.randperm(nr, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    gid = torch.searchsorted(offs[1:B + 1].contiguous(), perm, right=True).to(torch.int32)
    hx = torch.stack([c[:nr][perm] for c in x_cols], dim=1).contiguous().cpu().pin_memory()
    hy = y[:nr][perm].contiguous().cpu().pin_memory()
    hw = w[:nr][perm].contiguous().cpu().pin_memory() if weighted else None
    n_batches = (G + B - 1) // B
    hslots = [(gid + k * B).cpu().pin_memory() for k in range(n_batches)]     # batch k: the same rows under slots k B ...
    last = G - (n_batches - 1) * B         # the last batch may be partial: rows of groups >= `last` are skipped (valid = 0)
    hvalid_last = None
    if last < B:
        hslots[-1] = torch.where(gid < last, gid + (n_batches - 1) * B, torch.zeros_like(gid)).cpu().pin_memory()
        hvalid_last = (gid < last).to(torch.uint8).cpu().pin_memory()
    rows_valid_last = int(hvalid_last.sum()) if hvalid_last is not None else nr
    del perm
    abi = importlib.import_module(PKG + "._abi")
    best = None
    for _rep in range(2):
        # moments only: the state keeps no row log (a log lets Finalize refit the groups its moments cannot resolve; the DuckDB
        # arena keeps one by default, 64 GiB of HBM then 32 GiB of host memory — `with_row_log` below times that configuration)
        st = pkg.AggState(ctx, p, opts, initial_slots=G, retain_bytes=(None if with_row_log else 0))
        lib = st._lib
        err = abi.AnofoxError()
        ctx.synchronize()
        t0 = time.perf_counter()
        rows_sent = (n_batches - 1) * nr + (nr if hvalid_last is None else rows_valid_last)
        for k in range(n_batches):
            hs = hslots[k]
            for r0 in range(0, nr, chunk_rows):
                r1 = min(nr, r0 + chunk_rows)
                valid = hvalid_last[r0:r1].data_ptr() if (hvalid_last is not None and k == n_batches - 1) else None
                ok = lib.anofox_hip_agg_state_update_host(st._h, r1 - r0, G, hs[r0:r1].data_ptr(), hy[r0:r1].data_ptr(),
                                                          hx[r0:r1].data_ptr(), hw[r0:r1].data_ptr() if weighted else None,
                                                          valid, C.byref(err))
                if not ok:
                    raise RuntimeError(err.text())
        ctx.synchronize()
        t1 = time.perf_counter()
        core, _inf, unref = st.finalize()
        t2 = time.perf_counter()
        st.close()
        if best is None or (t2 - t0) < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1, core, int(unref), rows_sent)
    total, t_upd, t_fin, core, unref, rows_sent = best
    # parity: the first 256 slots of the LAST batch (their rows in arrival order) against the oracle
    S = min(256, last)
    g = gid.cpu().numpy().astype(np.int64)
    rows = np.nonzero(g < S)[0]
    order = rows[np.argsort(g[rows], kind="stable")]
    go = np.concatenate([[0], np.cumsum(np.bincount(g[rows], minlength=S))]).astype(np.int64)
    hxn, hyn = hx.numpy(), hy.numpy()
    rcore, _ = oracle.fit_groups(hyn[order], [np.ascontiguousarray(hxn[order, j]) for j in range(p)], go,
                                 w=hw.numpy()[order] if weighted else None, model=model,
                                 **{k: v for k, v in kw.items() if k != "compute_inference"})
    c = np.asarray(core)[(n_batches - 1) * B:(n_batches - 1) * B + S]
    scale = np.max(np.abs(rcore[:, :p + 1]), axis=1, keepdims=True)
    cerr = float(np.max(np.abs(c[:, :p + 1] - rcore[:, :p + 1]) / np.maximum(np.abs(rcore[:, :p + 1]), 1e-3 * scale)))
    derr = float(np.max(np.abs(c[:, p + 1:p + 4] / rcore[:, p + 1:p + 4] - 1.0)))
    ok = bool(np.array_equal(c[:, p + 5], rcore[:, p + 5]) and cerr < 1e-9 and derr < 1e-6)
    bytes_row = 8 * (p + 1) + (8 if weighted else 0) + 4
    return {"path": "anofox_hip_agg_state_update_host (page-locked host rows, shuffled group order) -> finalize (records on the host)",
            "fits_per_s": (G / total) if ok else None, "rows_per_s": rows_sent / total, "seconds": total,
            "update_seconds": t_upd, "finalize_seconds": t_fin, "GBps_pcie": rows_sent * bytes_row / t_upd / 1e9,
            "bytes_per_row_over_pcie": bytes_row, "groups": G, "rows_per_group": n, "rows_streamed": rows_sent,
            "batch_groups": B, "rows_per_update_call": chunk_rows, "groups_flagged_unrefined": unref,
            "row_log": ("HBM up to ANOFOX_HIP_RETAIN_BYTES (64 GiB), then page-locked host memory (32 GiB), then dropped: the arena's default"
                        if with_row_log else "none (moments only)"),
            "parity": {"ok": ok, "sample_groups": S, "max_coef_rel_err": cerr, "max_diag_rel_err": derr},
            "note": "PCIe-bound; reported beside `value`, never mixed into it (SURVEY.md 8d)"}


def multi_rank_audit(pkg, ctx, dmod, sharded, core_all, lo, hi, G, p, kt, steps, rehearsal, dev, stage=None):
    """Per-rank figures of an N-rank run, gathered to rank 0: each rank's accumulate-kernel time per step, the time of one
    all-gather of the records alone, and the number of ranks RCCL itself counts in a communicator created through the
    library's C ABI (anofox_hip_comm_create -> ncclCommCount) — over which the same records are gathered once more
    (anofox_hip_gather_records_device) and compared with torch.distributed's result."""
    import ctypes as C
    world, rank = dist.get_world_size(), dist.get_rank()
    stage = stage if stage is not None else [""]
    if os.environ.get("ANOFOX_BENCH_TEST_HANG_RANK") == str(rank):      # test hook of the watchdog path: this rank never joins
        stage[0] = "(test hook: rank held back on purpose)"
        time.sleep(10 ** 6)
    mine = {"rank": rank, "groups": hi - lo, "kernel_ms_per_step": kt["accumulate_ms"] / steps, "solve_span_ms_per_step": kt["solve_ms"] / steps}
    # one gather alone, timed with events on this rank's stream (5 repetitions after a warm-up)
    gather_ms = None
    if not rehearsal:
        per = dmod.padded_shard_len(G, world)
        local = torch.full((per, p + 6), float("nan"), dtype=torch.float64, device=dev)
        local[: hi - lo] = core_all[lo:hi]
        outb = torch.empty((per * world, p + 6), dtype=torch.float64, device=dev)
        stage[0] = "torch.distributed all_gather_into_tensor of the records (warm-up)"
        dist.all_gather_into_tensor(outb, local)
        torch.cuda.synchronize()
        stage[0] = "torch.distributed all_gather_into_tensor of the records (5 timed repetitions)"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dist.all_gather_into_tensor(outb, local)
        e1.record()
        torch.cuda.synchronize()
        gather_ms = e0.elapsed_time(e1) / 5
        mine["gather_ms"] = gather_ms
        # the exchange step behind the C ABI: RCCL's own count of the ranks, and the same gather through it.  Every step
        # below is a collective: the ranks first AGREE that each of them got as far as the unique id (a rank that raised
        # alone would leave the others waiting in the broadcast), and the audit must not cost the run its number
        lib = abi = err = None
        uid = (C.c_uint8 * 128)()
        problem = None
        try:
            abi = importlib.import_module(PKG + "._abi")
            lib = abi.load()
            err = abi.AnofoxError()
            if rank == 0 and not lib.anofox_hip_comm_unique_id(uid, C.byref(err)):
                problem = err.text()
        except Exception as exc:
            problem = str(exc)[:200]
        flags = [None] * world
        stage[0] = "all_gather_object of the ranks' unique-id status"
        dist.all_gather_object(flags, problem)
        if any(f is not None for f in flags):
            mine["c_abi_comm_error"] = next(f for f in flags if f is not None)
        else:
            try:
                t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
                stage[0] = "broadcast of the RCCL unique id"
                dist.broadcast(t, 0)
                uid = (C.c_uint8 * 128)(*t.cpu().tolist())
                comm = C.c_void_p()
                ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
                stage[0] = "anofox_hip_comm_create (ncclCommInitRank of the second communicator)"
                if not lib.anofox_hip_comm_create(ctx._h, world, rank, uid, C.byref(comm), C.byref(err)):
                    raise RuntimeError(err.text())
                mine["n_ranks_seen"] = int(lib.anofox_hip_comm_ranks_seen(comm))
                out2 = torch.empty_like(outb)
                stage[0] = "anofox_hip_gather_records_device (ncclAllGather through the C ABI)"
                if not lib.anofox_hip_gather_records_device(comm, C.c_void_p(local.data_ptr()), per, p + 6, C.c_void_p(out2.data_ptr()), C.byref(err)):
                    raise RuntimeError(err.text())
                torch.cuda.synchronize()
                mine["c_abi_gather_matches_torch"] = bool(torch.equal(torch.nan_to_num(out2, nan=-1.0), torch.nan_to_num(outb, nan=-1.0)))
                lib.anofox_hip_comm_destroy(comm)
            except Exception as exc:
                mine["c_abi_comm_error"] = str(exc)[:200]
    everyone = [None] * world
    stage[0] = "all_gather_object of the per-rank figures"
    dist.all_gather_object(everyone, mine)
    stage[0] = "(audit complete)"
    if rank != 0:
        return None
    return {"backend": dist.get_backend(), "world_size": world, "per_rank": everyone,
            "n_ranks_seen": everyone[0].get("n_ranks_seen"),
            "kernel_ms_per_step_max": max(r["kernel_ms_per_step"] for r in everyone),
            "kernel_ms_per_step_min": min(r["kernel_ms_per_step"] for r in everyone),
            "gather_ms": gather_ms}


def launch_ranks(n: int) -> int:
    """Start `n` ranks of this script (one process per GPU) through torch.distributed.run on 127.0.0.1 and wait."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--groups", type=int, default=1_000_000, help="total groups over all ranks")
    ap.add_argument("--rows", type=int, default=1000, help="rows per group")
    ap.add_argument("--features", type=int, default=8)
    ap.add_argument("--model", default="ols", choices=["ols", "ridge", "wls"])
    ap.add_argument("--inference", action="store_true")
    ap.add_argument("--window", action="store_true",
                    help="expanding-window fit + predict (*_fit_predict OVER ...): one fit per ROW")
    ap.add_argument("--frame", default="u,0",
                    help="with --window: ROWS BETWEEN a PRECEDING AND b PRECEDING as 'a,b' (a = u for UNBOUNDED)")
    ap.add_argument("--vif", action="store_true",
                    help="vif_agg: variance inflation factors (p OLS fits per group from one Gram matrix)")
    ap.add_argument("--predict", action="store_true",
                    help="fit + per-row predictions (*_fit_predict_agg); every 5th row is a prediction row (NULL y)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--end-to-end-row-log", action="store_true",
                    help="also time the end-to-end leg with the aggregate state's row log on (the DuckDB arena's default budgets)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the host-memory -> aggregate state -> records leg (p <= 8 only; reported as `end_to_end`)")
    ap.add_argument("--parity-sample", type=int, default=1024)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: this process becomes the launcher.  It has made no GPU call (importing
        # torch makes none) and makes none: the N ranks are child processes, rank 0's JSON line goes straight to
        # the inherited stdout, and the launcher exits with the children's return code.
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}: launch N ranks for --gpus N "
                         f"(or run `python bench.py --gpus N` without a launcher)")
    # ANOFOX_BENCH_REHEARSAL=1: several ranks share cuda:0 over gloo (to rehearse the N > 1 path on a 1-GPU box)
    rehearsal = os.environ.get("ANOFOX_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    dmod = importlib.import_module(PKG + ".distributed")

    G, n, p = args.groups, args.rows, args.features
    weighted = args.model == "wls"
    lo, hi = dmod.shard_range(G, rank, world)
    G_local = hi - lo
    # bound the footprint by what the card has free (288 GB nominal): inputs are resident, never re-generated
    need = G_local * n * (p + 1 + int(weighted)) * 8
    free, _ = torch.cuda.mem_get_info()
    if need > 0.9 * free:
        raise SystemExit(f"rank {rank}: workload needs {need / 1e9:.1f} GB, only {free / 1e9:.1f} GB free")
    offs, y, x_cols, w = synth.make_grouped(G_local, n, p, group_start=lo, weights=weighted, device=dev,
                                            chunk_groups=max(1, min(32768, (1 << 25) // n)))
    if args.predict:
        if world > 1:
            raise SystemExit("--predict is a single-GPU measurement")
        hold = (torch.arange(y.numel(), device=dev) % 5) == 4
        y = torch.where(hold, torch.full_like(y, float("nan")), y)
        del hold
    kw = {"compute_inference": args.inference}
    if args.model == "ridge":
        kw["alpha"] = 1.0
    opts = pkg.RegressionOptions(**kw).batch_options(args.model)
    ctx = pkg.Context(dev_index)
    sharded = dmod.ShardedBatchFit(ctx, G)

    pred_buf = torch.empty((y.numel(), 3), dtype=torch.float64, device=dev) if args.predict else None
    core_buf = torch.empty((G_local, p + 6), dtype=torch.float64, device=dev) if args.predict else None

    if args.window:
        if world > 1:
            raise SystemExit("--window is a single-GPU measurement")
        pred_buf = torch.empty((y.numel(), 3), dtype=torch.float64, device=dev)
    fa, fb = args.frame.split(",")
    frame = (None if fa.strip().lower().startswith("u") else int(fa), int(fb))

    vif_buf = torch.empty((G_local, p + 1), dtype=torch.float64, device=dev) if args.vif else None
    if args.vif and world > 1:
        raise SystemExit("--vif is a single-GPU measurement")

    def step():
        if args.vif:
            ctx.vif_batch_device(offs, x_cols, out=vif_buf)
            return None, None
        if args.window:
            ctx.fit_predict_window_device(offs, y, x_cols, w, opts, frame, pred=pred_buf)
            return None, None
        if args.predict:
            c, _ = ctx.fit_predict_batch_device(offs, y, x_cols, w, opts, core=core_buf, pred=pred_buf)
            return c, None
        return sharded.fit(offs, y, x_cols, w, opts)

    if not (args.vif or args.window or args.predict):
        sharded.prepare(offs, y, x_cols, w, opts)   # both pipeline slots allocated before anything is timed
    for _ in range(args.warmup):
        step()
    sharded.finish()
    torch.cuda.synchronize()
    timed = sharded.contexts() if sharded.contexts() else [ctx]
    if ctx not in timed:
        timed.append(ctx)
    for c in timed:
        c.enable_timing(True)
        c.collect_timing()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []                # an event at the end of every step's kernels (on the stream the step ran on): per-step spread
    for _ in range(args.steps):
        core_all, inf_all = step()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(sharded.last_stream() or torch.cuda.current_stream(dev))
        marks.append(ev)
    sharded.finish()          # every all-gather issued inside the timed region completes inside it
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # spacing of those marks (steady state: the steps are pipelined); SURVEY.md 8(d) asks for median and min.  Consecutive
    # steps alternate between two streams and a step's tail (solve, refinement) ends under the NEXT step's accumulate kernel,
    # so marks are compared two steps apart — the same stream's consecutive steps — and halved
    gaps = sorted(marks[i].elapsed_time(marks[i + 2]) / 2 for i in range(len(marks) - 2))
    step_ms_median = gaps[len(gaps) // 2] if gaps else None
    # (no minimum: a step's END moves with how its tail interleaves with the next step's accumulate kernel, so the smallest
    # spacing of two marks is not the duration of any step; roofline.avg_launch_ms is the kernel's own average)
    kt = None
    for c in timed:   # the sharded driver alternates between contexts: sum their kernel times
        k = c.collect_timing()
        c.enable_timing(False)
        if kt is None:
            kt = k
        elif k["accumulate_count"]:     # min / max of single launches do not add
            lo_ms = min(kt["accumulate_ms_min"], k["accumulate_ms_min"]) if kt["accumulate_count"] else k["accumulate_ms_min"]
            hi_ms = max(kt["accumulate_ms_max"], k["accumulate_ms_max"])
            kt = {key: kt[key] + k[key] for key in kt}
            kt["accumulate_ms_min"], kt["accumulate_ms_max"] = lo_ms, hi_ms
    refined = ctx.last_refine_count() if not (args.window or args.vif) else 0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ok, cerr, derr = True, 0.0, 0.0
    if args.vif:
        import oracle
        S = min(512, G_local)
        nr = int(offs[S].item())
        ref = oracle.vif_groups([c[:nr].cpu().numpy() for c in x_cols], offs[:S + 1].cpu().numpy())
        got = vif_buf[:S].cpu().numpy()
        fin = np.isfinite(ref[:, :p])
        ok = bool(np.array_equal(np.isfinite(got[:, :p]), fin) and np.array_equal(got[:, p], ref[:, p]))
        derr = float(np.max(np.abs(got[:, :p][fin] / ref[:, :p][fin] - 1.0))) if fin.any() else 0.0
        ok = ok and derr < 1e-6
        args.parity_sample = 0
    if args.window:
        # parity gate of the window path: a few partitions against the oracle's O(n^2) refits
        import oracle
        S = min(8, G_local)
        nr = int(offs[S].item())
        ref = oracle.fit_predict_window(y[:nr].cpu().numpy(), [c[:nr].cpu().numpy() for c in x_cols],
                                        offs[:S + 1].cpu().numpy(), w=w[:nr].cpu().numpy() if w is not None else None,
                                        start_preceding=frame[0], end_preceding=frame[1],
                                        model=args.model, **{k: v for k, v in kw.items() if k != "compute_inference"})
        got = pred_buf[:nr].cpu().numpy()
        m = ~np.isnan(ref[:, 0])
        ok = bool(np.array_equal(np.isnan(got[:, 0]), ~m))
        # every row of the sample, yhat and both interval bounds (ill-conditioned frames are refitted with refinement)
        sc = np.maximum(np.abs(ref[m, 0]), 1.0)
        cerr = float(np.max(np.abs(got[m, 0] - ref[m, 0]) / sc)) if m.any() else 0.0
        fin = m & np.isfinite(ref[:, 1]) & np.isfinite(ref[:, 2])
        derr = float(np.max(np.abs(got[fin, 1:] - ref[fin, 1:]) / np.maximum(np.abs(ref[fin, 1:]), 1.0))) if fin.any() else 0.0
        ok = ok and cerr < 1e-8 and derr < 1e-6
        args.parity_sample = 0
    if args.parity_sample > 0:
        # every rank checks head, middle and tail of its own shard (the gathered block [lo:hi] must be its own records)
        mine = core_all[lo:hi]
        mine_inf = inf_all[lo:hi] if inf_all is not None else None
        ok, cerr, derr = parity_gate(pkg, mine, mine_inf, offs, y, x_cols, w, args.model, kw, p, args.parity_sample)
        if world > 1:
            flag = torch.tensor([1.0 if ok else 0.0], device="cpu" if rehearsal else dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item() > 0.5)

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        fits_per_s = G * args.steps / elapsed
        bytes_fit = algorithmic_bytes_per_fit(n, p, weighted, args.inference)
        if args.vif:
            bytes_fit = 8 * n * p + 8 * (p + 1)     # the features once, the VIF record out
        acc_ms = kt["accumulate_ms"] / max(kt["accumulate_count"], 1)   # per launch (wide designs: per slab)
        acc_step_ms = kt["accumulate_ms"] / args.steps                      # all launches of one step
        if p <= 8:
            kernel, bound, unit, peak = "accumulate_narrow_kernel", "hbm", "GB/s", HBM_PEAK_GBS
            per_step = G_local * bytes_fit
            achieved = per_step / (acc_step_ms * 1e-3) / 1e9 if acc_step_ms > 0 else 0.0
        else:
            # (the speculative accumulate_quad kernel also takes p = 27 .. 33 of the unweighted fit with an intercept)
            # (csrc/accumulate_quad.hip: accumulate_quad_supports — the speculative LDS-DMA kernel takes 27 .. 42 and 49, 50)
            spec = not weighted and os.environ.get("ANOFOX_QUAD_SPEC", "1") != "0"
            quad = p <= 26 or (spec and (p <= 42 or p in (49, 50)))
            kernel = "accumulate_quad_kernel" if quad else ("accumulate_mid_kernel" if p <= 32 else "accumulate_wide_kernel")
            # which roof bounds this width: arithmetic intensity against the ridge point peak_flops / peak_bytes
            # (78.6 TFLOP/s / 8 TB/s = 9.8 flop/B).  p + 1 = 2 * 9.8 - 3 => widths up to p ~ 75 are HBM-bound
            intensity = algorithmic_flops_per_fit(n, p) / bytes_fit
            ridge = FP64_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
            if intensity < ridge:
                bound, unit, peak = "hbm", "GB/s", HBM_PEAK_GBS
                per_step = G_local * bytes_fit
                achieved = per_step / (acc_step_ms * 1e-3) / 1e9 if acc_step_ms > 0 else 0.0
            else:
                bound, unit, peak = "mfma", "TFLOP/s", FP64_MFMA_PEAK_TFLOPS
                per_step = G_local * algorithmic_flops_per_fit(n, p)
                achieved = per_step / (acc_step_ms * 1e-3) / 1e12 if acc_step_ms > 0 else 0.0
        if args.window:
            # the window path has its own kernel: one fit per ROW from running / rolling moments, 8 (p + 1 [+ 1]) B in and the
            # three prediction doubles out per row; its time is what the predict events bracket
            kernel = "expanding_predict_kernel" if frame[0] is None else "rolling_predict_kernel"
            bound, unit, peak = "hbm", "GB/s", HBM_PEAK_GBS
            acc_step_ms = kt["predict_ms"] / args.steps
            acc_ms = kt["predict_ms"] / max(kt["predict_count"], 1)
            per_step = G_local * n * (8 * (p + 1 + (1 if weighted else 0)) + 24)
            achieved = per_step / (acc_step_ms * 1e-3) / 1e9 if acc_step_ms > 0 else 0.0
        # the same work over the WHOLE step (accumulate + solve + refinement [+ gather]) — what `value` is quoted on
        step_achieved = per_step / (ms_per_step * 1e-3) / (1e9 if bound == "hbm" else 1e12)
        kernel_achieved = achieved
        if p > 8 and not args.window:
            # wide designs: the per-slab solve is a material part of the step, so the headline fraction is the step's;
            # the accumulate kernel's own rate stays in kernel_achieved / kernel_frac
            achieved = step_achieved
        # how much of the solve / refinement span ran under another step's accumulate kernel (two contexts on two
        # streams alternate): spans add up to more than the step exactly by the overlapped part
        solve_step_ms = kt["solve_ms"] / args.steps
        overlap_ms = max(0.0, acc_step_ms + solve_step_ms - ms_per_step)
        # the part of a step that no accumulate kernel covers: solve / refinement / launch gaps that are NOT hidden
        exposed_ms = max(0.0, ms_per_step - acc_step_ms)
        traffic = None   # HBM bytes per launch from the rocprofv3 PMC passes (profiles/hbm_traffic.json), if recorded
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and not (args.vif or args.window):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(f"{args.model}_G{G_local}_n{n}_p{p}")
                if traffic is None and not (args.inference and p <= 8) and tj.get(f"{args.model}_n{n}_p{p}_bytes_per_group"):
                    traffic = tj[f"{args.model}_n{n}_p{p}_bytes_per_group"] * G_local     # per step, like `achieved`
            except Exception:
                traffic = None
        extra = {}
        if args.vif:
            extra = {"equivalent_ols_fits_per_sec": G * p * args.steps / elapsed}
        if args.window:
            extra = {"rows_per_sec": G * n * args.steps / elapsed, "row_fits_per_sec": G * n * args.steps / elapsed,
                     "window_kernel_ms_per_step": kt["predict_ms"] / args.steps,
                     "frames_refitted_with_refinement_last_step": ctx.last_window_refit_count()}
        if args.predict:
            extra = {"rows_per_sec": G * n * args.steps / elapsed, "predict_kernel_ms_per_step": kt["predict_ms"] / args.steps,
                     "predict_GBps": G_local * n * (8 * p + 24) / (kt["predict_ms"] / args.steps * 1e-3) / 1e9
                     if kt["predict_ms"] > 0 else 0.0}
        out = {
            "metric": "group_fits_per_sec", "value": fits_per_s if ok else None, "unit": "fits/s", **extra,
            "ns_per_row": (elapsed / args.steps) * 1e9 / (G * n),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_step_median": step_ms_median,   # rank 0's own steps (HIP events at the end of every step)
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": (f"vif_agg: {G} groups x n={n} x p={p}, device-resident grouped columns" if args.vif else "") or f"{args.model}_fit{'_predict' if (args.predict or args.window) else ''}{(' OVER (ROWS BETWEEN ' + ('UNBOUNDED' if frame[0] is None else str(frame[0])) + ' PRECEDING AND ' + str(frame[1]) + ' PRECEDING)') if args.window else '_agg'}: {G} groups x n={n} x p={p}, device-resident grouped columns, "
                                   f"fit_intercept=true, compute_inference={str(args.inference).lower()}",
                       "groups_total": G, "groups_per_gpu": G_local, "rows_per_group": n, "features": p,
                       "partition": f"contiguous key ranges over {world} rank(s); all-gather of {p + 6}-double records"},
            "parity": {"ok": ok, "sample_groups_per_rank": min(args.parity_sample, G_local),
                       "sample_windows": ([list(wd) for wd in sample_windows(G_local, args.parity_sample)]
                                          if args.parity_sample > 0 else None),
                       "max_coef_rel_err": cerr, "max_diag_rel_err": derr},
            "roofline": {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak, "kernel_achieved": kernel_achieved, "kernel_frac": kernel_achieved / peak,
                         "step_achieved": step_achieved, "step_frac": step_achieved / peak, "traffic": traffic,
                         "traffic_source": ("profiles/hbm_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                            "passes of this kernel on this workload; not measured in this run)"
                                            if traffic is not None else None),
                         "kernel": kernel, "avg_launch_ms": acc_ms,
                         # shortest / longest single launch of that kernel inside the timed region (HIP events in the library)
                         "kernel_ms_min": kt.get("accumulate_ms_min") if not args.window else None,
                         "kernel_ms_max": kt.get("accumulate_ms_max") if not args.window else None,
                         "launches_per_step": kt["predict_count" if args.window else "accumulate_count"] / args.steps,
                         "kernel_ms_per_step": acc_step_ms, "groups_refined_last_launch": refined,
                         ("algorithmic_bytes_per_step" if bound == "hbm" else "algorithmic_flops_per_step"): per_step,
                         "hbm_GBps_algorithmic": (per_step if args.window else G_local * bytes_fit) / (acc_step_ms * 1e-3) / 1e9 if acc_step_ms > 0 else 0.0,
                         # solve_span: from the end of an accumulate kernel to the end of its solve / refinement.  For
                         # p <= 8 consecutive steps alternate between two streams and the span runs under the NEXT
                         # step's accumulate kernel; for wide designs the slabs of a step run on one stream and only
                         # the last slab's solve can overlap — solve_overlap_ms_per_step is computed, not assumed
                         # what a plain streaming-read kernel / a pure MFMA loop reach on this device (csrc/tools/
                         # hbm_read_rate, mfma_f64_rate; profiles/r01_hbm_read_rate.txt): the practical ceiling under `peak`
                         "measured_ceiling": MEASURED_STREAM_READ_GBS if bound == "hbm" else MEASURED_MFMA_F64_TFLOPS,
                         "frac_of_measured_ceiling": kernel_achieved / (MEASURED_STREAM_READ_GBS if bound == "hbm" else MEASURED_MFMA_F64_TFLOPS),
                         **({"arithmetic_intensity_flop_per_B": algorithmic_flops_per_fit(n, p) / bytes_fit,
                             "ridge_point_flop_per_B": FP64_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9),
                             # both roofs, whichever one bounds (SURVEY.md 8d asks for both on the wide configs)
                             "kernel_frac_of_hbm_peak": G_local * bytes_fit / (acc_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if acc_step_ms > 0 else 0.0,
                             "kernel_frac_of_mfma_peak": G_local * algorithmic_flops_per_fit(n, p) / (acc_step_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS if acc_step_ms > 0 else 0.0}
                            if (p > 8 and not (args.window or args.vif)) else {}),
                         "solve_span_ms_per_step": solve_step_ms,
                         "solve_overlap_ms_per_step": overlap_ms,
                         "exposed_non_accumulate_ms_per_step": exposed_ms,
                         # true only when (nearly) nothing of the solve is left outside the accumulate kernels' time
                         "solve_overlaps_next_accumulate": bool(exposed_ms <= 0.02 * ms_per_step)},
        }
    # ---- N > 1: what the SCALE record can be audited with (outside the timed region; the N = 1 path takes none of it).
    # The audit is a sequence of collectives, one of them on a second RCCL communicator: should any of them never return,
    # a watchdog prints the line — the timing above is complete — and ends the rank instead of hanging the run.
    if world > 1:
        import threading
        finished = threading.Event()
        audit_stage = ["(not started)"]      # the collective this rank entered last, for the watchdog's report

        def give_up():
            if finished.is_set():
                return
            # a collective that never returns is a FAILED run (exit code 3 on every rank), whatever the timing says: the line
            # is still printed — the timed region is complete — and names the collective each rank was waiting in
            sys.stderr.write(f"[bench rank {rank}] multi-rank audit hung in: {audit_stage[0]}\n")
            sys.stderr.flush()
            if rank == 0 and out is not None:
                out["multi_gpu"] = {"error": "the multi-rank audit did not finish in time (a collective never returned); the timed region above is complete",
                                    "rank0_pending_collective": audit_stage[0]}
                print(json.dumps(out), flush=True)
            os._exit(3)

        watchdog = threading.Timer(float(os.environ.get("ANOFOX_BENCH_AUDIT_TIMEOUT_S", "180")), give_up)
        watchdog.daemon = True
        watchdog.start()
        multi = multi_rank_audit(pkg, ctx, dmod, sharded, core_all, lo, hi, G, p, kt, args.steps, rehearsal, dev, audit_stage)
        finished.set()
        watchdog.cancel()
        if rank == 0 and multi is not None:
            out["multi_gpu"] = multi
    if rank == 0:
        if world == 1 and p <= 8 and not (args.no_end_to_end or args.vif or args.window or args.predict or args.inference):
            # The pipelined driver of the timed loop (two more contexts, each with its own streams) is finished with: while its streams
            # exist the H2D copies of the leg below run at 47-48 GB/s instead of the 55 GB/s the same leg reaches in a process without
            # them (scripts/e2e_bisect.py: with the driver alive 47.6, after dropping it 55.0).  A DuckDB process holds the arena's
            # contexts only.
            sharded.close()
            torch.cuda.synchronize()
            try:
                out["end_to_end"] = end_to_end_leg(pkg, ctx, offs, y, x_cols, w, opts, args.model, kw, G, n, p)
                if args.end_to_end_row_log:
                    out["end_to_end_with_row_log"] = end_to_end_leg(pkg, ctx, offs, y, x_cols, w, opts, args.model, kw, G, n, p, with_row_log=True)
            except Exception as exc:      # the leg is an addition to the line, never a reason to lose it
                out["end_to_end"] = {"error": str(exc)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(offs, y, x_cols, w, args.model, kw, n, p)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("parity gate failed: no throughput reported")


if __name__ == "__main__":
    main()
