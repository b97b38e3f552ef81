"""GPU tests of the sharded driver: `ShardedBatchFit` fed by the HIP path — step pipelining on two
streams / two contexts, the one-shot accumulate gate, and the partition + all-gather over two ranks that share the
box's one GPU (gloo; RCCL refuses two ranks on one device, the 8-GPU run is the driver's)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, import_pkg

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("p,inference", [(8, True), (8, False), (20, True), (40, False)])
def test_sharded_driver_single_rank_equals_plain_calls(p, inference):
    """world = 1, depth = 2, several steps on changing inputs: every step's full output equals a plain
    fit_batch_device call on the same inputs bit for bit (same kernels, same order of operations)."""
    import torch
    pkg = import_pkg()
    synth = import_pkg("synth")
    dmod = import_pkg("distributed")
    G, n = (3000, 200) if p <= 8 else (600, 150)
    ctx = pkg.Context(0)
    plain = pkg.Context(0)
    opts = pkg.RegressionOptions(compute_inference=inference).batch_options("ols")
    sharded = dmod.ShardedBatchFit(ctx, G)
    steps = []
    for k in range(5):
        offs, y, x_cols, _ = synth.make_grouped(G, n, p, group_start=k * G, device="cuda:0")
        core, inf = sharded.fit(offs, y, x_cols, None, opts)
        steps.append((offs, y, x_cols, core, inf))
        if k % 2 == 1:      # results of a slot are read before the slot is reused (depth = 2)
            sharded.finish()
            torch.cuda.synchronize()
            for (o2, y2, x2, c2, i2) in steps:
                want_c, want_i = plain.fit_batch_device(o2, y2, x2, None, opts)
                torch.cuda.synchronize()
                assert torch.equal(c2.view(torch.int64), want_c.view(torch.int64))
                if inference:
                    assert torch.equal(i2.view(torch.int64), want_i.view(torch.int64))
            steps = []
    sharded.finish()
    torch.cuda.synchronize()


def test_gate_does_not_outlive_the_sharded_driver():
    """The accumulate gate is one-shot: after a sharded run is dropped (its torch events destroyed) the caller's
    context is used directly and must neither wait on nor record into a dead event."""
    import gc
    import torch
    pkg = import_pkg()
    synth = import_pkg("synth")
    dmod = import_pkg("distributed")
    ctx = pkg.Context(0)
    opts = pkg.RegressionOptions().batch_options("ols")
    offs, y, x_cols, _ = synth.make_grouped(500, 100, 4, device="cuda:0")
    sharded = dmod.ShardedBatchFit(ctx, 500)
    for _ in range(3):
        a, _ = sharded.fit(offs, y, x_cols, None, opts)
    sharded.finish()
    torch.cuda.synchronize()
    want = a.clone()
    del sharded, a
    gc.collect()
    for _ in range(3):
        got, _ = ctx.fit_batch_device(offs, y, x_cols, None, opts)
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int64), want.view(torch.int64))
    # setting a gate and then failing validation must not leave it armed either
    ev = torch.cuda.Event()
    ev.record()
    ctx.set_accumulate_gate(None, ev)
    del ev
    gc.collect()
    got, _ = ctx.fit_batch_device(offs, y, x_cols, None, opts)
    got, _ = ctx.fit_batch_device(offs, y, x_cols, None, opts)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int64), want.view(torch.int64))


def _rank_main(rank, world, port, G, n, p, inference, tmp):
    """One rank: fits ITS key range with the HIP path on cuda:0 and takes part in the all-gather (gloo)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("anofox-statistics_amd")
    synth = importlib.import_module("anofox-statistics_amd.synth")
    dmod = importlib.import_module("anofox-statistics_amd.distributed")
    torch.cuda.set_device(0)
    lo, hi = dmod.shard_range(G, rank, world)
    offs, y, x_cols, _ = synth.make_grouped(hi - lo, n, p, group_start=lo, device="cuda:0")
    ctx = pkg.Context(0)
    opts = pkg.RegressionOptions(compute_inference=inference).batch_options("ols")
    sharded = dmod.ShardedBatchFit(ctx, G)
    for _ in range(3):          # both pipeline slots and a reuse
        core_all, inf_all = sharded.fit(offs, y, x_cols, None, opts)
    sharded.finish()
    torch.cuda.synchronize()
    np.save(os.path.join(tmp, f"core{rank}.npy"), core_all.cpu().numpy())
    if inference:
        np.save(os.path.join(tmp, f"inf{rank}.npy"), inf_all.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("G,n,p,inference", [(1001, 300, 8, True), (257, 120, 24, False)])
def test_two_ranks_hip_fed_gather_equals_single_rank_fit(tmp_path, G, n, p, inference):
    """2 ranks (gloo) sharing the GPU, each fitting its contiguous key range with the HIP kernels, then the gather:
    every rank ends with every group's record, bit for bit what ONE rank computes for the whole batch (odd G: the
    last shard is padded)."""
    import multiprocessing as mp
    import torch
    world = 2
    port = _free_port()
    mpc = mp.get_context("forkserver")
    procs = [mpc.Process(target=_rank_main, args=(r, world, port, G, n, p, inference, str(tmp_path))) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
    for pr in procs:
        if pr.is_alive():
            pr.kill()
            pytest.fail("a rank did not finish")
        assert pr.exitcode == 0, f"rank exited with {pr.exitcode}"
    pkg = import_pkg()
    synth = import_pkg("synth")
    offs, y, x_cols, _ = synth.make_grouped(G, n, p, device="cuda:0")
    ctx = pkg.Context(0)
    opts = pkg.RegressionOptions(compute_inference=inference).batch_options("ols")
    core, inf = ctx.fit_batch_device(offs, y, x_cols, None, opts)
    torch.cuda.synchronize()
    want_c = core.cpu().numpy()
    for r in range(world):
        got = np.load(tmp_path / f"core{r}.npy")
        assert got.shape == want_c.shape
        assert np.array_equal(got.view(np.int64), want_c.view(np.int64)), f"rank {r}: gathered core records differ"
        if inference:
            gi = np.load(tmp_path / f"inf{r}.npy")
            assert np.array_equal(gi.view(np.int64), inf.cpu().numpy().view(np.int64)), f"rank {r}: inference differs"


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a bare invocation (no torchrun): the launcher starts the ranks itself and
    relays rank 0's JSON line.  Rehearsal mode: the two ranks share cuda:0 over gloo."""
    import json
    import multiprocessing as mp
    mpc = mp.get_context("forkserver")
    out = str(tmp_path / "bench.json")
    pr = mpc.Process(target=_run_bench, args=(out,))
    pr.start()
    pr.join(600)
    if pr.is_alive():
        pr.kill()
        pytest.fail("bench.py --gpus 2 did not finish")
    assert pr.exitcode == 0
    line = [ln for ln in open(out).read().splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["parity"]["ok"] and rec["value"] > 0
    assert rec["config"]["groups_per_gpu"] == 10_000
    # the audit block the driver's SCALE run is read with: one entry per rank, group counts that add up, kernel times
    mg = rec["multi_gpu"]
    assert "error" not in mg and mg["world_size"] == 2
    ranks = sorted(mg["per_rank"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1]
    assert sum(r["groups"] for r in ranks) == 20_000
    assert all(r["kernel_ms_per_step"] > 0 and r["solve_span_ms_per_step"] >= 0 for r in ranks)
    assert mg["kernel_ms_per_step_min"] <= mg["kernel_ms_per_step_max"]
    assert rec["roofline"]["kernel_ms_min"] > 0


def test_bench_audit_watchdog_fails_the_run(tmp_path):
    """A collective of the multi-rank audit that never returns must FAIL the run (non-zero exit), with the timing line
    still printed and the pending collective named — a hung RCCL call reported as rc 0 would never be investigated."""
    import json
    import multiprocessing as mp
    mpc = mp.get_context("forkserver")
    out = str(tmp_path / "bench.json")
    pr = mpc.Process(target=_run_bench, args=(out, {"ANOFOX_BENCH_TEST_HANG_RANK": "1", "ANOFOX_BENCH_AUDIT_TIMEOUT_S": "8"}))
    pr.start()
    pr.join(600)
    if pr.is_alive():
        pr.kill()
        pytest.fail("bench.py --gpus 2 with a held-back rank did not end")
    assert pr.exitcode != 0
    lines = [ln for ln in open(out).read().splitlines() if ln.startswith("{")]
    assert lines, "the timing line is printed even when the audit hangs"
    rec = json.loads(lines[-1])
    assert rec["parity"]["ok"] and rec["value"] > 0
    assert "error" in rec["multi_gpu"] and rec["multi_gpu"]["rank0_pending_collective"]


def _run_bench(out, extra_env=None):
    """Runs in a process forked from the clean fork server (no GPU state): exec is allowed here."""
    import subprocess
    env = dict(os.environ, ANOFOX_BENCH_REHEARSAL="1", **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    with open(out, "w") as f:
        rc = subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--groups", "20000",
                              "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], stdout=f, env=env)
    sys.exit(rc)


def test_c_abi_gather_on_a_single_rank_communicator():
    """anofox_hip_comm_* / anofox_hip_gather_records_device (RCCL behind the C ABI): with the one GPU of this box a
    communicator of world size 1 — unique id, init, an all-gather that is stream-ordered behind a fit on the same
    context, destroy.  (N > 1 ranks need N GPUs: the driver's scaling run.)"""
    import ctypes as C
    import torch
    pkg = import_pkg()
    abi = import_pkg("_abi")
    synth = import_pkg("synth")
    lib = abi.load()
    err = abi.AnofoxError()
    uid = (C.c_uint8 * 128)()
    assert lib.anofox_hip_comm_unique_id(uid, C.byref(err)), err.text()
    ctx = pkg.Context(0)
    comm = C.c_void_p()
    assert lib.anofox_hip_comm_create(ctx._h, 1, 0, uid, C.byref(comm), C.byref(err)), err.text()
    assert lib.anofox_hip_comm_world_size(comm) == 1 and lib.anofox_hip_comm_rank(comm) == 0
    G, n, p = 4000, 200, 8
    offs, y, x_cols, _ = synth.make_grouped(G, n, p, device="cuda:0")
    opts = pkg.RegressionOptions().batch_options("ols")
    core, _ = ctx.fit_batch_device(offs, y, x_cols, None, opts)
    out = torch.full((G, p + 6), float("nan"), dtype=torch.float64, device="cuda:0")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    assert lib.anofox_hip_gather_records_device(comm, C.c_void_p(core.data_ptr()), G, p + 6, C.c_void_p(out.data_ptr()), C.byref(err)), err.text()
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int64), core.view(torch.int64))
    assert not lib.anofox_hip_gather_records_device(None, None, 1, 1, None, C.byref(err)) and err.code == 1
    assert not lib.anofox_hip_comm_create(ctx._h, 2, 5, uid, C.byref(C.c_void_p()), C.byref(err)) and err.code == 1
    lib.anofox_hip_comm_destroy(comm)
    lib.anofox_hip_comm_destroy(None)
    ctx.close()
