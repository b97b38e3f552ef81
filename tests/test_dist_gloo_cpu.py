"""world_size-2 gloo test of the multi-GPU plumbing on CPU: contiguous key-range partition + one all-gather
of the per-group records.  The per-rank records are produced by the CPU oracle here (no GPU in this tier);
on a GPU box the same gather is fed by the HIP path (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, G, n, p, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module("anofox-statistics_amd.distributed")
    synth = importlib.import_module("anofox-statistics_amd.synth")
    import oracle
    lo, hi = d.shard_range(G, rank, world)
    offs, y, xc, _ = synth.make_grouped(hi - lo, n, p, group_start=lo)
    core, _ = oracle.fit_groups(y.numpy(), [c.numpy() for c in xc], offs.numpy())
    full = d.gather_records(torch.from_numpy(core), G)
    # async variant
    full2, work = d.gather_records(torch.from_numpy(core), G, async_op=True)
    work.wait()
    assert torch.equal(full, full2)
    np.save(os.path.join(tmp, f"rank{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_and_gather(tmp_path):
    G, n, p, world = 37, 50, 3, 2       # odd group count: the last shard is padded
    port = _free_port()
    mp.spawn(_worker, args=(world, port, G, n, p, str(tmp_path)), nprocs=world, join=True)
    import importlib
    import oracle
    synth = importlib.import_module("anofox-statistics_amd.synth")
    offs, y, xc, _ = synth.make_grouped(G, n, p)
    want, _ = oracle.fit_groups(y.numpy(), [c.numpy() for c in xc], offs.numpy())
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == want.shape
        assert np.array_equal(got, want)      # every rank ends with every group's record, bit for bit


def _route_worker(rank, world, port, tmp):
    """Every rank holds an arbitrary slice of the arriving rows, routes them by hash64(key) % world, and the ranks
    exchange them (gloo all_to_all on CPU tensors; RCCL would move device buffers): afterwards rank r holds exactly the
    rows of the keys it owns, every key whole."""
    sys.path.insert(0, ROOT)
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module("anofox-statistics_amd.distributed")
    rng = np.random.default_rng(5)
    N, K = 20_000, 700
    keys_all = rng.integers(0, 1 << 62, K, dtype=np.int64)[rng.integers(0, K, N)]
    vals_all = rng.standard_normal(N)
    mine = np.arange(N) % world == rank                      # rows arrive at ranks in no relation to their keys
    keys, vals = keys_all[mine], vals_all[mine]
    shard = d.hash_partition(keys, world)
    send = [torch.from_numpy(np.stack([keys[shard == r].astype(np.float64), vals[shard == r]], 1)) for r in range(world)]
    counts = torch.tensor([t.shape[0] for t in send])
    all_counts = [torch.zeros(world, dtype=torch.long) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    # gloo has no all_to_all for CPU tensors of uneven sizes: pairwise send / recv in rank order
    got = []
    for src in range(world):
        for dst in range(world):
            if src == dst:
                if rank == src:
                    got.append(send[dst])
            elif rank == src:
                dist.send(send[dst], dst)
            elif rank == dst:
                buf = torch.empty((int(all_counts[src][dst]), 2), dtype=torch.float64)
                dist.recv(buf, src)
                got.append(buf)
    mine_rows = torch.cat(got).numpy()
    np.save(os.path.join(tmp, f"route{rank}.npy"), mine_rows)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hash_routing(tmp_path):
    import importlib
    world = 2
    port = _free_port()
    mp.spawn(_route_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    d = importlib.import_module("anofox-statistics_amd.distributed")
    rng = np.random.default_rng(5)
    N, K = 20_000, 700
    keys_all = rng.integers(0, 1 << 62, K, dtype=np.int64)[rng.integers(0, K, N)]
    vals_all = rng.standard_normal(N)
    owner = d.hash_partition(keys_all, world)
    seen = 0
    for r in range(world):
        rows = np.load(tmp_path / f"route{r}.npy")
        want_keys = keys_all[owner == r].astype(np.float64)
        assert rows.shape[0] == want_keys.shape[0]
        assert np.array_equal(np.sort(rows[:, 0]), np.sort(want_keys))          # exactly the rows of the keys this rank owns
        assert np.isclose(rows[:, 1].sum(), vals_all[owner == r].sum())
        seen += rows.shape[0]
    assert seen == N
    # the partition is balanced and deterministic: the known answers pin the hash (shared with the C++ ingest)
    assert [int(v) for v in d.hash64(np.array([0, 1, 2, 12345678901234567], dtype=np.uint64))] == \
        [16294208416658607535, 10451216379200822465, 10905525725756348110, 13463060612230490842]
    assert abs(np.mean(owner) - 0.5) < 0.05
