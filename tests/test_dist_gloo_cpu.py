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
