"""Why is bench.py's end_to_end leg slower inside bench.py (43-48 GB/s) than on its own (55)?  (a) with the bench's 72 GB of inputs resident,
(b) after a few batch fits on the same context."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
pkg = importlib.import_module("anofox-statistics_amd")
synth = importlib.import_module("anofox-statistics_amd.synth")
G, n, p = 1_000_000, 1000, 8
offs, y, x_cols, w = synth.make_grouped(G, n, p, device="cuda:0")
ctx = pkg.Context(0)
opts = pkg.RegressionOptions().batch_options("ols")
def leg(tag):
    r = bench.end_to_end_leg(pkg, ctx, offs, y, x_cols, None, opts, "ols", {}, G, n, p)
    print(tag, json.dumps({k: r[k] for k in ("fits_per_s", "GBps_pcie", "update_seconds")}), flush=True)
leg("(a) 72 GB resident, fresh context:")
for _ in range(3):
    ctx.fit_batch_device(offs, y, x_cols, None, opts)
torch.cuda.synchronize()
leg("(b) after three batch fits:")
dmod = importlib.import_module("anofox-statistics_amd.distributed")
ctx.enable_timing(True); ctx.collect_timing()
for _ in range(3):
    ctx.fit_batch_device(offs, y, x_cols, None, opts)
torch.cuda.synchronize(); ctx.collect_timing(); ctx.enable_timing(False)
leg("(c) after timed batch fits (enable_timing / collect_timing):")
sharded = dmod.ShardedBatchFit(ctx, G)
sharded.prepare(offs, y, x_cols, None, opts)
for _ in range(4):
    sharded.fit(offs, y, x_cols, None, opts)
sharded.finish(); torch.cuda.synchronize()
leg("(d) after four steps of the sharded driver (two contexts, own streams, gates):")
del sharded
import gc; gc.collect(); torch.cuda.synchronize()
leg("(e) after dropping the sharded driver:")
