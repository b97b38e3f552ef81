"""bench.py's end_to_end leg on its own (62 500 groups of data resident, 1M slots streamed): does it reach the rate of scripts/e2e_probe.py?"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
pkg = importlib.import_module("anofox-statistics_amd")
synth = importlib.import_module("anofox-statistics_amd.synth")
G, n, p = 1_000_000, 1000, 8
offs, y, x_cols, w = synth.make_grouped(62_500, n, p, device="cuda:0")
ctx = pkg.Context(0)
kw = {}
opts = pkg.RegressionOptions().batch_options("ols")
r = bench.end_to_end_leg(pkg, ctx, offs, y, x_cols, None, opts, "ols", kw, G, n, p)
print(json.dumps({k: r[k] for k in ("fits_per_s", "GBps_pcie", "update_seconds", "finalize_seconds", "parity")}))
