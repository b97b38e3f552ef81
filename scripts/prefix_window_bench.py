"""Expanding-window fit + predict for designs wider than the in-register window kernels (p > 8), inputs resident in HBM:
python scripts/prefix_window_bench.py --partitions 20000 --rows 500 --features 16   (ANOFOX_FRAMES_PREFIX=0: frames as virtual groups)"""
import argparse, importlib, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("anofox-statistics_amd")
synth = importlib.import_module("anofox-statistics_amd.synth")
ap = argparse.ArgumentParser()
ap.add_argument("--partitions", type=int, default=20000)
ap.add_argument("--rows", type=int, default=500)
ap.add_argument("--features", type=int, default=16)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
offs, y, x_cols, _ = synth.make_grouped(a.partitions, a.rows, a.features, device="cuda:0")
ctx = pkg.Context(0)
opts = pkg.RegressionOptions().batch_options("ols")
pred = ctx.fit_predict_expanding_device(offs, y, x_cols, None, opts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    pred = ctx.fit_predict_expanding_device(offs, y, x_cols, None, opts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
# parity of a sample against the oracle
sys.path.insert(0, ROOT)
import oracle
S = 4
n = a.rows
ref = oracle.fit_predict_expanding(y[:S * n].cpu().numpy(), [c[:S * n].cpu().numpy() for c in x_cols], offs[:S + 1].cpu().numpy(), model="ols")
got = pred[:S * n].cpu().numpy()
m = ~np.isnan(ref)
ok = bool(np.array_equal(np.isnan(got), np.isnan(ref)) and np.max(np.abs(got[m] - ref[m]) / np.maximum(np.abs(ref[m]), 1e-3 * np.abs(ref[m]).max())) < 1e-8)
print(json.dumps({"metric": "window_row_fits_per_sec", "partitions": a.partitions, "rows": a.rows, "features": a.features,
                  "prefix": os.environ.get("ANOFOX_FRAMES_PREFIX", "1"), "ms": dt * 1e3, "row_fits_per_s": a.partitions * a.rows / dt, "parity": ok}))
