"""Times the grouped residual-diagnostics kernel (anofox_hip_residuals_batch_device) on device-resident columns.
usage: python scripts/residuals_bench.py G n p [steps]"""
import importlib
import json
import sys

import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pkg = importlib.import_module("anofox-statistics_amd")
G, n, p = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
ctx = pkg.Context(0)
gen = torch.Generator(device=dev).manual_seed(1)
N = G * n
y = torch.randn(N, dtype=torch.float64, device=dev, generator=gen)
y_hat = y + 0.1 * torch.randn(N, dtype=torch.float64, device=dev, generator=gen)
x_cols = [torch.randn(N, dtype=torch.float64, device=dev, generator=gen) + j for j in range(p)]
offs = torch.arange(0, N + 1, n, dtype=torch.int64, device=dev)
rse = torch.full((G,), 0.1, dtype=torch.float64, device=dev)
out = torch.empty((N, 4), dtype=torch.float64, device=dev)
grp = torch.empty((G, 2), dtype=torch.float64, device=dev)
for _ in range(3):
    ctx.residuals_batch_device(offs, y, y_hat, x_cols, rse, out=out, group=grp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    ctx.residuals_batch_device(offs, y, y_hat, x_cols, rse, out=out, group=grp)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
# algorithmic traffic: y, y_hat and p feature columns read once, 4 doubles written per row (the second pass over x
# is expected to hit in L2 for groups of this size)
gb = N * (8 * (p + 2) + 32) / 1e9
hsum = out[:, 3].view(G, n).sum(1)
print(json.dumps({"workload": f"residuals_diagnostics_agg: {G} groups x n={n} x p={p}", "ms": ms, "rows_per_s": N / ms * 1e3,
                  "algorithmic_GBps": gb / ms * 1e3, "hat_trace_max_err": float((hsum - (p + 1)).abs().max()) if p else None}))
