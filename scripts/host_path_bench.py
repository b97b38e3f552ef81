"""PCIe-inclusive rate of the host-pointer entry point (anofox_hip_fit_batch_host): pageable vs pinned inputs."""
import importlib
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("anofox-statistics_amd")

G, n, p = int(sys.argv[1]) if len(sys.argv) > 1 else 50000, 1000, 8
N = G * n
rng = np.random.default_rng(0)
offs = np.arange(G + 1, dtype=np.int64) * n
opts = pkg.RegressionOptions().batch_options("ols")
ctx = pkg.Context()
for kind in ("pageable", "pinned"):
    if kind == "pageable":
        cols = [rng.standard_normal(N) for _ in range(p)]
        y = rng.standard_normal(N)
    else:
        t = [torch.empty(N, dtype=torch.float64).pin_memory() for _ in range(p + 1)]
        for a, c in zip(t, cols + [y]):
            a.numpy()[:] = c
        cols = [a.numpy() for a in t[:p]]
        y = t[p].numpy()
    pkg.fit_batch_host(offs[:1001], y[:1000 * n], [c[:1000 * n] for c in cols], None, opts, ctx=ctx)   # warm-up
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        core, _ = pkg.fit_batch_host(offs, y, cols, None, opts, ctx=ctx)
        best = min(best, time.perf_counter() - t0)
    print(json.dumps({"inputs": kind, "groups": G, "rows": n, "features": p, "seconds": best,
                      "fits_per_s": G / best, "GBps": N * (p + 1) * 8 / best / 1e9, "status_ok": int((core[:, p + 5] == 0).sum())}))
