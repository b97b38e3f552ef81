#!/usr/bin/env python3
"""Which HIP call does a chunk of the streaming Update wait in?  16 device-resident chunks of 2^20 rows into a 1M-slot state;
run under `rocprofv3 --hip-runtime-trace --stats` (no counters) and read the HIP API statistics."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("anofox-statistics_amd")
dev = torch.device("cuda", 0)
p, N, G = 8, 16 << 20, 1_000_000
g = torch.Generator(device=dev).manual_seed(3)
X = torch.rand((N, p), device=dev, dtype=torch.float64, generator=g)
y = X.sum(dim=1) + torch.randn(N, device=dev, dtype=torch.float64, generator=g)
slot = torch.randint(0, G, (N,), device=dev, dtype=torch.int32, generator=g)
ctx = pkg.Context(0)
opts = pkg.RegressionOptions().batch_options("ols")
for rep in range(3):
    st = pkg.AggState(ctx, p, opts, initial_slots=G, retain_bytes=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    times = []
    for r0 in range(0, N, 1 << 20):
        ta = time.perf_counter()
        st.update_device(slot[r0:r0 + (1 << 20)], y[r0:r0 + (1 << 20)], X[r0:r0 + (1 << 20)], None, n_slots=G)
        times.append((time.perf_counter() - ta) * 1e3)
    t_issue = time.perf_counter() - t0
    ctx.synchronize()
    t = time.perf_counter() - t0
    print(f"rep {rep}: issue {t_issue * 1e3:.1f} ms, done {t * 1e3:.1f} ms = {N / t / 1e9:.3f} G rows/s; per-call host ms: " + " ".join(f"{v:.2f}" for v in times), flush=True)
    st.close()
