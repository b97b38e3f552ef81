#!/usr/bin/env python3
"""Where does the host-rows -> aggregate-state path spend its time?  H2D rate of the box (torch copy from pinned memory), then
anofox_hip_agg_state_update_host from torch-pinned memory and from anofox_hip_host_alloc memory, sorted and shuffled arrival,
for a few slot counts.  One line per case."""
import ctypes as C
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

pkg = importlib.import_module("anofox-statistics_amd")
abi = importlib.import_module("anofox-statistics_amd._abi")
lib = abi.load()
dev = torch.device("cuda", 0)
p = 8
ctx = pkg.Context(0)
opts = pkg.RegressionOptions().batch_options("ols")

N = 32 << 20      # rows per experiment (2.4 GB)
g = torch.Generator(device=dev).manual_seed(3)
X = torch.rand((N, p), device=dev, dtype=torch.float64, generator=g)
y = X.sum(dim=1) + torch.randn(N, device=dev, dtype=torch.float64, generator=g)
hx, hy = X.cpu().pin_memory(), y.cpu().pin_memory()
# raw H2D rate
d = torch.empty_like(X)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(hx, non_blocking=True); torch.cuda.synchronize(); t = time.perf_counter() - t0
print(f"torch H2D from pin_memory(): {hx.numel() * 8 / t / 1e9:.1f} GB/s", flush=True)
del d
# the library's own page-locked allocation
nbx, nby = N * p * 8, N * 8
ax, ay, as_ = lib.anofox_hip_host_alloc(nbx), lib.anofox_hip_host_alloc(nby), lib.anofox_hip_host_alloc(N * 4)
C.memmove(ax, hx.data_ptr(), nbx); C.memmove(ay, hy.data_ptr(), nby)

def run(label, n_slots, rows_per_slot_in_order, px, py, ps, chunk=1 << 22):
    err = abi.AnofoxError()
    best = None
    for _ in range(2):
        st = pkg.AggState(ctx, p, opts, initial_slots=n_slots, retain_bytes=0)
        ctx.synchronize(); t0 = time.perf_counter()
        for r0 in range(0, N, chunk):
            r1 = min(N, r0 + chunk)
            ok = lib.anofox_hip_agg_state_update_host(st._h, r1 - r0, n_slots, ps + 4 * r0, py + 8 * r0, px + 8 * p * r0, None, None, C.byref(err))
            assert ok, err.text()
        ctx.synchronize(); t = time.perf_counter() - t0
        st.close()
        best = t if best is None else min(best, t)
    print(f"{label}: n_slots={n_slots} {N / best / 1e9:.3f} G rows/s = {N * 76 / best / 1e9:.1f} GB/s", flush=True)

for n_slots in (1 << 15, 1 << 20):
    for order in ("sorted", "shuffled"):
        if order == "sorted":
            slot = (torch.arange(N, device=dev) // (N // n_slots)).to(torch.int32)
        else:
            slot = torch.randint(0, n_slots, (N,), device=dev, dtype=torch.int32, generator=g)
        hs = slot.cpu().pin_memory()
        C.memmove(as_, hs.data_ptr(), N * 4)
        run(f"torch pinned, {order}", n_slots, None, hx.data_ptr(), hy.data_ptr(), hs.data_ptr())
        run(f"host_alloc,   {order}", n_slots, None, ax, ay, as_)
# slots that move in windows: rows of 62 500 slots per 4M-row chunk (the bench's end_to_end pattern), 1M slots in the state
n_slots = 1_000_000
slot = (torch.randint(0, 62500, (N,), device=dev, dtype=torch.int32, generator=g) + 62500 * ((torch.arange(N, device=dev) // (4 << 20)) % 16).to(torch.int32))
hs = slot.cpu().pin_memory()
run("torch pinned, windowed 62500-slot batches", n_slots, None, hx.data_ptr(), hy.data_ptr(), hs.data_ptr())
# the same pattern sustained for as many rows as the bench's end_to_end leg streams (1e9): does the rate hold over 1.5 s?
err = abi.AnofoxError()
st = pkg.AggState(ctx, p, opts, initial_slots=n_slots, retain_bytes=0)
ctx.synchronize(); t0 = time.perf_counter()
rows = 0
for rep in range(30):
    for r0 in range(0, N, 1 << 22):
        r1 = min(N, r0 + (1 << 22))
        ok = lib.anofox_hip_agg_state_update_host(st._h, r1 - r0, n_slots, hs.data_ptr() + 4 * r0, hy.data_ptr() + 8 * r0, hx.data_ptr() + 8 * p * r0, None, None, C.byref(err))
        assert ok, err.text()
        rows += r1 - r0
    if rep in (0, 9, 19, 29):
        ctx.synchronize()
        print(f"sustained windowed, after {rows / 1e9:.2f} G rows: {rows * 76 / (time.perf_counter() - t0) / 1e9:.1f} GB/s", flush=True)
st.close()
