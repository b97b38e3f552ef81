#!/usr/bin/env python3
"""Rows/s of the streaming aggregate state (anofox_hip_agg_state_*): Update from pinned host memory (PCIe-bound),
Update from device-resident chunks (kernel-bound), Finalize — for rows arriving in shuffled and in sorted group
order.  Prints one JSON line per arrival order; parity-gated against the oracle on a sample of groups.

    python scripts/ingest_bench.py [--groups 1000000 --rows 64 --features 8 --model ols --steps 3]
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=1_000_000)
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--features", type=int, default=8)
    ap.add_argument("--model", default="ols", choices=["ols", "ridge", "wls"])
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--chunk", type=int, default=1 << 22, help="rows per update call")
    ap.add_argument("--hard-frac", type=float, default=0.0,
                    help="fraction of the groups made nearly exact fits (noise 1e-7): Finalize queues them for refinement")
    ap.add_argument("--retain-gib", type=float, default=0.0, help="row log budget (anofox_hip_agg_state_retain_rows); 0 = off")
    args = ap.parse_args()
    pkg = importlib.import_module("anofox-statistics_amd")
    synth = importlib.import_module("anofox-statistics_amd.synth")
    import oracle
    dev = torch.device("cuda", 0)
    G, n, p = args.groups, args.rows, args.features
    weighted = args.model == "wls"
    offs, y, x_cols, w = synth.make_grouped(G, n, p, weights=weighted, device=dev)
    N = G * n
    X = torch.stack(x_cols, dim=1).contiguous()
    n_hard = int(args.hard_frac * G)
    if n_hard:
        g = torch.Generator(device=dev).manual_seed(5)
        b = torch.rand(p, device=dev, dtype=torch.float64, generator=g) * 4 - 2
        y[:n_hard * n] = X[:n_hard * n] @ b + 3.0 + 1e-7 * torch.randn(n_hard * n, device=dev, dtype=torch.float64, generator=g)
    ctx = pkg.Context(0)
    opts = pkg.RegressionOptions().batch_options(args.model)
    bytes_row = 8 * (p + 1) + (8 if weighted else 0) + 4
    for order in ("shuffled", "sorted"):
        if order == "shuffled":
            perm = torch.randperm(N, device=dev)
            slot = (perm // n).to(torch.int32)
            Xo, yo, wo = X[perm].contiguous(), y[perm].contiguous(), (w[perm].contiguous() if weighted else None)
        else:
            perm = None
            slot = (torch.arange(N, device=dev) // n).to(torch.int32)
            Xo, yo, wo = X, y, w
        # pinned host copies (what a DuckDB shim's arenas would be)
        hs, hx, hy = slot.cpu().pin_memory(), Xo.cpu().pin_memory(), yo.cpu().pin_memory()
        hw = wo.cpu().pin_memory() if weighted else None
        res = {}
        for where in ("host_pinned", "device"):
            best = None
            for _ in range(args.steps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st = pkg.AggState(ctx, p, opts, initial_slots=G, retain_bytes=int(args.retain_gib * (1 << 30)))
                t1 = time.perf_counter()
                for r0 in range(0, N, args.chunk):
                    r1 = min(N, r0 + args.chunk)
                    if where == "device":
                        st.update_device(slot[r0:r1], yo[r0:r1], Xo[r0:r1], wo[r0:r1] if weighted else None, n_slots=G)
                    else:
                        lib, C = st._lib, __import__("ctypes")
                        err = pkg._abi.AnofoxError()
                        ok = lib.anofox_hip_agg_state_update_host(
                            st._h, r1 - r0, G, hs[r0:r1].data_ptr(), hy[r0:r1].data_ptr(), hx[r0:r1].data_ptr(),
                            hw[r0:r1].data_ptr() if weighted else None, None, C.byref(err))
                        assert ok, err.text()
                ctx.synchronize()
                t2 = time.perf_counter()
                core, _, unref = st.finalize()
                t3 = time.perf_counter()
                st_bytes = st.retained_bytes
                st.close()
                rec = {"create_ms": (t1 - t0) * 1e3, "update_ms": (t2 - t1) * 1e3, "finalize_ms": (t3 - t2) * 1e3,
                       "groups_unrefined": int(unref), "retained_GB": st_bytes / 1e9}
                if best is None or rec["update_ms"] < best["update_ms"]:
                    best = rec
            best["rows_per_sec"] = N / (best["update_ms"] * 1e-3)
            best["input_GBps"] = N * bytes_row / (best["update_ms"] * 1e-3) / 1e9
            res[where] = best
        # parity: a sample of groups against the oracle (rows in arrival order)
        S = 256
        rows = torch.nonzero(slot < S).squeeze(1)
        so = slot[rows].cpu().numpy().astype(np.int64)
        ordr = np.argsort(so, kind="stable")
        counts = np.bincount(so, minlength=S)
        go = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        yy = yo[rows].cpu().numpy()[ordr]
        xx = Xo[rows].cpu().numpy()[ordr]
        ww = wo[rows].cpu().numpy()[ordr] if weighted else None
        kw = {"alpha": 1.0} if args.model == "ridge" else {}
        rcore, _ = oracle.fit_groups(yy, [np.ascontiguousarray(xx[:, j]) for j in range(p)], go, w=ww, model=args.model, **kw)
        c = core[:S]
        scale = np.max(np.abs(rcore[:, :p + 1]), axis=1, keepdims=True)
        cerr = float(np.max(np.abs(c[:, :p + 1] - rcore[:, :p + 1]) / np.maximum(np.abs(rcore[:, :p + 1]), 1e-3 * scale)))
        derr = float(np.max(np.abs(c[:, p + 1:p + 4] / rcore[:, p + 1:p + 4] - 1.0)))
        ok = bool(np.array_equal(c[:, p + 5], rcore[:, p + 5]) and cerr < 1e-9 and derr < 1e-6)
        print(json.dumps({"metric": "ingest_rows_per_sec", "arrival_order": order, "model": args.model,
                          "groups": G, "rows_per_group": n, "features": p, "rows": N, "bytes_per_row": bytes_row,
                          "update_chunk_rows": args.chunk, "retain_gib": args.retain_gib, "hard_groups": n_hard, **{k: v for k, v in res.items()},
                          "groups_unrefined": unref,
                          "parity": {"ok": ok, "sample_groups": S, "max_coef_rel_err": cerr, "max_diag_rel_err": derr}}), flush=True)
        del hs, hx, hy, hw


if __name__ == "__main__":
    main()
