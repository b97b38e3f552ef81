#!/usr/bin/env python3
"""Per-group errors of fuzz cases against the oracle (debugging aid for refit_dd.hip): python scripts/diag_refit_dd.py very 30560 30561 ..."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
pkg = importlib.import_module("anofox-statistics_amd")
import test_gpu_fuzz as F
ctx = pkg.Context()
family = sys.argv[1]
for seed in map(int, sys.argv[2:]):
    p, offs, y, x_cols, w, model, kw, deg = F._case(seed, {"very": "very", "wide": True, "narrow": False}[family])
    wv = w if model == "wls" else None
    core, inf = pkg.fit_batch_host(offs, y, x_cols, wv, pkg.RegressionOptions(**kw).batch_options(model), ctx=ctx)
    rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, model=model, **kw)
    print(f"seed {seed} {model} p={p} {kw} REFIT_DD={os.environ.get('ANOFOX_REFIT_DD', '1')}")
    for g in range(len(offs) - 1):
        n = offs[g + 1] - offs[g]
        c, r = core[g], rcore[g]
        if r[p + 5] != 0 or c[p + 5] != 0:
            print(f"  g{g} n={n} status {c[p + 5]} / {r[p + 5]}")
            continue
        nanpat = np.array_equal(np.isnan(c[:p]), np.isnan(r[:p]))
        sc = np.nanmax(np.abs(r[:p + 1]))
        err = np.nanmax(np.abs(c[:p + 1] - r[:p + 1]) / np.maximum(np.abs(r[:p + 1]), 1e-3 * sc))
        print(f"  g{g} n={n} nobs={r[p + 4]:.0f} kept={np.sum(~np.isnan(r[:p]))} nanpat_ok={nanpat} coef_err={err:.3e} r2 {c[p + 1]:.12g}/{r[p + 1]:.12g} sigma {c[p + 3]:.6g}/{r[p + 3]:.6g}")
