"""Per-group coefficient error of one fuzz case against the oracle (GPU box): python scripts/diag_fuzz_seed.py <seed> <wide|very|narrow>"""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle
import test_gpu_fuzz as F
from conftest import import_pkg
pkg = import_pkg(); ctx = pkg.Context()
kind = sys.argv[2] if len(sys.argv) > 2 else "very"
kind = {"narrow": False, "wide": True, "very": "very"}[kind]
for a in sys.argv[1:2]:
    seed = int(a)
    p, offs, y, x_cols, w, model, kw, degenerate = F._case(seed, kind)
    wv = w if model == "wls" else None
    opts = pkg.RegressionOptions(**kw).batch_options(model)
    core, inf = pkg.fit_batch_host(offs, y, x_cols, wv, opts, ctx=ctx)
    rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, model=model, **kw)
    pcore, _ = oracle.fit_groups(y, x_cols, offs, w=wv, model=model, plain_qr=True, **kw) if "plain_qr" in oracle.fit_groups.__code__.co_varnames else (rcore, None)
    for g in range(len(offs) - 1):
        if rcore[g, p + 5] != 0:
            continue
        with np.errstate(all="ignore"):
            sc = np.nanmax(np.abs(rcore[g, :p + 1]))
            e = np.nanmax(np.abs(core[g, :p + 1] - rcore[g, :p + 1]) / np.maximum(np.abs(rcore[g, :p + 1]), 1e-3 * sc))
            e2 = np.nanmax(np.abs(pcore[g, :p + 1] - rcore[g, :p + 1]) / np.maximum(np.abs(rcore[g, :p + 1]), 1e-3 * sc))
            X = np.stack([c[offs[g]:offs[g + 1]] for c in x_cols], 1)
            ok = np.all(np.isfinite(X), 1) & np.isfinite(y[offs[g]:offs[g + 1]])
            A = X[ok]
            if kw["fit_intercept"]:
                A = np.column_stack([np.ones(len(A)), A])
            A = A / np.linalg.norm(A, axis=0)
            cond = np.linalg.cond(A)
        if e < 1e-10:
            continue
        print(f"seed {seed} {model} p={p} group {g}: n={int(rcore[g, p + 4])} rows, err HIP {e:.2e}, plain-QR oracle vs refined oracle {e2:.2e}, cond of the column-scaled design {cond:.2e}")
