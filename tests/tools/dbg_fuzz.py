"""Debug helper: re-run one case of tests/test_gpu_fuzz.py and compare against a long-double solve.
Run from the repo root on a GPU box: python tests/tools/dbg_fuzz.py SEED [wide].  Test infrastructure (uses the oracle)."""
import sys, importlib
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from conftest import import_pkg
import test_gpu_fuzz as F
pkg = import_pkg()
ctx = pkg.Context()
seed = int(sys.argv[1]); wide = len(sys.argv) > 2
p, offs, y, x_cols, w, model, kw, deg = F._case(seed, wide)
wv = w if model == "wls" else None
core, inf = pkg.fit_batch_host(offs, y, x_cols, wv, pkg.RegressionOptions(**kw).batch_options(model), ctx=ctx)
rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, model=model, **kw)
print(model, p, kw)
X = np.stack(x_cols, 1)
for g in range(len(offs) - 1):
    if rcore[g, p + 5] != 0: continue
    c, r = core[g, :p], rcore[g, :p]
    sc = np.nanmax(np.abs(r))
    err = np.nanmax(np.abs(c - r) / np.maximum(np.abs(r), 1e-3 * sc))
    if err > 2e-10:
        lo, hi = offs[g], offs[g + 1]
        Xg, yg = X[lo:hi].astype(np.longdouble), y[lo:hi].astype(np.longdouble)
        ok = np.isfinite(Xg).all(1) & np.isfinite(yg)
        Xg, yg = Xg[ok], yg[ok]
        if model == "ridge" and not kw["fit_intercept"] and kw.get("lambda_scaling", "raw") == "raw":
            A = Xg.T @ Xg + np.longdouble(kw["alpha"]) * np.eye(p, dtype=np.longdouble)
            b = Xg.T @ yg
            # solve in long double by Gaussian elimination
            M = np.concatenate([A, b[:, None]], 1)
            for i in range(p):
                piv = i + np.argmax(np.abs(M[i:, i])); M[[i, piv]] = M[[piv, i]]
                M[i] /= M[i, i]
                for k in range(p):
                    if k != i: M[k] -= M[k, i] * M[i]
            truth = M[:, p].astype(np.float64)
            eg = np.max(np.abs(c - truth) / np.maximum(np.abs(truth), 1e-3 * sc))
            eo = np.max(np.abs(r - truth) / np.maximum(np.abs(truth), 1e-3 * sc))
            print(f"group {g} n={hi-lo} valid={ok.sum()} err(gpu,oracle)={err:.2e} gpu-vs-truth={eg:.2e} oracle-vs-truth={eo:.2e} cond={np.linalg.cond(A.astype(np.float64)):.2e}")
        else:
            print(f"group {g} n={hi-lo} err={err:.2e}")
