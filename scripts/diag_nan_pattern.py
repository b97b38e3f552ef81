"""Groups of one fuzz case whose NaN pattern (dropped columns) differs from the oracle's (GPU box): python scripts/diag_nan_pattern.py <seed> [narrow|wide]"""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle
import test_gpu_fuzz as F
from conftest import import_pkg
pkg = import_pkg(); ctx = pkg.Context()
seed = int(sys.argv[1]); wide = (sys.argv[2] == "wide") if len(sys.argv) > 2 else False
p, offs, y, x_cols, w, model, kw, degenerate = F._case(seed, wide)
wv = w if model == "wls" else None
opts = pkg.RegressionOptions(**kw).batch_options(model)
core, inf = pkg.fit_batch_host(offs, y, x_cols, wv, opts, ctx=ctx)
rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, model=model, **kw)
print("seed", seed, model, "p", p, kw)
np.set_printoptions(precision=6, linewidth=200)
for g in range(len(offs) - 1):
    a, b = np.isnan(core[g, :p]), np.isnan(rcore[g, :p])
    if np.array_equal(a, b) and core[g, p + 5] == rcore[g, p + 5]:
        continue
    lo, hi = offs[g], offs[g + 1]
    X = np.stack([c[lo:hi] for c in x_cols], 1); yy = y[lo:hi]
    ok = np.all(np.isfinite(X), 1) & np.isfinite(yy)
    if wv is not None:
        ok &= np.isfinite(wv[lo:hi]) & (wv[lo:hi] > 0)
    A = X[ok]
    print(f"group {g}: rows {hi - lo}, valid {ok.sum()}, degenerate flag {degenerate[g]}, status hip {core[g, p + 5]} oracle {rcore[g, p + 5]}")
    print("  hip   :", core[g, :p + 1]); print("  oracle:", rcore[g, :p + 1])
    if len(A):
        An = A / np.maximum(np.linalg.norm(A, axis=0), 1e-300)
        print("  column norms", np.linalg.norm(A, axis=0), "singular values of the column-scaled design", np.linalg.svd(An, compute_uv=False))
        print("  max |x - x_first| per column", np.abs(A - A[0]).max(0))
