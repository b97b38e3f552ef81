"""Debug helper: for one fuzz seed compare the intercepts of the GPU path and of the oracle with a long-double solve of
the centred normal equations.  Run on a GPU box: python tests/tools/dbg_intercept.py SEED [wide].  Test infrastructure."""
import sys
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
LD = np.longdouble
for g in range(len(offs) - 1):
    if rcore[g, p + 5] != 0 or np.isnan(rcore[g, p]):
        continue
    scale = np.nanmax(np.abs(rcore[g, :p + 1]))
    e = abs(core[g, p] - rcore[g, p]) / max(abs(rcore[g, p]), 1e-3 * scale)
    if e < 3e-10:
        continue
    lo, hi = offs[g], offs[g + 1]
    Xg, yg = X[lo:hi], y[lo:hi]
    wg = np.ones(hi - lo) if wv is None else wv[lo:hi]
    ok = np.isfinite(Xg).all(1) & np.isfinite(yg) & np.isfinite(wg) & (wg > 0)
    keep = ~np.isnan(rcore[g, :p])
    Xg, yg, wg = Xg[ok][:, keep].astype(LD), yg[ok].astype(LD), wg[ok].astype(LD)
    sw = wg.sum()
    xm, ym = (wg[:, None] * Xg).sum(0) / sw, (wg * yg).sum() / sw
    Xc, yc = Xg - xm, yg - ym
    A = (Xc * wg[:, None]).T @ Xc
    if model == "ridge":
        A = A + LD(kw["alpha"]) * np.eye(A.shape[0], dtype=LD)
    b = (Xc * wg[:, None]).T @ yc
    M = np.concatenate([A, b[:, None]], 1)
    q = A.shape[0]
    for i in range(q):
        piv = i + np.argmax(np.abs(M[i:, i])); M[[i, piv]] = M[[piv, i]]
        M[i] /= M[i, i]
        for k in range(q):
            if k != i:
                M[k] -= M[k, i] * M[i]
    beta = M[:, q]
    b0 = float(ym - (beta * xm).sum())
    den = max(abs(b0), 1e-3 * scale)
    print(f"group {g} n={ok.sum()} b0={b0:.6g} scale={scale:.3g} sum|beta*xbar|={float(np.abs(beta * xm).sum()):.3g} "
          f"gpu-oracle={e:.2e} gpu-truth={abs(core[g, p] - b0) / den:.2e} oracle-truth={abs(rcore[g, p] - b0) / den:.2e}")
