"""Per-coefficient relative error of one group of one fuzz case (GPU box): python scripts/diag_coef_errors.py <seed> <narrow|wide|very> <group>"""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle
import test_gpu_fuzz as F
from conftest import import_pkg
pkg = import_pkg(); ctx = pkg.Context()
seed = int(sys.argv[1]); kind = {"narrow": False, "wide": True, "very": "very"}[sys.argv[2]]; g = int(sys.argv[3])
p, offs, y, x_cols, w, model, kw, degenerate = F._case(seed, kind)
wv = w if model == "wls" else None
core, inf = pkg.fit_batch_host(offs, y, x_cols, wv, pkg.RegressionOptions(**kw).batch_options(model), ctx=ctx)
rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, model=model, **kw)
sc = np.nanmax(np.abs(rcore[g, :p + 1]))
np.set_printoptions(precision=3, linewidth=220)
print("oracle coef + intercept:", rcore[g, :p + 1])
print("rel err (floor 1e-3 of the largest):", np.abs(core[g, :p + 1] - rcore[g, :p + 1]) / np.maximum(np.abs(rcore[g, :p + 1]), 1e-3 * sc))
# the correction an exact residual asks for at the HIP coefficients (long double residual, least squares in double)
lo, hi = offs[g], offs[g + 1]
X = np.stack([c[lo:hi] for c in x_cols], 1); yy = y[lo:hi]
ok = np.isfinite(yy) & np.isfinite(X).all(1)
if wv is not None:
    ok &= np.isfinite(wv[lo:hi]) & (wv[lo:hi] > 0)
X = X[ok]; yy = yy[ok]
icpt = kw["fit_intercept"]
A = np.hstack([X, np.ones((len(yy), 1))]) if icpt else X
bh = core[g, :p + 1] if icpt else core[g, :p]
live = ~np.isnan(bh)
r = (yy.astype(np.longdouble) - (A[:, live].astype(np.longdouble) * bh[live].astype(np.longdouble)).sum(1)).astype(np.float64)
sw_ = np.sqrt(wv[lo:hi][ok]) if wv is not None else np.ones(len(yy))
d = np.linalg.lstsq(A[:, live] * sw_[:, None], r * sw_, rcond=None)[0]
print("model", model, kw, "rows", len(yy), "cond(A scaled)", np.linalg.cond(A[:, live] / np.linalg.norm(A[:, live], axis=0)))
print("exact step / scale:", np.abs(d) / np.maximum(np.abs(rcore[g, :p + 1][live] if icpt else rcore[g, :p][live]), 1e-3 * sc))
print("max |r|", np.abs(r).max(), "max |y|", np.abs(yy).max())
