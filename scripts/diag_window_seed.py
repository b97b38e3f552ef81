"""Rows of one window fuzz case whose NULL pattern / value differs from the oracle (GPU box): python scripts/diag_window_seed.py <seed index>"""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle
from conftest import import_pkg
pkg = import_pkg(); ctx = pkg.Context()
seed = int(sys.argv[1])
rng = np.random.default_rng(40_000 + seed)
p = int(rng.integers(1, 9)); G = int(rng.integers(1, 8))
ns = rng.choice([0, 1, 2, 5, 17, 64, 65, 130], size=G)
offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64); N = int(offs[-1])
X = rng.uniform(-5, 5, (N, p)); y = 1.0 + X @ rng.uniform(-2, 2, p) + 0.3 * rng.standard_normal(N)
y[rng.random(N) < 0.15] = np.nan
X[rng.random(N) < 0.02, int(rng.integers(0, p))] = np.nan
w = rng.uniform(0.3, 2.0, N)
b = int(rng.choice([0, 0, 1, 3, -1, -4])) if rng.random() < 0.9 else None
a = None if rng.random() < 0.35 else (b if b is not None else -5) + int(rng.integers(0, 40))
model = ["ols", "ridge", "wls"][int(rng.integers(0, 3))]
kw = dict(fit_intercept=bool(rng.integers(0, 2)), confidence_level=0.9)
if model == "ridge":
    kw["alpha"] = float(10.0 ** rng.uniform(-2, 0.5))
wv = w if model == "wls" else None
x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
pred = pkg.fit_predict_window_host(offs, y, x_cols, wv, pkg.RegressionOptions(**kw).batch_options(model), (a, b), ctx=ctx)
ref = oracle.fit_predict_window(y, x_cols, offs, w=wv, start_preceding=a, end_preceding=b, model=model, **kw)
print("seed", seed, model, "p", p, "frame", (a, b), kw, "ns", ns)
np.set_printoptions(precision=8, linewidth=200)
for i in np.nonzero(np.isnan(pred[:, 0]) != np.isnan(ref[:, 0]))[0]:
    g = int(np.searchsorted(offs, i, side="right") - 1)
    lo = max(offs[g], i - a) if a is not None else offs[g]
    hi = min(offs[g + 1], i - b + 1) if b is not None else offs[g + 1]
    print(f"row {i} (group {g}, rows {offs[g]}..{offs[g + 1]}): hip {pred[i]} oracle {ref[i]}; frame rows {lo}..{hi}")
    print("   y", y[lo:hi]); print("   X", X[lo:hi].T)
