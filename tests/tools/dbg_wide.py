import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import numpy as np
pkg = importlib.import_module("anofox-statistics_amd")
import oracle
p = int(sys.argv[1]) if len(sys.argv) > 1 else 9
rng = np.random.default_rng(0)
G = 4
ns = rng.integers(p + 5, 3 * p + 40, size=G)
offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
N = int(offs[-1])
x_cols = [rng.uniform(-10, 10, N) for _ in range(p)]
y = sum(0.5 * c for c in x_cols) + rng.standard_normal(N)
print("calling", flush=True)
core, inf = pkg.fit_batch_host(offs, y, x_cols, None, pkg.RegressionOptions(compute_inference=True).batch_options("ols"))
print("status", core[:, p + 5])
rcore, rinf = oracle.fit_groups(y, x_cols, offs, compute_inference=True)
print("max coef diff", np.nanmax(np.abs(core[:, :p+1] - rcore[:, :p+1])))
print("diag", core[:, p+1:p+5], rcore[:, p+1:p+5])
print("inf diff", np.nanmax(np.abs(inf - rinf) / np.abs(rinf)))
