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
