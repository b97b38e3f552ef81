import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from conftest import import_pkg
import test_gpu_fuzz as F
pkg = import_pkg(); ctx = pkg.Context()
seed = int(sys.argv[1])
p, offs, y, x_cols, w, model, kw, deg = F._case(seed, False)
wv = w if model == "wls" else None
core, inf = pkg.fit_batch_host(offs, y, x_cols, wv, pkg.RegressionOptions(**kw).batch_options(model), ctx=ctx)
rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, model=model, **kw)
ok = rcore[:, p + 5] == 0
with np.errstate(all="ignore"):
    rel = np.abs(core[:, p + 1] - rcore[:, p + 1]) / np.abs(rcore[:, p + 1])
for g in np.argsort(np.where(ok & np.isfinite(rel), rel, 0))[-4:]:
    print(g, "n", rcore[g, p + 4], "r2 gpu", repr(core[g, p + 1]), "ref", repr(rcore[g, p + 1]), "rel", rel[g], "rse", core[g, p + 3], rcore[g, p + 3], "coef", core[g, :p], rcore[g, :p])
