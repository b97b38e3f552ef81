"""Debug helper: for one wide/narrow fuzz seed list the p-values whose GPU and oracle values differ most, with the
t statistics behind them.  Run on a GPU box: python tests/tools/dbg_pvalues.py SEED [wide].  Test infrastructure."""
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
# inference record: se[p], t[p], pv[p], lo[p], hi[p], f, f_p
t, rt = inf[:, p:2 * p], rinf[:, p:2 * p]
pv, rpv = inf[:, 2 * p:3 * p], rinf[:, 2 * p:3 * p]
with np.errstate(all="ignore"):
    rel = np.abs(pv - rpv) / np.abs(rpv)
rel[~np.isfinite(rel)] = 0
for g, j in zip(*np.unravel_index(np.argsort(rel, axis=None)[-6:], rel.shape)):
    n_obs = rcore[g, p + 4]
    print(f"group {g} coef {j}: n={n_obs:.0f} p_gpu={pv[g, j]:.17g} p_ref={rpv[g, j]:.17g} rel={rel[g, j]:.3g} "
          f"t_gpu={t[g, j]:.17g} t_ref={rt[g, j]:.17g} rel_t={abs(t[g, j] - rt[g, j]) / abs(rt[g, j]):.3g}")
