"""Row log of the streaming aggregate state: per kind of hard group (tests/test_gpu_streaming.py::_hard_rows), the error
against the oracle of the state with the log, without it, and of the batch entry point (run on the GPU box)."""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle
from conftest import import_pkg
import test_gpu_streaming as T
pkg = import_pkg(); ctx = pkg.Context()
model, p = "ols", 8
rng = np.random.default_rng(4000 + 10 * p + len(model))
G = 200
slot, y, X, w, kind = T._hard_rows(rng, G, p, p + 3, 300)
valid = (rng.random(len(slot)) > 0.05).astype(np.uint8)
kw = T._kw(model, True)
opts = pkg.RegressionOptions(**kw).batch_options(model)
offs, yg, xg, wg = T._grouped(slot, y, X, w, G, keep=valid)
rcore, rinf = oracle.fit_groups(yg, xg, offs, model=model, **kw)
st = pkg.AggState(ctx, p, opts, retain_bytes=1 << 30)
T._feed(st, slot, y, X, None, G, [2048, 1, 777, 5000, 64], valid=valid)
core, inf, unref = st.finalize()
bcore, binf = ctx.fit_batch_host(offs, yg, xg, None, opts)
def err(a, b):
    sc = np.nanmax(np.abs(b[:, :p + 1]), axis=1, keepdims=True)
    return np.nanmax(np.abs(a[:, :p + 1] - b[:, :p + 1]) / np.maximum(np.abs(b[:, :p + 1]), 1e-3 * sc), axis=1)
n = np.diff(offs)
for k in range(4):
    m = (kind == k) & (rcore[:, p + 5] == 0)
    print("kind", k, "groups", m.sum(), "rows", n[m].min(), n[m].max(), "retained-vs-oracle %.2e batch-vs-oracle %.2e retained-vs-batch %.2e" % (err(core, rcore)[m].max(), err(bcore, rcore)[m].max(), err(core, bcore)[m].max()))
g = int(np.nanargmax(np.where(rcore[:, p + 5] == 0, err(core, rcore), 0)))
xs = np.stack([c[offs[g]:offs[g+1]] for c in xg], 1)
A = np.column_stack([np.ones(len(xs)), xs])
print("worst group", g, "kind", kind[g], "rows", n[g], "cond(A) %.3e" % np.linalg.cond(A), "status", core[g, p + 5], bcore[g, p+5], rcore[g, p+5])
pl = pkg.AggState(ctx, p, opts)
T._feed(pl, slot, y, X, None, G, [2048, 1, 777, 5000, 64], valid=valid)
pcore, pinf, punref = pl.finalize()
for k in range(4):
    m = (kind == k) & (rcore[:, p + 5] == 0)
    with np.errstate(all="ignore"):
        ds = np.abs(pcore[m, p + 3] - rcore[m, p + 3]) / np.abs(rcore[m, p + 3])
        dr = np.abs(core[m, p + 3] - rcore[m, p + 3]) / np.abs(rcore[m, p + 3])
    print("kind", k, "plain-vs-oracle coef %.2e sigma %.2e | retained sigma %.2e" % (err(pcore, rcore)[m].max(), np.nanmax(ds[np.isfinite(ds)]) if np.isfinite(ds).any() else 0, np.nanmax(dr[np.isfinite(dr)]) if np.isfinite(dr).any() else 0))
print("plain unrefined", punref)
