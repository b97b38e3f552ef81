"""GPU tests of the window functions beyond the in-register kernels: more than 8 features, and
explicit RANGE / GROUPS-style frames — every frame fitted as a virtual group of the batch path (csrc/frames.hip).
The oracle refits every frame from scratch (oracle.fit_predict_window, or one oracle.fit per explicit frame)."""
import numpy as np
import pytest

import oracle
from conftest import import_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return import_pkg()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context()
    yield c
    c.close()


def _data(rng, G, p, n_lo, n_hi):
    ns = rng.integers(n_lo, n_hi + 1, size=G)
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    N = int(offs[-1])
    x_cols = [rng.uniform(-10, 10, N) for _ in range(p)]
    gid = np.repeat(np.arange(G), ns)
    beta = rng.uniform(-5, 5, (G, p))
    y = rng.uniform(-10, 10, G)[gid] + sum(beta[gid, j] * x_cols[j] for j in range(p)) + 2.0 * rng.standard_normal(N)
    w = rng.uniform(0.5, 1.5, N)
    y[rng.random(N) < 0.1] = np.nan               # prediction rows (NULL y)
    x_cols[-1][rng.random(N) < 0.01] = np.nan     # NULL feature
    return offs, y, x_cols, w


def _check(pred, ref, what, tol=1e-8):
    assert np.array_equal(np.isnan(pred[:, 0]), np.isnan(ref[:, 0])), f"NULL pattern {what}"
    m = ~np.isnan(ref[:, 0])
    if not m.any():
        return
    scale = np.maximum(np.abs(ref[m, 0]), 1.0)
    err = np.abs(pred[m, 0] - ref[m, 0]) / scale
    assert err.max() < tol, (what, err.max())
    for k in (1, 2):                                  # interval bounds: same tolerance against the larger magnitude
        fin = np.isfinite(ref[m, k])
        e2 = np.abs(pred[m, k][fin] - ref[m, k][fin]) / np.maximum(np.abs(ref[m, k][fin]), scale[fin])
        assert e2.size == 0 or e2.max() < 10 * tol, (what, k, e2.max())


@pytest.mark.parametrize("model", ["ols", "ridge", "wls"])
@pytest.mark.parametrize("p,frame", [(9, (None, 0)), (12, (40, 0)), (20, (None, 1)), (20, (60, -5)), (40, (None, None)),
                                     (33, (90, 0))])
def test_wide_window_frames_match_oracle(pkg, ctx, model, p, frame):
    rng = np.random.default_rng(100 * p + len(model) + abs(frame[0] or 7))
    offs, y, x_cols, w = _data(rng, 4, p, 2 * p, 3 * p + 40)
    for icpt in (True, False):
        kw = dict(fit_intercept=icpt, confidence_level=0.9)
        if model == "ridge":
            kw["alpha"] = 0.5
        wv = w if model == "wls" else None
        opts = pkg.RegressionOptions(**kw).batch_options(model)
        pred = pkg.fit_predict_window_host(offs, y, x_cols, wv, opts, frame, ctx=ctx)
        ref = oracle.fit_predict_window(y, x_cols, offs, w=wv, start_preceding=frame[0], end_preceding=frame[1], model=model, **kw)
        _check(pred, ref, f"{model} p={p} icpt={icpt} frame={frame}")


def _oracle_frames(y, x_cols, w, lo, hi, model, kw):
    """One oracle fit per explicit frame (the window function's rules: ols_fit_predict.cpp:157-162,253-262)."""
    N, p = len(y), len(x_cols)
    icpt = kw.get("fit_intercept", True)
    out = np.full((N, 3), np.nan)
    for e in range(N):
        if hi[e] <= lo[e]:
            continue
        sl = slice(lo[e], hi[e])
        tr = ~np.isnan(y[sl])
        if tr.sum() <= p + int(icpt):
            continue
        code, r = oracle.fit(y[sl][tr], [c[sl][tr] for c in x_cols], w=(w[sl][tr] if w is not None else None), model=model, **kw)
        if code != 0:
            continue
        ok, pr = oracle.predict_with_interval(r["coefficients"], r["intercept"], [c[hi[e] - 1] for c in x_cols],
                                              r["residual_std_error"], r["n_observations"], kw.get("confidence_level", 0.95))
        if ok and np.isfinite(pr[0]):
            out[e] = pr
    return out


@pytest.mark.parametrize("model,p", [("ols", 2), ("wls", 5), ("ols", 11), ("ridge", 18)])
def test_explicit_range_and_groups_frames_match_oracle(pkg, ctx, model, p):
    """RANGE BETWEEN 2.5 PRECEDING AND 1.0 FOLLOWING over a non-uniform ORDER BY key with ties, and GROUPS BETWEEN 3
    PRECEDING AND CURRENT ROW (peer groups of the tied keys): the bounds are computed the way DuckDB's window executor
    resolves them and handed over as explicit row ranges."""
    rng = np.random.default_rng(5 * p + len(model))
    N = 260
    key = np.sort(np.round(rng.uniform(0, 40, N), 1))            # ties included
    part = np.concatenate([np.zeros(150, dtype=int), np.ones(N - 150, dtype=int)])
    x_cols = [rng.uniform(-5, 5, N) for _ in range(p)]
    y = 1.5 + sum((j + 1) * 0.3 * x_cols[j] for j in range(p)) + 0.2 * key + rng.standard_normal(N)
    w = rng.uniform(0.5, 2.0, N)
    y[rng.random(N) < 0.08] = np.nan
    kw = dict(fit_intercept=True, confidence_level=0.95)
    if model == "ridge":
        kw["alpha"] = 0.4
    wv = w if model == "wls" else None
    opts = pkg.RegressionOptions(**kw).batch_options(model)
    lo_r, hi_r, lo_g, hi_g = (np.zeros(N, dtype=np.int64) for _ in range(4))
    for e in range(N):
        rows = np.nonzero(part == part[e])[0]
        k = key[rows]
        inr = rows[(k >= key[e] - 2.5) & (k <= key[e] + 1.0)]
        lo_r[e], hi_r[e] = inr[0], inr[-1] + 1
        peers = np.unique(k)
        gi = np.searchsorted(peers, key[e])
        gsel = rows[(k >= peers[max(0, gi - 3)]) & (k <= key[e])]
        lo_g[e], hi_g[e] = gsel[0], gsel[-1] + 1
    for what, lo, hi in (("RANGE", lo_r, hi_r), ("GROUPS", lo_g, hi_g)):
        pred = pkg.fit_predict_frames_host(y, x_cols, wv, lo, hi, opts, ctx=ctx)
        ref = _oracle_frames(y, x_cols, wv, lo, hi, model, kw)
        _check(pred, ref, f"{what} {model} p={p}")
    # an empty frame and a frame of one row are NULL
    lo2, hi2 = lo_r.copy(), hi_r.copy()
    hi2[3] = lo2[3]
    hi2[7] = lo2[7] + 1
    pred = pkg.fit_predict_frames_host(y, x_cols, wv, lo2, hi2, opts, ctx=ctx)
    assert np.all(np.isnan(pred[3])) and np.all(np.isnan(pred[7]))


def test_frames_argument_errors(pkg, ctx):
    opts = pkg.RegressionOptions().batch_options("ols")
    y = np.arange(5.0)
    with pytest.raises(pkg.AnofoxStatsError):
        pkg.fit_predict_frames_host(y, [y], None, [0, 0, 0, 0, 0], [9, 1, 1, 1, 1], opts, ctx=ctx)     # hi beyond n_rows
    with pytest.raises(pkg.AnofoxStatsError):
        pkg.fit_predict_frames_host(y, [y], None, [0, 0, 0, 0, 0], [5, 5, 5, 5, 5], pkg.RegressionOptions().batch_options("wls"), ctx=ctx)


@pytest.mark.parametrize("model,p,frame", [("ols", 9, (None, 0)), ("wls", 17, (None, 2)), ("ols", 33, (None, -3)), ("ridge", 12, (None, 0)),
                                           ("ols", 100, (None, 0)), ("wls", 128, (None, 1))])
def test_expanding_frames_of_wide_designs_across_block_boundaries(pkg, ctx, model, p, frame):
    """(r4) csrc/accumulate_prefix.hip writes the moment records of `UNBOUNDED PRECEDING` frames incrementally, 128 consecutive frames
    per workgroup: partitions shorter and longer than a workgroup's share, ending on and next to its boundaries, one of a single row;
    the first rows of a partition without y (the first valid row comes later), rows with a NULL feature, weights <= 0; frames that
    end before and after the current row — against the oracle's refit of every frame."""
    rng = np.random.default_rng(7 * p + len(model))
    ns = np.array([1, 127, 128, 129, 2 * p + 150, 3, 64, p + 2, 300])
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    N = int(offs[-1])
    G = len(ns)
    x_cols = [rng.uniform(-10, 10, N) for _ in range(p)]
    gid = np.repeat(np.arange(G), ns)
    beta = rng.uniform(-3, 3, (G, p))
    y = rng.uniform(-10, 10, G)[gid] + sum(beta[gid, j] * x_cols[j] for j in range(p)) + 0.5 * rng.standard_normal(N)
    w = rng.uniform(0.5, 1.5, N)
    y[rng.random(N) < 0.08] = np.nan
    x_cols[p // 2][rng.random(N) < 0.01] = np.nan
    for g in (1, 4):                                   # partitions whose first rows do not train
        y[offs[g]:offs[g] + 3] = np.nan
    w[rng.random(N) < 0.02] = 0.0
    for icpt in (True, False):
        kw = dict(fit_intercept=icpt, confidence_level=0.95)
        if model == "ridge":
            kw["alpha"] = 0.3
        wv = w if model == "wls" else None
        opts = pkg.RegressionOptions(**kw).batch_options(model)
        pred = pkg.fit_predict_window_host(offs, y, x_cols, wv, opts, frame, ctx=ctx)
        ref = oracle.fit_predict_window(y, x_cols, offs, w=wv, start_preceding=frame[0], end_preceding=frame[1], model=model, **kw)
        _check(pred, ref, f"prefix {model} p={p} icpt={icpt} frame={frame}")
