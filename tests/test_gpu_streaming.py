"""GPU tests of the streaming aggregate state (anofox_hip_agg_state_*): Update with rows
arriving in shuffled group order across many chunks, Combine of partial states, Finalize — against the oracle's fit
of the same rows grouped (the reference buffers the rows, src/aggregate_functions/ols_aggregate.cpp:120-338, so its
result is the fit of each group's rows in arrival order)."""
import os
import warnings

import numpy as np
import pytest

import oracle
from conftest import COEF_RTOL, DIAG_RTOL, ROOT, assert_records_match, import_pkg, load_csv, load_json, nan_or

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return import_pkg()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context()
    yield c
    c.close()


def _rows(rng, G, p, n_lo, n_hi, offset=0.0):
    """Rows of G groups in ARRIVAL order (shuffled): slot, y, X[n, p], w."""
    ns = rng.integers(n_lo, n_hi + 1, size=G)
    slot = np.repeat(np.arange(G, dtype=np.uint32), ns)
    rng.shuffle(slot)
    N = len(slot)
    X = rng.uniform(-10, 10, (N, p)) + offset
    beta = rng.uniform(-5, 5, (G, p))
    b0 = rng.uniform(-10, 10, G)
    y = b0[slot] + np.einsum("ij,ij->i", beta[slot], X) + 2.0 * rng.standard_normal(N)
    w = rng.uniform(0.5, 1.5, N)
    return slot, y, X, w


def _grouped(slot, y, X, w, G, keep=None):
    """What the reference's state holds at Finalize: each group's accepted rows in arrival order."""
    idx = np.arange(len(slot)) if keep is None else np.nonzero(keep)[0]
    order = idx[np.argsort(slot[idx], kind="stable")]
    counts = np.bincount(slot[idx], minlength=G)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return offs, y[order], [np.ascontiguousarray(X[order, j]) for j in range(X.shape[1])], w[order]


def _kw(model, icpt, inference=True):
    kw = dict(fit_intercept=icpt, compute_inference=inference, confidence_level=0.9)
    if model == "ridge":
        kw["alpha"] = 1.5
    return kw


def _feed(st, slot, y, X, w, G, sizes, valid=None):
    r0 = 0
    k = 0
    while r0 < len(slot):
        n = sizes[k % len(sizes)]
        k += 1
        sl = slice(r0, r0 + n)
        st.update(slot[sl], y[sl], X[sl], None if w is None else w[sl], None if valid is None else valid[sl], n_slots=G)
        r0 += n


@pytest.mark.parametrize("model", ["ols", "ridge", "wls"])
@pytest.mark.parametrize("p", [1, 3, 8])
@pytest.mark.parametrize("icpt", [True, False])
def test_shuffled_chunks_match_oracle(pkg, ctx, model, p, icpt):
    rng = np.random.default_rng(100 * p + 10 * len(model) + int(icpt))
    G = 300
    slot, y, X, w = _rows(rng, G, p, 2, 400)
    kw = _kw(model, icpt)
    st = pkg.AggState(ctx, p, pkg.RegressionOptions(**kw).batch_options(model))
    wv = w if model == "wls" else None
    _feed(st, slot, y, X, wv, G, [2048, 1, 777, 5000, 64])          # DuckDB-sized vectors and odd ones
    core, inf, unref = st.finalize()
    offs, yg, xg, wg = _grouped(slot, y, X, w, G)
    rcore, rinf = oracle.fit_groups(yg, xg, offs, w=(wg if model == "wls" else None), model=model, **kw)
    assert core.shape == (G, p + 6) and st.n_rows == len(slot)
    zero_df = [g for g in range(G) if rcore[g, p + 5] == 0 and rcore[g, p + 4] <= p + int(icpt)]
    assert_records_match(core, rcore, p, inf, rinf, what=f"streaming {model} p={p} icpt={icpt}", skip_diag_groups=zero_df)
    st.close()


@pytest.mark.parametrize("model", ["ols", "wls"])
def test_one_call_with_millions_of_rows_and_huge_runs(pkg, ctx, model):
    """One update call above kIngestChunkRows (sub-chunked inside the library), rows partly sorted so that single
    groups contribute runs far longer than kIngestPieceRows (the piece path), partly shuffled."""
    rng = np.random.default_rng(9)
    p, G = 4, 40
    ns = np.full(G, 30_000)
    ns[0] = 900_000                       # one group is most of the batch: hundreds of pieces per chunk
    ns[1] = 2                             # and tiny ones
    ns[2] = 1
    slot = np.repeat(np.arange(G, dtype=np.uint32), ns)      # sorted arrival: long runs
    tail = slot[-300_000:].copy()
    rng.shuffle(tail)
    slot[-300_000:] = tail
    N = len(slot)
    assert N > (1 << 20)
    X = rng.uniform(-10, 10, (N, p)) + 50.0
    beta = rng.uniform(-5, 5, (G, p))
    y = np.einsum("ij,ij->i", beta[slot], X) + 3.0 + 2.0 * rng.standard_normal(N)
    w = rng.uniform(0.5, 1.5, N)
    kw = _kw(model, True)
    st = pkg.AggState(ctx, p, pkg.RegressionOptions(**kw).batch_options(model))
    st.update(slot, y, X, w if model == "wls" else None, n_slots=G)
    core, inf, _ = st.finalize()
    offs, yg, xg, wg = _grouped(slot, y, X, w, G)
    rcore, rinf = oracle.fit_groups(yg, xg, offs, w=(wg if model == "wls" else None), model=model, n_threads=8, **kw)
    assert_records_match(core, rcore, p, inf, rinf, what=f"streaming huge runs {model}")
    assert core[2, p + 5] == 100 and core[1, p + 5] == 6      # one row -> NULL; two rows, five parameters -> InsufficientData
    st.close()


def test_skipped_rows_invalid_values_and_empty_slots(pkg, ctx):
    """valid = 0 rows are not accumulated at all (Update's NULL skip: they do not count towards the '< 2 rows -> NULL'
    rule); NaN / inf values are accumulated rows that the fit's row filter drops (ols.rs:59-66); slots that never
    receive an accepted row come out NULL; constant and collinear columns as in the batch path."""
    rng = np.random.default_rng(3)
    p, G = 3, 64
    slot, y, X, w = _rows(rng, G, p, 1, 60)
    N = len(slot)
    valid = (rng.uniform(size=N) > 0.15).astype(np.uint8)
    valid[slot == 5] = 0                               # slot 5: every row skipped -> NULL (0 accumulated rows)
    y[rng.uniform(size=N) < 0.05] = np.nan             # accumulated, then filtered
    X[rng.uniform(size=N) < 0.03, 1] = np.inf
    w[rng.uniform(size=N) < 0.05] = -1.0               # WLS: non-positive weight -> filtered
    X[slot == 7, 2] = 4.25                             # constant column -> NaN coefficient
    X[slot == 9, 1] = 2.0 * X[slot == 9, 0] + 1.0      # collinear -> aliased
    y[slot == 11] = np.nan                             # no valid row at all -> NoValidData
    kw = _kw("wls", True)
    st = pkg.AggState(ctx, p, pkg.RegressionOptions(**kw).batch_options("wls"))
    _feed(st, slot, y, X, w, G + 3, [500, 33], valid=valid)      # 3 slots beyond the last one that gets rows
    core, inf, _ = st.finalize()
    assert core.shape[0] == G + 3
    keep = valid != 0
    offs, yg, xg, wg = _grouped(slot, y, X, w, G + 3, keep=keep)
    rcore, rinf = oracle.fit_groups(yg, xg, offs, w=wg, model="wls", **kw)
    zero_df = [g for g in range(G + 3) if rcore[g, p + 5] == 0 and rcore[g, p + 4] <= p + 1]
    assert_records_match(core, rcore, p, inf, rinf, what="streaming NULLs", skip_diag_groups=zero_df)
    assert core[5, p + 5] == 100 and np.all(core[G:, p + 5] == 100)
    assert core[11, p + 5] == 10          # NoValidData
    assert np.isnan(core[7, 2]) and core[7, p + 5] == 0
    st.close()


@pytest.mark.parametrize("model,icpt", [("ols", True), ("wls", True), ("ridge", False)])
def test_combine_merges_partial_states(pkg, ctx, model, icpt):
    """Three thread-local hash tables share one device state: keys overlap, Combine merges (source rows count as
    arriving after the target's), and the result equals the oracle's fit of the concatenated buffers."""
    rng = np.random.default_rng(17 + len(model))
    p, G = 5, 120
    kw = _kw(model, icpt)
    parts = []
    for t in range(3):
        slot, y, X, w = _rows(rng, G, p, 0, 90, offset=20.0 * t)
        present = rng.uniform(size=G) < 0.8                      # not every key is seen by every thread
        m = present[slot]
        parts.append((slot[m], y[m], X[m], w[m]))
    st = pkg.AggState(ctx, p, pkg.RegressionOptions(**kw).batch_options(model))
    # thread t's state of key g lives in slot t * G + g
    for t, (slot, y, X, w) in enumerate(parts):
        _feed(st, slot + np.uint32(t * G), y, X, w if model == "wls" else None, 3 * G, [1000, 17])
    ar = np.arange(G, dtype=np.uint32)
    st.combine(ar + G, ar)                 # thread 1 into thread 0
    st.combine(ar + 2 * G, ar)             # thread 2 into thread 0
    core, inf, _ = st.finalize()
    slot = np.concatenate([q[0] for q in parts])
    y = np.concatenate([q[1] for q in parts])
    X = np.concatenate([q[2] for q in parts])
    w = np.concatenate([q[3] for q in parts])
    offs, yg, xg, wg = _grouped(slot, y, X, w, G)                # stable: thread 0's rows, then 1's, then 2's
    rcore, rinf = oracle.fit_groups(yg, xg, offs, w=(wg if model == "wls" else None), model=model, **kw)
    zero_df = [g for g in range(G) if rcore[g, p + 5] == 0 and rcore[g, p + 4] <= p + 1]
    assert_records_match(core[:G], rcore, p, inf[:G], rinf, what=f"combine {model}", skip_diag_groups=zero_df)
    assert np.all(core[G:, p + 5] == 100)                        # the sources were emptied
    with pytest.raises(pkg.AnofoxStatsError):
        st.combine([1, 1], [2, 3])                               # a slot twice in one call
    with pytest.raises(pkg.AnofoxStatsError):
        st.combine([1], [3 * G + 5])                             # out of range
    st.close()


def test_creation_and_argument_errors(pkg, ctx):
    o = pkg.RegressionOptions().batch_options("ols")
    with pytest.raises(pkg.AnofoxStatsError) as ei:
        pkg.AggState(ctx, 129, o)
    assert ei.value.code == 1 and "features" in str(ei.value)
    with pytest.raises(pkg.AnofoxStatsError):
        pkg.AggState(ctx, 0, o)
    st = pkg.AggState(ctx, 2, pkg.RegressionOptions().batch_options("wls"))
    with pytest.raises(pkg.AnofoxStatsError):
        st.update([0], [1.0], [[1.0, 2.0]], None, n_slots=1)     # WLS without weights
    st.close()
    st = pkg.AggState(ctx, 2, o)
    st.update([0, 0, 0, 7], [1.0, 2.0, 3.5, 1.0], [[1, 2], [2, 1], [3, 5], [0, 0]], n_slots=4)   # slot 7 >= n_slots
    with pytest.raises(pkg.AnofoxStatsError) as ei:
        st.finalize()
    assert "slot index" in str(ei.value)
    st.close()
    st = pkg.AggState(ctx, 2, o)                                 # nothing ever updated
    core, inf, _ = st.finalize()
    assert core.shape == (0, 8) and inf is None
    st.close()


def test_device_chunks_and_agreement_with_the_batch_path(pkg, ctx):
    """update_device on shuffled device-resident chunks at a size the batch path fits in milliseconds: every group
    against the batch kernels (an independent accumulation order, so agreement is to rounding, not bit for bit), a
    sample against the oracle."""
    import torch
    synth = import_pkg("synth")
    G, n, p = 50_000, 64, 8
    offs, y, x_cols, w = synth.make_grouped(G, n, p, weights=True, device="cuda:0")
    N = G * n
    perm = torch.randperm(N, device="cuda:0")
    slot = (perm // n).to(torch.int32)
    X = torch.stack(x_cols, dim=1)[perm].contiguous()
    ys, ws = y[perm].contiguous(), w[perm].contiguous()
    for model in ("ols", "wls"):
        opts = pkg.RegressionOptions(compute_inference=True).batch_options(model)
        st = pkg.AggState(ctx, p, opts, initial_slots=G)
        step = 700_001
        for r0 in range(0, N, step):
            sl = slice(r0, min(N, r0 + step))
            st.update_device(slot[sl], ys[sl], X[sl], ws[sl] if model == "wls" else None, n_slots=G)
        core = torch.empty((G, p + 6), dtype=torch.float64, device="cuda:0")
        inf = torch.empty((G, 5 * p + 2), dtype=torch.float64, device="cuda:0")
        st.finalize_device(core, inf)
        bcore, binf = ctx.fit_batch_device(offs, y, x_cols, w if model == "wls" else None, opts)
        torch.cuda.synchronize()
        assert bool((core[:, p + 5] == 0).all()) and bool((core[:, p + 4] == n).all())
        scale = bcore[:, :p + 1].abs().max(dim=1, keepdim=True).values
        assert float(((core[:, :p + 1] - bcore[:, :p + 1]).abs() / torch.maximum(bcore[:, :p + 1].abs(), 1e-3 * scale)).max()) < COEF_RTOL
        assert float((core[:, p + 1:p + 4] / bcore[:, p + 1:p + 4] - 1.0).abs().max()) < DIAG_RTOL
        assert float(((inf - binf).abs() / binf.abs().clamp_min(1e-300))[:, :2 * p].max()) < DIAG_RTOL
        # the oracle sees a group's rows in arrival order = ascending position in the permutation
        S = 64
        rows = torch.nonzero(slot < S).squeeze(1)
        so, ysr, Xs, wsr = slot[rows].cpu().numpy().astype(np.uint32), ys[rows].cpu().numpy(), X[rows].cpu().numpy(), ws[rows].cpu().numpy()
        go, yg, xg, wg = _grouped(so, ysr, Xs, wsr, S)
        rcore, rinf = oracle.fit_groups(yg, xg, go, w=(wg if model == "wls" else None), model=model, compute_inference=True)
        assert_records_match(core[:S].cpu().numpy(), rcore, p, inf[:S].cpu().numpy(), rinf, what=f"streaming device {model}")
        st.close()


@pytest.mark.parametrize("case,xn,icpt", [("simple_linear", ["x"], True), ("multiple_regression", ["x1", "x2", "x3"], True),
                                          ("no_intercept", ["x"], False), ("rank_deficient", ["x1", "x2"], True),
                                          ("perfect_collinearity", ["x1", "x2"], True)])
def test_reference_fixtures_through_the_streaming_aggregate(pkg, ctx, case, xn, icpt):
    """The R fixtures of test/data/ols_tests through the aggregate mirror with a GPU-resident state: the fixture's
    rows (group 1) arrive interleaved with rows of another group, in small updates; the fixture's rows keep their
    own order."""
    from conftest import rel_err
    d = load_csv(f"ols_tests/input/{case}.csv")
    e = load_json(f"ols_tests/expected/{case}.json")
    X = np.stack([d[c] for c in xn], axis=1)
    y = d["y"]
    n = len(y)
    rng = np.random.default_rng(1)
    agg = pkg.OlsFitAgg({"intercept": icpt}, ctx, streaming=True)
    Xa = np.concatenate([X, rng.uniform(-1, 1, (n, len(xn)))])
    ya = np.concatenate([y, rng.uniform(-1, 1, n)])
    keys = np.concatenate([np.full(n, 1), np.full(n, 2)])
    order = np.argsort(np.concatenate([np.arange(n) + 0.25, rng.uniform(0, n, n)]), kind="stable")
    for c0 in range(0, 2 * n, 37):
        sel = order[c0:c0 + 37]
        agg.update(keys[sel], ya[sel], Xa[sel])
    res = agg.finalize()
    assert list(res.keys) == [1, 2]
    r = res.row(0)
    assert r is not None
    coefs = e["coefficients"] if isinstance(e["coefficients"], list) else [e["coefficients"]]
    coefs = [nan_or(c) for c in coefs]
    slopes = coefs[1:] if icpt else coefs
    if icpt:
        assert rel_err(r["intercept"], coefs[0]) < COEF_RTOL
    else:
        assert np.isnan(r["intercept"])
    for got, want in zip(r["coefficients"], slopes):
        assert (np.isnan(got) and np.isnan(want)) or rel_err(got, want) < COEF_RTOL
    assert rel_err(r["r_squared"], e["r_squared"]) < DIAG_RTOL
    assert rel_err(r["adj_r_squared"], e["adj_r_squared"]) < DIAG_RTOL
    assert rel_err(r["residual_std_error"], e["sigma"]) < DIAG_RTOL
    assert r["n_observations"] == n and r["n_features"] == len(xn)


def test_streaming_mirror_equals_buffered_mirror_with_combine(pkg, ctx):
    """OlsFitAgg / WlsFitAgg with streaming=<pool>: two 'threads' update disjoint row sets with NULLs in y, x and w,
    Combine, Finalize — same SQL-level result as the buffered mirror (same keys, same NULL groups, values to 1e-9)."""
    rng = np.random.default_rng(23)
    n, p = 4000, 3
    keys = rng.integers(0, 50, n)
    X = rng.uniform(-3, 3, (n, p))
    y = X @ np.array([1.0, -2.0, 0.5]) + 0.1 * keys + rng.standard_normal(n)
    w = rng.uniform(0.5, 2.0, n)
    yl = [None if rng.uniform() < 0.05 else float(v) for v in y]
    xl = [None if rng.uniform() < 0.05 else r.tolist() for r in X]
    wl = [None if rng.uniform() < 0.05 else float(v) for v in w]
    keys[-3:] = [900, 901, 901]                       # a one-row group and a two-row group
    for cls, has_w in ((pkg.OlsFitAgg, False), (pkg.WlsFitAgg, True)):
        opts = {"compute_inference": True}
        pool = pkg.StreamingStates(ctx)
        a, b = cls(opts, ctx, streaming=pool), cls(opts, ctx, streaming=pool)
        ra, rb = cls(opts, ctx), cls(opts, ctx)
        half = n // 2
        for agg in (a, ra):
            for c0 in range(0, half, 512):
                sl = slice(c0, min(half, c0 + 512))
                agg.update(keys[sl], yl[sl], xl[sl], *([wl[sl]] if has_w else []))
        for agg in (b, rb):
            for c0 in range(half, n, 512):
                sl = slice(c0, min(n, c0 + 512))
                agg.update(keys[sl], yl[sl], xl[sl], *([wl[sl]] if has_w else []))
        got = a.combine(b).finalize()
        want = ra.combine(rb).finalize()
        assert np.array_equal(got.keys, want.keys) and np.array_equal(got.is_null, want.is_null)
        assert np.array_equal(got.status, want.status) and np.array_equal(got.n_observations, want.n_observations)
        ok = ~want.is_null
        for f in ("coefficients", "intercept", "r_squared", "residual_std_error", "std_errors", "t_values", "f_statistic"):
            g, r = getattr(got, f)[ok], getattr(want, f)[ok]
            assert np.allclose(g, r, rtol=1e-9, atol=1e-12, equal_nan=True), f


# ---- the optional row log: Finalize refits the groups its solve queued (anofox_hip_agg_state_retain_rows) ----

def _hard_rows(rng, G, p, n_lo, n_hi):
    """Groups the moments alone do not resolve to the ordinary tolerances: nearly collinear columns (pivot ratio far
    below 1e-3), nearly exact fits (rss / tss ~ 1e-12), square systems (rows == parameters) — mixed with easy ones."""
    ns = rng.integers(n_lo, n_hi + 1, size=G)
    kind = rng.integers(0, 4, size=G)
    ns[kind == 3] = p + 1                                     # exact interpolation with an intercept
    slot = np.repeat(np.arange(G, dtype=np.uint32), ns)
    rng.shuffle(slot)
    N = len(slot)
    X = rng.uniform(-10, 10, (N, p)) + 5.0
    if p >= 2:
        near = kind[slot] == 1
        # cond(A) ~ 3e4
        X[near, p - 1] = X[near, 0] * 1.5 + 1e-3 * rng.standard_normal(int(near.sum()))
    beta = rng.uniform(-5, 5, (G, p))
    noise = np.where(kind[slot] == 2, 1e-6, 1.0) * rng.standard_normal(N)
    y = rng.uniform(-10, 10, G)[slot] + np.einsum("ij,ij->i", beta[slot], X) + noise
    w = rng.uniform(0.5, 1.5, N)
    return slot, y, X, w, kind


def _coef_err(a, b, p):
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)          # unfitted groups: all-NaN rows
        sc = np.nanmax(np.abs(b[:, :p + 1]), axis=1, keepdims=True)
        return np.nanmax(np.abs(a[:, :p + 1] - b[:, :p + 1]) / np.maximum(np.abs(b[:, :p + 1]), 1e-3 * sc), axis=1)


@pytest.mark.parametrize("model", ["ols", "ridge", "wls"])
@pytest.mark.parametrize("p", [2, 5, 8])
def test_retained_rows_refit_the_queued_groups(pkg, ctx, model, p):
    """Measured on MI355X for ols p = 8 (scripts/diag_retained.py, profiles/r02_stream_unrefined.md):
       nearly collinear groups (cond 3e4)  coefficients  4.2e-5 without the log, 6.1e-9 with it (= the batch path, bit for bit)
       nearly exact fits (noise 1e-6)      sigma         2.3 (relative!) without the log, 5.2e-9 with it."""
    rng = np.random.default_rng(4000 + 10 * p + len(model))
    G = 200
    slot, y, X, w, kind = _hard_rows(rng, G, p, p + 3, 300)
    valid = (rng.random(len(slot)) > 0.05).astype(np.uint8)
    kw = _kw(model, True)
    if model == "ridge":
        kw["alpha"] = 1e-6                                    # (a large penalty would regularise the hard groups away)
    opts = pkg.RegressionOptions(**kw).batch_options(model)
    wv = w if model == "wls" else None
    offs, yg, xg, wg = _grouped(slot, y, X, w, G, keep=valid)
    rcore, rinf = oracle.fit_groups(yg, xg, offs, w=(wg if model == "wls" else None), model=model, **kw)
    fitted = rcore[:, p + 5] == 0
    zero_df = {g for g in range(G) if fitted[g] and rcore[g, p + 4] <= p + 1}

    plain = pkg.AggState(ctx, p, opts, retain_bytes=0, retain_host_bytes=0)      # moments only
    _feed(plain, slot, y, X, wv, G, [2048, 1, 777, 5000, 64], valid=valid)
    pcore, _, unref_plain = plain.finalize()
    assert unref_plain > G // 4 and not plain.retaining and len(plain.unrefined_slots) == unref_plain
    # what the moments alone cannot resolve is flagged (status 101, NaN record), everything else meets the contract
    u = plain.unrefined_slots
    assert np.all(pcore[u, p + 5] == 101) and np.all(np.isnan(pcore[u, :p + 5]))
    rest = np.setdiff1d(np.arange(G), u)
    assert_records_match(pcore[rest], rcore[rest], p, None, None, what=f"moments only, resolved groups {model} p={p}",
                         skip_diag_groups=[k for k, g in enumerate(rest) if int(g) in zero_df])
    plain.close()

    st = pkg.AggState(ctx, p, opts, retain_bytes=1 << 30)
    _feed(st, slot, y, X, wv, G, [2048, 1, 777, 5000, 64], valid=valid)
    assert st.retaining and st.retained_bytes >= len(slot) * (8 * (p + 1) + 5)
    core, inf, unref = st.finalize()
    assert unref == 0 and len(st.unrefined_slots) == 0

    def check(a_core, a_inf, idx, what, **tol):
        idx = np.asarray(idx)
        assert_records_match(core[idx], a_core[idx], p, inf[idx], a_inf[idx], what=f"{what} {model} p={p}",
                             skip_diag_groups=[k for k, g in enumerate(idx) if int(g) in zero_df], **tol)

    # the refit IS the batch entry point's path on a sub-batch: same records (bit for bit on the refitted groups,
    # rounding-level on the others, whose moments were summed in another order)
    bcore, binf = ctx.fit_batch_host(offs, yg, xg, wg if model == "wls" else None, opts)
    check(bcore, binf, np.arange(G), "retained vs batch", coef_rtol=1e-10, diag_rtol=1e-8)
    # against the oracle: ordinary tolerances, except the coefficients of the cond 3e4 groups, where the batch path
    # itself sits at 6e-9 (per-coefficient error with a 1e-3 normwise floor; 7e-12 normwise = cond eps)
    collinear = np.nonzero(kind == 1)[0] if p >= 2 else np.empty(0, dtype=int)
    check(rcore, rinf, np.setdiff1d(np.arange(G), collinear), "retained")
    if collinear.size:
        check(rcore, rinf, collinear, "retained, collinear", coef_rtol=1e-7)
    # and the log is what bought that: without it the hard groups (nearly collinear, nearly exact) are flagged NULL
    # (status 101) — round 2 handed out their values: coefficients 1e-5 off, sigma off by its own size
    m1 = fitted & (kind == 1)
    m2 = fitted & (kind == 2) & ~np.isin(np.arange(G), sorted(zero_df))
    assert np.all(pcore[m2, p + 5] == 101) and (not m1.any() or np.mean(pcore[m1, p + 5] == 101) > 0.5)
    assert np.all(core[m1 | m2, p + 5] == 0)
    d = np.abs(core[m2, p + 3] - rcore[m2, p + 3]) / rcore[m2, p + 3]
    assert np.max(d) < 1e-6
    core2, inf2, unref2 = st.finalize()                        # Finalize does not consume the state or its log
    assert unref2 == 0 and np.array_equal(core, core2, equal_nan=True) and np.array_equal(inf, inf2, equal_nan=True)
    st.close()


def test_retained_rows_follow_combine_and_span_slabs(pkg, ctx):
    """More rows than the first slab holds (65536), two partial states per key merged by Combine: the refit sees the
    rows of both under the target's slot; torch-side finalize_device takes the same path."""
    import torch
    rng = np.random.default_rng(4100)
    p, G = 4, 600
    kw = _kw("ols", True)
    parts = [_hard_rows(rng, G, p, p + 3, 250)[:4] for _ in range(2)]
    assert sum(len(q[0]) for q in parts) > 100_000
    st = pkg.AggState(ctx, p, pkg.RegressionOptions(**kw).batch_options("ols"), retain_bytes=1 << 30)
    for t, (slot, y, X, w) in enumerate(parts):
        _feed(st, slot + np.uint32(t * G), y, X, None, 2 * G, [30_000, 2048, 3])
    ar = np.arange(G, dtype=np.uint32)
    st.combine(ar + G, ar)
    core, inf, unref = st.finalize()
    slot = np.concatenate([q[0] for q in parts])
    y = np.concatenate([q[1] for q in parts])
    X = np.concatenate([q[2] for q in parts])
    offs, yg, xg, _ = _grouped(slot, y, X, np.ones(len(slot)), G)
    rcore, rinf = oracle.fit_groups(yg, xg, offs, model="ols", **kw)
    assert unref == 0
    assert_records_match(core[:G], rcore, p, inf[:G], rinf, what="retained + combine")
    assert np.all(core[G:, p + 5] == 100)
    dcore = torch.empty((2 * G, p + 6), dtype=torch.float64, device="cuda")
    dinf = torch.empty((2 * G, 5 * p + 2), dtype=torch.float64, device="cuda")
    st.finalize_device(dcore, dinf)
    torch.cuda.synchronize()
    assert np.array_equal(dcore.cpu().numpy(), core, equal_nan=True) and np.array_equal(dinf.cpu().numpy(), inf, equal_nan=True)
    st.close()


def test_row_log_budget_and_call_order(pkg, ctx):
    rng = np.random.default_rng(4200)
    p, G = 3, 50
    slot, y, X, w, _ = _hard_rows(rng, G, p, p + 3, 200)
    kw = _kw("ols", True)
    opts = pkg.RegressionOptions(**kw).batch_options("ols")
    ref = pkg.AggState(ctx, p, opts, retain_bytes=0, retain_host_bytes=0)
    _feed(ref, slot, y, X, None, G, [512])
    rcore, rinf, runref = ref.finalize()
    rlist = ref.unrefined_slots.copy()
    ref.close()
    st = pkg.AggState(ctx, p, opts, retain_bytes=1000 * (8 * (p + 1) + 5), retain_host_bytes=0)     # room for 1000 rows only
    _feed(st, slot, y, X, None, G, [512])
    assert len(slot) > 1000 and not st.retaining and st.retained_bytes == 0    # dropped, not an error
    core, inf, unref = st.finalize()
    assert unref == runref > 0 and np.array_equal(st.unrefined_slots, rlist)
    assert np.array_equal(core, rcore, equal_nan=True) and np.array_equal(inf, rinf, equal_nan=True)
    assert np.all(core[rlist, p + 5] == 101)
    lib = pkg._abi.load()
    err = pkg._abi.AnofoxError()
    import ctypes
    assert not lib.anofox_hip_agg_state_retain_rows(st._h, 1 << 20, ctypes.byref(err)) and "before the first update" in err.text()
    assert not lib.anofox_hip_agg_state_retain_rows_host(st._h, 1 << 20, ctypes.byref(err)) and "before the first update" in err.text()
    st.close()
    # the same HBM budget with a host budget behind it: the log continues in page-locked host memory, nothing is
    # dropped or flagged, and the records are the batch entry point's
    sp = pkg.AggState(ctx, p, opts, retain_bytes=1000 * (8 * (p + 1) + 5), retain_host_bytes=1 << 28)
    _feed(sp, slot, y, X, None, G, [512])
    assert sp.retaining and 0 < sp.retained_bytes <= 1000 * (8 * (p + 1) + 5) and sp.retained_host_bytes >= (len(slot) - 1000) * (8 * (p + 1) + 5)
    score, sinf, sunref = sp.finalize()
    assert sunref == 0 and np.all(score[:, p + 5] == 0)
    offs, yg, xg, _ = _grouped(slot, y, X, np.ones(len(slot)), G)
    bcore, binf = ctx.fit_batch_host(offs, yg, xg, None, opts)
    assert_records_match(score, bcore, p, sinf, binf, what="row log spilled to host", coef_rtol=1e-10, diag_rtol=1e-8)
    sp.close()


# ---- log-only states: designs wider than 8 features and HC errors keep the rows, not moments ----

@pytest.mark.parametrize("model,p,hc", [("ols", 12, None), ("wls", 20, None), ("ridge", 40, None), ("ols", 128, None),
                                        ("ols", 5, "hc1"), ("wls", 8, "hc3"), ("ols", 17, "hc2")])
def test_log_only_states_wide_designs_and_hc_errors(pkg, ctx, model, p, hc):
    """No O(p^2) record to stream into for p > 8, and HC errors need a second pass over the rows: the state then IS the
    row log in HBM (the reference's per-group row buffers, ols_aggregate.cpp:19-42), Update appends, Combine re-labels,
    Finalize runs the batch path over all of it.  Same entry points; against the oracle's fit of each key's rows in
    the reference's order (thread 0's, then thread 1's)."""
    rng = np.random.default_rng(5000 + 7 * p + len(model))
    G = 60 if p <= 40 else 12
    kw = _kw(model, True)
    if hc:
        kw["hc_type"] = hc
    opts = pkg.RegressionOptions(**kw).batch_options(model)
    parts = [_rows(rng, G, p, 0 if t else 2, 3 * p + 40, offset=3.0 * t) for t in range(2)]
    valid = [(rng.random(len(q[0])) > 0.07).astype(np.uint8) for q in parts]
    for q in parts:                                              # NaN / inf values are accumulated rows the fit drops
        q[2][rng.random(q[2].shape[0]) < 0.01, 0] = np.nan
    st = pkg.AggState(ctx, p, opts)
    assert st.retaining
    for t, (slot, y, X, w) in enumerate(parts):
        _feed(st, slot + np.uint32(t * G), y, X, w if model == "wls" else None, 2 * G, [2048, 1, 333], valid=valid[t])
    ar = np.arange(G, dtype=np.uint32)
    st.combine(ar + G, ar)
    core, inf, unref = st.finalize()
    assert unref == 0 and st.retained_bytes > 0
    slot = np.concatenate([q[0] for q in parts])
    y = np.concatenate([q[1] for q in parts])
    X = np.concatenate([q[2] for q in parts])
    w = np.concatenate([q[3] for q in parts])
    keep = np.concatenate(valid)
    offs, yg, xg, wg = _grouped(slot, y, X, w, G, keep=keep)
    rcore, rinf = oracle.fit_groups(yg, xg, offs, w=(wg if model == "wls" else None), model=model, n_threads=8, **kw)
    slack = 3 if hc in ("hc2", "hc3") else 0
    n_par = np.sum(~np.isnan(rcore[:, :p]), axis=1) + 1
    skip = [g for g in range(G) if rcore[g, p + 5] == 0 and rcore[g, p + 4] - n_par[g] <= slack]
    assert_records_match(core[:G], rcore, p, inf[:G], rinf, what=f"log-only {model} p={p} hc={hc}", skip_diag_groups=skip)
    assert np.all(core[G:, p + 5] == 100)                        # the sources were emptied
    assert (rcore[:, p + 5] == 0).sum() >= G // 2
    # the batch entry point on the same grouped rows gives the same records (it is the same path): bit for bit for the
    # wide kernels; for p <= 8 the batch of 2 G slots (the emptied sources included) may pick another accumulate kernel
    # than the batch of G groups (packed small groups / a wave per group), and the HC sums of a group are completed by
    # whichever wave finishes last — equal to rounding there
    bcore, binf = ctx.fit_batch_host(offs, yg, xg, wg if model == "wls" else None, opts)
    if p > 8 and hc is None:
        assert np.array_equal(core[:G], bcore, equal_nan=True) and np.array_equal(inf[:G], binf, equal_nan=True)
    else:
        assert np.allclose(core[:G], bcore, rtol=1e-10, atol=1e-12, equal_nan=True)
        assert np.allclose(inf[:G], binf, rtol=1e-8, atol=1e-12, equal_nan=True)
    st.close()


def test_log_only_state_budget_and_bad_slots(pkg, ctx):
    o = pkg.RegressionOptions().batch_options("ols")
    p = 12
    rng = np.random.default_rng(1)
    st = pkg.AggState(ctx, p, o, retain_bytes=50 * (8 * (p + 1) + 5), retain_host_bytes=0)        # room for 50 rows: the log IS the state
    X = rng.standard_normal((200, p))
    with pytest.raises(pkg.AnofoxStatsError) as ei:
        st.update(np.zeros(200, dtype=np.uint32), X[:, 0], X, n_slots=1)
    assert ei.value.code == 7 and "budget" in str(ei.value)
    st.close()
    st = pkg.AggState(ctx, p, o)
    st.update(np.array([0] * 30 + [5], dtype=np.uint32), rng.standard_normal(31), rng.standard_normal((31, p)), n_slots=2)
    with pytest.raises(pkg.AnofoxStatsError) as ei:
        st.finalize()
    assert "slot index" in str(ei.value)
    st.close()
    st = pkg.AggState(ctx, p, o)                                 # slots announced, no row ever
    st.reserve(3)
    core, inf, _ = st.finalize(3)
    assert np.all(core[:, p + 5] == 100)
    st.close()


def test_radix_sort_against_stable_sort():
    """csrc/radix_sort.h — the hand-written sort under the ingest passes and the row log (rocPRIM until round 3) — against
    std::stable_sort: sizes around the wavefront / sub-tile / tile boundaries up to 8 Mi keys, constant / sorted / reversed /
    two-valued / skewed inputs, every end_bit class, pairs (the value is the original position: stability) and 64-bit keys,
    a guard element past the output and an untouched input (csrc/tools/radix_sort_check.hip, built with the library)."""
    import subprocess
    exe = os.path.join(ROOT, "anofox-statistics_amd", "csrc", "tools", "radix_sort_check")
    assert os.path.exists(exe), "csrc/tools/radix_sort_check is not built (python -c 'import __graft_entry__ as g; g.build()')"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "radix_sort_check: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
