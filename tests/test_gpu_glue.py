"""The DuckDB glue (duckdb_shim/fit_agg_hip.cpp) on a GPU: compiled against the stand-in of DuckDB's headers, linked with
the real library and driven as DuckDB drives an aggregate (tests/tools/glue_driver.hpp) — a parallel hash aggregate with
thread-local states and Combine, the naive window aggregator, and a segment tree's Combine — against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from conftest import ROOT, assert_records_match

pytestmark = pytest.mark.gpu
LIB = os.path.join(ROOT, "anofox-statistics_amd", "duckdb_shim", "libanofox_glue_capi.so")
_DP = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def lib():
    lib = C.CDLL(LIB)
    lib.glue_open.restype = C.c_void_p
    lib.glue_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p]
    lib.glue_close.argtypes = [C.c_void_p]
    lib.glue_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.glue_group_by.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_char_p]
    lib.glue_group_by_ragged.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
    lib.glue_window.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
    lib.glue_tree_window.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                     C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
    return lib


def _stats(lib, q):
    s = (C.c_int64 * 6)()
    lib.glue_stats(q, s)
    return dict(rows=s[0], unrefined=s[1], slots=s[2], live=s[3], fit_calls=s[4], slots_fitted=s[5])


def _ptr(a):
    return None if a is None else a.ctypes.data


def _fix_last_column(core, p):
    """The SQL struct carries n_features where the library's record carries the status: a row that is not NULL has status 0."""
    out = core.copy()
    out[:, p + 5] = 0.0
    return out


@pytest.mark.parametrize("fn,model,p,spec,kw", [
    ("anofox_stats_ols_fit_agg", "ols", 3, None, {}),
    ("ols_fit_agg", "ols", 8, b"compute_inference=true;confidence_level=0.9", dict(compute_inference=True, confidence_level=0.9)),
    ("anofox_stats_ridge_fit_agg", "ridge", 5, b"alpha=0.5;inference=true", dict(alpha=0.5, compute_inference=True)),
    ("wls_fit_agg", "wls", 4, b"intercept=false;compute_inference=true", dict(fit_intercept=False, compute_inference=True)),
    ("ols_fit_agg", "ols", 20, b"compute_inference=true", dict(compute_inference=True)),            # log-only state
    ("anofox_stats_ols_fit_agg", "ols", 3, b"compute_inference=true;hc_type=hc1", dict(compute_inference=True, hc_type="hc1")),
])
def test_group_by_through_the_glue_matches_oracle(lib, fn, model, p, spec, kw):
    rng = np.random.default_rng(len(fn) * 31 + p)
    K, n = 400, 60_000
    key = rng.integers(0, K, n).astype(np.uint32)
    X = rng.uniform(-5, 5, (n, p)) + 1.0
    beta = rng.uniform(-3, 3, (K, p))
    y = np.einsum("ij,ij->i", beta[key], X) + 4.0 + rng.standard_normal(n)
    w = rng.uniform(0.5, 1.5, n)
    y_null = (rng.random(n) < 0.03).astype(np.uint8)        # NULL y / NULL x list / NULL weight: the row is skipped
    x_null = (rng.random(n) < 0.02).astype(np.uint8)
    w_null = (rng.random(n) < 0.02).astype(np.uint8)
    xe_null = (rng.random((n, p)) < 0.002).astype(np.uint8)  # NULL list elements: NaN, the fit drops the row
    key[:7] = K - 1                                          # a key with few rows
    msg = C.create_string_buffer(512)
    q = lib.glue_open(fn.encode(), spec, 0, msg)
    assert q, msg.value
    inference = kw.get("compute_inference", False)
    core = np.full((K, p + 6), np.nan)
    inf = np.full((K, 5 * p + 2), np.nan) if inference else None
    nn = np.zeros(K, dtype=np.uint8)
    rc = lib.glue_group_by(q, n, p, _ptr(key), K, _ptr(y), _ptr(X), _ptr(w), _ptr(y_null), _ptr(x_null), _ptr(xe_null), _ptr(w_null),
                           6, 2048, 1, _ptr(core), _ptr(inf), _ptr(nn), msg)
    assert rc == 0, msg.value
    st = _stats(lib, q)
    lib.glue_close(q)
    keep = ~(y_null.astype(bool) | x_null.astype(bool) | ((model == "wls") & w_null.astype(bool)))
    Xn = np.where(xe_null.astype(bool), np.nan, X)
    idx = np.nonzero(keep)[0]
    order = idx[np.argsort(key[idx], kind="stable")]
    offs = np.concatenate([[0], np.cumsum(np.bincount(key[idx], minlength=K))]).astype(np.int64)
    rcore, rinf = oracle.fit_groups(y[order], [np.ascontiguousarray(Xn[order, j]) for j in range(p)], offs,
                                    w=(w[order] if model == "wls" else None), model=model, **kw)
    assert st["rows"] == int(keep.sum()) and st["live"] == 0 and st["unrefined"] == 0
    assert st["fit_calls"] == 1                                   # ONE batched fit for the whole GROUP BY
    fitted = rcore[:, p + 5] == 0
    assert np.array_equal(nn == 0, fitted)                        # NULL exactly where the reference returns NULL
    assert np.all(core[fitted, p + 5] == p)                       # n_features
    assert_records_match(_fix_last_column(core[fitted], p), rcore[fitted], p, None if inf is None else inf[fitted],
                         None if rinf is None else rinf[fitted], what=f"glue GROUP BY {fn} p={p}")


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0"])
@pytest.mark.parametrize("fn,model,p,spec,kw", [
    ("ols_fit_agg", "ols", 8, b"compute_inference=true", dict(compute_inference=True)),
    ("wls_fit_agg", "wls", 4, b"intercept=false", dict(fit_intercept=False)),
    ("ridge_fit_agg", "ridge", 3, b"alpha=0.5", dict(alpha=0.5)),
    ("ols_fit_agg", "ols", 20, None, {}),                       # log-only state: stays on one shard, same answers
])
def test_group_by_sharded_over_devices_through_the_glue(lib, devices, fn, model, p, spec, kw):
    """SURVEY.md 8(e) behind the SQL aggregate: ANOFOX_HIP_DEVICES routes every aggregate state to one of W device states by
    hash64(state address) % W at its first accepted row; the parallel hash aggregate's Combine then pairs thread-local sources
    with targets on OTHER shards (moment records exported, imported and merged on the target's device), Finalize fits every
    shard in its own batched call.  W = 2 and 4 shards on the one GPU of this box: the same records as one shard (merge order
    may differ: 1e-12) and the oracle's; every row counted once; nothing flagged."""
    rng = np.random.default_rng(77 + p)
    K, n = 300, 50_000
    key = rng.integers(0, K, n).astype(np.uint32)
    X = rng.uniform(-5, 5, (n, p)) + 1.0
    beta = rng.uniform(-3, 3, (K, p))
    y = np.einsum("ij,ij->i", beta[key], X) + 4.0 + rng.standard_normal(n)
    w = rng.uniform(0.5, 1.5, n)
    zeros = np.zeros(n, dtype=np.uint8)
    xe = np.zeros((n, p), dtype=np.uint8)
    inference = kw.get("compute_inference", False)

    def run(dev):
        old = os.environ.pop("ANOFOX_HIP_DEVICES", None)
        if dev is not None:
            os.environ["ANOFOX_HIP_DEVICES"] = dev
        try:
            msg = C.create_string_buffer(512)
            q = lib.glue_open(fn.encode(), spec, 0, msg)
            assert q, msg.value
            core = np.full((K, p + 6), np.nan)
            inf = np.full((K, 5 * p + 2), np.nan) if inference else None
            nn = np.zeros(K, dtype=np.uint8)
            rc = lib.glue_group_by(q, n, p, _ptr(key), K, _ptr(y), _ptr(X), _ptr(w), _ptr(zeros), _ptr(zeros), _ptr(xe), _ptr(zeros),
                                   6, 2048, 1, _ptr(core), _ptr(inf), _ptr(nn), msg)
            assert rc == 0, msg.value
            st = _stats(lib, q)
            lib.glue_close(q)
            return core, inf, nn, st
        finally:
            os.environ.pop("ANOFOX_HIP_DEVICES", None)
            if old is not None:
                os.environ["ANOFOX_HIP_DEVICES"] = old

    core1, inf1, nn1, st1 = run(None)
    coreW, infW, nnW, stW = run(devices)
    W = len(devices.split(","))
    assert st1["rows"] == n and stW["rows"] == n and stW["live"] == 0 and stW["unrefined"] == 0
    assert st1["fit_calls"] == 1
    assert stW["fit_calls"] == (1 if p > 8 else W)        # one batched fit per device state that holds groups
    assert np.array_equal(nn1, nnW) and np.all(nnW == 0)
    np.testing.assert_allclose(coreW[:, :p + 4], core1[:, :p + 4], rtol=1e-10, atol=1e-12)
    order = np.argsort(key, kind="stable")
    offs = np.concatenate([[0], np.cumsum(np.bincount(key, minlength=K))]).astype(np.int64)
    rcore, rinf = oracle.fit_groups(y[order], [np.ascontiguousarray(X[order, j]) for j in range(p)], offs,
                                    w=(w[order] if model == "wls" else None), model=model, **kw)
    assert_records_match(_fix_last_column(coreW, p), rcore, p, infW, rinf, what=f"glue GROUP BY over {W} shards, {fn} p={p}")


def test_groups_of_different_widths_in_one_query(lib):
    """The reference fixes the feature count per STATE, at the state's first accepted row (ols_aggregate.cpp:164-175): the
    groups of one query may have x lists of different lengths, each result carries its own n_features and LIST lengths; a
    row of another length inside a group is the reference's error.  Widths 3 (moment records), 11 (tiles solve over the
    row log) and 0 (empty lists: NULL) in one GROUP BY, against the oracle run per width."""
    rng = np.random.default_rng(404)
    K, n, P = 300, 45_000, 11
    width_of_key = np.where(np.arange(K) % 3 == 0, 11, 3).astype(np.uint32)
    width_of_key[K - 1] = 0
    key = rng.integers(0, K, n).astype(np.uint32)
    X = rng.uniform(-5, 5, (n, P)) + 1.0
    y = X[:, :3] @ np.array([1.5, -2.0, 0.5]) + 4.0 + rng.standard_normal(n)
    x_len = width_of_key[key].copy()
    msg = C.create_string_buffer(512)
    q = lib.glue_open(b"ols_fit_agg", b"compute_inference=true", 0, msg)
    assert q, msg.value
    core = np.full((K, P + 6), np.nan)
    inf = np.full((K, 5 * P + 2), np.nan)
    nn = np.zeros(K, dtype=np.uint8)
    rc = lib.glue_group_by_ragged(q, n, P, _ptr(x_len), _ptr(key), K, _ptr(y), _ptr(X), None, 4, 2048, _ptr(core), _ptr(inf), _ptr(nn), msg)
    assert rc == 0, msg.value
    st = _stats(lib, q)
    assert st["live"] == 0 and st["unrefined"] == 0 and st["rows"] == int(np.sum(x_len > 0)) and st["fit_calls"] == 2
    assert nn[K - 1] == 1 and np.all(nn[:K - 1] == 0)
    for width in (3, 11):
        ks = np.nonzero(width_of_key == width)[0]
        sel = np.nonzero(np.isin(key, ks))[0]
        order = sel[np.argsort(key[sel], kind="stable")]
        offs = np.concatenate([[0], np.cumsum(np.bincount(key[sel], minlength=K)[ks])]).astype(np.int64)
        rcore, rinf = oracle.fit_groups(y[order], [np.ascontiguousarray(X[order, j]) for j in range(width)], offs, compute_inference=True)
        got = np.concatenate([core[ks, :width], core[ks, P:P + 6]], axis=1)
        assert np.all(got[:, width + 5] == width)                              # n_features of each group
        assert np.all(np.isnan(core[ks, width:P]))                              # ... and its LIST has exactly that many entries
        got_inf = np.concatenate([inf[ks, l * P:l * P + width] for l in range(5)] + [inf[ks, 5 * P:]], axis=1)
        assert_records_match(_fix_last_column(got, width), rcore, width, got_inf, rinf, what=f"mixed widths, p={width}")
    # one row of another width inside a group: the reference's message, with the state's own count
    x_len2 = x_len.copy()
    victim = np.nonzero(key == 1)[0][5]
    x_len2[victim] = 5
    rc = lib.glue_group_by_ragged(q, n, P, _ptr(x_len2), _ptr(key), K, _ptr(y), _ptr(X), None, 1, 2048, _ptr(core), _ptr(inf), _ptr(nn), msg)
    assert rc != 0 and msg.value == b"Inconsistent feature count: expected 3, got 5"
    assert _stats(lib, q)["live"] == 0                                          # the failing query destroyed its states
    lib.glue_close(q)


def test_reference_window_test_through_the_glue(lib):
    """test/sql/comprehensive_tests.test:425-444: anofox_stats_ols_fit_agg(y, [x]) OVER (ORDER BY idx ROWS BETWEEN 4 PRECEDING
    AND CURRENT ROW) over y = 2 i + 1, i = 1..20 — 16 rows have n_observations = 5.  The frames fit exactly (rss = 0): the
    device state queues every one of them for refinement and refits them from its row log."""
    i = np.arange(1, 21, dtype=np.float64)
    y, X = 2 * i + 1, i.reshape(-1, 1).copy()
    for vsize in (2048, 4):
        msg = C.create_string_buffer(512)
        q = lib.glue_open(b"anofox_stats_ols_fit_agg", None, 0, msg)
        core = np.full((20, 7), np.nan)
        nn = np.zeros(20, dtype=np.uint8)
        assert lib.glue_window(q, 20, 1, _ptr(y), _ptr(X), None, 4, vsize, _ptr(core), None, _ptr(nn), msg) == 0, msg.value
        st = _stats(lib, q)
        lib.glue_close(q)
        assert int(np.sum((nn == 0) & (core[:, 5] == 5))) == 16
        assert nn[0] == 1 and np.all(nn[1:] == 0)                 # a single row -> NULL (ols_aggregate.cpp:263-267)
        assert np.allclose(core[1:, 0], 2.0, rtol=1e-9) and np.allclose(core[1:, 1], 1.0, rtol=0, atol=1e-8) and np.allclose(core[1:, 2], 1.0)
        assert st["live"] == 0 and st["slots_fitted"] == 20 and st["unrefined"] == 0
        if vsize == 4:
            assert st["slots"] <= 8                               # destroyed states' slots are handed out again


@pytest.mark.parametrize("p,model", [(2, "ols"), (6, "wls")])
def test_rolling_window_through_the_glue_matches_oracle(lib, p, model):
    rng = np.random.default_rng(7 + p)
    n, back = 3000, 39
    X = rng.uniform(-5, 5, (n, p)) + np.linspace(0, 3, n)[:, None]
    y = X @ rng.uniform(-2, 2, p) + 1.5 + rng.standard_normal(n)
    w = rng.uniform(0.5, 1.5, n)
    msg = C.create_string_buffer(512)
    fn = b"anofox_stats_wls_fit_agg" if model == "wls" else b"ols_fit_agg"
    q = lib.glue_open(fn, b"compute_inference=true", 0, msg)
    core = np.full((n, p + 6), np.nan)
    inf = np.full((n, 5 * p + 2), np.nan)
    nn = np.zeros(n, dtype=np.uint8)
    assert lib.glue_window(q, n, p, _ptr(y), _ptr(X), _ptr(w), back, 512, _ptr(core), _ptr(inf), _ptr(nn), msg) == 0, msg.value
    st = _stats(lib, q)
    lib.glue_close(q)
    # the oracle: every frame as a group of its own
    lo = np.maximum(0, np.arange(n) - back)
    lens = np.arange(n) + 1 - lo
    rows = np.concatenate([np.arange(a, b + 1) for a, b in zip(lo, np.arange(n))])
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rcore, rinf = oracle.fit_groups(y[rows], [np.ascontiguousarray(X[rows, j]) for j in range(p)], offs,
                                    w=(w[rows] if model == "wls" else None), model=model, compute_inference=True)
    fitted = rcore[:, p + 5] == 0
    assert np.array_equal(nn == 0, fitted)
    zero_df = [k for k, g in enumerate(np.nonzero(fitted)[0]) if lens[g] <= p + 1]
    assert_records_match(_fix_last_column(core[fitted], p), rcore[fitted], p, inf[fitted], rinf[fitted],
                         what=f"glue window {model} p={p}", skip_diag_groups=zero_df)
    assert st["live"] == 0 and st["slots"] <= 2 * 512 and st["slots_fitted"] == n     # bounded state, each frame fitted once


def test_segment_tree_combine_through_the_glue(lib):
    """PRESERVE_INPUT Combine: every leaf state is the source of up to three frames of one Combine call and lives on."""
    rng = np.random.default_rng(11)
    p, leaf, back, n = 3, 50, 2, 50 * 40
    X = rng.uniform(-5, 5, (n, p))
    y = X @ np.array([1.0, -2.0, 0.5]) + 3.0 + rng.standard_normal(n)
    msg = C.create_string_buffer(512)
    q = lib.glue_open(b"ols_fit_agg", None, 0, msg)
    L = n // leaf
    core = np.full((L, p + 6), np.nan)
    nn = np.zeros(L, dtype=np.uint8)
    assert lib.glue_tree_window(q, n, p, _ptr(y), _ptr(X), None, leaf, back, 2048, _ptr(core), None, _ptr(nn), msg) == 0, msg.value
    lib.glue_close(q)
    los = np.maximum(0, np.arange(L) - back) * leaf
    his = (np.arange(L) + 1) * leaf
    rows = np.concatenate([np.arange(a, b) for a, b in zip(los, his)])
    offs = np.concatenate([[0], np.cumsum(his - los)]).astype(np.int64)
    rcore, _ = oracle.fit_groups(y[rows], [np.ascontiguousarray(X[rows, j]) for j in range(p)], offs, model="ols")
    assert np.all(nn == 0)
    assert_records_match(_fix_last_column(core, p), rcore, p, None, None, what="glue segment tree")


@pytest.mark.parametrize("p,exact", [(3, True), (12, False), (12, True)])
def test_segment_tree_combine_keeps_frames_refinable(lib, p, exact):
    """(r4) A preserved Combine copies the sources' logged rows to their targets, as the reference's Combine copies
    the row buffers: frames that FIT EXACTLY (every such frame is queued for the refinement passes) come back as fits, not
    NULL, and designs of more than 8 features — whose state is the rows themselves — go through the segment tree as well."""
    rng = np.random.default_rng(29 + p)
    leaf, back = 40, 2
    n = leaf * 24
    X = rng.uniform(-5, 5, (n, p))
    y = X @ rng.uniform(-2, 2, p) + 3.0 + (0.0 if exact else 1.0) * rng.standard_normal(n)
    msg = C.create_string_buffer(512)
    q = lib.glue_open(b"ols_fit_agg", None, 0, msg)
    L = n // leaf
    core = np.full((L, p + 6), np.nan)
    nn = np.zeros(L, dtype=np.uint8)
    assert lib.glue_tree_window(q, n, p, _ptr(y), _ptr(X), None, leaf, back, 2048, _ptr(core), None, _ptr(nn), msg) == 0, msg.value
    st = _stats(lib, q)
    lib.glue_close(q)
    assert np.all(nn == 0) and st["unrefined"] == 0                       # no frame NULL, none flagged
    los = np.maximum(0, np.arange(L) - back) * leaf
    his = (np.arange(L) + 1) * leaf
    rows = np.concatenate([np.arange(a, b) for a, b in zip(los, his)])
    offs = np.concatenate([[0], np.cumsum(his - los)]).astype(np.int64)
    rcore, _ = oracle.fit_groups(y[rows], [np.ascontiguousarray(X[rows, j]) for j in range(p)], offs, model="ols")
    if exact:       # sigma of an exact fit is rounding noise on either side: coefficients, r^2 and the row counts are compared
        scale = np.max(np.abs(rcore[:, :p + 1]), axis=1, keepdims=True)
        assert np.max(np.abs(core[:, :p + 1] - rcore[:, :p + 1]) / np.maximum(np.abs(rcore[:, :p + 1]), 1e-3 * scale)) < 1e-9
        assert np.max(np.abs(core[:, p + 1] - 1.0)) < 1e-12 and np.all(core[:, p + 3] < 1e-8)
        assert np.array_equal(core[:, p + 4], rcore[:, p + 4])
    else:
        assert_records_match(_fix_last_column(core, p), rcore, p, None, None, what=f"glue segment tree p={p}")


# ------------------------------------------------------------------------------------------------------------------------
# the rest of the family through its DuckDB glue (duckdb_shim/family_agg_hip.cpp): *_fit_predict_agg, the *_fit_predict
# window aggregates and vif_agg — Update / Combine / Finalize as DuckDB drives them, one batched library call per Finalize
# vector, against the oracle (the reference's files: src/aggregate_functions/*_predict_aggregate.cpp,
# src/window_functions/*_fit_predict.cpp, src/aggregate_functions/vif_aggregate.cpp)
# ------------------------------------------------------------------------------------------------------------------------
SPLIT_STRINGS = [None, "train", "Training", "test", "TRAIN", "a-validation-partition-name", "training"]   # family_driver.hpp


@pytest.fixture(scope="module")
def fam(lib):
    lib.family_open.restype = C.c_void_p
    lib.family_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p]
    lib.family_close.argtypes = [C.c_void_p]
    lib.family_result_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.family_registered.argtypes = [C.c_void_p, C.c_char_p]
    lib.family_predict_group_by.restype = C.c_int64
    lib.family_predict_group_by.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t] + [C.c_void_p] * 8 + [
        C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
    lib.family_vif_group_by.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p]
    lib.family_window.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t] + [C.c_void_p] * 7 + [C.c_size_t] * 4 + [C.c_void_p, C.c_void_p, C.c_char_p]
    return lib


def _driver_order(key, keep, n_threads, vector_size):
    """Row order of every group's output: the driver hands vector v to thread v % n_threads and combines thread by thread."""
    idx = np.nonzero(keep)[0]
    thread = (idx // vector_size) % n_threads
    return idx[np.lexsort((idx, thread, key[idx]))]


def _pred_close(got, want, what):
    assert np.array_equal(np.isnan(got), np.isnan(want)), what
    m = ~np.isnan(want)
    if m.any():
        scale = np.maximum(np.abs(want[m]), 1e-3 * np.abs(want[m]).max())
        assert np.max(np.abs(got[m] - want[m]) / scale) < 1e-9, what


@pytest.mark.parametrize("fn,model,p,split,spec,kw", [
    ("anofox_stats_ols_fit_predict_agg", "ols", 3, False, None, {}),
    ("ols_predict_agg", "ols", 5, True, b"confidence_level=0.9;intercept=false", dict(confidence_level=0.9, fit_intercept=False)),
    ("ridge_fit_predict_agg", "ridge", 4, False, b"alpha=0.7;null_policy=drop_y_zero_x", dict(alpha=0.7)),
    ("anofox_stats_wls_fit_predict_agg", "wls", 2, True, None, {}),
    ("wls_predict_agg", "wls", 6, False, b"null_policy=drop_y_zero_x;fit_intercept=true", {}),
    ("ols_fit_predict_agg", "ols", 20, False, None, {}),                 # a width of the MFMA paths
])
def test_fit_predict_agg_through_the_glue_matches_oracle(fam, fn, model, p, split, spec, kw):
    rng = np.random.default_rng(len(fn) * 17 + p)
    K, n, threads, vsize = 150, 24_000, 5, 512
    key = rng.integers(0, K, n).astype(np.uint32)
    X = rng.uniform(-5, 5, (n, p)) + 1.0
    X[rng.random((n, p)) < 0.01] = 0.0                                  # exact zeros: null_policy = 'drop_y_zero_x'
    beta = rng.uniform(-3, 3, (K, p))
    y = np.einsum("ij,ij->i", beta[key], X) + 4.0 + rng.standard_normal(n)
    w = rng.uniform(0.5, 1.5, n)
    w[rng.random(n) < 0.02] = 0.0                                       # weight <= 0: kept for the output, does not train
    w[rng.random(n) < 0.01] = -1.0
    y_null = (rng.random(n) < 0.2).astype(np.uint8)                     # prediction rows
    x_null = (rng.random(n) < 0.02).astype(np.uint8)                    # NULL x list: the row does not exist
    w_null = (rng.random(n) < 0.02).astype(np.uint8)
    xe_null = (rng.random((n, p)) < 0.004).astype(np.uint8)             # NULL list elements
    code = rng.integers(0, len(SPLIT_STRINGS), n).astype(np.uint8)
    key[:3] = K - 1                                                     # one key with (almost) nothing to train on
    key[3:] = np.minimum(key[3:], K - 2)
    y_null[:3] = [0, 1, 1]
    msg = C.create_string_buffer(512)
    q = fam.family_open(fn.encode(), spec, 0, int(split), msg)
    assert q, msg.value
    offs = np.zeros(K + 1, dtype=np.int64)
    vals = np.full((n, 4), np.nan)
    flags = np.zeros(n, dtype=np.uint8)
    nn = np.zeros(K, dtype=np.uint8)
    rows = fam.family_predict_group_by(q, n, p, _ptr(key), K, _ptr(y), _ptr(X), _ptr(w), _ptr(y_null), _ptr(x_null), _ptr(xe_null), _ptr(w_null),
                                       _ptr(code) if split else None, threads, vsize, 1, _ptr(offs), _ptr(vals), _ptr(flags), _ptr(nn), msg)
    fam.family_close(q)
    assert rows >= 0, msg.value
    weighted = model == "wls"
    keep = ~x_null.astype(bool) & ~(weighted & w_null.astype(bool))
    order = _driver_order(key, keep, threads, vsize)
    Xn = np.where(xe_null.astype(bool), np.nan, X)
    train = ~y_null.astype(bool)
    if split:
        train &= np.array([s is not None and s.lower() in ("train", "training") for s in SPLIT_STRINGS])[code]
    if weighted:
        train &= w > 0
    if model == "ols":
        train &= ~xe_null.any(axis=1)                                  # only the OLS file clears the flag on a NULL feature
    if spec and b"drop_y_zero_x" in spec:
        train &= ~np.any(Xn == 0.0, axis=1)
    ks, yo, Xo, wo, tro = key[order], y[order], Xn[order], w[order], train[order]
    roffs = np.concatenate([[0], np.cumsum(np.bincount(ks, minlength=K))]).astype(np.int64)
    counts = np.bincount(ks, weights=tro, minlength=K).astype(np.int64)
    rcore, rpred = oracle.fit_predict_groups(np.where(tro, yo, np.nan), [np.ascontiguousarray(Xo[:, j]) for j in range(p)], roffs,
                                             w=(np.where(tro, wo, 1.0) if weighted else None), train_counts=counts, model=model, **kw)
    null_ref = (counts < 2) | (rcore[:, p + 5] != 0)
    assert np.array_equal(nn.astype(bool), null_ref)                    # NULL exactly where the reference returns NULL
    assert nn[K - 1] == 1
    fitted = ~null_ref
    assert np.array_equal(np.diff(offs)[fitted], np.diff(roffs)[fitted]) and np.all(np.diff(offs)[~fitted] == 0)
    sel = np.concatenate([np.arange(roffs[k], roffs[k + 1]) for k in np.nonzero(fitted)[0]])
    assert rows == len(sel)
    got, fl = vals[:rows], flags[:rows]
    assert np.array_equal((fl & 1) != 0, y_null[order][sel] != 0)                      # y: NULL where it was NULL ...
    assert np.array_equal(got[(fl & 1) == 0, 0], yo[sel][(fl & 1) == 0])               # ... and otherwise the value, bit for bit
    assert np.array_equal((fl & 16) != 0, tro[sel])                                    # is_training
    assert np.all(((fl & 14) == 0) | ((fl & 14) == 14))                                # the three prediction fields are NULL together
    _pred_close(got[:, 1:4], rpred[sel], f"{fn} p={p}")


@pytest.mark.parametrize("fn,model,p,spec,kw", [
    ("anofox_stats_ols_fit_predict", "ols", 2, None, {}),
    ("ridge_fit_predict", "ridge", 3, b"alpha=0.5;intercept=false", dict(alpha=0.5, fit_intercept=False)),
    ("wls_fit_predict", "wls", 4, b"confidence_level=0.8", dict(confidence_level=0.8)),
    ("ols_fit_predict", "ols", 12, None, {}),
])
def test_fit_predict_window_through_the_glue_matches_oracle(fam, fn, model, p, spec, kw):
    """The window aggregates as DuckDB's window operator drives them: a state per output row fed its frame (naive aggregator),
    and leaf states combined under PRESERVE_INPUT into a fresh state per frame (segment tree).  Every Finalize vector is one
    batched call; the oracle refits every frame."""
    rng = np.random.default_rng(len(fn) + 5 * p)
    n, preceding = 700, 3 * p + 14
    X = rng.uniform(-3, 3, (n, p)) + 0.5
    y = X @ rng.uniform(-2, 2, p) + 1.5 + 0.3 * rng.standard_normal(n)
    w = rng.uniform(0.5, 2.0, n)
    y_null = (rng.random(n) < 0.1).astype(np.uint8)
    x_null = (rng.random(n) < 0.03).astype(np.uint8)
    w_null = (rng.random(n) < 0.03).astype(np.uint8)
    weighted = model == "wls"
    msg = C.create_string_buffer(512)
    q = fam.family_open(fn.encode(), spec, 0, 0, msg)
    assert q, msg.value
    out = np.full((n, 3), np.nan)
    nn = np.zeros(n, dtype=np.uint8)
    rc = fam.family_window(q, n, p, _ptr(y), _ptr(X), _ptr(w), _ptr(y_null), _ptr(x_null), None, _ptr(w_null), preceding, 0, 0, 256,
                           _ptr(out), _ptr(nn), msg)
    assert rc == 0, msg.value
    # the oracle: a group per frame = its training rows, then the row to predict (y NaN)
    icpt = kw.get("fit_intercept", True)
    trains = ~y_null.astype(bool) & ~x_null.astype(bool) & ~(weighted & w_null.astype(bool))

    def reference(frames):
        """frames: per output row (training rows in arrival order, the row whose x is predicted or None)"""
        ys, xs, ws, offs, want_null, cnt = [], [], [], [0], [], []
        for tr, cur in frames:
            null = cur is None or len(tr) <= p + int(icpt)
            want_null.append(null)
            if null:
                continue
            ys.append(np.append(y[tr], np.nan))
            xs.append(np.vstack([X[tr], X[cur]]))
            ws.append(np.append(w[tr], 1.0))
            offs.append(offs[-1] + len(tr) + 1)
            cnt.append(len(tr))
        yy, xx, ww = np.concatenate(ys), np.vstack(xs), np.concatenate(ws)
        rcore, rpred = oracle.fit_predict_groups(yy, [np.ascontiguousarray(xx[:, j]) for j in range(p)], np.array(offs, dtype=np.int64),
                                                 w=(ww if weighted else None), train_counts=np.array(cnt, dtype=np.int64), model=model, **kw)
        wn = np.array(want_null)
        want = np.full((len(frames), 3), np.nan)
        want[~wn] = rpred[np.array(offs[1:]) - 1]
        wn[np.nonzero(~wn)[0][rcore[:, p + 5] != 0]] = True
        return wn, want

    def naive_frame(o):                      # has_current_x follows the LAST row Update saw (ols_fit_predict.cpp:141-166)
        fr = list(range(max(0, o - preceding), o + 1))
        return [r for r in fr if trains[r]], (None if x_null[fr[-1]] else fr[-1])

    wn, want = reference([naive_frame(o) for o in range(n)])
    assert np.array_equal(nn.astype(bool), wn)
    _pred_close(out[~wn], want[~wn], f"{fn} naive window")
    leaf, back = 8, 5
    n_leaves = (n + leaf - 1) // leaf
    out_t = np.full((n_leaves, 3), np.nan)
    nn_t = np.zeros(n_leaves, dtype=np.uint8)
    rc = fam.family_window(q, n, p, _ptr(y), _ptr(X), _ptr(w), _ptr(y_null), _ptr(x_null), None, _ptr(w_null), 0, leaf, back, 64,
                           _ptr(out_t), _ptr(nn_t), msg)
    fam.family_close(q)
    assert rc == 0, msg.value

    def tree_frame(o):
        """Combine (ols_fit_predict.cpp:196-243): a leaf that never saw a non-NULL x list is skipped; the target keeps its row to
        predict unless the source has one."""
        tr, cur = [], None
        for l in range(max(0, o - back), o + 1):
            lr = list(range(l * leaf, min(n, (l + 1) * leaf)))
            if all(x_null[r] for r in lr):
                continue
            tr += [r for r in lr if trains[r]]
            if not x_null[lr[-1]]:
                cur = lr[-1]
        return tr, cur
    wn, want = reference([tree_frame(o) for o in range(n_leaves)])
    assert np.array_equal(nn_t.astype(bool), wn)
    _pred_close(out_t[~wn], want[~wn], f"{fn} segment tree")


@pytest.mark.parametrize("p,threads,vsize", [(4, 4, 256), (2, 1, 2048), (12, 3, 100)])
def test_vif_agg_through_the_glue_matches_oracle(fam, p, threads, vsize):
    rng = np.random.default_rng(3 + p)
    K, n = 60, 9000
    key = rng.integers(0, K, n).astype(np.uint32)
    X = rng.standard_normal((n, p))
    X[:, 1] += 0.8 * X[:, 0]                                            # some collinearity
    x_null = (rng.random(n) < 0.03).astype(np.uint8)
    key[:2] = K - 1                                                     # fewer than 3 rows -> NULL
    key[2:] = np.minimum(key[2:], K - 2)
    ragged_key = 7
    ragged_row = np.nonzero(key == ragged_key)[0][3]
    X[ragged_row, 1] = np.nan                                           # one NaN shortens that column only -> NULL (vif.rs:40-51)
    x_null[ragged_row] = 0
    all_nan_rows = np.nonzero(key == 9)[0][:2]
    X[all_nan_rows] = np.nan                                            # a row of NaNs shortens every column alike: the row is gone
    msg = C.create_string_buffer(512)
    q = fam.family_open(b"vif_agg", None, 0, 0, msg)
    assert q, msg.value
    shape = C.c_int(-1)
    assert fam.family_result_shape(q, C.byref(shape)) == 2 and fam.family_registered(q, b"anofox_stats_vif_agg") == 1
    out = np.full((K, p), np.nan)
    nn = np.zeros(K, dtype=np.uint8)
    rc = fam.family_vif_group_by(q, n, p, _ptr(key), K, _ptr(X), _ptr(x_null), None, threads, vsize, 1, _ptr(out), _ptr(nn), msg)
    fam.family_close(q)
    assert rc == 0, msg.value
    keep = ~x_null.astype(bool) & ~np.isnan(X).all(axis=1)
    order = _driver_order(key, keep, threads, vsize)
    roffs = np.concatenate([[0], np.cumsum(np.bincount(key[order], minlength=K))]).astype(np.int64)
    rv = oracle.vif_groups([np.ascontiguousarray(X[order, j]) for j in range(p)], roffs)
    null_ref = rv[:, p] != 0
    null_ref[ragged_key] = True
    assert np.array_equal(nn.astype(bool), null_ref) and nn[K - 1] == 1
    ok = ~null_ref
    g, r = out[ok], rv[ok, :p]
    assert np.array_equal(np.isinf(g), np.isinf(r))
    m = np.isfinite(r)
    assert np.max(np.abs(g[m] - r[m]) / np.abs(r[m])) < 1e-6


def test_family_registration_and_errors_through_the_glue(fam):
    msg = C.create_string_buffer(512)
    q = fam.family_open(b"anofox_stats_ols_fit_predict_agg", None, 0, 0, msg)
    assert q, msg.value
    fields = C.c_int(0)
    assert fam.family_result_shape(q, C.byref(fields)) == 0 and fields.value == 5    # LIST(STRUCT(y, yhat, yhat_lower, yhat_upper, is_training))
    for name in ("ols_fit_predict_agg", "ols_predict_agg", "anofox_stats_ols_predict_agg", "anofox_stats_ridge_fit_predict_agg",
                 "wls_predict_agg", "anofox_stats_ols_fit_predict", "ridge_fit_predict", "anofox_stats_wls_fit_predict", "vif_agg"):
        assert fam.family_registered(q, name.encode()) == 1, name
    fam.family_close(q)
    q = fam.family_open(b"wls_fit_predict", b"fit_intercept=false", 1, 0, msg)      # a MAP literal
    assert q, msg.value
    assert fam.family_result_shape(q, C.byref(fields)) == 1 and fields.value == 3    # STRUCT(yhat, yhat_lower, yhat_upper)
    fam.family_close(q)
    assert not fam.family_open(b"ols_fit_predict_agg", b"null_policy=keep", 0, 0, msg)
    assert b"Invalid null_policy: 'keep'. Valid values are 'drop', 'drop_y_zero_x'" in msg.value
    assert not fam.family_open(b"ols_fit_predict", None, 0, 1, msg)                  # the window aggregates take no split column
