"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of
libanofox_stats_hip.so and is compared with the CPU oracle on identical inputs, with the reference's
R fixtures / known answers, and — at the BASELINE sizes — through size-independent properties.
Tolerances (north_star): coefficients 1e-9 relative, diagnostics 1e-6 (conftest.assert_records_match)."""
import numpy as np
import pytest

import oracle
from conftest import (COEF_RTOL, DIAG_RTOL, assert_records_match, import_pkg, load_csv, load_json, nan_or, rel_err,
                      release_device_memory)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return import_pkg()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context()
    yield c
    c.close()


def _opts(pkg, model, **kw):
    o = pkg.RegressionOptions(**kw)
    return o.batch_options(model)


def _oracle_kw(model, kw):
    d = dict(model=model)
    d.update(kw)
    d.pop("solver", None)
    return d


def _host_fit(pkg, ctx, model, offsets, y, x_cols, w=None, **kw):
    return pkg.fit_batch_host(offsets, y, x_cols, w, _opts(pkg, model, **kw), ctx=ctx)


# --------------------------------------------------------------------------------------------------
# reference fixtures (R lm / glmnet), through the aggregate mirror
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,xn,icpt", [
    ("simple_linear", ["x"], True),
    ("multiple_regression", ["x1", "x2", "x3"], True),
    ("no_intercept", ["x"], False),
    ("rank_deficient", ["x1", "x2"], True),
    ("perfect_collinearity", ["x1", "x2"], True),
])
def test_ols_fixtures(pkg, ctx, case, xn, icpt):
    d = load_csv(f"ols_tests/input/{case}.csv")
    e = load_json(f"ols_tests/expected/{case}.json")
    X = np.stack([d[n] for n in xn], axis=1)
    res = pkg.ols_fit_agg(np.zeros(len(d["y"]), dtype=np.int64), d["y"], X, {"intercept": icpt}, context=ctx)
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
    assert r["n_observations"] == len(d["y"]) and r["n_features"] == len(xn)


@pytest.mark.parametrize("case", ["wls_equal_weights", "wls_inverse_variance"])
def test_wls_fixtures(pkg, ctx, case):
    d = load_csv(f"wls_tests/input/{case}.csv")
    e = load_json(f"wls_tests/expected/{case}.json")
    res = pkg.wls_fit_agg(np.zeros(len(d["y"]), dtype=np.int64), d["y"], d["x"][:, None], d["weight"], context=ctx)
    r = res.row(0)
    assert rel_err(r["intercept"], e["coefficients"][0]) < COEF_RTOL
    assert rel_err(r["coefficients"][0], e["coefficients"][1]) < COEF_RTOL
    assert rel_err(r["r_squared"], e["r_squared"]) < DIAG_RTOL
    assert rel_err(r["adj_r_squared"], e["adj_r_squared"]) < DIAG_RTOL
    assert rel_err(r["residual_std_error"], e["sigma"]) < DIAG_RTOL


@pytest.mark.parametrize("case,xn", [("simple_inference", ["x"]), ("multiple_inference", ["x1", "x2", "x3"])])
def test_inference_fixtures(pkg, ctx, case, xn):
    d = load_csv(f"inference_tests/input/{case}.csv")
    e = load_json(f"inference_tests/expected/{case}.json")
    X = np.stack([d[n] for n in xn], axis=1)
    res = pkg.ols_fit_agg(np.zeros(len(d["y"]), dtype=np.int64), d["y"], X,
                          {"compute_inference": True, "confidence_level": 0.95}, context=ctx)
    r = res.row(0)
    c = e["coefficients"]
    assert rel_err(r["intercept"], c["estimates"][0]) < COEF_RTOL
    assert np.all(rel_err(r["coefficients"], c["estimates"][1:]) < COEF_RTOL)
    assert np.all(rel_err(r["std_errors"], c["std_errors"][1:]) < DIAG_RTOL)
    assert np.all(rel_err(r["t_values"], c["t_values"][1:]) < DIAG_RTOL)
    assert np.all(rel_err(r["p_values"], c["p_values"][1:]) < DIAG_RTOL)     # down to 1.28e-85
    assert np.all(rel_err(r["ci_lower"], e["confidence_intervals"]["lower_95"][1:]) < DIAG_RTOL)
    assert np.all(rel_err(r["ci_upper"], e["confidence_intervals"]["upper_95"][1:]) < DIAG_RTOL)
    assert rel_err(r["f_statistic"], e["model_stats"]["fstatistic"][0]) < DIAG_RTOL
    assert rel_err(r["r_squared"], e["model_stats"]["r_squared"]) < DIAG_RTOL


@pytest.mark.parametrize("case", ["ridge_lambda_0.1", "ridge_lambda_1.0"])
def test_ridge_glmnet_fixtures(pkg, ctx, case):
    d = load_csv(f"ridge_tests/input/{case}.csv")
    e = load_json(f"ridge_tests/expected/{case}.json")
    X = np.stack([d[n] for n in ("x1", "x2", "x3")], axis=1)
    res = pkg.ridge_fit_agg(np.zeros(len(d["y"]), dtype=np.int64), d["y"], X,
                            {"alpha": e["lambda"], "lambda_scaling": "glmnet"}, context=ctx)
    r = res.row(0)
    got = np.array([r["intercept"]] + r["coefficients"])
    assert np.all(rel_err(got, e["coefficients"]) < 2e-5)   # glmnet's own convergence, SURVEY.md §8c-(i)


def test_ridge_raw_identity(pkg, known_answers):
    k = known_answers["ridge_raw_identity"]
    x = [-2.0, -1.0, 0.0, 1.0, 2.0]
    y = (k["ols_slope"] * np.array(x) + np.array([0.3, -0.6, 0.6, -0.6, 0.3])).tolist()
    r = pkg.ridge_fit(y, [x], {"alpha": k["lambda"]})
    assert rel_err(r["coefficients"][0], k["ridge_slope"]) < 1e-12


# --------------------------------------------------------------------------------------------------
# the reference's sqllogictest known answers: scalar functions through the reference-compatible symbols
# --------------------------------------------------------------------------------------------------
def _check_expect(r, exp):
    for key, val in exp.items():
        if key == "coefficients":
            for got, w in zip(r["coefficients"], val):
                if w is not None:
                    assert round(float(got), w[1]) + 0.0 == w[0]
        elif key == "coefficients_abs":
            for got, w in zip(r["coefficients"], val):
                assert abs(got - w[0]) < w[1]
        elif key == "intercept_abs":
            assert abs(r["intercept"] - val[0]) < val[1]
        elif key == "intercept_is_nan":
            assert np.isnan(r["intercept"])
        elif key == "residual_std_error_lt":
            assert r["residual_std_error"] < val
        elif key == "r_squared_gt":
            assert r["r_squared"] > val
        else:
            assert round(float(r[key]), val[1]) + 0.0 == val[0]


def test_scalar_known_answers(pkg, known_answers):
    for case in known_answers["scalar_fits"]:
        r = pkg.ols_fit(case["y"], case["x"], {"intercept": case["fit_intercept"]})
        _check_expect(r, case["expect"])
        assert r["n_features"] == len(case["x"]) and len(r["coefficients"]) == len(case["x"])


def test_group_by_known_answers(pkg, ctx, known_answers):
    for case in known_answers["group_by"]:
        rows = case["rows"]
        keys = [r[0] for r in rows]
        X = [r[1:-1] for r in rows]
        y = [r[-1] for r in rows]
        res = pkg.ols_fit_agg(keys, y, X, {"intercept": case["fit_intercept"]}, context=ctx).as_dict()
        for g, exp in case["expect"].items():
            if exp == "NULL":
                assert res[g] is None
            elif exp == "OK":
                assert res[g] is not None and not np.isnan(res[g]["r_squared"])
            else:
                _check_expect(res[g], exp)


def test_series_known_answers(pkg, known_answers):
    for case in known_answers["series"]:
        i = np.arange(1, case["n"] + 1, dtype=np.float64)
        fn = pkg.ols_fit if case["model"] == "ols" else pkg.ridge_fit
        opts = {"alpha": case["alpha"]} if "alpha" in case else None
        _check_expect(fn((2 * i + 1).tolist(), [i.tolist()], opts), case["expect"])


def test_cfg1_example_calls_through_the_c_symbols(pkg, known_answers):
    """BASELINE cfg1: the eight example calls of examples/ols_single_series.sql (:27-31,49-57,78-82,98-102,113-117,
    140-157,186,200-230) through anofox_ols_fit, against the closed form rounded as the example rounds."""
    from test_oracle_golden import _check_example
    cases = known_answers["ols_single_series_examples"]["cases"]
    assert len(cases) == 11
    for case in cases:
        r = pkg.ols_fit(case["y"], case["x"], case["options"])
        _check_example(r, case)
        assert r["n_features"] == len(case["x"]) and len(r["coefficients"]) == len(case["x"])
        if case["name"] == "ex3_full_inference":      # an exact fit: errors and p-values collapse to ~0
            assert r["std_errors"][0] < 1e-9 and r["p_values"][0] < 1e-12 and r["f_pvalue"] < 1e-12


def test_single_fit_error_conventions(pkg):
    a = import_pkg("_abi")
    with pytest.raises(pkg.InvalidInputException) as ei:
        pkg.ridge_fit([1.0, 2.0, 3.0], [[1.0, 2.0, 4.0]], {"alpha": -1.0})
    assert ei.value.code == a.ERROR_INVALID_ALPHA and "Invalid alpha parameter" in str(ei.value)
    with pytest.raises(pkg.InvalidInputException) as ei:
        pkg.ols_fit([1.0, 2.0], [[1.0, 2.0], [2.0, 5.0]])
    assert ei.value.code == a.ERROR_INSUFFICIENT_DATA and "Insufficient data: 2 rows, 2 features" in str(ei.value)
    with pytest.raises(pkg.InvalidInputException) as ei:
        pkg.ols_fit([None, None], [[1.0, 2.0]])
    assert ei.value.code == a.ERROR_NO_VALID_DATA
    with pytest.raises(pkg.InvalidInputException) as ei:
        pkg.ols_fit([1.0, 2.0, 3.0], [[1.0, 2.0]])
    assert ei.value.code == a.ERROR_DIMENSION_MISMATCH
    with pytest.raises(pkg.InvalidInputException) as ei:
        pkg.ols_fit([], [[]])
    assert ei.value.code == a.ERROR_INVALID_INPUT
    # NULLs through the validity bitmask are dropped rows
    r = pkg.ols_fit([1.0, None, 3.0, 4.0, 5.5], [[1.0, 2.0, None, 4.0, 5.0]])
    assert r["n_observations"] == 3
    # single row: both columns constant -> intercept-only result, inference absent (len 0)
    r = pkg.ols_fit([2.5], [[1.0], [3.0]], {"compute_inference": True})
    assert r["intercept"] == 2.5 and all(np.isnan(c) for c in r["coefficients"]) and r["std_errors"] is None
    assert r["r_squared"] == 0.0 and np.isnan(r["residual_std_error"])


# --------------------------------------------------------------------------------------------------
# randomised parity against the oracle
# --------------------------------------------------------------------------------------------------
def _random_groups(rng, G, p, n_lo, n_hi, offset=0.0):
    ns = rng.integers(n_lo, n_hi + 1, size=G)
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    N = int(offs[-1])
    x_cols = [rng.uniform(-10, 10, N) + offset for _ in range(p)]
    beta = rng.uniform(-5, 5, (G, p))
    b0 = rng.uniform(-10, 10, G)
    gid = np.repeat(np.arange(G), ns)
    y = b0[gid] + sum(beta[gid, j] * x_cols[j] for j in range(p)) + 2.0 * rng.standard_normal(N)
    w = rng.uniform(0.5, 1.5, N)
    return offs, y, x_cols, w


@pytest.mark.parametrize("model", ["ols", "ridge", "wls"])
@pytest.mark.parametrize("p", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("icpt", [True, False])
def test_random_groups_match_oracle(pkg, ctx, model, p, icpt):
    rng = np.random.default_rng(1000 * p + (7 if icpt else 0) + len(model))
    offs, y, x_cols, w = _random_groups(rng, 192, p, 2, 700)
    kw = dict(fit_intercept=icpt, compute_inference=True, confidence_level=0.9)
    if model == "ridge":
        kw["alpha"] = 2.5
    wv = w if model == "wls" else None
    core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
    rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, **_oracle_kw(model, kw))
    assert_records_match(core, rcore, p, inf, rinf, what=f"{model} p={p} icpt={icpt}")


def test_bench_shape_groups_match_oracle(pkg, ctx):
    """cfg2 shape at a size the oracle finishes in seconds: n = 1000, p = 8."""
    synth = import_pkg("synth")
    offs, y, x_cols, w = synth.make_grouped(512, 1000, 8, weights=True)
    offs, y, w = offs.numpy(), y.numpy(), w.numpy()
    x_cols = [c.numpy() for c in x_cols]
    for model, kw in (("ols", {}), ("ols", {"compute_inference": True}), ("ridge", {"alpha": 1.0}),
                      ("wls", {"compute_inference": True})):
        wv = w if model == "wls" else None
        core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
        rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, n_threads=8, **_oracle_kw(model, kw))
        assert_records_match(core, rcore, 8, inf, rinf, what=f"bench-shape {model} {kw}")
        assert np.all(core[:, 8 + 5] == 0)


def test_large_offsets_and_scales(pkg, ctx):
    """Columns far from the origin and badly scaled columns: the shifted one-pass accumulation must hold the
    tolerance wherever the reference's own QR on the raw design does."""
    rng = np.random.default_rng(5)
    offs, y, x_cols, w = _random_groups(rng, 64, 4, 300, 900, offset=1000.0)
    x_cols[1] = x_cols[1] * 1e-4
    x_cols[2] = (x_cols[2] - 1000.0) * 1e5
    core, _ = _host_fit(pkg, ctx, "ols", offs, y, x_cols)
    rcore, _ = oracle.fit_groups(y, x_cols, offs, model="ols")
    # the oracle's own error here is ~cond(X)*eps ~ 1e-11 on the intercept: compare at 1e-8 / 1e-6
    assert_records_match(core, rcore, 4, coef_rtol=1e-8, what="offset 1e3")


def test_edge_cases_match_oracle(pkg, ctx):
    rng = np.random.default_rng(11)
    p = 3
    groups = []

    def add(n, mutate=None):
        X = rng.uniform(-10, 10, (n, p))
        y = 1.0 + X @ np.array([2.0, -1.0, 0.5]) + rng.standard_normal(n)
        w = rng.uniform(0.5, 1.5, n)
        if mutate:
            mutate(X, y, w)
        groups.append((X, y, w))

    add(0)                                                  # empty group -> NULL
    add(1)                                                  # one row -> NULL (aggregate rule)
    add(2)                                                  # 2 rows, 3 features -> InsufficientData
    add(4)                                                  # n == p + 1: zero residual df
    add(5)
    add(127); add(128); add(129); add(255); add(256); add(257)   # around the 128-row tile
    add(300, lambda X, y, w: X.__setitem__((slice(None), 1), 7.0))            # constant column
    add(300, lambda X, y, w: X.__setitem__((slice(None), 1), 7.0 + 5e-11))    # constant within 1e-10
    add(300, lambda X, y, w: X.__setitem__((slice(None), slice(None)), 3.0))  # all constant -> intercept only
    add(300, lambda X, y, w: X.__setitem__((slice(None), 2), 2 * X[:, 0] + 3))  # collinear -> aliased
    add(300, lambda X, y, w: y.__setitem__(slice(0, 300, 7), np.nan))          # NaN rows
    add(300, lambda X, y, w: X.__setitem__((slice(5, 300, 11), 0), np.inf))    # Inf rows
    add(300, lambda X, y, w: y.__setitem__(slice(None), np.nan))               # nothing valid
    add(300, lambda X, y, w: w.__setitem__(slice(0, 300, 3), 0.0))             # zero weights (WLS only)
    add(300, lambda X, y, w: w.__setitem__(slice(0, 300, 5), -1.0))            # negative weights
    add(200, lambda X, y, w: (X.__setitem__((slice(0, 130), slice(None)), np.nan)))  # first tile entirely invalid
    add(300, lambda X, y, w: y.__setitem__(slice(None), 1.0 + X @ np.array([2.0, -1.0, 0.5])))  # exact fit
    add(300, lambda X, y, w: y.__setitem__(slice(None), 4.25))                 # constant y
    ns = [len(g[1]) for g in groups]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    X = np.concatenate([g[0] for g in groups])
    y = np.concatenate([g[1] for g in groups])
    w = np.concatenate([g[2] for g in groups])
    x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    for model in ("ols", "ridge", "wls"):
        for icpt in (True, False):
            kw = dict(fit_intercept=icpt, compute_inference=True)
            wv = w if model == "wls" else None
            core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
            rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, **_oracle_kw(model, kw))
            # exact / constant-y fits: RSS ~ 1e-28, ratios of rounding noise -> compare those diagnostics loosely
            exact = [len(groups) - 2, len(groups) - 1]
            keep = np.ones(len(groups), dtype=bool)
            keep[exact] = False
            # group 3 has n == p + 1: zero residual degrees of freedom when an intercept is fitted
            assert_records_match(core[keep], rcore[keep], p, inf[keep], rinf[keep], what=f"edge {model} icpt={icpt}",
                                 skip_diag_groups=(3,) if icpt else ())
            assert np.array_equal(core[exact, p + 5], rcore[exact, p + 5])
            g = exact[0]
            if core[g, p + 5] == 0:
                assert np.allclose(core[g, :p], rcore[g, :p], rtol=1e-9, atol=1e-12, equal_nan=True)
                if icpt and model != "ridge":                    # y = 1 + x'b is exact only with an intercept
                    assert core[g, p + 3] < 1e-9                 # residual_std_error of an exact fit
                    assert abs(core[g, p + 1] - 1.0) < 1e-12


@pytest.mark.parametrize("hc", ["hc0", "hc1", "hc2", "hc3"])
@pytest.mark.parametrize("model", ["ols", "wls"])
def test_hc_standard_errors_match_oracle(pkg, ctx, model, hc):
    """hc_type replaces se / t / p / ci and keeps the classical F (ols.rs:209-231, wls.rs:230-252).  The
    estimator is upstream's un-vendored compute_hc_inference: parity is UNPINNED, the oracle restates the
    published sandwich estimator (checked against a dense numpy sandwich in tests/test_oracle_golden.py)."""
    for p, icpt in ((1, True), (3, True), (3, False), (8, True), (8, False)):
        rng = np.random.default_rng(31 * p + len(hc) + ord(hc[2]) + (5 if icpt else 0))
        offs, y, x_cols, w = _random_groups(rng, 96, p, p + 5, 700)
        y = y + np.abs(x_cols[0]) * rng.standard_normal(len(y))         # variance grows with |x_1|
        y[offs[5]:offs[6]:9] = np.nan                                   # invalid rows inside a group
        if p >= 3:
            x_cols[1][offs[7]:offs[8]] = 2.5                            # constant column -> NaN slot
            x_cols[2][offs[9]:offs[10]] = 3.0 * x_cols[0][offs[9]:offs[10]] - 1.0   # aliased column
        kw = dict(fit_intercept=icpt, compute_inference=True, confidence_level=0.9, hc_type=hc)
        wv = w if model == "wls" else None
        core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
        rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, **_oracle_kw(model, kw))
        assert_records_match(core, rcore, p, inf, rinf, what=f"{hc} {model} p={p} icpt={icpt}")
        kw["hc_type"] = "none"
        _, inf_classical = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
        assert np.array_equal(inf[:, 5 * p:], inf_classical[:, 5 * p:], equal_nan=True)   # F, F p-value
        assert not np.allclose(inf[:, :p], inf_classical[:, :p], equal_nan=True)


@pytest.mark.parametrize("hc", ["hc0", "hc1", "hc2", "hc3"])
@pytest.mark.parametrize("model", ["ols", "wls"])
def test_hc_standard_errors_wide_match_oracle(pkg, ctx, model, hc):
    for p, icpt in ((9, True), (20, False), (33, True), (128, True)):
        rng = np.random.default_rng(77 * p + len(hc) + ord(hc[2]) + (5 if icpt else 0))
        G = 6 if p == 128 else 12
        offs, y, x_cols, w = _random_groups(rng, G, p, p + 6, 3 * p + 60)
        y = y + np.abs(x_cols[0]) * rng.standard_normal(len(y))
        y[offs[1]:offs[2]:9] = np.nan                                   # invalid rows inside a group
        x_cols[1][offs[2]:offs[3]] = 2.5                                # constant column -> NaN slot
        x_cols[2][offs[3]:offs[4]] = 3.0 * x_cols[0][offs[3]:offs[4]] - 1.0   # aliased column
        kw = dict(fit_intercept=icpt, compute_inference=True, confidence_level=0.9, hc_type=hc)
        wv = w if model == "wls" else None
        core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
        rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, **_oracle_kw(model, kw))
        assert_records_match(core, rcore, p, inf, rinf, what=f"{hc} {model} p={p} icpt={icpt}")
        kw["hc_type"] = "none"
        _, inf_classical = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
        assert np.array_equal(inf[:, 5 * p:], inf_classical[:, 5 * p:], equal_nan=True)   # F, F p-value
        assert not np.allclose(inf[:, :p], inf_classical[:, :p], equal_nan=True)


def test_hc_is_ignored_where_the_reference_ignores_it(pkg, ctx):
    rng = np.random.default_rng(8)
    offs, y, x_cols, w = _random_groups(rng, 16, 3, 10, 50)
    base = _host_fit(pkg, ctx, "ridge", offs, y, x_cols, alpha=1.0, compute_inference=True)
    hc = _host_fit(pkg, ctx, "ridge", offs, y, x_cols, alpha=1.0, compute_inference=True, hc_type="hc3")
    assert np.array_equal(base[1], hc[1], equal_nan=True)              # ridge has no HC branch (ridge.rs)
    core_a, inf_a = _host_fit(pkg, ctx, "ols", offs, y, x_cols, hc_type="hc1")       # no inference requested
    core_b, _ = _host_fit(pkg, ctx, "ols", offs, y, x_cols)
    assert inf_a is None and np.array_equal(core_a, core_b, equal_nan=True)
    r = pkg.ols_fit(y[:40], [c[:40] for c in x_cols], {"compute_inference": True, "hc_type": "hc1"})
    code, d = oracle.fit(y[:40], [c[:40] for c in x_cols], compute_inference=True, hc_type="hc1")
    assert code == 0 and np.allclose(r["std_errors"], d["std_errors"], rtol=1e-8)


@pytest.mark.parametrize("model", ["ols", "ridge", "wls"])
@pytest.mark.parametrize("icpt", [True, False])
def test_very_large_groups_are_split_and_merged(pkg, ctx, model, icpt):
    """Groups with more than seg_rows (>= 8192) rows are cut into segments, one wavefront each, and the segment
    records are merged with a change of shift (accumulate_narrow.hip) — same results as the one-wave path."""
    rng = np.random.default_rng(91 + len(model) + (1 if icpt else 0))
    p = 5
    ns = [50_000, 3, 20_001, 100, 0, 8192, 8193, 30_000, 17_000, 1]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    N = int(offs[-1])
    X = rng.uniform(-10, 10, (N, p)) + np.array([0.0, 100.0, -3.0, 0.5, 1e3])
    gid = np.repeat(np.arange(len(ns)), ns)
    pos = np.arange(N) - offs[gid]
    X[:, 0] += 1e-3 * pos                                        # trend: the segments sit in different places
    y = 2.0 + X @ np.array([1.5, -0.2, 0.7, 3.0, 0.01]) + rng.standard_normal(N)
    w = rng.uniform(0.5, 1.5, N)
    lo = offs[0]
    y[lo:lo + 9000] = np.nan                                     # group 0: its whole first segment is invalid
    lo = offs[2]
    X[lo:lo + 20_001, 2] = 4.0                                   # group 2: constant column over every segment
    lo = offs[7]
    X[lo:lo + 30_000, 3] = np.where(np.arange(30_000) < 16_384, 1.0, 2.0)   # constant INSIDE each segment, not overall
    X[lo + 5:lo + 30_000:977, 1] = np.inf                        # scattered invalid rows
    lo = offs[8]
    X[lo:lo + 17_000, 4] = 2.0 * X[lo:lo + 17_000, 1] + 1.0      # aliased column
    x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    kw = dict(fit_intercept=icpt, compute_inference=True)
    if model == "ridge":
        kw["alpha"] = 0.7
    wv = w if model == "wls" else None
    core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
    rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, **_oracle_kw(model, kw))
    assert_records_match(core, rcore, p, inf, rinf, what=f"split {model} icpt={icpt}")
    assert np.isnan(core[2, 2]) and core[7, 5 + 5] == 0 and not np.isnan(core[7, 3])
    if model != "ridge":                                         # HC pass: overflow rows go to extra wavefronts, V by atomics
        kw["hc_type"] = "hc3"
        core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
        rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, **_oracle_kw(model, kw))
        assert_records_match(core, rcore, p, inf, rinf, what=f"split hc3 {model} icpt={icpt}")
    v = pkg.vif_batch_host(offs, x_cols[:4], ctx=ctx)            # the same accumulate kernel under vif_agg
    rv = oracle.vif_groups(x_cols[:4], offs)
    _assert_vif_match(v, rv, 4, "split vif")


@pytest.mark.parametrize("model,p", [("ols", 12), ("wls", 20), ("ridge", 32), ("ols", 40), ("wls", 100)])
def test_very_large_groups_are_split_and_merged_mid(pkg, ctx, model, p):
    """The same for the MFMA paths (8 < p <= 32: a wave per segment; beyond: a workgroup per segment): segments
    share the group's first valid row, the last finisher sums the segment records."""
    rng = np.random.default_rng(17 + p)
    ns = [30_000, 5, 9_000, 0, 8193, 200]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    N = int(offs[-1])
    X = rng.uniform(-10, 10, (N, p)) + rng.uniform(-50, 50, p)
    gid = np.repeat(np.arange(len(ns)), ns)
    pos = np.arange(N) - offs[gid]
    X[:, 0] += 1e-3 * pos
    y = 2.0 + X @ rng.uniform(-1, 1, p) + rng.standard_normal(N)
    w = rng.uniform(0.5, 1.5, N)
    y[0:9000] = np.nan                                           # group 0: first segment entirely invalid
    lo = offs[2]
    X[lo:lo + 9000, 3] = 4.0                                     # constant column over both segments
    X[lo:lo + 9000, 5] = np.where(np.arange(9000) < 8192, 1.0, 2.0)   # constant inside each segment only
    X[lo + 3:lo + 9000:501, 1] = np.nan
    x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    for icpt in (True, False):
        kw = dict(fit_intercept=icpt, compute_inference=True)
        if model == "ridge":
            kw["alpha"] = 0.7
        wv = w if model == "wls" else None
        core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
        rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, **_oracle_kw(model, kw))
        assert_records_match(core, rcore, p, inf, rinf, what=f"split mid {model} p={p} icpt={icpt}")
        assert np.isnan(core[2, 3]) and not np.isnan(core[2, 5])


@pytest.mark.parametrize("p", [3, 8, 12])
def test_fit_predict_on_very_large_groups(pkg, ctx, p):
    """Rows beyond the first seg_rows of a group are predicted by extra wavefronts (predict.hip)."""
    rng = np.random.default_rng(5 + p)
    ns = [40_000, 7, 8193, 0, 20_000]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    N = int(offs[-1])
    X = rng.uniform(-5, 5, (N, p))
    y = 1.0 + X @ rng.uniform(-1, 1, p) + 0.1 * rng.standard_normal(N)
    y[::5] = np.nan                                               # prediction rows
    x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    core, pred = pkg.fit_predict_batch_host(offs, y, x_cols, None, _opts(pkg, "ols", confidence_level=0.9), ctx=ctx)
    rcore, rpred = oracle.fit_predict_groups(y, x_cols, offs, model="ols", confidence_level=0.9)
    assert_records_match(core, rcore, p, what=f"big fit_predict p={p}")
    assert np.array_equal(np.isnan(pred), np.isnan(rpred))
    m = ~np.isnan(rpred)
    assert np.max(np.abs(pred[m] - rpred[m]) / np.maximum(np.abs(rpred[m]), 1.0)) < 1e-9


@pytest.mark.parametrize("avg", [6, 18, 40])
def test_fit_predict_small_groups_with_a_very_large_one(pkg, ctx, avg):
    """Batches averaging <= 64 rows per group predict several groups per wavefront (predict.hip, segments of 8 / 16 /
    32 lanes); a group beyond seg_rows inside such a batch still hands its tail to the extra wavefronts."""
    p = 4
    rng = np.random.default_rng(900 + avg)
    ns = rng.integers(0, 2 * avg + 1, size=3000)
    ns[1234] = 9000
    ns[17] = 700
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    N = int(offs[-1])
    assert N / len(ns) <= {6: 12, 18: 24, 40: 64}[avg]
    X = rng.uniform(-5, 5, (N, p))
    gid = np.repeat(np.arange(len(ns)), ns)
    y = rng.uniform(-2, 2, len(ns))[gid] + X @ rng.uniform(-1, 1, p) + 0.1 * rng.standard_normal(N)
    y[rng.random(N) < 0.2] = np.nan                               # prediction rows
    X[rng.random(N) < 0.01, 1] = np.nan                           # NULL feature: NULL prediction
    x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    core, pred = pkg.fit_predict_batch_host(offs, y, x_cols, None, _opts(pkg, "ols"), ctx=ctx)
    rcore, rpred = oracle.fit_predict_groups(y, x_cols, offs, model="ols")
    n_par = np.sum(~np.isnan(rcore[:, :p]), axis=1) + 1
    tight = np.flatnonzero((rcore[:, p + 5] == 0) & (rcore[:, p + 4] - n_par <= 1))
    assert_records_match(core, rcore, p, what=f"small fit_predict avg={avg}", skip_diag_groups=list(tight))
    assert np.array_equal(np.isnan(pred[:, 0]), np.isnan(rpred[:, 0]))
    m = ~np.isnan(rpred) & ~np.isin(gid, tight)[:, None]
    assert np.array_equal(np.isnan(pred[m]), np.isnan(rpred[m]))
    assert np.max(np.abs(pred[m] - rpred[m]) / np.maximum(np.abs(rpred[m]), 1.0)) < 1e-8


def test_inference_with_millions_of_rows(pkg, ctx):
    """df ~ 3e6: the incomplete-beta continued fraction needs thousands of terms there and ln Gamma(a + 1/2) -
    ln Gamma(a) cancels; p-values and critical values are checked against scipy's Student-t."""
    from scipy import stats as sps
    rng = np.random.default_rng(77)
    n = 3_000_000
    X = rng.standard_normal((n, 3))
    y = 0.5 + X @ np.array([0.0008, 0.3, 0.0]) + rng.standard_normal(n)     # weak, strong and null effects
    offs = np.array([0, n], dtype=np.int64)
    x_cols = [np.ascontiguousarray(X[:, j]) for j in range(3)]
    for hc in ("none", "hc1"):
        core, inf = _host_fit(pkg, ctx, "ols", offs, y, x_cols, compute_inference=True, confidence_level=0.99, hc_type=hc)
        assert core[0, 8] == 0
        df = n - 4
        coef, se, tv, pv, lo, hi = core[0, :3], inf[0, 0:3], inf[0, 3:6], inf[0, 6:9], inf[0, 9:12], inf[0, 12:15]
        assert np.allclose(tv, coef / se, rtol=1e-12)
        want_p = 2.0 * sps.t.sf(np.abs(tv), df)
        assert np.allclose(pv, want_p, rtol=1e-6, atol=0.0), (pv, want_p)
        tcrit = sps.t.ppf(0.995, df)
        assert np.allclose(hi - coef, tcrit * se, rtol=1e-9) and np.allclose(coef - lo, tcrit * se, rtol=1e-9)
    rcore, rinf = oracle.fit_groups(y, x_cols, offs, model="ols", compute_inference=True, confidence_level=0.99)
    core, inf = _host_fit(pkg, ctx, "ols", offs, y, x_cols, compute_inference=True, confidence_level=0.99)
    assert_records_match(core, rcore, 3, inf, rinf, what="3M rows")
    assert abs(pkg.t_critical(0.99, 10_000_000) / sps.t.ppf(0.995, 10_000_000) - 1.0) < 1e-12
    assert abs(pkg.t_critical(0.9, 250_000) / sps.t.ppf(0.95, 250_000) - 1.0) < 1e-12


def test_segment_tables_never_overflow(pkg, ctx):
    """The row-segment tables are sized from n_rows.  A caller of the device entry point that understates n_rows
    makes more segments than fit: the reservations are compare-and-swap bounded and a group that does not fit
    is accumulated by its own wavefront instead (slow, but the same result)."""
    import ctypes as C
    import torch
    a = import_pkg("_abi")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    ns = [8192 * 3000, 8192 * 1500 + 7, 500]                     # 4501 segments at seg_rows = 8192: more than the 4112 slots
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    N = int(offs[-1])
    x = rng.standard_normal(N)
    y = 1.0 + 2.0 * x + rng.standard_normal(N)
    t_off, t_y, t_x = (torch.from_numpy(v).to(dev) for v in (offs, y, x))
    opts = _opts(pkg, "ols", compute_inference=True)
    core = torch.empty((3, 7), dtype=torch.float64, device=dev)
    inf = torch.empty((3, 7), dtype=torch.float64, device=dev)
    cols = (C.c_void_p * 1)(t_x.data_ptr())
    err = a.AnofoxError()
    ok = ctx._lib.anofox_hip_fit_batch_device(ctx._h, 3, 1, 1, C.c_void_p(t_off.data_ptr()), C.c_void_p(t_y.data_ptr()), cols,
                                              C.c_void_p(0), opts, C.c_void_p(core.data_ptr()), C.c_void_p(inf.data_ptr()),
                                              C.byref(err))                       # n_rows = 1: a lie
    assert ok, err.text()
    torch.cuda.synchronize()
    good, good_inf = ctx.fit_batch_device(t_off, t_y, [t_x], None, opts)           # n_rows = N: everything is split
    torch.cuda.synchronize()
    assert torch.allclose(core, good, rtol=1e-10, atol=0.0, equal_nan=True)
    assert torch.allclose(inf, good_inf, rtol=1e-8, atol=0.0, equal_nan=True)
    rcore, _ = oracle.fit_groups(y[-500:], [x[-500:]], [0, 500], model="ols")
    assert np.allclose(good[2, :2].cpu().numpy(), rcore[0, :2], rtol=1e-9)


def test_device_calls_are_ordered_with_torch_streams(pkg, ctx):
    """The device entry points launch on torch's current stream — including the default stream, whose handle is 0 —
    so inputs produced by torch kernels just before the call are complete when the fit reads them, on the default
    stream and on a side stream alike."""
    import torch
    dev = torch.device("cuda:0")
    G, n = 20_000, 1000
    offs = torch.arange(0, G + 1, dtype=torch.int64, device=dev) * n
    opts = _opts(pkg, "ols")
    gen = torch.Generator(device=dev).manual_seed(5)
    for stream in (None, torch.cuda.Stream(device=dev)):
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream(dev)):
            x = torch.randn(G * n, dtype=torch.float64, device=dev, generator=gen)
            y = torch.zeros_like(x)
            torch.cuda.synchronize()
            # a chain of element-wise kernels that is still running when the fit is enqueued
            t = x.clone()
            for _ in range(20):
                t = torch.sin(t) + x
            y.copy_(1.0 + 2.0 * x + 0.0 * t)
            core, _ = ctx.fit_batch_device(offs, y, [x], None, opts)       # no synchronisation in between
            torch.cuda.synchronize()
            assert float((core[:, 0] - 2.0).abs().max()) < 1e-9 and float((core[:, 1] - 1.0).abs().max()) < 1e-9
            assert bool((core[:, 6] == 0).all())


@pytest.mark.parametrize("model", ["ols", "ridge", "wls"])
@pytest.mark.parametrize("avg", [6, 12, 28, 50])
def test_small_groups_share_a_wavefront(pkg, ctx, model, avg):
    """Batches that average at most 64 rows per group run several groups per wavefront (accumulate_small.hip:
    16- or 32-lane segments); groups too long for that kernel go through a list to the one-wave-per-group kernel,
    very large ones are split on top of that."""
    for p in (1, 3, 8):
        rng = np.random.default_rng(1000 * avg + 10 * p + len(model))
        G = 700
        ns = rng.integers(0, 2 * avg + 1, size=G)
        ns[5] = 300                                              # too long for the packed kernel -> list
        ns[77] = 1000
        if avg == 28:
            ns[200] = 9000                                       # > seg_rows: split into segments as well
        offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
        N = int(offs[-1])
        X = rng.uniform(-10, 10, (N, p)) + rng.uniform(-20, 20, p)
        gid = np.repeat(np.arange(G), ns)
        y = rng.uniform(-5, 5, G)[gid] + np.einsum("ij,ij->i", X, rng.uniform(-3, 3, (G, p))[gid]) + rng.standard_normal(N)
        w = rng.uniform(0.5, 1.5, N)
        y[rng.random(N) < 0.03] = np.nan                         # invalid rows, also first rows of groups
        X[rng.random(N) < 0.01, 0] = np.inf
        for g in range(10, G, 37):                               # constant columns
            X[offs[g]:offs[g + 1], p - 1] = 2.5
        if p >= 3:
            for g in range(20, G, 53):                           # aliased columns
                X[offs[g]:offs[g + 1], 2] = 3.0 * X[offs[g]:offs[g + 1], 0] - 1.0
        w[rng.random(N) < 0.02] = 0.0
        x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
        for icpt in (True, False):
            kw = dict(fit_intercept=icpt, compute_inference=True)
            if model == "ridge":
                kw["alpha"] = 0.8
            wv = w if model == "wls" else None
            core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
            rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, **_oracle_kw(model, kw))
            n_obs = rcore[:, p + 4]
            n_par = np.sum(~np.isnan(rcore[:, :p]), axis=1) + (1 if icpt else 0)
            tight = [g for g in range(G) if rcore[g, p + 5] == 0 and n_obs[g] - n_par[g] <= 0]
            assert_records_match(core, rcore, p, inf, rinf, what=f"small {model} avg={avg} p={p} icpt={icpt}",
                                 skip_diag_groups=tight)


def test_alpha_negative_and_bad_arguments(pkg, ctx):
    a = import_pkg("_abi")
    rng = np.random.default_rng(2)
    offs, y, x_cols, w = _random_groups(rng, 4, 2, 10, 20)
    core, _ = _host_fit(pkg, ctx, "ridge", offs, y, x_cols, alpha=-0.5)
    assert np.all(core[:, 2 + 5] == a.ERROR_INVALID_ALPHA) and np.all(np.isnan(core[:, :7]))
    with pytest.raises(pkg.AnofoxStatsError):
        _host_fit(pkg, ctx, "wls", offs, y, x_cols, None)                    # weights missing
    with pytest.raises(pkg.AnofoxStatsError):
        _host_fit(pkg, ctx, "ols", offs[::-1].copy(), y, x_cols)             # decreasing offsets
    with pytest.raises(pkg.AnofoxStatsError):
        _host_fit(pkg, ctx, "ols", offs, y, [x_cols[0]] * 129)               # more than anofox_hip_max_features()


def test_aggregate_update_semantics(pkg, ctx):
    """NULL y / NULL x-list rows are skipped by Update and do not count towards the 2-row rule; chunks may
    arrive in any order and through Combine."""
    agg = pkg.OlsFitAgg({"fit_intercept": True}, context=ctx)
    agg.update(["a", "a", "b", "a", "b"], [1.0, None, 5.0, 3.1, None], [[1.0], [2.0], [1.0], None, [4.0]])
    other = pkg.OlsFitAgg({"fit_intercept": True}, context=ctx)
    other.update(["a", "b", "c", "a", "c"], [5.0, 9.0, None, 7.2, None], [[3.0], [2.0], [1.0], [4.0], [2.0]])
    res = agg.combine(other).finalize().as_dict()
    code, r = oracle.fit([1.0, 5.0, 7.2], [[1.0, 3.0, 4.0]])
    assert res["a"]["n_observations"] == 3 and rel_err(res["a"]["coefficients"][0], r["coefficients"][0]) < 1e-12
    assert res["b"]["n_observations"] == 2 and abs(res["b"]["coefficients"][0] - 4.0) < 1e-12
    assert res["c"] is None                       # only NULL rows: state never initialised
    with pytest.raises(pkg.InvalidInputException, match="Inconsistent feature count: expected 1, got 2"):
        agg.update(["a"], [1.0], [[1.0, 2.0]])


# --------------------------------------------------------------------------------------------------
# device-resident path and size-independent properties at the BASELINE sizes
# --------------------------------------------------------------------------------------------------
def _device_fit(pkg, ctx, model, offs, y, x_cols, w=None, **kw):
    import torch
    core, inf = ctx.fit_batch_device(offs, y, x_cols, w, _opts(pkg, model, **kw))
    torch.cuda.synchronize()
    return core, inf


def test_device_path_cfg2_sample_vs_oracle_and_linearity(pkg, ctx):
    """BASELINE cfg2: OLS, 10k groups x 1000 x 8, device resident.  A 256-group sample is checked against the
    oracle; the whole batch is checked through linearity of least squares:
    fit(a*y + b*x_1 + c) has slopes a*beta + b*e_1, intercept a*beta0 + c, and R^2 ... of the same residuals."""
    import torch
    synth = import_pkg("synth")
    G, n, p = 10_000, 1000, 8
    offs, y, x_cols, _ = synth.make_grouped(G, n, p, device="cuda")
    core, inf = _device_fit(pkg, ctx, "ols", offs, y, x_cols, compute_inference=True)
    core_h, inf_h = core.cpu().numpy(), inf.cpu().numpy()
    assert np.all(core_h[:, p + 5] == 0) and np.all(core_h[:, p + 4] == n)
    S = 256
    ys = y[:S * n].cpu().numpy()
    xs = [c[:S * n].cpu().numpy() for c in x_cols]
    rcore, rinf = oracle.fit_groups(ys, xs, offs[:S + 1].cpu().numpy(), model="ols", compute_inference=True,
                                    n_threads=8)
    assert_records_match(core_h[:S], rcore, p, inf_h[:S], rinf, what="cfg2 sample")
    a, b, c = 2.0, 3.0, -1.0
    y2 = a * y + b * x_cols[0] + c
    core2, _ = _device_fit(pkg, ctx, "ols", offs, y2, x_cols)
    c2 = core2.cpu().numpy()
    want = a * core_h[:, :p].copy()
    want[:, 0] += b
    scale = np.max(np.abs(want), axis=1, keepdims=True)
    assert np.max(np.abs(c2[:, :p] - want) / np.maximum(np.abs(want), 1e-3 * scale)) < COEF_RTOL
    assert np.max(np.abs(c2[:, p] - (a * core_h[:, p] + c)) / np.maximum(np.abs(a * core_h[:, p] + c), 1e-3 * scale[:, 0])) < COEF_RTOL
    # residuals scale by a: sigma' = |a| sigma
    assert np.max(np.abs(c2[:, p + 3] / (abs(a) * core_h[:, p + 3]) - 1.0)) < DIAG_RTOL


def test_device_path_cfg3_full_size_properties(pkg, ctx):
    """BASELINE cfg3 at full size: ridge (alpha = 1, raw) and WLS on 1M groups x 1000 x 8, device resident
    (80 GB with weights).  Checked by (i) a 128-group sample against the oracle, (ii) WLS with all weights
    scaled by 4 leaves every statistic unchanged, (iii) ridge with alpha = 0 equals OLS."""
    import torch
    synth = import_pkg("synth")
    G, n, p = 1_000_000, 1000, 8
    release_device_memory()
    free, total = torch.cuda.mem_get_info()
    assert free >= 100e9, f"cfg3 needs ~100 GB of free HBM, {free / 1e9:.0f} of {total / 1e9:.0f} GB free"
    offs, y, x_cols, w = synth.make_grouped(G, n, p, weights=True, device="cuda", chunk_groups=32768)
    S = 128
    ys = y[:S * n].cpu().numpy()
    xs = [c[:S * n].cpu().numpy() for c in x_cols]
    ws = w[:S * n].cpu().numpy()
    so = offs[:S + 1].cpu().numpy()

    ridge, _ = _device_fit(pkg, ctx, "ridge", offs, y, x_cols, alpha=1.0)
    rr = ridge[:S].cpu().numpy()
    rcore, _ = oracle.fit_groups(ys, xs, so, model="ridge", alpha=1.0, n_threads=8)
    assert_records_match(rr, rcore, p, what="cfg3 ridge sample")
    assert bool((ridge[:, p + 5] == 0).all()) and bool((ridge[:, p + 4] == n).all())

    ridge0, _ = _device_fit(pkg, ctx, "ridge", offs, y, x_cols, alpha=0.0)
    ols, _ = _device_fit(pkg, ctx, "ols", offs, y, x_cols)
    # same factorisation, so the coefficients agree to rounding; RSS goes through b'Sxy instead of |L^-1 Sxy|^2
    assert float(((ridge0[:, :p + 1] - ols[:, :p + 1]).abs() / ols[:, :p + 1].abs().clamp_min(1e-3)).max()) < 1e-12
    assert float(((ridge0[:, p + 1:p + 4] / ols[:, p + 1:p + 4]) - 1.0).abs().max()) < 1e-9
    assert bool(torch.equal(ridge0[:, p + 4:], ols[:, p + 4:]))
    del ridge, ridge0

    wls, winf = _device_fit(pkg, ctx, "wls", offs, y, x_cols, w, compute_inference=True)
    rcore, rinf = oracle.fit_groups(ys, xs, so, w=ws, model="wls", compute_inference=True, n_threads=8)
    assert_records_match(wls[:S].cpu().numpy(), rcore, p, winf[:S].cpu().numpy(), rinf, what="cfg3 wls sample")
    w4 = w * 4.0
    wls4, _ = _device_fit(pkg, ctx, "wls", offs, y, x_cols, w4)
    d = (wls4[:, :p + 3] - wls[:, :p + 3]).abs() / wls[:, :p + 3].abs().clamp_min(1e-3)
    assert float(d.max()) < COEF_RTOL
    # sigma scales with sqrt(4)
    assert float(((wls4[:, p + 3] / (2.0 * wls[:, p + 3])) - 1.0).abs().max()) < DIAG_RTOL
    # OLS recovers the generating coefficients within sampling error (sigma = 2, n = 1000, x ~ U(-10, 10))
    assert float((ols[:, p + 3] - 2.0).abs().max()) < 0.5


# --------------------------------------------------------------------------------------------------
# wide designs (8 < p <= 128): FP64-MFMA accumulation + LDS Cholesky
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("p", [9, 16, 17, 33, 64, 100, 128])
@pytest.mark.parametrize("model", ["ols", "ridge", "wls"])
def test_wide_random_groups_match_oracle(pkg, ctx, p, model):
    rng = np.random.default_rng(77 * p + len(model))
    G = 24 if p > 64 else 48
    offs, y, x_cols, w = _random_groups(rng, G, p, max(2, p - 3), 3 * p + 40)   # some groups have n < p + 1
    for icpt in (True, False):
        kw = dict(fit_intercept=icpt, compute_inference=True, confidence_level=0.95)
        if model == "ridge":
            kw["alpha"] = 3.0
        wv = w if model == "wls" else None
        core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
        rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, n_threads=8, **_oracle_kw(model, kw))
        assert_records_match(core, rcore, p, inf, rinf, what=f"wide {model} p={p} icpt={icpt}")
        assert (rcore[:, p + 5] == 0).sum() >= G // 2


@pytest.mark.parametrize("p", list(range(9, 33)))
def test_every_mid_width_matches_oracle(pkg, ctx, p):
    """Every width 9 .. 32 on its own: the 4 x 4-block kernel's last column group holds y, the ones and 0 .. 3 padding
    columns depending on p mod 4 (accumulate_quad.hip), accumulate_mid switches load paths at 12, 16/17 and 27 — ragged
    groups (some with n < p + 1), NaN / inf rows, every model with and without intercept, inference on."""
    rng = np.random.default_rng(9000 + p)
    G = 40
    offs, y, x_cols, w = _random_groups(rng, G, p, max(2, p - 3), 3 * p + 150)
    X = np.stack(x_cols, 1)
    for g in range(0, G, 5):                                   # NaN / inf rows in every fifth group
        lo, hi = offs[g], offs[g + 1]
        if hi - lo > 6:
            X[lo + 1, rng.integers(0, p)] = np.nan
            y[lo + (hi - lo) // 2] = np.inf
    x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    for model in ("ols", "ridge", "wls"):
        for icpt in (True, False):
            kw = dict(fit_intercept=icpt, compute_inference=True, confidence_level=0.9)
            if model == "ridge":
                kw["alpha"] = 0.7
            wv = w if model == "wls" else None
            core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
            rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, n_threads=8, **_oracle_kw(model, kw))
            n_par = np.sum(~np.isnan(rcore[:, :p]), axis=1) + (1 if icpt else 0)
            tight = [g for g in range(G) if rcore[g, p + 5] == 0 and rcore[g, p + 4] - n_par[g] <= 0]
            assert_records_match(core, rcore, p, inf, rinf, what=f"width {p} {model} icpt={icpt}", skip_diag_groups=tight)


@pytest.mark.parametrize("p", [13, 16, 20, 30, 32])
def test_wide_edge_cases_match_oracle(pkg, ctx, p):
    """8 < p <= 32, every load path of accumulate_mid: p = 13 / 16 rows staged through LDS in 128-row blocks (with / without
    y and the ones inside the column block), 20 straight into fragment layout, 30 / 32 staged in 64-row blocks."""
    rng = np.random.default_rng(123 + p)
    groups = []

    def add(n, mutate=None):
        X = rng.uniform(-10, 10, (n, p))
        y = 1.0 + X @ rng.uniform(-2, 2, p) + rng.standard_normal(n)
        w = rng.uniform(0.5, 1.5, n)
        if mutate:
            mutate(X, y, w)
        groups.append((X, y, w))

    add(0); add(1); add(5); add(p + 1); add(p + 2); add(15); add(16); add(17); add(31); add(32); add(33); add(100)
    for n in (63, 64, 65, 127, 128, 129, 130, 191, 192, 193, 255, 256, 257, 300):           # around the 64- / 128-row blocks
        add(n)
    add(90, lambda X, y, w: X.__setitem__((slice(None), 3), 7.0))                     # constant column
    add(90, lambda X, y, w: X.__setitem__((slice(None), p - 1), -2.0 + 5e-11))        # constant within 1e-10 (last column)
    add(90, lambda X, y, w: X.__setitem__((slice(None), slice(None)), 3.0))           # all constant
    add(90, lambda X, y, w: X.__setitem__((slice(None), p - 3), 2 * X[:, 0] - X[:, 5] + 3))  # collinear -> aliased
    add(90, lambda X, y, w: y.__setitem__(slice(0, 90, 7), np.nan))
    add(90, lambda X, y, w: X.__setitem__((slice(5, 90, 11), p - 2), np.inf))
    add(90, lambda X, y, w: y.__setitem__(slice(None), np.nan))
    add(90, lambda X, y, w: w.__setitem__(slice(0, 90, 3), 0.0))
    add(90, lambda X, y, w: w.__setitem__(slice(0, 90, 5), -1.0))
    add(90, lambda X, y, w: X.__setitem__((slice(0, 40), slice(None)), np.nan))        # first chunks entirely invalid
    add(400, lambda X, y, w: X.__setitem__((slice(0, 131), 2), np.nan))                # first valid row: an odd row of the second block
    add(400, lambda X, y, w: y.__setitem__(slice(128, 256), np.nan))                   # a whole block invalid in the middle
    add(400, lambda X, y, w: y.__setitem__(slice(1, 400, 2), np.inf))                  # every odd row invalid
    add(333, lambda X, y, w: X.__setitem__((slice(300, 333), 0), np.nan))              # the partial last block entirely invalid
    ns = [len(g[1]) for g in groups]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    X = np.concatenate([g[0] for g in groups])
    y = np.concatenate([g[1] for g in groups])
    w = np.concatenate([g[2] for g in groups])
    x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    for model in ("ols", "ridge", "wls"):
        for icpt in (True, False):
            kw = dict(fit_intercept=icpt, compute_inference=True)
            wv = w if model == "wls" else None
            core, inf = _host_fit(pkg, ctx, model, offs, y, x_cols, wv, **kw)
            rcore, rinf = oracle.fit_groups(y, x_cols, offs, w=wv, **_oracle_kw(model, kw))
            # groups 3 / 4 (n = 21 / 22) have zero residual degrees of freedom with / without... n == p + 1 or n == p
            zero_df = (3,) if icpt else ()
            assert_records_match(core, rcore, p, inf, rinf, what=f"wide edge {model} p={p} icpt={icpt}", skip_diag_groups=zero_df)


@pytest.mark.parametrize("p", [9, 10, 12, 15, 16, 18, 22, 26, 27, 30, 31, 33, 34, 40, 46, 47, 48, 49, 62, 63, 64, 100, 128])
def test_wide_speculative_kernel_and_its_fallback(pkg, ctx, p):
    """p > 8, OLS with an intercept: the speculative accumulate kernels (every row valid, first row as the shift, constant
    columns read off the diagonal of the moments) — accumulate_quad's register-staged one (p <= 26: every column-group count
    and both block sizes), its LDS-DMA one (27 .. 33), accumulate_mid's LDS-DMA kernel on three / four column tiles (34 .. 64:
    with y and the ones inside the last tile, 46 / 62, and with side sums, 47 / 48 / 63 / 64) and accumulate_wide's (65 .. 128)
    — and the groups they have to hand to the full kernel: a NaN / inf anywhere, an invalid first row, columns in the band
    where sum d^2 does not decide |x - x_first| < 1e-10 — next to groups they keep, of every block-count shape (0, 1, 2
    rows; exactly k x 32 rows; one row more or less)."""
    rng = np.random.default_rng(1000 + p)
    groups = []

    def add(n, mutate=None):
        X = rng.uniform(-10, 10, (n, p))
        y = 1.0 + X @ rng.uniform(-2, 2, p) + rng.standard_normal(n)
        if mutate:
            mutate(X, y)
        groups.append((X, y))

    big = 3 * p + 40
    base = 32 * ((p + 2 + 31) // 32)            # a whole number of 32-row chunks with enough rows to fit
    for n in (0, 1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 128, 129, base + 31, base + 32, base + 33, base + 64, base + 65,
              big, 2 * big + 1):
        add(n)
    add(big, lambda X, y: X.__setitem__((slice(None), 3), 7.0))                          # constant column: kept, flagged from M_jj = 0
    # constant within 1e-10 but not exactly: sum d^2 = n / 2 * 2.5e-21 lies between 1e-20 and n 1e-20, where the diagonal
    # of the moments does not decide — the full kernel's per-row test does.  (A column that moves by MORE than 1e-10
    # in a few rows only is outside this test: it is "not constant" for ols.rs:76-87 and then a matter of the rank
    # tolerances, which the oracle takes against the uncentred and the Cholesky against the centred column norm.)
    add(big, lambda X, y: X.__setitem__((slice(None), p - 1), -2.0 + 5e-11 * (np.arange(big) % 2)))
    add(big, lambda X, y: X.__setitem__((0, 7), np.nan))                                  # invalid first row
    add(big, lambda X, y: y.__setitem__(min(70, big // 2 + 3), np.nan))                   # NaN in a middle chunk
    add(big, lambda X, y: X.__setitem__((big - 1, p - 2), np.inf))                        # inf in the last row
    add(big, lambda X, y: X.__setitem__((slice(0, 40), slice(None)), np.nan))             # first chunk entirely invalid
    add(big, lambda X, y: X.__setitem__((slice(None), p // 2), 2 * X[:, 0] - X[:, 1] + 3))   # aliased column
    add(big, lambda X, y: X.__setitem__((slice(None), slice(None)), X * 1e150))           # squares overflow: moments inf -> full kernel, same answer
    ns = [len(g[1]) for g in groups]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    X = np.concatenate([g[0] for g in groups])
    y = np.concatenate([g[1] for g in groups])
    x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    kw = dict(fit_intercept=True, compute_inference=True)
    core, inf = _host_fit(pkg, ctx, "ols", offs, y, x_cols, None, **kw)
    rcore, rinf = oracle.fit_groups(y, x_cols, offs, n_threads=8, **_oracle_kw("ols", kw))
    zero_df = [g for g in range(len(ns)) if rcore[g, p + 5] == 0 and rcore[g, p + 4] <= p + 1]
    skip = zero_df + [len(ns) - 1]            # (the overflow group: status and NaN pattern only)
    with np.errstate(all="ignore"):
        assert_records_match(core, rcore, p, inf, rinf, what=f"wide speculative p={p}", skip_diag_groups=skip)
    assert (rcore[:, p + 5] == 0).sum() >= 14


def test_wide_exact_fit_uses_residual_pass(pkg, ctx):
    rng = np.random.default_rng(9)
    p, n = 12, 200
    X = rng.uniform(-10, 10, (n, p))
    beta = rng.uniform(-2, 2, p)
    y = 0.5 + X @ beta
    core, _ = _host_fit(pkg, ctx, "ols", np.array([0, n]), y, [np.ascontiguousarray(X[:, j]) for j in range(p)])
    assert core[0, p + 5] == 0
    assert np.allclose(core[0, :p], beta, rtol=1e-9) and abs(core[0, p] - 0.5) < 1e-9
    assert core[0, p + 3] < 1e-9 and abs(core[0, p + 1] - 1.0) < 1e-12


def test_wide_cfg5_shape_sample_and_properties(pkg, ctx):
    """BASELINE cfg5 shape (n = 4096, p = 128, full diagnostics), device resident, at a group count the box
    generates in seconds; 16 groups against the oracle, all groups through the linearity property."""
    import torch
    synth = import_pkg("synth")
    G, n, p = 2048, 4096, 128
    offs, y, x_cols, _ = synth.make_grouped(G, n, p, device="cuda", chunk_groups=64)
    core, inf = _device_fit(pkg, ctx, "ols", offs, y, x_cols, compute_inference=True)
    ch, ih = core.cpu().numpy(), inf.cpu().numpy()
    assert np.all(ch[:, p + 5] == 0) and np.all(ch[:, p + 4] == n)
    S = 16
    rcore, rinf = oracle.fit_groups(y[:S * n].cpu().numpy(), [c[:S * n].cpu().numpy() for c in x_cols],
                                    offs[:S + 1].cpu().numpy(), compute_inference=True, n_threads=8)
    assert_records_match(ch[:S], rcore, p, ih[:S], rinf, what="cfg5 sample")
    y2 = 2.0 * y + 3.0 * x_cols[0] - 1.0
    core2, _ = _device_fit(pkg, ctx, "ols", offs, y2, x_cols)
    c2 = core2.cpu().numpy()
    want = 2.0 * ch[:, :p].copy()
    want[:, 0] += 3.0
    scale = np.max(np.abs(want), axis=1, keepdims=True)
    assert np.max(np.abs(c2[:, :p] - want) / np.maximum(np.abs(want), 1e-3 * scale)) < COEF_RTOL
    assert np.max(np.abs(c2[:, p + 3] / (2.0 * ch[:, p + 3]) - 1.0)) < DIAG_RTOL
    assert float(np.max(np.abs(ch[:, p + 3] - 2.0))) < 0.2        # sigma of the generator


# --------------------------------------------------------------------------------------------------
# fit-predict family: *_fit_predict_agg, predict, predict_with_interval, t_critical
# --------------------------------------------------------------------------------------------------
def test_predict_agg_reference_structural_tests(pkg, ctx):
    """test/sql/predict_agg/test_ols_predict_agg.test:8-118: y = 2i + 1 for i <= 7, NULL for i = 8..10."""
    i = np.arange(1, 11, dtype=np.float64)
    y = [float(2 * v + 1) if v <= 7 else None for v in i]
    res = pkg.ols_fit_predict_agg(np.zeros(10, dtype=np.int64), y, [[v] for v in i], context=ctx)
    rows = res.rows(0)
    assert len(rows) == 10
    assert sum(r["is_training"] for r in rows) == 7 and sum(not r["is_training"] for r in rows) == 3
    assert all(r["yhat"] is not None and r["yhat_upper"] >= r["yhat_lower"] for r in rows)
    assert all(r["y"] is not None for r in rows if r["is_training"])
    assert all(r["y"] is None for r in rows if not r["is_training"])
    assert max(abs(r["yhat"] - (2 * v + 1)) for r, v in zip(rows, i)) < 1e-9
    # null_policy = drop_y_zero_x: a zero feature value removes the row from training (test :87-107)
    x2 = [0.0 if v == 3 else v for v in i]
    res = pkg.SQL_FUNCTIONS["ols_predict_agg"](np.zeros(10, dtype=np.int64), y, [[a, b] for a, b in zip(i, x2)],
                                               {"null_policy": "drop_y_zero_x"}, context=ctx)
    assert sum(r["is_training"] for r in res.rows(0)) == 6
    with pytest.raises(pkg.InvalidInputException, match="Invalid null_policy"):
        pkg.parse_options({"null_policy": "keep"})


@pytest.mark.parametrize("model", ["ols", "ridge", "wls"])
@pytest.mark.parametrize("p", [2, 8, 20])
def test_fit_predict_batch_matches_oracle(pkg, ctx, model, p):
    rng = np.random.default_rng(31 * p + len(model))
    G = 40
    offs, y, x_cols, w = _random_groups(rng, G, p, 1, 4 * p + 30)
    # prediction rows (NULL y), rows with a NULL feature, groups with too few training rows
    y = y.copy()
    y[rng.random(len(y)) < 0.25] = np.nan
    x_cols = [c.copy() for c in x_cols]
    x_cols[0][rng.random(len(y)) < 0.03] = np.nan
    train = ~np.isnan(y) & ~np.isnan(x_cols[0])
    gid = np.repeat(np.arange(G), np.diff(offs))
    train_counts = np.bincount(gid, weights=train, minlength=G).astype(np.int64)
    for icpt in (True, False):
        kw = dict(fit_intercept=icpt, confidence_level=0.9)
        if model == "ridge":
            kw["alpha"] = 0.7
        wv = w if model == "wls" else None
        core, pred = pkg.fit_predict_batch_host(offs, y, x_cols, wv, _opts(pkg, model, **kw),
                                                train_counts=train_counts, ctx=ctx)
        rcore, rpred = oracle.fit_predict_groups(y, x_cols, offs, w=wv, train_counts=train_counts,
                                                 **_oracle_kw(model, kw))
        zero_df = [g for g in range(G) if rcore[g, p + 5] == 0 and rcore[g, p + 4] - np.sum(~np.isnan(rcore[g, :p])) - icpt == 0]
        assert_records_match(core, rcore, p, what=f"fit_predict {model} p={p} icpt={icpt}", skip_diag_groups=zero_df)
        assert np.array_equal(np.isnan(pred), np.isnan(rpred))
        m = ~np.isnan(rpred)
        rowgrp = np.repeat(np.arange(G), np.diff(offs))
        skip = np.isin(rowgrp, zero_df)
        m &= ~skip[:, None]
        scale = np.maximum(np.abs(rpred[m]), 1e-3 * np.abs(rpred[~np.isnan(rpred)]).max())
        assert np.max(np.abs(pred[m] - rpred[m]) / scale) < 1e-9


def test_scalar_prediction_helpers(pkg):
    from scipy import stats as sps
    for df in (1, 2, 5, 17, 146, 991):
        for c in (0.8, 0.95, 0.99):
            assert rel_err(pkg.t_critical(c, df), sps.t.ppf(0.5 * (1 + c), df)) < 1e-10
            assert rel_err(pkg.t_critical(c, df), oracle.t_critical(c, df)) < 1e-12
    assert np.isnan(pkg.t_critical(0.95, 0)) and np.isnan(pkg.t_critical(1.0, 5))
    for coef, icpt, xn, rse, n in (([2.0, np.nan], 1.0, [3.0, 5.0], 0.5, 20), ([2.0], np.nan, [3.0], 0.5, 20),
                                   ([2.0], 1.0, [3.0], np.nan, 20), ([2.0], 1.0, [3.0], 0.5, 2)):
        got = pkg.predict_with_interval(coef, icpt, xn, rse, n, 0.95)
        ok, want = oracle.predict_with_interval(coef, icpt, xn, rse, n, 0.95)
        assert ok and max(abs(got[k] - want[i]) for i, k in enumerate(("yhat", "yhat_lower", "yhat_upper"))) < 1e-12
    # anofox_predict: plain X beta (+ intercept); runs on the GPU
    x = [[1.0, 2.0, 3.0, None], [0.5, -1.0, 2.0, 4.0]]
    out = pkg.predict(x, [2.0, -3.0], 0.25)
    assert out[:3] == [0.25 + 2 * 1 - 3 * 0.5, 0.25 + 4 + 3, 0.25 + 6 - 6] and np.isnan(out[3])
    assert pkg.predict(x, [2.0, -3.0]) [0] == 2 * 1 - 3 * 0.5           # NaN intercept = none
    assert all(np.isnan(v) for v in pkg.predict(x, [2.0, float("nan")], 1.0))   # NaN coefficients poison (predict.rs:55-61)
    with pytest.raises(pkg.InvalidInputException):
        pkg.predict(x, [1.0, 2.0, 3.0])


def test_fit_predict_device_cfg_shape_properties(pkg, ctx):
    """Device-resident fit + predict at the reference's published 1M-group benchmark shape scaled to fit a test
    (100k groups x 100 rows x p = 3, examples/performance_1m_groups/benchmark_ols.sql): every 5th row is a
    prediction row.  yhat of the training rows must reproduce y - residual (mean residual 0 with an intercept),
    and a sample is checked against the oracle."""
    import torch
    synth = import_pkg("synth")
    G, n, p = 100_000, 100, 3
    offs, y, x_cols, _ = synth.make_grouped(G, n, p, device="cuda")
    hold = (torch.arange(G * n, device="cuda") % 5) == 4
    y_fit = torch.where(hold, torch.full_like(y, float("nan")), y)
    core, pred = ctx.fit_predict_batch_device(offs, y_fit, x_cols, None, _opts(pkg, "ols"))
    torch.cuda.synchronize()
    assert bool((core[:, p + 5] == 0).all()) and bool((core[:, p + 4] == 80).all())
    assert not bool(torch.isnan(pred).any())
    resid = torch.where(hold, torch.zeros_like(y), y - pred[:, 0]).reshape(G, n)
    assert float(resid.sum(dim=1).abs().max()) < 1e-8                      # residuals of the training rows sum to 0
    assert bool((pred[:, 2] >= pred[:, 1]).all())
    S = 64
    rcore, rpred = oracle.fit_predict_groups(y_fit[:S * n].cpu().numpy(), [c[:S * n].cpu().numpy() for c in x_cols],
                                             offs[:S + 1].cpu().numpy(), model="ols")
    assert_records_match(core[:S].cpu().numpy(), rcore, p, what="fit_predict device sample")
    assert np.max(np.abs(pred[:S * n].cpu().numpy() - rpred) / np.maximum(np.abs(rpred), 1.0)) < 1e-9


# --------------------------------------------------------------------------------------------------
# *_fit_predict window functions (expanding frames)
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model", ["ols", "ridge", "wls"])
@pytest.mark.parametrize("p", [1, 3, 8])
def test_expanding_window_matches_oracle(pkg, ctx, model, p):
    rng = np.random.default_rng(17 * p + len(model))
    G = 24
    offs, y, x_cols, w = _random_groups(rng, G, p, 1, 180)
    y = y.copy()
    y[rng.random(len(y)) < 0.15] = np.nan                       # prediction rows (NULL y)
    x_cols = [c.copy() for c in x_cols]
    x_cols[0][rng.random(len(y)) < 0.02] = np.nan               # NULL feature: no prediction, no training
    if p >= 3:                                                  # a column that only starts to vary late
        lo, hi = offs[3], offs[4]
        x_cols[1][lo:hi] = 2.5
        if hi - lo > 20:
            x_cols[1][lo + 15:hi] = rng.uniform(-1, 1, hi - lo - 15)
    for icpt in (True, False):
        kw = dict(fit_intercept=icpt, confidence_level=0.9)
        if model == "ridge":
            kw["alpha"] = 0.5
        wv = w if model == "wls" else None
        pred = pkg.fit_predict_expanding_host(offs, y, x_cols, wv, _opts(pkg, model, **kw), ctx=ctx)
        ref = oracle.fit_predict_expanding(y, x_cols, offs, w=wv, **_oracle_kw(model, kw))
        assert np.array_equal(np.isnan(pred[:, 0]), np.isnan(ref[:, 0])), f"NULL pattern {model} p={p} icpt={icpt}"
        m = ~np.isnan(ref[:, 0])
        scale = np.maximum(np.abs(ref[m, 0]), 1.0)
        # every row: ill-conditioned early prefixes (n barely above p) are flagged and refitted with refinement
        err = np.abs(pred[m, 0] - ref[m, 0]) / scale
        assert err.max() < 1e-8, (model, p, icpt, err.max())
        wid_ref = ref[m, 2] - ref[m, 1]
        wid = pred[m, 2] - pred[m, 1]
        ok = np.isfinite(wid_ref) & (wid_ref > 1e-6 * scale)
        assert np.quantile(np.abs(wid[ok] / wid_ref[ok] - 1.0), 0.98) < 1e-6


@pytest.mark.parametrize("model", ["ols", "ridge", "wls"])
@pytest.mark.parametrize("frame", [(9, 0), (20, 0), (9, 1), (7, 3), (2, 0), (89, 0), (150, 150), (None, 2), (None, 0),
                                   (5, -5), (0, -12), (-2, -9), (10, None), (None, -3), (None, None), (-250, -250)])
def test_window_frames_match_oracle(pkg, ctx, model, frame):
    """ROWS BETWEEN a PRECEDING AND b PRECEDING — the rolling and lagged frames of the reference's docs and tests
    (e.g. test/sql: '9 PRECEDING AND CURRENT ROW', '7 PRECEDING AND 3 PRECEDING', '89 PRECEDING AND CURRENT ROW',
    'UNBOUNDED PRECEDING AND 1 PRECEDING'): every frame refitted by the oracle."""
    for p in (1, 3, 8):
        rng = np.random.default_rng(1000 + 7 * p + len(model) + abs(frame[0] or 0) + 13 * abs(frame[1] or 0))
        offs, y, x_cols, w = _random_groups(rng, 10, p, 1, 200)
        x_cols = [c.copy() for c in x_cols]
        gid = np.repeat(np.arange(len(offs) - 1), np.diff(offs))
        pos = np.arange(len(y)) - offs[gid]
        x_cols[0] = 1000.0 + pos + 0.3 * rng.standard_normal(len(y))   # trending regressor far from zero
        y = y + 0.05 * pos
        y[rng.random(len(y)) < 0.15] = np.nan                          # prediction rows (NULL y)
        x_cols[-1][rng.random(len(y)) < 0.02] = np.nan                 # NULL feature
        for icpt in (True, False):
            kw = dict(fit_intercept=icpt, confidence_level=0.9)
            if model == "ridge":
                kw["alpha"] = 0.5
            wv = w if model == "wls" else None
            pred = pkg.fit_predict_window_host(offs, y, x_cols, wv, _opts(pkg, model, **kw), frame, ctx=ctx)
            ref = oracle.fit_predict_window(y, x_cols, offs, w=wv, start_preceding=frame[0],
                                            end_preceding=frame[1], **_oracle_kw(model, kw))
            what = f"{model} p={p} icpt={icpt} frame={frame}"
            assert np.array_equal(np.isnan(pred[:, 0]), np.isnan(ref[:, 0])), f"NULL pattern {what}"
            m = ~np.isnan(ref[:, 0])
            if not m.any():
                continue
            scale = np.maximum(np.abs(ref[m, 0]), 1.0)
            # every row: frames with barely more rows than parameters, and (without an intercept) the trending column
            # (values ~1000, spread ~frame: conditioned like 1e6 in the normal equations), are flagged by the kernel and
            # refitted with the fit path's refinement passes
            err = np.abs(pred[m, 0] - ref[m, 0]) / scale
            assert err.max() < (1e-8 if icpt else 1e-7), (what, err.max())
            wid_ref = ref[m, 2] - ref[m, 1]
            wid = pred[m, 2] - pred[m, 1]
            ok = np.isfinite(wid_ref) & (wid_ref > 1e-6 * scale)
            if ok.any():
                assert np.max(np.abs(wid[ok] / wid_ref[ok] - 1.0)) < 1e-5, what


def test_window_reference_sql_structural_tests(pkg, ctx):
    """test/sql/fit_predict/test_ridge_fit_predict_rolling.test (non-NULL counts 13 / 13 / 8 / 13 and the partitioned
    case) and test_wls_fit_predict_edge.test (8): which rows get a prediction depends only on the frame, not on the
    random values of the reference's tables."""
    rng = np.random.default_rng(0)
    t = np.arange(1, 16)
    x = t.astype(float)
    y = 2.0 * t + 0.5 * rng.random(15)
    keys = np.zeros(15, dtype=np.int64)
    opts = {"intercept": 1.0, "alpha": 1.0}
    X = x[:, None].tolist()

    def count(frame, sel=slice(None)):
        yh, _, _ = pkg.ridge_fit_predict(keys, t, y, X, opts, context=ctx, frame=frame)
        return int(np.sum(~np.isnan(yh[sel])))

    assert count(("unbounded", "current row")) == 13
    assert count(("4 preceding", "current row")) == 13
    assert count(("7 preceding", "3 preceding"), t >= 8) == 8
    assert count(("3 preceding", "current row")) == 13
    tt = np.tile(np.arange(1, 11), 2)
    kk = np.array(["A"] * 10 + ["B"] * 10)
    yy = tt * np.where(kk == "A", 2.0, 3.0) + 0.3 * rng.random(20)
    yh, _, _ = pkg.ridge_fit_predict(kk, tt, yy, tt.astype(float)[:, None].tolist(), {"intercept": 1.0, "alpha": 0.5},
                                     context=ctx, frame=("3 preceding", "current row"))
    assert np.all(~np.isnan(yh[(tt >= 3) & (tt <= 5)])) and np.all(np.isnan(yh[tt <= 2]))
    i = np.arange(1, 11)
    ye = [float(2.0 * v + 1.0) if v <= 6 else None for v in i]
    yh, lo, hi = pkg.wls_fit_predict(np.zeros(10, dtype=np.int64), i, ye, i.astype(float)[:, None].tolist(), np.ones(10), context=ctx)
    assert int(np.sum(~np.isnan(yh))) == 8 and np.all(np.isnan(yh[:2]))
    assert np.allclose(yh[2:], 2.0 * i[2:] + 1.0, rtol=1e-9)        # exact line: the later rows are extrapolated


def test_window_frame_validation(pkg, ctx):
    rng = np.random.default_rng(5)
    offs, y, x_cols, w = _random_groups(rng, 3, 2, 5, 30)
    for bad in ((3, 5), (-2, -1), (0, 1)):                           # the frame must start at or before its end
        with pytest.raises(pkg.AnofoxStatsError):
            pkg.fit_predict_window_host(offs, y, x_cols, None, _opts(pkg, "ols"), bad, ctx=ctx)
    a = pkg.ols_fit_predict(np.zeros(len(y), dtype=np.int64), np.arange(len(y)), y, np.stack(x_cols, 1).tolist(), context=ctx,
                            frame=("2 preceding", "3 following"))
    b = pkg.fit_predict_window_host(np.array([0, len(y)]), y, x_cols, None, _opts(pkg, "ols"), (2, -3), ctx=ctx)
    assert np.array_equal(a[0], b[:, 0], equal_nan=True) and not np.all(np.isnan(b[:, 0]))
    # UNBOUNDED PRECEDING AND UNBOUNDED FOLLOWING: every row of a partition gets the fit on the whole partition,
    # predicting the partition's last row (ols_fit_predict.cpp:157-162)
    full = pkg.fit_predict_window_host(offs, y, x_cols, None, _opts(pkg, "ols"), (None, None), ctx=ctx)
    for g in range(3):
        blk = full[offs[g]:offs[g + 1]]
        assert np.all(blk == blk[0]) and np.isfinite(blk[0, 0])
    keys = np.zeros(len(y), dtype=np.int64)
    a = pkg.ols_fit_predict(keys, np.arange(len(y)), y, np.stack(x_cols, 1).tolist(), context=ctx, frame=("9 preceding", "current row"))
    b = pkg.fit_predict_window_host(np.array([0, len(y)]), y, x_cols, None, _opts(pkg, "ols"), (9, 0), ctx=ctx)
    assert np.array_equal(a[0], b[:, 0], equal_nan=True)
    with pytest.raises(Exception):
        pkg.ols_fit_predict(keys, np.arange(len(y)), y, np.stack(x_cols, 1).tolist(), context=ctx, frame=("current row", "3 preceding"))


def test_window_function_mirror_and_frames(pkg, ctx):
    """ols_fit_predict OVER (PARTITION BY g ORDER BY t): unsorted input, NULL y rows are predicted, and the
    '1 preceding' frame of the reference's benchmark (examples/performance_1m_groups/benchmark_ols.sql:16-19) is
    the 'current row' result shifted by one row inside each partition."""
    rng = np.random.default_rng(4)
    n = 60
    keys = np.array(["a"] * n + ["b"] * n)
    t = np.concatenate([rng.permutation(n), rng.permutation(n)])
    X = rng.uniform(-5, 5, (2 * n, 2))
    y = 1.5 + X @ np.array([2.0, -1.0]) + 0.1 * rng.standard_normal(2 * n)
    yl = [None if (ti % 7 == 6) else float(v) for v, ti in zip(y, t)]
    yh, lo, hi = pkg.ols_fit_predict(keys, t, yl, X.tolist(), {"confidence_level": 0.9}, context=ctx)
    yh1, _, _ = pkg.ols_fit_predict(keys, t, yl, X.tolist(), {"confidence_level": 0.9}, context=ctx, frame_end="1 preceding")
    for k in ("a", "b"):
        sel = np.nonzero(keys == k)[0]
        order = sel[np.argsort(t[sel])]
        assert np.all(np.isnan(yh[order[:3]]))                 # needs more than p + 1 = 3 training rows
        assert np.all(np.isfinite(yh[order[8:]])) and np.all(hi[order[8:]] >= lo[order[8:]])
        assert np.isnan(yh1[order[0]])
        # shifted: row i of the '1 preceding' frame = prediction made at row i-1 (with x of row i-1)
        assert np.array_equal(yh1[order[1:]], yh[order[:-1]], equal_nan=True)
        late = order[30:]
        assert np.max(np.abs(yh[late] - (1.5 + X[late] @ np.array([2.0, -1.0])))) < 0.3


def test_expanding_window_device_reference_benchmark_shape(pkg, ctx):
    """The reference's published window benchmark shape (1M partitions x 100 rows x p = 3,
    examples/performance_1m_groups/benchmark_ols.sql) at 200k partitions, device resident: a sample against the
    oracle, and the last row of every partition against the whole-partition fit."""
    import torch
    synth = import_pkg("synth")
    G, n, p = 200_000, 100, 3
    offs, y, x_cols, _ = synth.make_grouped(G, n, p, device="cuda")
    pred = ctx.fit_predict_expanding_device(offs, y, x_cols, None, _opts(pkg, "ols"))
    core, pall = ctx.fit_predict_batch_device(offs, y, x_cols, None, _opts(pkg, "ols"))
    torch.cuda.synchronize()
    last = (offs[1:] - 1)
    assert float((pred[last] - pall[last]).abs().max() / pall[last].abs().max()) < 1e-9
    first_rows = pred.reshape(G, n, 3)[:, :4, 0]
    assert bool(torch.isnan(first_rows).all()) and not bool(torch.isnan(pred.reshape(G, n, 3)[:, 4:, :]).any())
    S = 16
    ref = oracle.fit_predict_expanding(y[:S * n].cpu().numpy(), [c[:S * n].cpu().numpy() for c in x_cols],
                                       offs[:S + 1].cpu().numpy(), model="ols")
    got = pred[:S * n].cpu().numpy()
    m = ~np.isnan(ref[:, 0])
    assert np.array_equal(np.isnan(got[:, 0]), ~m)
    assert np.quantile(np.abs(got[m] - ref[m]) / np.maximum(np.abs(ref[m]), 1.0), 0.98) < 1e-9


# --------------------------------------------------------------------------------------------------
# variance inflation factors (vif_agg / vif): p OLS fits per group from one Gram matrix
# --------------------------------------------------------------------------------------------------
def _assert_vif_match(got, ref, p, what):
    assert np.array_equal(got[:, p], ref[:, p]), f"{what}: status"
    g, r = got[:, :p], ref[:, :p]
    assert np.array_equal(np.isnan(g), np.isnan(r)), f"{what}: NaN pattern"
    assert np.array_equal(np.isinf(g), np.isinf(r)), f"{what}: inf pattern"
    m = np.isfinite(r)
    # VIF = 1/(1 - R^2): R^2 is a diagnostic (1e-6); the map amplifies its error by VIF <= 1e4
    assert np.all(np.abs(g[m] - r[m]) <= 1e-6 * r[m] * np.maximum(r[m], 1.0) * 1e-2 + 1e-9 * r[m]), f"{what}: values"


@pytest.mark.parametrize("p", [1, 2, 3, 5, 8, 9, 12])
def test_vif_batch_matches_oracle(pkg, ctx, p):
    rng = np.random.default_rng(40 + p)
    offs, _, x_cols, _ = _random_groups(rng, 64, p, 1, 400)
    x_cols = [c.copy() for c in x_cols]
    if p >= 2:
        lo, hi = offs[5], offs[6]
        x_cols[1][lo:hi] = 0.9 * x_cols[0][lo:hi] + 0.2 * rng.standard_normal(hi - lo)     # strong, not perfect
        lo, hi = offs[7], offs[8]
        x_cols[1][lo:hi] = 2.0 * x_cols[0][lo:hi] + 1.0                                    # perfect -> inf
        lo, hi = offs[9], offs[10]
        x_cols[0][lo:hi] = 4.0                                                             # constant feature
        lo, hi = offs[11], offs[12]
        x_cols[p - 1][lo:hi:5] = np.inf                                                    # rows the fits drop
    got = pkg.vif_batch_host(offs, x_cols, ctx=ctx)
    ref = oracle.vif_groups(x_cols, offs)
    _assert_vif_match(got, ref, p, f"vif p={p}")


def test_vif_reference_sql_tests(pkg, ctx):
    """test/sql/diagnostics/test_vif_agg.test and test/sql/scalar/test_diagnostics_scalar.test:64-83."""
    low = np.array([[1, 5, 9], [2, 3, 7], [3, 8, 2], [4, 1, 6], [5, 9, 4], [6, 2, 8], [7, 7, 3], [8, 4, 5], [9, 6, 1], [10, 10, 10]], float)
    high = np.array([[1, 2.1, 5], [2, 4.0, 6], [3, 6.1, 7], [4, 7.9, 8], [5, 10.0, 9], [6, 12.1, 10], [7, 13.9, 11], [8, 16.0, 12],
                     [9, 18.1, 13], [10, 19.9, 14]], float)
    keys, res = pkg.vif_agg(["low"] * 10 + ["high"] * 10, np.concatenate([low, high]).tolist(), context=ctx)
    d = dict(zip(keys.tolist(), res))
    assert len(d["low"]) == 3 and min(d["low"]) >= 1.0 and max(d["low"]) < 5
    assert max(d["high"]) > 5
    _, two = pkg.vif_agg([0] * 10, low[:, :2].tolist(), context=ctx)
    assert len(two[0]) == 2
    keys, res = pkg.vif_agg(["low"] * 5 + ["high"] * 5, np.concatenate([low[:5], high[:5]]).tolist(), context=ctx)
    d = dict(zip(keys.tolist(), res))
    assert max(d["low"]) < 5 and not max(d["high"]) < 5
    # NULL rules of the aggregate: one feature, fewer than 3 rows, a NaN that shortens one column
    assert pkg.vif_agg([0] * 10, low[:, :1].tolist(), context=ctx)[1] == [None]
    assert pkg.vif_agg([0] * 2, low[:2].tolist(), context=ctx)[1] == [None]
    bad = low.tolist()
    bad[3][1] = float("nan")
    assert pkg.vif_agg([0] * 10, bad, context=ctx)[1] == [None]
    rows = low.tolist()
    rows[4] = None                                                              # NULL list: skipped
    ref = oracle.vif_groups([np.delete(low[:, j], 4) for j in range(3)], [0, 9])[0, :3]
    assert np.allclose(pkg.vif_agg([0] * 10, rows, context=ctx)[1][0], ref, rtol=1e-8)
    # scalar: the argument is a list of feature COLUMNS (5 "features" of 2 observations here -> every fit fails)
    v = pkg.vif([[1.0, 5.0], [2.0, 3.0], [3.0, 8.0], [4.0, 1.0], [5.0, 9.0]])
    assert len(v) == 5 and min(v) >= 1.0
    assert pkg.vif([[1.0, 2, 3, 4, 5]]) == [1.0]
    v = pkg.vif([low[:, 0].tolist(), low[:, 1].tolist(), low[:, 2].tolist()])
    assert np.allclose(v, oracle.vif_groups([low[:, j] for j in range(3)], [0, 10], min_rows=0)[0, :3], rtol=1e-8)
    with pytest.raises(pkg.InvalidInputException, match="Feature 1 has 2 observations, expected 3"):
        pkg.vif([[1.0, 2.0, 3.0], [1.0, 2.0]])
    with pytest.raises(pkg.InvalidInputException, match="x is NULL or empty"):
        pkg.vif([])


def test_vif_device_full_size_properties(pkg, ctx):
    """1M groups x 100 rows x p = 3 independent features, device resident: VIF ~ 1, and a sample against the oracle."""
    import torch
    synth = import_pkg("synth")
    dev = torch.device("cuda:0")
    G, n, p = 1_000_000, 100, 3
    offs, _, x_cols, _ = synth.make_grouped(G, n, p, device=dev)
    out = ctx.vif_batch_device(offs, x_cols)
    torch.cuda.synchronize()
    assert bool((out[:, p] == 0).all())
    v = out[:, :p]
    assert bool(torch.isfinite(v).all()) and float(v.min()) >= 1.0 and float(v.median()) < 1.1
    S = 512
    nr = S * n
    ref = oracle.vif_groups([c[:nr].cpu().numpy() for c in x_cols], offs[:S + 1].cpu().numpy())
    _assert_vif_match(out[:S].cpu().numpy(), ref, p, "vif device sample")


def test_entry_points_are_reentrant_across_threads(pkg):
    """DuckDB calls Update/Finalize from its worker pool (SURVEY.md §8b 'Threading'): the single-group symbols use a
    per-thread default context and must be safe to call concurrently; so must batch calls on separate contexts."""
    import threading
    rng = np.random.default_rng(8)
    problems = []
    for k in range(16):
        n = 30 + 5 * k
        X = rng.uniform(-5, 5, (n, 3))
        y = 0.5 * k + X @ np.array([1.0, -2.0, 0.25 * k]) + 0.01 * rng.standard_normal(n)
        problems.append((y.tolist(), [X[:, j].tolist() for j in range(3)]))
    expected = [oracle.fit(p[0], p[1])[1] for p in problems]
    errors = []

    def worker(tid):
        try:
            ctx = pkg.Context()
            for rep in range(5):
                for k in range(tid, len(problems), 4):
                    r = pkg.ols_fit(problems[k][0], problems[k][1], {"compute_inference": True})
                    assert np.allclose(r["coefficients"], expected[k]["coefficients"], rtol=1e-9)
                    off, yy, xc, _ = _random_groups(np.random.default_rng(100 * tid + rep), 8, 2, 10, 40)
                    core, _ = pkg.fit_batch_host(off, yy, xc, None, pkg.RegressionOptions().batch_options("ols"), ctx=ctx)
                    ref, _ = oracle.fit_groups(yy, xc, off)
                    assert np.allclose(core[:, :3], ref[:, :3], rtol=1e-9)
            ctx.close()
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


# --------------------------------------------------------------------------------------------------
# residual diagnostics: residuals_diagnostics_agg / residuals_diagnostics / anofox_compute_residuals
# --------------------------------------------------------------------------------------------------
def _assert_residuals_match(out, grp, rout, rgrp, what):
    assert np.array_equal(grp, rgrp), what
    assert np.array_equal(np.isnan(out), np.isnan(rout)), what
    m = ~np.isnan(rout)
    assert np.array_equal(out[:, 0][m[:, 0]], rout[:, 0][m[:, 0]]), what          # raw = y - yhat: same subtraction
    inf = np.isinf(rout)                                                          # s = 0: studentized = raw / 0
    assert np.array_equal(out[inf], rout[inf]), what
    m &= ~inf
    scale = np.maximum(np.abs(rout[m]), 1.0)
    assert np.max(np.abs(out[m] - rout[m]) / scale, initial=0.0) < 1e-9, what


@pytest.mark.parametrize("p", [0, 1, 3, 8, 9, 16, 32, 33, 64, 128])
def test_residuals_batch_matches_oracle(pkg, ctx, p):
    """Tolerance 1e-9 relative on every part (the stated bar is 1e-6 for diagnostics).  p <= 8: residuals_narrow.hip,
    9 .. 128: residuals_wide.hip."""
    rng = np.random.default_rng(4100 + p)
    G = 120 if p <= 32 else 24
    offs, y, x_cols, _ = _random_groups(rng, G, p, 2 * p + 4, 300, offset=25.0)
    ns = np.diff(offs)
    N = len(y)
    y_hat = y + 0.5 * rng.standard_normal(N)
    y = y.copy()
    y[rng.random(N) < 0.05] = np.nan                                  # skipped rows, also first rows of groups
    y_hat[rng.random(N) < 0.03] = np.nan
    rse = rng.uniform(0.2, 2.0, G)
    rse[::7] = np.nan                                                 # no residual standard error for these groups
    rse[3] = 0.0                                                      # s <= 0: standardized = raw (residuals.rs:56-60)
    rse[5] = -1.0
    for drop in (True, False):
        for stud in (True, False):
            for r in (rse, None):
                out, grp = pkg.residuals_batch_host(offs, y, y_hat, x_cols, r, include_studentized=stud, drop_nan_rows=drop, ctx=ctx)
                rout, rgrp = oracle.residuals_groups(y, y_hat, x_cols, offs, rse=r, include_studentized=stud, drop_nan_rows=drop)
                _assert_residuals_match(out, grp, rout, rgrp, f"residuals p={p} drop={drop} stud={stud} rse={r is not None}")
    if p:
        flags = grp[:, 1].astype(int)
        assert np.all(flags & 4 == (4 if stud else 0)) or stud         # every group of this data has full rank
        out, grp = pkg.residuals_batch_host(offs, y, y_hat, x_cols, rse, ctx=ctx)
        gid = np.repeat(np.arange(G), ns)
        used = ~np.isnan(out[:, 0])
        hsum = np.bincount(gid[used], weights=out[used, 3], minlength=G)
        assert np.allclose(hsum, p + 1, rtol=0, atol=1e-8)              # trace of the hat matrix = p + 1
        assert np.all(out[used, 3] > 0) and np.all(out[used, 3] < 1 + 1e-12)


def test_residuals_rank_deficient_empty_and_poisoned_groups(pkg, ctx):
    x1 = np.arange(1.0, 11.0)
    rng = np.random.default_rng(8)
    blocks = [
        (x1, 2.0 * x1),                                   # exactly collinear: no leverage here and upstream
        (x1, np.full(10, 4.0)),                           # constant column: collinear with the intercept
        (x1, rng.standard_normal(10)),                    # full rank
        (np.array([1.0, 2.0]), np.array([0.5, -1.0])),    # fewer rows than parameters
        (np.empty(0), np.empty(0)),                       # empty group
        (x1, np.where(np.arange(10) == 4, np.nan, rng.standard_normal(10))),   # NaN feature value in a used row
    ]
    ns = [len(b[0]) for b in blocks]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    x_cols = [np.concatenate([b[j] for b in blocks]) for j in range(2)]
    N = int(offs[-1])
    y = rng.standard_normal(N)
    y_hat = y + 0.1 * rng.standard_normal(N)
    rse = np.full(len(blocks), 0.1)
    out, grp = pkg.residuals_batch_host(offs, y, y_hat, x_cols, rse, ctx=ctx)
    rout, rgrp = oracle.residuals_groups(y, y_hat, x_cols, offs, rse=rse)
    assert list(grp[:, 0]) == ns
    assert list(grp[:, 1].astype(int)) == [1, 1, 7, 1, 1, 7]
    # upstream agrees wherever its elimination meets an exact zero (integer data); the 2-row group is one of the
    # cases where it would divide by rounding noise instead (residuals_narrow.hip header)
    assert list(rgrp[[0, 1, 2, 4, 5], 1].astype(int)) == [1, 1, 7, 1, 7]
    lo, hi = offs[2], offs[3]
    assert np.allclose(out[lo:hi], rout[lo:hi], rtol=1e-10, atol=0)
    lo, hi = offs[5], offs[6]
    assert np.all(np.isnan(out[lo:hi, 3])) and np.all(np.isnan(rout[lo:hi, 3]))     # poisoned leverage, flag still set
    assert np.allclose(out[lo:hi, 2], rout[lo:hi, 2], rtol=1e-12)                   # max(1 - NaN, 1e-10) = 1e-10
    assert np.array_equal(out[:, :2], rout[:, :2])


@pytest.mark.parametrize("p", [12, 40, 100])
def test_residuals_mid_width_rank_deficient_and_poisoned_groups(pkg, ctx, p):
    """9 .. 128 features (residuals_wide.hip, one workgroup per group): collinear and
    constant columns give no leverage, a NaN feature value in a used row poisons the group's leverage, empty and tiny
    groups, everything else against the oracle."""
    rng = np.random.default_rng(12)
    big = max(60, p + 20)
    ns = [big, big, big, 5, 0, big, 5 * big]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    N = int(offs[-1])
    X = rng.uniform(-3, 3, (N, p)) + 10.0
    X[offs[0]:offs[1], 7] = 2.0 * X[offs[0]:offs[1], 2] - 1.0          # exactly collinear
    X[offs[1]:offs[2], 4] = 4.0                                          # constant: collinear with the intercept
    X[offs[5] + 9, 3] = np.nan                                           # NaN feature value in a used row
    x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    y = rng.standard_normal(N)
    y_hat = y + 0.1 * rng.standard_normal(N)
    y[offs[6] + 5] = np.nan                                              # a skipped row
    rse = np.full(len(ns), 0.1)
    out, grp = pkg.residuals_batch_host(offs, y, y_hat, x_cols, rse, ctx=ctx)
    rout, rgrp = oracle.residuals_groups(y, y_hat, x_cols, offs, rse=rse)
    assert list(grp[:, 0]) == [big, big, big, 5, 0, big, 5 * big - 1]
    assert list(grp[:, 1].astype(int)) == [1, 1, 7, 1, 1, 7, 7]
    for g in (2, 6):
        lo, hi = offs[g], offs[g + 1]
        assert np.allclose(out[lo:hi], rout[lo:hi], rtol=1e-9, atol=0, equal_nan=True)
    lo, hi = offs[5], offs[6]
    assert np.all(np.isnan(out[lo:hi, 3]))                               # poisoned leverage, flag still set
    assert np.array_equal(np.nan_to_num(out[:, :2], nan=-7.0), np.nan_to_num(rout[:, :2], nan=-7.0))
    with pytest.raises(pkg.AnofoxStatsError):
        pkg.residuals_batch_host(offs, y, y_hat, [x_cols[0]] * 129, rse, ctx=ctx)


def test_residuals_reference_structural_tests(pkg, ctx):
    """test/sql/diagnostics/test_residuals_diagnostics_agg.test (reg_data, reg_data_with_x), the scalar form of
    test/sql/scalar/test_diagnostics_scalar.test:126-140 and the unit tests of residuals.rs:204-263."""
    ya = [5.2, 9.8, 15.1, 20.0, 24.9, 30.2, 35.0, 39.8, 45.1, 50.0]
    yp = [5.0, 10.0, 15.0, 20.0, 25.0, 30.0, 35.0, 40.0, 45.0, 50.0]
    keys = np.zeros(10, dtype=np.int64)
    _, res = pkg.residuals_diagnostics_agg(keys, ya, yp, context=ctx)
    assert res[0]["raw"] == [a - b for a, b in zip(ya, yp)]
    assert res[0]["standardized"] is None and res[0]["studentized"] is None and res[0]["leverage"] is None
    x = [[float(i), 2.0 * i] for i in range(1, 11)]                  # x2 = 2 x1: singular design -> leverage NULL
    _, res = pkg.SQL_FUNCTIONS["anofox_stats_residuals_diagnostics_agg"](keys, ya, yp, x, context=ctx)
    assert res[0]["raw"] == [a - b for a, b in zip(ya, yp)] and res[0]["leverage"] is None
    x = [[float(i), float((i * 7) % 5)] for i in range(1, 11)]
    ya2 = list(ya)
    ya2[3] = None                                                    # NULL y: the row is skipped
    _, res = pkg.residuals_diagnostics_agg(keys, ya2, yp, x, context=ctx)
    assert len(res[0]["raw"]) == 9 and abs(sum(res[0]["leverage"]) - 3.0) < 1e-10
    _, res = pkg.residuals_diagnostics_agg([0, 0, 1, 1, 1], ya[:5], yp[:5], context=ctx)
    assert res[0] is None and len(res[1]["raw"]) == 3                # fewer than 3 rows -> NULL (:223)
    # scalar: residuals.rs unit tests
    y = [1.0, 2.0, 3.0, 4.0, 5.0]
    yh = [1.1, 1.9, 3.0, 4.1, 4.9]
    r = pkg.residuals_diagnostics(y, yh)
    assert max(abs(a - b) for a, b in zip(r["raw"], [-0.1, 0.1, 0.0, -0.1, 0.1])) < 1e-10
    assert r["standardized"] is None and r["studentized"] is None and r["leverage"] is None
    r = pkg.residuals_diagnostics(y, y, None, 0.1, False)
    assert r["standardized"] == [0.0] * 5
    r = pkg.SQL_FUNCTIONS["residuals_diagnostics"](y, yh, [[1.0, 2.0, 3.0, 4.0, 5.0]], 0.1, True)
    assert np.allclose(r["leverage"], [0.6, 0.3, 0.2, 0.3, 0.6], rtol=1e-13)          # textbook hat values of x = 1..5
    assert np.allclose(r["studentized"], np.array(r["raw"]) / (0.1 * np.sqrt(1 - np.array(r["leverage"]))), rtol=1e-13)
    assert pkg.residuals_diagnostics([1.0, 2.0], [1.0, 2.0]) is None                 # fewer than 3 values (:85-88)
    assert pkg.residuals_diagnostics(y, yh[:4]) is None
    r = pkg.residuals_diagnostics([1.0, float("nan"), 3.0], [1.0, 2.0, 2.5])          # the scalar form keeps NaN rows
    assert np.isnan(r["raw"][1]) and r["raw"][2] == 0.5


def test_compute_residuals_c_symbol_errors(pkg):
    import ctypes as C
    abi = import_pkg("_abi")
    lib = abi.load()
    arr = (C.c_double * 3)(1.0, 2.0, 3.0)
    a3 = abi.AnofoxDataArray(arr, None, 3)
    a2 = abi.AnofoxDataArray(arr, None, 2)
    a0 = abi.AnofoxDataArray(arr, None, 0)
    res = abi.AnofoxResidualsResult()
    err = abi.AnofoxError()
    assert not lib.anofox_compute_residuals(a0, a0, None, 0, float("nan"), False, C.byref(res), C.byref(err))
    assert err.code == 1 and "Empty y array" in err.text()                           # InvalidInput (residuals.rs:39-41)
    assert not lib.anofox_compute_residuals(a3, a2, None, 0, float("nan"), False, C.byref(res), C.byref(err))
    assert err.code == 9 and "y has 3 elements, y_hat has 2" in err.text()           # DimensionMismatch (:43-49)
    assert not lib.anofox_compute_residuals(a3, a3, None, 0, float("nan"), False, None, C.byref(err))
    assert lib.anofox_compute_residuals(a3, a3, None, 0, 2.0, True, C.byref(res), C.byref(err))
    assert res.len == 3 and res.has_standardized and not res.has_studentized and not res.has_leverage
    assert [res.raw[i] for i in range(3)] == [0.0, 0.0, 0.0]
    lib.anofox_free_residuals(C.byref(res))
    assert not res.raw and res.len == 0
    xs = (abi.AnofoxDataArray * 129)(*[a3] * 129)
    assert not lib.anofox_compute_residuals(a3, a3, xs, 129, 1.0, True, C.byref(res), C.byref(err))
    assert err.code == 1 and "maximum of 128" in err.text()
    xs = (abi.AnofoxDataArray * 9)(*[a3] * 9)          # nine identical columns: accepted, rank deficient -> no leverage
    assert lib.anofox_compute_residuals(a3, a3, xs, 9, 1.0, True, C.byref(res), C.byref(err))
    assert res.len == 3 and not res.has_leverage
    lib.anofox_free_residuals(C.byref(res))


def test_residuals_device_full_size_properties(pkg, ctx):
    """1M-row device-resident batch: per-group trace of the hat matrix = p + 1, raw = y - yhat exactly, and a sample of
    groups against the oracle."""
    import torch
    G, n, p = 10_000, 100, 8
    gen = torch.Generator(device="cuda").manual_seed(11)
    y = torch.randn(G * n, dtype=torch.float64, device="cuda", generator=gen)
    y_hat = y + 0.1 * torch.randn(G * n, dtype=torch.float64, device="cuda", generator=gen)
    x_cols = [torch.randn(G * n, dtype=torch.float64, device="cuda", generator=gen) * (j + 1) + 10.0 * j for j in range(p)]
    offs = torch.arange(0, G * n + 1, n, dtype=torch.int64, device="cuda")
    rse = torch.full((G,), 0.1, dtype=torch.float64, device="cuda")
    out, grp = ctx.residuals_batch_device(offs, y, y_hat, x_cols, rse)
    torch.cuda.synchronize()
    assert torch.equal(out[:, 0], y - y_hat)
    assert torch.all(grp[:, 0] == n) and torch.all(grp[:, 1] == 7)
    assert torch.allclose(out[:, 3].view(G, n).sum(1), torch.full((G,), p + 1.0, dtype=torch.float64, device="cuda"), rtol=0, atol=1e-9)
    S = 50
    rout, rgrp = oracle.residuals_groups(y[:S * n].cpu().numpy(), y_hat[:S * n].cpu().numpy(),
                                         [c[:S * n].cpu().numpy() for c in x_cols], offs[:S + 1].cpu().numpy(),
                                         rse=rse[:S].cpu().numpy())
    _assert_residuals_match(out[:S * n].cpu().numpy(), grp[:S].cpu().numpy(), rout, rgrp, "device sample")


# --------------------------------------------------------------------------------------------------
# information criteria as batched outputs of the fit records
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,p,icpt", [("ols", 3, True), ("ols", 8, False), ("wls", 5, True), ("ridge", 12, True), ("ols", 40, True)])
def test_information_criteria_batch_matches_reference_formula(pkg, ctx, model, p, icpt):
    """out[g] = {rss, aic, bic}: rss against the residuals of the oracle's coefficients summed in numpy, aic / bic
    against information_criteria.rs:15-33,67-85 (oracle.aic / oracle.bic) on (rss, n, k)."""
    rng = np.random.default_rng(31 * p + len(model))
    G = 40
    offs, y, x_cols, w = _random_groups(rng, G, p, p + 3, 300)
    # degenerate groups: a constant column (k shrinks), a one-row group (NULL), an exact fit (rss -> 0)
    x_cols[0][offs[1]:offs[2]] = 2.5
    lo, hi = offs[3], offs[4]
    y[lo:hi] = (1.0 if icpt else 0.0) + sum((j + 1) * x_cols[j][lo:hi] for j in range(p))
    kw = dict(fit_intercept=icpt)
    if model == "ridge":
        kw["alpha"] = 0.3
    wv = w if model == "wls" else None
    opts = _opts(pkg, model, **kw)
    core, _ = pkg.fit_batch_host(offs, y, x_cols, wv, opts, ctx=ctx)
    out = pkg.information_criteria_host(core, opts, ctx=ctx)
    rcore, _ = oracle.fit_groups(y, x_cols, offs, w=wv, **_oracle_kw(model, kw))
    X = np.stack(x_cols, axis=1)
    for g in range(G):
        if rcore[g, p + 5] != 0:
            assert np.all(np.isnan(out[g]))
            continue
        sl = slice(offs[g], offs[g + 1])
        b = np.nan_to_num(rcore[g, :p], nan=0.0)
        res = y[sl] - (rcore[g, p] if icpt else 0.0) - X[sl] @ b
        rss = float(np.sum((w[sl] if model == "wls" else 1.0) * res * res))
        k = int(np.sum(~np.isnan(rcore[g, :p]))) + int(icpt)
        n = int(rcore[g, p + 4])
        if n == k:
            assert np.all(np.isnan(out[g]))
            continue
        if g == 3 and model != "ridge":      # exact fit: rss is rounding noise, the criteria are hugely negative (or -inf)
            assert out[g, 0] < 1e-18 * np.sum(y[sl] ** 2) and out[g, 1] < -20 * n
            continue
        assert abs(out[g, 0] / rss - 1.0) < DIAG_RTOL, (g, out[g, 0], rss)
        assert abs(out[g, 1] - oracle.aic(rss, n, k)[1]) <= 1e-6 * max(1.0, abs(out[g, 1]))
        assert abs(out[g, 2] - oracle.bic(rss, n, k)[1]) <= 1e-6 * max(1.0, abs(out[g, 2]))
    # device entry point: same numbers
    import torch
    d = ctx.information_criteria_device(torch.from_numpy(core).cuda(), opts)
    torch.cuda.synchronize()
    assert np.array_equal(np.nan_to_num(d.cpu().numpy(), nan=-7.0), np.nan_to_num(out, nan=-7.0))


@pytest.mark.parametrize("p,kind", [(4, "copy"), (6, "dummy"), (12, "copy"), (20, "dummy"), (40, "copy"), (70, "dummy")])
def test_exactly_aliased_columns_are_not_queued(pkg, ctx, p, kind):
    """(r4) A solve that drops a non-constant column queues the group for the double-double refit only when the pivot is ABOVE the
    rounding noise of the moments (1e-13 .. 1e-11 of the diagonal: the band in which the reference's rule may keep the column).
    Exact copies (x_b = 2 x_a) and dummy-variable traps (indicator columns that sum to the intercept) — every group of many real
    workloads — must stay on the fast path: NaN for the later column as the reference's rule gives it, and (almost) nothing queued."""
    rng = np.random.default_rng(1000 + p)
    G, n = 4000, 120 + 3 * p
    offs = (np.arange(G + 1) * n).astype(np.int64)
    N = G * n
    X = rng.standard_normal((N, p)) * 10.0 ** rng.uniform(-1, 2, p) + rng.choice([0.0, 5.0, 300.0], p)
    if kind == "copy":
        a, b = 1, p - 1
        X[:, b] = 2.0 * X[:, a]
        dropped = b
    else:                                                           # three indicator columns that sum to one: the last is aliased to the intercept
        lvl = rng.integers(0, 3, N)
        for k in range(3):
            X[:, k] = (lvl == k).astype(np.float64)
        dropped = 2
    beta = rng.uniform(-2, 2, p)
    y = X @ beta + 3.0 + rng.standard_normal(N)
    x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    core, inf = _host_fit(pkg, ctx, "ols", offs, y, x_cols, None, compute_inference=True)
    queued = ctx.last_refine_count()
    assert queued <= G // 100, f"{queued} of {G} groups queued"
    assert np.all(core[:, p + 5] == 0) and np.all(np.isnan(core[:, dropped])) and np.all(np.isnan(inf[:, dropped]))
    keep = [j for j in range(p) if j != dropped]
    assert not np.isnan(core[:, keep]).any()
    S = 64                                                          # a sample against the oracle
    rcore, rinf = oracle.fit_groups(y[:S * n], [c[:S * n] for c in x_cols], offs[:S + 1], model="ols", compute_inference=True)
    assert_records_match(core[:S], rcore, p, inf[:S], rinf, what=f"aliased {kind} p={p}")
