"""pytest configuration: marker registration and shared fixture loaders."""
import csv
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-process GPU tests fork their ranks from a fork server that is started HERE, before anything in this
    # process has touched the GPU: a process that has initialised HIP must never exec (or hand a forked copy of
    # itself to) another program, and a rank forked from the clean server initialises the GPU on its own.
    import multiprocessing as mp
    from multiprocessing import forkserver
    try:
        mp.set_forkserver_preload([])
        forkserver.ensure_running()
    except Exception:        # pragma: no cover  (platforms without a fork server: the tests that need it skip)
        pass


def release_device_memory():
    """Hand every block torch's caching allocator holds back to the device.  The full-size tests (cfg3: 80 GB, cfg5:
    213 GB) size themselves from `torch.cuda.mem_get_info()`, which counts cached blocks of EARLIER tests as used: round 3's
    full suite skipped cfg5 at its BASELINE size for that reason alone."""
    import gc
    gc.collect()
    try:
        import torch
    except Exception:        # pragma: no cover
        return
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first(request):
    """torch's lazy CUDA initialisation fails ("No HIP GPUs are available") when it happens AFTER the DuckDB-glue test driver
    (libanofox_glue_capi.so and its worker threads) has used the device in the same process — seen when tests/test_gpu_glue.py
    runs before the first torch-using GPU test (the full suite's alphabetical order initialises torch earlier and never showed
    it).  Initialise it once, up front, whenever GPU tests are selected (after the fork server of pytest_configure)."""
    if any(item.get_closest_marker("gpu") is not None for item in request.session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:        # pragma: no cover
            pass
    yield


@pytest.fixture(autouse=True)
def _free_gpu_blocks_after_gpu_tests(request):
    """After a GPU test that left more than 4 GiB in torch's caching allocator: drop the references the test frame held and
    empty the cache (the full-size tests call release_device_memory() themselves before sizing).  Doing it after EVERY test
    cost 0.15 s a test — a 30 x slower deep fuzz sweep."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        try:
            import torch
            if torch.cuda.is_available() and torch.cuda.is_initialized() and torch.cuda.memory_reserved() > (4 << 30):
                release_device_memory()
        except Exception:        # pragma: no cover
            pass


def load_csv(rel):
    """Return dict column-name -> float64 array for a fixture CSV under tests/golden."""
    with open(os.path.join(GOLDEN, rel), newline="") as f:
        rows = list(csv.reader(f))
    names = rows[0]
    data = np.array([[float(v) for v in r] for r in rows[1:]], dtype=np.float64)
    return {n: np.ascontiguousarray(data[:, i]) for i, n in enumerate(names)}


def load_json(rel):
    with open(os.path.join(GOLDEN, rel)) as f:
        return json.load(f)


def nan_or(v):
    return np.nan if v in ("NA", None) else float(v)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


@pytest.fixture(scope="session")
def known_answers():
    return load_json("known_answers.json")


def import_pkg(sub: str = ""):
    """Import the product package (its directory name has a hyphen, so go through importlib)."""
    import importlib
    return importlib.import_module("anofox-statistics_amd" + (("." + sub) if sub else ""))


COEF_RTOL = 1e-9      # north_star: coefficients within 1e-9 relative
DIAG_RTOL = 1e-6      # north_star: diagnostic statistics within 1e-6


def assert_records_match(core, ref_core, p, inf=None, ref_inf=None, coef_rtol=COEF_RTOL, diag_rtol=DIAG_RTOL,
                         what="", skip_diag_groups=(), xbar=None):
    """Compare (core, inference) records of the HIP path with the oracle's, group by group.

    Coefficients: |got - ref| <= coef_rtol * max(|ref_j|, 1e-3 * max_k |ref_k|)  — strict relative error for
    every coefficient that is not negligible against the group's largest one, normwise for the rest
    (the achievable absolute error of any least-squares solver scales with ||beta||, not with |beta_j|).
    Diagnostics: relative diag_rtol (absolute for values that are exactly 0).
    NaN patterns and status words must agree exactly.
    `xbar` [G, p] (optional): the groups' feature means.  The intercept is ybar - sum_j xbar_j beta_j, so a coefficient
    difference within its tolerance moves it by up to sum_j |xbar_j| tol_j — far above 1e-9 |intercept| when the
    intercept is the small remainder of large cancelling terms; with `xbar` the intercept may differ by that much more.
    `skip_diag_groups`: groups whose diagnostics are ratios of rounding noise (zero residual degrees of
    freedom: RSS/0 is +inf or NaN depending on whether RSS rounds to exactly 0) — only coefficients are compared.
    """
    core = np.asarray(core)
    ref_core = np.asarray(ref_core)
    assert core.shape == ref_core.shape, what
    st, rst = core[:, p + 5], ref_core[:, p + 5]
    bad = np.nonzero(st != rst)[0]
    assert bad.size == 0, f"{what}: status differs at groups {bad[:10]}: {st[bad[:10]]} vs {rst[bad[:10]]}"
    ok = rst == 0
    c, rc = core[ok, :p], ref_core[ok, :p]
    assert np.array_equal(np.isnan(c), np.isnan(rc)), f"{what}: NaN pattern of coefficients differs"
    scale = np.nanmax(np.abs(np.concatenate([rc, ref_core[ok, p:p + 1]], axis=1)), axis=1, keepdims=True)
    scale = np.where(np.isfinite(scale), scale, 0.0)

    def chk_coef(g, r, name, extra=0.0):
        tol = coef_rtol * np.maximum(np.abs(r), 1e-3 * scale) + extra
        err = np.abs(g - r)
        m = ~np.isnan(r)
        worst = np.max((err[m] / np.maximum(tol[m], 1e-300))) if m.any() else 0.0
        assert worst <= 1.0, f"{what}: {name} off by {worst:.3g} x tolerance"

    chk_coef(c, rc, "coefficients")
    gi, ri = core[ok, p:p + 1], ref_core[ok, p:p + 1]
    assert np.array_equal(np.isnan(gi), np.isnan(ri)), f"{what}: intercept NaN pattern differs"
    extra = 0.0
    if xbar is not None:
        ctol = coef_rtol * np.maximum(np.abs(rc), 1e-3 * scale)
        extra = np.nansum(np.abs(np.asarray(xbar)[ok]) * np.where(np.isnan(rc), 0.0, ctol), axis=1, keepdims=True)
    chk_coef(gi, ri, "intercept", extra)

    def chk_diag(g, r, name, rtol=diag_rtol, stat=None, atol=0.0):
        """`stat` = (got, ref) of the statistic a tail probability was computed from: an extreme p-value amplifies the
        statistic's relative error by |d ln p / d ln t| <= 2 |ln p| + 2 (p = 1e-246 at t = 85.7, df = 371 turns a
        2.9e-9 difference in t — itself held to `rtol` — into 1e-6), so the p-value may differ by that much more."""
        g = np.asarray(g, dtype=np.float64)
        r = np.asarray(r, dtype=np.float64)
        assert np.array_equal(np.isnan(g), np.isnan(r)), f"{what}: NaN pattern of {name} differs"
        m = ~np.isnan(r)
        fin = m & np.isfinite(r)
        assert np.array_equal(g[m & ~fin], r[m & ~fin]), f"{what}: infinities of {name} differ"
        err = np.abs(g[fin] - r[fin])
        rel = np.full(err.shape, rtol)
        if stat is not None:
            sg, sr = np.asarray(stat[0], dtype=np.float64)[fin], np.asarray(stat[1], dtype=np.float64)[fin]
            with np.errstate(all="ignore"):
                rel_stat = np.where(np.isfinite(sr) & (sr != 0), np.abs(sg - sr) / np.abs(sr), 0.0)
                amp = 2.0 * np.abs(np.log(np.maximum(np.abs(r[fin]), 1e-320))) + 2.0
            rel = np.maximum(rel, amp * rel_stat)
        tol = rel * np.abs(r[fin]) + (np.asarray(atol)[fin] if np.ndim(atol) else atol) + 1e-300
        worst = np.max(err / tol) if err.size else 0.0
        assert worst <= 1.0, f"{what}: {name} off by {worst:.3g} x tolerance"

    okd = ok.copy()
    okd[list(skip_diag_groups)] = False
    for k, name in ((1, "r_squared"), (2, "adj_r_squared"), (3, "residual_std_error"), (4, "n_observations")):
        # r^2 and adjusted r^2 are 1 - (a ratio): their rounding error is absolute (~1e-15 x conditioning), so a value
        # that is itself ~1e-10 (no signal) cannot be held to a relative 1e-6
        chk_diag(core[okd, p + k], ref_core[okd, p + k], name, rtol=(0.0 if k == 4 else diag_rtol),
                 atol=(1e-12 if k in (1, 2) else 0.0))
    if ref_inf is not None:
        inf = np.asarray(inf)
        ref_inf = np.asarray(ref_inf)
        names = ["std_errors", "t_values", "p_values", "ci_lower", "ci_upper"]
        # an interval bound b -+ t se that happens to fall near zero is a difference of two larger numbers: its error
        # is measured against the larger of the two bounds' magnitudes
        ci_scale = np.maximum(np.abs(ref_inf[okd, 3 * p:4 * p]), np.abs(ref_inf[okd, 4 * p:5 * p]))
        ci_scale = np.where(np.isfinite(ci_scale), ci_scale, 0.0)
        for k, name in enumerate(names):
            stat = (inf[okd, p:2 * p], ref_inf[okd, p:2 * p]) if name == "p_values" else None
            atol = diag_rtol * ci_scale if name in ("ci_lower", "ci_upper") else 0.0
            chk_diag(inf[okd, k * p:(k + 1) * p], ref_inf[okd, k * p:(k + 1) * p], name, stat=stat, atol=atol)
        # F = (TSS - RSS) / dfm / (RSS / df): with no signal at all TSS - RSS cancels (same absolute floor as r^2, x df/dfm)
        chk_diag(inf[okd, 5 * p], ref_inf[okd, 5 * p], "f_statistic", atol=1e-9)
        chk_diag(inf[okd, 5 * p + 1], ref_inf[okd, 5 * p + 1], "f_pvalue", stat=(inf[okd, 5 * p], ref_inf[okd, 5 * p]))
