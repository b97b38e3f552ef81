"""ctypes loader for the CPU oracle (oracle/anofox_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libanofox_oracle.so")

MODEL = {"ols": 0, "ridge": 1, "wls": 2}
STATUS_NULL_TOO_FEW_ROWS = 100


class OracleOptions(C.Structure):
    _fields_ = [
        ("model", C.c_int32),
        ("fit_intercept", C.c_int32),
        ("compute_inference", C.c_int32),
        ("lambda_scaling", C.c_int32),
        ("confidence_level", C.c_double),
        ("alpha", C.c_double),
        ("hc_type", C.c_int32),
        ("plain_qr", C.c_int32),
        ("plain_svd", C.c_int32),
    ]


_DP = C.POINTER(C.c_double)


class OracleResult(C.Structure):
    _fields_ = [
        ("coefficients", _DP),
        ("std_errors", _DP),
        ("t_values", _DP),
        ("p_values", _DP),
        ("ci_lower", _DP),
        ("ci_upper", _DP),
        ("intercept", C.c_double),
        ("r_squared", C.c_double),
        ("adj_r_squared", C.c_double),
        ("residual_std_error", C.c_double),
        ("f_statistic", C.c_double),
        ("f_pvalue", C.c_double),
        ("rss", C.c_double),
        ("tss", C.c_double),
        ("n_observations", C.c_int64),
        ("n_features", C.c_int64),
        ("rank", C.c_int32),
        ("has_inference", C.c_int32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "anofox_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_fit.restype = C.c_int
        L.oracle_fit.argtypes = [_DP, C.POINTER(_DP), _DP, C.c_size_t, C.c_size_t,
                                 C.POINTER(OracleOptions), C.POINTER(OracleResult)]
        L.oracle_fit_groups.restype = C.c_int
        L.oracle_fit_groups.argtypes = [_DP, C.POINTER(_DP), _DP, C.POINTER(C.c_int64), C.c_int64, C.c_size_t,
                                        C.POINTER(OracleOptions), _DP, _DP, C.c_int]
        for name in ("oracle_betainc",):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_double] * 3
        L.oracle_t_two_sided_p.restype = C.c_double
        L.oracle_t_two_sided_p.argtypes = [C.c_double, C.c_double]
        L.oracle_f_sf.restype = C.c_double
        L.oracle_f_sf.argtypes = [C.c_double] * 3
        L.oracle_t_quantile.restype = C.c_double
        L.oracle_t_quantile.argtypes = [C.c_double, C.c_double]
        L.oracle_t_critical.restype = C.c_double
        L.oracle_t_critical.argtypes = [C.c_double, C.c_int64]
        L.oracle_predict_with_interval.restype = C.c_int
        L.oracle_predict_with_interval.argtypes = [_DP, C.c_size_t, C.c_double, _DP, C.c_double, C.c_int64, C.c_double, _DP]
        L.oracle_fit_predict_groups.restype = C.c_int
        L.oracle_fit_predict_groups.argtypes = [_DP, C.POINTER(_DP), _DP, C.POINTER(C.c_int64), C.c_int64, C.c_size_t,
                                                C.POINTER(OracleOptions), C.POINTER(C.c_int64), _DP, _DP]
        L.oracle_fit_predict_expanding.restype = C.c_int
        L.oracle_fit_predict_expanding.argtypes = [_DP, C.POINTER(_DP), _DP, C.POINTER(C.c_int64), C.c_int64, C.c_size_t,
                                                   C.POINTER(OracleOptions), _DP]
        L.oracle_fit_predict_window.restype = C.c_int
        L.oracle_fit_predict_window.argtypes = [_DP, C.POINTER(_DP), _DP, C.POINTER(C.c_int64), C.c_int64, C.c_size_t,
                                                C.POINTER(OracleOptions), C.c_int64, C.c_int64, _DP]
        L.oracle_vif_groups.restype = C.c_int
        L.oracle_vif_groups.argtypes = [C.POINTER(_DP), C.POINTER(C.c_int64), C.c_int64, C.c_size_t, C.c_int64, _DP]
        L.oracle_residuals_groups.restype = C.c_int
        L.oracle_residuals_groups.argtypes = [_DP, _DP, C.POINTER(_DP), C.POINTER(C.c_int64), C.c_int64, C.c_size_t, _DP,
                                              C.c_int, C.c_int, _DP, _DP]
        for name in ("oracle_aic", "oracle_bic"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.c_double, C.c_int64, C.c_int64, _DP]
        _lib = L
    return _lib


def _opts(model="ols", fit_intercept=True, compute_inference=False, confidence_level=0.95, alpha=1.0,
          lambda_scaling="raw", hc_type="none", plain_qr=False, plain_svd=False) -> OracleOptions:
    """plain_qr=True stops after the QR solve, plain_svd=True solves through an SVD (QR, then one-sided Jacobi on the
    triangular factor) — the reference's algorithm classes as they are: `solver` qr and svd, the latter the aggregates'
    default (ols_aggregate.cpp:51); bench.py times both as CPU baselines.  The default adds the extended-precision
    refinement that makes the oracle the checker."""
    return OracleOptions(MODEL[model], int(bool(fit_intercept)), int(bool(compute_inference)),
                         {"raw": 0, "glmnet": 1}[lambda_scaling], float(confidence_level), float(alpha),
                         {"none": 0, "hc0": 1, "hc1": 2, "hc2": 3, "hc3": 4}[hc_type], int(bool(plain_qr)),
                         int(bool(plain_svd)))


def _col_ptrs(cols):
    arr = (_DP * len(cols))()
    for j, c in enumerate(cols):
        arr[j] = c.ctypes.data_as(_DP)
    return arr


def fit(y, x_cols, w=None, **kw):
    """One group.  ``x_cols`` = sequence of p arrays (one per feature).
    Returns (error_code, dict)."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    p = len(cols)
    wv = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    out = {k: np.full(p, np.nan) for k in ("coefficients", "std_errors", "t_values", "p_values", "ci_lower", "ci_upper")}
    res = OracleResult()
    for k, v in out.items():
        setattr(res, k, v.ctypes.data_as(_DP))
    o = _opts(**kw)
    code = lib().oracle_fit(y.ctypes.data_as(_DP), _col_ptrs(cols), None if wv is None else wv.ctypes.data_as(_DP),
                            len(y), p, C.byref(o), C.byref(res))
    d = dict(out)
    for k in ("intercept", "r_squared", "adj_r_squared", "residual_std_error", "f_statistic", "f_pvalue", "rss",
              "tss", "n_observations", "n_features", "rank", "has_inference"):
        d[k] = getattr(res, k)
    return code, d


def fit_groups(y, x_cols, offsets, w=None, n_threads=1, **kw):
    """Grouped fit, rows of group g = [offsets[g], offsets[g+1]).
    Returns (core[G, p+6], inf[G, 5p+2] or None) in the record layout of include/anofox_stats_hip.h."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    p = len(cols)
    G = len(offsets) - 1
    wv = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    o = _opts(**kw)
    core = np.empty((G, p + 6))
    inf = np.empty((G, 5 * p + 2)) if o.compute_inference else None
    rc = lib().oracle_fit_groups(y.ctypes.data_as(_DP), _col_ptrs(cols),
                                 None if wv is None else wv.ctypes.data_as(_DP),
                                 offsets.ctypes.data_as(C.POINTER(C.c_int64)), G, p, C.byref(o),
                                 core.ctypes.data_as(_DP), None if inf is None else inf.ctypes.data_as(_DP),
                                 int(n_threads))
    if rc != 0:
        raise RuntimeError(f"oracle_fit_groups failed: {rc}")
    return core, inf


def aic(rss, n, k):
    out = C.c_double()
    rc = lib().oracle_aic(rss, n, k, C.byref(out))
    return rc, out.value


def bic(rss, n, k):
    out = C.c_double()
    rc = lib().oracle_bic(rss, n, k, C.byref(out))
    return rc, out.value


def t_critical(confidence_level, df):
    return lib().oracle_t_critical(float(confidence_level), int(df))


def predict_with_interval(coef, intercept, x_new, rse, n_obs, confidence_level=0.95):
    c = np.ascontiguousarray(coef, dtype=np.float64)
    xn = np.ascontiguousarray(x_new, dtype=np.float64)
    out = np.empty(3)
    ok = lib().oracle_predict_with_interval(c.ctypes.data_as(_DP), len(c), float(intercept), xn.ctypes.data_as(_DP),
                                            float(rse), int(n_obs), float(confidence_level), out.ctypes.data_as(_DP))
    return bool(ok), out


def fit_predict_groups(y, x_cols, offsets, w=None, train_counts=None, **kw):
    """(core[G, p+6], pred[N, 3]) — y carries NaN at non-training rows."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    p, G = len(cols), len(offsets) - 1
    wv = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    tc = None if train_counts is None else np.ascontiguousarray(train_counts, dtype=np.int64)
    o = _opts(**kw)
    core = np.empty((G, p + 6))
    pred = np.empty((len(y), 3))
    rc = lib().oracle_fit_predict_groups(y.ctypes.data_as(_DP), _col_ptrs(cols),
                                         None if wv is None else wv.ctypes.data_as(_DP),
                                         offsets.ctypes.data_as(C.POINTER(C.c_int64)), G, p, C.byref(o),
                                         None if tc is None else tc.ctypes.data_as(C.POINTER(C.c_int64)),
                                         core.ctypes.data_as(_DP), pred.ctypes.data_as(_DP))
    if rc != 0:
        raise RuntimeError(f"oracle_fit_predict_groups failed: {rc}")
    return core, pred


def fit_predict_expanding(y, x_cols, offsets, w=None, **kw):
    """pred[N, 3]: prediction of x_e from the fit on rows 0..e of its partition (window functions)."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    wv = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    o = _opts(**kw)
    pred = np.empty((len(y), 3))
    rc = lib().oracle_fit_predict_expanding(y.ctypes.data_as(_DP), _col_ptrs(cols),
                                            None if wv is None else wv.ctypes.data_as(_DP),
                                            offsets.ctypes.data_as(C.POINTER(C.c_int64)), len(offsets) - 1, len(cols),
                                            C.byref(o), pred.ctypes.data_as(_DP))
    if rc != 0:
        raise RuntimeError(f"oracle_fit_predict_expanding failed: {rc}")
    return pred


def fit_predict_window(y, x_cols, offsets, w=None, start_preceding=None, end_preceding=0, **kw):
    """pred[N, 3] of the window functions over ROWS BETWEEN start_preceding PRECEDING AND end_preceding PRECEDING
    (negative = FOLLOWING; None = UNBOUNDED PRECEDING / FOLLOWING)."""
    start_preceding = 2 ** 63 - 1 if start_preceding is None else start_preceding
    end_preceding = -(2 ** 63 - 1) if end_preceding is None else end_preceding
    y = np.ascontiguousarray(y, dtype=np.float64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    wv = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    o = _opts(**kw)
    pred = np.empty((len(y), 3))
    rc = lib().oracle_fit_predict_window(y.ctypes.data_as(_DP), _col_ptrs(cols),
                                         None if wv is None else wv.ctypes.data_as(_DP),
                                         offsets.ctypes.data_as(C.POINTER(C.c_int64)), len(offsets) - 1, len(cols),
                                         C.byref(o), int(start_preceding), int(end_preceding), pred.ctypes.data_as(_DP))
    if rc != 0:
        raise RuntimeError(f"oracle_fit_predict_window failed: {rc}")
    return pred


def vif_groups(x_cols, offsets, min_rows=3):
    """out[G, p+1] = { vif[p], status } per group (status 100 = fewer than min_rows rows -> NULL)."""
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    p, G = len(cols), len(offsets) - 1
    out = np.empty((G, p + 1))
    rc = lib().oracle_vif_groups(_col_ptrs(cols), offsets.ctypes.data_as(C.POINTER(C.c_int64)), G, p, int(min_rows),
                                 out.ctypes.data_as(_DP))
    if rc != 0:
        raise RuntimeError(f"oracle_vif_groups failed: {rc}")
    return out


def residuals_groups(y, y_hat, x_cols, offsets, rse=None, include_studentized=True, drop_nan_rows=True):
    """(out[N, 4] = raw / standardized / studentized / leverage, group[G, 2] = rows used / flags); restates
    crates/anofox-stats-core/src/diagnostics/residuals.rs:30-145 per group."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    y_hat = np.ascontiguousarray(y_hat, dtype=np.float64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in (x_cols or [])]
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    p, G, N = len(cols), len(offsets) - 1, len(y)
    out = np.empty((N, 4))
    group = np.empty((G, 2))
    rse_arr = None if rse is None else np.ascontiguousarray(rse, dtype=np.float64)
    rc = lib().oracle_residuals_groups(y.ctypes.data_as(_DP), y_hat.ctypes.data_as(_DP), _col_ptrs(cols) if p else None,
                                       offsets.ctypes.data_as(C.POINTER(C.c_int64)), G, p,
                                       None if rse_arr is None else rse_arr.ctypes.data_as(_DP),
                                       int(bool(include_studentized)), int(bool(drop_nan_rows)),
                                       out.ctypes.data_as(_DP), group.ctypes.data_as(_DP))
    if rc != 0:
        raise RuntimeError(f"oracle_residuals_groups failed: {rc}")
    return out, group
