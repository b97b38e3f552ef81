"""Scalar `anofox_stats_{ols,ridge,wls}_fit(y LIST, x LIST(LIST) [, w] [, options])` — mirror of
src/table_functions/{ols,ridge,wls}_fit.cpp (one group per call, x column-major, failures THROW,
ols_fit.cpp:176-178), bound to the reference-compatible C symbols of libanofox_stats_hip.so
(anofox_ols_fit / anofox_ridge_fit / anofox_wls_fit / anofox_free_result_*)."""
from __future__ import annotations

import ctypes as C
from typing import Any, Mapping, Optional, Sequence

import numpy as np

from . import _abi
from .options import InvalidInputException, parse_options

_DP = C.POINTER(C.c_double)


def _data_array(v):
    """(AnofoxDataArray, keepalive): None entries become NULLs through the validity bitmask."""
    vals = list(v)
    nulls = [x is None for x in vals]
    data = np.array([0.0 if x is None else float(x) for x in vals], dtype=np.float64)
    arr = _abi.AnofoxDataArray()
    arr.data = data.ctypes.data_as(_DP)
    arr.len = len(vals)
    keep = [data]
    if any(nulls):
        bits = np.zeros((len(vals) + 7) // 8, dtype=np.uint8)
        for i, isnull in enumerate(nulls):
            if not isnull:
                bits[i // 8] |= 1 << (i % 8)
        arr.validity = bits.ctypes.data_as(C.POINTER(C.c_uint8))
        keep.append(bits)
    else:
        arr.validity = None
    return arr, keep


def _call(model: str, y, x: Sequence[Sequence[float]], weights, options: Optional[Mapping[str, Any]]) -> dict:
    lib = _abi.load()
    o = parse_options(options)
    ya, k0 = _data_array(y)
    xs = (_abi.AnofoxDataArray * max(len(x), 1))()
    keep = [k0]
    for j, col in enumerate(x):
        a, k = _data_array(col)
        xs[j] = a
        keep.append(k)
    core = _abi.AnofoxFitResultCore()
    inf = _abi.AnofoxFitResultInference()
    err = _abi.AnofoxError()
    infp = C.byref(inf) if o.compute_inference else None
    if model == "ols":
        opt = _abi.AnofoxOlsOptions(o.fit_intercept, o.compute_inference, o.confidence_level, _abi.SOLVER[o.solver],
                                    _abi.HC_TYPE[o.hc_type])
        ok = lib.anofox_ols_fit(ya, xs, len(x), opt, C.byref(core), infp, C.byref(err))
        label = "OLS"
    elif model == "ridge":
        opt = _abi.AnofoxRidgeOptions(o.alpha, o.fit_intercept, o.compute_inference, o.confidence_level,
                                      _abi.SOLVER[o.solver], _abi.LAMBDA_SCALING[o.lambda_scaling])
        ok = lib.anofox_ridge_fit(ya, xs, len(x), opt, C.byref(core), infp, C.byref(err))
        label = "Ridge"
    else:
        wa, kw = _data_array(weights)
        keep.append(kw)
        opt = _abi.AnofoxWlsOptions(o.fit_intercept, o.compute_inference, o.confidence_level, _abi.SOLVER[o.solver],
                                    _abi.HC_TYPE[o.hc_type])
        ok = lib.anofox_wls_fit(ya, xs, len(x), wa, opt, C.byref(core), infp, C.byref(err))
        label = "WLS"
    if not ok:
        e = InvalidInputException(f"{label} fit failed: {err.text()}")
        e.code = err.code
        raise e
    try:
        p = core.coefficients_len
        out = {"coefficients": [core.coefficients[i] for i in range(p)], "intercept": core.intercept,
               "r_squared": core.r_squared, "adj_r_squared": core.adj_r_squared,
               "residual_std_error": core.residual_std_error, "n_observations": core.n_observations,
               "n_features": core.n_features}
        if o.compute_inference:
            n = inf.len
            for name in ("std_errors", "t_values", "p_values", "ci_lower", "ci_upper"):
                ptr = getattr(inf, name)
                out[name] = [ptr[i] for i in range(n)] if n else None
            out["f_statistic"] = inf.f_statistic
            out["f_pvalue"] = inf.f_pvalue
        return out
    finally:
        lib.anofox_free_result_core(C.byref(core))
        if o.compute_inference:
            lib.anofox_free_result_inference(C.byref(inf))


def ols_fit(y, x, options=None) -> dict:
    """anofox_stats_ols_fit([y...], [[x1...], [x2...]], {...})"""
    return _call("ols", y, x, None, options)


def ridge_fit(y, x, options=None) -> dict:
    return _call("ridge", y, x, None, options)


def wls_fit(y, x, weights, options=None) -> dict:
    return _call("wls", y, x, weights, options)


def aic(rss: float, n: int, k: int) -> Optional[float]:
    """aic(rss, n, k) scalar function (src/scalar_functions/aic_bic.cpp:12-60): NULL (None) on error."""
    lib = _abi.load()
    out = C.c_double()
    err = _abi.AnofoxError()
    return out.value if lib.anofox_compute_aic(float(rss), int(n), int(k), C.byref(out), C.byref(err)) else None


def bic(rss: float, n: int, k: int) -> Optional[float]:
    lib = _abi.load()
    out = C.c_double()
    err = _abi.AnofoxError()
    return out.value if lib.anofox_compute_bic(float(rss), int(n), int(k), C.byref(out), C.byref(err)) else None


def t_critical(confidence_level: float, df: int) -> float:
    """anofox_t_critical (crates/anofox-stats-ffi/src/lib.rs:2217-2231)."""
    return float(_abi.load().anofox_t_critical(float(confidence_level), int(df)))


def predict_with_interval(coefficients, intercept, x_new, residual_std_error, n_observations,
                          confidence_level=0.95) -> Optional[dict]:
    """anofox_predict_with_interval (lib.rs:2264-2349); None when the call fails."""
    lib = _abi.load()
    c = np.ascontiguousarray(coefficients, dtype=np.float64)
    xn = np.ascontiguousarray(x_new, dtype=np.float64)
    out = _abi.AnofoxPredictionResult()
    ok = lib.anofox_predict_with_interval(c.ctypes.data_as(_DP), len(c), float(intercept), xn.ctypes.data_as(_DP),
                                          len(xn), float(residual_std_error), int(n_observations),
                                          float(confidence_level), C.byref(out))
    return {"yhat": out.yhat, "yhat_lower": out.yhat_lower, "yhat_upper": out.yhat_upper} if ok else None


def predict(x, coefficients, intercept=float("nan")):
    """anofox_stats_predict(x LIST(LIST), coefficients, intercept) (src/table_functions/predict.cpp); x is
    column-major.  Runs on the GPU."""
    lib = _abi.load()
    xs = (_abi.AnofoxDataArray * max(len(x), 1))()
    keep = []
    for j, col in enumerate(x):
        a, k = _data_array(col)
        xs[j] = a
        keep.append(k)
    c = np.ascontiguousarray(coefficients, dtype=np.float64)
    outp = _DP()
    outn = C.c_size_t()
    err = _abi.AnofoxError()
    if not lib.anofox_predict(xs, len(x), c.ctypes.data_as(_DP), len(c), float(intercept), C.byref(outp),
                              C.byref(outn), C.byref(err)):
        e = InvalidInputException(f"Prediction failed: {err.text()}")
        e.code = err.code
        raise e
    try:
        return [outp[i] for i in range(outn.value)]
    finally:
        lib.anofox_free_predictions(outp)


def vif(x):
    """anofox_stats_vif(x LIST(LIST(DOUBLE))) -> LIST(DOUBLE) (src/scalar_functions/vif.cpp:26-76): x is a list of
    feature COLUMNS; throws on failure like the SQL function.  Runs on the GPU."""
    lib = _abi.load()
    xs = (_abi.AnofoxDataArray * max(len(x), 1))()
    keep = []
    for j, col in enumerate(x):
        a, k = _data_array(col)
        xs[j] = a
        keep.append(k)
    outp = _DP()
    outn = C.c_size_t()
    err = _abi.AnofoxError()
    if not lib.anofox_compute_vif(xs, len(x), C.byref(outp), C.byref(outn), C.byref(err)):
        e = InvalidInputException(f"VIF computation failed: {err.text()}")
        e.code = err.code
        raise e
    try:
        return [outp[i] for i in range(outn.value)]
    finally:
        lib.anofox_free_vif(outp)


def residuals_diagnostics(y, y_hat, x=None, residual_std_error=None, include_studentized=True):
    """anofox_stats_residuals_diagnostics(y LIST, y_hat LIST[, x LIST(LIST), rse DOUBLE, include_studentized BOOL])
    -> STRUCT(raw, standardized, studentized, leverage) (src/scalar_functions/residuals_diagnostics.cpp:75-192): x is a
    list of feature COLUMNS; NULL for fewer than 3 values or lists of unequal length (:85-88) and when the C call
    fails (:158-161).  Runs on the GPU."""
    if len(y) < 3 or len(y) != len(y_hat):
        return None
    lib = _abi.load()
    ya, k1 = _data_array(y)
    ha, k2 = _data_array(y_hat)
    cols = list(x) if x is not None else []
    xs = (_abi.AnofoxDataArray * max(len(cols), 1))()
    keep = [k1, k2]
    for j, col in enumerate(cols):
        a, k = _data_array(col)
        xs[j] = a
        keep.append(k)
    res = _abi.AnofoxResidualsResult()
    err = _abi.AnofoxError()
    rse = float("nan") if residual_std_error is None else float(residual_std_error)
    if not lib.anofox_compute_residuals(ya, ha, xs if cols else None, len(cols), rse, bool(include_studentized),
                                        C.byref(res), C.byref(err)):
        return None
    try:
        n = res.len

        def take(ptr, has):
            return [ptr[i] for i in range(n)] if has and ptr else None
        return dict(raw=take(res.raw, True), standardized=take(res.standardized, res.has_standardized),
                    studentized=take(res.studentized, res.has_studentized), leverage=take(res.leverage, res.has_leverage))
    finally:
        lib.anofox_free_residuals(C.byref(res))
