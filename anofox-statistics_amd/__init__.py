"""anofox-statistics_amd — MI355X-native grouped least-squares (ols_fit_agg / ridge_fit_agg / wls_fit_agg).

The product is libanofox_stats_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/anofox_stats_hip.h).  This package is its host-side mirror of the reference's operator
interface: the aggregates (aggregate.py), the scalar functions (scalar.py), the options parser
(options.py), plus device-resident and multi-GPU drivers (runtime.py, distributed.py).

Importing the package loads the shared library and fails if it is missing — there is no CPU fallback.
"""
from . import _abi

_abi.load()

from ._abi import AnofoxStatsError  # noqa: E402
from .aggregate import (StreamingStates, FitAggResult, FitPredictAggResult, OlsFitAgg, OlsFitPredictAgg, RidgeFitAgg,  # noqa: E402
                        RidgeFitPredictAgg, WlsFitAgg, WlsFitPredictAgg, SQL_FUNCTIONS, ols_fit_agg,
                        ols_fit_predict_agg, ridge_fit_agg, ridge_fit_predict_agg, wls_fit_agg, wls_fit_predict_agg,
                        result_from_records, ols_fit_predict, ridge_fit_predict, wls_fit_predict, vif_agg,
                        residuals_diagnostics_agg)
from .options import InvalidInputException, RegressionOptions, parse_options  # noqa: E402
from .runtime import AggState, Context, information_criteria_host, fit_predict_frames_host, fit_batch_host, fit_predict_batch_host, fit_predict_expanding_host, fit_predict_window_host, vif_batch_host, residuals_batch_host  # noqa: E402
from .scalar import aic, bic, ols_fit, predict, predict_with_interval, ridge_fit, t_critical, vif, wls_fit, residuals_diagnostics  # noqa: E402

# the scalar functions under their SQL names (src/table_functions/{ols,ridge,wls}_fit.cpp, predict.cpp,
# src/scalar_functions/{aic_bic,vif}.cpp) and the deprecated aggregate aliases
SQL_FUNCTIONS.update({
    "anofox_stats_ols_fit": ols_fit, "ols_fit": ols_fit,
    "anofox_stats_ridge_fit": ridge_fit, "ridge_fit": ridge_fit,
    "anofox_stats_wls_fit": wls_fit, "wls_fit": wls_fit,
    "anofox_stats_predict": predict,
    "anofox_stats_aic": aic, "aic": aic, "anofox_stats_bic": bic, "bic": bic,
    "anofox_stats_vif": vif, "vif": vif,
    "anofox_stats_residuals_diagnostics": residuals_diagnostics, "residuals_diagnostics": residuals_diagnostics,
    "ridge_predict_agg": ridge_fit_predict_agg, "wls_predict_agg": wls_fit_predict_agg,
})

__all__ = [
    "AggState", "StreamingStates", "information_criteria_host", "fit_predict_frames_host", "AnofoxStatsError", "Context", "FitAggResult", "InvalidInputException", "OlsFitAgg", "RegressionOptions",
    "RidgeFitAgg", "SQL_FUNCTIONS", "WlsFitAgg", "aic", "bic", "fit_batch_host", "ols_fit", "ols_fit_agg",
    "parse_options", "result_from_records", "ridge_fit", "ridge_fit_agg", "wls_fit", "wls_fit_agg",
    "FitPredictAggResult", "OlsFitPredictAgg", "RidgeFitPredictAgg", "WlsFitPredictAgg", "fit_predict_batch_host",
    "ols_fit_predict_agg", "ridge_fit_predict_agg", "wls_fit_predict_agg", "predict", "predict_with_interval",
    "t_critical", "fit_predict_expanding_host", "fit_predict_window_host", "ols_fit_predict", "ridge_fit_predict", "wls_fit_predict",
    "vif", "vif_agg", "vif_batch_host", "residuals_diagnostics", "residuals_diagnostics_agg", "residuals_batch_host",
]


def version() -> str:
    return _abi.load().anofox_hip_version().decode()
