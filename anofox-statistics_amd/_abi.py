"""ctypes binding of libanofox_stats_hip.so (include/anofox_stats_hip.h).

The library is the product; this module only declares its C ABI.  There is no
fallback: if the shared object is missing or a symbol is absent, import fails.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ANOFOX_STATS_HIP_LIB selects another build of the same ABI (used for the diagnostic build with stamps)
LIB_PATH = os.environ.get("ANOFOX_STATS_HIP_LIB") or os.path.join(_HERE, "libanofox_stats_hip.so")

# --- enums (anofox_stats_hip.h) ------------------------------------------------
ERROR_SUCCESS = 0
ERROR_INVALID_INPUT = 1
ERROR_SINGULAR_MATRIX = 2
ERROR_CONVERGENCE_FAILURE = 3
ERROR_INVALID_ALPHA = 4
ERROR_INVALID_L1_RATIO = 5
ERROR_INSUFFICIENT_DATA = 6
ERROR_ALLOCATION_FAILURE = 7
ERROR_SERIALIZATION_ERROR = 8
ERROR_DIMENSION_MISMATCH = 9
ERROR_NO_VALID_DATA = 10
ERROR_INTERNAL = 99
STATUS_NULL_TOO_FEW_ROWS = 100

SOLVER = {"qr": 0, "svd": 1, "cholesky": 2}
LAMBDA_SCALING = {"raw": 0, "glmnet": 1}
HC_TYPE = {"none": 0, "hc0": 1, "hc1": 2, "hc2": 3, "hc3": 4}
MODEL = {"ols": 0, "ridge": 1, "wls": 2}

_DP = C.POINTER(C.c_double)


class AnofoxError(C.Structure):
    _fields_ = [("code", C.c_int), ("message", C.c_char * 256)]

    def text(self) -> str:
        return self.message.decode("utf-8", "replace")


class AnofoxDataArray(C.Structure):
    _fields_ = [("data", _DP), ("validity", C.POINTER(C.c_uint8)), ("len", C.c_size_t)]


class AnofoxFitResultCore(C.Structure):
    _fields_ = [("coefficients", _DP), ("coefficients_len", C.c_size_t), ("intercept", C.c_double),
                ("r_squared", C.c_double), ("adj_r_squared", C.c_double), ("residual_std_error", C.c_double),
                ("n_observations", C.c_size_t), ("n_features", C.c_size_t)]


class AnofoxFitResultInference(C.Structure):
    _fields_ = [("std_errors", _DP), ("t_values", _DP), ("p_values", _DP), ("ci_lower", _DP), ("ci_upper", _DP),
                ("len", C.c_size_t), ("confidence_level", C.c_double), ("f_statistic", C.c_double),
                ("f_pvalue", C.c_double)]


class AnofoxOlsOptions(C.Structure):
    _fields_ = [("fit_intercept", C.c_bool), ("compute_inference", C.c_bool), ("confidence_level", C.c_double),
                ("solver", C.c_int), ("hc_type", C.c_int)]


class AnofoxRidgeOptions(C.Structure):
    _fields_ = [("alpha", C.c_double), ("fit_intercept", C.c_bool), ("compute_inference", C.c_bool),
                ("confidence_level", C.c_double), ("solver", C.c_int), ("lambda_scaling", C.c_int)]


class AnofoxWlsOptions(C.Structure):
    _fields_ = AnofoxOlsOptions._fields_


class AnofoxHipBatchOptions(C.Structure):
    _fields_ = [("model", C.c_int), ("fit_intercept", C.c_bool), ("compute_inference", C.c_bool),
                ("confidence_level", C.c_double), ("alpha", C.c_double), ("solver", C.c_int),
                ("lambda_scaling", C.c_int), ("hc_type", C.c_int)]


class AnofoxHipKernelTimes(C.Structure):
    _fields_ = [("accumulate_ms", C.c_double), ("accumulate_count", C.c_int64), ("solve_ms", C.c_double),
                ("solve_count", C.c_int64), ("predict_ms", C.c_double), ("predict_count", C.c_int64),
                ("accumulate_ms_min", C.c_double), ("accumulate_ms_max", C.c_double)]


class AnofoxHipWindowFrame(C.Structure):
    """ROWS BETWEEN start_preceding PRECEDING AND end_preceding PRECEDING; start_preceding < 0 = UNBOUNDED."""
    _fields_ = [("start_preceding", C.c_int64), ("end_preceding", C.c_int64)]


class AnofoxPredictionResult(C.Structure):
    _fields_ = [("yhat", C.c_double), ("yhat_lower", C.c_double), ("yhat_upper", C.c_double)]


RESIDUALS_HAS_STANDARDIZED, RESIDUALS_HAS_STUDENTIZED, RESIDUALS_HAS_LEVERAGE = 1, 2, 4


class AnofoxResidualsResult(C.Structure):  # anofox_stats_ffi.h:527-536
    _fields_ = [("raw", C.POINTER(C.c_double)), ("standardized", C.POINTER(C.c_double)),
                ("studentized", C.POINTER(C.c_double)), ("leverage", C.POINTER(C.c_double)), ("len", C.c_size_t),
                ("has_standardized", C.c_bool), ("has_studentized", C.c_bool), ("has_leverage", C.c_bool)]


# every symbol include/anofox_stats_hip.h declares: name -> (restype, argtypes)
_ERRP = C.POINTER(AnofoxError)
_CTX = C.c_void_p
SYMBOLS = {
    "anofox_ols_fit": (C.c_bool, [AnofoxDataArray, C.POINTER(AnofoxDataArray), C.c_size_t, AnofoxOlsOptions,
                                  C.POINTER(AnofoxFitResultCore), C.POINTER(AnofoxFitResultInference), _ERRP]),
    "anofox_ridge_fit": (C.c_bool, [AnofoxDataArray, C.POINTER(AnofoxDataArray), C.c_size_t, AnofoxRidgeOptions,
                                    C.POINTER(AnofoxFitResultCore), C.POINTER(AnofoxFitResultInference), _ERRP]),
    "anofox_wls_fit": (C.c_bool, [AnofoxDataArray, C.POINTER(AnofoxDataArray), C.c_size_t, AnofoxDataArray,
                                  AnofoxWlsOptions, C.POINTER(AnofoxFitResultCore),
                                  C.POINTER(AnofoxFitResultInference), _ERRP]),
    "anofox_free_result_core": (None, [C.POINTER(AnofoxFitResultCore)]),
    "anofox_free_result_inference": (None, [C.POINTER(AnofoxFitResultInference)]),
    "anofox_compute_aic": (C.c_bool, [C.c_double, C.c_size_t, C.c_size_t, _DP, _ERRP]),
    "anofox_compute_bic": (C.c_bool, [C.c_double, C.c_size_t, C.c_size_t, _DP, _ERRP]),
    "anofox_hip_core_record_len": (C.c_size_t, [C.c_size_t]),
    "anofox_hip_inference_record_len": (C.c_size_t, [C.c_size_t]),
    "anofox_hip_max_features": (C.c_size_t, []),
    "anofox_hip_context_create": (C.c_bool, [C.c_int, C.POINTER(_CTX), _ERRP]),
    "anofox_hip_context_destroy": (None, [_CTX]),
    "anofox_hip_context_set_stream": (C.c_bool, [_CTX, C.c_void_p, _ERRP]),
    "anofox_hip_context_synchronize": (C.c_bool, [_CTX, _ERRP]),
    "anofox_hip_fit_batch_device": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.c_void_p, C.c_void_p,
                                               C.POINTER(C.c_void_p), C.c_void_p, AnofoxHipBatchOptions, C.c_void_p,
                                               C.c_void_p, _ERRP]),
    "anofox_hip_fit_batch_host": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.POINTER(C.c_int64), _DP,
                                             C.POINTER(_DP), _DP, AnofoxHipBatchOptions, _DP, _DP, _ERRP]),
    "anofox_t_critical": (C.c_double, [C.c_double, C.c_size_t]),
    "anofox_predict_with_interval": (C.c_bool, [_DP, C.c_size_t, C.c_double, _DP, C.c_size_t, C.c_double, C.c_size_t,
                                                C.c_double, C.POINTER(AnofoxPredictionResult)]),
    "anofox_predict": (C.c_bool, [C.POINTER(AnofoxDataArray), C.c_size_t, _DP, C.c_size_t, C.c_double,
                                  C.POINTER(_DP), C.POINTER(C.c_size_t), _ERRP]),
    "anofox_free_predictions": (None, [_DP]),
    "anofox_compute_vif": (C.c_bool, [C.POINTER(AnofoxDataArray), C.c_size_t, C.POINTER(_DP), C.POINTER(C.c_size_t), _ERRP]),
    "anofox_free_vif": (None, [_DP]),
    "anofox_compute_residuals": (C.c_bool, [AnofoxDataArray, AnofoxDataArray, C.POINTER(AnofoxDataArray), C.c_size_t, C.c_double,
                                            C.c_bool, C.POINTER(AnofoxResidualsResult), _ERRP]),
    "anofox_free_residuals": (None, [C.POINTER(AnofoxResidualsResult)]),
    "anofox_hip_residuals_max_features": (C.c_size_t, []),
    "anofox_hip_residuals_batch_device": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.POINTER(C.c_void_p), C.c_void_p, C.c_bool, C.c_bool, C.c_void_p,
                                                     C.c_void_p, _ERRP]),
    "anofox_hip_residuals_batch_host": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.POINTER(C.c_int64), _DP, _DP,
                                                   C.POINTER(_DP), _DP, C.c_bool, C.c_bool, _DP, _DP, _ERRP]),
    "anofox_hip_vif_record_len": (C.c_size_t, [C.c_size_t]),
    "anofox_hip_vif_max_features": (C.c_size_t, []),
    "anofox_hip_vif_batch_device": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p),
                                               C.c_void_p, _ERRP]),
    "anofox_hip_vif_batch_host": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.POINTER(C.c_int64), C.POINTER(_DP),
                                             _DP, _ERRP]),
    "anofox_hip_fit_predict_batch_device": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.c_void_p, C.c_void_p,
                                                       C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                                       AnofoxHipBatchOptions, C.c_void_p, C.c_void_p, _ERRP]),
    "anofox_hip_fit_predict_batch_host": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.POINTER(C.c_int64), _DP,
                                                     C.POINTER(_DP), _DP, C.POINTER(C.c_int64), AnofoxHipBatchOptions,
                                                     _DP, _DP, _ERRP]),
    "anofox_hip_fit_predict_expanding_device": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.c_void_p,
                                                           C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p,
                                                           AnofoxHipBatchOptions, C.c_void_p, _ERRP]),
    "anofox_hip_fit_predict_expanding_host": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.POINTER(C.c_int64),
                                                         _DP, C.POINTER(_DP), _DP, AnofoxHipBatchOptions, _DP, _ERRP]),
    "anofox_hip_fit_predict_window_device": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.c_void_p,
                                                        C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p,
                                                        AnofoxHipWindowFrame, AnofoxHipBatchOptions, C.c_void_p, _ERRP]),
    "anofox_hip_fit_predict_window_host": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.POINTER(C.c_int64),
                                                      _DP, C.POINTER(_DP), _DP, AnofoxHipWindowFrame,
                                                      AnofoxHipBatchOptions, _DP, _ERRP]),
    "anofox_hip_predict_batch_device": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_int64, C.c_void_p,
                                                   C.POINTER(C.c_void_p), C.c_void_p, C.c_double, C.c_void_p, _ERRP]),
    "anofox_hip_context_enable_timing": (C.c_bool, [_CTX, C.c_bool, _ERRP]),
    "anofox_hip_context_collect_timing": (C.c_bool, [_CTX, C.POINTER(AnofoxHipKernelTimes), _ERRP]),
    "anofox_hip_context_use_own_stream": (C.c_bool, [_CTX, _ERRP]),
    "anofox_hip_context_set_accumulate_gate": (C.c_bool, [_CTX, C.c_void_p, C.c_void_p, _ERRP]),
    "anofox_hip_context_last_refine_count": (C.c_bool, [_CTX, C.POINTER(C.c_int64), _ERRP]),
    "anofox_hip_version": (C.c_char_p, []),
    "anofox_hip_comm_unique_id": (C.c_bool, [C.c_void_p, _ERRP]),
    "anofox_hip_comm_create": (C.c_bool, [_CTX, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), _ERRP]),
    "anofox_hip_comm_destroy": (None, [C.c_void_p]),
    "anofox_hip_comm_world_size": (C.c_int, [C.c_void_p]),
    "anofox_hip_comm_rank": (C.c_int, [C.c_void_p]),
    "anofox_hip_comm_ranks_seen": (C.c_int, [C.c_void_p]),
    "anofox_hip_gather_records_device": (C.c_bool, [C.c_void_p, C.c_void_p, C.c_int64, C.c_size_t, C.c_void_p, _ERRP]),
    "anofox_hip_context_last_window_refit_count": (C.c_bool, [_CTX, C.POINTER(C.c_int64), _ERRP]),
    "anofox_hip_fit_predict_frames_device": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p,
                                                        C.c_void_p, C.c_void_p, AnofoxHipBatchOptions, C.c_void_p, _ERRP]),
    "anofox_hip_fit_predict_frames_host": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, _DP, C.POINTER(_DP), _DP, C.POINTER(C.c_int64),
                                                      C.POINTER(C.c_int64), AnofoxHipBatchOptions, _DP, _ERRP]),
    "anofox_hip_information_criteria_batch_device": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, C.c_void_p, AnofoxHipBatchOptions,
                                                                C.c_void_p, _ERRP]),
    "anofox_hip_information_criteria_batch_host": (C.c_bool, [_CTX, C.c_int64, C.c_size_t, _DP, AnofoxHipBatchOptions, _DP, _ERRP]),
    "anofox_hip_agg_state_max_features": (C.c_size_t, []),
    "anofox_hip_agg_state_create": (C.c_bool, [_CTX, C.c_size_t, AnofoxHipBatchOptions, C.c_int64, C.POINTER(C.c_void_p), _ERRP]),
    "anofox_hip_agg_state_destroy": (None, [C.c_void_p]),
    "anofox_hip_agg_state_reserve": (C.c_bool, [C.c_void_p, C.c_int64, _ERRP]),
    "anofox_hip_agg_state_retain_rows": (C.c_bool, [C.c_void_p, C.c_size_t, _ERRP]),
    "anofox_hip_agg_state_retain_rows_host": (C.c_bool, [C.c_void_p, C.c_size_t, _ERRP]),
    "anofox_hip_agg_state_retaining": (C.c_int, [C.c_void_p]),
    "anofox_hip_agg_state_retained_host_bytes": (C.c_size_t, [C.c_void_p]),
    "anofox_hip_agg_state_retained_bytes": (C.c_size_t, [C.c_void_p]),
    "anofox_hip_agg_state_slots": (C.c_int64, [C.c_void_p]),
    "anofox_hip_agg_state_rows": (C.c_int64, [C.c_void_p]),
    "anofox_hip_agg_state_update_host": (C.c_bool, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, _ERRP]),
    "anofox_hip_agg_state_update_device": (C.c_bool, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_void_p, _ERRP]),
    "anofox_hip_agg_state_combine": (C.c_bool, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, _ERRP]),
    "anofox_hip_agg_state_combine_ex": (C.c_bool, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_bool, _ERRP]),
    "anofox_hip_agg_state_finalize_slots_host": (C.c_bool, [C.c_void_p, C.c_int64, C.c_void_p, _DP, _DP, C.POINTER(C.c_int64), _ERRP]),
    "anofox_hip_agg_state_release_slots": (C.c_bool, [C.c_void_p, C.c_int64, C.c_void_p, _ERRP]),
    "anofox_hip_agg_state_reset": (C.c_bool, [C.c_void_p, _ERRP]),
    "anofox_hip_agg_state_record_len": (C.c_size_t, [C.c_void_p]),
    "anofox_hip_agg_state_export_slots_host": (C.c_bool, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, _ERRP]),
    "anofox_hip_agg_state_import_slots_host": (C.c_bool, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, _ERRP]),
    "anofox_hip_agg_state_finalize_host": (C.c_bool, [C.c_void_p, C.c_int64, _DP, _DP, C.POINTER(C.c_int64), C.c_void_p, _ERRP]),
    "anofox_hip_agg_state_finalize_device": (C.c_bool, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, _ERRP]),
    "anofox_hip_host_alloc": (C.c_void_p, [C.c_size_t]),
    "anofox_hip_host_free": (None, [C.c_void_p]),
}

_lib = None


def load():
    """Load the shared library and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C anofox-statistics_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class AnofoxStatsError(RuntimeError):
    """A failed library call; carries the AnofoxErrorCode."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code
