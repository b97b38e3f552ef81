"""Bind-time options of the three aggregates — mirror of the reference's
RegressionMapOptions::ParseFromValue (src/include/map_options_parser.cpp:637-750),
ExtractBool (:21-45), ExtractSolverType / ExtractHcType / ExtractLambdaScaling (:222-266)
and GetRegularizationStrength (src/include/map_options_parser.hpp:265-270), with the
bind-data defaults of ols_aggregate.cpp:48-52, ridge_aggregate.cpp:49-54, wls_aggregate.cpp:49-54.

Keys are case-insensitive; unknown keys are silently ignored (map_options_parser.cpp:798);
`alpha` wins over `lambda`; there is no range check on confidence_level.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Mapping, Optional

from . import _abi


class InvalidInputException(ValueError):
    """Counterpart of duckdb::InvalidInputException for option / input errors."""


def _extract_bool(val: Any) -> Optional[bool]:
    if val is None:
        return None
    if isinstance(val, bool):
        return val
    if isinstance(val, int):
        return val != 0
    if isinstance(val, float):
        return val != 0.0
    try:  # DECIMAL-like
        import decimal
        if isinstance(val, decimal.Decimal):
            return val != 0
    except Exception:  # pragma: no cover
        pass
    raise InvalidInputException(f"Cannot convert value of type {type(val).__name__.upper()} to boolean")


def _extract_double(val: Any) -> Optional[float]:
    return None if val is None else float(val)


def _extract_enum(val: Any, table: Mapping[str, int], what: str, valid: str) -> Optional[str]:
    if val is None:
        return None
    s = str(val).lower()
    if s not in table:
        raise InvalidInputException(f"Invalid {what}: '{s}'. Valid values are {valid}")
    return s


@dataclass
class RegressionOptions:
    """Resolved options (defaults = the C++ bind data of the aggregates)."""
    fit_intercept: bool = True
    compute_inference: bool = False
    confidence_level: float = 0.95
    alpha: float = 1.0                 # ridge_aggregate.cpp:49
    solver: str = "svd"                # ols_aggregate.cpp:51 (accepted, ignored by the GPU path)
    hc_type: str = "none"
    lambda_scaling: str = "raw"
    null_policy: str = "drop"          # predict aggregates only (ols_predict_aggregate.cpp:68)

    def batch_options(self, model: str) -> _abi.AnofoxHipBatchOptions:
        return _abi.AnofoxHipBatchOptions(
            _abi.MODEL[model], self.fit_intercept, self.compute_inference, self.confidence_level, self.alpha,
            _abi.SOLVER[self.solver], _abi.LAMBDA_SCALING[self.lambda_scaling], _abi.HC_TYPE[self.hc_type])


def parse_options(opts: Optional[Mapping[str, Any]]) -> RegressionOptions:
    """Parse a constant MAP / STRUCT literal, given as a Python mapping."""
    out = RegressionOptions()
    if opts is None:
        return out
    if not isinstance(opts, Mapping):
        raise InvalidInputException("Options parameter must be a constant expression")
    alpha = lam = None
    for raw_key, val in opts.items():
        key = str(raw_key).lower()
        if key in ("intercept", "fit_intercept"):
            v = _extract_bool(val)
            if v is not None:
                out.fit_intercept = v
        elif key in ("compute_inference", "inference"):
            v = _extract_bool(val)
            if v is not None:
                out.compute_inference = v
        elif key in ("confidence_level", "confidence"):
            v = _extract_double(val)
            if v is not None:
                out.confidence_level = v
        elif key == "alpha":
            alpha = _extract_double(val)
        elif key == "lambda":
            lam = _extract_double(val)
        elif key == "solver":
            v = _extract_enum(val, _abi.SOLVER, "solver", "'qr', 'svd', 'cholesky'")
            if v is not None:
                out.solver = v
        elif key == "hc_type":
            v = _extract_enum(val, _abi.HC_TYPE, "hc_type", "'none', 'hc0', 'hc1', 'hc2', 'hc3'")
            if v is not None:
                out.hc_type = v
        elif key == "lambda_scaling":
            v = _extract_enum(val, _abi.LAMBDA_SCALING, "lambda_scaling", "'raw', 'glmnet'")
            if v is not None:
                out.lambda_scaling = v
        elif key == "null_policy":
            if val is not None:
                v = str(val).lower()
                if v not in ("drop", "drop_y_zero_x"):
                    raise InvalidInputException(
                        f"Invalid null_policy: '{v}'. Valid values are 'drop', 'drop_y_zero_x'")
                out.null_policy = v
        # every other key: ignored, as in the reference
    if alpha is not None:      # GetRegularizationStrength: alpha first, then lambda
        out.alpha = alpha
    elif lam is not None:
        out.alpha = lam
    return out
