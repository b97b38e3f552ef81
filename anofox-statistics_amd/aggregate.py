"""Host-side mirror of the reference's three DuckDB aggregates
(src/aggregate_functions/{ols,ridge,wls}_aggregate.cpp): Bind -> Update -> Combine -> Finalize.

  * Bind      = constructor: parses the constant options argument (options.parse_options)
  * Update    = update(): appends rows; rows whose y (or weight) is NULL or whose x list is NULL are
                skipped (ols_aggregate.cpp:150-159, wls_aggregate.cpp:160-166); the first accumulated
                row fixes the feature count, a different length raises
                "Inconsistent feature count" (ols_aggregate.cpp:165-175)
  * Combine   = combine(): merges another partial state (ols_aggregate.cpp:189-234)
  * Finalize  = finalize(): where the reference loops `anofox_*_fit` over the states
                (ols_aggregate.cpp:257-337) this makes ONE batched GPU call; groups the reference
                maps to SQL NULL (fewer than 2 accumulated rows, or a failed fit) come back with
                is_null set.

The row buffers here are plain numpy chunks; the arithmetic happens in libanofox_stats_hip.so.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, List, Mapping, Optional, Sequence

import numpy as np

from . import _abi
from .options import InvalidInputException, RegressionOptions, parse_options
from .runtime import Context, fit_batch_host


@dataclass
class FitAggResult:
    """Struct-of-arrays form of the aggregate's result column (ols_aggregate.cpp:74-96)."""
    keys: np.ndarray
    coefficients: np.ndarray          # [G, p]
    intercept: np.ndarray             # [G]
    r_squared: np.ndarray
    adj_r_squared: np.ndarray
    residual_std_error: np.ndarray
    n_observations: np.ndarray        # int64 (0 where NULL)
    n_features: int
    status: np.ndarray                # AnofoxErrorCode / 100 per group
    is_null: np.ndarray               # bool: the SQL value is NULL
    std_errors: Optional[np.ndarray] = None
    t_values: Optional[np.ndarray] = None
    p_values: Optional[np.ndarray] = None
    ci_lower: Optional[np.ndarray] = None
    ci_upper: Optional[np.ndarray] = None
    f_statistic: Optional[np.ndarray] = None
    f_pvalue: Optional[np.ndarray] = None

    def __len__(self):
        return len(self.keys)

    def row(self, i: int) -> Optional[dict]:
        """The STRUCT value of group i, or None for SQL NULL."""
        if self.is_null[i]:
            return None
        d = {"coefficients": self.coefficients[i].tolist(), "intercept": float(self.intercept[i]),
             "r_squared": float(self.r_squared[i]), "adj_r_squared": float(self.adj_r_squared[i]),
             "residual_std_error": float(self.residual_std_error[i]),
             "n_observations": int(self.n_observations[i]), "n_features": int(self.n_features)}
        if self.std_errors is not None:
            for k in ("std_errors", "t_values", "p_values", "ci_lower", "ci_upper"):
                d[k] = getattr(self, k)[i].tolist()
            d["f_statistic"] = float(self.f_statistic[i])
            d["f_pvalue"] = float(self.f_pvalue[i])
        return d

    def as_dict(self) -> dict:
        return {k: self.row(i) for i, k in enumerate(self.keys.tolist())}


def result_from_records(keys, core: np.ndarray, inf: Optional[np.ndarray], p: int) -> FitAggResult:
    status = core[:, p + 5].astype(np.int64)
    is_null = status != 0
    nobs = np.where(is_null, 0, np.nan_to_num(core[:, p + 4], nan=0.0)).astype(np.int64)
    res = FitAggResult(keys=np.asarray(keys), coefficients=core[:, :p], intercept=core[:, p], r_squared=core[:, p + 1],
                       adj_r_squared=core[:, p + 2], residual_std_error=core[:, p + 3], n_observations=nobs,
                       n_features=p, status=status, is_null=is_null)
    if inf is not None:
        res.std_errors, res.t_values, res.p_values = inf[:, :p], inf[:, p:2 * p], inf[:, 2 * p:3 * p]
        res.ci_lower, res.ci_upper = inf[:, 3 * p:4 * p], inf[:, 4 * p:5 * p]
        res.f_statistic, res.f_pvalue = inf[:, 5 * p], inf[:, 5 * p + 1]
    return res


def _null_mask_1d(a) -> (np.ndarray, np.ndarray):
    """float64 values + NULL mask from a list (None = NULL), masked array or plain array."""
    if isinstance(a, np.ma.MaskedArray):
        return np.ascontiguousarray(a.filled(np.nan), dtype=np.float64), np.ma.getmaskarray(a).copy()
    arr = np.asarray(a)
    if arr.dtype == object:
        mask = np.array([v is None for v in arr], dtype=bool)
        vals = np.array([np.nan if v is None else float(v) for v in arr], dtype=np.float64)
        return vals, mask
    return np.ascontiguousarray(arr, dtype=np.float64), np.zeros(arr.shape[0], dtype=bool)


class _FitAgg:
    model = "ols"
    sql_name = "anofox_stats_ols_fit_agg"
    has_weights = False

    def __init__(self, options: Optional[Mapping[str, Any]] = None, context: Optional[Context] = None):
        self.options: RegressionOptions = parse_options(options)   # Bind
        self._ctx = context
        self.n_features: Optional[int] = None
        self._keys: List[np.ndarray] = []      # every key seen (also of skipped rows: the group exists)
        self._rkeys: List[np.ndarray] = []     # keys of accumulated rows
        self._y: List[np.ndarray] = []
        self._x: List[np.ndarray] = []         # [n, p] row-major chunks, as the LIST child delivers them
        self._w: List[np.ndarray] = []

    # ---- Update ------------------------------------------------------------------------------
    def update(self, group_keys, y, x, weights=None):
        keys = np.asarray(group_keys)
        yv, ynull = _null_mask_1d(y)
        n = len(yv)
        if len(keys) != n:
            raise InvalidInputException("group_keys and y differ in length")
        skip = ynull.copy()
        if self.has_weights:
            if weights is None:
                raise InvalidInputException(f"{self.sql_name} needs a weight argument")
            wv, wnull = _null_mask_1d(weights)
            skip |= wnull
        # x: list of per-row lists (None = NULL list) or a 2-D array / masked array (fully masked row = NULL list)
        if isinstance(x, np.ma.MaskedArray):
            xnull = np.ma.getmaskarray(x).all(axis=1)
            xa = np.ascontiguousarray(x.filled(np.nan), dtype=np.float64)
            lens = np.full(n, xa.shape[1])
        elif isinstance(x, np.ndarray) and x.dtype != object:
            xa = np.ascontiguousarray(x, dtype=np.float64)
            if xa.ndim != 2:
                raise InvalidInputException("x must be a LIST(DOUBLE) per row, i.e. a 2-D array")
            xnull = np.zeros(n, dtype=bool)
            lens = np.full(n, xa.shape[1])
        else:
            rows = list(x)
            xnull = np.array([r is None for r in rows], dtype=bool)
            lens = np.array([0 if r is None else len(r) for r in rows])
            width = int(lens[~xnull].max()) if (~xnull).any() else 0
            xa = np.full((n, width), np.nan)
            for i, r in enumerate(rows):
                if r is not None:
                    xa[i, :len(r)] = [np.nan if v is None else float(v) for v in r]
        if xa.shape[0] != n:
            raise InvalidInputException("x and y differ in length")
        skip |= xnull
        keep = ~skip
        # feature count is fixed by the first accumulated row; later rows must agree
        for ln in np.unique(lens[keep]):
            if self.n_features is None:
                self.n_features = int(lens[keep][0])
            if int(ln) != self.n_features:
                raise InvalidInputException(
                    f"Inconsistent feature count: expected {self.n_features}, got {int(ln)}")
        self._keys.append(keys)
        if keep.any():
            self._rkeys.append(keys[keep])
            self._y.append(yv[keep])
            self._x.append(np.ascontiguousarray(xa[keep][:, :self.n_features]))
            if self.has_weights:
                self._w.append(wv[keep])
        return self

    # ---- Combine -----------------------------------------------------------------------------
    def combine(self, other: "_FitAgg"):
        if other.n_features is not None:
            if self.n_features is None:
                self.n_features = other.n_features
            elif self.n_features != other.n_features:
                raise InvalidInputException(
                    f"Inconsistent feature count: expected {self.n_features}, got {other.n_features}")
        for name in ("_keys", "_rkeys", "_y", "_x", "_w"):
            getattr(self, name).extend(getattr(other, name))
        return self

    # ---- Finalize ----------------------------------------------------------------------------
    def grouped_columns(self):
        """Sort the accumulated rows by key: (unique keys, row_offsets, y, x_cols, w)."""
        all_keys = np.concatenate(self._keys) if self._keys else np.empty(0)
        ukeys = np.unique(all_keys)
        G = len(ukeys)
        p = self.n_features or 0
        if not self._rkeys:
            return ukeys, np.zeros(G + 1, dtype=np.int64), np.empty(0), [np.empty(0) for _ in range(p)], \
                (np.empty(0) if self.has_weights else None)
        rkeys = np.concatenate(self._rkeys)
        gid = np.searchsorted(ukeys, rkeys)
        order = np.argsort(gid, kind="stable")          # rows keep their arrival order inside a group
        counts = np.bincount(gid, minlength=G)
        offsets = np.zeros(G + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        y = np.concatenate(self._y)[order]
        X = np.concatenate(self._x, axis=0)[order]
        x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]   # row-major LIST rows -> one array per feature
        w = np.concatenate(self._w)[order] if self.has_weights else None
        return ukeys, offsets, y, x_cols, w

    def finalize(self) -> FitAggResult:
        ukeys, offsets, y, x_cols, w = self.grouped_columns()
        G = len(ukeys)
        if self.n_features is None:       # no row was ever accumulated: every group is NULL
            core = np.full((G, 6), np.nan)
            core[:, 5] = _abi.STATUS_NULL_TOO_FEW_ROWS
            return result_from_records(ukeys, core, None, 0)
        p = self.n_features
        opts = self.options.batch_options(self.model)
        core, inf = fit_batch_host(offsets, y, x_cols, w, opts, ctx=self._ctx)
        return result_from_records(ukeys, core, inf, p)


class OlsFitAgg(_FitAgg):
    """anofox_stats_ols_fit_agg(y DOUBLE, x LIST(DOUBLE) [, options]) — alias ols_fit_agg
    (ols_aggregate.cpp:377-426)."""
    model = "ols"
    sql_name = "anofox_stats_ols_fit_agg"


class RidgeFitAgg(_FitAgg):
    """anofox_stats_ridge_fit_agg(y, x [, options]) — alias ridge_fit_agg (ridge_aggregate.cpp:388-440)."""
    model = "ridge"
    sql_name = "anofox_stats_ridge_fit_agg"


class WlsFitAgg(_FitAgg):
    """anofox_stats_wls_fit_agg(y, x, weight [, options]) — alias wls_fit_agg (wls_aggregate.cpp:401-452)."""
    model = "wls"
    sql_name = "anofox_stats_wls_fit_agg"
    has_weights = True


def ols_fit_agg(group_keys, y, x, options=None, context=None) -> FitAggResult:
    """SELECT g, ols_fit_agg(y, x [, options]) FROM t GROUP BY g"""
    return OlsFitAgg(options, context).update(group_keys, y, x).finalize()


def ridge_fit_agg(group_keys, y, x, options=None, context=None) -> FitAggResult:
    return RidgeFitAgg(options, context).update(group_keys, y, x).finalize()


def wls_fit_agg(group_keys, y, x, weights, options=None, context=None) -> FitAggResult:
    return WlsFitAgg(options, context).update(group_keys, y, x, weights).finalize()


# SQL function names (and short aliases) -> implementation, as registered by the reference
SQL_FUNCTIONS = {
    "anofox_stats_ols_fit_agg": ols_fit_agg, "ols_fit_agg": ols_fit_agg,
    "anofox_stats_ridge_fit_agg": ridge_fit_agg, "ridge_fit_agg": ridge_fit_agg,
    "anofox_stats_wls_fit_agg": wls_fit_agg, "wls_fit_agg": wls_fit_agg,
}
