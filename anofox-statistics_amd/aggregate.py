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
from .runtime import AggState, Context, fit_batch_host


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


class StreamingStates:
    """Every aggregate state of one query, kept on the GPU (runtime.AggState / anofox_hip_agg_state_*): what a
    DuckDB shim holds once per query.  Each aggregate object created with `streaming=<this>` plays one thread-local
    hash table: its Initialize takes slot numbers from here, its Update sends accepted rows straight to the GPU
    state (no row buffer), its Combine merges slot pairs, its Finalize reads the solved records of its slots."""

    def __init__(self, context: Optional[Context] = None):
        self.context = context or Context()
        self.state: Optional[AggState] = None
        self.n_slots = 0
        self.unrefined = 0
        self._solved = None        # (core, inf) of all slots, computed by the first Finalize after the last change

    def new_slots(self, n: int) -> np.ndarray:
        out = np.arange(self.n_slots, self.n_slots + n, dtype=np.uint32)
        self.n_slots += n
        return out

    def ensure(self, n_features: int, batch_options) -> AggState:
        if self.state is None:
            self.state = AggState(self.context, n_features, batch_options)
        elif self.state.p != n_features:
            raise InvalidInputException(f"Inconsistent feature count: expected {self.state.p}, got {n_features}")
        return self.state

    def touched(self):
        self._solved = None

    def solved(self):
        if self._solved is None:
            self.state.reserve(self.n_slots)      # slots whose rows were all skipped never reached the GPU
            core, inf, self.unrefined = self.state.finalize(self.n_slots)
            self._solved = (core, inf)
        return self._solved


class _FitAgg:
    model = "ols"
    sql_name = "anofox_stats_ols_fit_agg"
    has_weights = False

    def __init__(self, options: Optional[Mapping[str, Any]] = None, context: Optional[Context] = None,
                 streaming=None):
        self.options: RegressionOptions = parse_options(options)   # Bind
        self._ctx = context
        # streaming: None/False = buffer rows on the host and fit them in one batched call at Finalize (the layout
        # of the reference's state); True or a StreamingStates = O(p^2) state per group on the GPU
        self._pool: Optional[StreamingStates] = None
        if streaming:
            self._pool = streaming if isinstance(streaming, StreamingStates) else StreamingStates(context)
        self._slot_of: dict = {}               # key -> slot (streaming)
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
        if self._pool is not None:
            return self._update_streaming(keys, keep, yv, xa, wv if self.has_weights else None)
        self._keys.append(keys)
        if keep.any():
            self._rkeys.append(keys[keep])
            self._y.append(yv[keep])
            self._x.append(np.ascontiguousarray(xa[keep][:, :self.n_features]))
            if self.has_weights:
                self._w.append(wv[keep])
        return self

    def _update_streaming(self, keys, keep, yv, xa, wv):
        """Update of the streaming state: Initialize a slot for every new key (also for keys whose rows are all
        skipped: the group exists and comes out NULL), then hand the accepted rows to the GPU state."""
        uk, inv = np.unique(keys, return_inverse=True)
        new = [k for k in uk.tolist() if k not in self._slot_of]
        if new:
            for k, sl in zip(new, self._pool.new_slots(len(new)).tolist()):
                self._slot_of[k] = sl
        if keep.any():
            slots = np.array([self._slot_of[k] for k in uk.tolist()], dtype=np.uint32)[inv]
            st = self._pool.ensure(self.n_features, self.options.batch_options(self.model))
            st.update(slots[keep], yv[keep], np.ascontiguousarray(xa[keep][:, :self.n_features]),
                      None if wv is None else wv[keep], n_slots=self._pool.n_slots)
            self._pool.touched()
        return self

    # ---- Combine -----------------------------------------------------------------------------
    def combine(self, other: "_FitAgg"):
        if self._pool is not None:
            if other._pool is not self._pool:
                raise InvalidInputException("streaming aggregates combine only within one StreamingStates pool")
            if other.n_features is not None:
                if self.n_features is None:
                    self.n_features = other.n_features
                elif self.n_features != other.n_features:
                    raise InvalidInputException(
                        f"Cannot combine states with different feature counts: {other.n_features} vs {self.n_features}")
            src, dst = [], []
            for k, sl in other._slot_of.items():
                if k in self._slot_of:
                    src.append(sl)
                    dst.append(self._slot_of[k])
                else:
                    self._slot_of[k] = sl          # the target hash table adopts the source's state
            if src and self._pool.state is not None:
                self._pool.state.reserve(self._pool.n_slots)
                self._pool.state.combine(src, dst)
                self._pool.touched()
            other._slot_of = {}
            return self
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
        if self._pool is not None:
            ukeys = np.array(sorted(self._slot_of.keys()))
            G = len(ukeys)
            if self._pool.state is None:      # no row was ever accumulated: every group is NULL
                core = np.full((G, 6), np.nan)
                core[:, 5] = _abi.STATUS_NULL_TOO_FEW_ROWS
                return result_from_records(ukeys, core, None, 0)
            core, inf = self._pool.solved()
            idx = np.array([self._slot_of[k] for k in ukeys.tolist()], dtype=np.int64)
            return result_from_records(ukeys, core[idx], None if inf is None else inf[idx], self._pool.state.p)
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


# ------------------------------------------------------------------------------------------------------
# *_fit_predict_agg(y, x [, options]): fit on the training rows of each group, predict every row
# (src/aggregate_functions/{ols,ridge,wls}_predict_aggregate.cpp)
# ------------------------------------------------------------------------------------------------------
@dataclass
class FitPredictAggResult:
    """Per group: LIST(STRUCT(y, yhat, yhat_lower, yhat_upper, is_training)) (ols_predict_aggregate.cpp:93-104)."""
    keys: np.ndarray
    row_offsets: np.ndarray           # rows of group i = [row_offsets[i], row_offsets[i+1])
    y: np.ndarray                     # NaN where y was NULL
    y_is_null: np.ndarray
    yhat: np.ndarray                  # NaN = SQL NULL
    yhat_lower: np.ndarray
    yhat_upper: np.ndarray
    is_training: np.ndarray
    is_null: np.ndarray               # per group: the whole LIST is NULL
    core: np.ndarray                  # fit records [G, p+6]

    def rows(self, i: int) -> Optional[list]:
        if self.is_null[i]:
            return None
        lo, hi = int(self.row_offsets[i]), int(self.row_offsets[i + 1])
        out = []
        for r in range(lo, hi):
            nn = lambda v: None if np.isnan(v) else float(v)  # noqa: E731
            out.append({"y": None if self.y_is_null[r] else float(self.y[r]), "yhat": nn(self.yhat[r]),
                        "yhat_lower": nn(self.yhat_lower[r]), "yhat_upper": nn(self.yhat_upper[r]),
                        "is_training": bool(self.is_training[r])})
        return out


class _FitPredictAgg:
    """Bind / Update / Combine / Finalize of the predict aggregates.  Update keeps ALL rows whose x list is not
    NULL; a row trains iff its y is not NULL and no feature is NULL (ols_predict_aggregate.cpp:150-250), or iff the
    optional split column says 'train'/'training' and y is not NULL."""
    model = "ols"
    sql_name = "anofox_stats_ols_fit_predict_agg"
    has_weights = False

    def __init__(self, options: Optional[Mapping[str, Any]] = None, context: Optional[Context] = None):
        self.options: RegressionOptions = parse_options(options)
        self._ctx = context
        self.n_features: Optional[int] = None
        self._chunks: List[tuple] = []
        self._seen_keys: List[np.ndarray] = []

    def update(self, group_keys, y, x, weights=None, split=None):
        keys = np.asarray(group_keys)
        yv, ynull = _null_mask_1d(y)
        n = len(yv)
        rows = [None if r is None else list(r) for r in (x.tolist() if isinstance(x, np.ndarray) else x)]
        if len(rows) != n or len(keys) != n:
            raise InvalidInputException("group_keys, y and x differ in length")
        xnull = np.array([r is None for r in rows], dtype=bool)
        keep = ~xnull                                            # rows with a NULL x list are skipped entirely
        lens = {len(r) for r in rows if r is not None}
        for ln in sorted(lens):
            if self.n_features is None:
                self.n_features = next(len(r) for r in rows if r is not None)
            if ln != self.n_features:
                raise InvalidInputException(f"Inconsistent feature count: expected {self.n_features}, got {ln}")
        p = self.n_features or 0
        X = np.full((n, p), np.nan)
        feat_null = np.zeros(n, dtype=bool)
        for i, r in enumerate(rows):
            if r is not None:
                for j, v in enumerate(r):
                    if v is None:
                        feat_null[i] = True
                    else:
                        X[i, j] = float(v)
        if split is not None:
            sp = [None if s is None else str(s).lower() for s in split]
            train = np.array([s in ("train", "training") for s in sp], dtype=bool) & ~ynull
        else:
            train = ~ynull
        if self.model == "ols":                                  # only the OLS file clears the flag on a NULL feature
            train &= ~feat_null                                  # (ols_predict_aggregate.cpp:236-239); ridge / wls hand the row
                                                                 # to the fit, whose row filter drops it
        if self.options.null_policy == "drop_y_zero_x":          # ols_predict_aggregate.cpp:241-249
            train &= ~np.any(X == 0.0, axis=1)
        wv = None
        if self.has_weights:
            if weights is None:
                raise InvalidInputException(f"{self.sql_name} needs a weight argument")
            wv, wnull = _null_mask_1d(weights)
            keep &= ~wnull                                       # a NULL weight: the row does not exist (wls_predict_aggregate.cpp:170-174)
            train &= ~wnull & (np.nan_to_num(wv, nan=0.0) > 0)   # weight <= 0: kept for the output, does not train (:212-217)
            wv = np.where(train, wv, 1.0)
        self._seen_keys.append(keys)
        if keep.any():
            self._chunks.append((keys[keep], yv[keep], ynull[keep], X[keep], train[keep],
                                 None if wv is None else wv[keep]))
        return self

    def combine(self, other: "_FitPredictAgg"):
        if other.n_features is not None:
            if self.n_features is None:
                self.n_features = other.n_features
            elif self.n_features != other.n_features:
                raise InvalidInputException(
                    f"Cannot combine states with different feature counts: {self.n_features} vs {other.n_features}")
        self._chunks.extend(other._chunks)
        self._seen_keys.extend(other._seen_keys)
        return self

    def finalize(self) -> FitPredictAggResult:
        from .runtime import fit_predict_batch_host
        ukeys = np.unique(np.concatenate(self._seen_keys)) if self._seen_keys else np.empty(0)
        G = len(ukeys)
        p = self.n_features or 0
        if not self._chunks:
            z = np.empty(0)
            return FitPredictAggResult(ukeys, np.zeros(G + 1, dtype=np.int64), z, z.astype(bool), z, z, z,
                                       z.astype(bool), np.ones(G, dtype=bool), np.full((G, p + 6), np.nan))
        keys = np.concatenate([c[0] for c in self._chunks])
        gid = np.searchsorted(ukeys, keys)
        order = np.argsort(gid, kind="stable")
        counts = np.bincount(gid, minlength=G)
        offsets = np.zeros(G + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        y = np.concatenate([c[1] for c in self._chunks])[order]
        ynull = np.concatenate([c[2] for c in self._chunks])[order]
        X = np.concatenate([c[3] for c in self._chunks], axis=0)[order]
        train = np.concatenate([c[4] for c in self._chunks])[order]
        w = np.concatenate([c[5] for c in self._chunks])[order] if self.has_weights else None
        # non-training rows must not enter the fit: their y goes in as NaN (the batch ABI's NULL)
        y_fit = np.where(train, y, np.nan)
        train_counts = np.bincount(gid[order], weights=train.astype(np.float64), minlength=G).astype(np.int64)
        x_cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
        opts = self.options.batch_options(self.model)
        core, pred = fit_predict_batch_host(offsets, y_fit, x_cols, w, opts, train_counts=train_counts, ctx=self._ctx)
        is_null = core[:, p + 5] != 0
        return FitPredictAggResult(ukeys, offsets, np.where(ynull, np.nan, y), ynull, pred[:, 0], pred[:, 1],
                                   pred[:, 2], train, is_null, core)


class OlsFitPredictAgg(_FitPredictAgg):
    model = "ols"
    sql_name = "anofox_stats_ols_fit_predict_agg"


class RidgeFitPredictAgg(_FitPredictAgg):
    model = "ridge"
    sql_name = "anofox_stats_ridge_fit_predict_agg"


class WlsFitPredictAgg(_FitPredictAgg):
    model = "wls"
    sql_name = "anofox_stats_wls_fit_predict_agg"
    has_weights = True


def ols_fit_predict_agg(group_keys, y, x, options=None, context=None, split=None) -> FitPredictAggResult:
    return OlsFitPredictAgg(options, context).update(group_keys, y, x, split=split).finalize()


def ridge_fit_predict_agg(group_keys, y, x, options=None, context=None, split=None) -> FitPredictAggResult:
    return RidgeFitPredictAgg(options, context).update(group_keys, y, x, split=split).finalize()


def wls_fit_predict_agg(group_keys, y, x, weights, options=None, context=None, split=None) -> FitPredictAggResult:
    return WlsFitPredictAgg(options, context).update(group_keys, y, x, weights, split=split).finalize()


SQL_FUNCTIONS.update({
    "anofox_stats_ols_fit_predict_agg": ols_fit_predict_agg, "ols_fit_predict_agg": ols_fit_predict_agg,
    "ols_predict_agg": ols_fit_predict_agg, "anofox_stats_ols_predict_agg": ols_fit_predict_agg,  # deprecated aliases
    "anofox_stats_ridge_fit_predict_agg": ridge_fit_predict_agg, "ridge_fit_predict_agg": ridge_fit_predict_agg,
    "anofox_stats_wls_fit_predict_agg": wls_fit_predict_agg, "wls_fit_predict_agg": wls_fit_predict_agg,
})


# ------------------------------------------------------------------------------------------------------
# vif_agg(x LIST(DOUBLE)) -> LIST(DOUBLE)  (src/aggregate_functions/vif_aggregate.cpp)
# ------------------------------------------------------------------------------------------------------
def vif_agg(group_keys, x, context=None):
    """GROUP BY mirror of vif_agg.  Returns (keys, [list of p VIFs or None per group]).

    Update (vif_aggregate.cpp:50-95): NULL x lists are skipped; every non-NaN value is appended to ITS column, so a
    NaN shortens only that column.  Finalize (:144-185): NULL unless >= 2 features and >= 3 values in the first
    column; columns of unequal length make compute_vif fail (vif.rs:40-51) -> NULL."""
    from .runtime import vif_batch_host
    keys = np.asarray(group_keys)
    rows = [None if r is None else [np.nan if v is None else float(v) for v in r] for r in x]
    ukeys, gid = np.unique(keys, return_inverse=True)
    per_group = [[] for _ in ukeys]
    for i, r in enumerate(rows):
        if r is not None:
            per_group[gid[i]].append(r)
    result = [None] * len(ukeys)
    batch, batch_ids, p_batch = [], [], None
    for g, rs in enumerate(per_group):
        if not rs:
            continue
        p = len(rs[0])
        for r in rs:
            if len(r) != p:
                raise InvalidInputException(f"Inconsistent feature count: expected {p}, got {len(r)}")
        cols = [np.array([r[j] for r in rs if not np.isnan(r[j])]) for j in range(p)]
        if p < 2 or len(cols[0]) < 3 or any(len(c) != len(cols[0]) for c in cols):
            continue                                             # SQL NULL
        if p_batch is not None and p != p_batch:
            raise InvalidInputException("vif_agg: all groups of one call must have the same feature count")
        p_batch = p
        batch.append(cols)
        batch_ids.append(g)
    if batch:
        offsets = np.zeros(len(batch) + 1, dtype=np.int64)
        np.cumsum([len(c[0]) for c in batch], out=offsets[1:])
        x_cols = [np.concatenate([c[j] for c in batch]) for j in range(p_batch)]
        out = vif_batch_host(offsets, x_cols, ctx=context)
        for k, g in enumerate(batch_ids):
            result[g] = None if out[k, p_batch] != 0 else [float(v) for v in out[k, :p_batch]]
    return ukeys, result


SQL_FUNCTIONS.update({"anofox_stats_vif_agg": vif_agg, "vif_agg": vif_agg})


# ------------------------------------------------------------------------------------------------------
# *_fit_predict(y, x [, options]) OVER (PARTITION BY k ORDER BY o ROWS BETWEEN UNBOUNDED PRECEDING AND
# CURRENT ROW | k PRECEDING, or any ROWS BETWEEN a PRECEDING AND b PRECEDING): the window functions (src/window_functions/{ols,ridge,wls}_fit_predict.cpp)
# ------------------------------------------------------------------------------------------------------
def _parse_frame(frame, frame_end):
    """(start, end) in rows PRECEDING the current row (negative = FOLLOWING; start None = UNBOUNDED PRECEDING, end
    None = UNBOUNDED FOLLOWING) from `frame` = (start, end), each "unbounded preceding" / "unbounded following" /
    "<k> preceding" / "current row" / "<k> following" or an integer; `frame_end` is the older spelling for frames
    that start UNBOUNDED PRECEDING."""
    def bound(v, what, is_start):
        if isinstance(v, str):
            t = v.strip().lower()
            if t in ("current row", "current"):
                return 0
            if t == "unbounded":
                return None
            if t == "unbounded preceding" and is_start:
                return None
            if t == "unbounded following" and not is_start:
                return None
            if t.endswith(" preceding") and t[:-10].strip().isdigit():
                return int(t[:-10])
            if t.endswith(" following") and t[:-10].strip().isdigit():
                return -int(t[:-10])
            raise InvalidInputException(f"{what} must be 'unbounded preceding|following', 'current row', '<k> preceding' "
                                        "or '<k> following'")
        return v if v is None else int(v)
    if frame is None:
        frame = (None, frame_end)
    start, end = bound(frame[0], "frame start", True), bound(frame[1], "frame end", False)
    if start is not None and end is not None and start < end:
        raise InvalidInputException("the frame must start at or before its end")
    return start, end


def _fit_predict_window(model, partition_keys, order, y, x, weights, options, context, frame_end, frame=None):
    """Returns (yhat, yhat_lower, yhat_upper) per input row, in input order; NaN = SQL NULL.
    ROWS BETWEEN frame[0] PRECEDING AND frame[1] PRECEDING; the default frame starts UNBOUNDED PRECEDING and
    ends at `frame_end` ("current row" or "<k> preceding")."""
    from .runtime import fit_predict_window_host
    opts = parse_options(options)
    start, end = _parse_frame(frame, frame_end)
    keys = np.asarray(partition_keys)
    yv, ynull = _null_mask_1d(y)
    yv = np.where(ynull, np.nan, yv)
    rows = [None if r is None else [np.nan if v is None else float(v) for v in r] for r in x]
    p = max((len(r) for r in rows if r is not None), default=0)
    Xd = np.full((len(yv), p), np.nan)
    for i, r in enumerate(rows):
        if r is None:
            continue                                             # NULL x list: no current x, no training
        if len(r) != p:
            raise InvalidInputException(f"Inconsistent feature count: expected {p}, got {len(r)}")
        Xd[i] = r
    if opts.null_policy == "drop_y_zero_x":                      # ols_fit_predict.cpp:171-178
        yv = np.where(np.any(Xd == 0.0, axis=1), np.nan, yv)
    ukeys, gid = np.unique(keys, return_inverse=True)
    perm = np.lexsort((np.asarray(order), gid))                  # PARTITION BY keys ORDER BY order
    counts = np.bincount(gid, minlength=len(ukeys))
    offsets = np.zeros(len(ukeys) + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    wv = None
    if weights is not None:
        wv, wnull = _null_mask_1d(weights)
        wv = np.where(wnull, np.nan, wv)[perm]
    pred_sorted = fit_predict_window_host(offsets, yv[perm], [np.ascontiguousarray(Xd[perm, j]) for j in range(p)],
                                          wv, opts.batch_options(model), (start, end), ctx=context)
    out = np.empty_like(pred_sorted)
    out[perm] = pred_sorted
    return out[:, 0], out[:, 1], out[:, 2]


def ols_fit_predict(partition_keys, order, y, x, options=None, context=None, frame_end="current row", frame=None):
    return _fit_predict_window("ols", partition_keys, order, y, x, None, options, context, frame_end, frame)


def ridge_fit_predict(partition_keys, order, y, x, options=None, context=None, frame_end="current row", frame=None):
    return _fit_predict_window("ridge", partition_keys, order, y, x, None, options, context, frame_end, frame)


def wls_fit_predict(partition_keys, order, y, x, weights, options=None, context=None, frame_end="current row", frame=None):
    return _fit_predict_window("wls", partition_keys, order, y, x, weights, options, context, frame_end, frame)


SQL_FUNCTIONS.update({
    "anofox_stats_ols_fit_predict": ols_fit_predict, "ols_fit_predict": ols_fit_predict,
    "anofox_stats_ridge_fit_predict": ridge_fit_predict, "ridge_fit_predict": ridge_fit_predict,
    "anofox_stats_wls_fit_predict": wls_fit_predict, "wls_fit_predict": wls_fit_predict,
})


# ------------------------------------------------------------------------------------------------------
# residuals_diagnostics_agg(y, y_hat[, x LIST(DOUBLE)]) -> STRUCT(raw, standardized, studentized, leverage)
# (src/aggregate_functions/residuals_diagnostics_aggregate.cpp)
# ------------------------------------------------------------------------------------------------------
def residuals_diagnostics_agg(group_keys, y, y_hat, x=None, context=None):
    """GROUP BY mirror of residuals_diagnostics_agg.  Returns (keys, [dict(raw=, standardized=, studentized=,
    leverage=) or None per group]).

    Update (residuals_diagnostics_aggregate.cpp:71-163): rows with a NULL / NaN y or y_hat (or a NULL x list) are
    skipped.  Finalize (:213-286): NULL for fewer than 3 buffered rows; the residual standard error is passed as NaN
    (:232), so `standardized` and `studentized` are always NULL and the 3-argument form adds `leverage` (NULL when
    the design is rank deficient)."""
    from .runtime import residuals_batch_host
    from . import _abi
    keys = np.asarray(group_keys)
    yv = np.array([np.nan if v is None else float(v) for v in y], dtype=np.float64)
    yh = np.array([np.nan if v is None else float(v) for v in y_hat], dtype=np.float64)
    keep = ~np.isnan(yv) & ~np.isnan(yh)
    p = 0
    xm = None
    if x is not None:
        keep &= np.array([r is not None for r in x], dtype=bool)
        first = next((r for r, k in zip(x, keep) if k and len(r) > 0), None)   # :141-145: first non-empty valid row
        p = 0 if first is None else len(first)
        for r, k in zip(x, keep):
            if k and len(r) != p:
                raise InvalidInputException(f"Inconsistent feature count: expected {p}, got {len(r)}")
        xm = np.array([[np.nan if v is None else float(v) for v in r] if k else [np.nan] * p
                       for r, k in zip(x, keep)], dtype=np.float64).reshape(len(keys), p)
    ukeys, gid = np.unique(keys, return_inverse=True)
    order = np.argsort(gid[keep], kind="stable")
    rows = np.flatnonzero(keep)[order]
    counts = np.bincount(gid[keep], minlength=len(ukeys))
    offsets = np.zeros(len(ukeys) + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    x_cols = [np.ascontiguousarray(xm[rows, j]) for j in range(p)] if p else []
    result = [None] * len(ukeys)
    if len(ukeys) == 0:
        return ukeys, result
    out, group = residuals_batch_host(offsets, yv[rows], yh[rows], x_cols, None, include_studentized=x is not None,
                                      drop_nan_rows=True, ctx=context)
    for g in range(len(ukeys)):
        lo, hi = offsets[g], offsets[g + 1]
        if hi - lo < 3:
            continue                                             # SQL NULL (:223)
        flags = int(group[g, 1])
        result[g] = dict(raw=[float(v) for v in out[lo:hi, 0]], standardized=None, studentized=None,
                         leverage=[float(v) for v in out[lo:hi, 3]] if flags & _abi.RESIDUALS_HAS_LEVERAGE else None)
    return ukeys, result


SQL_FUNCTIONS.update({"anofox_stats_residuals_diagnostics_agg": residuals_diagnostics_agg,
                      "residuals_diagnostics_agg": residuals_diagnostics_agg})
