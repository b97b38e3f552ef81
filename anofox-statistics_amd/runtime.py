"""Contexts and the two batch entry points (host pointers / device pointers)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import _abi
from ._abi import AnofoxStatsError


class Context:
    """Owns one AnofoxHipContext (device + stream + reusable workspace)."""

    def __init__(self, device: int = -1):
        lib = _abi.load()
        err = _abi.AnofoxError()
        handle = C.c_void_p()
        if not lib.anofox_hip_context_create(int(device), C.byref(handle), C.byref(err)):
            raise AnofoxStatsError(err.code, err.text())
        self._h = handle
        self._lib = lib

    def close(self):
        if getattr(self, "_h", None):
            self._lib.anofox_hip_context_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _check(self, ok, err):
        if not ok:
            raise AnofoxStatsError(err.code, err.text())

    def set_stream(self, hip_stream: Optional[int]):
        """Launch on the given hipStream_t handle.  0 is HIP's default stream (= torch's default stream) and is used
        as given; None goes back to the context's own non-blocking stream."""
        err = _abi.AnofoxError()
        if hip_stream is None:
            self._check(self._lib.anofox_hip_context_use_own_stream(self._h, C.byref(err)), err)
        else:
            self._check(self._lib.anofox_hip_context_set_stream(self._h, C.c_void_p(int(hip_stream)), C.byref(err)), err)

    def set_accumulate_gate(self, wait_event=None, record_event=None):
        """torch.cuda.Event objects (or None): wait for `wait_event` before the accumulate kernel of the next fit
        calls, record `record_event` right after it (see anofox_hip_context_set_accumulate_gate)."""
        err = _abi.AnofoxError()
        h = lambda ev: C.c_void_p(0 if ev is None else int(ev.cuda_event))
        # the library holds the raw hipEvent_t handles until the next fit call consumes the (one-shot) gate: keep the
        # torch objects alive at least that long, whatever the caller does with its own references
        self._gate_events = (wait_event, record_event)
        self._check(self._lib.anofox_hip_context_set_accumulate_gate(self._h, h(wait_event), h(record_event), C.byref(err)), err)

    def synchronize(self):
        err = _abi.AnofoxError()
        self._check(self._lib.anofox_hip_context_synchronize(self._h, C.byref(err)), err)

    def enable_timing(self, enable: bool = True):
        err = _abi.AnofoxError()
        self._check(self._lib.anofox_hip_context_enable_timing(self._h, bool(enable), C.byref(err)), err)

    def collect_timing(self) -> dict:
        err = _abi.AnofoxError()
        t = _abi.AnofoxHipKernelTimes()
        self._check(self._lib.anofox_hip_context_collect_timing(self._h, C.byref(t), C.byref(err)), err)
        return {"accumulate_ms": t.accumulate_ms, "accumulate_count": t.accumulate_count,
                "solve_ms": t.solve_ms, "solve_count": t.solve_count,
                "predict_ms": t.predict_ms, "predict_count": t.predict_count,
                "accumulate_ms_min": t.accumulate_ms_min, "accumulate_ms_max": t.accumulate_ms_max}

    # ---- device-resident batch (torch tensors on this context's GPU) -------------------------
    def fit_batch_device(self, row_offsets, y, x_cols: Sequence, w, options: _abi.AnofoxHipBatchOptions,
                         core=None, inference=None, use_current_torch_stream: bool = True):
        """row_offsets: int64[G+1]; y, x_cols[j], w: float64[N] CUDA tensors.  Asynchronous.
        Returns (core[G, p+6], inference[G, 5p+2] or None) CUDA tensors."""
        import torch

        p = len(x_cols)
        G = int(row_offsets.numel()) - 1
        N = int(y.numel())
        for t in (row_offsets, y, *x_cols) + ((w,) if w is not None else ()):
            if not t.is_cuda or not t.is_contiguous():
                raise ValueError("device batch needs contiguous CUDA tensors")
        if row_offsets.dtype != torch.int64 or y.dtype != torch.float64 or any(c.dtype != torch.float64 for c in x_cols):
            raise ValueError("row_offsets must be int64 and data float64")
        if any(int(c.numel()) != N for c in x_cols) or (w is not None and int(w.numel()) != N):
            raise ValueError("every column must have y's length")
        if core is None:
            core = torch.empty((G, p + 6), dtype=torch.float64, device=y.device)
        if options.compute_inference and inference is None:
            inference = torch.empty((G, 5 * p + 2), dtype=torch.float64, device=y.device)
        if use_current_torch_stream:
            self.set_stream(torch.cuda.current_stream(y.device).cuda_stream)
        cols = (C.c_void_p * p)(*[c.data_ptr() for c in x_cols])
        err = _abi.AnofoxError()
        ok = self._lib.anofox_hip_fit_batch_device(
            self._h, G, p, N, C.c_void_p(row_offsets.data_ptr()), C.c_void_p(y.data_ptr()), cols,
            C.c_void_p(w.data_ptr() if w is not None else 0), options, C.c_void_p(core.data_ptr()),
            C.c_void_p(inference.data_ptr() if inference is not None else 0), C.byref(err))
        self._gate_events = None   # consumed (or dropped) by the call above
        self._check(ok, err)
        return core, (inference if options.compute_inference else None)

    def fit_predict_batch_device(self, row_offsets, y, x_cols: Sequence, w, options: _abi.AnofoxHipBatchOptions,
                                 train_counts=None, core=None, pred=None, use_current_torch_stream: bool = True):
        """fit + per-row predictions, device resident.  Returns (core[G, p+6], pred[N, 3]) CUDA tensors."""
        import torch

        p = len(x_cols)
        G = int(row_offsets.numel()) - 1
        N = int(y.numel())
        if core is None:
            core = torch.empty((G, p + 6), dtype=torch.float64, device=y.device)
        if pred is None:
            pred = torch.empty((N, 3), dtype=torch.float64, device=y.device)
        if use_current_torch_stream:
            self.set_stream(torch.cuda.current_stream(y.device).cuda_stream)
        cols = (C.c_void_p * p)(*[c.data_ptr() for c in x_cols])
        err = _abi.AnofoxError()
        ok = self._lib.anofox_hip_fit_predict_batch_device(
            self._h, G, p, N, C.c_void_p(row_offsets.data_ptr()), C.c_void_p(y.data_ptr()), cols,
            C.c_void_p(w.data_ptr() if w is not None else 0),
            C.c_void_p(train_counts.data_ptr() if train_counts is not None else 0), options,
            C.c_void_p(core.data_ptr()), C.c_void_p(pred.data_ptr()), C.byref(err))
        self._check(ok, err)
        return core, pred

    def fit_predict_expanding_device(self, row_offsets, y, x_cols: Sequence, w, options: _abi.AnofoxHipBatchOptions,
                                     pred=None, use_current_torch_stream: bool = True):
        """Expanding-window fit + predict, device resident.  Returns pred[N, 3] (CUDA tensor)."""
        return self.fit_predict_window_device(row_offsets, y, x_cols, w, options, (None, 0), pred, use_current_torch_stream)

    def fit_predict_window_device(self, row_offsets, y, x_cols: Sequence, w, options: _abi.AnofoxHipBatchOptions,
                                  frame=(None, 0), pred=None, use_current_torch_stream: bool = True):
        """Window fit + predict over ROWS BETWEEN frame[0] PRECEDING AND frame[1] PRECEDING (frame[0] None =
        UNBOUNDED), device resident.  Returns pred[N, 3] (CUDA tensor)."""
        import torch

        p = len(x_cols)
        G = int(row_offsets.numel()) - 1
        N = int(y.numel())
        if pred is None:
            pred = torch.empty((N, 3), dtype=torch.float64, device=y.device)
        if use_current_torch_stream:
            self.set_stream(torch.cuda.current_stream(y.device).cuda_stream)
        cols = (C.c_void_p * p)(*[c.data_ptr() for c in x_cols])
        err = _abi.AnofoxError()
        ok = self._lib.anofox_hip_fit_predict_window_device(
            self._h, G, p, N, C.c_void_p(row_offsets.data_ptr()), C.c_void_p(y.data_ptr()), cols,
            C.c_void_p(w.data_ptr() if w is not None else 0), _frame(frame), options, C.c_void_p(pred.data_ptr()),
            C.byref(err))
        self._check(ok, err)
        return pred

    def information_criteria_device(self, core, options: _abi.AnofoxHipBatchOptions, out=None, use_current_torch_stream: bool = True):
        """CUDA fit records [G, p+6] -> out[G, 3] = {rss, aic, bic}; asynchronous."""
        import torch
        G, p = int(core.shape[0]), int(core.shape[1]) - 6
        if out is None:
            out = torch.empty((G, 3), dtype=torch.float64, device=core.device)
        if use_current_torch_stream:
            self.set_stream(torch.cuda.current_stream(core.device).cuda_stream)
        err = _abi.AnofoxError()
        ok = self._lib.anofox_hip_information_criteria_batch_device(self._h, G, p, C.c_void_p(core.data_ptr()), options,
                                                                    C.c_void_p(out.data_ptr()), C.byref(err))
        self._check(ok, err)
        return out

    def last_window_refit_count(self) -> int:
        """Output rows of the most recent window call that were refitted with refinement (diagnostic)."""
        n = C.c_int64()
        err = _abi.AnofoxError()
        self._check(self._lib.anofox_hip_context_last_window_refit_count(self._h, C.byref(n), C.byref(err)), err)
        return int(n.value)

    def last_refine_count(self) -> int:
        """Groups of the most recent fit launch that took the on-device refinement passes (diagnostic)."""
        n = C.c_int64()
        err = _abi.AnofoxError()
        self._check(self._lib.anofox_hip_context_last_refine_count(self._h, C.byref(n), C.byref(err)), err)
        return int(n.value)

    def vif_batch_device(self, row_offsets, x_cols: Sequence, out=None, use_current_torch_stream: bool = True):
        """Grouped variance inflation factors, device resident.  Returns out[G, p+1] = {vif[p], status}."""
        import torch

        p = len(x_cols)
        G = int(row_offsets.numel()) - 1
        N = int(x_cols[0].numel())
        if out is None:
            out = torch.empty((G, p + 1), dtype=torch.float64, device=x_cols[0].device)
        if use_current_torch_stream:
            self.set_stream(torch.cuda.current_stream(x_cols[0].device).cuda_stream)
        cols = (C.c_void_p * p)(*[c.data_ptr() for c in x_cols])
        err = _abi.AnofoxError()
        ok = self._lib.anofox_hip_vif_batch_device(self._h, G, p, N, C.c_void_p(row_offsets.data_ptr()), cols,
                                                   C.c_void_p(out.data_ptr()), C.byref(err))
        self._check(ok, err)
        return out

    def residuals_batch_device(self, row_offsets, y, y_hat, x_cols: Sequence = (), rse=None, include_studentized: bool = True,
                               drop_nan_rows: bool = True, out=None, group=None, use_current_torch_stream: bool = True):
        """Grouped residual diagnostics, device resident.  Returns (out[N, 4] = raw / standardized / studentized /
        leverage per row, group[G, 2] = rows used / ANOFOX_HIP_RESIDUALS_HAS_* flags)."""
        import torch

        p = len(x_cols)
        G = int(row_offsets.numel()) - 1
        N = int(y.numel())
        if out is None:
            out = torch.empty((N, 4), dtype=torch.float64, device=y.device)
        if group is None:
            group = torch.empty((G, 2), dtype=torch.float64, device=y.device)
        if use_current_torch_stream:
            self.set_stream(torch.cuda.current_stream(y.device).cuda_stream)
        cols = (C.c_void_p * max(p, 1))(*[c.data_ptr() for c in x_cols])
        err = _abi.AnofoxError()
        ok = self._lib.anofox_hip_residuals_batch_device(
            self._h, G, p, N, C.c_void_p(row_offsets.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(y_hat.data_ptr()),
            cols if p else None, C.c_void_p(rse.data_ptr()) if rse is not None else None, bool(include_studentized),
            bool(drop_nan_rows), C.c_void_p(out.data_ptr()), C.c_void_p(group.data_ptr()), C.byref(err))
        self._check(ok, err)
        return out, group

    # ---- host-resident batch (numpy) ----------------------------------------------------------
    def fit_batch_host(self, row_offsets, y, x_cols: Sequence, w, options: _abi.AnofoxHipBatchOptions):
        return fit_batch_host(row_offsets, y, x_cols, w, options, ctx=self)


_DP = C.POINTER(C.c_double)


class AggState:
    """The streaming aggregate state of {ols,ridge,wls}_fit_agg on the GPU (anofox_hip_agg_state_*): one O(p^2)
    moment record per slot; `update` folds row chunks in, `combine` merges slots, `finalize` solves them."""

    def __init__(self, ctx: Context, n_features: int, options: _abi.AnofoxHipBatchOptions, initial_slots: int = 0,
                 retain_bytes: Optional[int] = None, retain_host_bytes: Optional[int] = None):
        """retain_bytes: HBM for a row log next to the moments (anofox_hip_agg_state_retain_rows), so that finalize can
        refit the groups the moments alone cannot resolve; retain_host_bytes: page-locked host memory the log continues
        in once that is spent (anofox_hip_agg_state_retain_rows_host).  None = the shim's defaults (64 GiB / 32 GiB,
        ANOFOX_HIP_RETAIN_BYTES / ANOFOX_HIP_RETAIN_HOST_BYTES; slabs are allocated as rows arrive), 0 = moments only.
        Groups that finalize can neither resolve nor refit come back as NaN records with status 101."""
        log_only = int(n_features) > 8 or (bool(options.compute_inference) and int(options.hc_type) != 0 and int(options.model) != 1)
        if retain_bytes is None:      # (a log-only state IS its log: uncapped unless the caller caps it)
            retain_bytes = 0 if log_only else int(os.environ.get("ANOFOX_HIP_RETAIN_BYTES", 64 << 30))
        if retain_host_bytes is None:
            # (an explicit retain_bytes = 0 on a moment state means "moments only", as documented: round 3 left the host
            # continuation on in that case, and every chunk was copied back into page-locked slabs — 9 GB/s instead of 55)
            moments_only = (not log_only) and retain_bytes == 0
            retain_host_bytes = 0 if moments_only else int(os.environ.get("ANOFOX_HIP_RETAIN_HOST_BYTES", 32 << 30))
        self._lib = _abi.load()
        self._ctx = ctx          # keeps the context alive
        self.p = int(n_features)
        self.options = options
        err = _abi.AnofoxError()
        h = C.c_void_p()
        if not self._lib.anofox_hip_agg_state_create(ctx._h, self.p, options, int(initial_slots), C.byref(h), C.byref(err)):
            raise AnofoxStatsError(err.code, err.text())
        self._h = h
        self.unrefined_slots = np.empty(0, dtype=np.int32)
        if retain_bytes:
            if not self._lib.anofox_hip_agg_state_retain_rows(self._h, int(retain_bytes), C.byref(err)):
                raise AnofoxStatsError(err.code, err.text())
        if retain_host_bytes:
            if not self._lib.anofox_hip_agg_state_retain_rows_host(self._h, int(retain_host_bytes), C.byref(err)):
                raise AnofoxStatsError(err.code, err.text())

    @property
    def retaining(self) -> bool:
        return bool(self._lib.anofox_hip_agg_state_retaining(self._h))

    @property
    def retained_bytes(self) -> int:
        return int(self._lib.anofox_hip_agg_state_retained_bytes(self._h))

    @property
    def retained_host_bytes(self) -> int:
        return int(self._lib.anofox_hip_agg_state_retained_host_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.anofox_hip_agg_state_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_slots(self) -> int:
        return int(self._lib.anofox_hip_agg_state_slots(self._h))

    @property
    def n_rows(self) -> int:
        return int(self._lib.anofox_hip_agg_state_rows(self._h))

    def reserve(self, n_slots: int):
        err = _abi.AnofoxError()
        if not self._lib.anofox_hip_agg_state_reserve(self._h, int(n_slots), C.byref(err)):
            raise AnofoxStatsError(err.code, err.text())

    def update(self, slot, y, x_rowmajor, w=None, valid=None, n_slots: Optional[int] = None):
        """Host chunk (numpy): slot uint32[n], y float64[n], x_rowmajor float64[n, p], w float64[n], valid uint8[n]."""
        sl = np.ascontiguousarray(slot, dtype=np.uint32)
        yv = np.ascontiguousarray(y, dtype=np.float64)
        xv = np.ascontiguousarray(x_rowmajor, dtype=np.float64)
        n = len(yv)
        if len(sl) != n or xv.size != n * self.p:
            raise ValueError("slot, y and x disagree in length")
        wv = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        vv = None if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
        if n_slots is None:
            n_slots = max(self.n_slots, int(sl.max()) + 1 if n else 0)
        err = _abi.AnofoxError()
        ok = self._lib.anofox_hip_agg_state_update_host(
            self._h, n, int(n_slots), sl.ctypes.data, yv.ctypes.data, xv.ctypes.data,
            None if wv is None else wv.ctypes.data, None if vv is None else vv.ctypes.data, C.byref(err))
        if not ok:
            raise AnofoxStatsError(err.code, err.text())

    def update_device(self, slot, y, x_rowmajor, w=None, valid=None, n_slots: Optional[int] = None,
                      use_current_torch_stream: bool = True):
        """Device chunk (CUDA tensors): slot int32/uint32-as-int32[n], y[n], x_rowmajor[n, p], w[n], valid uint8[n]."""
        import torch
        n = int(y.numel())
        if n_slots is None:
            n_slots = max(self.n_slots, int(slot.max().item()) + 1 if n else 0)
        if use_current_torch_stream:
            self._ctx.set_stream(torch.cuda.current_stream(y.device).cuda_stream)
        err = _abi.AnofoxError()
        ok = self._lib.anofox_hip_agg_state_update_device(
            self._h, n, int(n_slots), C.c_void_p(slot.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(x_rowmajor.data_ptr()),
            C.c_void_p(w.data_ptr() if w is not None else 0), C.c_void_p(valid.data_ptr() if valid is not None else 0),
            C.byref(err))
        if not ok:
            raise AnofoxStatsError(err.code, err.text())

    def combine(self, source_slots, target_slots):
        src = np.ascontiguousarray(source_slots, dtype=np.uint32)
        dst = np.ascontiguousarray(target_slots, dtype=np.uint32)
        if len(src) != len(dst):
            raise ValueError("source and target differ in length")
        err = _abi.AnofoxError()
        if not self._lib.anofox_hip_agg_state_combine(self._h, len(src), src.ctypes.data, dst.ctypes.data, C.byref(err)):
            raise AnofoxStatsError(err.code, err.text())

    def finalize(self, n_slots: Optional[int] = None):
        """-> (core[G, p+6], inference[G, 5p+2] or None, number of groups flagged as unrefined: status 101, NaN record —
        they asked for the refinement passes and their rows were not kept)."""
        G = self.n_slots if n_slots is None else int(n_slots)
        p = self.p
        core = np.empty((G, p + 6), dtype=np.float64)
        inf = np.empty((G, 5 * p + 2), dtype=np.float64) if self.options.compute_inference else None
        unref = C.c_int64()
        slots = np.empty(max(G, 1), dtype=np.int32)
        err = _abi.AnofoxError()
        ok = self._lib.anofox_hip_agg_state_finalize_host(
            self._h, G, core.ctypes.data_as(_DP), None if inf is None else inf.ctypes.data_as(_DP), C.byref(unref),
            slots.ctypes.data, C.byref(err))
        if not ok:
            raise AnofoxStatsError(err.code, err.text())
        self.unrefined_slots = np.sort(slots[:int(unref.value)])   # groups that would have taken the refinement passes
        return core, inf, int(unref.value)

    def finalize_device(self, core, inference=None, n_slots: Optional[int] = None, use_current_torch_stream: bool = True):
        import torch
        G = self.n_slots if n_slots is None else int(n_slots)
        if use_current_torch_stream:
            self._ctx.set_stream(torch.cuda.current_stream(core.device).cuda_stream)
        err = _abi.AnofoxError()
        ok = self._lib.anofox_hip_agg_state_finalize_device(
            self._h, G, C.c_void_p(core.data_ptr()), C.c_void_p(inference.data_ptr() if inference is not None else 0),
            C.byref(err))
        if not ok:
            raise AnofoxStatsError(err.code, err.text())
        return core, inference


def fit_batch_host(row_offsets, y, x_cols: Sequence, w, options: _abi.AnofoxHipBatchOptions,
                   ctx: Optional[Context] = None):
    """numpy in, numpy out: (core[G, p+6], inference[G, 5p+2] or None)."""
    lib = _abi.load()
    off = np.ascontiguousarray(row_offsets, dtype=np.int64)
    yv = np.ascontiguousarray(y, dtype=np.float64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    wv = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    p = len(cols)
    G = len(off) - 1
    N = len(yv)
    if any(len(c) != N for c in cols) or (wv is not None and len(wv) != N):
        raise ValueError("every column must have y's length")
    core = np.empty((G, p + 6), dtype=np.float64)
    inf = np.empty((G, 5 * p + 2), dtype=np.float64) if options.compute_inference else None
    colp = (_DP * max(p, 1))(*[c.ctypes.data_as(_DP) for c in cols])
    err = _abi.AnofoxError()
    ok = lib.anofox_hip_fit_batch_host(
        ctx._h if ctx is not None else None, G, p, N, off.ctypes.data_as(C.POINTER(C.c_int64)),
        yv.ctypes.data_as(_DP), colp, None if wv is None else wv.ctypes.data_as(_DP), options,
        core.ctypes.data_as(_DP), None if inf is None else inf.ctypes.data_as(_DP), C.byref(err))
    if not ok:
        raise AnofoxStatsError(err.code, err.text())
    return core, inf


def fit_predict_batch_host(row_offsets, y, x_cols: Sequence, w, options: _abi.AnofoxHipBatchOptions, train_counts=None,
                           ctx: Optional[Context] = None):
    """numpy in, numpy out: (core[G, p+6], pred[N, 3] = yhat / yhat_lower / yhat_upper, NaN = SQL NULL)."""
    lib = _abi.load()
    off = np.ascontiguousarray(row_offsets, dtype=np.int64)
    yv = np.ascontiguousarray(y, dtype=np.float64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    wv = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    tc = None if train_counts is None else np.ascontiguousarray(train_counts, dtype=np.int64)
    p, G, N = len(cols), len(off) - 1, len(yv)
    core = np.empty((G, p + 6), dtype=np.float64)
    pred = np.empty((N, 3), dtype=np.float64)
    colp = (_DP * max(p, 1))(*[c.ctypes.data_as(_DP) for c in cols])
    err = _abi.AnofoxError()
    ok = lib.anofox_hip_fit_predict_batch_host(
        ctx._h if ctx is not None else None, G, p, N, off.ctypes.data_as(C.POINTER(C.c_int64)), yv.ctypes.data_as(_DP),
        colp, None if wv is None else wv.ctypes.data_as(_DP),
        None if tc is None else tc.ctypes.data_as(C.POINTER(C.c_int64)), options, core.ctypes.data_as(_DP),
        pred.ctypes.data_as(_DP), C.byref(err))
    if not ok:
        raise AnofoxStatsError(err.code, err.text())
    return core, pred


def information_criteria_host(core, options: _abi.AnofoxHipBatchOptions, ctx: Optional[Context] = None):
    """numpy fit records [G, p+6] -> out[G, 3] = {rss, aic, bic} (anofox_hip_information_criteria_batch_host)."""
    lib = _abi.load()
    c = np.ascontiguousarray(core, dtype=np.float64)
    G, p = c.shape[0], c.shape[1] - 6
    out = np.empty((G, 3), dtype=np.float64)
    err = _abi.AnofoxError()
    ok = lib.anofox_hip_information_criteria_batch_host(ctx._h if ctx is not None else None, G, p, c.ctypes.data_as(_DP), options,
                                                        out.ctypes.data_as(_DP), C.byref(err))
    if not ok:
        raise AnofoxStatsError(err.code, err.text())
    return out


FRAME_UNBOUNDED = 2 ** 63 - 1        # ANOFOX_HIP_FRAME_UNBOUNDED


def _frame(frame) -> _abi.AnofoxHipWindowFrame:
    """(start, end) in rows PRECEDING the current row (negative = FOLLOWING); start None = UNBOUNDED PRECEDING,
    end None = UNBOUNDED FOLLOWING."""
    start, end = frame
    return _abi.AnofoxHipWindowFrame(FRAME_UNBOUNDED if start is None else int(start),
                                     -FRAME_UNBOUNDED if end is None else int(end))


def fit_predict_expanding_host(row_offsets, y, x_cols: Sequence, w, options: _abi.AnofoxHipBatchOptions,
                               ctx: Optional[Context] = None):
    """numpy in, numpy out: pred[N, 3] — prediction of x_e from the fit on rows 0..e of its partition."""
    return fit_predict_window_host(row_offsets, y, x_cols, w, options, (None, 0), ctx=ctx)


def fit_predict_window_host(row_offsets, y, x_cols: Sequence, w, options: _abi.AnofoxHipBatchOptions,
                            frame=(None, 0), ctx: Optional[Context] = None):
    """numpy in, numpy out: pred[N, 3] of the window functions over ROWS BETWEEN frame[0] PRECEDING AND frame[1]
    PRECEDING (negative = FOLLOWING; frame[0] None = UNBOUNDED PRECEDING, frame[1] 0 = CURRENT ROW, None = UNBOUNDED
    FOLLOWING)."""
    lib = _abi.load()
    off = np.ascontiguousarray(row_offsets, dtype=np.int64)
    yv = np.ascontiguousarray(y, dtype=np.float64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    wv = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    p, G, N = len(cols), len(off) - 1, len(yv)
    pred = np.empty((N, 3), dtype=np.float64)
    colp = (_DP * max(p, 1))(*[c.ctypes.data_as(_DP) for c in cols])
    err = _abi.AnofoxError()
    ok = lib.anofox_hip_fit_predict_window_host(
        ctx._h if ctx is not None else None, G, p, N, off.ctypes.data_as(C.POINTER(C.c_int64)), yv.ctypes.data_as(_DP),
        colp, None if wv is None else wv.ctypes.data_as(_DP), _frame(frame), options, pred.ctypes.data_as(_DP),
        C.byref(err))
    if not ok:
        raise AnofoxStatsError(err.code, err.text())
    return pred


def fit_predict_frames_host(y, x_cols: Sequence, w, frame_lo, frame_hi, options: _abi.AnofoxHipBatchOptions,
                            ctx: Optional[Context] = None):
    """Window fit + predict over explicit frames: frame of row e = rows [frame_lo[e], frame_hi[e]).  numpy in / out:
    pred[N, 3]."""
    lib = _abi.load()
    yv = np.ascontiguousarray(y, dtype=np.float64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    wv = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    lo = np.ascontiguousarray(frame_lo, dtype=np.int64)
    hi = np.ascontiguousarray(frame_hi, dtype=np.int64)
    p, N = len(cols), len(yv)
    pred = np.empty((N, 3), dtype=np.float64)
    colp = (_DP * max(p, 1))(*[c.ctypes.data_as(_DP) for c in cols])
    err = _abi.AnofoxError()
    ok = lib.anofox_hip_fit_predict_frames_host(
        ctx._h if ctx is not None else None, N, p, yv.ctypes.data_as(_DP), colp, None if wv is None else wv.ctypes.data_as(_DP),
        lo.ctypes.data_as(C.POINTER(C.c_int64)), hi.ctypes.data_as(C.POINTER(C.c_int64)), options, pred.ctypes.data_as(_DP),
        C.byref(err))
    if not ok:
        raise AnofoxStatsError(err.code, err.text())
    return pred


def vif_batch_host(row_offsets, x_cols: Sequence, ctx: Optional[Context] = None):
    """numpy in, numpy out: out[G, p+1] = {vif[p], status} (status 100 = fewer than 3 rows -> SQL NULL)."""
    lib = _abi.load()
    off = np.ascontiguousarray(row_offsets, dtype=np.int64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    p, G = len(cols), len(off) - 1
    N = len(cols[0]) if cols else 0
    out = np.empty((G, p + 1), dtype=np.float64)
    colp = (_DP * max(p, 1))(*[c.ctypes.data_as(_DP) for c in cols])
    err = _abi.AnofoxError()
    ok = lib.anofox_hip_vif_batch_host(ctx._h if ctx is not None else None, G, p, N, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                       colp, out.ctypes.data_as(_DP), C.byref(err))
    if not ok:
        raise AnofoxStatsError(err.code, err.text())
    return out


def residuals_batch_host(row_offsets, y, y_hat, x_cols: Sequence = (), rse=None, include_studentized: bool = True,
                         drop_nan_rows: bool = True, ctx: Optional[Context] = None):
    """numpy in, numpy out: (out[N, 4] = raw / standardized / studentized / leverage, group[G, 2] = rows used / flags)."""
    lib = _abi.load()
    off = np.ascontiguousarray(row_offsets, dtype=np.int64)
    yv = np.ascontiguousarray(y, dtype=np.float64)
    yh = np.ascontiguousarray(y_hat, dtype=np.float64)
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in x_cols]
    p, G, N = len(cols), len(off) - 1, len(yv)
    out = np.empty((N, 4), dtype=np.float64)
    group = np.empty((G, 2), dtype=np.float64)
    colp = (_DP * max(p, 1))(*[c.ctypes.data_as(_DP) for c in cols])
    rse_arr = None if rse is None else np.ascontiguousarray(rse, dtype=np.float64)
    err = _abi.AnofoxError()
    ok = lib.anofox_hip_residuals_batch_host(
        ctx._h if ctx is not None else None, G, p, N, off.ctypes.data_as(C.POINTER(C.c_int64)), yv.ctypes.data_as(_DP),
        yh.ctypes.data_as(_DP), colp if p else None, None if rse_arr is None else rse_arr.ctypes.data_as(_DP),
        bool(include_studentized), bool(drop_nan_rows), out.ctypes.data_as(_DP), group.ctypes.data_as(_DP), C.byref(err))
    if not ok:
        raise AnofoxStatsError(err.code, err.text())
    return out, group
