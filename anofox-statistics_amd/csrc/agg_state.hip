// agg_state.hip — anofox_hip_agg_state_*: the aggregate state of {ols,ridge,wls}_fit_agg kept on the GPU.
//
// The reference's aggregate state is a per-group row buffer on the host (src/aggregate_functions/
// ols_aggregate.cpp:19-42) that Update appends to row by row (:120-186), Combine concatenates (:189-234) and
// Finalize hands to one FFI call per group (:249-338).  Here the state of ALL groups of a query is one object
// holding an O(p^2) moment record per "slot" (= one DuckDB aggregate state) in HBM:
//   update    a chunk of rows in arrival order -> ingest.hip folds it into the records (rows are not kept);
//   combine   pairs (source slot, target slot) -> the source record is merged into the target and emptied;
//   finalize  the unchanged solve kernel of the batch path (solve_narrow.hip) over the records.
// Optionally (anofox_hip_agg_state_retain_rows) Update also appends the chunk to a row log in HBM (rowlog.hip), and
// Finalize refits from it exactly the groups its solve queued for refinement, through the unchanged batch path.
// Designs wider than 8 features and heteroskedasticity-consistent errors have no moment record to stream into: such a
// state keeps ONLY the row log ("log-only": the reference's row buffers, in HBM) and Finalize runs the batch path on
// all of it — same entry points, any p <= 128, any hc_type.
// Host chunks are staged through two device buffers on a copy stream, so the H2D copy of chunk k + 1 overlaps
// the kernels of chunk k; update() returns once its inputs have been copied (the caller may reuse them).
#include <stdlib.h>

#include <memory>
#include <unordered_set>
#include <vector>

#include <algorithm>

#include "context.h"

using namespace anofox;
using namespace anofox::host;

struct AnofoxHipAggState {
	AnofoxHipContext *ctx = nullptr;
	size_t p = 0;
	AnofoxHipBatchOptions opt{};
	std::mutex mu;
	// per-slot state
	double *moments = nullptr;
	int64_t *n_accum = nullptr;
	int32_t *run_start = nullptr, *run_end = nullptr;
	int64_t capacity = 0; // slots allocated
	int64_t n_slots = 0;  // slots in use (largest count announced by the caller)
	int64_t rows = 0;     // rows passed to update so far
	// per-pass scratch (one set: the passes of one state are serialised on the context's stream)
	void *scratch = nullptr;
	size_t scratch_bytes = 0;
	size_t sort_temp_bytes = 0;
	int32_t *counters = nullptr; // [0] runs of the current pass, [1] sticky out-of-range flag (own small allocation)
	// staging of host chunks
	struct Stage {
		void *buf = nullptr;
		size_t bytes = 0;
		hipEvent_t copied = nullptr, done = nullptr;
	} stage[2];
	int next_stage = 0;
	hipStream_t copy_stream = nullptr;
	void *pair_buf = nullptr; // combine: src | dst
	size_t pair_bytes = 0;
	// optional row log (anofox_hip_agg_state_retain_rows): slabs in arrival order
	bool log_only = false;     // p > 8 or HC errors: no moments, the row log IS the state
	bool retain = false;       // asked for
	bool log_dropped = false;  // ... and given up because the budget was exceeded
	size_t log_budget = 0, log_bytes = 0;            // HBM part of the log
	size_t log_host_budget = 0, log_host_bytes = 0;  // page-locked host part (the spill beyond the HBM budget)
	int64_t log_rows = 0;
	std::vector<RowLogSlab> slabs;
	void *refit_idx = nullptr, *refit_rows = nullptr; // Finalize's refit scratch
	size_t refit_idx_bytes = 0, refit_rows_bytes = 0;
	void *remap_buf = nullptr;
	size_t remap_bytes = 0;
};

namespace {
void slab_release(RowLogSlab &sl) {
	void *parts[] = {sl.x, sl.y, sl.w, sl.slot, sl.valid};
	for (void *q : parts)
		if (q) (void)(sl.on_host ? hipHostFree(q) : hipFree(q));
	sl = RowLogSlab{};
}
} // namespace

namespace anofox {
namespace host {
// Release everything the state holds on the device and cut it loose from its context (called by
// anofox_hip_agg_state_destroy and by anofox_hip_context_destroy for the states that outlive their context).
void agg_state_detach(AnofoxHipAggState *s) {
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!s->ctx) return;
	(void)hipSetDevice(s->ctx->device);
	(void)hipStreamSynchronize(s->ctx->stream);
	if (s->copy_stream) {
		(void)hipStreamSynchronize(s->copy_stream);
		(void)hipStreamDestroy(s->copy_stream);
		s->copy_stream = nullptr;
	}
	for (auto &st : s->stage) {
		if (st.buf) (void)hipFree(st.buf);
		if (st.copied) (void)hipEventDestroy(st.copied);
		if (st.done) (void)hipEventDestroy(st.done);
		st = AnofoxHipAggState::Stage();
	}
	for (auto &sl : s->slabs) slab_release(sl);
	s->slabs.clear();
	s->log_bytes = 0;
	s->log_host_bytes = 0;
	s->log_rows = 0;
	void **bufs[] = {(void **)&s->moments, (void **)&s->n_accum, (void **)&s->run_start, (void **)&s->run_end, &s->scratch,
	                 (void **)&s->counters, &s->pair_buf, &s->refit_idx, &s->refit_rows, &s->remap_buf};
	for (void **b : bufs) {
		if (*b) (void)hipFree(*b);
		*b = nullptr;
	}
	s->capacity = 0;
	s->scratch_bytes = s->pair_bytes = s->refit_idx_bytes = s->refit_rows_bytes = s->remap_bytes = 0;
	s->ctx = nullptr;
}
} // namespace host
} // namespace anofox

namespace {

// every entry point: the state must still be attached to a live context
bool attached(AnofoxHipAggState *s, AnofoxError *e) {
	if (s && !s->ctx) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT, "the aggregate state's context has been destroyed");
		return false;
	}
	return true;
}

bool state_reserve(AnofoxHipAggState *s, int64_t n_slots, AnofoxError *e) {
	if (n_slots > (int64_t)0x7fffffff) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "too many aggregate states (limit 2^31 - 1)"); return false; }
	if (n_slots > s->n_slots) s->n_slots = n_slots;
	if (s->log_only) return true; // nothing is kept per slot
	if (n_slots <= s->capacity) return true;
	int64_t cap = s->capacity * 2;
	if (cap < n_slots) cap = n_slots;
	if (cap < 1024) cap = 1024;
	if (cap > (int64_t)0x7fffffff) cap = 0x7fffffff;
	const size_t rec = (size_t)moment_record_len((int)s->p) * sizeof(double);
	double *m = nullptr;
	int64_t *na = nullptr;
	int32_t *rs = nullptr, *re = nullptr;
	if (hipMalloc((void **)&m, (size_t)cap * rec) != hipSuccess || hipMalloc((void **)&na, (size_t)cap * sizeof(int64_t)) != hipSuccess ||
	    hipMalloc((void **)&rs, (size_t)cap * sizeof(int32_t)) != hipSuccess || hipMalloc((void **)&re, (size_t)cap * sizeof(int32_t)) != hipSuccess) {
		(void)hipGetLastError();
		if (m) (void)hipFree(m);
		if (na) (void)hipFree(na);
		if (rs) (void)hipFree(rs);
		if (re) (void)hipFree(re);
		set_error(e, ANOFOX_ERROR_ALLOCATION_FAILURE, "hipMalloc failed for the aggregate state");
		return false;
	}
	hipStream_t st = s->ctx->stream;
	const size_t old = (size_t)s->capacity;
	bool bad = false;
	if (old) {
		bad = bad || hip_fail(hipMemcpyAsync(m, s->moments, old * rec, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync", e);
		bad = bad || hip_fail(hipMemcpyAsync(na, s->n_accum, old * sizeof(int64_t), hipMemcpyDeviceToDevice, st), "hipMemcpyAsync", e);
	}
	// an empty slot is all zeros (cnt = 0)
	bad = bad || hip_fail(hipMemsetAsync((char *)m + old * rec, 0, ((size_t)cap - old) * rec, st), "hipMemsetAsync", e);
	bad = bad || hip_fail(hipMemsetAsync(na + old, 0, ((size_t)cap - old) * sizeof(int64_t), st), "hipMemsetAsync", e);
	bad = bad || hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", e);
	if (bad) {
		(void)hipFree(m); (void)hipFree(na); (void)hipFree(rs); (void)hipFree(re);
		return false;
	}
	if (s->moments) (void)hipFree(s->moments);
	if (s->n_accum) (void)hipFree(s->n_accum);
	if (s->run_start) (void)hipFree(s->run_start);
	if (s->run_end) (void)hipFree(s->run_end);
	s->moments = m;
	s->n_accum = na;
	s->run_start = rs;
	s->run_end = re;
	s->capacity = cap;
	return true;
}

// scratch of one pass: keys_in | keys_out | rows_out | run_list | piece table | sort temp
bool state_scratch(AnofoxHipAggState *s, AnofoxError *e) {
	if (s->scratch) return true;
	const size_t b_n = align_up((size_t)kIngestChunkRows * sizeof(uint32_t), 256);
	const size_t b_pt = align_up(ingest_piece_table_bytes((int)s->p), 256);
	s->sort_temp_bytes = align_up(ingest_sort_temp_bytes(kIngestChunkRows), 256);
	return ensure_buffer(&s->scratch, &s->scratch_bytes, 4 * b_n + b_pt + s->sort_temp_bytes, "ingest scratch", e);
}

size_t log_row_bytes(const AnofoxHipAggState *s) {
	return s->p * sizeof(double) + sizeof(double) + (s->opt.model == ANOFOX_HIP_MODEL_WLS ? sizeof(double) : 0) + sizeof(uint32_t) + 1;
}

void log_free(AnofoxHipAggState *s) {
	(void)hipStreamSynchronize(s->ctx->stream);
	for (auto &sl : s->slabs) slab_release(sl);
	s->slabs.clear();
	s->log_bytes = 0;
	s->log_host_bytes = 0;
	s->log_rows = 0;
}

// A new slab at the end of the log with room for at least `need` rows (1 = whatever the growth policy gives).  Slabs grow
// geometrically from 64 Ki rows to 16 Mi rows or 2 GiB; HBM while its budget lasts, then page-locked host memory (the same kernels
// read it over PCIe), then nothing: *dropped = the budgets (or the memory itself) are exhausted and the log was given up — an
// error only for a log-only state, whose log IS the state.
bool log_grow(AnofoxHipAggState *s, int64_t need, bool *dropped, AnofoxError *e) {
	*dropped = false;
	const bool weighted = s->opt.model == ANOFOX_HIP_MODEL_WLS;
	const size_t p = s->p, rb = log_row_bytes(s);
	int64_t cap = s->slabs.empty() ? 65536 : s->slabs.back().cap * 2;
	if (cap > (int64_t)1 << 24) cap = (int64_t)1 << 24;
	if ((size_t)cap * rb > ((size_t)2 << 30) && cap > 65536) { // never more than 2 GiB per slab (1 KB rows at p = 128)
		cap = (int64_t)(((size_t)2 << 30) / rb);
		if (cap < 65536) cap = 65536;
	}
	auto fit_cap = [&](size_t left) { // rows of the next slab within `left` bytes; 0 = `need` rows do not fit
		int64_t c = cap;
		if ((size_t)c * rb > left) c = (int64_t)(left / rb);
		if (c < need && (size_t)need * rb <= left) c = need;
		return (c <= 0 || c < need) ? (int64_t)0 : c;
	};
	auto alloc_slab = [&](int64_t c, bool on_host, RowLogSlab *out) {
		RowLogSlab nsl{};
		nsl.cap = c;
		nsl.first_row = s->log_rows;
		nsl.on_host = on_host ? 1 : 0;
		auto get = [&](void **q, size_t bytes) {
			return (on_host ? hipHostMalloc(q, bytes, hipHostMallocDefault) : hipMalloc(q, bytes)) == hipSuccess;
		};
		const bool ok = get((void **)&nsl.x, (size_t)c * p * sizeof(double)) && get((void **)&nsl.y, (size_t)c * sizeof(double)) &&
		                (!weighted || get((void **)&nsl.w, (size_t)c * sizeof(double))) &&
		                get((void **)&nsl.slot, (size_t)c * sizeof(uint32_t)) && get((void **)&nsl.valid, (size_t)c);
		if (!ok) {
			(void)hipGetLastError();
			slab_release(nsl);
			return false;
		}
		*out = nsl;
		return true;
	};
	RowLogSlab sl{};
	bool have = false;
	const int64_t cap_dev = fit_cap(s->log_budget > s->log_bytes ? s->log_budget - s->log_bytes : 0);
	if (cap_dev > 0 && alloc_slab(cap_dev, false, &sl)) {
		have = true;
		s->log_bytes += (size_t)cap_dev * rb;
	} else {
		const int64_t cap_host = fit_cap(s->log_host_budget > s->log_host_bytes ? s->log_host_budget - s->log_host_bytes : 0);
		if (cap_host > 0 && alloc_slab(cap_host, true, &sl)) {
			have = true;
			s->log_host_bytes += (size_t)cap_host * rb;
		}
	}
	if (!have) { // both budgets (or the memory itself) are exhausted: stop retaining
		if (s->log_only) {
			set_error(e, ANOFOX_ERROR_ALLOCATION_FAILURE,
			          "the aggregate state's row log exceeds its budgets (anofox_hip_agg_state_retain_rows / _retain_rows_host)");
			return false;
		}
		log_free(s);
		s->log_dropped = true;
		*dropped = true;
		return true;
	}
	s->slabs.push_back(sl);
	return true;
}

// Append one chunk (device pointers, stream-ordered after whatever produced them) to the row log.  Exceeding the
// budget is not an error: the log is dropped and Finalize reports the unrefined groups as it does without one.
bool log_append(AnofoxHipAggState *s, int64_t n, const uint32_t *d_slot, const double *d_y, const double *d_x, const double *d_w,
                const uint8_t *d_valid, AnofoxError *e) {
	if (!s->retain || s->log_dropped || n <= 0) return true;
	const bool weighted = s->opt.model == ANOFOX_HIP_MODEL_WLS;
	const size_t p = s->p;
	hipStream_t st = s->ctx->stream;
	int64_t done = 0;
	while (done < n) {
		if (s->slabs.empty() || s->slabs.back().rows == s->slabs.back().cap) {
			bool dropped = false;
			if (!log_grow(s, n - done, &dropped, e)) return false;
			if (dropped) return true;
		}
		RowLogSlab &sl = s->slabs.back();
		const int64_t m = (n - done) < (sl.cap - sl.rows) ? (n - done) : (sl.cap - sl.rows);
		const size_t at = (size_t)sl.rows;
		bool bad = hip_fail(hipMemcpyAsync(sl.x + at * p, d_x + (size_t)done * p, (size_t)m * p * sizeof(double), hipMemcpyDefault, st), "row log", e);
		bad = bad || hip_fail(hipMemcpyAsync(sl.y + at, d_y + done, (size_t)m * sizeof(double), hipMemcpyDefault, st), "row log", e);
		if (weighted) bad = bad || hip_fail(hipMemcpyAsync(sl.w + at, d_w + done, (size_t)m * sizeof(double), hipMemcpyDefault, st), "row log", e);
		bad = bad || hip_fail(hipMemcpyAsync(sl.slot + at, d_slot + done, (size_t)m * sizeof(uint32_t), hipMemcpyDefault, st), "row log", e);
		if (d_valid)
			bad = bad || hip_fail(hipMemcpyAsync(sl.valid + at, d_valid + done, (size_t)m, hipMemcpyDefault, st), "row log", e);
		else
			bad = bad || hip_fail(hipMemsetAsync(sl.valid + at, 1, (size_t)m, st), "row log", e);
		if (bad) return false;
		sl.rows += m;
		s->log_rows += m;
		done += m;
	}
	return true;
}

// A Combine that preserves its sources (DuckDB's AggregateCombineType::PRESERVE_INPUT: window segment trees): the targets get
// their own copies of the sources' logged rows, as the reference's Combine copies the row buffers (ols_aggregate.cpp:224-233), so
// that Finalize can refit the frames its moments cannot resolve (r3 gave the log up here: every exactly fitting frame of a
// windowed aggregate came back NULL, and log-only states refused the call).  Host arrays; synchronises the stream.
bool log_duplicate(AnofoxHipAggState *s, int64_t n_pairs, const uint32_t *src, const uint32_t *dst, AnofoxError *e) {
	if (!(s->retain || s->log_only) || s->log_dropped || s->log_rows == 0 || n_pairs <= 0) return true;
	hipStream_t st = s->ctx->stream;
	// the pairs as a map source -> its targets
	std::vector<int64_t> order((size_t)n_pairs);
	for (int64_t i = 0; i < n_pairs; ++i) order[(size_t)i] = i;
	std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return src[a] < src[b]; });
	std::vector<uint32_t> usrc, utgt;
	std::vector<int32_t> uoff;
	for (int64_t i : order) {
		if (src[i] == dst[i]) continue;
		if (usrc.empty() || usrc.back() != src[i]) {
			usrc.push_back(src[i]);
			uoff.push_back((int32_t)utgt.size());
		}
		utgt.push_back(dst[i]);
	}
	if (usrc.empty()) return true;
	uoff.push_back((int32_t)utgt.size());
	const int m = (int)usrc.size();
	int64_t n_tiles = 0;
	std::vector<int64_t> src_rows;
	for (auto &sl : s->slabs) {
		n_tiles += rowlog_dup_tiles(sl.rows);
		src_rows.push_back(sl.rows);
	}
	const size_t b_u = align_up((size_t)m * sizeof(uint32_t), 256), b_o = align_up((size_t)(m + 1) * sizeof(int32_t), 256);
	const size_t b_t = align_up(utgt.size() * sizeof(uint32_t), 256), b_c = align_up((size_t)(n_tiles + 1) * sizeof(int64_t), 256);
	if (hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", e)) return false; // (the buffer may still be read by an earlier remap)
	if (!ensure_buffer(&s->remap_buf, &s->remap_bytes, b_u + b_o + b_t + b_c, "row log duplication", e)) return false;
	char *base = (char *)s->remap_buf;
	uint32_t *d_usrc = (uint32_t *)base, *d_utgt = (uint32_t *)(base + b_u + b_o);
	int32_t *d_uoff = (int32_t *)(base + b_u);
	int64_t *d_cnt = (int64_t *)(base + b_u + b_o + b_t);
	bool bad = hip_fail(hipMemcpyAsync(d_usrc, usrc.data(), (size_t)m * sizeof(uint32_t), hipMemcpyHostToDevice, st), "H2D", e);
	bad = bad || hip_fail(hipMemcpyAsync(d_uoff, uoff.data(), (size_t)(m + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st), "H2D", e);
	bad = bad || hip_fail(hipMemcpyAsync(d_utgt, utgt.data(), utgt.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st), "H2D", e);
	bad = bad || hip_fail(launch_rowlog_dup_count(s->slabs.data(), (int)s->slabs.size(), d_usrc, d_uoff, d_utgt, m, d_cnt, n_tiles, st),
	                      "row log duplication (count)", e);
	int64_t total = 0;
	bad = bad || hip_fail(hipMemcpyAsync(&total, d_cnt + n_tiles, sizeof total, hipMemcpyDeviceToHost, st), "D2H", e);
	bad = bad || hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", e);
	if (bad) return false;
	if (total == 0) return true;
	if (total > ((int64_t)1 << 24)) { // more copies than one slab may hold: not a window frame's Combine
		if (s->log_only) {
			set_error(e, ANOFOX_ERROR_INVALID_INPUT, "combine with preserved sources: more than 2^24 rows to copy in one call");
			return false;
		}
		log_free(s);
		s->log_dropped = true;
		return true;
	}
	if (s->slabs.back().cap - s->slabs.back().rows < total) {
		bool dropped = false;
		if (!log_grow(s, total, &dropped, e)) return false;
		if (dropped) return true;
		src_rows.push_back(0); // (the new slab holds no source rows)
	}
	RowLogSlab &dsl = s->slabs.back();
	if (hip_fail(launch_rowlog_dup_fill(s->slabs.data(), src_rows.data(), (int)s->slabs.size(), (int)s->p, s->opt.model == ANOFOX_HIP_MODEL_WLS ? 1 : 0,
	                                    d_usrc, d_uoff, d_utgt, m, d_cnt, dsl, dsl.rows, st),
	             "row log duplication (fill)", e))
		return false;
	dsl.rows += total;
	s->log_rows += total;
	return !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", e);
}

// one pass (<= kIngestChunkRows rows) on device-resident inputs
bool run_pass(AnofoxHipAggState *s, int64_t n, const uint32_t *d_slot, const double *d_y, const double *d_x, const double *d_w,
              const uint8_t *d_valid, AnofoxError *e) {
	if (s->log_only) return log_append(s, n, d_slot, d_y, d_x, d_w, d_valid, e);
	if (!state_scratch(s, e)) return false;
	const size_t b_n = align_up((size_t)kIngestChunkRows * sizeof(uint32_t), 256);
	const size_t b_pt = align_up(ingest_piece_table_bytes((int)s->p), 256);
	char *base = (char *)s->scratch;
	IngestArgs a;
	memset(&a, 0, sizeof a);
	a.slot = d_slot;
	a.y = d_y;
	a.x = d_x;
	a.w = d_w;
	a.valid = d_valid;
	a.n = n;
	a.p = (int)s->p;
	a.weighted = s->opt.model == ANOFOX_HIP_MODEL_WLS ? 1 : 0;
	a.center = s->opt.fit_intercept ? 1 : 0;
	a.moments = s->moments;
	a.n_accum = s->n_accum;
	a.n_slots = s->n_slots;
	a.keys_in = (uint32_t *)base;
	a.keys_out = (uint32_t *)(base + b_n);
	a.rows_out = (uint32_t *)(base + 2 * b_n);
	a.run_list = (uint32_t *)(base + 3 * b_n);
	a.piece_table = base + 4 * b_n;
	a.sort_temp = base + 4 * b_n + b_pt;
	a.sort_temp_bytes = s->sort_temp_bytes;
	a.run_start = s->run_start;
	a.run_end = s->run_end;
	a.counters = s->counters;
	hipStream_t st = s->ctx->stream;
	if (hip_fail(hipMemsetAsync(a.counters, 0, sizeof(int32_t), st), "hipMemsetAsync", e)) return false;
	if (hip_fail(hipMemsetAsync(a.piece_table, 0, 64, st), "hipMemsetAsync", e)) return false; // PieceHeader
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (s->ctx->timing) {
		e0 = get_event(s->ctx);
		e1 = get_event(s->ctx);
		(void)hipEventRecord(e0, st);
	}
	if (hip_fail(launch_ingest_chunk(a, st), "ingest kernel launch", e)) return false;
	if (s->ctx->timing) {
		(void)hipEventRecord(e1, st);
		s->ctx->acc_events.emplace_back(e0, e1); // the ingest pass is this path's "accumulate" stage
	}
	return log_append(s, n, d_slot, d_y, d_x, d_w, d_valid, e);
}

bool validate_update(AnofoxHipAggState *s, int64_t n_rows, int64_t n_slots, const void *slot, const void *y, const void *x, const void *w,
                     AnofoxError *e) {
	if (!s) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "state is NULL"); return false; }
	if (n_rows < 0 || n_slots < 0) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "negative n_rows or n_slots"); return false; }
	if (n_rows > 0 && (!slot || !y || !x)) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "slot, y or x is NULL"); return false; }
	if (n_rows > 0 && s->opt.model == ANOFOX_HIP_MODEL_WLS && !w) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "weights is NULL"); return false; }
	if (n_rows > 0 && n_slots == 0) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "rows without any aggregate state (n_slots == 0)"); return false; }
	return true;
}

} // namespace

extern "C" {

bool anofox_hip_agg_state_create(AnofoxHipContext *ctx, size_t n_features, AnofoxHipBatchOptions options, int64_t initial_slots,
                                 AnofoxHipAggState **out_state, AnofoxError *out_error) {
	reset_error(out_error);
	if (!out_state) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "out_state is NULL"); return false; }
	*out_state = nullptr;
	if (!ctx) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	if (n_features == 0 || n_features > anofox_hip_agg_state_max_features()) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT,
		          "streaming aggregate states support 1.." + std::to_string(anofox_hip_agg_state_max_features()) +
		              " features (got " + std::to_string(n_features) + ")");
		return false;
	}
	if (options.model != ANOFOX_HIP_MODEL_OLS && options.model != ANOFOX_HIP_MODEL_RIDGE && options.model != ANOFOX_HIP_MODEL_WLS) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "unknown model");
		return false;
	}
	if ((int)options.hc_type < ANOFOX_HC_NONE || (int)options.hc_type > ANOFOX_HC_HC3) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "unknown hc_type");
		return false;
	}
	if (initial_slots < 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "negative initial_slots"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	auto *s = new (std::nothrow) AnofoxHipAggState();
	if (!s) { set_error(out_error, ANOFOX_ERROR_ALLOCATION_FAILURE, "state allocation failed"); return false; }
	s->ctx = ctx;
	s->p = n_features;
	s->opt = options;
	// no moment record for these: the rows themselves are the state (HC errors need a second pass over them)
	s->log_only = n_features > (size_t)kNarrowMaxP ||
	              (options.compute_inference && options.hc_type != ANOFOX_HC_NONE && options.model != ANOFOX_HIP_MODEL_RIDGE);
	if (s->log_only) {
		s->retain = true;
		s->log_budget = ~(size_t)0; // until anofox_hip_agg_state_retain_rows caps it
	}
	bool ok = !hip_fail(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking), "hipStreamCreate", out_error);
	for (int k = 0; ok && k < 2; ++k) {
		ok = !hip_fail(hipEventCreateWithFlags(&s->stage[k].copied, hipEventDisableTiming), "hipEventCreate", out_error) &&
		     !hip_fail(hipEventCreateWithFlags(&s->stage[k].done, hipEventDisableTiming), "hipEventCreate", out_error);
	}
	ok = ok && !hip_fail(hipMalloc((void **)&s->counters, 256), "hipMalloc", out_error);
	ok = ok && !hip_fail(hipMemsetAsync(s->counters, 0, 256, ctx->stream), "hipMemsetAsync", out_error);
	if (ok && initial_slots > 0) {
		ok = state_reserve(s, initial_slots, out_error);
		s->n_slots = 0; // reserved, not yet in use
	}
	if (!ok) {
		agg_state_detach(s);
		delete s;
		return false;
	}
	ctx->agg_states.push_back(s);
	*out_state = s;
	return true;
}

void anofox_hip_agg_state_destroy(AnofoxHipAggState *s) {
	if (!s) return;
	AnofoxHipContext *ctx = nullptr;
	{
		std::lock_guard<std::mutex> lk0(s->mu);
		ctx = s->ctx;
	}
	if (ctx) {
		{
			std::lock_guard<std::mutex> lk(ctx->mu);
			auto &v = ctx->agg_states;
			for (size_t i = 0; i < v.size(); ++i)
				if (v[i] == s) { v.erase(v.begin() + (long)i); break; }
		}
		agg_state_detach(s);
	}
	delete s;
}

size_t anofox_hip_agg_state_max_features(void) { return (size_t)kWideMaxP; }

int64_t anofox_hip_agg_state_slots(const AnofoxHipAggState *s) { return s ? s->n_slots : 0; }
int64_t anofox_hip_agg_state_rows(const AnofoxHipAggState *s) { return s ? s->rows : 0; }

bool anofox_hip_agg_state_reserve(AnofoxHipAggState *s, int64_t n_slots, AnofoxError *out_error) {
	reset_error(out_error);
	if (!s || n_slots < 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "state is NULL or n_slots negative"); return false; }
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	std::lock_guard<std::mutex> lk(s->ctx->mu);
	if (hip_fail(hipSetDevice(s->ctx->device), "hipSetDevice", out_error)) return false;
	return state_reserve(s, n_slots, out_error);
}

bool anofox_hip_agg_state_retain_rows(AnofoxHipAggState *s, size_t max_bytes, AnofoxError *out_error) {
	reset_error(out_error);
	if (!s) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "state is NULL"); return false; }
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	if (s->rows > 0) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "retain_rows has to be called before the first update");
		return false;
	}
	if (s->log_only) { // the log is the state: max_bytes only caps it (0 = no cap); exceeding it fails the update
		s->log_budget = max_bytes ? max_bytes : ~(size_t)0;
		return true;
	}
	s->log_budget = max_bytes;
	s->retain = s->log_budget > 0 || s->log_host_budget > 0;
	s->log_dropped = false;
	return true;
}

bool anofox_hip_agg_state_retain_rows_host(AnofoxHipAggState *s, size_t max_host_bytes, AnofoxError *out_error) {
	reset_error(out_error);
	if (!s) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "state is NULL"); return false; }
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	if (s->rows > 0) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "retain_rows_host has to be called before the first update");
		return false;
	}
	s->log_host_budget = max_host_bytes;
	if (!s->log_only) s->retain = s->log_budget > 0 || s->log_host_budget > 0; // (a log-only state always keeps its rows)
	return true;
}

int anofox_hip_agg_state_retaining(const AnofoxHipAggState *s) { return s && s->retain && !s->log_dropped ? 1 : 0; }
size_t anofox_hip_agg_state_retained_bytes(const AnofoxHipAggState *s) { return s ? s->log_bytes : 0; }
size_t anofox_hip_agg_state_retained_host_bytes(const AnofoxHipAggState *s) { return s ? s->log_host_bytes : 0; }

bool anofox_hip_agg_state_update_device(AnofoxHipAggState *s, int64_t n_rows, int64_t n_slots, const uint32_t *d_slot, const double *d_y,
                                        const double *d_x_rowmajor, const double *d_w, const uint8_t *d_valid, AnofoxError *out_error) {
	reset_error(out_error);
	if (!validate_update(s, n_rows, n_slots, d_slot, d_y, d_x_rowmajor, d_w, out_error)) return false;
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	std::lock_guard<std::mutex> lk(s->ctx->mu);
	if (hip_fail(hipSetDevice(s->ctx->device), "hipSetDevice", out_error)) return false;
	if (!state_reserve(s, n_slots, out_error)) return false;
	const bool weighted = s->opt.model == ANOFOX_HIP_MODEL_WLS;
	for (int64_t r0 = 0; r0 < n_rows; r0 += kIngestChunkRows) {
		const int64_t n = n_rows - r0 < kIngestChunkRows ? n_rows - r0 : kIngestChunkRows;
		if (!run_pass(s, n, d_slot + r0, d_y + r0, d_x_rowmajor + (size_t)r0 * s->p, weighted ? d_w + r0 : nullptr,
		              d_valid ? d_valid + r0 : nullptr, out_error))
			return false;
	}
	s->rows += n_rows;
	return true;
}

bool anofox_hip_agg_state_update_host(AnofoxHipAggState *s, int64_t n_rows, int64_t n_slots, const uint32_t *slot, const double *y,
                                      const double *x_rowmajor, const double *w, const uint8_t *valid, AnofoxError *out_error) {
	reset_error(out_error);
	if (!validate_update(s, n_rows, n_slots, slot, y, x_rowmajor, w, out_error)) return false;
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	std::lock_guard<std::mutex> lk(s->ctx->mu);
	if (hip_fail(hipSetDevice(s->ctx->device), "hipSetDevice", out_error)) return false;
	if (!state_reserve(s, n_slots, out_error)) return false;
	const bool weighted = s->opt.model == ANOFOX_HIP_MODEL_WLS;
	const size_t p = s->p;
	// staging layout of one chunk: x | y | w | slot | valid
	const size_t C = (size_t)kIngestStageRows;
	const size_t b_x = align_up(C * p * sizeof(double), 256), b_y = align_up(C * sizeof(double), 256);
	const size_t b_s = align_up(C * sizeof(uint32_t), 256), b_v = align_up(C, 256);
	const size_t total = b_x + 2 * b_y + b_s + b_v;
	hipStream_t st = s->ctx->stream, cs = s->copy_stream;
	for (int64_t r0 = 0; r0 < n_rows; r0 += kIngestStageRows) {
		const size_t n = (size_t)(n_rows - r0 < kIngestStageRows ? n_rows - r0 : kIngestStageRows);
		auto &sg = s->stage[s->next_stage];
		s->next_stage ^= 1;
		if (!sg.buf) {
			if (!ensure_buffer(&sg.buf, &sg.bytes, total, "ingest staging", out_error)) return false;
		} else if (hip_fail(hipStreamWaitEvent(cs, sg.done, 0), "hipStreamWaitEvent", out_error)) { // the pass that last read this buffer
			return false;
		}
		char *base = (char *)sg.buf;
		double *d_x = (double *)base, *d_y = (double *)(base + b_x), *d_w = (double *)(base + b_x + b_y);
		uint32_t *d_slot = (uint32_t *)(base + b_x + 2 * b_y);
		uint8_t *d_valid = (uint8_t *)(base + b_x + 2 * b_y + b_s);
		bool bad = hip_fail(hipMemcpyAsync(d_x, x_rowmajor + (size_t)r0 * p, n * p * sizeof(double), hipMemcpyHostToDevice, cs), "H2D x", out_error);
		bad = bad || hip_fail(hipMemcpyAsync(d_y, y + r0, n * sizeof(double), hipMemcpyHostToDevice, cs), "H2D y", out_error);
		if (weighted) bad = bad || hip_fail(hipMemcpyAsync(d_w, w + r0, n * sizeof(double), hipMemcpyHostToDevice, cs), "H2D w", out_error);
		bad = bad || hip_fail(hipMemcpyAsync(d_slot, slot + r0, n * sizeof(uint32_t), hipMemcpyHostToDevice, cs), "H2D slot", out_error);
		if (valid) bad = bad || hip_fail(hipMemcpyAsync(d_valid, valid + r0, n, hipMemcpyHostToDevice, cs), "H2D valid", out_error);
		bad = bad || hip_fail(hipEventRecord(sg.copied, cs), "hipEventRecord", out_error);
		bad = bad || hip_fail(hipStreamWaitEvent(st, sg.copied, 0), "hipStreamWaitEvent", out_error);
		if (bad) return false;
		if (!run_pass(s, (int64_t)n, d_slot, d_y, d_x, weighted ? d_w : nullptr, valid ? d_valid : nullptr, out_error)) return false;
		if (hip_fail(hipEventRecord(sg.done, st), "hipEventRecord", out_error)) return false;
	}
	// the caller may reuse its buffers on return: wait for the copies (not for the kernels)
	if (hip_fail(hipStreamSynchronize(cs), "hipStreamSynchronize", out_error)) return false;
	s->rows += n_rows;
	return true;
}

bool anofox_hip_agg_state_combine(AnofoxHipAggState *s, int64_t n_pairs, const uint32_t *source_slots, const uint32_t *target_slots,
                                  AnofoxError *out_error) {
	return anofox_hip_agg_state_combine_ex(s, n_pairs, source_slots, target_slots, false, out_error);
}

bool anofox_hip_agg_state_combine_ex(AnofoxHipAggState *s, int64_t n_pairs, const uint32_t *source_slots, const uint32_t *target_slots,
                                     bool preserve_sources, AnofoxError *out_error) {
	reset_error(out_error);
	if (!s || n_pairs < 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "state is NULL or n_pairs negative"); return false; }
	if (n_pairs == 0) return true;
	if (!source_slots || !target_slots) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "slot arrays are NULL"); return false; }
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	{
		// every pair is merged by its own wavefront: a target (or a slot that is both source and target) may appear once
		std::unordered_set<uint32_t> seen;
		seen.reserve((size_t)n_pairs * 2);
		for (int64_t i = 0; i < n_pairs; ++i) {
			const uint32_t a = source_slots[i], b = target_slots[i];
			if ((int64_t)a >= s->n_slots || (int64_t)b >= s->n_slots) {
				set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "combine: slot index out of range");
				return false;
			}
			if (a == b) continue;
			// (sources that are preserved are only read: the same one may feed several targets of a call)
			const bool dup = preserve_sources ? (!seen.insert(b).second) : (!seen.insert(a).second || !seen.insert(b).second);
			if (dup) {
				set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "combine: a slot may take part in one pair per call");
				return false;
			}
		}
		if (preserve_sources)
			for (int64_t i = 0; i < n_pairs; ++i)
				if (source_slots[i] != target_slots[i] && seen.count(source_slots[i])) {
					set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "combine: a slot is source and target in one call");
					return false;
				}
	}
	std::lock_guard<std::mutex> lk(s->ctx->mu);
	if (hip_fail(hipSetDevice(s->ctx->device), "hipSetDevice", out_error)) return false;
	const size_t b = align_up((size_t)n_pairs * sizeof(uint32_t), 256);
	hipStream_t st = s->ctx->stream;
	if (2 * b > s->pair_bytes) {
		if (hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error)) return false;
		if (!ensure_buffer(&s->pair_buf, &s->pair_bytes, 2 * b, "combine pairs", out_error)) return false;
	}
	uint32_t *d_src = (uint32_t *)s->pair_buf, *d_dst = (uint32_t *)((char *)s->pair_buf + b);
	if (hip_fail(hipMemcpyAsync(d_src, source_slots, (size_t)n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, st), "H2D", out_error)) return false;
	if (hip_fail(hipMemcpyAsync(d_dst, target_slots, (size_t)n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, st), "H2D", out_error)) return false;
	if (!s->log_only && hip_fail(launch_ingest_combine(s->moments, s->n_accum, s->n_slots, d_src, d_dst, n_pairs, (int)s->p,
	                                                   s->opt.fit_intercept ? 1 : 0, preserve_sources ? 1 : 0, st),
	                             "combine kernel launch", out_error))
		return false;
	if (preserve_sources) {
		// (r4) the sources live on AND count for their targets: the targets get copies of the sources' logged rows
		if (!log_duplicate(s, n_pairs, source_slots, target_slots, out_error)) return false;
	} else if ((s->retain || s->log_only) && !s->log_dropped && s->log_rows > 0) { // the sources' rows in the log now belong to the targets
		if (!ensure_buffer(&s->remap_buf, &s->remap_bytes, (size_t)s->n_slots * sizeof(uint32_t), "row log remap", out_error)) return false;
		if (hip_fail(launch_rowlog_remap((uint32_t *)s->remap_buf, s->n_slots, d_src, d_dst, n_pairs, s->slabs.data(), (int)s->slabs.size(), st),
		             "row log remap launch", out_error))
			return false;
	}
	// the pair arrays are pageable host memory: the copies above have completed on return, the kernel is stream-ordered
	return !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error);
}

} // extern "C"

namespace {

// solve n records (the state's own: every slot [0, n); or a gathered subset); d_core / d_inf are device buffers
bool run_solve(AnofoxHipAggState *s, int64_t n, double *d_moments, const int64_t *d_rule_counts, double *d_core, double *d_inf, AnofoxError *e);
bool run_finalize(AnofoxHipAggState *s, int64_t n, double *d_core, double *d_inf, AnofoxError *e) {
	return run_solve(s, n, s->moments, s->n_accum, d_core, d_inf, e);
}
bool run_solve(AnofoxHipAggState *s, int64_t n, double *d_moments, const int64_t *d_rule_counts, double *d_core, double *d_inf, AnofoxError *e) {
	AnofoxHipContext *ctx = s->ctx;
	const size_t p = s->p;
	// workspace: refine list | refine vec | counters | t memo  (the refinement passes need the rows and do not run here;
	// the queue only counts the groups that would have taken them)
	const size_t b_lst = align_up((size_t)n * sizeof(int32_t), 256);
	const size_t b_vec = align_up((size_t)n * refine_vec_len((int)p) * sizeof(double), 256);
	if (!ensure_buffer(&ctx->ws, &ctx->ws_bytes, b_lst + b_vec + 256 + kTcritTableBytes, "workspace", e)) return false;
	char *base = (char *)ctx->ws;
	BatchArgs a;
	memset(&a, 0, sizeof a);
	a.n_groups = n;
	a.p = (int)p;
	a.model = (int)s->opt.model;
	a.fit_intercept = s->opt.fit_intercept ? 1 : 0;
	a.compute_inference = s->opt.compute_inference ? 1 : 0;
	a.lambda_scaling = (int)s->opt.lambda_scaling;
	a.hc_type = ANOFOX_HC_NONE;
	a.confidence_level = s->opt.confidence_level;
	a.alpha = s->opt.alpha;
	a.moments = d_moments;
	a.core = d_core;
	a.inference = s->opt.compute_inference ? d_inf : nullptr;
	a.refine_list = (int32_t *)base;
	a.refine_vec = (double *)(base + b_lst);
	a.refine_count = (int32_t *)(base + b_lst + b_vec);
	a.tcrit_table = base + b_lst + b_vec + 256;
	a.rule_counts = d_rule_counts; // the aggregate's "< 2 accumulated rows -> NULL" (ols_aggregate.cpp:263-267)
	ctx->last_refine_count = a.refine_count;
	hipStream_t st = ctx->stream;
	if (hip_fail(hipMemsetAsync(a.refine_count, 0, 256 + kTcritTableBytes, st), "hipMemsetAsync", e)) return false;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (ctx->timing) {
		e0 = get_event(ctx);
		e1 = get_event(ctx);
		(void)hipEventRecord(e0, st);
	}
	if (hip_fail(launch_solve_narrow(a, st), "solve kernel launch", e)) return false;
	if (ctx->timing) {
		(void)hipEventRecord(e1, st);
		ctx->predict_events.emplace_back(e0, e1); // reported as the state's finalize stage (predict_ms of the timing struct)
	}
	return true;
}

// The batch path (accumulate -> solve -> refine [-> HC]) on the logged rows of K slots — d_list[0 .. K) (device), or
// every slot 0 .. n - 1 when d_list is nullptr (K == n) — rowlog.hip's header has the steps.  Records go to rows
// d_list[k] of d_core / d_inf (scatter) or, for all slots, straight to rows 0 .. n - 1.  Synchronises the stream once
// (the number of selected rows has to reach the host).
bool refit_from_log(AnofoxHipAggState *s, int64_t n, int64_t K, const int32_t *d_list, bool keep_hc, double *d_core, double *d_inf,
                    AnofoxError *e, const int32_t *d_pos = nullptr) {
	AnofoxHipContext *ctx = s->ctx;
	hipStream_t st = ctx->stream;
	const size_t p = s->p;
	const bool weighted = s->opt.model == ANOFOX_HIP_MODEL_WLS;
	const bool all = d_list == nullptr;
	unsigned row_bits = 0, end_bit = 0;
	if (!rowlog_key_bits(s->log_rows, K, &row_bits, &end_bit)) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT, "finalize: " + std::to_string(K) + " groups x " + std::to_string(s->log_rows) +
		                                             " logged rows do not fit one 64-bit refit key");
		return false;
	}
	// index scratch: sorted slots | dense map | counters | slab table | sort temp for K keys
	const size_t n_tab = s->slabs.size() ? s->slabs.size() : 1;
	const size_t b_sorted = align_up((size_t)K * sizeof(int32_t), 256), b_dense = align_up((size_t)n * sizeof(int32_t), 256);
	const size_t b_tab = align_up(n_tab * sizeof(RowLogSlab), 256), b_t1 = align_up(rowlog_sort_temp_bytes(K), 256);
	if (!ensure_buffer(&s->refit_idx, &s->refit_idx_bytes, b_sorted + b_dense + 256 + b_tab + b_t1, "refit scratch", e)) return false;
	char *ib = (char *)s->refit_idx;
	int32_t *d_sorted = (int32_t *)ib, *d_dense = (int32_t *)(ib + b_sorted);
	unsigned long long *d_counter = (unsigned long long *)(ib + b_sorted + b_dense); // [0] selected rows, [1] rows of slots >= n
	RowLogSlab *d_tab = (RowLogSlab *)(ib + b_sorted + b_dense + 256);
	void *d_t1 = ib + b_sorted + b_dense + 256 + b_tab;
	bool bad = all ? hip_fail(launch_rowlog_iota(d_sorted, K, st), "refit iota", e)
	               : hip_fail(launch_rowlog_sort_slots(d_list, d_sorted, K, d_t1, b_t1, st), "refit sort", e);
	bad = bad || hip_fail(launch_rowlog_dense(d_sorted, K, d_dense, n, st), "refit mark", e);
	bad = bad || hip_fail(hipMemsetAsync(d_counter, 0, 256, st), "hipMemsetAsync", e);
	if (!s->slabs.empty())
		bad = bad || hip_fail(hipMemcpyAsync(d_tab, s->slabs.data(), s->slabs.size() * sizeof(RowLogSlab), hipMemcpyHostToDevice, st), "H2D", e);
	for (size_t k = 0; !bad && k < s->slabs.size(); ++k) {
		const RowLogSlab &sl = s->slabs[k];
		bad = hip_fail(launch_rowlog_select(false, sl.slot, sl.valid, sl.rows, sl.first_row, d_dense, n, d_counter, nullptr, row_bits, st), "refit count", e);
	}
	unsigned long long counts[2] = {0, 0};
	bad = bad || hip_fail(hipMemcpyAsync(counts, d_counter, sizeof counts, hipMemcpyDeviceToHost, st), "D2H", e);
	bad = bad || hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", e); // (also: the slab table copy has left the vector)
	if (bad) return false;
	if (all && counts[1] != 0) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT, "update: a row named a slot index >= n_slots (the row was dropped)");
		return false;
	}
	if (!all && counts[0] == 0) return true; // nothing logged for them (cannot happen for a fitted group): leave them as they are
	const size_t M = (size_t)counts[0];
	const size_t Mb = M ? M : 1;
	// row scratch: keys a | keys b | y | w | x columns | offsets | core | inference | sort temp for M keys
	const size_t b_k = align_up(Mb * sizeof(uint64_t), 256), b_c = align_up(Mb * sizeof(double), 256);
	const size_t b_off = align_up((size_t)(K + 1) * sizeof(int64_t), 256);
	const size_t b_core = all ? 0 : align_up((size_t)K * (p + 6) * sizeof(double), 256);
	const size_t b_inf = (!all && s->opt.compute_inference) ? align_up((size_t)K * (5 * p + 2) * sizeof(double), 256) : 0;
	const size_t b_t2 = align_up(rowlog_sort_temp_bytes((int64_t)Mb), 256);
	if (!ensure_buffer(&s->refit_rows, &s->refit_rows_bytes, 2 * b_k + (2 + p) * b_c + b_off + b_core + b_inf + b_t2, "refit scratch", e)) return false;
	char *rb = (char *)s->refit_rows;
	uint64_t *d_ka = (uint64_t *)rb, *d_kb = (uint64_t *)(rb + b_k);
	double *d_y = (double *)(rb + 2 * b_k), *d_w = (double *)(rb + 2 * b_k + b_c), *d_x = (double *)(rb + 2 * b_k + 2 * b_c);
	int64_t *d_off = (int64_t *)(rb + 2 * b_k + (2 + p) * b_c);
	double *d_core2 = all ? d_core : (double *)((char *)d_off + b_off);
	double *d_inf2 = all ? d_inf : (b_inf ? (double *)((char *)d_off + b_off + b_core) : nullptr);
	void *d_t2 = (char *)d_off + b_off + b_core + b_inf;
	bad = hip_fail(hipMemsetAsync(d_counter, 0, 256, st), "hipMemsetAsync", e);
	for (size_t k = 0; !bad && k < s->slabs.size(); ++k) {
		const RowLogSlab &sl = s->slabs[k];
		bad = hip_fail(launch_rowlog_select(true, sl.slot, sl.valid, sl.rows, sl.first_row, d_dense, n, d_counter, d_ka, row_bits, st), "refit fill", e);
	}
	if (M) bad = bad || hip_fail(launch_rowlog_sort_keys(d_ka, d_kb, (int64_t)M, end_bit, d_t2, b_t2, st), "refit sort", e);
	// (columns are b_c bytes apart, not M doubles: every column starts 256-byte aligned)
	const size_t col_stride = b_c / sizeof(double);
	bad = bad || hip_fail(launch_rowlog_gather(d_kb, (int64_t)M, K, d_tab, (int)s->slabs.size(), (int)p, weighted ? 1 : 0, d_y, d_x, col_stride, d_w,
	                                           d_off, row_bits, st),
	                      "refit gather", e);
	if (bad) return false;
	const double *x_cols[kWideMaxP];
	for (size_t j = 0; j < p; ++j) x_cols[j] = d_x + j * col_stride;
	AnofoxHipBatchOptions opt = s->opt;
	if (!keep_hc) opt.hc_type = ANOFOX_HC_NONE;
	if (!refit_groups_device(ctx, K, p, (int64_t)M, d_off, d_y, x_cols, weighted ? d_w : nullptr, opt, d_core2, d_inf2, e)) return false;
	if (all) return true;
	bad = hip_fail(launch_rowlog_scatter(d_core2, d_sorted, K, (int)(p + 6), d_core, d_pos, st), "refit scatter", e);
	if (d_inf2) bad = bad || hip_fail(launch_rowlog_scatter(d_inf2, d_sorted, K, (int)(5 * p + 2), d_inf, d_pos, st), "refit scatter", e);
	return !bad;
}

// After run_finalize: how many groups its solve queued for refinement, and — with a row log — their refit through
// the batch path.  *remaining = the groups still unrefined afterwards; when it is not 0 their slot numbers are still
// the first *remaining words of the context's workspace.  Synchronises the stream (the counts have to reach the host).
bool refit_queued(AnofoxHipAggState *s, int64_t n, double *d_core, double *d_inf, int64_t *remaining, AnofoxError *e) {
	AnofoxHipContext *ctx = s->ctx;
	hipStream_t st = ctx->stream;
	int32_t queued = 0;
	if (hip_fail(hipMemcpyAsync(&queued, ctx->last_refine_count, sizeof queued, hipMemcpyDeviceToHost, st), "D2H", e)) return false;
	if (hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", e)) return false;
	if (queued > n) queued = (int32_t)n;
	*remaining = queued;
	if (queued == 0 || !s->retain || s->log_dropped || s->log_rows == 0) return true;
	// (the queue is the first words of the context's workspace, which the refit's own batch call reuses: refit_from_log
	// sorts it into the state's scratch before that)
	if (!refit_from_log(s, n, queued, (const int32_t *)ctx->ws, false, d_core, d_inf, e)) return false;
	*remaining = 0;
	return true;
}

// The groups run_finalize queued and nothing refitted are not handed out as numbers: NaN records with status
// ANOFOX_HIP_STATUS_UNREFINED.  The queue and its length are read on the device (no synchronisation).
bool flag_unrefined(AnofoxHipAggState *s, int64_t n, double *d_core, double *d_inf, AnofoxError *e) {
	AnofoxHipContext *ctx = s->ctx;
	return !hip_fail(launch_rowlog_flag_unrefined((const int32_t *)ctx->ws, ctx->last_refine_count, n, (int)s->p, d_core,
	                                              s->opt.compute_inference ? d_inf : nullptr, ctx->stream),
	                 "flag kernel launch", e);
}

bool check_finalize(AnofoxHipAggState *s, int64_t n, const void *core, const void *inf, AnofoxError *e) {
	if (!s) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "state is NULL"); return false; }
	if (n < 0 || n > s->n_slots) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "n_slots exceeds the slots in use"); return false; }
	if (n > 0 && !core) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "core is NULL"); return false; }
	if (n > 0 && s->opt.compute_inference && !inf) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "inference buffer is NULL"); return false; }
	return true;
}

// the sticky flag of ingest_keys_kernel: a row named a slot >= n_slots (its row was dropped)
bool check_slot_flag(AnofoxHipAggState *s, AnofoxError *e) {
	int32_t c[2] = {0, 0};
	if (hip_fail(hipMemcpyAsync(c, s->counters, sizeof c, hipMemcpyDeviceToHost, s->ctx->stream), "hipMemcpy", e)) return false;
	if (hip_fail(hipStreamSynchronize(s->ctx->stream), "hipStreamSynchronize", e)) return false;
	if (c[1] != 0) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT, "update: a row named a slot index >= n_slots (the row was dropped)");
		return false;
	}
	return true;
}

} // namespace

extern "C" {

bool anofox_hip_agg_state_finalize_device(AnofoxHipAggState *s, int64_t n_slots, double *d_core, double *d_inference, AnofoxError *out_error) {
	reset_error(out_error);
	if (!check_finalize(s, n_slots, d_core, d_inference, out_error)) return false;
	if (n_slots == 0) return true;
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	std::lock_guard<std::mutex> lk(s->ctx->mu);
	if (hip_fail(hipSetDevice(s->ctx->device), "hipSetDevice", out_error)) return false;
	if (s->log_only) return refit_from_log(s, n_slots, n_slots, nullptr, true, d_core, d_inference, out_error);
	if (!run_finalize(s, n_slots, d_core, d_inference, out_error)) return false;
	if (!s->retain || s->log_dropped) // no log: nothing to refit and no reason to synchronise — the queued groups are flagged
		return flag_unrefined(s, n_slots, d_core, d_inference, out_error);
	int64_t remaining = 0;
	if (!refit_queued(s, n_slots, d_core, d_inference, &remaining, out_error)) return false;
	if (remaining != 0 && !flag_unrefined(s, n_slots, d_core, d_inference, out_error)) return false;
	// (this path has synchronised already: report rows that named a slot >= n_slots, as finalize_host does; without a
	// log the device entry point stays asynchronous and leaves that check to the caller's own validation)
	return check_slot_flag(s, out_error);
}

bool anofox_hip_agg_state_finalize_host(AnofoxHipAggState *s, int64_t n_slots, double *core, double *inference, int64_t *out_unrefined,
                                        int32_t *out_unrefined_slots, AnofoxError *out_error) {
	reset_error(out_error);
	if (out_unrefined) *out_unrefined = 0;
	if (!check_finalize(s, n_slots, core, inference, out_error)) return false;
	if (n_slots == 0) return true;
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	std::lock_guard<std::mutex> lk(s->ctx->mu);
	AnofoxHipContext *ctx = s->ctx;
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	const size_t p = s->p, G = (size_t)n_slots;
	const size_t b_core = align_up(G * (p + 6) * sizeof(double), 256);
	const size_t b_inf = s->opt.compute_inference ? align_up(G * (5 * p + 2) * sizeof(double), 256) : 0;
	if (!ensure_buffer(&ctx->stage, &ctx->stage_bytes, b_core + b_inf, "staging", out_error)) return false;
	double *d_core = (double *)ctx->stage;
	double *d_inf = b_inf ? (double *)((char *)ctx->stage + b_core) : nullptr;
	int64_t queued = 0;
	if (s->log_only) {
		if (!refit_from_log(s, n_slots, n_slots, nullptr, true, d_core, d_inf, out_error)) return false;
	} else {
		if (!run_finalize(s, n_slots, d_core, d_inf, out_error)) return false;
		if (!refit_queued(s, n_slots, d_core, d_inf, &queued, out_error)) return false;
		if (queued > 0 && !flag_unrefined(s, n_slots, d_core, d_inf, out_error)) return false;
	}
	hipStream_t st = ctx->stream;
	if (hip_fail(hipMemcpyAsync(core, d_core, G * (p + 6) * sizeof(double), hipMemcpyDeviceToHost, st), "D2H core", out_error)) return false;
	if (d_inf && hip_fail(hipMemcpyAsync(inference, d_inf, G * (5 * p + 2) * sizeof(double), hipMemcpyDeviceToHost, st), "D2H inference", out_error)) return false;
	if (!check_slot_flag(s, out_error)) return false; // (synchronises the stream)
	if (out_unrefined) *out_unrefined = queued;
	if (out_unrefined_slots && queued > 0) { // the queue itself: the first words of the workspace (run_finalize)
		if (hip_fail(hipMemcpy(out_unrefined_slots, ctx->ws, (size_t)queued * sizeof(int32_t), hipMemcpyDeviceToHost), "D2H", out_error)) return false;
	}
	return true;
}

// Finalize of a SUBSET of the slots (DuckDB finalizes vectors of states; under the windowed-aggregate protocol states
// are created, finalized and destroyed per frame, so fitting every slot at every Finalize would cost frames x slots):
// record k of core / inference belongs to slots[k].  The listed slots must be distinct.
bool anofox_hip_agg_state_finalize_slots_host(AnofoxHipAggState *s, int64_t n_list, const uint32_t *slots, double *core, double *inference,
                                              int64_t *out_unrefined, AnofoxError *out_error) {
	reset_error(out_error);
	if (out_unrefined) *out_unrefined = 0;
	if (!s || n_list < 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "state is NULL or n_list negative"); return false; }
	if (n_list == 0) return true;
	if (!slots || !core) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "slots or core is NULL"); return false; }
	if (s->opt.compute_inference && !inference) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "inference buffer is NULL"); return false; }
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	for (int64_t k = 0; k < n_list; ++k)
		if ((int64_t)slots[k] >= s->n_slots) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "finalize: slot index out of range"); return false; }
	{
		// record k belongs to slots[k]: a slot listed twice would leave one of its two rows unwritten (stale staging memory
		// handed out as a fit).  The arena passes a sorted unique list; direct callers get an error instead.
		std::vector<uint32_t> sorted(slots, slots + n_list);
		std::sort(sorted.begin(), sorted.end());
		if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
			set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "finalize: a slot is listed twice");
			return false;
		}
	}
	std::lock_guard<std::mutex> lk(s->ctx->mu);
	AnofoxHipContext *ctx = s->ctx;
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	if (!s->log_only && !state_reserve(s, s->n_slots, out_error)) return false; // slots handed out but never updated
	hipStream_t st = ctx->stream;
	const size_t p = s->p, K = (size_t)n_list, rec = (size_t)moment_record_len((int)p);
	// staging: core | inference | list | positions | gathered records | gathered counts | mapped queue
	const size_t b_core = align_up(K * (p + 6) * sizeof(double), 256);
	const size_t b_inf = s->opt.compute_inference ? align_up(K * (5 * p + 2) * sizeof(double), 256) : 0;
	const size_t b_list = align_up(K * sizeof(uint32_t), 256), b_pos = align_up((size_t)s->n_slots * sizeof(int32_t), 256);
	const size_t b_mom = s->log_only ? 0 : align_up(K * rec * sizeof(double), 256), b_cnt = s->log_only ? 0 : align_up(K * sizeof(int64_t), 256);
	const size_t b_q = align_up(K * sizeof(int32_t), 256);
	if (!ensure_buffer(&ctx->stage, &ctx->stage_bytes, b_core + b_inf + b_list + b_pos + b_mom + b_cnt + b_q, "staging", out_error)) return false;
	char *sb = (char *)ctx->stage;
	double *d_core = (double *)sb;
	double *d_inf = b_inf ? (double *)(sb + b_core) : nullptr;
	uint32_t *d_sel = (uint32_t *)(sb + b_core + b_inf);
	int32_t *d_pos = (int32_t *)(sb + b_core + b_inf + b_list);
	double *d_mom = (double *)(sb + b_core + b_inf + b_list + b_pos);
	int64_t *d_cnt = (int64_t *)((char *)d_mom + b_mom);
	int32_t *d_q = (int32_t *)((char *)d_cnt + b_cnt);
	bool bad = hip_fail(hipMemcpyAsync(d_sel, slots, K * sizeof(uint32_t), hipMemcpyHostToDevice, st), "H2D", out_error);
	bad = bad || hip_fail(launch_rowlog_positions(d_sel, n_list, d_pos, s->n_slots, st), "positions kernel launch", out_error);
	if (bad) return false;
	int64_t queued = 0;
	if (s->log_only) {
		if (!refit_from_log(s, s->n_slots, n_list, (const int32_t *)d_sel, true, d_core, d_inf, out_error, d_pos)) return false;
	} else {
		if (hip_fail(launch_ingest_gather_slots(s->moments, s->n_accum, d_sel, n_list, (int)p, d_mom, d_cnt, st), "gather kernel launch", out_error))
			return false;
		if (!run_solve(s, n_list, d_mom, d_cnt, d_core, d_inf, out_error)) return false;
		int32_t q32 = 0;
		if (hip_fail(hipMemcpyAsync(&q32, ctx->last_refine_count, sizeof q32, hipMemcpyDeviceToHost, st), "D2H", out_error)) return false;
		if (hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error)) return false;
		queued = q32 > n_list ? n_list : q32;
		if (queued > 0) {
			if (s->retain && !s->log_dropped && s->log_rows > 0) {
				// the queue holds positions in the list: the refit wants slot numbers, its scatter goes back to positions
				if (hip_fail(launch_rowlog_map_queue((const int32_t *)ctx->ws, ctx->last_refine_count, d_sel, d_q, st), "queue kernel launch", out_error))
					return false;
				if (!refit_from_log(s, s->n_slots, queued, d_q, false, d_core, d_inf, out_error, d_pos)) return false;
				queued = 0;
			} else if (hip_fail(launch_rowlog_flag_unrefined((const int32_t *)ctx->ws, ctx->last_refine_count, n_list, (int)p, d_core, d_inf, st),
			                    "flag kernel launch", out_error)) {
				return false;
			}
		}
	}
	if (hip_fail(hipMemcpyAsync(core, d_core, K * (p + 6) * sizeof(double), hipMemcpyDeviceToHost, st), "D2H core", out_error)) return false;
	if (d_inf && hip_fail(hipMemcpyAsync(inference, d_inf, K * (5 * p + 2) * sizeof(double), hipMemcpyDeviceToHost, st), "D2H inference", out_error)) return false;
	if (!check_slot_flag(s, out_error)) return false; // (synchronises the stream)
	if (out_unrefined) *out_unrefined = queued;
	return true;
}

// Destroy of aggregate states (ols_aggregate.cpp:108-118): the listed slots are emptied — all-zero record, no accepted
// rows, their logged rows invalidated — and may be handed out again by the caller.
bool anofox_hip_agg_state_release_slots(AnofoxHipAggState *s, int64_t n_list, const uint32_t *slots, AnofoxError *out_error) {
	reset_error(out_error);
	if (!s || n_list < 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "state is NULL or n_list negative"); return false; }
	if (n_list == 0) return true;
	if (!slots) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "slots is NULL"); return false; }
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	for (int64_t k = 0; k < n_list; ++k)
		if ((int64_t)slots[k] >= s->n_slots) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "release: slot index out of range"); return false; }
	std::lock_guard<std::mutex> lk(s->ctx->mu);
	if (hip_fail(hipSetDevice(s->ctx->device), "hipSetDevice", out_error)) return false;
	hipStream_t st = s->ctx->stream;
	const size_t b = align_up((size_t)n_list * sizeof(uint32_t), 256);
	if (b > s->pair_bytes) {
		if (hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error)) return false;
		if (!ensure_buffer(&s->pair_buf, &s->pair_bytes, b, "release list", out_error)) return false;
	}
	uint32_t *d_list = (uint32_t *)s->pair_buf;
	if (hip_fail(hipMemcpyAsync(d_list, slots, (size_t)n_list * sizeof(uint32_t), hipMemcpyHostToDevice, st), "H2D", out_error)) return false;
	if (!s->log_only && s->moments) {
		// (slots beyond the capacity were never written: nothing to clear)
		for (int64_t k = 0; k < n_list; ++k)
			if ((int64_t)slots[k] >= s->capacity) { if (!state_reserve(s, s->n_slots, out_error)) return false; break; }
		if (hip_fail(launch_ingest_clear_slots(s->moments, s->n_accum, d_list, n_list, (int)s->p, st), "clear kernel launch", out_error)) return false;
	}
	if ((s->retain || s->log_only) && !s->log_dropped && s->log_rows > 0) {
		if (!ensure_buffer(&s->remap_buf, &s->remap_bytes, (size_t)s->n_slots * sizeof(uint32_t), "row log marks", out_error)) return false;
		if (hip_fail(launch_rowlog_invalidate((uint8_t *)s->remap_buf, s->n_slots, d_list, n_list, s->slabs.data(), (int)s->slabs.size(), st),
		             "row log invalidate launch", out_error))
			return false;
	}
	return !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error); // the list is pageable host memory
}

// Every slot empty, the row log released, the slot count back to zero — the state as it was created, its device buffers kept.
// (The DuckDB arena calls it between two executions of a prepared statement: the arena lives in the bind data.)
bool anofox_hip_agg_state_reset(AnofoxHipAggState *s, AnofoxError *out_error) {
	reset_error(out_error);
	if (!s) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "state is NULL"); return false; }
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	std::lock_guard<std::mutex> lk(s->ctx->mu);
	if (hip_fail(hipSetDevice(s->ctx->device), "hipSetDevice", out_error)) return false;
	hipStream_t st = s->ctx->stream;
	if (s->copy_stream && hip_fail(hipStreamSynchronize(s->copy_stream), "hipStreamSynchronize", out_error)) return false;
	if (s->moments && s->capacity > 0) {
		const size_t rec = (size_t)moment_record_len((int)s->p) * sizeof(double);
		if (hip_fail(hipMemsetAsync(s->moments, 0, (size_t)s->capacity * rec, st), "hipMemsetAsync", out_error)) return false;
		if (hip_fail(hipMemsetAsync(s->n_accum, 0, (size_t)s->capacity * sizeof(int64_t), st), "hipMemsetAsync", out_error)) return false;
	}
	log_free(s); // (synchronises the stream)
	s->log_dropped = false;
	s->n_slots = 0;
	s->rows = 0;
	return !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error);
}

size_t anofox_hip_agg_state_record_len(const AnofoxHipAggState *s) { return (s && !s->log_only) ? (size_t)moment_record_len((int)s->p) : 0; }

// Cross-device Combine of moment states (the DuckDB glue shards a query's aggregate states over the node's GPUs):
// the records of the listed slots as the device keeps them, out of one state and into another.  Rows kept in a row log do not
// travel: both calls give this state's log up — a group its moments cannot resolve is then FLAGGED by Finalize (status 101),
// never handed out as a number.
static bool slots_transfer(AnofoxHipAggState *s, int64_t n_list, const uint32_t *slots, double *records, int64_t *counts, bool import,
                           AnofoxError *out_error) {
	reset_error(out_error);
	if (!s || n_list < 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "state is NULL or n_list negative"); return false; }
	if (n_list == 0) return true;
	if (!slots || !records || !counts) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "slots, records or counts is NULL"); return false; }
	std::lock_guard<std::mutex> lk0(s->mu);
	if (!attached(s, out_error)) return false;
	if (s->log_only) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT,
		          "designs of more than 8 features and HC standard errors keep rows, not moment records: their states stay on one device");
		return false;
	}
	for (int64_t k = 0; k < n_list; ++k)
		if ((int64_t)slots[k] >= s->n_slots) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "slot index out of range"); return false; }
	if (import) { // (a slot written twice by one call would depend on the kernel's scheduling)
		std::vector<uint32_t> sorted(slots, slots + n_list);
		std::sort(sorted.begin(), sorted.end());
		if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "import: a slot is listed twice"); return false; }
	}
	std::lock_guard<std::mutex> lk(s->ctx->mu);
	AnofoxHipContext *ctx = s->ctx;
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	if (!state_reserve(s, s->n_slots, out_error)) return false;
	hipStream_t st = ctx->stream;
	const size_t K = (size_t)n_list, rec = (size_t)moment_record_len((int)s->p);
	const size_t b_list = align_up(K * sizeof(uint32_t), 256), b_mom = align_up(K * rec * sizeof(double), 256), b_cnt = align_up(K * sizeof(int64_t), 256);
	if (!ensure_buffer(&ctx->stage, &ctx->stage_bytes, b_list + b_mom + b_cnt, "staging", out_error)) return false;
	char *sb = (char *)ctx->stage;
	uint32_t *d_sel = (uint32_t *)sb;
	double *d_mom = (double *)(sb + b_list);
	int64_t *d_cnt = (int64_t *)(sb + b_list + b_mom);
	if (hip_fail(hipMemcpyAsync(d_sel, slots, K * sizeof(uint32_t), hipMemcpyHostToDevice, st), "H2D", out_error)) return false;
	if (import) {
		if (hip_fail(hipMemcpyAsync(d_mom, records, K * rec * sizeof(double), hipMemcpyHostToDevice, st), "H2D", out_error)) return false;
		if (hip_fail(hipMemcpyAsync(d_cnt, counts, K * sizeof(int64_t), hipMemcpyHostToDevice, st), "H2D", out_error)) return false;
		if (hip_fail(launch_ingest_scatter_slots(s->moments, s->n_accum, d_sel, n_list, (int)s->p, d_mom, d_cnt, st), "scatter kernel launch", out_error)) return false;
	} else {
		if (hip_fail(launch_ingest_gather_slots(s->moments, s->n_accum, d_sel, n_list, (int)s->p, d_mom, d_cnt, st), "gather kernel launch", out_error)) return false;
		if (hip_fail(hipMemcpyAsync(records, d_mom, K * rec * sizeof(double), hipMemcpyDeviceToHost, st), "D2H", out_error)) return false;
		if (hip_fail(hipMemcpyAsync(counts, d_cnt, K * sizeof(int64_t), hipMemcpyDeviceToHost, st), "D2H", out_error)) return false;
	}
	if (s->retain && !s->log_dropped) { // the rows behind these records are on the other side, or about to be
		log_free(s);
		s->log_dropped = true;
	}
	return !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error);
}

bool anofox_hip_agg_state_export_slots_host(AnofoxHipAggState *s, int64_t n_list, const uint32_t *slots, double *records, int64_t *counts,
                                            AnofoxError *out_error) {
	return slots_transfer(s, n_list, slots, records, counts, false, out_error);
}

bool anofox_hip_agg_state_import_slots_host(AnofoxHipAggState *s, int64_t n_list, const uint32_t *slots, const double *records,
                                            const int64_t *counts, AnofoxError *out_error) {
	return slots_transfer(s, n_list, slots, const_cast<double *>(records), const_cast<int64_t *>(counts), true, out_error);
}

void *anofox_hip_host_alloc(size_t bytes) {
	void *p = nullptr;
	if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		return nullptr;
	}
	return p;
}

void anofox_hip_host_free(void *p) {
	if (p) (void)hipHostFree(p);
}

} // extern "C"
