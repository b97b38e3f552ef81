// agg_arena.hpp — the DuckDB-independent half of the aggregate shim: what Update / Combine /
// Finalize of {ols,ridge,wls}_fit_agg do with their rows once the per-group row buffers of the reference
// (src/aggregate_functions/ols_aggregate.cpp:19-42) are replaced by ONE GPU-resident state per query.
//
//   NewSlot()   Initialize of one DuckDB aggregate state: hands out the next slot number.
//   Writer      one Update call (a vector of <= 2048 rows): locks the arena once, appends the accepted rows
//               {slot, y, x[p], w} to page-locked chunk buffers ("columnar arenas": y and w arrays, row-major x as
//               the LIST child delivers it), and ships a full buffer with anofox_hip_agg_state_update_host — the
//               rows then live on as O(p^2) moments on the GPU and the buffer is reused.  The first accepted row
//               fixes the feature count; a different LIST length throws the reference's message
//               (ols_aggregate.cpp:165-175).
//   Combine()   pairs of (source slot, target slot) -> anofox_hip_agg_state_combine (ols_aggregate.cpp:189-234).
//   Solve()     flushes, then ONE finalize for every slot of the query; Core(slot) / Inference(slot) serve the
//               Finalize vectors from that result (ols_aggregate.cpp:249-338 loops one FFI call per group).
//
// Wider designs (9 .. 128 features) and HC standard errors go through the same calls: the library then keeps the rows
// themselves in HBM instead of moments (a "log-only" state, include/anofox_stats_hip.h) and fits them at Finalize.
// (Round 2's first version buffered such rows in host chunks here and made one batched call at Solve; removed.)
//
// Plain C++17 over the C ABI of include/anofox_stats_hip.h; no DuckDB types, so it is compiled and tested in this
// repository (duckdb_shim/arena_capi.cpp + tests/test_gpu_arena.py).  fit_agg_hip.cpp is the thin DuckDB glue on top.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <stdexcept>
#include <utility>
#include <string>
#include <vector>

#include "anofox_stats_hip.h"

namespace anofox_shim {

class AggArena {
public:
	// flush_rows: rows per page-locked chunk buffer (1M rows = 76 MB at p = 8: one ingest pass of the library)
	// retain_bytes: HBM the device state may spend on keeping the rows (anofox_hip_agg_state_retain_rows), so that
	//   Finalize refits ill-conditioned and nearly exactly fitting groups as the batch path does; the reference keeps
	//   every row on the host anyway (ols_aggregate.cpp:19-42).  0 = moments only.  Default 64 GiB of the 288;
	//   ANOFOX_HIP_RETAIN_BYTES overrides it.  A query that outgrows it continues without the log (Unrefined() > 0).
	explicit AggArena(AnofoxHipBatchOptions options, size_t flush_rows = (size_t)1 << 20, size_t retain_bytes = (size_t)64 << 30)
	    : opt_(options), cap_(flush_rows), retain_bytes_(retain_bytes) {
		if (cap_ == 0) cap_ = 1;
		if (const char *v = getenv("ANOFOX_HIP_RETAIN_BYTES")) retain_bytes_ = (size_t)strtoull(v, nullptr, 10);
	}
	AggArena(const AggArena &) = delete;
	AggArena &operator=(const AggArena &) = delete;
	~AggArena() {
		if (state_) anofox_hip_agg_state_destroy(state_);
		if (ctx_) anofox_hip_context_destroy(ctx_);
		FreeBuffers();
	}

	uint32_t NewSlot() {
		std::lock_guard<std::mutex> lk(mu_);
		solved_ = false;
		return n_slots_++;
	}
	uint32_t SlotCount() const { return n_slots_; }
	size_t FeatureCount() const { return p_; } // 0 until the first accepted row
	uint64_t RowsAccepted() const { return rows_; }
	int64_t Unrefined() const { return unrefined_; }
	bool RetainingRows() const { return state_ && anofox_hip_agg_state_retaining(state_) != 0; }

	// One Update call: holds the arena's lock for the lifetime of the object.
	class Writer {
	public:
		explicit Writer(AggArena &a) : a_(a), lk_(a.mu_) { a_.solved_ = false; }
		// Initialize of a state that this Update call is the first to touch (the lock is already held)
		uint32_t NewSlot() { return a_.n_slots_++; }
		// x: the row's LIST(DOUBLE) entries, NULL entries already replaced by NaN; w ignored unless the model is WLS
		void Append(uint32_t slot, double y, const double *x, size_t n_features, double w = 1.0) { a_.AppendLocked(slot, y, x, n_features, w); }

	private:
		AggArena &a_;
		std::lock_guard<std::mutex> lk_;
	};

	void Combine(const uint32_t *source_slots, const uint32_t *target_slots, size_t n) {
		std::lock_guard<std::mutex> lk(mu_);
		if (n == 0 || !state_) return; // no accepted row anywhere: every slot is empty already
		FlushLocked();
		Reserve();
		AnofoxError err;
		if (!anofox_hip_agg_state_combine(state_, (int64_t)n, source_slots, target_slots, &err)) Throw(err);
		solved_ = false;
	}

	// Flush the pending rows and fit every slot (once; later calls are free until the state changes again).
	void Solve() {
		std::lock_guard<std::mutex> lk(mu_);
		if (solved_) return;
		core_.clear();
		inf_.clear();
		if (state_) {
			FlushLocked();
			Reserve();
			core_.resize((size_t)n_slots_ * (p_ + 6));
			if (opt_.compute_inference) inf_.resize((size_t)n_slots_ * (5 * p_ + 2));
			AnofoxError err;
			if (!anofox_hip_agg_state_finalize_host(state_, n_slots_, core_.data(), inf_.empty() ? nullptr : inf_.data(), &unrefined_, nullptr, &err))
				Throw(err);
		}
		solved_ = true;
	}
	// Records of one slot after Solve(): nullptr = SQL NULL (no accepted row in the whole query, or a status != 0).
	const double *Core(uint32_t slot) const {
		if (core_.empty() || slot >= n_slots_) return nullptr;
		const double *rec = &core_[(size_t)slot * (p_ + 6)];
		return rec[p_ + 5] != 0.0 ? nullptr : rec;
	}
	const double *Inference(uint32_t slot) const { return inf_.empty() || slot >= n_slots_ ? nullptr : &inf_[(size_t)slot * (5 * p_ + 2)]; }
	int Status(uint32_t slot) const {
		if (core_.empty() || slot >= n_slots_) return ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS;
		return (int)core_[(size_t)slot * (p_ + 6) + p_ + 5];
	}

private:
	[[noreturn]] static void Throw(const AnofoxError &e) { throw std::runtime_error(std::string("anofox_stats fit_agg (HIP): ") + e.message); }

	void AppendLocked(uint32_t slot, double y, const double *x, size_t n_features, double w) {
		if (p_ == 0) { // first accepted row of the query fixes the feature count (per state in the reference; the
			// aggregate's x argument is one column, so every state sees the same LIST length or the query fails)
			if (n_features == 0 || n_features > anofox_hip_max_features())
				throw std::invalid_argument("anofox_stats fit_agg (HIP): 1.." + std::to_string(anofox_hip_max_features()) +
				                            " features are supported, got " + std::to_string(n_features));
			p_ = n_features;
			AnofoxError err;
			if (!anofox_hip_context_create(-1, &ctx_, &err)) Throw(err);
			// up to 8 features without HC errors: O(p^2) moments per slot (+ the row log for the groups they cannot
			// resolve); wider designs and HC errors: the library keeps the rows themselves in HBM (log-only state)
			if (!anofox_hip_agg_state_create(ctx_, p_, opt_, 0, &state_, &err)) Throw(err);
			// the budget is for the OPTIONAL log of a moment state; a log-only state needs every row it is given (its
			// log grows until the device is full, and an Update beyond that fails like any allocation)
			const bool log_only = p_ > 8 || (opt_.compute_inference && opt_.hc_type != ANOFOX_HC_NONE && opt_.model != ANOFOX_HIP_MODEL_RIDGE);
			if (!log_only && retain_bytes_ && !anofox_hip_agg_state_retain_rows(state_, retain_bytes_, &err)) Throw(err);
			AllocBuffers();
		}
		if (n_features != p_)
			throw std::invalid_argument("Inconsistent feature count: expected " + std::to_string(p_) + ", got " + std::to_string(n_features));
		slot_[fill_] = slot;
		y_[fill_] = y;
		memcpy(x_ + fill_ * p_, x, p_ * sizeof(double));
		if (w_) w_[fill_] = w;
		++rows_;
		if (++fill_ == cap_) FlushLocked();
	}

	void Reserve() {
		AnofoxError err;
		if (!anofox_hip_agg_state_reserve(state_, n_slots_, &err)) Throw(err);
	}
	void FlushLocked() {
		if (!state_ || fill_ == 0) return;
		AnofoxError err;
		// returns once the rows have been copied to the GPU; the kernels run on while the buffer refills
		if (!anofox_hip_agg_state_update_host(state_, (int64_t)fill_, n_slots_, slot_, y_, x_, w_, nullptr, &err)) Throw(err);
		fill_ = 0;
	}
	void AllocBuffers() {
		const bool weighted = opt_.model == ANOFOX_HIP_MODEL_WLS;
		slot_ = (uint32_t *)anofox_hip_host_alloc(cap_ * sizeof(uint32_t));
		y_ = (double *)anofox_hip_host_alloc(cap_ * sizeof(double));
		x_ = (double *)anofox_hip_host_alloc(cap_ * p_ * sizeof(double));
		w_ = weighted ? (double *)anofox_hip_host_alloc(cap_ * sizeof(double)) : nullptr;
		if (!slot_ || !y_ || !x_ || (weighted && !w_)) {
			FreeBuffers();
			throw std::bad_alloc();
		}
	}
	void FreeBuffers() {
		anofox_hip_host_free(slot_);
		anofox_hip_host_free(y_);
		anofox_hip_host_free(x_);
		anofox_hip_host_free(w_);
		slot_ = nullptr;
		y_ = x_ = w_ = nullptr;
	}

	AnofoxHipBatchOptions opt_;
	size_t cap_;
	std::mutex mu_;
	AnofoxHipContext *ctx_ = nullptr;
	AnofoxHipAggState *state_ = nullptr;
	size_t p_ = 0;
	uint32_t n_slots_ = 0;
	uint64_t rows_ = 0;
	// page-locked chunk buffers
	uint32_t *slot_ = nullptr;
	double *y_ = nullptr, *x_ = nullptr, *w_ = nullptr;
	size_t fill_ = 0;
	// solved records
	bool solved_ = false;
	int64_t unrefined_ = 0;
	size_t retain_bytes_ = 0;
	std::vector<double> core_, inf_;
};

} // namespace anofox_shim
