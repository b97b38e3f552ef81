// agg_arena.hpp — the DuckDB-independent half of the aggregate shim: what Initialize / Update /
// Combine / Finalize / Destroy of {ols,ridge,wls}_fit_agg do once the per-group row buffers of the reference
// (src/aggregate_functions/ols_aggregate.cpp:19-42) are replaced by ONE GPU-resident state per query.
//
//   NewSlot() / ReleaseSlot()   Initialize / Destroy of one DuckDB aggregate state (ols_aggregate.cpp:103-118): a slot
//               number of the query's GPU state.  Released slots are emptied on the device and handed out again, so
//               the windowed-aggregate protocol (states created, finalized and destroyed frame by frame) does not grow
//               the state without bound.
//   Writer      one Update call (a vector of <= 2048 rows, ols_aggregate.cpp:120-186).  Appends go WITHOUT a lock into
//               the calling thread's own page-locked chunk buffer ({slot, y, x[p] row-major as the LIST child delivers
//               it, w}); only a full chunk is handed — under the shipping lock — to anofox_hip_agg_state_update_host,
//               after which the rows live on as O(p^2) moments (or as logged rows) on the GPU and the buffer is reused.
//               (Round 2 held ONE mutex for the whole Update and copied to the GPU inside it: every worker thread of
//               the query serialised on it.)  The arena's first accepted row fixes its feature count; a different LIST
//               length throws (the glue keeps one arena per feature count and applies the reference's per-state rule,
//               ols_aggregate.cpp:165-175, before rows get here).
//   Combine()   pairs of (source slot, target slot) -> anofox_hip_agg_state_combine_ex (ols_aggregate.cpp:189-234);
//               a target that occurs several times in a call is served in rounds, in order; preserve = the sources live
//               on (DuckDB's AggregateCombineType::PRESERVE_INPUT: window segment trees).
//   Fetch()     Finalize of one vector of states (ols_aggregate.cpp:249-338 loops one FFI call per group): fits the
//               slots that changed since they were last fitted — ALL of a GROUP BY's groups in one batched call at the
//               first Finalize, only a window frame's new states later — and copies the asked records out.
//
// Wider designs (9 .. 128 features) and HC standard errors go through the same calls: the library then keeps the rows
// themselves in HBM instead of moments (a "log-only" state, include/anofox_stats_hip.h) and fits them at Finalize.
//
// Plain C++17 over the C ABI of include/anofox_stats_hip.h; no DuckDB types, so it is compiled and tested in this
// repository (duckdb_shim/arena_capi.cpp + tests/test_gpu_arena.py, tests/tools/arena_sanitize.cpp under ASan / UBSan).
// fit_agg_hip.cpp is the DuckDB glue on top.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "anofox_stats_hip.h"

namespace anofox_shim {

class AggArena {
public:
	// chunk_rows: rows per page-locked chunk buffer, one buffer per thread that calls Update (256 Ki rows = 19 MB at
	//   p = 8; capped at 32 MiB of x per buffer for wide designs)
	// retain_bytes / retain_host_bytes: HBM, then page-locked host memory, that the device state may spend on keeping
	//   the rows (anofox_hip_agg_state_retain_rows / _retain_rows_host) so that Finalize refits ill-conditioned and
	//   nearly exactly fitting groups as the batch path does; the reference keeps every row on the host anyway
	//   (ols_aggregate.cpp:19-42).  Defaults 64 GiB of the 288 and 32 GiB; ANOFOX_HIP_RETAIN_BYTES /
	//   ANOFOX_HIP_RETAIN_HOST_BYTES override them.  Groups of a query that outgrows both come back as SQL NULL with
	//   status ANOFOX_HIP_STATUS_UNREFINED when (and only when) the moments cannot resolve them; Unrefined() counts them.
	// device: the GPU of this arena's state (-1 = the calling thread's current device); ShardedAggArena gives every shard its own
	explicit AggArena(AnofoxHipBatchOptions options, size_t chunk_rows = (size_t)1 << 18, size_t retain_bytes = (size_t)64 << 30,
	                  size_t retain_host_bytes = (size_t)32 << 30, int device = -1)
	    : opt_(options), chunk_rows_(chunk_rows ? chunk_rows : 1), retain_bytes_(retain_bytes), retain_host_bytes_(retain_host_bytes), device_(device) {
		if (const char *v = getenv("ANOFOX_HIP_RETAIN_BYTES")) retain_bytes_ = (size_t)strtoull(v, nullptr, 10);
		if (const char *v = getenv("ANOFOX_HIP_RETAIN_HOST_BYTES")) retain_host_bytes_ = (size_t)strtoull(v, nullptr, 10);
	}
	AggArena(const AggArena &) = delete;
	AggArena &operator=(const AggArena &) = delete;
	~AggArena() {
		if (state_) anofox_hip_agg_state_destroy(state_);
		if (ctx_) anofox_hip_context_destroy(ctx_);
		for (auto &c : chunks_) FreeChunk(*c);
	}

	uint32_t NewSlot() {
		// (r4) a new execution of a prepared statement?  The arena lives in the bind data, which DuckDB re-uses: once an
		// execution's states have all been finalized AND destroyed, the next Initialize starts from a clean device state
		// (slot numbers from 0, the row log's budget whole again) instead of resting on what the last execution left.
		if (fetched_.load(std::memory_order_acquire)) ResetIfIdle();
		std::lock_guard<std::mutex> lk(mu_);
		uint32_t s;
		if (!free_.empty()) {
			s = free_.back();
			free_.pop_back();
		} else {
			s = n_slots_.fetch_add(1, std::memory_order_relaxed);
		}
		++live_slots_;
		return s;
	}
	// Destroy of a state: the slot is emptied on the device before it is handed out again (at the next Combine / Fetch)
	void ReleaseSlot(uint32_t slot) {
		std::lock_guard<std::mutex> lk(mu_);
		pending_release_.push_back(slot);
		if (live_slots_) --live_slots_;
	}
	uint32_t SlotCount() const { return n_slots_.load(std::memory_order_relaxed); } // high-water mark
	uint32_t LiveSlots() const {
		std::lock_guard<std::mutex> lk(mu_);
		return live_slots_;
	}
	size_t FeatureCount() const { return p_.load(std::memory_order_acquire); } // 0 until the first accepted row
	uint64_t RowsAccepted() const { return rows_.load(std::memory_order_relaxed); }
	int64_t Unrefined() const { return unrefined_.load(std::memory_order_relaxed); } // groups flagged so far (status 101)
	uint64_t FitCalls() const { return fit_calls_.load(std::memory_order_relaxed); }
	uint64_t SlotsFitted() const { return slots_fitted_.load(std::memory_order_relaxed); }
	bool RetainingRows() const { return state_ && anofox_hip_agg_state_retaining(state_) != 0; }
	const AnofoxHipBatchOptions &Options() const { return opt_; }

private:
	// page-locked rows of ONE thread that calls Update, waiting for their trip to the GPU
	struct Chunk {
		uint32_t *slot = nullptr;
		double *y = nullptr, *x = nullptr, *w = nullptr;
		size_t fill = 0, cap = 0;
		std::thread::id owner;
		std::atomic<bool> busy {false}; // an Update call (or a flush) is using it: whoever flips it false -> true owns the chunk
		// slots the Update calls of this chunk's thread touched since the marks were last merged into the arena's dirty list:
		// appended by the owner without a lock, merged when the chunk ships or is flushed (r4: one lock per 256 Ki rows
		// instead of two per 2048-row vector — with 32 and more writer threads those were the arena's bottleneck)
		std::vector<uint32_t> touched;
	};
	static constexpr size_t kMaxChunks = 4096; // writer threads of one query

public:
	// One Update call.  Not shared between threads; appends take no lock.
	class Writer {
	public:
		explicit Writer(AggArena &a) : a_(a) {}
		Writer(const Writer &) = delete;
		Writer &operator=(const Writer &) = delete;
		~Writer() { a_.EndWriter(chunk_, n_rows_); }
		// Initialize of a state that this Update call is the first to touch
		uint32_t NewSlot() { return a_.NewSlot(); }
		// Starts a row and returns where its n_features values go (NULL list entries as NaN: the fit's row filter drops
		// such rows, ols.rs:59-66); w is ignored unless the model is WLS.
		double *Begin(uint32_t slot, double y, size_t n_features, double w = 1.0) {
			size_t p = a_.p_.load(std::memory_order_acquire);
			if (p == 0) p = a_.Init(n_features);
			if (n_features != p)
				throw std::invalid_argument("Inconsistent feature count: expected " + std::to_string(p) + ", got " + std::to_string(n_features));
			if (!chunk_) chunk_ = a_.AcquireChunk();
			if (chunk_->fill == chunk_->cap) a_.Ship(*chunk_);
			const size_t i = chunk_->fill++;
			chunk_->slot[i] = slot;
			chunk_->y[i] = y;
			if (chunk_->w) chunk_->w[i] = w;
			if (chunk_->touched.empty() || chunk_->touched.back() != slot) chunk_->touched.push_back(slot);
			++n_rows_;
			return chunk_->x + i * p;
		}
		void Append(uint32_t slot, double y, const double *x, size_t n_features, double w = 1.0) {
			double *dst = Begin(slot, y, n_features, w);
			memcpy(dst, x, n_features * sizeof(double));
		}

	private:
		friend class AggArena;
		AggArena &a_;
		Chunk *chunk_ = nullptr;
		uint64_t n_rows_ = 0;
	};

	// preserve: the sources keep their rows (AggregateCombineType::PRESERVE_INPUT); otherwise they are emptied, as the
	// reference's Combine moves or appends them (ols_aggregate.cpp:189-234).
	void Combine(const uint32_t *source_slots, const uint32_t *target_slots, size_t n, bool preserve = false) {
		if (n == 0) return;
		std::lock_guard<std::mutex> ship(ship_mu_);
		if (!state_) return; // no accepted row anywhere: every slot is empty already
		FlushAllShipLocked();
		DrainReleasesShipLocked();
		Reserve();
		// Rounds: within one library call every pair is merged by its own wavefront, so a slot may be written by one
		// pair only, and the sequential meaning of the pair list (ols_aggregate.cpp:196-233 loops it in order) has to
		// survive: pair (a -> b) runs after every earlier pair that wrote a or b, and after every earlier pair that read b.
		std::unordered_map<uint32_t, uint32_t> after_write, after_read; // slot -> first round that may touch it again
		std::vector<std::vector<uint32_t>> src, dst;
		for (size_t i = 0; i < n; ++i) {
			const uint32_t a = source_slots[i], b = target_slots[i];
			if (a == b) continue;
			uint32_t r = std::max(after_write[a], after_write[b]);
			r = std::max(r, after_read[b]);
			if (!preserve) r = std::max(r, after_read[a]); // (a destructive combine writes its source as well)
			if (r >= src.size()) {
				src.resize(r + 1);
				dst.resize(r + 1);
			}
			src[r].push_back(a);
			dst[r].push_back(b);
			after_write[b] = r + 1;
			if (preserve) after_read[a] = std::max(after_read[a], r + 1);
			else after_write[a] = r + 1;
		}
		AnofoxError err;
		for (size_t r = 0; r < src.size(); ++r)
			if (!src[r].empty() && !anofox_hip_agg_state_combine_ex(state_, (int64_t)src[r].size(), src[r].data(), dst[r].data(), preserve, &err)) Throw(err);
		std::lock_guard<std::mutex> lk(mu_);
		for (size_t i = 0; i < n; ++i) {
			MarkDirtyLocked(target_slots[i]);
			if (!preserve) MarkDirtyLocked(source_slots[i]);
		}
	}

	// ---- (r4) Combine across devices: the source lives in ANOTHER arena (the glue shards a query's states over the node's GPUs) ----
	// ExportRecords: the listed slots' moment records and accepted-row counts, as the device keeps them (pending rows flushed
	// first).  MergeRecords: the same records into `targets` of THIS arena — through fresh slots, the library's ordinary combine
	// (the imported rows count as arriving after the target's, ols_aggregate.cpp:224-233) and the slots' release.  ClearSlots:
	// what a consuming Combine leaves of its sources (emptied on the device; the DuckDB states keep their slot numbers until
	// Destroy).  Moment states only (up to 8 features, no HC errors): the library refuses log-only states, whose rows stay put.
	size_t RecordLen() const { return state_ ? anofox_hip_agg_state_record_len(state_) : 0; }
	void ExportRecords(const uint32_t *slots, size_t n, std::vector<double> &records, std::vector<int64_t> &counts) {
		records.clear();
		counts.clear();
		if (n == 0) return;
		std::lock_guard<std::mutex> ship(ship_mu_);
		if (!state_) throw std::runtime_error("anofox_stats fit_agg (HIP): export from an arena without a device state");
		FlushAllShipLocked();
		DrainReleasesShipLocked();
		Reserve();
		const size_t rec = anofox_hip_agg_state_record_len(state_);
		records.resize(n * rec);
		counts.resize(n);
		AnofoxError err;
		if (!anofox_hip_agg_state_export_slots_host(state_, (int64_t)n, slots, records.data(), counts.data(), &err)) Throw(err);
	}
	void MergeRecords(const double *records, const int64_t *counts, const uint32_t *targets, size_t n) {
		if (n == 0) return;
		std::vector<uint32_t> tmp(n);
		for (size_t i = 0; i < n; ++i) tmp[i] = NewSlot();
		{
			std::lock_guard<std::mutex> ship(ship_mu_);
			if (!state_) throw std::runtime_error("anofox_stats fit_agg (HIP): import into an arena without a device state");
			FlushAllShipLocked();
			DrainReleasesShipLocked();
			Reserve();
			AnofoxError err;
			if (!anofox_hip_agg_state_import_slots_host(state_, (int64_t)n, tmp.data(), records, counts, &err)) Throw(err);
		}
		Combine(tmp.data(), targets, n, false);
		for (size_t i = 0; i < n; ++i) ReleaseSlot(tmp[i]);
	}
	void ClearSlots(const uint32_t *slots, size_t n) {
		if (n == 0) return;
		std::vector<uint32_t> rel(slots, slots + n);
		std::sort(rel.begin(), rel.end());
		rel.erase(std::unique(rel.begin(), rel.end()), rel.end());
		std::lock_guard<std::mutex> ship(ship_mu_);
		if (!state_) return;
		FlushAllShipLocked();
		Reserve();
		AnofoxError err;
		if (!anofox_hip_agg_state_release_slots(state_, (int64_t)rel.size(), rel.data(), &err)) Throw(err);
		std::lock_guard<std::mutex> lk(mu_);
		for (uint32_t s : rel) MarkDirtyLocked(s);
	}
	uint64_t Resets() const { return resets_.load(std::memory_order_relaxed); }

	// the device state exists from the arena's first accepted row on; a cross-device Combine may reach an arena before that
	void EnsureState(size_t n_features) {
		if (p_.load(std::memory_order_acquire) == 0) (void)Init(n_features);
	}

	// Finalize of n states: out_core [n x (p + 6)], out_inf [n x (5 p + 2)] (nullptr unless inference was asked for),
	// out_status[n] = the record's status word (0 = a fit; 100 = fewer than 2 accumulated rows; an AnofoxErrorCode; 101 =
	// unrefined).  A query without any accepted row: every status 100.
	void Fetch(const uint32_t *slots, size_t n, double *out_core, double *out_inf, int *out_status) {
		std::lock_guard<std::mutex> ship(ship_mu_);
		SolveShipLocked();
		fetched_.store(true, std::memory_order_release);
		const size_t p = p_.load(std::memory_order_acquire);
		const size_t lc = p + 6, li = 5 * p + 2;
		for (size_t i = 0; i < n; ++i) {
			const size_t s = slots[i];
			if (!state_ || (s + 1) * lc > core_.size()) {
				out_status[i] = ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS;
				continue;
			}
			const double *rec = &core_[s * lc];
			out_status[i] = (int)rec[p + 5];
			if (out_core) memcpy(out_core + i * lc, rec, lc * sizeof(double));
			if (out_inf && !inf_.empty()) memcpy(out_inf + i * li, &inf_[s * li], li * sizeof(double));
		}
	}

private:
	friend class Writer;

	[[noreturn]] static void Throw(const AnofoxError &e) { throw std::runtime_error(std::string("anofox_stats fit_agg (HIP): ") + e.message); }

	// First accepted row of this arena: fixes ITS feature count and creates the device state.  (The reference fixes the
	// count per state, ols_aggregate.cpp:164-175; the glue keeps one arena per count that occurs and checks every row
	// against its own state's count, so this arena only ever sees one width.)  Context and state are built into locals and
	// committed only when all of it succeeded.
	size_t Init(size_t n_features) {
		std::lock_guard<std::mutex> ship(ship_mu_);
		size_t p = p_.load(std::memory_order_acquire);
		if (p) return p;
		if (n_features == 0 || n_features > anofox_hip_max_features())
			throw std::invalid_argument("anofox_stats fit_agg (HIP): 1.." + std::to_string(anofox_hip_max_features()) +
			                            " features are supported, got " + std::to_string(n_features));
		AnofoxError err;
		AnofoxHipContext *ctx = nullptr;
		AnofoxHipAggState *state = nullptr;
		if (!anofox_hip_context_create(device_, &ctx, &err)) Throw(err);
		// up to 8 features without HC errors: O(p^2) moments per slot (+ the row log for the groups they cannot resolve);
		// wider designs and HC errors: the library keeps the rows themselves (log-only state)
		bool ok = anofox_hip_agg_state_create(ctx, n_features, opt_, 0, &state, &err);
		const bool log_only = n_features > 8 || (opt_.compute_inference && opt_.hc_type != ANOFOX_HC_NONE && opt_.model != ANOFOX_HIP_MODEL_RIDGE);
		// the HBM budget is for the OPTIONAL log of a moment state; a log-only state needs every row it is given (HBM until
		// the device is full, then the host budget; an Update beyond both fails like any allocation)
		if (ok && !log_only && retain_bytes_) ok = anofox_hip_agg_state_retain_rows(state, retain_bytes_, &err);
		if (ok && retain_host_bytes_ && (log_only || retain_bytes_)) ok = anofox_hip_agg_state_retain_rows_host(state, retain_host_bytes_, &err);
		if (!ok) {
			if (state) anofox_hip_agg_state_destroy(state);
			anofox_hip_context_destroy(ctx);
			Throw(err);
		}
		ctx_ = ctx;
		state_ = state;
		p_.store(n_features, std::memory_order_release);
		return n_features;
	}

	// the calling thread's chunk buffer (created on its first Update).  No lock on the way: the chunk table is an array of
	// pointers that only grows, and the thread remembers its chunk of the arena it wrote to last.
	Chunk *AcquireChunk() {
		static thread_local uint64_t tl_arena_id = 0;
		static thread_local Chunk *tl_chunk = nullptr;
		Chunk *mine = nullptr;
		if (tl_arena_id == id_ && tl_chunk) {
			mine = tl_chunk;
		} else {
			const std::thread::id me = std::this_thread::get_id();
			const size_t n = n_chunks_.load(std::memory_order_acquire);
			for (size_t k = 0; k < n; ++k) {
				Chunk *c = chunk_tab_[k].load(std::memory_order_acquire);
				if (c && c->owner == me) {
					mine = c;
					break;
				}
			}
		}
		if (mine) {
			bool expect = false;
			while (!mine->busy.compare_exchange_weak(expect, true, std::memory_order_acquire)) { // (a flush from another thread may hold it for a moment)
				expect = false;
				std::this_thread::yield();
			}
			tl_arena_id = id_;
			tl_chunk = mine;
			return mine;
		}
		const size_t p = p_.load(std::memory_order_acquire);
		auto c = std::make_unique<Chunk>();
		size_t cap = chunk_rows_;
		const size_t max_rows = ((size_t)32 << 20) / (p * sizeof(double));
		if (cap > max_rows) cap = max_rows < 2048 ? 2048 : max_rows;
		c->cap = cap;
		c->owner = std::this_thread::get_id();
		c->busy.store(true, std::memory_order_relaxed);
		const bool weighted = opt_.model == ANOFOX_HIP_MODEL_WLS;
		c->slot = (uint32_t *)anofox_hip_host_alloc(cap * sizeof(uint32_t));
		c->y = (double *)anofox_hip_host_alloc(cap * sizeof(double));
		c->x = (double *)anofox_hip_host_alloc(cap * p * sizeof(double));
		c->w = weighted ? (double *)anofox_hip_host_alloc(cap * sizeof(double)) : nullptr;
		if (!c->slot || !c->y || !c->x || (weighted && !c->w)) {
			FreeChunk(*c);
			throw std::bad_alloc();
		}
		std::lock_guard<std::mutex> lk(mu_);
		const size_t k = n_chunks_.load(std::memory_order_relaxed);
		if (k >= kMaxChunks) {
			FreeChunk(*c);
			throw std::runtime_error("anofox_stats fit_agg (HIP): more than " + std::to_string(kMaxChunks) + " writer threads in one query");
		}
		chunks_.push_back(std::move(c));
		Chunk *raw = chunks_.back().get();
		chunk_tab_[k].store(raw, std::memory_order_release);
		n_chunks_.store(k + 1, std::memory_order_release);
		tl_arena_id = id_;
		tl_chunk = raw;
		return raw;
	}
	static void FreeChunk(Chunk &c) {
		anofox_hip_host_free(c.slot);
		anofox_hip_host_free(c.y);
		anofox_hip_host_free(c.x);
		anofox_hip_host_free(c.w);
		c.slot = nullptr;
		c.y = c.x = c.w = nullptr;
	}
	void EndWriter(Chunk *c, uint64_t n_rows) noexcept {
		rows_.fetch_add(n_rows, std::memory_order_relaxed);
		if (c) c->busy.store(false, std::memory_order_release); // (its dirty marks travel with the chunk: MergeTouched)
	}
	// the slots a chunk's Update calls touched -> the arena's dirty list; the caller owns the chunk (busy)
	void MergeTouched(Chunk &c) {
		if (c.touched.empty()) return;
		std::lock_guard<std::mutex> lk(mu_);
		for (uint32_t s : c.touched) MarkDirtyLocked(s);
		c.touched.clear();
	}
	void MarkDirtyLocked(uint32_t slot) {
		if (slot >= dirty_.size()) dirty_.resize((size_t)slot + 1 + dirty_.size() / 2, 0);
		if (dirty_[slot] == 0) dirty_list_.push_back(slot);
		dirty_[slot] = 1; // (2 = released while dirty: already listed)
	}

	// a full chunk of the calling Writer goes to the GPU; returns once the rows have been copied (the kernels run on
	// while the buffer refills)
	void Ship(Chunk &c) {
		std::lock_guard<std::mutex> ship(ship_mu_);
		ShipLocked(c);
	}
	void ShipLocked(Chunk &c) {
		MergeTouched(c);
		if (c.fill == 0) return;
		AnofoxError err;
		if (!anofox_hip_agg_state_update_host(state_, (int64_t)c.fill, (int64_t)n_slots_.load(std::memory_order_relaxed), c.slot, c.y, c.x, c.w, nullptr, &err))
			Throw(err);
		c.fill = 0;
	}
	// Every thread's pending rows, except those of a thread that is inside an Update call right now: DuckDB never
	// finalizes or combines a state while another thread is still feeding it (hash aggregates sink completely before they
	// combine; a window task creates, feeds, finalizes and destroys its states on its own thread), so such a chunk holds
	// no row of the states this call is about.  (Waiting for it instead would deadlock: its thread may be waiting for the
	// shipping lock this thread holds.)
	void FlushAllShipLocked() {
		const size_t n = n_chunks_.load(std::memory_order_acquire);
		for (size_t k = 0; k < n; ++k) {
			Chunk *c = chunk_tab_[k].load(std::memory_order_acquire);
			bool expect = false;
			if (!c || !c->busy.compare_exchange_strong(expect, true, std::memory_order_acquire)) continue; // inside an Update call: see above
			try {
				ShipLocked(*c); // (also merges the chunk's dirty marks when it holds no row)
			} catch (...) {
				c->busy.store(false, std::memory_order_release);
				throw;
			}
			c->busy.store(false, std::memory_order_release);
		}
	}
	// the slots Destroy gave back: emptied on the device (their pending rows have been flushed before), then reusable
	void DrainReleasesShipLocked() {
		std::vector<uint32_t> rel;
		{
			std::lock_guard<std::mutex> lk(mu_);
			rel.swap(pending_release_);
		}
		if (rel.empty()) return;
		std::sort(rel.begin(), rel.end());
		rel.erase(std::unique(rel.begin(), rel.end()), rel.end());
		if (state_) {
			Reserve();
			AnofoxError err;
			if (!anofox_hip_agg_state_release_slots(state_, (int64_t)rel.size(), rel.data(), &err)) Throw(err);
		}
		const size_t p = p_.load(std::memory_order_acquire);
		std::lock_guard<std::mutex> lk(mu_);
		for (uint32_t s : rel) {
			if (s < dirty_.size() && dirty_[s]) dirty_[s] = 2; // (stays in dirty_list_, skipped there)
			if (p && ((size_t)s + 1) * (p + 6) <= core_.size()) core_[(size_t)s * (p + 6) + p + 5] = (double)ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS;
			free_.push_back(s);
		}
	}
	// see NewSlot: only when nothing of the last execution is alive — no state holds a slot, no thread is inside an Update
	void ResetIfIdle() {
		std::lock_guard<std::mutex> ship(ship_mu_);
		{
			std::lock_guard<std::mutex> lk(mu_);
			if (!fetched_.load(std::memory_order_acquire) || live_slots_ != 0) return;
			const size_t n = n_chunks_.load(std::memory_order_acquire);
			for (size_t k = 0; k < n; ++k) {
				Chunk *c = chunk_tab_[k].load(std::memory_order_acquire);
				if (c && (c->busy.load(std::memory_order_acquire) || c->fill != 0)) return; // rows of a running Update: not idle
			}
			fetched_.store(false, std::memory_order_release);
			free_.clear();
			pending_release_.clear();
			std::fill(dirty_.begin(), dirty_.end(), 0);
			dirty_list_.clear();
			core_.clear();
			inf_.clear();
			n_slots_.store(0, std::memory_order_relaxed);
			for (size_t k = 0; k < n; ++k)
				if (Chunk *c = chunk_tab_[k].load(std::memory_order_acquire)) c->touched.clear();
		}
		if (state_) {
			AnofoxError err;
			if (!anofox_hip_agg_state_reset(state_, &err)) Throw(err);
		}
		resets_.fetch_add(1, std::memory_order_relaxed);
	}
	void Reserve() {
		AnofoxError err;
		if (!anofox_hip_agg_state_reserve(state_, (int64_t)n_slots_.load(std::memory_order_relaxed), &err)) Throw(err);
	}

	// fit the slots that changed since they were last fitted
	void SolveShipLocked() {
		if (!state_) return;
		FlushAllShipLocked();
		DrainReleasesShipLocked();
		std::vector<uint32_t> todo;
		{
			std::lock_guard<std::mutex> lk(mu_);
			for (uint32_t s : dirty_list_) {
				if (dirty_[s] == 1) todo.push_back(s);
				dirty_[s] = 0;
			}
			dirty_list_.clear();
		}
		if (todo.empty()) return;
		Reserve();
		const size_t p = p_.load(std::memory_order_acquire), lc = p + 6, li = 5 * p + 2;
		const size_t n_all = n_slots_.load(std::memory_order_relaxed);
		const bool inference = opt_.compute_inference;
		if (core_.size() < n_all * lc) {
			const size_t old = core_.size() / lc;
			core_.resize(n_all * lc);
			for (size_t s = old; s < n_all; ++s) core_[s * lc + p + 5] = (double)ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS;
			if (inference) inf_.resize(n_all * li);
		}
		AnofoxError err;
		int64_t unref = 0;
		if (todo.size() == n_all) { // a GROUP BY's first Finalize: one batched call for every group
			if (!anofox_hip_agg_state_finalize_host(state_, (int64_t)n_all, core_.data(), inference ? inf_.data() : nullptr, &unref, nullptr, &err)) Throw(err);
		} else {
			std::sort(todo.begin(), todo.end());
			std::vector<double> c(todo.size() * lc), f(inference ? todo.size() * li : 0);
			if (!anofox_hip_agg_state_finalize_slots_host(state_, (int64_t)todo.size(), todo.data(), c.data(), inference ? f.data() : nullptr, &unref, &err))
				Throw(err);
			for (size_t k = 0; k < todo.size(); ++k) {
				memcpy(&core_[(size_t)todo[k] * lc], &c[k * lc], lc * sizeof(double));
				if (inference) memcpy(&inf_[(size_t)todo[k] * li], &f[k * li], li * sizeof(double));
			}
		}
		unrefined_.fetch_add(unref, std::memory_order_relaxed);
		fit_calls_.fetch_add(1, std::memory_order_relaxed);
		slots_fitted_.fetch_add(todo.size(), std::memory_order_relaxed);
	}

	AnofoxHipBatchOptions opt_;
	size_t chunk_rows_;
	size_t retain_bytes_, retain_host_bytes_;
	int device_;
	mutable std::mutex mu_; // slots, dirty marks, the chunk table
	std::mutex ship_mu_;    // every call into the library (taken before mu_, never while holding it)
	AnofoxHipContext *ctx_ = nullptr;
	AnofoxHipAggState *state_ = nullptr;
	std::atomic<size_t> p_ {0};
	std::atomic<uint32_t> n_slots_ {0};
	uint32_t live_slots_ = 0;
	std::vector<uint32_t> free_, pending_release_;
	std::vector<std::unique_ptr<Chunk>> chunks_;           // owns the chunks (under mu_)
	std::atomic<Chunk *> chunk_tab_[kMaxChunks] = {};        // the same pointers for lock-free readers; only grows
	std::atomic<size_t> n_chunks_ {0};
	static uint64_t NextId() {
		static std::atomic<uint64_t> next {1};
		return next.fetch_add(1, std::memory_order_relaxed);
	}
	const uint64_t id_ = NextId();                           // (a thread's cached chunk belongs to THIS arena, not to one that lived at its address)
	std::vector<uint8_t> dirty_; // 1 = changed since its last fit, 2 = released while dirty
	std::vector<uint32_t> dirty_list_;
	std::atomic<uint64_t> rows_ {0}, fit_calls_ {0}, slots_fitted_ {0};
	std::atomic<int64_t> unrefined_ {0};
	std::atomic<bool> fetched_ {false};  // a Finalize has handed records out since the last reset
	std::atomic<uint64_t> resets_ {0};
	std::vector<double> core_, inf_; // last fitted record of every slot
};

} // namespace anofox_shim
