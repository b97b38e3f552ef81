// arena_capi.cpp — C entry points around anofox_shim::AggArena so that the shim's Update / Combine / Finalize logic
// (agg_arena.hpp) is compiled and exercised in this repository although DuckDB's headers are not available:
// tests/test_gpu_arena.py drives it the way duckdb_shim/fit_agg_hip.cpp does (vectors of <= 2048 rows from several
// threads, lazily initialised states, Combine of thread-local states, Finalize vector by vector).
// Test infrastructure for the shim: builds into libanofox_arena_capi.so next to this file.
#include "agg_arena.hpp"
#include "sharded_arena.hpp"

using anofox_shim::AggArena;
using anofox_shim::ShardedAggArena;

extern "C" {

#define ARENA_API __attribute__((visibility("default")))

ARENA_API void *arena_create(AnofoxHipBatchOptions options, size_t flush_rows) {
	try {
		return new AggArena(options, flush_rows);
	} catch (...) {
		return nullptr;
	}
}
ARENA_API void arena_destroy(void *a) { delete static_cast<AggArena *>(a); }
ARENA_API uint32_t arena_new_slot(void *a) { return static_cast<AggArena *>(a)->NewSlot(); }

// One Update call, as HipAggUpdate makes it: row i belongs to DuckDB state state_of_row[i]; state_slots[k] is that
// state's slot, -1 while no Update has touched it (assigned here, also when every row of the state is skipped);
// rows with accept[i] == 0 are the ones Update skips (NULL y / NULL x list / NULL weight).  x is row-major with
// `n_features` entries per row.  Returns 0, or -1 with the exception text in msg (<= 255 chars).
ARENA_API int arena_update(void *a, size_t n, const uint32_t *state_of_row, int64_t *state_slots, const double *y, const double *x,
                           size_t n_features, const double *w, const uint8_t *accept, char *msg) {
	try {
		AggArena::Writer wr(*static_cast<AggArena *>(a));
		for (size_t i = 0; i < n; ++i) {
			int64_t &slot = state_slots[state_of_row[i]];
			if (slot < 0) slot = wr.NewSlot();
			if (!accept || accept[i]) wr.Append((uint32_t)slot, y[i], x + i * n_features, n_features, w ? w[i] : 1.0);
		}
		return 0;
	} catch (const std::exception &e) {
		if (msg) {
			strncpy(msg, e.what(), 255);
			msg[255] = 0;
		}
		return -1;
	}
}
ARENA_API int arena_combine(void *a, const uint32_t *src, const uint32_t *dst, size_t n, char *msg) {
	try {
		static_cast<AggArena *>(a)->Combine(src, dst, n);
		return 0;
	} catch (const std::exception &e) {
		if (msg) {
			strncpy(msg, e.what(), 255);
			msg[255] = 0;
		}
		return -1;
	}
}
// Combine with the sources left intact (AggregateCombineType::PRESERVE_INPUT)
ARENA_API int arena_combine_preserve(void *a, const uint32_t *src, const uint32_t *dst, size_t n, char *msg) {
	try {
		static_cast<AggArena *>(a)->Combine(src, dst, n, true);
		return 0;
	} catch (const std::exception &e) {
		if (msg) {
			strncpy(msg, e.what(), 255);
			msg[255] = 0;
		}
		return -1;
	}
}
// Destroy of n states
ARENA_API void arena_release(void *a, const uint32_t *slot, size_t n) {
	for (size_t i = 0; i < n; ++i) static_cast<AggArena *>(a)->ReleaseSlot(slot[i]);
}
// Finalize of one vector of states: out_core [n x (p+6)], out_inf [n x (5p+2)] or NULL, is_null [n] (status != 0).
ARENA_API int arena_finalize(void *a, size_t n, const uint32_t *slot, double *out_core, double *out_inf, uint8_t *is_null, char *msg) {
	try {
		AggArena &ar = *static_cast<AggArena *>(a);
		std::vector<int> status(n);
		ar.Fetch(slot, n, out_core, out_inf, status.data());
		for (size_t i = 0; i < n; ++i) is_null[i] = status[i] != 0 ? 1 : 0;
		return 0;
	} catch (const std::exception &e) {
		if (msg) {
			strncpy(msg, e.what(), 255);
			msg[255] = 0;
		}
		return -1;
	}
}
ARENA_API size_t arena_feature_count(void *a) { return static_cast<AggArena *>(a)->FeatureCount(); }
ARENA_API uint64_t arena_rows(void *a) { return static_cast<AggArena *>(a)->RowsAccepted(); }
// groups flagged as unrefined by the Finalize calls so far (0 while the device state keeps the rows), and whether it does
ARENA_API int64_t arena_unrefined(void *a) { return static_cast<AggArena *>(a)->Unrefined(); }
ARENA_API int arena_retaining(void *a) { return static_cast<AggArena *>(a)->RetainingRows() ? 1 : 0; }
// bookkeeping the tests look at: slot numbers handed out (high-water mark), live states, library fit calls, slots fitted
ARENA_API uint32_t arena_slot_count(void *a) { return static_cast<AggArena *>(a)->SlotCount(); }
ARENA_API uint32_t arena_live_slots(void *a) { return static_cast<AggArena *>(a)->LiveSlots(); }
ARENA_API uint64_t arena_fit_calls(void *a) { return static_cast<AggArena *>(a)->FitCalls(); }
ARENA_API uint64_t arena_slots_fitted(void *a) { return static_cast<AggArena *>(a)->SlotsFitted(); }

// ---- hash-partitioned ingest over W device states (sharded_arena.hpp) ----
ARENA_API uint32_t sharded_shard_of(uint64_t key, uint32_t n_shards) { return anofox_shim::shard_of_key(key, n_shards); }
ARENA_API void *sharded_create(AnofoxHipBatchOptions options, uint32_t n_shards, const int *devices, size_t chunk_rows) {
	try {
		return new ShardedAggArena(options, n_shards, devices, chunk_rows);
	} catch (...) {
		return nullptr;
	}
}
ARENA_API void sharded_destroy(void *a) { delete static_cast<ShardedAggArena *>(a); }
// one Update call: row i has group key keys[i]; rows with accept[i] == 0 register their key but are skipped
ARENA_API int sharded_update(void *a, size_t n, const uint64_t *keys, const double *y, const double *x, size_t n_features, const double *w,
                             const uint8_t *accept, char *msg) {
	try {
		ShardedAggArena::Writer wr(*static_cast<ShardedAggArena *>(a));
		for (size_t i = 0; i < n; ++i) {
			if (accept && !accept[i]) { wr.Touch(keys[i]); continue; }
			wr.Append(keys[i], y[i], x + i * n_features, n_features, w ? w[i] : 1.0);
		}
		return 0;
	} catch (const std::exception &e) {
		if (msg) {
			strncpy(msg, e.what(), 255);
			msg[255] = 0;
		}
		return -1;
	}
}
ARENA_API uint64_t sharded_rows_of_shard(void *a, uint32_t s) { return static_cast<ShardedAggArena *>(a)->RowsOfShard(s); }
ARENA_API size_t sharded_keys_of_shard(void *a, uint32_t s) { return static_cast<ShardedAggArena *>(a)->KeysOfShard(s); }
ARENA_API size_t sharded_key_count(void *a) {
	auto &ar = *static_cast<ShardedAggArena *>(a);
	size_t n = 0;
	for (uint32_t s = 0; s < ar.ShardCount(); ++s) n += ar.KeysOfShard(s);
	return n;
}
// Finalize: out_keys [K], out_core [K x (p + 6)], out_inf [K x (5 p + 2)] or NULL, out_status [K]; K = sharded_key_count
ARENA_API int sharded_finalize(void *a, size_t capacity, uint64_t *out_keys, double *out_core, double *out_inf, int32_t *out_status, char *msg) {
	try {
		auto &ar = *static_cast<ShardedAggArena *>(a);
		std::vector<uint64_t> keys;
		std::vector<double> core, inf;
		std::vector<int> status;
		ar.Fetch(keys, core, out_inf ? &inf : nullptr, status);
		if (keys.size() > capacity) throw std::runtime_error("sharded_finalize: output capacity too small");
		const size_t p = ar.FeatureCount();
		memcpy(out_keys, keys.data(), keys.size() * sizeof(uint64_t));
		memcpy(out_core, core.data(), keys.size() * (p + 6) * sizeof(double));
		if (out_inf && !inf.empty()) memcpy(out_inf, inf.data(), keys.size() * (5 * p + 2) * sizeof(double));
		for (size_t k = 0; k < keys.size(); ++k) out_status[k] = status[k];
		return 0;
	} catch (const std::exception &e) {
		if (msg) {
			strncpy(msg, e.what(), 255);
			msg[255] = 0;
		}
		return -1;
	}
}

} // extern "C"
