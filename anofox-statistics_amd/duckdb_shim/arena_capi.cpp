// arena_capi.cpp — C entry points around anofox_shim::AggArena so that the shim's Update / Combine / Finalize logic
// (agg_arena.hpp) is compiled and exercised in this repository although DuckDB's headers are not available:
// tests/test_gpu_arena.py drives it the way duckdb_shim/fit_agg_hip.cpp does (vectors of <= 2048 rows from several
// threads, lazily initialised states, Combine of thread-local states, Finalize vector by vector).
// Test infrastructure for the shim: builds into libanofox_arena_capi.so next to this file.
#include "agg_arena.hpp"

using anofox_shim::AggArena;

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

} // extern "C"
