// mock_abi.hpp — a recording MOCK of the C ABI (include/anofox_stats_hip.h) for the CPU tier: it keeps the rows it is
// given (per slot, in arrival order) and "fits" a group by three order-sensitive sums, so that a test can tell whether
// every accepted row reached the right slot in the right order — through Update vectors from several threads, flushes,
// Combine, Destroy and Finalize.  Included by tests/tools/arena_sanitize.cpp and glue_sanitize.cpp (ASan / UBSan builds,
// no library, no GPU).  Test infrastructure only; nothing here is shipped.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <atomic>
#include <vector>

#include "../../include/anofox_stats_hip.h"

struct Row {
	double y, x0, w;
};
// the mock's "fit": core[0] = sum y, core[1] = sum (k + 1) y_k (arrival order), core[2] = sum x0 w; n at p + 4, status 0
void mock_fit(const std::vector<Row> &rows, size_t p, double *core) {
	for (size_t k = 0; k < p + 6; ++k) core[k] = 0.0;
	double a = 0, b = 0, c = 0;
	for (size_t k = 0; k < rows.size(); ++k) {
		a += rows[k].y;
		b += (double)(k + 1) * rows[k].y;
		c += rows[k].x0 * rows[k].w;
	}
	core[0] = a;
	if (p > 1) core[1] = b;
	if (p > 2) core[2] = c;
	core[p + 4] = (double)rows.size();
	core[p + 5] = rows.size() < 2 ? (double)ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS : 0.0;
}
// (atomic: the glue keeps one arena per feature count, and arenas of different widths initialise concurrently)
std::atomic<int> g_contexts {0}, g_states {0}, g_host_allocs {0}, g_batch_calls {0}, g_update_calls {0}, g_combine_calls {0}, g_release_calls {0},
    g_subset_calls {0}, g_full_calls {0}, g_fail_state_create {0};
int g_mock_unrefined_rows = -1; // >= 0: a group of exactly that many rows is flagged ANOFOX_HIP_STATUS_UNREFINED


struct AnofoxHipContext {
	int dummy;
};
struct AnofoxHipAggState {
	size_t p;
	bool weighted;
	std::vector<std::vector<Row>> slots;
};

extern "C" {
size_t anofox_hip_max_features(void) { return 128; }
size_t anofox_hip_agg_state_max_features(void) { return 128; }
bool anofox_hip_context_create(int, AnofoxHipContext **out, AnofoxError *) {
	*out = new AnofoxHipContext{0};
	++g_contexts;
	return true;
}
void anofox_hip_context_destroy(AnofoxHipContext *c) {
	if (c) --g_contexts;
	delete c;
}
void *anofox_hip_host_alloc(size_t bytes) {
	++g_host_allocs;
	return malloc(bytes);
}
void anofox_hip_host_free(void *p) {
	if (p) --g_host_allocs;
	free(p);
}
bool anofox_hip_agg_state_create(AnofoxHipContext *, size_t p, AnofoxHipBatchOptions opt, int64_t, AnofoxHipAggState **out, AnofoxError *err) {
	if (g_fail_state_create) { // (the failing-mock case of main: the arena must stay usable and leak nothing)
		err->code = ANOFOX_ERROR_ALLOCATION_FAILURE;
		snprintf(err->message, sizeof err->message, "mock: no device memory");
		return false;
	}
	*out = new AnofoxHipAggState{p, opt.model == ANOFOX_HIP_MODEL_WLS, {}};
	++g_states;
	return true;
}
void anofox_hip_agg_state_destroy(AnofoxHipAggState *s) {
	if (s) --g_states;
	delete s;
}
bool anofox_hip_agg_state_retain_rows(AnofoxHipAggState *, size_t, AnofoxError *) { return true; }
bool anofox_hip_agg_state_retain_rows_host(AnofoxHipAggState *, size_t, AnofoxError *) { return true; }
int anofox_hip_agg_state_retaining(const AnofoxHipAggState *) { return 1; }
bool anofox_hip_agg_state_reserve(AnofoxHipAggState *s, int64_t n, AnofoxError *) {
	if ((size_t)n > s->slots.size()) s->slots.resize((size_t)n);
	return true;
}
bool anofox_hip_agg_state_update_host(AnofoxHipAggState *s, int64_t n_rows, int64_t n_slots, const uint32_t *slot, const double *y,
                                      const double *x, const double *w, const uint8_t *valid, AnofoxError *err) {
	++g_update_calls;
	if ((size_t)n_slots > s->slots.size()) s->slots.resize((size_t)n_slots);
	for (int64_t i = 0; i < n_rows; ++i) {
		if (valid && !valid[i]) continue;
		if ((int64_t)slot[i] >= n_slots) {
			err->code = ANOFOX_ERROR_INVALID_INPUT;
			snprintf(err->message, sizeof err->message, "slot out of range");
			return false;
		}
		s->slots[slot[i]].push_back(Row{y[i], x[(size_t)i * s->p], s->weighted ? w[i] : 1.0});
	}
	return true;
}
bool anofox_hip_agg_state_combine_ex(AnofoxHipAggState *s, int64_t n, const uint32_t *src, const uint32_t *dst, bool preserve, AnofoxError *err) {
	++g_combine_calls;
	// the library merges the pairs of one call concurrently: a target may appear once, a destroyed source nowhere else
	std::map<uint32_t, int> writes, reads;
	for (int64_t i = 0; i < n; ++i) {
		if (src[i] == dst[i]) continue;
		++writes[dst[i]];
		if (preserve) ++reads[src[i]]; else ++writes[src[i]];
	}
	for (auto &kv : writes)
		if (kv.second > 1 || reads.count(kv.first)) {
			err->code = ANOFOX_ERROR_INVALID_INPUT;
			snprintf(err->message, sizeof err->message, "combine: a slot may take part in one pair per call");
			return false;
		}
	std::vector<std::vector<Row>> before;
	if (preserve) before = s->slots; // all pairs of a call read the state as it was
	for (int64_t i = 0; i < n; ++i) {
		if (src[i] == dst[i]) continue;
		auto &a = preserve ? before[src[i]] : s->slots[src[i]];
		auto &b = s->slots[dst[i]];
		b.insert(b.end(), a.begin(), a.end()); // the source's rows count as arriving after the target's
		if (!preserve) a.clear();
	}
	return true;
}
bool anofox_hip_agg_state_release_slots(AnofoxHipAggState *s, int64_t n, const uint32_t *slots, AnofoxError *) {
	++g_release_calls;
	for (int64_t i = 0; i < n; ++i)
		if (slots[i] < s->slots.size()) s->slots[slots[i]].clear();
	return true;
}
// cross-device Combine: the mock's "record" of a slot is a handle to a copy of its rows (one double), the count its row count
std::mutex g_export_mu;
std::vector<std::vector<Row>> g_exported;
std::atomic<int> g_export_calls {0}, g_import_calls {0};
std::atomic<int> g_reset_calls {0};
bool anofox_hip_agg_state_reset(AnofoxHipAggState *s, AnofoxError *) {
	++g_reset_calls;
	s->slots.clear();
	return true;
}
size_t anofox_hip_agg_state_record_len(const AnofoxHipAggState *s) { return s ? 1 : 0; }
bool anofox_hip_agg_state_export_slots_host(AnofoxHipAggState *s, int64_t n, const uint32_t *slots, double *records, int64_t *counts, AnofoxError *) {
	++g_export_calls;
	std::lock_guard<std::mutex> lk(g_export_mu);
	for (int64_t i = 0; i < n; ++i) {
		const std::vector<Row> rows = slots[i] < s->slots.size() ? s->slots[slots[i]] : std::vector<Row>();
		records[i] = (double)g_exported.size();
		counts[i] = (int64_t)rows.size();
		g_exported.push_back(rows);
	}
	return true;
}
bool anofox_hip_agg_state_import_slots_host(AnofoxHipAggState *s, int64_t n, const uint32_t *slots, const double *records, const int64_t *counts,
                                            AnofoxError *err) {
	++g_import_calls;
	std::lock_guard<std::mutex> lk(g_export_mu);
	for (int64_t i = 0; i < n; ++i) {
		const size_t h = (size_t)records[i];
		if (h >= g_exported.size() || (int64_t)g_exported[h].size() != counts[i] || slots[i] >= s->slots.size()) {
			err->code = ANOFOX_ERROR_INVALID_INPUT;
			snprintf(err->message, sizeof err->message, "mock: bad import");
			return false;
		}
		s->slots[slots[i]] = g_exported[h];
	}
	return true;
}
bool anofox_hip_agg_state_finalize_slots_host(AnofoxHipAggState *s, int64_t n, const uint32_t *slots, double *core, double *, int64_t *unrefined,
                                              AnofoxError *) {
	++g_subset_calls;
	int64_t flagged = 0;
	for (int64_t k = 0; k < n; ++k) {
		const std::vector<Row> rows = slots[k] < s->slots.size() ? s->slots[slots[k]] : std::vector<Row>();
		mock_fit(rows, s->p, core + (size_t)k * (s->p + 6));
		if ((int)rows.size() == g_mock_unrefined_rows) { core[(size_t)k * (s->p + 6) + s->p + 5] = (double)ANOFOX_HIP_STATUS_UNREFINED; ++flagged; }
	}
	if (unrefined) *unrefined = flagged;
	return true;
}
bool anofox_hip_agg_state_finalize_host(AnofoxHipAggState *s, int64_t n_slots, double *core, double *, int64_t *unrefined, int32_t *, AnofoxError *) {
	++g_full_calls;
	int64_t flagged = 0;
	for (int64_t g = 0; g < n_slots; ++g) {
		const std::vector<Row> rows = (size_t)g < s->slots.size() ? s->slots[(size_t)g] : std::vector<Row>();
		mock_fit(rows, s->p, core + (size_t)g * (s->p + 6));
		if ((int)rows.size() == g_mock_unrefined_rows) { core[(size_t)g * (s->p + 6) + s->p + 5] = (double)ANOFOX_HIP_STATUS_UNREFINED; ++flagged; }
	}
	if (unrefined) *unrefined = flagged;
	return true;
}
bool anofox_hip_fit_batch_host(AnofoxHipContext *, int64_t G, size_t p, int64_t n_rows, const int64_t *offs, const double *y,
                               const double *const *x_cols, const double *w, AnofoxHipBatchOptions, double *core, double *, AnofoxError *) {
	++g_batch_calls;
	if (offs[G] != n_rows) return false;
	for (int64_t g = 0; g < G; ++g) {
		std::vector<Row> rows;
		for (int64_t r = offs[g]; r < offs[g + 1]; ++r) rows.push_back(Row{y[r], x_cols[0][r], w ? w[r] : 1.0});
		mock_fit(rows, p, core + (size_t)g * (p + 6));
	}
	return true;
}
// the mock's fit + predict: S = sum over the group's training rows (non-NaN y, in order, k = position in the group) of (k + 1) y_k w_k;
// row r: { S + x0[r] + x_last[r], that - #training rows, that + train_counts[g] }; status 100 when fewer than 2 rows train
std::atomic<int> g_predict_calls {0}, g_vif_calls {0};
bool anofox_hip_fit_predict_batch_host(AnofoxHipContext *, int64_t G, size_t p, int64_t n_rows, const int64_t *offs, const double *y,
                                       const double *const *x_cols, const double *w, const int64_t *train_counts, AnofoxHipBatchOptions,
                                       double *core, double *pred, AnofoxError *) {
	++g_predict_calls;
	if (offs[G] != n_rows || p == 0) return false;
	for (int64_t g = 0; g < G; ++g) {
		double S = 0.0;
		int64_t n_train = 0;
		for (int64_t r = offs[g]; r < offs[g + 1]; ++r)
			if (!isnan(y[r])) {
				S += (double)(r - offs[g] + 1) * y[r] * (w ? w[r] : 1.0);
				++n_train;
			}
		double *c = core + (size_t)g * (p + 6);
		for (size_t k = 0; k < p + 6; ++k) c[k] = 0.0;
		c[p + 4] = (double)n_train;
		c[p + 5] = n_train < 2 ? (double)ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS : 0.0;
		for (int64_t r = offs[g]; r < offs[g + 1]; ++r) {
			const double v = S + x_cols[0][r] + x_cols[p - 1][r];
			pred[3 * r] = v;
			pred[3 * r + 1] = v - (double)n_train;
			pred[3 * r + 2] = v + (double)(train_counts ? train_counts[g] : offs[g + 1] - offs[g]);
		}
	}
	return true;
}
size_t anofox_hip_vif_record_len(size_t p) { return p + 1; }
size_t anofox_hip_vif_max_features(void) { return 129; }
// the mock's VIF of feature j: sum over the group's rows of (k + 1) x_j[k]
bool anofox_hip_vif_batch_host(AnofoxHipContext *, int64_t G, size_t p, int64_t n_rows, const int64_t *offs, const double *const *x_cols, double *vif,
                               AnofoxError *) {
	++g_vif_calls;
	if (offs[G] != n_rows) return false;
	for (int64_t g = 0; g < G; ++g) {
		for (size_t j = 0; j < p; ++j) {
			double s = 0.0;
			for (int64_t r = offs[g]; r < offs[g + 1]; ++r) s += (double)(r - offs[g] + 1) * x_cols[j][r];
			vif[(size_t)g * (p + 1) + j] = s;
		}
		vif[(size_t)g * (p + 1) + p] = offs[g + 1] - offs[g] < 3 ? 100.0 : 0.0;
	}
	return true;
}
}

