// arena_sanitize.cpp — the DuckDB shim's arena (anofox-statistics_amd/duckdb_shim/agg_arena.hpp) on the CPU, under
// ASan / UBSan, against a MOCK of the C ABI: the mock keeps the rows it is given (streaming state: per slot, in
// arrival order) and "fits" a group by three order-sensitive sums, so that the
// test can tell whether the arena handed every accepted row to the right slot in the right order — through Update
// vectors from several threads, flushes, Combine and Finalize.  Test infrastructure only (tests/test_sanitizers_cpu.py); nothing here is shipped.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <stdexcept>
#include <vector>

#include "../../anofox-statistics_amd/duckdb_shim/agg_arena.hpp"

namespace {
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
int g_contexts = 0, g_states = 0, g_host_allocs = 0, g_batch_calls = 0, g_update_calls = 0;
} // namespace

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
bool anofox_hip_agg_state_create(AnofoxHipContext *, size_t p, AnofoxHipBatchOptions opt, int64_t, AnofoxHipAggState **out, AnofoxError *) {
	*out = new AnofoxHipAggState{p, opt.model == ANOFOX_HIP_MODEL_WLS, {}};
	++g_states;
	return true;
}
void anofox_hip_agg_state_destroy(AnofoxHipAggState *s) {
	if (s) --g_states;
	delete s;
}
bool anofox_hip_agg_state_retain_rows(AnofoxHipAggState *, size_t, AnofoxError *) { return true; }
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
bool anofox_hip_agg_state_combine(AnofoxHipAggState *s, int64_t n, const uint32_t *src, const uint32_t *dst, AnofoxError *) {
	for (int64_t i = 0; i < n; ++i) {
		if (src[i] == dst[i]) continue;
		auto &a = s->slots[src[i]];
		auto &b = s->slots[dst[i]];
		b.insert(b.end(), a.begin(), a.end()); // the source's rows count as arriving after the target's
		a.clear();
	}
	return true;
}
bool anofox_hip_agg_state_finalize_host(AnofoxHipAggState *s, int64_t n_slots, double *core, double *, int64_t *unrefined, int32_t *, AnofoxError *) {
	for (int64_t g = 0; g < n_slots; ++g) mock_fit((size_t)g < s->slots.size() ? s->slots[(size_t)g] : std::vector<Row>(), s->p, core + (size_t)g * (s->p + 6));
	if (unrefined) *unrefined = 0;
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
}

#define CHECK(c)                                                        \
	do {                                                                \
		if (!(c)) {                                                     \
			fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); \
			exit(1);                                                    \
		}                                                               \
	} while (0)

// three "threads" with their own state tables over the same keys; thread t's state of key k lives in its own slot
static void scenario(size_t p, AnofoxHipModel model, size_t flush_rows) {
	using anofox_shim::AggArena;
	AnofoxHipBatchOptions opt;
	memset(&opt, 0, sizeof opt);
	opt.model = model;
	opt.fit_intercept = true;
	opt.confidence_level = 0.95;
	const int before_batch = g_batch_calls, before_update = g_update_calls;
	{
		AggArena arena(opt, flush_rows, 0);
		const int T = 3, K = 40;
		std::vector<std::vector<int64_t>> slot_of(T, std::vector<int64_t>(K, -1));
		std::map<int, std::vector<Row>> expect; // key -> rows in the reference's order: thread 0's, then 1's, then 2's
		std::vector<std::map<int, std::vector<Row>>> per_thread(T);
		unsigned long long rng = 12345;
		auto next = [&] { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng >> 33); };
		std::vector<double> x(p);
		// interleave the threads' Update vectors (each vector: one Writer = one lock)
		for (int round = 0; round < 25; ++round) {
			for (int t = 0; t < T; ++t) {
				AggArena::Writer wr(arena);
				const int n = 1 + (int)(next() % 70);
				for (int i = 0; i < n; ++i) {
					const int key = (int)(next() % K);
					if (t == 2 && key % 5 == 0) continue; // thread 2 never sees some keys
					int64_t &slot = slot_of[t][key];
					if (slot < 0) slot = wr.NewSlot();
					const bool accept = next() % 10 != 0; // NULL y / x / w rows are skipped by Update
					if (!accept) continue;
					Row r{(double)(next() % 1000) / 7.0, (double)(next() % 100), model == ANOFOX_HIP_MODEL_WLS ? 0.5 + (double)(next() % 4) : 1.0};
					for (size_t j = 0; j < p; ++j) x[j] = r.x0 + (double)j;
					wr.Append((uint32_t)slot, r.y, x.data(), p, r.w);
					per_thread[t][key].push_back(r);
				}
			}
		}
		CHECK(arena.FeatureCount() == p);
		// Combine threads 1 and 2 into thread 0 the way the glue does: adopt where the target has no slot, merge otherwise
		for (int t = 1; t < T; ++t) {
			std::vector<uint32_t> src, dst;
			for (int k = 0; k < K; ++k) {
				if (slot_of[t][k] < 0) continue;
				if (slot_of[0][k] < 0) { slot_of[0][k] = slot_of[t][k]; continue; }
				src.push_back((uint32_t)slot_of[t][k]);
				dst.push_back((uint32_t)slot_of[0][k]);
			}
			arena.Combine(src.data(), dst.data(), src.size());
		}
		for (int k = 0; k < K; ++k)
			for (int t = 0; t < T; ++t) {
				auto it = per_thread[t].find(k);
				if (it != per_thread[t].end()) expect[k].insert(expect[k].end(), it->second.begin(), it->second.end());
			}
		arena.Solve();
		arena.Solve(); // free the second time
		std::vector<double> want(p + 6);
		for (int k = 0; k < K; ++k) {
			if (slot_of[0][k] < 0) continue;
			mock_fit(expect[k], p, want.data());
			const double *rec = arena.Core((uint32_t)slot_of[0][k]);
			if (want[p + 5] != 0.0) { CHECK(rec == nullptr); continue; }
			CHECK(rec != nullptr);
			for (size_t j = 0; j < p + 6; ++j) CHECK(rec[j] == want[j]); // same rows, same order: the sums are bit-identical
		}
		CHECK(g_batch_calls == before_batch && g_update_calls > before_update); // everything through the state object
	}
	CHECK(g_contexts == 0 && g_states == 0 && g_host_allocs == 0); // everything released
}

int main() {
	scenario(3, ANOFOX_HIP_MODEL_OLS, 64);      // many flushes
	scenario(8, ANOFOX_HIP_MODEL_WLS, 1 << 20); // one flush at Solve
	scenario(12, ANOFOX_HIP_MODEL_OLS, 100);    // wider designs: the same calls (the library keeps the rows instead of moments)
	scenario(20, ANOFOX_HIP_MODEL_WLS, 1 << 20);
	{ // an arena nobody wrote to
		anofox_shim::AggArena arena(AnofoxHipBatchOptions{});
		arena.Solve();
		CHECK(arena.Core(0) == nullptr && arena.SlotCount() == 0);
	}
	{ // inconsistent feature counts
		AnofoxHipBatchOptions opt;
		memset(&opt, 0, sizeof opt);
		anofox_shim::AggArena arena(opt, 16, 0);
		bool threw = false;
		try {
			anofox_shim::AggArena::Writer wr(arena);
			const double x3[3] = {1, 2, 3}, x2[2] = {1, 2};
			wr.Append(wr.NewSlot(), 1.0, x3, 3);
			wr.Append(0, 1.0, x2, 2);
		} catch (const std::invalid_argument &) {
			threw = true;
		}
		CHECK(threw);
	}
	printf("arena_sanitize: all scenarios passed\n");
	return 0;
}
