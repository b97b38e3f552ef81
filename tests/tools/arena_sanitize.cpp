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
#include <thread>
#include <vector>

#include "../../anofox-statistics_amd/duckdb_shim/agg_arena.hpp"

#include "mock_abi.hpp"

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
		std::vector<double> want(p + 6), got(p + 6);
		const uint64_t fits0 = arena.FitCalls();
		for (int pass = 0; pass < 2; ++pass) { // the second Finalize of unchanged states costs no library call
			for (int k = 0; k < K; ++k) {
				if (slot_of[0][k] < 0) continue;
				mock_fit(expect[k], p, want.data());
				const uint32_t sl = (uint32_t)slot_of[0][k];
				int status = -1;
				arena.Fetch(&sl, 1, got.data(), nullptr, &status);
				CHECK(status == (int)want[p + 5]);
				if (status != 0) continue;
				for (size_t j = 0; j < p + 6; ++j) CHECK(got[j] == want[j]); // same rows, same order: the sums are bit-identical
			}
			CHECK(arena.FitCalls() == fits0 + 1);
		}
		CHECK(g_batch_calls == before_batch && g_update_calls > before_update); // everything through the state object
	}
	CHECK(g_contexts == 0 && g_states == 0 && g_host_allocs == 0); // everything released
}

// The aggregate as a window function, the way DuckDB's naive window aggregator drives it (the reference's test:
// test/sql/comprehensive_tests.test:425-444, ROWS BETWEEN 4 PRECEDING AND CURRENT ROW over 20 rows): per output row a
// state is initialised, fed its frame, finalized and destroyed.  Slots must be reused, every Finalize must fit only the
// new state, and the records must be the frames' own.
static void windowed() {
	using anofox_shim::AggArena;
	AnofoxHipBatchOptions opt;
	memset(&opt, 0, sizeof opt);
	opt.fit_intercept = true;
	AggArena arena(opt, 64, 0);
	const int N = 200, W = 5;
	const size_t p = 1;
	const uint64_t fitted0 = arena.SlotsFitted();
	int n5 = 0;
	for (int i = 0; i < N; ++i) {
		uint32_t slot;
		std::vector<Row> frame;
		{
			AggArena::Writer wr(arena);
			slot = wr.NewSlot();
			for (int r = std::max(0, i - W + 1); r <= i; ++r) {
				const double x = (double)(r + 1), y = 2.0 * x + 1.0;
				wr.Append(slot, y, &x, 1);
				frame.push_back(Row{y, x, 1.0});
			}
		}
		double rec[7], want[7];
		int status = -1;
		arena.Fetch(&slot, 1, rec, nullptr, &status);
		mock_fit(frame, p, want);
		CHECK(status == (int)want[p + 5]);
		if (status == 0) {
			CHECK(rec[0] == want[0] && rec[p + 4] == want[p + 4]);
			if (rec[p + 4] == 5.0) ++n5;
		}
		arena.ReleaseSlot(slot);
	}
	CHECK(n5 == N - W + 1);
	CHECK(arena.SlotCount() <= 2);                    // slots are handed out again: no growth with the number of frames
	CHECK(arena.SlotsFitted() - fitted0 == (uint64_t)N); // one slot fitted per Finalize, not slots x frames
	CHECK(arena.LiveSlots() == 0);
	// a segment tree's use of Combine: the same source feeds several targets of one call and lives on
	{
		AggArena tree(opt, 64, 0);
		uint32_t leaf[2], res[3];
		{
			AggArena::Writer wr(tree);
			for (int k = 0; k < 2; ++k) {
				leaf[k] = wr.NewSlot();
				for (int r = 0; r < 4; ++r) {
					const double x = (double)(4 * k + r);
					wr.Append(leaf[k], 10.0 * k + r, &x, 1);
				}
			}
			for (int k = 0; k < 3; ++k) res[k] = wr.NewSlot();
		}
		const uint32_t src[4] = {leaf[0], leaf[0], leaf[1], leaf[1]}, dst[4] = {res[0], res[1], res[1], res[2]};
		tree.Combine(src, dst, 4, true);
		double rec[3 * 7];
		int st[3];
		tree.Fetch(res, 3, rec, nullptr, st);
		CHECK(st[0] == 0 && st[1] == 0 && st[2] == 0);
		CHECK(rec[0 * 7 + 5] == 4.0 && rec[1 * 7 + 5] == 8.0 && rec[2 * 7 + 5] == 4.0);
		CHECK(rec[0 * 7] == 0 + 1 + 2 + 3 && rec[2 * 7] == 10 + 11 + 12 + 13 && rec[1 * 7] == rec[0 * 7] + rec[2 * 7]);
		double lrec[2 * 7];
		int lst[2];
		tree.Fetch(leaf, 2, lrec, nullptr, lst);
		CHECK(lst[0] == 0 && lrec[5] == 4.0 && lrec[7 + 5] == 4.0); // the leaves are still what they were
	}
}

// Update from several threads at once (each with its own states, as DuckDB's thread-local hash tables), small chunks so
// that shipping happens under contention; then Combine and Finalize.  Run under ASan / UBSan (and TSan-clean by design:
// appends touch only the calling thread's chunk).
static void concurrent() {
	using anofox_shim::AggArena;
	AnofoxHipBatchOptions opt;
	memset(&opt, 0, sizeof opt);
	opt.fit_intercept = true;
	AggArena arena(opt, 128, 0);
	const int T = 8, K = 50, V = 60;
	const size_t p = 3;
	std::vector<std::vector<int64_t>> slot_of(T, std::vector<int64_t>(K, -1));
	std::vector<std::vector<std::vector<Row>>> rows(T, std::vector<std::vector<Row>>(K));
	std::vector<std::thread> th;
	for (int t = 0; t < T; ++t)
		th.emplace_back([&, t] {
			unsigned long long rng = 777 + t;
			auto next = [&] { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng >> 33); };
			double x[3];
			for (int v = 0; v < V; ++v) {
				AggArena::Writer wr(arena);
				for (int i = 0; i < 97; ++i) {
					const int key = (int)(next() % K);
					int64_t &slot = slot_of[t][key];
					if (slot < 0) slot = wr.NewSlot();
					Row r{(double)(next() % 1000) / 3.0, (double)(next() % 100), 1.0};
					x[0] = r.x0; x[1] = 1.0; x[2] = 2.0;
					wr.Append((uint32_t)slot, r.y, x, p);
					rows[t][key].push_back(r);
				}
			}
		});
	for (auto &t : th) t.join();
	CHECK(arena.RowsAccepted() == (uint64_t)T * V * 97);
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
	std::vector<double> want(p + 6), got(p + 6);
	for (int k = 0; k < K; ++k) {
		if (slot_of[0][k] < 0) continue;
		std::vector<Row> all;
		for (int t = 0; t < T; ++t) all.insert(all.end(), rows[t][k].begin(), rows[t][k].end());
		mock_fit(all, p, want.data());
		const uint32_t sl = (uint32_t)slot_of[0][k];
		int status = -1;
		arena.Fetch(&sl, 1, got.data(), nullptr, &status);
		CHECK(status == (int)want[p + 5]);
		for (size_t j = 0; j < p + 6; ++j) CHECK(got[j] == want[j]);
	}
}

int main() {
	scenario(3, ANOFOX_HIP_MODEL_OLS, 64);      // many flushes
	scenario(8, ANOFOX_HIP_MODEL_WLS, 1 << 20); // one flush at Solve
	scenario(12, ANOFOX_HIP_MODEL_OLS, 100);    // wider designs: the same calls (the library keeps the rows instead of moments)
	scenario(20, ANOFOX_HIP_MODEL_WLS, 1 << 20);
	{ // an arena nobody wrote to
		anofox_shim::AggArena arena(AnofoxHipBatchOptions{});
		const uint32_t sl = 0;
		int status = 0;
		arena.Fetch(&sl, 1, nullptr, nullptr, &status);
		CHECK(status == ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS && arena.SlotCount() == 0);
	}
	windowed();
	concurrent();
	{ // the device state cannot be created: the Update throws, nothing leaks, and the arena is not left half-built
		AnofoxHipBatchOptions opt;
		memset(&opt, 0, sizeof opt);
		anofox_shim::AggArena arena(opt, 16, 0);
		const double x2[2] = {1, 2};
		g_fail_state_create = 1;
		bool threw = false;
		try {
			anofox_shim::AggArena::Writer wr(arena);
			wr.Append(wr.NewSlot(), 1.0, x2, 2);
		} catch (const std::runtime_error &) {
			threw = true;
		}
		g_fail_state_create = 0;
		CHECK(threw && arena.FeatureCount() == 0 && g_contexts == 0 && g_states == 0);
		{ // ... and works once the device has room again
			anofox_shim::AggArena::Writer wr(arena);
			wr.Append(0, 1.0, x2, 2);
			wr.Append(0, 2.0, x2, 2);
		}
		const uint32_t sl = 0;
		int status = -1;
		double rec[8];
		arena.Fetch(&sl, 1, rec, nullptr, &status);
		CHECK(status == 0 && rec[0] == 3.0);
	}
	CHECK(g_contexts == 0 && g_states == 0 && g_host_allocs == 0);
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
