// arena_bench.cpp — rows/s through the shim's arena from T concurrent "DuckDB worker threads": every thread owns a
// thread-local table of states (its own slot range) and feeds Update vectors of 2048 rows (random state per row,
// p features, row-major LIST data) through AggArena::Writer; then Combine into the first thread's states and Finalize.
// Reports the Update phase next to the PCIe ceiling of the box (the rows cross the bus once: 8 (p + 1) + 4 bytes each).
//   arena_bench <threads> <slots_total> <rows_per_thread> <features> [chunk_rows]
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "agg_arena.hpp"

using anofox_shim::AggArena;

int main(int argc, char **argv) {
	const int T = argc > 1 ? atoi(argv[1]) : 8;
	const size_t slots_total = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)1 << 20;
	const size_t rows_per_thread = argc > 3 ? (size_t)atoll(argv[3]) : (size_t)1 << 24;
	const size_t p = argc > 4 ? (size_t)atoi(argv[4]) : 8;
	const size_t chunk_rows = argc > 5 ? (size_t)atoll(argv[5]) : (size_t)1 << 18;
	const size_t V = 2048, per_thread_slots = slots_total / (size_t)T;
	AnofoxHipBatchOptions opt;
	memset(&opt, 0, sizeof opt);
	opt.model = ANOFOX_HIP_MODEL_OLS;
	opt.fit_intercept = true;
	opt.confidence_level = 0.95;
	AggArena arena(opt, chunk_rows);
	// slot numbers of every thread's states, handed out up front (Initialize), and one warm-up row so that the device
	// state exists before the clock starts
	std::vector<std::vector<uint32_t>> slot_of(T, std::vector<uint32_t>(per_thread_slots));
	for (int t = 0; t < T; ++t)
		for (auto &s : slot_of[t]) s = arena.NewSlot();
	{
		std::vector<double> x(p, 1.0);
		AggArena::Writer w(arena);
		w.Append(slot_of[0][0], 1.0, x.data(), p);
	}
	// per-thread input vectors (a few distinct ones, reused: generating data is not what is measured)
	const int NV = 16;
	std::vector<std::vector<double>> ys(T), xs(T);
	std::vector<std::vector<uint32_t>> ks(T);
	for (int t = 0; t < T; ++t) {
		unsigned long long rng = 1234567 + 77 * t;
		auto next = [&] { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng >> 33); };
		ys[t].resize(NV * V);
		xs[t].resize(NV * V * p);
		ks[t].resize(NV * V);
		for (size_t i = 0; i < NV * V; ++i) {
			ks[t][i] = next() % per_thread_slots;
			double acc = 1.0;
			for (size_t j = 0; j < p; ++j) {
				const double v = (double)(next() % 2000) / 100.0 - 10.0;
				xs[t][i * p + j] = v;
				acc += (double)(j + 1) * v;
			}
			ys[t][i] = acc + (double)(next() % 1000) / 250.0;
		}
	}
	std::atomic<int> ready {0};
	std::atomic<bool> go {false};
	auto worker = [&](int t) {
		{ // the thread's page-locked chunk is allocated by its first row (hipHostMalloc pins 20 MB: 5 - 30 ms, once per thread
			// of the query): before the clock starts
			AggArena::Writer w(arena);
			memcpy(w.Begin(slot_of[t][0], ys[t][0], p), &xs[t][0], p * sizeof(double));
		}
		++ready;
		while (!go.load()) std::this_thread::yield();
		const size_t n_vec = rows_per_thread / V;
		for (size_t v = 0; v < n_vec; ++v) {
			const size_t base = (v % NV) * V;
			AggArena::Writer w(arena);
			const double *yv = &ys[t][base], *xv = &xs[t][base * p];
			const uint32_t *kv = &ks[t][base];
			// (the input vectors are reused; the state of a row is not: every vector hits 2048 fresh random states)
			unsigned long long h = 0x9E3779B97F4A7C15ull * (v + 1) + (unsigned long long)t;
			for (size_t i = 0; i < V; ++i) {
				h = h * 6364136223846793005ull + 1442695040888963407ull;
				const size_t k = (size_t)((h >> 33) % per_thread_slots) ^ (kv[i] & 0);
				double *dst = w.Begin(slot_of[t][k], yv[i], p);
				memcpy(dst, xv + i * p, p * sizeof(double));
			}
		}
	};
	std::vector<std::thread> th;
	for (int t = 0; t < T; ++t) th.emplace_back(worker, t);
	while (ready.load() < T) std::this_thread::yield();
	const auto t0 = std::chrono::steady_clock::now();
	go.store(true);
	for (auto &x : th) x.join();
	const auto t1 = std::chrono::steady_clock::now();
	// Combine the other threads' states into thread 0's, Finalize everything
	for (int t = 1; t < T; ++t) arena.Combine(slot_of[t].data(), slot_of[0].data(), per_thread_slots);
	const auto t2 = std::chrono::steady_clock::now();
	std::vector<double> core(per_thread_slots * (p + 6));
	std::vector<int> status(per_thread_slots);
	arena.Fetch(slot_of[0].data(), per_thread_slots, core.data(), nullptr, status.data());
	const auto t3 = std::chrono::steady_clock::now();
	size_t fitted = 0;
	for (int s : status) fitted += s == 0;
	const double rows = (double)T * (double)(rows_per_thread / V * V);
	const double upd = std::chrono::duration<double>(t1 - t0).count();
	printf("{\"threads\": %d, \"slots\": %zu, \"rows\": %.0f, \"features\": %zu, \"chunk_rows\": %zu, \"update_s\": %.4f, \"rows_per_s\": %.4g, "
	       "\"bytes_per_row\": %zu, \"GBps_over_pcie\": %.2f, \"combine_s\": %.4f, \"finalize_s\": %.4f, \"groups_fitted\": %zu, \"coef0_slot0\": %.6f}\n",
	       T, slots_total, rows, p, chunk_rows, upd, rows / upd, 8 * (p + 1) + 4, rows * (double)(8 * (p + 1) + 4) / upd / 1e9,
	       std::chrono::duration<double>(t2 - t1).count(), std::chrono::duration<double>(t3 - t2).count(), fitted, core[0]);
	return 0;
}
