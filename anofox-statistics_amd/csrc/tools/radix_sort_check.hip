// radix_sort_check.hip — radix_sort.h against std::stable_sort on the GPU: sizes around the tile / sub-tile / wavefront
// boundaries, skewed digit distributions (constant, sorted, reversed, two values), every end_bit class, 32-bit pairs (the value
// must be the ORIGINAL position: stability) and 64-bit / 31-bit keys.  Prints "radix_sort_check: ok (N cases)" and exits 0.
// Test infrastructure (tests/test_gpu_streaming.py runs it); also times the 4 Mi-element sort of an ingest pass.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <numeric>
#include <vector>

#include "../radix_sort.h"

using namespace anofox;

#define HIP_OK(x)                                                                  \
	do {                                                                           \
		hipError_t e_ = (x);                                                       \
		if (e_ != hipSuccess) {                                                    \
			fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
			exit(2);                                                               \
		}                                                                          \
	} while (0)

static unsigned long long g_rng = 88172645463325252ull;
static unsigned long long rnd() {
	g_rng ^= g_rng << 13;
	g_rng ^= g_rng >> 7;
	g_rng ^= g_rng << 17;
	return g_rng;
}

template <class KEY>
static std::vector<KEY> make_keys(size_t n, int pattern, unsigned end_bit) {
	std::vector<KEY> k(n);
	const KEY mask = end_bit >= 8 * sizeof(KEY) ? ~(KEY)0 : (((KEY)1 << end_bit) - 1);
	for (size_t i = 0; i < n; ++i) {
		KEY v;
		switch (pattern) {
		case 0: v = (KEY)rnd(); break;                                  // uniform
		case 1: v = (KEY)7; break;                                      // constant
		case 2: v = (KEY)i; break;                                      // sorted
		case 3: v = (KEY)(n - i); break;                                // reversed
		case 4: v = (KEY)((rnd() & 1) ? 0x00ff00ffull : 0x01000100ull); break; // two values
		case 5: v = (KEY)(rnd() % 1000003ull); break;                   // slot numbers of a million states
		default: v = (KEY)((rnd() % 3 == 0) ? rnd() : (rnd() & 0xff));  // heavy low digits
		}
		k[i] = v & mask;
	}
	return k;
}

template <class KEY, bool PAIRS>
static int check(size_t n, int pattern, unsigned end_bit, bool high_garbage) {
	std::vector<KEY> k = make_keys<KEY>(n, pattern, end_bit);
	if (high_garbage && end_bit < 8 * sizeof(KEY))
		for (auto &v : k) v |= (KEY)(rnd() & 0xf) << end_bit; // bits at and above end_bit must not influence the order
	KEY *d_in = nullptr, *d_out = nullptr;
	uint32_t *d_val = nullptr;
	void *d_tmp = nullptr;
	const size_t tb = rsort::temp_bytes<KEY, PAIRS>(n);
	HIP_OK(hipMalloc(&d_in, (n + 1) * sizeof(KEY)));
	HIP_OK(hipMalloc(&d_out, (n + 1) * sizeof(KEY)));
	HIP_OK(hipMalloc(&d_val, (n + 1) * sizeof(uint32_t)));
	HIP_OK(hipMalloc(&d_tmp, tb));
	HIP_OK(hipMemcpy(d_in, k.data(), n * sizeof(KEY), hipMemcpyHostToDevice));
	HIP_OK(hipMemset(d_out, 0xee, (n + 1) * sizeof(KEY)));
	HIP_OK((rsort::sort<KEY, PAIRS>(d_in, d_out, d_val, n, end_bit, d_tmp, tb, nullptr)));
	HIP_OK(hipDeviceSynchronize());
	std::vector<KEY> got(n + 1);
	std::vector<uint32_t> gv(n + 1);
	HIP_OK(hipMemcpy(got.data(), d_out, (n + 1) * sizeof(KEY), hipMemcpyDeviceToHost));
	if (PAIRS) HIP_OK(hipMemcpy(gv.data(), d_val, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
	std::vector<KEY> in_after(n);
	HIP_OK(hipMemcpy(in_after.data(), d_in, n * sizeof(KEY), hipMemcpyDeviceToHost));
	HIP_OK(hipFree(d_in));
	HIP_OK(hipFree(d_out));
	HIP_OK(hipFree(d_val));
	HIP_OK(hipFree(d_tmp));
	const KEY mask = end_bit >= 8 * sizeof(KEY) ? ~(KEY)0 : (((KEY)1 << end_bit) - 1);
	std::vector<uint32_t> order(n);
	std::iota(order.begin(), order.end(), 0u);
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (k[a] & mask) < (k[b] & mask); });
	int bad = 0;
	for (size_t i = 0; i < n && bad < 5; ++i) {
		if (got[i] != k[order[i]] || (PAIRS && gv[i] != order[i])) {
			fprintf(stderr, "n=%zu pattern=%d end_bit=%u key=%zu pairs=%d: position %zu holds %llx / %u, expected %llx / %u\n", n, pattern, end_bit,
			        sizeof(KEY), (int)PAIRS, i, (unsigned long long)got[i], PAIRS ? gv[i] : 0u, (unsigned long long)k[order[i]], order[i]);
			++bad;
		}
	}
	unsigned char guard[sizeof(KEY)];
	memcpy(guard, &got[n], sizeof(KEY));
	for (size_t b = 0; b < sizeof(KEY); ++b)
		if (guard[b] != 0xee) {
			fprintf(stderr, "n=%zu: wrote past the output\n", n);
			++bad;
			break;
		}
	if (in_after != k) {
		fprintf(stderr, "n=%zu: the input was modified\n", n);
		++bad;
	}
	return bad;
}

int main(int argc, char **argv) {
	int cases = 0, bad = 0;
	const size_t sizes[] = {1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 12289, 100000, (size_t)1 << 20, ((size_t)1 << 22) + 3,
	                        (size_t)rsort::kSubTile * rsort::kMaxBlocks + 5};
	for (size_t n : sizes) {
		for (int pattern = 0; pattern < 7; ++pattern) {
			if (n > 200000 && pattern != 0 && pattern != 1 && pattern != 5) continue;
			for (unsigned end_bit : {20u, 32u, 1u, 9u}) {
				if (n > 200000 && end_bit != 20u && end_bit != 32u) continue;
				bad += check<uint32_t, true>(n, pattern, end_bit, true);
				++cases;
			}
			bad += check<uint32_t, false>(n, pattern, 31u, false);
			bad += check<uint64_t, false>(n, pattern, 47u, true);
			cases += 2;
			if (n <= 100000) {
				bad += check<uint64_t, false>(n, pattern, 64u, false);
				bad += check<uint32_t, true>(n, pattern, 0u, true); // no key bits: a stable copy
				cases += 2;
			}
		}
	}
	if (bad) {
		fprintf(stderr, "radix_sort_check: %d mismatches\n", bad);
		return 1;
	}
	// timing: the sort of one ingest pass (4 Mi slot numbers below 2^20, pairs)
	if (argc > 1) {
		const size_t n = (size_t)1 << 22;
		std::vector<uint32_t> k = make_keys<uint32_t>(n, 5, 20);
		uint32_t *d_in, *d_out, *d_val;
		void *d_tmp;
		const size_t tb = rsort::temp_bytes<uint32_t, true>(n);
		HIP_OK(hipMalloc(&d_in, n * 4));
		HIP_OK(hipMalloc(&d_out, n * 4));
		HIP_OK(hipMalloc(&d_val, n * 4));
		HIP_OK(hipMalloc(&d_tmp, tb));
		HIP_OK(hipMemcpy(d_in, k.data(), n * 4, hipMemcpyHostToDevice));
		for (int rep = 0; rep < 3; ++rep) {
			HIP_OK(hipDeviceSynchronize());
			const auto t0 = std::chrono::steady_clock::now();
			for (int it = 0; it < 10; ++it) HIP_OK((rsort::sort<uint32_t, true>(d_in, d_out, d_val, n, 20u, d_tmp, tb, nullptr)));
			HIP_OK(hipDeviceSynchronize());
			const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 10;
			printf("sort of 2^22 (slot, row) pairs, 20 key bits: %.3f ms = %.2f G keys/s\n", ms, n / ms / 1e6);
		}
	}
	printf("radix_sort_check: ok (%d cases)\n", cases);
	return 0;
}
