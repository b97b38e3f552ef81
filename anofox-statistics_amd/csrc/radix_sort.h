// radix_sort.h — a stable LSD radix sort for gfx950, 8-bit digits, written for the two sorts of this library: (slot, row number)
// pairs of an ingest chunk (ingest.hip: 32-bit keys, the value is the row's position) and the refit / log keys of the row log
// (rowlog.hip: 64-bit `group << row_bits | row` keys and 31-bit slot numbers, keys only).  Replaces rocPRIM's device sort
// (rounds 2-3: merge sort for 2^20-element chunks = 20 launches of 7-8 us per chunk).
//
// One pass per digit, three launches per pass, nothing synchronises with the host:
//   histogram  a workgroup counts the digits of ITS tile of the keys (wave-aggregated: the lanes of a wavefront that hold the same
//              digit are found with eight ballots and their leader adds the population count — a sorted or constant input does not
//              serialise on one LDS address) -> hist[digit][block];
//   scan       256 workgroups, one per digit, turn their row of hist into exclusive prefix sums and write the row's total; the
//              scatter workgroups scan the 256 totals themselves: start of digit d + prefix of (d, b) = where block b's keys of
//              digit d start in the output;
//   scatter    the workgroup walks its tile again in 4096-key sub-tiles, wavefront w taking the w-th 1024 keys of the sub-tile in
//              rounds of 64 consecutive keys.  A key's position = start of (digit, block) + keys of that digit in earlier sub-tiles
//              + in earlier wavefronts of the sub-tile + in earlier rounds of the wavefront + in lower lanes of its round — every
//              term a count in memory order, so equal digits keep their order (STABLE: ingest.hip relies on a state's rows staying
//              in arrival order) and the result does not depend on timing.
// Tiles are sized so that at most kMaxBlocks workgroups exist (a row of hist stays within one workgroup's reach whatever n).  Traffic per pass: keys
// (and values) read twice, written once; three passes sort the 20-bit slot numbers of a 4 Mi-row chunk.
#pragma once
#include <hip/hip_runtime.h>

#include <stddef.h>
#include <stdint.h>

namespace anofox {
namespace rsort {

constexpr int kBlock = 256;
constexpr int kItems = 16;                  // keys per lane and sub-tile
constexpr int kSubTile = kBlock * kItems;   // 4096 keys
constexpr int kWaveKeys = 64 * kItems;      // 1024 consecutive keys per wavefront and sub-tile
constexpr unsigned kMaxBlocks = 2048;

struct Plan {
	unsigned n_blocks;
	size_t tile; // keys per workgroup, a multiple of kSubTile
};
inline Plan make_plan(size_t n) {
	const size_t sub = (n + kSubTile - 1) / kSubTile;
	size_t per = (sub + kMaxBlocks - 1) / kMaxBlocks;
	if (per == 0) per = 1;
	Plan p;
	p.tile = per * kSubTile;
	p.n_blocks = (unsigned)((n + p.tile - 1) / p.tile);
	if (p.n_blocks == 0) p.n_blocks = 1;
	return p;
}
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

template <class KEY, bool PAIRS>
inline size_t temp_bytes(size_t n) {
	return align256((size_t)256 * (kMaxBlocks + 1) * sizeof(uint32_t)) + align256(n * sizeof(KEY)) + (PAIRS ? align256(n * sizeof(uint32_t)) : 0);
}

// lanes of this wavefront that are active and hold the same 8-bit digit as this lane
__device__ __forceinline__ unsigned long long same_digit_lanes(unsigned d, bool active) {
	unsigned long long m = __ballot(active);
#pragma unroll
	for (int b = 0; b < 8; ++b) {
		const bool bit = (d >> b) & 1u;
		const unsigned long long v = __ballot(bit);
		m &= bit ? v : ~v;
	}
	return m;
}

template <class KEY>
__device__ __forceinline__ unsigned digit_of(KEY k, unsigned shift, unsigned mask) {
	return (unsigned)(k >> shift) & mask;
}

template <class KEY>
__global__ __launch_bounds__(kBlock) void histogram_kernel(const KEY *keys, size_t n, size_t tile, unsigned shift, unsigned mask, uint32_t *hist) {
	__shared__ uint32_t h[256];
	const int lane = threadIdx.x & 63;
	h[threadIdx.x] = 0;
	__syncthreads();
	const size_t lo = (size_t)blockIdx.x * tile, hi = lo + tile < n ? lo + tile : n;
	for (size_t i0 = lo; i0 < hi; i0 += kBlock) { // (every lane of the workgroup runs every trip: the ballots need them)
		const size_t i = i0 + threadIdx.x;
		const bool act = i < hi;
		const unsigned d = act ? digit_of(keys[i], shift, mask) : 0u;
		const unsigned long long m = same_digit_lanes(d, act);
		if (act && (m & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&h[d], (uint32_t)__popcll(m));
	}
	__syncthreads();
	hist[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

// hist[d][0 .. nb) -> exclusive prefix sums within the row of digit d, totals[d] = the row's sum.  One workgroup per digit: the
// 256 rows are scanned side by side and the 256 totals are scanned by every scatter workgroup itself ((r4) first version: ONE
// workgroup walked all 256 x nb counters, each thread through 256 dependent loads — 0.3 of a pass's 0.45 ms).
template <int UNUSED = 0> // (a template only so that two translation units may include this header)
__global__ __launch_bounds__(kBlock) void scan_rows_kernel(uint32_t *hist, unsigned nb, uint32_t *totals) {
	__shared__ uint32_t part[kBlock];
	uint32_t *row = hist + (size_t)blockIdx.x * nb;
	const unsigned span = (nb + kBlock - 1) / kBlock; // <= kMaxBlocks / kBlock entries per thread
	const unsigned lo = threadIdx.x * span < nb ? threadIdx.x * span : nb, hi = lo + span < nb ? lo + span : nb;
	uint32_t s = 0;
	for (unsigned i = lo; i < hi; ++i) s += row[i];
	part[threadIdx.x] = s;
	__syncthreads();
	for (int off = 1; off < kBlock; off <<= 1) {
		const uint32_t v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	uint32_t base = threadIdx.x ? part[threadIdx.x - 1] : 0u;
	for (unsigned i = lo; i < hi; ++i) {
		const uint32_t c = row[i];
		row[i] = base;
		base += c;
	}
	if (threadIdx.x == kBlock - 1) totals[blockIdx.x] = part[kBlock - 1];
}

// IOTA: the value of a key is its position in `kin` (the first pass of a pairs sort)
template <class KEY, bool PAIRS, bool IOTA>
__global__ __launch_bounds__(kBlock) void scatter_kernel(const KEY *kin, const uint32_t *vin, KEY *kout, uint32_t *vout, size_t n, size_t tile,
                                                         unsigned shift, unsigned mask, const uint32_t *hist, const uint32_t *totals) {
	__shared__ uint32_t cnt[4][256]; // per wavefront: digit counts of its 1024 keys, then where they start
	__shared__ uint32_t run[256];    // where the workgroup's next key of each digit goes
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	{ // where digit d starts in the output: the exclusive scan of the 256 row totals (every workgroup does it for itself)
		const uint32_t own = totals[threadIdx.x];
		run[threadIdx.x] = own;
		__syncthreads();
		for (int off = 1; off < kBlock; off <<= 1) {
			const uint32_t v = (int)threadIdx.x >= off ? run[threadIdx.x - off] : 0u;
			__syncthreads();
			run[threadIdx.x] += v;
			__syncthreads();
		}
		const uint32_t start = run[threadIdx.x] - own;
		__syncthreads();
		run[threadIdx.x] = start + hist[(size_t)threadIdx.x * gridDim.x + blockIdx.x];
	}
	const size_t lo = (size_t)blockIdx.x * tile, hi = lo + tile < n ? lo + tile : n;
	for (size_t s0 = lo; s0 < hi; s0 += kSubTile) {
#pragma unroll
		for (int k = 0; k < 4; ++k) cnt[w][lane + 64 * k] = 0;
		KEY key[kItems];
		uint32_t val[kItems];
		uint32_t rank[kItems];
		const size_t wbase = s0 + (size_t)w * kWaveKeys;
#pragma unroll
		for (int r = 0; r < kItems; ++r) {
			const size_t i = wbase + (size_t)r * 64 + lane;
			const bool act = i < hi;
			key[r] = act ? kin[i] : (KEY)0;
			if (PAIRS) val[r] = IOTA ? (uint32_t)i : (act ? vin[i] : 0u);
			const unsigned d = digit_of(key[r], shift, mask);
			const unsigned long long m = same_digit_lanes(d, act);
			// (one wavefront, LDS in program order: every lane reads the count before the leader moves it on)
			const uint32_t base = cnt[w][d];
			const uint32_t below = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
			rank[r] = base + below;
			if (act && below == 0u) cnt[w][d] = base + (uint32_t)__popcll(m);
		}
		__syncthreads();
		{
			uint32_t o = run[threadIdx.x];
#pragma unroll
			for (int w2 = 0; w2 < 4; ++w2) {
				const uint32_t c = cnt[w2][threadIdx.x];
				cnt[w2][threadIdx.x] = o;
				o += c;
			}
			run[threadIdx.x] = o;
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < kItems; ++r) {
			const size_t i = wbase + (size_t)r * 64 + lane;
			if (i < hi) {
				const uint32_t pos = cnt[w][digit_of(key[r], shift, mask)] + rank[r];
				kout[pos] = key[r];
				if (PAIRS) vout[pos] = val[r];
			}
		}
		__syncthreads();
	}
}

// Sorts keys_in[0 .. n) by their bits [0, end_bit) into keys_out (stable); PAIRS: vals_out[k] = position in keys_in of the key that
// ends up at k.  keys_in is only read.  `temp` holds temp_bytes<KEY, PAIRS>(n) bytes.  n < 2^32.
template <class KEY, bool PAIRS>
inline hipError_t sort(const KEY *keys_in, KEY *keys_out, uint32_t *vals_out, size_t n, unsigned end_bit, void *temp, size_t temp_size, hipStream_t st) {
	if (n == 0) return hipSuccess;
	if (n >= ((size_t)1 << 32) || temp_size < temp_bytes<KEY, PAIRS>(n) || end_bit > 8 * sizeof(KEY)) return hipErrorInvalidValue;
	const Plan pl = make_plan(n);
	char *t = static_cast<char *>(temp);
	uint32_t *hist = reinterpret_cast<uint32_t *>(t);
	uint32_t *totals = hist + (size_t)256 * kMaxBlocks;
	t += align256((size_t)256 * (kMaxBlocks + 1) * sizeof(uint32_t));
	KEY *ktmp = reinterpret_cast<KEY *>(t);
	t += align256(n * sizeof(KEY));
	uint32_t *vtmp = PAIRS ? reinterpret_cast<uint32_t *>(t) : nullptr;
	const unsigned passes = end_bit == 0 ? 1u : (end_bit + 7u) / 8u;
	const KEY *ksrc = keys_in;
	const uint32_t *vsrc = nullptr;
	for (unsigned j = 0; j < passes; ++j) {
		const unsigned shift = 8u * j;
		const unsigned bits = end_bit > shift ? (end_bit - shift < 8u ? end_bit - shift : 8u) : 0u;
		const unsigned mask = (1u << bits) - 1u;
		const bool to_out = ((passes - 1u - j) & 1u) == 0u;
		KEY *kdst = to_out ? keys_out : ktmp;
		uint32_t *vdst = to_out ? vals_out : vtmp;
		hipLaunchKernelGGL((histogram_kernel<KEY>), dim3(pl.n_blocks), dim3(kBlock), 0, st, ksrc, n, pl.tile, shift, mask, hist);
		hipLaunchKernelGGL((scan_rows_kernel<0>), dim3(256), dim3(kBlock), 0, st, hist, pl.n_blocks, totals);
		if (j == 0) hipLaunchKernelGGL((scatter_kernel<KEY, PAIRS, true>), dim3(pl.n_blocks), dim3(kBlock), 0, st, ksrc, vsrc, kdst, vdst, n, pl.tile, shift, mask, hist, totals);
		else hipLaunchKernelGGL((scatter_kernel<KEY, PAIRS, false>), dim3(pl.n_blocks), dim3(kBlock), 0, st, ksrc, vsrc, kdst, vdst, n, pl.tile, shift, mask, hist, totals);
		ksrc = kdst;
		vsrc = vdst;
	}
	return hipGetLastError();
}

} // namespace rsort
} // namespace anofox
