// rowlog.hip — the optional row log of a streaming aggregate state (agg_state.hip, anofox_hip_agg_state_retain_rows).
//
// The reference's aggregate state IS a row buffer (src/aggregate_functions/ols_aggregate.cpp:19-42); the streaming
// state replaces it by O(p^2) moments, which is what makes Update cheap — and what leaves Finalize without the rows
// the batch path's refinement passes re-read for ill-conditioned or (nearly) exactly fitting groups.  With 288 GB of
// HBM the rows can simply stay: Update appends every chunk to slabs in arrival order (device-to-device copies, no
// kernel), and Finalize pulls out the rows of exactly the groups its solve queued for refinement:
//   mark      dense[slot] = k for the k-th queued slot (ascending), -1 elsewhere
//   count     rows of the log that belong to a queued slot (one pass over the 4-byte slot column)
//   fill      key = k << row_bits | global row number (row_bits = the bits the log's row count needs, so that
//             2^31 slots and 2^33 rows both fit a 64-bit key), appended through a wave-aggregated counter
//   sort      rocPRIM radix sort of the keys: groups contiguous, arrival order inside a group, whatever order the
//             appends happened in
//   gather    key -> slab row -> y / w / x columns of an ordinary batch (column-major), row_offsets by binary search
// and hands that batch to the unchanged batch path (accumulate -> solve -> refine), whose records replace the
// streaming ones.  Combine re-labels the source slots' rows in the log (one pass over the slot column).
#include <hip/hip_runtime.h>

#include "common.h"
#include "radix_sort.h"

namespace anofox {

namespace {

constexpr int kLogBlock = 256;

__global__ void rowlog_dense_kernel(const int32_t *sorted_slots, int64_t k_n, int32_t *dense) {
	const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k < k_n) dense[sorted_slots[k]] = (int32_t)k;
}

__global__ void rowlog_remap_kernel(uint32_t *slot, int64_t n, const uint32_t *remap, int64_t n_slots) {
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const uint32_t s = slot[i];
		if ((int64_t)s < n_slots) {
			const uint32_t t = remap[s];
			if (t != s) slot[i] = t;
		}
	}
}

__global__ void rowlog_iota_kernel(int32_t *v, int64_t n) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) v[i] = (int32_t)i;
}

__global__ void rowlog_remap_init_kernel(uint32_t *remap, int64_t n_slots) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_slots) remap[i] = (uint32_t)i;
}

__global__ void rowlog_remap_pairs_kernel(uint32_t *remap, const uint32_t *src, const uint32_t *dst, int64_t n_pairs) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_pairs) remap[src[i]] = dst[i];
}

// FILL = false: count the selected rows;  FILL = true: append their keys.
// One workgroup owns kSelectTile consecutive rows and makes ONE atomic on the shared counter for them (a wave-level
// append measured 12 ms for 64 M shuffled rows with 1 % selected: 370 k same-address atomics per pass); the keys of
// a tile land in one contiguous range, in any tile order — the sort that follows does not care.
constexpr int kSelectItems = 16;
constexpr int kSelectTile = kLogBlock * kSelectItems;

// -1: the row is not selected; -2: it names a slot >= n_slots (Update's mistake, reported by the log-only Finalize)
__device__ __forceinline__ int32_t rowlog_group_of(const uint32_t *slot, const uint8_t *valid, int64_t i, int64_t n, const int32_t *dense,
                                                   int64_t n_slots) {
	if (i >= n || !valid[i]) return -1;
	const uint32_t s = slot[i];
	return (int64_t)s < n_slots ? dense[s] : -2;
}

template <bool FILL>
__global__ void __launch_bounds__(kLogBlock) rowlog_select_kernel(const uint32_t *slot, const uint8_t *valid, int64_t n, int64_t base_row,
                                                                  const int32_t *dense, int64_t n_slots, unsigned long long *counter,
                                                                  uint64_t *keys, unsigned row_bits) {
	__shared__ unsigned long long s_base;
	__shared__ int s_wave[kLogBlock / 64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// wave w of the block reads rows tile0 + (it * 4 + w) * 64 + lane
	const int64_t tile0 = (int64_t)blockIdx.x * kSelectTile;
	int mine = 0; // selected rows of this wave (uniform)
	bool oob = false;
#pragma unroll 4
	for (int it = 0; it < kSelectItems; ++it) {
		const int64_t i = tile0 + (int64_t)((it * (kLogBlock / 64) + wave) * 64 + lane);
		const int32_t k = rowlog_group_of(slot, valid, i, n, dense, n_slots);
		mine += __popcll(__ballot(k >= 0));
		oob = oob || k == -2;
	}
	if (!FILL && __ballot(oob) != 0ull && lane == 0) atomicAdd(counter + 1, 1ull); // counter[1]: rows of slots >= n_slots
	if (lane == 0) s_wave[wave] = mine;
	__syncthreads();
	if (threadIdx.x == 0) {
		int total = 0;
		for (int w = 0; w < kLogBlock / 64; ++w) total += s_wave[w];
		s_base = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
	}
	if (!FILL) return;
	__syncthreads();
	unsigned long long at = s_base;
	for (int w = 0; w < wave; ++w) at += (unsigned long long)s_wave[w];
	if (mine == 0) return;
	for (int it = 0; it < kSelectItems; ++it) {
		const int64_t i = tile0 + (int64_t)((it * (kLogBlock / 64) + wave) * 64 + lane);
		const int32_t k = rowlog_group_of(slot, valid, i, n, dense, n_slots);
		const uint64_t m = __ballot(k >= 0);
		if (k >= 0) keys[at + (unsigned long long)__popcll(m & ((1ull << lane) - 1))] = ((uint64_t)k << row_bits) | (uint64_t)(base_row + i);
		at += (unsigned long long)__popcll(m);
	}
}

__device__ __forceinline__ int slab_of_row(const RowLogSlab *slabs, int n_slabs, int64_t row) {
	int lo = 0, hi = n_slabs - 1;
	while (lo < hi) { // last slab whose first row is <= row
		const int mid = (lo + hi + 1) >> 1;
		if (slabs[mid].first_row <= row) lo = mid; else hi = mid - 1;
	}
	return lo;
}

__global__ void rowlog_gather_kernel(const uint64_t *keys, int64_t m, const RowLogSlab *slabs, int n_slabs, int p, int weighted, double *y,
                                     double *x_cols, size_t col_stride, double *w, unsigned row_bits) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= m) return;
	const int64_t row = (int64_t)(keys[i] & ((1ull << row_bits) - 1));
	const RowLogSlab sl = slabs[slab_of_row(slabs, n_slabs, row)];
	const int64_t r = row - sl.first_row;
	y[i] = sl.y[r];
	if (weighted) w[i] = sl.w[r];
	const double *xr = sl.x + (size_t)r * (size_t)p;
	for (int j = 0; j < p; ++j) x_cols[(size_t)j * col_stride + (size_t)i] = xr[j];
}

// offs[k] = first position whose key belongs to group >= k (k = 0 .. k_n)
__global__ void rowlog_offsets_kernel(const uint64_t *keys, int64_t m, int64_t k_n, int64_t *offs, unsigned row_bits) {
	const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k > k_n) return;
	const uint64_t want = (uint64_t)k << row_bits; // (k == k_n may need bit 64: row_bits + bits(k_n) <= 63, rowlog_key_bits)
	int64_t lo = 0, hi = m;
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if (keys[mid] < want) lo = mid + 1; else hi = mid;
	}
	offs[k] = lo;
}

__global__ void rowlog_scatter_kernel(const double *src, const int32_t *sorted_slots, int64_t k_n, int len, double *dst, const int32_t *pos) {
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= k_n * len) return;
	const int64_t k = t / len;
	const int j = (int)(t - k * len);
	const int64_t row = pos ? pos[sorted_slots[k]] : sorted_slots[k];
	if (row >= 0) dst[(size_t)row * (size_t)len + (size_t)j] = src[t];
}

__global__ void rowlog_positions_kernel(const uint32_t *list, int64_t n_list, int32_t *pos) {
	const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k < n_list) pos[list[k]] = (int32_t)k;
}

__global__ void rowlog_map_queue_kernel(const int32_t *queue, const int32_t *count, const uint32_t *sel, int32_t *out) {
	const int n = *count;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = (int32_t)sel[queue[i]];
}

__global__ void rowlog_mark_kernel(const uint32_t *list, int64_t n_list, uint8_t *mark) {
	const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k < n_list) mark[list[k]] = 1;
}

__global__ void rowlog_invalidate_kernel(const uint32_t *slot, uint8_t *valid, int64_t n, const uint8_t *mark, int64_t n_slots) {
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const uint32_t s = slot[i];
		if ((int64_t)s < n_slots && mark[s]) valid[i] = 0;
	}
}

// one workgroup per list entry (grid-stride): a group the state could not refine is not handed out as a number
__global__ void rowlog_flag_unrefined_kernel(const int32_t *list, const int32_t *count, int64_t n_slots, int p, double *core, double *inf) {
	const int n = *count;
	for (int k = blockIdx.x; k < n; k += gridDim.x) {
		const int64_t g = list[k];
		if (g < 0 || g >= n_slots) continue;
		double *c = core + g * (int64_t)(p + 6);
		for (int j = threadIdx.x; j < p + 6; j += blockDim.x) c[j] = (j == p + 5) ? (double)ANOFOX_HIP_STATUS_UNREFINED : __builtin_nan("");
		if (inf) {
			double *f = inf + g * (int64_t)(5 * p + 2);
			for (int j = threadIdx.x; j < 5 * p + 2; j += blockDim.x) f[j] = __builtin_nan("");
		}
	}
}

// ---- (r4) duplication for a Combine that PRESERVES its sources (window segment trees) ----
// A logged row belongs to one slot.  When a source lives on after it has been merged into one or several targets, the
// targets need their own copies of its rows — the reference copies them too (ols_aggregate.cpp:224-233: target.x.insert(...)) —
// or Finalize could not refit the frames its moments cannot resolve (every exactly fitting frame is one).  The copies are
// appended to the log in the order of the rows they copy, labelled with the target.
struct DupMap {
	const uint32_t *usrc; // [m] distinct sources, ascending
	const int32_t *uoff;  // [m + 1] their targets: utgt[uoff[k] .. uoff[k + 1])
	const uint32_t *utgt;
	int m;
};
constexpr int kDupIt = 16; // a wavefront's tile: 64 x 16 rows

__device__ __forceinline__ int dup_find(const DupMap &d, uint32_t s) {
	int lo = 0, hi = d.m - 1;
	while (lo <= hi) {
		const int mid = (lo + hi) >> 1;
		const uint32_t v = d.usrc[mid];
		if (v == s) return mid;
		if (v < s) lo = mid + 1; else hi = mid - 1;
	}
	return -1;
}
__device__ __forceinline__ int dup_fanout(const DupMap &d, const uint32_t *slot, const uint8_t *valid, int64_t i, int64_t n, int *k_out) {
	*k_out = -1;
	if (i >= n || !valid[i]) return 0;
	const int k = dup_find(d, slot[i]);
	*k_out = k;
	return k < 0 ? 0 : d.uoff[k + 1] - d.uoff[k];
}
// copies per wavefront tile
__global__ void __launch_bounds__(kLogBlock) rowlog_dup_count_kernel(const uint32_t *slot, const uint8_t *valid, int64_t n, DupMap d, int64_t *tile_cnt) {
	const int lane = threadIdx.x & 63;
	const int64_t tile = (int64_t)blockIdx.x * (kLogBlock / 64) + (threadIdx.x >> 6);
	const int64_t r0 = tile * 64 * kDupIt;
	if (r0 >= n) return;
	int64_t mine = 0;
	for (int it = 0; it < kDupIt; ++it) {
		int k;
		mine += dup_fanout(d, slot, valid, r0 + (int64_t)it * 64 + lane, n, &k);
	}
	for (int msk = 32; msk >= 1; msk >>= 1) mine += __shfl_xor(mine, msk, 64);
	if (lane == 0) tile_cnt[tile] = mine;
}
// in-place exclusive scan by ONE workgroup (a few hundred thousand tiles at most); total -> *total
__global__ void __launch_bounds__(1024) rowlog_dup_scan_kernel(int64_t *cnt, int64_t n_tiles, int64_t *total) {
	__shared__ int64_t part[1024];
	const int t = threadIdx.x;
	const int64_t per = (n_tiles + 1023) / 1024;
	const int64_t lo = (int64_t)t * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
	int64_t sum = 0;
	for (int64_t i = lo; i < hi; ++i) sum += cnt[i];
	part[t] = sum;
	__syncthreads();
	if (t == 0) {
		int64_t run = 0;
		for (int k = 0; k < 1024; ++k) {
			const int64_t v = part[k];
			part[k] = run;
			run += v;
		}
		*total = run;
	}
	__syncthreads();
	int64_t run = part[t];
	for (int64_t i = lo; i < hi; ++i) {
		const int64_t v = cnt[i];
		cnt[i] = run;
		run += v;
	}
}
// the copies of one source slab, written behind `dst_at` of the destination arrays in row order
__global__ void __launch_bounds__(kLogBlock) rowlog_dup_fill_kernel(RowLogSlab src, int p, int weighted, DupMap d, const int64_t *tile_base,
                                                                    RowLogSlab dst, int64_t dst_at) {
	const int lane = threadIdx.x & 63;
	const int64_t tile = (int64_t)blockIdx.x * (kLogBlock / 64) + (threadIdx.x >> 6);
	const int64_t n = src.rows, r0 = tile * 64 * kDupIt;
	if (r0 >= n) return;
	int64_t at = dst_at + tile_base[tile];
	for (int it = 0; it < kDupIt; ++it) {
		const int64_t i = r0 + (int64_t)it * 64 + lane;
		int k;
		const int f = dup_fanout(d, src.slot, src.valid, i, n, &k);
		int incl = f; // inclusive scan over the wave's lanes
		for (int off = 1; off < 64; off <<= 1) {
			const int v = __shfl_up(incl, off, 64);
			if (lane >= off) incl += v;
		}
		const int total = __shfl(incl, 63, 64);
		int64_t o = at + (incl - f);
		for (int c = 0; c < f; ++c, ++o) {
			dst.slot[o] = d.utgt[d.uoff[k] + c];
			dst.valid[o] = 1;
			dst.y[o] = src.y[i];
			if (weighted) dst.w[o] = src.w[i];
			const double *xs = src.x + (size_t)i * (size_t)p;
			double *xd = dst.x + (size_t)o * (size_t)p;
			for (int j = 0; j < p; ++j) xd[j] = xs[j];
		}
		at += total;
	}
}

inline unsigned grid_for(int64_t n, int64_t cap = 1 << 16) {
	int64_t g = (n + kLogBlock - 1) / kLogBlock;
	if (g < 1) g = 1;
	if (g > cap) g = cap;
	return (unsigned)g;
}

} // namespace

size_t rowlog_sort_temp_bytes(int64_t n) {
	const size_t a = rsort::temp_bytes<uint64_t, false>((size_t)n), b = rsort::temp_bytes<uint32_t, false>((size_t)n);
	return a > b ? a : b;
}

hipError_t launch_rowlog_sort_slots(const int32_t *in, int32_t *out, int64_t n, void *temp, size_t temp_bytes, hipStream_t st) {
	// slot numbers are < 2^31: sorted as unsigned 31-bit keys ((r4) radix_sort.h; rocPRIM until round 3)
	return rsort::sort<uint32_t, false>(reinterpret_cast<const uint32_t *>(in), reinterpret_cast<uint32_t *>(out), nullptr, (size_t)n, 31u, temp, temp_bytes, st);
}

hipError_t launch_rowlog_iota(int32_t *v, int64_t n, hipStream_t st) {
	if (n <= 0) return hipSuccess;
	rowlog_iota_kernel<<<grid_for(n, 1 << 30), kLogBlock, 0, st>>>(v, n);
	return hipGetLastError();
}

hipError_t launch_rowlog_dense(const int32_t *sorted_slots, int64_t k_n, int32_t *dense, int64_t n_slots, hipStream_t st) {
	hipError_t rc = hipMemsetAsync(dense, 0xff, (size_t)n_slots * sizeof(int32_t), st);
	if (rc != hipSuccess) return rc;
	rowlog_dense_kernel<<<grid_for(k_n, 1 << 30), kLogBlock, 0, st>>>(sorted_slots, k_n, dense);
	return hipGetLastError();
}

hipError_t launch_rowlog_select(bool fill, const uint32_t *slot, const uint8_t *valid, int64_t n, int64_t base_row, const int32_t *dense,
                                int64_t n_slots, unsigned long long *counter, uint64_t *keys, unsigned row_bits, hipStream_t st) {
	if (n <= 0) return hipSuccess;
	const unsigned tiles = (unsigned)((n + kSelectTile - 1) / kSelectTile); // n <= 2^24 rows per slab
	if (fill)
		rowlog_select_kernel<true><<<tiles, kLogBlock, 0, st>>>(slot, valid, n, base_row, dense, n_slots, counter, keys, row_bits);
	else
		rowlog_select_kernel<false><<<tiles, kLogBlock, 0, st>>>(slot, valid, n, base_row, dense, n_slots, counter, keys, row_bits);
	return hipGetLastError();
}

// bits of a refit key: row_bits for the global row number, then the dense group index.  false when log_rows rows and
// k_n groups do not fit 63 bits together (2^31 slots x 2^33 rows do not; the caller reports it instead of wrapping).
bool rowlog_key_bits(int64_t log_rows, int64_t k_n, unsigned *row_bits, unsigned *end_bit) {
	unsigned rb = 1;
	while (rb < 62 && ((uint64_t)log_rows >> rb) != 0) ++rb;
	unsigned kb = 1;
	while (kb < 62 && ((uint64_t)k_n >> kb) != 0) ++kb; // k_n itself must be representable (rowlog_offsets_kernel)
	*row_bits = rb;
	*end_bit = rb + kb;
	return rb + kb <= 63;
}

hipError_t launch_rowlog_sort_keys(const uint64_t *in, uint64_t *out, int64_t m, unsigned end_bit, void *temp, size_t temp_bytes, hipStream_t st) {
	return rsort::sort<uint64_t, false>(in, out, nullptr, (size_t)m, end_bit, temp, temp_bytes, st);
}

hipError_t launch_rowlog_gather(const uint64_t *keys, int64_t m, int64_t k_n, const RowLogSlab *d_slabs, int n_slabs, int p, int weighted,
                                double *y, double *x_cols, size_t col_stride, double *w, int64_t *offs, unsigned row_bits, hipStream_t st) {
	if (m > 0) rowlog_gather_kernel<<<grid_for(m, 1 << 30), kLogBlock, 0, st>>>(keys, m, d_slabs, n_slabs, p, weighted, y, x_cols, col_stride, w, row_bits);
	rowlog_offsets_kernel<<<grid_for(k_n + 1, 1 << 30), kLogBlock, 0, st>>>(keys, m, k_n, offs, row_bits);
	return hipGetLastError();
}

hipError_t launch_rowlog_scatter(const double *src, const int32_t *sorted_slots, int64_t k_n, int len, double *dst, const int32_t *pos,
                                 hipStream_t st) {
	if (k_n <= 0) return hipSuccess;
	rowlog_scatter_kernel<<<grid_for(k_n * len, 1 << 30), kLogBlock, 0, st>>>(src, sorted_slots, k_n, len, dst, pos);
	return hipGetLastError();
}

hipError_t launch_rowlog_positions(const uint32_t *list, int64_t n_list, int32_t *pos, int64_t n_slots, hipStream_t st) {
	hipError_t rc = hipMemsetAsync(pos, 0xff, (size_t)n_slots * sizeof(int32_t), st);
	if (rc != hipSuccess) return rc;
	if (n_list > 0) rowlog_positions_kernel<<<grid_for(n_list, 1 << 30), kLogBlock, 0, st>>>(list, n_list, pos);
	return hipGetLastError();
}

hipError_t launch_rowlog_map_queue(const int32_t *queue, const int32_t *count, const uint32_t *sel, int32_t *out, hipStream_t st) {
	rowlog_map_queue_kernel<<<256, kLogBlock, 0, st>>>(queue, count, sel, out);
	return hipGetLastError();
}

hipError_t launch_rowlog_invalidate(uint8_t *mark, int64_t n_slots, const uint32_t *list, int64_t n_list, const RowLogSlab *h_slabs, int n_slabs,
                                    hipStream_t st) {
	if (n_list <= 0) return hipSuccess;
	hipError_t rc = hipMemsetAsync(mark, 0, (size_t)n_slots, st);
	if (rc != hipSuccess) return rc;
	rowlog_mark_kernel<<<grid_for(n_list, 1 << 30), kLogBlock, 0, st>>>(list, n_list, mark);
	for (int k = 0; k < n_slabs; ++k)
		if (h_slabs[k].rows > 0)
			rowlog_invalidate_kernel<<<grid_for(h_slabs[k].rows, 4096), kLogBlock, 0, st>>>(h_slabs[k].slot, h_slabs[k].valid, h_slabs[k].rows, mark, n_slots);
	return hipGetLastError();
}

hipError_t launch_rowlog_flag_unrefined(const int32_t *list, const int32_t *count, int64_t n_slots, int p, double *core, double *inf,
                                        hipStream_t st) {
	rowlog_flag_unrefined_kernel<<<1024, 64, 0, st>>>(list, count, n_slots, p, core, inf);
	return hipGetLastError();
}

// tile_cnt: one int64 per wavefront tile of every slab, slab after slab (rowlog_dup_tiles), then one more for the total
int64_t rowlog_dup_tiles(int64_t rows) { return (rows + 64 * kDupIt - 1) / (64 * kDupIt); }

hipError_t launch_rowlog_dup_count(const RowLogSlab *h_slabs, int n_slabs, const uint32_t *usrc, const int32_t *uoff, const uint32_t *utgt, int m,
                                   int64_t *tile_cnt, int64_t n_tiles, hipStream_t st) {
	const DupMap d{usrc, uoff, utgt, m};
	int64_t t0 = 0;
	for (int k = 0; k < n_slabs; ++k) {
		const int64_t tiles = rowlog_dup_tiles(h_slabs[k].rows);
		if (tiles > 0)
			rowlog_dup_count_kernel<<<(unsigned)((tiles + kLogBlock / 64 - 1) / (kLogBlock / 64)), kLogBlock, 0, st>>>(h_slabs[k].slot, h_slabs[k].valid,
			                                                                                                      h_slabs[k].rows, d, tile_cnt + t0);
		t0 += tiles;
	}
	rowlog_dup_scan_kernel<<<1, 1024, 0, st>>>(tile_cnt, n_tiles, tile_cnt + n_tiles);
	return hipGetLastError();
}

// src_rows[k]: the rows slab k held when the copies were counted (the destination may be the last slab itself)
hipError_t launch_rowlog_dup_fill(const RowLogSlab *h_slabs, const int64_t *src_rows, int n_slabs, int p, int weighted, const uint32_t *usrc,
                                  const int32_t *uoff, const uint32_t *utgt, int m, const int64_t *tile_base, const RowLogSlab &dst, int64_t dst_at,
                                  hipStream_t st) {
	const DupMap d{usrc, uoff, utgt, m};
	int64_t t0 = 0;
	for (int k = 0; k < n_slabs; ++k) {
		RowLogSlab src = h_slabs[k];
		src.rows = src_rows[k];
		const int64_t tiles = rowlog_dup_tiles(src.rows);
		if (tiles > 0)
			rowlog_dup_fill_kernel<<<(unsigned)((tiles + kLogBlock / 64 - 1) / (kLogBlock / 64)), kLogBlock, 0, st>>>(src, p, weighted, d, tile_base + t0, dst,
			                                                                                                     dst_at);
		t0 += tiles;
	}
	return hipGetLastError();
}

hipError_t launch_rowlog_remap(uint32_t *remap, int64_t n_slots, const uint32_t *src, const uint32_t *dst, int64_t n_pairs,
                               const RowLogSlab *h_slabs, int n_slabs, hipStream_t st) {
	rowlog_remap_init_kernel<<<grid_for(n_slots, 1 << 30), kLogBlock, 0, st>>>(remap, n_slots);
	rowlog_remap_pairs_kernel<<<grid_for(n_pairs, 1 << 30), kLogBlock, 0, st>>>(remap, src, dst, n_pairs);
	for (int k = 0; k < n_slabs; ++k)
		if (h_slabs[k].rows > 0)
			rowlog_remap_kernel<<<grid_for(h_slabs[k].rows, 4096), kLogBlock, 0, st>>>(h_slabs[k].slot, h_slabs[k].rows, remap, n_slots);
	return hipGetLastError();
}

} // namespace anofox
