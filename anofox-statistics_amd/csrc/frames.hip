// frames.hip — window functions over ANY frame and any width as a batch of "virtual groups".
//
// The reference's window functions (src/window_functions/ols_fit_predict.cpp:110-324; ridge_fit_predict.cpp,
// wls_fit_predict.cpp) run the aggregate's Update / Finalize over every frame DuckDB hands them — ROWS, RANGE or
// GROUPS, any number of features.  window_narrow.hip covers ROWS frames for p <= 8 with in-register solves; this
// file covers everything else, and the frames window_narrow.hip flags as ill-conditioned: frame e is the row range
// [lo[e], hi[e]) and becomes group e of an ordinary batch fit whose row ranges may overlap (BatchArgs::row_ends),
// so the three accumulate / solve families (and their refinement passes) apply unchanged.  What is specific:
//   ynn     prefix count of rows whose y is not NULL (NaN): the training rows of a frame in O(1);
//   rule    the window's NULL rule: a frame needs MORE than p + [intercept] training rows
//           (ols_fit_predict.cpp:257-262) — frames that fail get rule count 0, which the solve kernels turn into
//           status 100 like the aggregate's "fewer than 2 rows";
//   spec    frame bounds of ROWS BETWEEN a PRECEDING AND b PRECEDING per row, clipped to the partition;
//   predict x of the LAST row of the frame (ols_fit_predict.cpp:157-162) with the simplified interval of
//           anofox_predict_with_interval (lib.rs:2264-2349): yhat -+ t sigma sqrt(1 + 1/n).
// Cost: O(frame) rows per output row (the reference refits every frame from scratch as well).
#include "common.h"

namespace anofox {

namespace {

__global__ __launch_bounds__(256) void frames_spec_kernel(const int64_t *row_offsets, int64_t n_groups, int64_t n_rows, int64_t start_p,
                                                          int64_t end_p, int64_t *lo_out, int64_t *hi_out, const int32_t *list,
                                                          int64_t n_list) {
	const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= (list ? n_list : n_rows)) return;
	const int64_t e = list ? (int64_t)list[k] : k;
	if (e < 0 || e >= n_rows) { // (a list entry out of range: an empty frame)
		lo_out[k] = hi_out[k] = 0;
		return;
	}
	// partition of row e: the last g with row_offsets[g] <= e
	int64_t a = 0, b = n_groups;
	while (b - a > 1) {
		const int64_t m = (a + b) >> 1;
		if (row_offsets[m] <= e) a = m;
		else b = m;
	}
	const int64_t plo = row_offsets[a], phi = row_offsets[a + 1];
	int64_t first = start_p == kFrameUnbounded ? plo : e - start_p;
	int64_t last = end_p == -kFrameUnbounded ? phi - 1 : e - end_p;
	if (first < plo) first = plo;
	if (last > phi - 1) last = phi - 1;
	const bool empty = last < first || e < plo || e >= phi;
	lo_out[k] = empty ? e : first;
	hi_out[k] = empty ? e : last + 1;
}

__global__ __launch_bounds__(256) void frames_rule_kernel(FrameArgs a) {
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= a.n_frames) return;
	const int64_t lo = a.lo[e], hi = a.hi[e];
	int64_t nt = 0;
	if (hi > lo) {
		if (a.ynn) nt = a.ynn[hi] - a.ynn[lo];
		else // few frames (the refits of flagged rows): count directly instead of scanning every row of the input
			for (int64_t r = lo; r < hi; ++r) nt += a.y[r] == a.y[r] ? 1 : 0;
	}
	a.rule_counts[e] = nt > (int64_t)(a.p + (a.fit_intercept ? 1 : 0)) ? nt : 0;
}

// critical value for an integer df: table lookup; beyond the table the Cornish-Fisher series (as window_narrow.hip)
__device__ __forceinline__ double frames_tcrit(const FrameArgs &a, double df) {
	const int i = (int)df;
	if (i <= a.tcrit_cap) return a.tcrit[i];
	const double z = a.tcrit[0], z2 = z * z, r = 1.0 / df;
	const double g1 = z * (z2 + 1.0) * 0.25;
	const double g2 = z * ((5.0 * z2 + 16.0) * z2 + 3.0) * (1.0 / 96.0);
	const double g3 = z * (((3.0 * z2 + 19.0) * z2 + 17.0) * z2 - 15.0) * (1.0 / 384.0);
	return z + r * (g1 + r * (g2 + r * g3));
}

__global__ __launch_bounds__(256) void frames_predict_kernel(FrameArgs a) {
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= a.n_frames) return;
	const int p = a.p;
	const double *c = a.core + e * (int64_t)(p + 6);
	const double nanv = __builtin_nan("");
	double yhat = nanv, ylo = nanv, yhi = nanv;
	const int64_t hi = a.hi[e];
	if (c[p + 5] == 0.0 && hi > a.lo[e]) {
		const int64_t r = hi - 1; // the frame's last row
		const double b0 = c[p];
		double v = isnan(b0) ? 0.0 : b0; // lib.rs:2264-2349: NaN intercept / coefficients contribute 0
		for (int j = 0; j < p; ++j) {
			const double bj = c[j];
			if (!isnan(bj)) v = fma(bj, a.x_table[j][r], v);
		}
		if (isfinite(v)) {
			yhat = ylo = yhi = v;
			const double rse = c[p + 3], n = c[p + 4];
			const double used = (double)p + (isnan(b0) ? 0.0 : 1.0);
			if (!isnan(rse) && rse > 0.0 && n > (double)p + 1.0 && n > used) {
				const double t = frames_tcrit(a, n - used);
				if (!isnan(t)) {
					const double margin = t * rse * sqrt(1.0 + 1.0 / n);
					ylo = v - margin;
					yhi = v + margin;
				}
			}
		}
	}
	double *out = a.pred + (a.list ? (int64_t)a.list[e] : e) * 3;
	out[0] = yhat;
	out[1] = ylo;
	out[2] = yhi;
}

} // namespace

// ---- ynn[i] = number of rows r < i with y[r] not NaN, i = 0 .. n_rows ((r4) hand-written; rocPRIM's exclusive_scan until round 3) ----
// Three launches: a workgroup counts the non-NaN rows of its tile (ballots), one workgroup scans the <= kScanMaxBlocks tile
// counts, the workgroups walk their tiles again 256 rows at a time and write the running counts.
constexpr unsigned kScanMaxBlocks = 2048;
struct ScanPlan {
	unsigned n_blocks;
	int64_t tile; // rows per workgroup, a multiple of 256
};
static ScanPlan scan_plan(int64_t n) {
	int64_t chunks = (n + 255) / 256, per = (chunks + kScanMaxBlocks - 1) / kScanMaxBlocks;
	if (per < 1) per = 1;
	ScanPlan p;
	p.tile = per * 256;
	p.n_blocks = (unsigned)((n + p.tile - 1) / p.tile);
	if (p.n_blocks == 0) p.n_blocks = 1;
	return p;
}

__global__ __launch_bounds__(256) void frames_ynn_count_kernel(const double *y, int64_t n_rows, int64_t tile, int64_t *counts) {
	__shared__ int64_t wsum[4];
	const int64_t lo = (int64_t)blockIdx.x * tile, hi = lo + tile < n_rows ? lo + tile : n_rows;
	int64_t c = 0;
	for (int64_t i0 = lo; i0 < hi; i0 += 256) {
		const int64_t i = i0 + threadIdx.x;
		c += (int64_t)__popcll(__ballot(i < hi && y[i] == y[i])); // (the same count in every lane of the wavefront)
	}
	if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void frames_ynn_scan_kernel(int64_t *counts, unsigned n_blocks, int64_t *total_out) {
	__shared__ int64_t part[1024];
	const unsigned span = (n_blocks + 1023u) / 1024u;
	const unsigned lo = threadIdx.x * span < n_blocks ? threadIdx.x * span : n_blocks, hi = lo + span < n_blocks ? lo + span : n_blocks;
	int64_t s = 0;
	for (unsigned i = lo; i < hi; ++i) s += counts[i];
	part[threadIdx.x] = s;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) {
		const int64_t v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	int64_t base = threadIdx.x ? part[threadIdx.x - 1] : 0;
	for (unsigned i = lo; i < hi; ++i) {
		const int64_t c = counts[i];
		counts[i] = base;
		base += c;
	}
	if (threadIdx.x == 1023) *total_out = part[1023];
}

__global__ __launch_bounds__(256) void frames_ynn_write_kernel(const double *y, int64_t n_rows, int64_t tile, const int64_t *counts, int64_t *ynn) {
	__shared__ int wcount[4];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t lo = (int64_t)blockIdx.x * tile, hi = lo + tile < n_rows ? lo + tile : n_rows;
	int64_t base = counts[blockIdx.x];
	for (int64_t i0 = lo; i0 < hi; i0 += 256) {
		const int64_t i = i0 + threadIdx.x;
		const unsigned long long b = __ballot(i < hi && y[i] == y[i]);
		if (lane == 0) wcount[w] = (int)__popcll(b);
		__syncthreads();
		int before = 0, all = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			before += k < w ? wcount[k] : 0;
			all += wcount[k];
		}
		if (i < hi) ynn[i] = base + before + (int64_t)__popcll(b & ((1ull << lane) - 1ull));
		base += all;
		__syncthreads();
	}
}

size_t frames_scan_temp_bytes(int64_t) { return (size_t)kScanMaxBlocks * sizeof(int64_t); }

hipError_t launch_frames_ynn(const double *y, int64_t n_rows, int64_t *ynn, void *temp, size_t temp_bytes, hipStream_t stream) {
	if (n_rows < 0 || temp_bytes < frames_scan_temp_bytes(n_rows)) return hipErrorInvalidValue;
	if (n_rows == 0) return hipMemsetAsync(ynn, 0, sizeof(int64_t), stream);
	const ScanPlan pl = scan_plan(n_rows);
	int64_t *counts = static_cast<int64_t *>(temp);
	hipLaunchKernelGGL(frames_ynn_count_kernel, dim3(pl.n_blocks), dim3(256), 0, stream, y, n_rows, pl.tile, counts);
	hipLaunchKernelGGL(frames_ynn_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, pl.n_blocks, ynn + n_rows);
	hipLaunchKernelGGL(frames_ynn_write_kernel, dim3(pl.n_blocks), dim3(256), 0, stream, y, n_rows, pl.tile, counts, ynn);
	return hipGetLastError();
}

hipError_t launch_frames_from_rows_spec(const int64_t *row_offsets, int64_t n_groups, int64_t n_rows, int64_t start_preceding,
                                        int64_t end_preceding, int64_t *lo, int64_t *hi, hipStream_t stream, const int32_t *list,
                                        int64_t n_list) {
	const int64_t n = list ? n_list : n_rows;
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(frames_spec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, row_offsets, n_groups, n_rows,
	                   start_preceding, end_preceding, lo, hi, list, n_list);
	return hipGetLastError();
}

hipError_t launch_frames_rule(const FrameArgs &a, hipStream_t stream) {
	if (a.n_frames <= 0) return hipSuccess;
	hipLaunchKernelGGL(frames_rule_kernel, dim3((unsigned)((a.n_frames + 255) / 256)), dim3(256), 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_frames_predict(const FrameArgs &a, hipStream_t stream) {
	if (a.n_frames <= 0) return hipSuccess;
	hipLaunchKernelGGL(frames_predict_kernel, dim3((unsigned)((a.n_frames + 255) / 256)), dim3(256), 0, stream, a);
	return hipGetLastError();
}

} // namespace anofox
