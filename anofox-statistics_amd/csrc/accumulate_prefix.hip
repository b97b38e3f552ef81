// accumulate_prefix.hip — (r4) moment records of EXPANDING window frames, 9 <= p <= 128.
//
// The `*_fit_predict` window functions over ROWS BETWEEN UNBOUNDED PRECEDING AND b PRECEDING | CURRENT ROW | b FOLLOWING
// (src/window_functions/ols_fit_predict.cpp:110-324: one refit from scratch per output row) run here, for designs wider than the
// in-register window kernels take, as "virtual groups" of the batch fit: frame e = rows [lo[e], hi[e]) is group e, and the
// accumulate kernels read every frame's rows again — O(partition length) rows per output row (frames.hip).  For frames that all
// start at their partition's first row that is n / 2 rows per row for nothing: frame e + 1 is frame e plus one row.
//
// This kernel writes the SAME moment records (the layout of accumulate_wide / accumulate_mid / accumulate_quad: upper-triangular
// 16 x 16 tiles of sum w d d', then sum w d, sum w d dy, the first valid row, the non-constant flags, the y sums and counts) for
// K consecutive frames per workgroup: the record of the first frame is accumulated from the frame's rows, every following frame
// adds the rows it has more than its predecessor and the record is written out again.  Rows read per frame: n / (2 K) + 1
// instead of n / 2.  Everything behind the records — the primary solve, the refinement passes (which re-read a queued frame's
// rows through row_offsets / row_ends exactly as before), the window's NULL rule, the prediction — is unchanged, so the
// frames keep the fit path's accuracy.  Any frame list is handled (a frame that does not extend its predecessor restarts the
// sums); the host uses the kernel where frames are prefix-shaped (host_api.hip: run_window).
//
// One 256-thread workgroup: the record lives in LDS (74 KB at p = 128); a row is added by all threads — thread j loads column j,
// the row filter (ols.rs:59-66, wls.rs:76-86) is a workgroup vote, thread t owns the record's elements t, t + 256, .. — and
// the record is copied out coalesced.  Vector units, not matrix cores: a rank-1 update per row is all there is.
#include "common.h"

namespace anofox {

namespace {

constexpr int kPrefixThreads = 256;

template <bool WEIGHTED, bool CENTER>
__global__ __launch_bounds__(kPrefixThreads) void accumulate_prefix_kernel(WideArgs args, int frames_per_block) {
	extern __shared__ double pf_lds[];
	const int p = args.p;
	const int T = wide_tiles(p), P16 = 16 * T, NT = T * (T + 1) / 2;
	const int reclen = wide_record_len(T);
	const int tid = threadIdx.x;
	double *rec = pf_lds;                 // [reclen]
	double *z = rec + reclen;             // [P16 + 2]: d_0 .. d_{P16 - 1} (0 beyond p), dy, w
	double *first = z + P16 + 2;          // [P16 + 1]: x of the first valid row, y at P16
	int *vote = reinterpret_cast<int *>(first + P16 + 2);
	double *vec = rec + NT * 256, *sc = vec + 4 * P16;

	const int64_t g_begin = (int64_t)blockIdx.x * frames_per_block;
	int64_t g_end = g_begin + frames_per_block;
	if (g_end > args.n_groups) g_end = args.n_groups;
	// this thread's column (x_tid, y, w), read once: the table is indexed by the thread number only here
	const double *col = tid < p ? args.x_table[tid] : (tid == p ? args.y : (WEIGHTED && tid == p + 1 ? args.w : nullptr));
	int64_t cur_lo = -1, cur_hi = -1; // rows [cur_lo, cur_hi) are in the record
	double v_next = 0.0;              // this thread's value of row `next_row`, loaded ahead
	int64_t next_row = -1;
	bool have_next = false;
	bool have_first = false;          // (uniform)

	auto reset = [&]() {
		__syncthreads(); // (the copy of the frame before this one reads `rec`)
		for (int k = tid; k < reclen; k += kPrefixThreads) rec[k] = 0.0;
		for (int k = tid; k <= P16; k += kPrefixThreads) first[k] = 0.0;
		have_first = false;
		__syncthreads();
	};
	// this thread's elements of the tiles: element k of the record, k = tile * 256 + r * 16 + c -> (16 I + r, 16 J + c)
	// `v` = this thread's value of row r (loaded one row ahead by the caller: a row per memory round trip would be the kernel)
	auto add_row = [&](double v) {
		// thread j < p column j, thread p: y, thread p + 1: w; everything finite (and w > 0) or the row does not take part
		bool bad = false;
		if (tid <= p + (WEIGHTED ? 1 : 0)) bad = !isfinite(v) || (WEIGHTED && tid == p + 1 && !(v > 0.0));
		if (tid == 0) *vote = 0;
		__syncthreads();
		if (bad) *vote = 1; // (benign race: every writer writes 1)
		__syncthreads();
		if (*vote) {
			__syncthreads(); // (the vote word is rewritten by the next row)
			return;
		}
		if (!have_first) { // the group's first valid row: the shift (with an intercept), the reference of the constant-column test
			if (tid < p) first[tid] = v;
			else if (tid == p) first[P16] = v;
			have_first = true;
		}
		if (tid < p) {
			const double f = first[tid];
			z[tid] = CENTER ? v - f : v;
			if (!(fabs(v - f) < 1e-10)) vec[3 * P16 + tid] = 1.0; // not constant (ols.rs:76-87)
		} else if (tid == p) {
			z[P16] = CENTER ? v - first[P16] : v;
		} else if (tid == p + 1) {
			z[P16 + 1] = WEIGHTED ? v : 1.0;
		}
		if (!WEIGHTED && tid == p + 1) z[P16 + 1] = 1.0;
		__syncthreads();
		const double w = z[P16 + 1], dy = z[P16];
		int tile = 0;
		for (int I = 0; I < T; ++I) {
			for (int J = I; J < T; ++J, ++tile) {
				const int rr = tid >> 4, cc = tid & 15;
				const double a = z[16 * I + rr], b = z[16 * J + cc];
				rec[tile * 256 + tid] = fma(w * a, b, rec[tile * 256 + tid]);
			}
		}
		if (tid < P16) {
			const double wd = w * z[tid];
			vec[tid] += wd;                                 // sum w d
			vec[P16 + tid] = fma(wd, dy, vec[P16 + tid]);   // sum w d dy
		}
		if (tid == 0) {
			const double wdy = w * dy;
			sc[0] += wdy;
			sc[1] = fma(wdy, dy, sc[1]);
			sc[2] += w;
			sc[3] += 1.0;
		}
		__syncthreads();
	};

	for (int k = tid; k < P16 + 2; k += kPrefixThreads) z[k] = 0.0; // (columns p .. 16 T - 1 stay zero)
	__syncthreads();
	for (int64_t g = g_begin; g < g_end; ++g) {
		const int64_t lo = args.row_offsets[args.group_base + g];
		const int64_t hi = group_row_end(args, args.group_base + g);
		if (lo != cur_lo || hi < cur_hi || cur_lo < 0) { // not an extension of the frame before it
			reset();
			cur_lo = lo;
			cur_hi = lo;
		}
		// rows this block will still add after this frame's (the frames that extend it): the prefetch may run that far
		if (hi > cur_hi) {
			int64_t run_end = hi;
			for (int64_t g2 = g + 1; g2 < g_end && g2 < g + 4; ++g2) {
				if (args.row_offsets[args.group_base + g2] != lo) break;
				const int64_t h2 = group_row_end(args, args.group_base + g2);
				if (h2 < run_end) break;
				run_end = h2;
			}
			if (!have_next || next_row != cur_hi) {
				v_next = col ? col[cur_hi] : 0.0;
				next_row = cur_hi;
				have_next = true;
			}
			for (int64_t r = cur_hi; r < hi; ++r) {
				const double v = v_next;
				if (r + 1 < run_end) {
					v_next = col ? col[r + 1] : 0.0;
					next_row = r + 1;
				} else {
					have_next = false;
				}
				add_row(v);
			}
			cur_hi = hi;
		}
		// the record of this frame (first valid row and y's first value with it)
		double *out = args.moments + g * (int64_t)reclen;
		for (int k = tid; k < reclen; k += kPrefixThreads) {
			double v = rec[k];
			if (k >= NT * 256 + 2 * P16 && k < NT * 256 + 3 * P16) v = first[k - NT * 256 - 2 * P16];
			else if (k == NT * 256 + 4 * P16 + 4) v = first[P16];
			out[k] = v;
		}
		// (the next frame's rows are added to `rec` only after every thread has copied its part: add_row starts with a barrier)
	}
}

} // namespace

size_t accumulate_prefix_lds_bytes(int p) {
	const int T = wide_tiles(p);
	return sizeof(double) * ((size_t)wide_record_len(T) + 2 * (size_t)(16 * T + 2) + 2 + 4);
}

hipError_t launch_accumulate_prefix(const WideArgs &a, int frames_per_block, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	const bool weighted = a.model == ANOFOX_HIP_MODEL_WLS;
	const bool center = a.fit_intercept != 0;
	const size_t lds = accumulate_prefix_lds_bytes(a.p);
	static const bool attr_set = [] {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_prefix_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_prefix_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_prefix_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_prefix_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		return true;
	}();
	(void)attr_set;
	const dim3 grid((unsigned)((a.n_groups + frames_per_block - 1) / frames_per_block)), block(kPrefixThreads);
	if (weighted) {
		if (center) hipLaunchKernelGGL((accumulate_prefix_kernel<true, true>), grid, block, lds, stream, a, frames_per_block);
		else hipLaunchKernelGGL((accumulate_prefix_kernel<true, false>), grid, block, lds, stream, a, frames_per_block);
	} else {
		if (center) hipLaunchKernelGGL((accumulate_prefix_kernel<false, true>), grid, block, lds, stream, a, frames_per_block);
		else hipLaunchKernelGGL((accumulate_prefix_kernel<false, false>), grid, block, lds, stream, a, frames_per_block);
	}
	return hipGetLastError();
}

} // namespace anofox
