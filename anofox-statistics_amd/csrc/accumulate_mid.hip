// accumulate_mid.hip — moment accumulation for moderately wide designs (8 < p <= 32): one WAVEFRONT per group.
//
// Same role and the same record as accumulate_wide.hip (tile-major X'WX blocks from v_mfma_f64_16x16x4_f64, X'Wy,
// column sums, y moments, first valid row, constant-column flags; reference: the row buffering + dense
// decomposition of src/aggregate_functions/ols_aggregate.cpp:120-186,249-296 and the row filter / constant test
// of crates/anofox-stats-core/src/models/ols.rs:59-87).  The workgroup-per-group kernel stages 16-row chunks
// through LDS behind a barrier per chunk; with one or two column blocks that overhead, not HBM or the matrix
// pipe, set the pace (1.7-2.8 TB/s at p = 9..32).  Here a wave owns its group and nothing is shared:
//   * a step is 16 rows; lane (kk, lj) loads rows 4 kk .. 4 kk + 3 of column 16 I + lj straight into MFMA
//     fragment layout (two 16-byte loads per column block, the 16 lanes of a column cover one 128-byte line),
//     one step ahead of its use;
//   * K-step m of the MFMA takes the lanes' m-th row: A[i = lj][k = kk] = w d[4 kk + m][16 I + lj],
//     B[k = kk][j = lj] = d[4 kk + m][16 J + lj]  (which rows share a K-step is irrelevant to the sum);
//   * row validity = 4 ballots per step (the 16 lanes of a kk group hold the 16 columns of a block), invalid
//     rows are removed with bit masks, so a step is branch-free;
//   * four groups per 256-thread workgroup, no LDS, no barrier.
#include "common.h"

namespace anofox {

typedef double mid_dbl2u __attribute__((ext_vector_type(2), aligned(8)));
typedef double mid_dbl4 __attribute__((ext_vector_type(4)));
typedef const double __attribute__((address_space(1))) *mid_gptr_t;
typedef const mid_dbl2u __attribute__((address_space(1))) *mid_gptr2_t;

namespace {

__device__ __forceinline__ double mid_mask(double v, long long m) {
	return __longlong_as_double(__double_as_longlong(v) & m);
}

// four consecutive rows r .. r + 3 of one column; rows at or past `hi` are clamped (their values are masked later)
__device__ __forceinline__ void load4(mid_gptr_t col, int64_t r, int64_t hi, bool full, double (&v)[4]) {
	if (full) {
		const mid_dbl2u a = *reinterpret_cast<mid_gptr2_t>(col + r);
		const mid_dbl2u b = *reinterpret_cast<mid_gptr2_t>(col + r + 2);
		v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
	} else {
#pragma unroll
		for (int m = 0; m < 4; ++m) v[m] = col[r + m < hi ? r + m : hi - 1];
	}
}

template <int T, bool WEIGHTED, bool CENTER>
__global__ __launch_bounds__(256) void accumulate_mid_kernel(WideArgs args) {
	constexpr int P16 = 16 * T;
	constexpr int NT = T * (T + 1) / 2;
	const int p = args.p;
	const int lane = threadIdx.x & 63;
	const int kk = lane >> 4, lj = lane & 15;
	const int64_t gl = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	if (gl >= args.n_groups) return;
	const int64_t lo = args.row_offsets[args.group_base + gl];
	const int64_t hi = args.row_offsets[args.group_base + gl + 1];

	mid_gptr_t col[T];
	bool real[T]; // column 16 I + lj exists (padding columns read column p - 1 and count as zeros)
#pragma unroll
	for (int I = 0; I < T; ++I) {
		const int j = 16 * I + lj;
		real[I] = j < p;
		col[I] = (mid_gptr_t)(uintptr_t)args.x_table[real[I] ? j : p - 1];
	}
	const mid_gptr_t ycol = (mid_gptr_t)(uintptr_t)args.y;
	const mid_gptr_t wcol = (mid_gptr_t)(uintptr_t)args.w;

	mid_dbl4 acc[NT];
#pragma unroll
	for (int t = 0; t < NT; ++t) acc[t] = mid_dbl4{0.0, 0.0, 0.0, 0.0};
	double sx[T], sxy[T], first[T];
#pragma unroll
	for (int I = 0; I < T; ++I) sx[I] = sxy[I] = first[I] = 0.0;
	unsigned ncmask = 0;
	double sy = 0.0, syy = 0.0, sw = 0.0, first_y = 0.0;
	bool have_first = false;
	int cnt = 0;

	double xn[T][4], yn[4], wn[4]; // the next step, in flight
	auto issue = [&](int64_t r0) {
		const bool full = r0 + 16 <= hi; // wave-uniform
		const int64_t r = r0 + 4 * kk;
#pragma unroll
		for (int I = 0; I < T; ++I) load4(col[I], r, hi, full, xn[I]);
		load4(ycol, r, hi, full, yn);
		if (WEIGHTED) load4(wcol, r, hi, full, wn);
	};
	if (lo < hi) issue(lo);
	for (int64_t r0 = lo; r0 < hi; r0 += 16) {
		double x[T][4], y[4], w[4];
#pragma unroll
		for (int m = 0; m < 4; ++m) {
#pragma unroll
			for (int I = 0; I < T; ++I) x[I][m] = real[I] ? xn[I][m] : 0.0;
			y[m] = yn[m];
			w[m] = WEIGHTED ? wn[m] : 1.0;
		}
		if (r0 + 16 < hi) issue(r0 + 16);

		// row validity (ols.rs:59-66, wls.rs:76-86): bit 4 kk + m of rowmask
		unsigned rowmask = 0;
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			bool ok = isfinite(y[m]) && (r0 + 4 * kk + m < hi);
			if (WEIGHTED) ok = ok && isfinite(w[m]) && (w[m] > 0.0);
#pragma unroll
			for (int I = 0; I < T; ++I) ok = ok && isfinite(x[I][m]);
			const unsigned long long b = __ballot(ok);
#pragma unroll
			for (int k = 0; k < 4; ++k) rowmask |= (((b >> (16 * k)) & 0xFFFFull) == 0xFFFFull) ? (1u << (4 * k + m)) : 0u;
		}
		rowmask = __builtin_amdgcn_readfirstlane(rowmask);
		if (rowmask == 0u) continue;
		if (!have_first) {
			const int r = __ffs((int)rowmask) - 1; // first valid row of the group: held by the lanes of kk = r / 4
			const int src = 16 * (r >> 2) + lj, m = r & 3;
#pragma unroll
			for (int I = 0; I < T; ++I) {
				const double mine = m == 0 ? x[I][0] : (m == 1 ? x[I][1] : (m == 2 ? x[I][2] : x[I][3]));
				first[I] = __shfl(mine, src, 64);
			}
			const double ym = m == 0 ? y[0] : (m == 1 ? y[1] : (m == 2 ? y[2] : y[3]));
			first_y = __shfl(ym, src, 64);
			have_first = true;
		}
		cnt += __popc(rowmask);
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			const long long rm = -(long long)((rowmask >> (4 * kk + m)) & 1u); // all ones when the row is valid
			double d[T], a[T];
#pragma unroll
			for (int I = 0; I < T; ++I) {
				const double dev = mid_mask(x[I][m] - first[I], rm); // deviation from the first valid row
				d[I] = CENTER ? dev : mid_mask(x[I][m], rm);
				// constant-column predicate of ols.rs:76-87: |x - x_first| < 1e-10 on every valid row
				ncmask |= !(fabs(dev) < 1e-10) ? (1u << I) : 0u;
			}
			const double dy = mid_mask(CENTER ? y[m] - first_y : y[m], rm);
			const double wv = mid_mask(w[m], rm);
#pragma unroll
			for (int I = 0; I < T; ++I) a[I] = WEIGHTED ? wv * d[I] : d[I];
			int tile = 0;
#pragma unroll
			for (int I = 0; I < T; ++I) {
#pragma unroll
				for (int J = I; J < T; ++J) {
					acc[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], d[J], acc[tile], 0, 0, 0);
					++tile;
				}
			}
#pragma unroll
			for (int I = 0; I < T; ++I) {
				sx[I] += a[I];
				sxy[I] = fma(a[I], dy, sxy[I]);
			}
			const double wdy = WEIGHTED ? wv * dy : dy;
			sy += wdy;
			syy = fma(wdy, dy, syy);
			sw += wv;
		}
	}

	// ---- the moment record, layout of accumulate_wide.hip ----
	double *rec = args.moments + gl * (int64_t)wide_record_len(T);
#pragma unroll
	for (int t = 0; t < NT; ++t) {
		double *tp = rec + (int64_t)t * 256; // tile-major, element (row, col) at row * 16 + col
#pragma unroll
		for (int r = 0; r < 4; ++r) tp[(kk + 4 * r) * 16 + lj] = acc[t][r];
	}
	double *vec = rec + (int64_t)NT * 256;
#pragma unroll
	for (int I = 0; I < T; ++I) { // reduce over the four kk groups (lanes l, l^16, l^32, l^48)
		double a = sx[I], b = sxy[I];
		a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
		b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
		unsigned nc = (ncmask >> I) & 1u;
		nc |= (unsigned)__shfl_xor((int)nc, 16, 64);
		nc |= (unsigned)__shfl_xor((int)nc, 32, 64);
		if (lane < 16) {
			vec[0 * P16 + 16 * I + lane] = a;
			vec[1 * P16 + 16 * I + lane] = b;
			vec[2 * P16 + 16 * I + lane] = first[I];
			vec[3 * P16 + 16 * I + lane] = (real[I] && nc) ? 1.0 : 0.0;
		}
	}
	// every lane of a kk group holds the same partial of the y moments: lanes 0, 16, 32, 48
	sy += __shfl_xor(sy, 16, 64); sy += __shfl_xor(sy, 32, 64);
	syy += __shfl_xor(syy, 16, 64); syy += __shfl_xor(syy, 32, 64);
	sw += __shfl_xor(sw, 16, 64); sw += __shfl_xor(sw, 32, 64);
	if (lane == 0) {
		double *sc = vec + 4 * P16;
		sc[0] = sy;
		sc[1] = syy;
		sc[2] = sw;
		sc[3] = (double)cnt;
		sc[4] = first_y;
	}
}

template <int T>
hipError_t launch_mid_T(const WideArgs &a, hipStream_t stream) {
	const bool weighted = a.model == ANOFOX_HIP_MODEL_WLS;
	const bool center = a.fit_intercept != 0;
	const dim3 grid((unsigned)((a.n_groups + 3) / 4)), block(256);
	if (weighted) {
		if (center) hipLaunchKernelGGL((accumulate_mid_kernel<T, true, true>), grid, block, 0, stream, a);
		else hipLaunchKernelGGL((accumulate_mid_kernel<T, true, false>), grid, block, 0, stream, a);
	} else {
		if (center) hipLaunchKernelGGL((accumulate_mid_kernel<T, false, true>), grid, block, 0, stream, a);
		else hipLaunchKernelGGL((accumulate_mid_kernel<T, false, false>), grid, block, 0, stream, a);
	}
	return hipGetLastError();
}

} // namespace

bool accumulate_mid_supports(int p) { return p > kNarrowMaxP && p <= 32; }

hipError_t launch_accumulate_mid(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	switch (wide_tiles(a.p)) {
	case 1: return launch_mid_T<1>(a, stream);
	case 2: return launch_mid_T<2>(a, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace anofox
