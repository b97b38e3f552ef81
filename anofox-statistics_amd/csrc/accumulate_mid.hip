// accumulate_mid.hip — moment accumulation for moderately wide designs (8 < p <= 32): one WAVEFRONT per group.
//
// Same role and the same record as accumulate_wide.hip (tile-major X'WX blocks from v_mfma_f64_16x16x4_f64, X'Wy,
// column sums, y moments, first valid row, constant-column flags; reference: the row buffering + dense
// decomposition of src/aggregate_functions/ols_aggregate.cpp:120-186,249-296 and the row filter / constant test
// of crates/anofox-stats-core/src/models/ols.rs:59-87).  The workgroup-per-group kernel stages 16-row chunks
// through LDS behind a barrier per chunk; with one or two column blocks that overhead, not HBM or the matrix
// pipe, set the pace (1.7-2.8 TB/s at p = 9..32).  Here a wave owns its group and nothing is shared:
//   * a step is 16 rows; lane (kk, lj) loads rows 2 kk, 2 kk + 1 and 8 + 2 kk, 9 + 2 kk of column 16 I + lj straight
//     into MFMA fragment layout (two 16-byte loads per column block; each instruction covers a contiguous 64-byte
//     half line per column);
//     a loop trip is 2 or 4 steps and the next trip's loads are in flight while this one computes;
//   * K-step m of the MFMA takes the lanes' m-th row r = mid_row(kk, m): A[i = lj][k = kk] = w d[r][16 I + lj],
//     B[k = kk][j = lj] = d[r][16 J + lj]  (which rows share a K-step is irrelevant to the sum);
//   * row validity: one ballot says whether all 16 rows of a step pass (then the step runs without masks);
//     otherwise 4 ballots give the row mask (the 16 lanes of a kk group hold the 16 columns of a block) and
//     invalid rows are removed with bit masks;
//   * AUX (p <= 16 T - 2): y and a column of ones ride in the two columns behind x_p of the last block, so X'Wy, the
//     column sums and the y moments come out of the same MFMAs and the per-row side sums (2 T + 3 vector
//     instructions per K-step, each of which costs matrix-core issue time) disappear, as do the separate y loads;
//     the record is written with those two columns taken out again;
//   * four groups per 256-thread workgroup, no LDS, no barrier;
//   * (measured and removed in round 2: the same kernel with the step buffers in a per-wave LDS ring filled by
//     global_load_lds_dwordx4, 3-8 steps ahead and no VGPR staging — correct, but slower at p = 9 / 16 / 24-wls
//     (3.1 / 4.1 / 3.1 TB/s against 3.8 / 4.7 / 3.5 here) and level at p = 32: one DMA instruction covers 8 columns,
//     so 10, 17 or 26 columns waste 37 / 29 / 19 % of the lanes, and the ring's LDS caps the CU at 8 waves)
//   * a group with more than seg_rows rows (>= 1/2048 of the batch) is cut into segments, one wavefront each
//     (accumulate_mid_segments_kernel, idle otherwise), all shifted by the group's first valid row, and the wave
//     that finishes the last segment sums the segment records.
#include <stddef.h>
#include <stdlib.h>

#include "common.h"
#include "lds_dma.h"

// steps per loop trip, by measurement (100 000 x 1000 rows): T = 1: 2 (4 changes nothing; weighted 1 -> 2: 3.19 -> 2.92 ms at
// p = 14); T = 2: 1 (2 spills: 4.9 -> 8.6 ms at p = 24).  More waves per SIMD do not help either (80 VGPRs, 6 waves:
// 1.79 against 1.82 ms at p = 9; 64 VGPRs spill).
#define ANOFOX_MID_S (T == 1 ? 2 : 1)
// what-if builds (never shipped: results are wrong): bit 0 = no row filter, 1 = no constant-column test, 2 = no side sums,
// 3 = no MFMA, 4 = no centring
#ifndef ANOFOX_MID_SKIP
#define ANOFOX_MID_SKIP 0
#endif

namespace anofox {

typedef double mid_dbl2u __attribute__((ext_vector_type(2), aligned(8)));
typedef double mid_dbl4 __attribute__((ext_vector_type(4)));
typedef const double __attribute__((address_space(1))) *mid_gptr_t;
typedef const mid_dbl2u __attribute__((address_space(1))) *mid_gptr2_t;

namespace {

__device__ __forceinline__ double mid_mask(double v, long long m) {
	return __longlong_as_double(__double_as_longlong(v) & m);
}

// Which row of a 16-row step lane group kk holds in its m-th value: rows 2 kk, 2 kk + 1 of the first half, then the
// same of the second half — so that one load instruction covers a contiguous 64-byte half line per column.
// SEQ (the LDS-staged path below): the lane group's four rows are consecutive, rows 4 kk .. 4 kk + 3.
template <bool SEQ>
__device__ __forceinline__ int mid_row(int kk, int m) { return SEQ ? 4 * kk + m : 8 * (m >> 1) + 2 * kk + (m & 1); }

// the lane's four rows of one column, all inside the group: two 16-byte loads (r0 = first row of the step)
__device__ __forceinline__ void load4_full(mid_gptr_t col, int64_t r0, int kk, double (&v)[4]) {
	const mid_dbl2u a = *reinterpret_cast<mid_gptr2_t>(col + r0 + 2 * kk);
	const mid_dbl2u b = *reinterpret_cast<mid_gptr2_t>(col + r0 + 8 + 2 * kk);
	v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
// the same at the end of a group: rows at or past `hi` are clamped (their values are masked later)
__device__ __forceinline__ void load4_tail(mid_gptr_t col, int64_t r0, int kk, int64_t hi, double (&v)[4]) {
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		const int64_t r = r0 + mid_row<false>(kk, m);
		v[m] = col[r < hi ? r : hi - 1];
	}
}

template <int T>
struct MidState {
	mid_dbl4 acc[T * (T + 1) / 2];
	double sx[T], sxy[T], first[T];
	double sy, syy, sw, first_y;
	unsigned ncmask;
	int cnt;
	bool have_first;
};

// One 16-row step of a wave.  ALLVALID: every row of the step passed the row filter (the common case) — no masks.
template <int T, bool WEIGHTED, bool CENTER, bool ALLVALID, bool AUX, bool SEQ>
__device__ __forceinline__ void mid_step(MidState<T> &st, const double (&x)[T][4], const double (&y)[4], const double (&w)[4],
                                         unsigned rowmask, int kk, int lj, bool is_one) {
	if (!st.have_first) {
		const int r = __ffs((int)rowmask) - 1; // first valid row of the group; mid_row(kk, m) == r
		const int src = SEQ ? 16 * (r >> 2) + lj : 16 * ((r >> 1) & 3) + lj, m = SEQ ? (r & 3) : (((r >> 3) << 1) | (r & 1));
#pragma unroll
		for (int I = 0; I < T; ++I) {
			const double mine = m == 0 ? x[I][0] : (m == 1 ? x[I][1] : (m == 2 ? x[I][2] : x[I][3]));
			st.first[I] = __shfl(mine, src, 64);
		}
		if (AUX) {
			if (is_one) st.first[T - 1] = 0.0; // the column of ones is not shifted
		} else {
			const double ym = m == 0 ? y[0] : (m == 1 ? y[1] : (m == 2 ? y[2] : y[3]));
			st.first_y = __shfl(ym, src, 64);
		}
		st.have_first = true;
	}
	st.cnt += ALLVALID ? 16 : __popc(rowmask);
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		const long long rm = ALLVALID ? -1ll : -(long long)((rowmask >> mid_row<SEQ>(kk, m)) & 1u); // all ones when the row is valid
		double d[T], a[T];
#pragma unroll
		for (int I = 0; I < T; ++I) {
			const double dev = ALLVALID ? x[I][m] - st.first[I] : mid_mask(x[I][m] - st.first[I], rm); // deviation from the first valid row
			d[I] = (CENTER && !(ANOFOX_MID_SKIP & 16)) ? dev : (ALLVALID ? x[I][m] : mid_mask(x[I][m], rm));
			// constant-column predicate of ols.rs:76-87: |x - x_first| < 1e-10 on every valid row
			if (!(ANOFOX_MID_SKIP & 2)) st.ncmask |= !(fabs(dev) < 1e-10) ? (1u << I) : 0u;
		}
		const double dy0 = AUX ? 0.0 : (CENTER ? y[m] - st.first_y : y[m]);
		const double dy = ALLVALID ? dy0 : mid_mask(dy0, rm);
		const double wv = ALLVALID ? w[m] : mid_mask(w[m], rm);
#pragma unroll
		for (int I = 0; I < T; ++I) a[I] = WEIGHTED ? wv * d[I] : d[I];
		int tile = 0;
#pragma unroll
		for (int I = 0; I < T; ++I) {
#pragma unroll
			for (int J = I; J < T; ++J) {
				if (!(ANOFOX_MID_SKIP & 8)) st.acc[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], d[J], st.acc[tile], 0, 0, 0);
				else st.acc[tile][0] += a[I] + d[J];
				++tile;
			}
		}
		if (!AUX && !(ANOFOX_MID_SKIP & 4)) {
#pragma unroll
			for (int I = 0; I < T; ++I) {
				st.sx[I] += a[I];
				st.sxy[I] = fma(a[I], dy, st.sxy[I]);
			}
			const double wdy = WEIGHTED ? wv * dy : dy;
			st.sy += wdy;
			st.syy = fma(wdy, dy, st.syy);
			st.sw += wv;
		}
	}
}

// Row validity of one 16-row step starting at row r0 (ols.rs:59-66, wls.rs:76-86), then the step itself.  Common case
// first: a full step whose 16 rows all pass.
template <int T, bool WEIGHTED, bool CENTER, bool AUX, bool SEQ>
__device__ __forceinline__ void mid_process_step(MidState<T> &st, const double (&x)[T][4], const double (&y)[4], const double (&w)[4],
                                                 int64_t r0, int64_t hi, int kk, int lj, bool is_one) {
	// (AUX: y is one of the x columns here — the lane that holds it tests it)
	bool ok_all = true;
	if (ANOFOX_MID_SKIP & 1) {
		if (r0 + 16 <= hi) {
			mid_step<T, WEIGHTED, CENTER, true, AUX, SEQ>(st, x, y, w, 0xFFFFu, kk, lj, is_one);
			return;
		}
	}
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		if (!AUX) ok_all = ok_all && isfinite(y[m]);
		if (WEIGHTED) ok_all = ok_all && isfinite(w[m]) && (w[m] > 0.0);
#pragma unroll
		for (int I = 0; I < T; ++I) ok_all = ok_all && isfinite(x[I][m]);
	}
	if (r0 + 16 <= hi && __ballot(ok_all) == ~0ull) {
		mid_step<T, WEIGHTED, CENTER, true, AUX, SEQ>(st, x, y, w, 0xFFFFu, kk, lj, is_one);
		return;
	}
	unsigned rowmask = 0; // bit = row of the step
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		bool ok = (AUX || isfinite(y[m])) && (r0 + mid_row<SEQ>(kk, m) < hi);
		if (WEIGHTED) ok = ok && isfinite(w[m]) && (w[m] > 0.0);
#pragma unroll
		for (int I = 0; I < T; ++I) ok = ok && isfinite(x[I][m]);
		const unsigned long long b = __ballot(ok);
#pragma unroll
		for (int k = 0; k < 4; ++k) rowmask |= (((b >> (16 * k)) & 0xFFFFull) == 0xFFFFull) ? (1u << mid_row<SEQ>(k, m)) : 0u;
	}
	rowmask = __builtin_amdgcn_readfirstlane(rowmask);
	if (rowmask == 0u) return;
	mid_step<T, WEIGHTED, CENTER, false, AUX, SEQ>(st, x, y, w, rowmask, kk, lj, is_one);
}

template <int T, bool AUX>
__device__ __forceinline__ void mid_init_state(MidState<T> &st, const bool (&real)[T], const double *forced_first, int lj, bool is_y) {
	constexpr int P16 = 16 * T, NT = T * (T + 1) / 2;
#pragma unroll
	for (int t = 0; t < NT; ++t) st.acc[t] = mid_dbl4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
	for (int I = 0; I < T; ++I) st.sx[I] = st.sxy[I] = st.first[I] = 0.0;
	st.ncmask = 0;
	st.sy = st.syy = st.sw = st.first_y = 0.0;
	st.have_first = false;
	st.cnt = 0;
	if (forced_first) {
#pragma unroll
		for (int I = 0; I < T; ++I) st.first[I] = real[I] ? forced_first[16 * I + lj] : 0.0;
		st.first_y = forced_first[P16];
		if (AUX && is_y) st.first[T - 1] = st.first_y; // the lane that carries y in the last block
		st.have_first = true;
	}
}

// the moment record, layout of accumulate_wide.hip.  AUX: columns p (y) and p + 1 (ones) of the last block are taken
// out of the tiles again — their entries ARE X'Wy, the column sums and the y moments — and written where the record
// keeps those.
template <int T, bool AUX>
__device__ __forceinline__ void mid_write_record(const MidState<T> &st, const bool (&real)[T], double *rec, int lane, int p) {
	constexpr int P16 = 16 * T, NT = T * (T + 1) / 2;
	const int kk = lane >> 4, lj = lane & 15;
	{
		int t = 0;
#pragma unroll
		for (int I = 0; I < T; ++I) {
#pragma unroll
			for (int J = I; J < T; ++J) {
				double *tp = rec + (int64_t)t * 256; // tile-major, element (row, col) at row * 16 + col
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const bool keep = !AUX || (16 * J + lj < p && 16 * I + kk + 4 * r < p);
					tp[(kk + 4 * r) * 16 + lj] = keep ? st.acc[t][r] : 0.0;
				}
				++t;
			}
		}
	}
	double *vec = rec + (int64_t)NT * 256;
	double *sc = vec + 4 * P16;
	if (AUX) {
		const int cy = p - 16 * (T - 1), co = cy + 1; // columns of y and of the ones inside the last block
#pragma unroll
		for (int I = 0; I < T; ++I) {
			constexpr int dummy = 0;
			(void)dummy;
			const int t = I * T - I * (I - 1) / 2 + (T - 1 - I); // tile (I, T - 1)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int j = 16 * I + kk + 4 * r; // the row of the tile this lane's element r belongs to
				const double v = j < p ? st.acc[t][r] : 0.0;
				if (lj == co) vec[0 * P16 + j] = v; // sum w d_j
				if (lj == cy) vec[1 * P16 + j] = v; // sum w d_j dy
			}
			unsigned nc = (st.ncmask >> I) & 1u;
			nc |= (unsigned)__shfl_xor((int)nc, 16, 64);
			nc |= (unsigned)__shfl_xor((int)nc, 32, 64);
			if (lane < 16) {
				vec[2 * P16 + 16 * I + lane] = real[I] ? st.first[I] : 0.0;
				vec[3 * P16 + 16 * I + lane] = (real[I] && nc) ? 1.0 : 0.0;
			}
		}
		// diagonal tile of the last block: (cy, co) = sum w dy, (cy, cy) = sum w dy^2, (co, co) = sum w
		constexpr int td = NT - 1;
		auto elem = [&](int r) { return r == 0 ? st.acc[td][0] : (r == 1 ? st.acc[td][1] : (r == 2 ? st.acc[td][2] : st.acc[td][3])); };
		if (lj == co && kk == (cy & 3)) sc[0] = elem(cy >> 2);
		if (lj == cy && kk == (cy & 3)) sc[1] = elem(cy >> 2);
		if (lj == co && kk == (co & 3)) sc[2] = elem(co >> 2);
		if (lane == cy) sc[4] = st.first[T - 1]; // y of the first valid row
		if (lane == 0) sc[3] = (double)st.cnt;
		return;
	}
#pragma unroll
	for (int I = 0; I < T; ++I) { // reduce over the four kk groups (lanes l, l^16, l^32, l^48)
		double a = st.sx[I], b = st.sxy[I];
		a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
		b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
		unsigned nc = (st.ncmask >> I) & 1u;
		nc |= (unsigned)__shfl_xor((int)nc, 16, 64);
		nc |= (unsigned)__shfl_xor((int)nc, 32, 64);
		if (lane < 16) {
			vec[0 * P16 + 16 * I + lane] = a;
			vec[1 * P16 + 16 * I + lane] = b;
			vec[2 * P16 + 16 * I + lane] = st.first[I];
			vec[3 * P16 + 16 * I + lane] = (real[I] && nc) ? 1.0 : 0.0;
		}
	}
	// every lane of a kk group holds the same partial of the y moments: lanes 0, 16, 32, 48
	double sy = st.sy, syy = st.syy, sw = st.sw;
	sy += __shfl_xor(sy, 16, 64); sy += __shfl_xor(sy, 32, 64);
	syy += __shfl_xor(syy, 16, 64); syy += __shfl_xor(syy, 32, 64);
	sw += __shfl_xor(sw, 16, 64); sw += __shfl_xor(sw, 32, 64);
	if (lane == 0) {
		sc[0] = sy;
		sc[1] = syy;
		sc[2] = sw;
		sc[3] = (double)st.cnt;
		sc[4] = st.first_y;
	}
}

// The rows [lo, hi) of one group — or of one segment of a very large group, then with the group's first valid
// row handed in (`forced_first`: x per column, y at index 16 T) — into one wide moment record at `rec`.
template <int T, bool WEIGHTED, bool CENTER, bool AUX>
__device__ __forceinline__ void mid_accumulate_rows(const WideArgs &args, int64_t lo, int64_t hi, double *rec,
                                                    const double *forced_first, int lane) {
	const int p = args.p;
	const int kk = lane >> 4, lj = lane & 15;

	mid_gptr_t col[T];
	bool real[T]; // column 16 I + lj exists (padding columns read column p - 1 and count as zeros)
	// AUX: the lane of the last block that carries y (column p) / the ones (column p + 1)
	const bool is_y = AUX && 16 * (T - 1) + lj == p, is_one = AUX && 16 * (T - 1) + lj == p + 1;
#pragma unroll
	for (int I = 0; I < T; ++I) {
		const int j = 16 * I + lj;
		real[I] = j < p;
		col[I] = (mid_gptr_t)(uintptr_t)args.x_table[real[I] ? j : p - 1];
	}
	if (is_y) col[T - 1] = (mid_gptr_t)(uintptr_t)args.y;
	const mid_gptr_t ycol = (mid_gptr_t)(uintptr_t)args.y;
	const mid_gptr_t wcol = (mid_gptr_t)(uintptr_t)args.w;

	MidState<T> st;
	mid_init_state<T, AUX>(st, real, forced_first, lj, is_y);

	// S steps (16 S rows) per loop trip, the next trip's loads in flight: per column a wave asks for 128 S
	// contiguous bytes at a time (DRAM locality: a single 128-byte line per stream and trip ran at ~4 TB/s)
	constexpr int S = ANOFOX_MID_S;
	double xn[S][T][4], yn[S][4], wn[S][4];
	// every load of a trip sits in ONE arm of the (wave-uniform) full / tail branch: a join between the x and the
	// y loads would make the compiler drain the former before issuing the latter
	auto issue = [&](int64_t r0) {
		if (r0 + 16 * S <= hi) {
#pragma unroll
			for (int q = 0; q < S; ++q) {
				const int64_t r = r0 + 16 * q;
#pragma unroll
				for (int I = 0; I < T; ++I) load4_full(col[I], r, kk, xn[q][I]);
				if (!AUX) load4_full(ycol, r, kk, yn[q]);
				if (WEIGHTED) load4_full(wcol, r, kk, wn[q]);
			}
		} else {
#pragma unroll
			for (int q = 0; q < S; ++q) {
				const int64_t r = r0 + 16 * q;
#pragma unroll
				for (int I = 0; I < T; ++I) load4_tail(col[I], r, kk, hi, xn[q][I]);
				if (!AUX) load4_tail(ycol, r, kk, hi, yn[q]);
				if (WEIGHTED) load4_tail(wcol, r, kk, hi, wn[q]);
			}
		}
	};
	if (lo < hi) issue(lo);
	for (int64_t t0 = lo; t0 < hi; t0 += 16 * S) {
		double xs[S][T][4], ys[S][4], ws[S][4];
#pragma unroll
		for (int q = 0; q < S; ++q) {
#pragma unroll
			for (int m = 0; m < 4; ++m) {
#pragma unroll
				for (int I = 0; I < T; ++I) xs[q][I][m] = real[I] ? xn[q][I][m] : 0.0;
				if (AUX) xs[q][T - 1][m] = (real[T - 1] || is_y) ? xn[q][T - 1][m] : (is_one ? 1.0 : 0.0);
				ys[q][m] = AUX ? 0.0 : yn[q][m];
				ws[q][m] = WEIGHTED ? wn[q][m] : 1.0;
			}
		}
		if (t0 + 16 * S < hi) issue(t0 + 16 * S);
#pragma unroll
		for (int q = 0; q < S; ++q) {
			const int64_t r0 = t0 + 16 * q;
			if (r0 >= hi) break; // wave-uniform
			const double (&x)[T][4] = xs[q];
			const double (&y)[4] = ys[q];
			const double (&w)[4] = ws[q];
			mid_process_step<T, WEIGHTED, CENTER, AUX, false>(st, x, y, w, r0, hi, kk, lj, is_one);
		}
	}

	mid_write_record<T, AUX>(st, real, rec, lane, p);
}

// ---- the same with the rows staged through a wave-private slice of LDS -------------------------------------------------
// The loop above asks for 64 contiguous bytes per column and instruction, 128 S per trip; with 10..34 column streams per
// wave DRAM sees short runs and the kernel stays at 4.4-4.8 TB/s where the narrow kernel (1 KiB per column and tile)
// reaches 6.  Here a block is 64 consecutive rows: lane l loads row l of EVERY column (512 contiguous bytes per
// instruction), D blocks are in flight per wave (D = 2: 1 KiB per column, T = 1; D = 1 for T = 2, whose 33 column
// registers per block are all the VGPR file has room for), the block is written to LDS column by column (stride 66
// doubles: conflict-free for the 8-byte writes and for the 16-byte fragment reads) and read back in MFMA fragment
// layout, lane (kk, lj) taking rows 16 s + 4 kk .. + 3 of column 16 I + lj (two ds_read_b128).  The slice belongs to one
// wavefront: no barrier; LDS operations of a wave execute in order.  y is column p of the slice, the weights p + 1.
typedef double mid_dbl2a __attribute__((ext_vector_type(2)));
__host__ __device__ constexpr int mid_lds_stride(int RL) { return 64 * RL + 2; } // 66 / 130 doubles: an odd number of 16-byte units
// columns of a wave's slice: x, y, (w), and — unless every fragment column is an x column — a column of zeros and one of ones
__host__ __device__ constexpr int mid_lds_columns(int p, bool weighted, int T) { return p + 1 + (weighted ? 1 : 0) + (p != 16 * T ? 2 : 0); }

// One 16-row step of the LDS-staged loop: mid_step without the bookkeeping that loop does per block (first valid row,
// row count), rows 4 kk + m, and for one column block (T = 1) the K-steps alternate between two accumulators — four
// back-to-back MFMAs into one accumulator wait for each other (16 passes each), and with 2 waves per SIMD nobody else fills
// the gap.
template <int T, bool WEIGHTED, bool CENTER, bool ALLVALID, bool AUX, bool SPEC = false>
__device__ __forceinline__ void mid_step_lds(MidState<T> &st, mid_dbl4 (&acc2)[T * (T + 1) / 2], double (&dmax)[T], const double (&x)[T][4],
                                             const double (&y)[4], const double (&w)[4], unsigned rowmask, int kk) {
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		const long long rm = ALLVALID ? -1ll : -(long long)((rowmask >> (4 * kk + m)) & 1u); // all ones when the row is valid
		double d[T], a[T];
#pragma unroll
		for (int I = 0; I < T; ++I) {
			const double dev = ALLVALID ? x[I][m] - st.first[I] : mid_mask(x[I][m] - st.first[I], rm); // deviation from the first valid row
			d[I] = (CENTER && !(ANOFOX_MID_SKIP & 16)) ? dev : (ALLVALID ? x[I][m] : mid_mask(x[I][m], rm));
			// constant-column predicate of ols.rs:76-87, |x - x_first| < 1e-10 on every valid row: the largest |deviation| per
			// lane and column block is kept (one instruction; 0 on rows that do not take part) and tested once per group
			if (!SPEC && !(ANOFOX_MID_SKIP & 2)) dmax[I] = fmax(dmax[I], fabs(dev));
		}
		const double dy0 = AUX ? 0.0 : (CENTER ? y[m] - st.first_y : y[m]);
		const double dy = ALLVALID ? dy0 : mid_mask(dy0, rm);
		const double wv = ALLVALID ? w[m] : mid_mask(w[m], rm);
#pragma unroll
		for (int I = 0; I < T; ++I) a[I] = WEIGHTED ? wv * d[I] : d[I];
		int tile = 0;
#pragma unroll
		for (int I = 0; I < T; ++I) {
#pragma unroll
			for (int J = I; J < T; ++J) {
				if (ANOFOX_MID_SKIP & 8) st.acc[tile][0] += a[I] + d[J];
				else if (T == 1 && (m & 1)) acc2[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], d[J], acc2[tile], 0, 0, 0);
				else st.acc[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], d[J], st.acc[tile], 0, 0, 0);
				++tile;
			}
		}
		if (!AUX && !(ANOFOX_MID_SKIP & 4)) {
#pragma unroll
			for (int I = 0; I < T; ++I) {
				st.sx[I] += a[I];
				st.sxy[I] = fma(a[I], dy, st.sxy[I]);
			}
			const double wdy = WEIGHTED ? wv * dy : dy;
			st.sy += wdy;
			st.syy = fma(wdy, dy, st.syy);
			if (WEIGHTED) st.sw += wv; // (unweighted: the row count, added once per block)
		}
	}
}

__device__ __forceinline__ unsigned mid_spread8(unsigned x) { // bit i of the low byte -> bit 2 i
	x = (x | (x << 4)) & 0x0F0Fu;
	x = (x | (x << 2)) & 0x3333u;
	x = (x | (x << 1)) & 0x5555u;
	return x;
}

// RL = rows per lane and block: 1 = 64-row blocks, 8-byte loads, D = 2 (T = 1) or 1 blocks in flight;
//                                2 = 128-row blocks, 16-byte loads (1 KiB contiguous per column and instruction, what the narrow kernel asks for), D = 1
// The row filter (ols.rs:59-66, wls.rs:76-86) runs on the loaded registers, where a lane holds ITS rows of every column:
// one ballot per block says whether all of its rows pass, and then its 4 RL steps are straight-line code without masks.
template <int T, bool WEIGHTED, bool CENTER, bool AUX, int RL>
__device__ __forceinline__ void mid_accumulate_rows_lds(const WideArgs &args, int64_t lo, int64_t hi, double *rec,
                                                        const double *forced_first, int lane, double *buf) {
	constexpr int NCOL = 16 * T + 2; // most columns a block can have: x (<= 16 T), y, w
	constexpr int NT = T * (T + 1) / 2;
	constexpr int D = (T == 1 && RL == 1) ? 2 : 1;
	constexpr int RS = mid_lds_stride(RL);
	constexpr int BR = 64 * RL; // rows per block
	const int p = args.p;
	const int ncol = p + 1 + (WEIGHTED ? 1 : 0);
	const int kk = lane >> 4, lj = lane & 15;
	const bool is_one = AUX && 16 * (T - 1) + lj == p + 1;
	const bool is_y = AUX && 16 * (T - 1) + lj == p;
	bool real[T], rd[T];
	int ccol[T]; // the slice column behind fragment column 16 I + lj (x, or y in the AUX layout)
#pragma unroll
	for (int I = 0; I < T; ++I) {
		const int c = 16 * I + lj;
		real[I] = c < p;
		rd[I] = real[I] || (AUX && c == p);
		ccol[I] = rd[I] ? c : ((AUX && I == T - 1 && is_one) ? ncol + 1 : ncol);
	}
	MidState<T> st;
	mid_init_state<T, AUX>(st, real, forced_first, lj, is_y);
	mid_dbl4 acc2[NT];
#pragma unroll
	for (int t = 0; t < NT; ++t) acc2[t] = mid_dbl4{0.0, 0.0, 0.0, 0.0};
	double dmax[T];
#pragma unroll
	for (int I = 0; I < T; ++I) dmax[I] = 0.0;
	// two constant columns behind the data in the slice: zeros for the padding columns of the last block, ones for the AUX
	// layout's column of ones — every lane reads its fragment from SOME column and no select follows the read
	// (p = 16 T has neither padding columns nor room for the AUX layout: no lane reads them and the slice ends at ncol)
	if (p != 16 * T) {
		const int czero = ncol, cone = ncol + 1;
#pragma unroll
		for (int e = 0; e < RL; ++e) {
			buf[czero * RS + RL * lane + e] = 0.0;
			buf[cone * RS + RL * lane + e] = 1.0;
		}
	}

	double reg[D][NCOL][RL];
	// Every block issues the same NCOL loads — a load behind a branch of its own (column c exists?) makes the compiler
	// wait for ALL outstanding loads, the prefetched block's included, before the first use; the slots past the last
	// column read y again (the line is in L1) and are not written to LDS.
	auto colp = [&](int c) -> mid_gptr_t {
		const double *ptr = c < p ? args.x_table[c < kWideMaxP ? c : 0] : ((WEIGHTED && c == p + 1) ? args.w : args.y);
		return (mid_gptr_t)(uintptr_t)ptr;
	};
	// all loads of a block in one arm of the wave-uniform full / tail branch (see above)
	auto issue = [&](double (&r)[NCOL][RL], int64_t b) {
		if (b + BR <= hi) {
			const int64_t row = b + RL * lane;
#pragma unroll
			for (int c = 0; c < NCOL; ++c) {
				if (RL == 2) {
					const mid_dbl2u v = *reinterpret_cast<mid_gptr2_t>(colp(c) + row);
					r[c][0] = v.x;
					r[c][RL - 1] = v.y;
				} else {
					r[c][0] = colp(c)[row];
				}
			}
		} else { // clamped; rows past the end fail the row filter below
#pragma unroll
			for (int c = 0; c < NCOL; ++c) {
#pragma unroll
				for (int e = 0; e < RL; ++e) {
					const int64_t row = b + RL * lane + e;
					r[c][e] = colp(c)[row < hi ? row : hi - 1];
				}
			}
		}
	};
	// the fragments of step sidx: lane (kk, lj) takes rows 16 sidx + 4 kk .. + 3 of its columns
	auto fragments = [&](int sidx, double (&x)[T][4], double (&y)[4], double (&w)[4]) {
		const int ro = 16 * sidx + 4 * kk;
#pragma unroll
		for (int I = 0; I < T; ++I) {
			const mid_dbl2a a = *reinterpret_cast<const mid_dbl2a *>(buf + ccol[I] * RS + ro);
			const mid_dbl2a c2 = *reinterpret_cast<const mid_dbl2a *>(buf + ccol[I] * RS + ro + 2);
			x[I][0] = a.x;
			x[I][1] = a.y;
			x[I][2] = c2.x;
			x[I][3] = c2.y;
		}
		if (!AUX) {
			const mid_dbl2a a = *reinterpret_cast<const mid_dbl2a *>(buf + p * RS + ro);
			const mid_dbl2a c2 = *reinterpret_cast<const mid_dbl2a *>(buf + p * RS + ro + 2);
			y[0] = a.x; y[1] = a.y; y[2] = c2.x; y[3] = c2.y;
		} else {
			y[0] = y[1] = y[2] = y[3] = 0.0;
		}
		if (WEIGHTED) {
			const mid_dbl2a a = *reinterpret_cast<const mid_dbl2a *>(buf + (p + 1) * RS + ro);
			const mid_dbl2a c2 = *reinterpret_cast<const mid_dbl2a *>(buf + (p + 1) * RS + ro + 2);
			w[0] = a.x; w[1] = a.y; w[2] = c2.x; w[3] = c2.y;
		} else {
			w[0] = w[1] = w[2] = w[3] = 1.0;
		}
	};
	auto block = [&](double (&r)[NCOL][RL], int64_t b, int64_t b_next) {
		// row filter on this lane's rows
		bool ok[RL];
#pragma unroll
		for (int e = 0; e < RL; ++e) {
			ok[e] = b + RL * lane + e < hi;
			if (!(ANOFOX_MID_SKIP & 1)) {
				// finiteness of the row as ONE number: z = sum 0 * v is NaN iff some v is not finite (the slots past the last
				// column hold y again) — a class test and a mask AND per value cost 14 % of accumulate_quad before this
				double z = 0.0, wv = 1.0;
#pragma unroll
				for (int c = 0; c < NCOL; ++c) {
					z = fma(0.0, r[c][e], z);
					if (WEIGHTED) wv = (c == p + 1) ? r[c][e] : wv;
				}
				ok[e] = ok[e] && (z == 0.0) && (wv > 0.0);
			}
		}
#pragma unroll
		for (int c = 0; c < NCOL; ++c) {
			if (c < ncol) {
				if (RL == 2) *reinterpret_cast<mid_dbl2a *>(buf + c * RS + 2 * lane) = mid_dbl2a{r[c][0], r[c][RL - 1]};
				else buf[c * RS + lane] = r[c][0];
			}
		}
		if (b_next < hi) issue(r, b_next);
		__builtin_amdgcn_wave_barrier();
		const unsigned long long v0 = __ballot(ok[0]), v1 = RL == 2 ? __ballot(ok[RL - 1]) : v0;
		const bool all = (v0 & v1) == ~0ull;
		if ((v0 | v1) != 0ull) {
			if (!st.have_first) { // the first valid row of the group: every lane fetches its columns' values of that row
				const int f0 = __ffsll((long long)v0) - 1, f1 = __ffsll((long long)v1) - 1;
				int fr;
				if (RL == 2) {
					const int r0 = v0 ? 2 * f0 : 1 << 20, r1 = v1 ? 2 * f1 + 1 : 1 << 20;
					fr = r0 < r1 ? r0 : r1;
				} else {
					fr = f0;
				}
#pragma unroll
				for (int I = 0; I < T; ++I) st.first[I] = rd[I] ? buf[ccol[I] * RS + fr] : 0.0; // (constant columns are not shifted)
				if (AUX) {
					if (is_one) st.first[T - 1] = 0.0; // the column of ones is not shifted
				} else {
					st.first_y = buf[p * RS + fr];
				}
				st.have_first = true;
			}
			if (all) {
				st.cnt += BR;
#pragma unroll
				for (int sidx = 0; sidx < 4 * RL; ++sidx) {
					double x[T][4], y[4], w[4];
					fragments(sidx, x, y, w);
					mid_step_lds<T, WEIGHTED, CENTER, true, AUX>(st, acc2, dmax, x, y, w, 0xFFFFu, kk);
				}
			} else {
				st.cnt += __popcll(v0) + (RL == 2 ? __popcll(v1) : 0);
#pragma unroll
				for (int sidx = 0; sidx < 4 * RL; ++sidx) {
					unsigned rowmask;
					if (RL == 2) rowmask = mid_spread8((unsigned)(v0 >> (8 * sidx)) & 0xFFu) | (mid_spread8((unsigned)(v1 >> (8 * sidx)) & 0xFFu) << 1);
					else rowmask = (unsigned)(v0 >> (16 * sidx)) & 0xFFFFu;
					if (rowmask == 0u) continue; // wave-uniform
					double x[T][4], y[4], w[4];
					fragments(sidx, x, y, w);
					mid_step_lds<T, WEIGHTED, CENTER, false, AUX>(st, acc2, dmax, x, y, w, rowmask, kk);
				}
			}
		}
		__builtin_amdgcn_wave_barrier(); // the reads above before the next block's writes
	};
#pragma unroll
	for (int d = 0; d < D; ++d)
		if (lo + BR * d < hi) issue(reg[d], lo + BR * d);
	for (int64_t b = lo; b < hi; b += BR * D) {
#pragma unroll
		for (int d = 0; d < D; ++d) {
			const int64_t bd = b + BR * d;
			if (bd < hi) block(reg[d], bd, bd + BR * D); // wave-uniform
		}
	}
	if (T == 1) {
#pragma unroll
		for (int t = 0; t < NT; ++t) st.acc[t] += acc2[t];
	}
#pragma unroll
	for (int I = 0; I < T; ++I) st.ncmask |= !(dmax[I] < 1e-10) ? (1u << I) : 0u;
	if (!AUX && !WEIGHTED) st.sw = 0.25 * (double)st.cnt; // every lane group adds its quarter: the record sums the four
	mid_write_record<T, AUX>(st, real, rec, lane, p);
}

// ---- (r4) the speculative version on LDS-DMA for THREE and FOUR column tiles (p = 34 .. 64) --------------------------------
// accumulate_wide's workgroup-per-group kernel runs these widths at 0.46-0.49 of the HBM peak with the matrix pipe 62-65 % busy
// (profiles/r04_pmc_wide_t3_t4.md): f64 vector and matrix time ADD, and per 32-row chunk its waves issue ~220 vector / scalar /
// LDS instructions next to 24-40 matrix ones.  Here a wavefront owns its group as in the kernels above, and — the scheme of
// accumulate_quad's speculative kernel — no row passes through a register on its way in: `global_load_lds_dword` moves 32 rows
// of one column per instruction into one half of the wave's 64-row ring (stride 66 doubles, the slice layout of the staged loop
// above), block k + 1 lands while the two 16-row steps of block k read the other half, every row is taken for valid and the
// first row for the shift, and what that assumed is checked on the result: every moment finite, every x column clearly constant
// (sum d^2 < 1e-20) or clearly not (>= n 1e-20).  Anything else goes to the redo list for accumulate_wide's full version.
// Per 32-row block a wave issues p + 1 DMA instructions, 2 x (2 T ds_read_b128 + 4 T subtractions) and the matrix instructions.
// One wavefront per workgroup: the slice is 18-34 KB and the CU holds as many groups as its LDS admits (8 at p = 36, 6 at
// p = 48, 4 at p = 64).
__host__ __device__ constexpr int mid_dma_slice_doubles(int p, int T) { return mid_lds_columns(p, false, T) * mid_lds_stride(1); }

template <int T, bool AUX>
__device__ __forceinline__ bool mid_spec_dma_rows(const WideArgs &args, int64_t lo, int64_t hi, double *rec, int lane, double *buf) {
	constexpr int NT = T * (T + 1) / 2;
	constexpr int RS = mid_lds_stride(1);
	const int p = args.p;
	const int ncol = p + 1;
	const int kk = lane >> 4, lj = lane & 15;
	const bool is_one = AUX && 16 * (T - 1) + lj == p + 1;
	const bool is_y = AUX && 16 * (T - 1) + lj == p;
	bool real[T], rd[T];
	int ccol[T];
#pragma unroll
	for (int I = 0; I < T; ++I) {
		const int c = 16 * I + lj;
		real[I] = c < p;
		rd[I] = real[I] || (AUX && c == p);
		ccol[I] = rd[I] ? c : ((AUX && I == T - 1 && is_one) ? ncol + 1 : ncol);
	}
	MidState<T> st;
	mid_init_state<T, AUX>(st, real, nullptr, lj, is_y);
	mid_dbl4 acc2[NT]; // (only the one-tile kernels use it)
	double dmax[T];
#pragma unroll
	for (int I = 0; I < T; ++I) dmax[I] = 0.0;
	if (p != 16 * T) { // the constant columns of the slice, both halves of the ring
		buf[ncol * RS + lane] = 0.0;
		buf[(ncol + 1) * RS + lane] = 1.0;
	}
	const unsigned lds0 = lds_dma_address(buf);
	// the column pointers are re-read from the kernel arguments for every block (lds_dma.h: kept resident they spill into VGPR lanes)
	const lds_dma_table_t tab = lds_dma_table((unsigned)offsetof(WideArgs, x_table)); // (y sits behind the last feature: host_api.hip)
	auto dma = [&](int64_t blk, int h) {
		lds_dma_block<RS * 8, 16 * T + 1>(tab, ncol, blk, lds_dma_offsets(lane, hi - blk), lds0 + (unsigned)h * 256u);
	};
	auto fragments = [&](const double *hb, int sidx, double (&x)[T][4], double (&y)[4]) {
		const int ro = 16 * sidx + 4 * kk;
#pragma unroll
		for (int I = 0; I < T; ++I) {
			const mid_dbl2a a = *reinterpret_cast<const mid_dbl2a *>(hb + ccol[I] * RS + ro);
			const mid_dbl2a c2 = *reinterpret_cast<const mid_dbl2a *>(hb + ccol[I] * RS + ro + 2);
			x[I][0] = a.x; x[I][1] = a.y; x[I][2] = c2.x; x[I][3] = c2.y;
		}
		if (!AUX) {
			const mid_dbl2a a = *reinterpret_cast<const mid_dbl2a *>(hb + p * RS + ro);
			const mid_dbl2a c2 = *reinterpret_cast<const mid_dbl2a *>(hb + p * RS + ro + 2);
			y[0] = a.x; y[1] = a.y; y[2] = c2.x; y[3] = c2.y;
		} else {
			y[0] = y[1] = y[2] = y[3] = 0.0;
		}
	};
	const double w1[4] = {1.0, 1.0, 1.0, 1.0};
	int h = 0;
	dma(lo, 0);
	lds_dma_wait_all();
	__builtin_amdgcn_wave_barrier();
	{ // the shift: the group's first row (constant columns are not shifted)
#pragma unroll
		for (int I = 0; I < T; ++I) st.first[I] = rd[I] ? buf[ccol[I] * RS] : 0.0;
		if (AUX) {
			if (is_one) st.first[T - 1] = 0.0;
		} else {
			st.first_y = buf[p * RS];
		}
		st.have_first = true;
	}
	// Full blocks in a loop WITHOUT a branch around the matrix instructions (with the full / partial choice inside the loop the
	// compiler moved the 40-80 accumulator registers between register banks at every join: 240 v_accvgpr_mov per trip), the
	// partial last block after it.
	int64_t blk = lo;
	for (; blk + 32 <= hi; blk += 32, h ^= 1) {
		if (blk + 32 < hi) dma(blk + 32, h ^ 1); // lands while this block's steps run
		__builtin_amdgcn_wave_barrier();
		const double *hb = buf + 32 * h;
#pragma unroll
		for (int sidx = 0; sidx < 2; ++sidx) {
			double x[T][4], y[4];
			fragments(hb, sidx, x, y);
			mid_step_lds<T, false, true, true, AUX, true>(st, acc2, dmax, x, y, w1, 0xFFFFu, kk);
		}
		__builtin_amdgcn_wave_barrier();
		lds_dma_wait_all(); // the next block has landed (it had this block's steps to do so)
	}
	if (blk < hi) {
		const int64_t left = hi - blk;
		const unsigned m32 = (1u << (unsigned)left) - 1u; // 1 <= left < 32
		const double *hb = buf + 32 * h;
#pragma unroll
		for (int sidx = 0; sidx < 2; ++sidx) {
			const unsigned rowmask = (m32 >> (16 * sidx)) & 0xFFFFu;
			if (rowmask == 0u) continue; // wave-uniform
			double x[T][4], y[4];
			fragments(hb, sidx, x, y);
			mid_step_lds<T, false, true, false, AUX, true>(st, acc2, dmax, x, y, w1, rowmask, kk);
		}
	}
	// ---- what the speculation assumed, checked on the result ----
	double zs = 0.0;
#pragma unroll
	for (int t = 0; t < NT; ++t) zs = fma(st.acc[t][0], 0.0, fma(st.acc[t][1], 0.0, fma(st.acc[t][2], 0.0, fma(st.acc[t][3], 0.0, zs))));
	if (!AUX) {
#pragma unroll
		for (int I = 0; I < T; ++I) zs = fma(st.sx[I], 0.0, fma(st.sxy[I], 0.0, zs));
		zs = fma(st.sy, 0.0, fma(st.syy, 0.0, zs));
	}
	bool bad = isnan(zs);
	const double n_d = (double)(hi - lo);
	st.ncmask = 0u;
#pragma unroll
	for (int I = 0; I < T; ++I) {
		// diagonal element (j, j) of tile (I, I), j = lj: on the lane with kk == j % 4, in element j / 4
		const int t = I * T - I * (I - 1) / 2; // tile (I, I)
		if (kk == (lj & 3)) {
			double mjj = st.acc[t][0];
#pragma unroll
			for (int r = 1; r < 4; ++r) mjj = (lj >> 2) == r ? st.acc[t][r] : mjj;
			const bool moved = mjj >= n_d * 1e-20;
			if (real[I]) {
				bad = bad || (!moved && !(mjj < 1e-20));
				if (moved) st.ncmask |= 1u << I;
			}
		}
	}
	if (__ballot(bad) != 0ull) return false;
	st.cnt = (int)(hi - lo);
	if (!AUX) st.sw = 0.25 * (double)st.cnt; // every lane group adds its quarter: the record sums the four
	mid_write_record<T, AUX>(st, real, rec, lane, p);
	return true;
}

template <int T, bool AUX>
__global__ __launch_bounds__(64) void accumulate_tile_spec_kernel(WideArgs args) {
	extern __shared__ double tile_lds[];
	const int lane = threadIdx.x;
	const int64_t gl = blockIdx.x;
	const int64_t lo = args.row_offsets[args.group_base + gl];
	const int64_t hi = group_row_end(args, args.group_base + gl);
	if (args.seg_table && hi - lo > args.seg_rows) {
		// (the host sized the table for accumulate_wide's workgroup-per-segment kernel, which takes these groups)
		if (wide_register_big_group(args, gl, lo, hi, T, lane, kWideSegMaxBig, kWideSegMaxSegments)) return;
	}
	if (hi > lo && mid_spec_dma_rows<T, AUX>(args, lo, hi, args.moments + gl * (int64_t)wide_record_len(T), lane, tile_lds)) return;
	if (lane == 0) args.refine_list[atomicAdd(args.refine_count + kWideRedoCounter, 1)] = (int32_t)gl;
}

template <int T>
hipError_t launch_tile_spec_T(const WideArgs &a, hipStream_t stream) {
	const dim3 grid((unsigned)a.n_groups), block(64);
	const size_t lds_bytes = (size_t)mid_dma_slice_doubles(a.p, T) * sizeof(double);
	static const bool attr_set = [] {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_tile_spec_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_tile_spec_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		return true;
	}();
	(void)attr_set;
	if (a.p + 2 <= 16 * T) hipLaunchKernelGGL((accumulate_tile_spec_kernel<T, true>), grid, block, lds_bytes, stream, a);
	else hipLaunchKernelGGL((accumulate_tile_spec_kernel<T, false>), grid, block, lds_bytes, stream, a);
	const hipError_t rc = hipGetLastError();
	if (rc != hipSuccess) return rc;
	return launch_accumulate_wide_followup(a, stream); // its segment kernel, and the full version on the redo list
}

template <int T, bool WEIGHTED, bool CENTER, bool AUX, int LDSX> // LDSX: 0 = straight into fragment layout, 1 / 2 = through LDS, rows per lane
__global__ __launch_bounds__(256) void accumulate_mid_kernel(WideArgs args) {
	const int lane = threadIdx.x & 63;
	int64_t gl = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	if (args.from_redo_list) { // the groups a speculative kernel (accumulate_quad.hip) gave up on
		if (gl >= args.refine_count[kWideRedoCounter]) return;
		gl = args.refine_list[gl];
	} else if (gl >= args.n_groups) {
		return;
	}
	const int64_t lo = args.row_offsets[args.group_base + gl];
	const int64_t hi = group_row_end(args, args.group_base + gl);
	if (!args.from_redo_list && args.seg_table && hi - lo > args.seg_rows) {
		if (wide_register_big_group(args, gl, lo, hi, T, lane, kSegMaxBig, kSegMaxSegments)) return;
	}
	double *rec = args.moments + gl * (int64_t)wide_record_len(T);
	if (LDSX) {
		extern __shared__ double mid_lds[];
		constexpr int RLX = LDSX ? LDSX : 1;
		mid_accumulate_rows_lds<T, WEIGHTED, CENTER, AUX, RLX>(args, lo, hi, rec, nullptr, lane, mid_lds + (threadIdx.x >> 6) * mid_lds_columns(args.p, WEIGHTED, T) * mid_lds_stride(RLX));
	} else {
		mid_accumulate_rows<T, WEIGHTED, CENTER, AUX>(args, lo, hi, rec, nullptr, lane);
	}
}

// One wavefront per registered segment; every segment of a group uses the group's first valid row as its shift,
// so the wave that completes the last one merges by plain (ordered) sums.
template <int T, bool WEIGHTED, bool CENTER, bool AUX>
__global__ __launch_bounds__(256) void accumulate_mid_segments_kernel(WideArgs args) {
	constexpr int P16 = 16 * T;
	constexpr int NT = T * (T + 1) / 2;
	const int reclen = wide_record_len(T);
	const int lane = threadIdx.x & 63;
	const int v = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (int)blockIdx.x * 4;
	SegHeader *h = wseg_header(args.seg_table);
	const int total = h->seg_total; // reservations never exceed the capacity
	if (v >= total) return;
	const SegEntry e = wseg_entries(args.seg_table, kSegMaxBig)[v];
	if (e.slot < 0) return; // reserved but unclaimed
	SegBigGroup *b = wseg_big(args.seg_table) + e.slot;
	const double *ff = wseg_first(args.seg_table, kSegMaxBig, kSegMaxSegments) + (size_t)e.slot * (P16 + 2);
	double *recs = wseg_records(args.seg_table, T, kSegMaxBig, kSegMaxSegments);
	mid_accumulate_rows<T, WEIGHTED, CENTER, AUX>(args, e.lo, e.hi, recs + (int64_t)v * reclen, ff, lane);
	__threadfence(); // this segment's record before the counter
	int old = 0;
	if (lane == 0) old = atomicAdd(&b->done, 1);
	old = __builtin_amdgcn_readfirstlane(old);
	if (old != b->nseg - 1) return;
	__threadfence(); // every other segment's record after the counter
	const double *src = recs + (int64_t)b->base * reclen;
	double *dst = args.moments + b->g * (int64_t)reclen;
	const int vec0 = NT * 256;
	for (int k = lane; k < reclen; k += 64) {
		const bool is_first = (k >= vec0 + 2 * P16 && k < vec0 + 3 * P16) || k == vec0 + 4 * P16 + 4; // first x / first y: shared
		const bool is_flag = k >= vec0 + 3 * P16 && k < vec0 + 4 * P16;                                   // non-constant flags: OR
		double acc = 0.0;
		for (int t = 0; t < b->nseg; ++t) acc += src[(int64_t)t * reclen + k];
		if (is_first) acc = src[k];
		if (is_flag) acc = acc > 0.0 ? 1.0 : 0.0;
		dst[k] = acc;
	}
}

template <int T>
hipError_t launch_mid_T(const WideArgs &a, hipStream_t stream) {
	const bool weighted = a.model == ANOFOX_HIP_MODEL_WLS;
	const bool center = a.fit_intercept != 0;
	const dim3 grid((unsigned)((a.n_groups + 3) / 4)), block(256);
	const dim3 seg_grid((unsigned)((kSegMaxSegments + 3) / 4)); // idle unless some group exceeded seg_rows
	static const bool aux_on = !(getenv("ANOFOX_MID_AUX") && atoi(getenv("ANOFOX_MID_AUX")) == 0); // A/B switch
	const bool aux = aux_on && a.p + 2 <= 16 * T; // room for y and the ones in the last block
	// The rows are loaded straight into fragment layout (0) or staged through LDS in 64-row (1) / 128-row (2) blocks.  By
	// measurement (100 000 groups x 1000 rows, profiles/r03_mid_paths.txt): 128-row blocks win at p = 12 .. 16 while two
	// workgroups' slices fit a CU (4.6-4.9 against 4.4-4.6 TB/s), 64-row blocks at p >= 29 (4.2-4.4 against 3.9-4.3), the
	// direct loads elsewhere (p <= 11: the staged loop always issues 18 column loads; 17 <= p <= 28: three tiles of MFMAs
	// set the pace and the fewer VGPRs of the direct loop keep one more wave per SIMD).  ANOFOX_MID_LDS=0/1/2 forces one.
	static const int forced = getenv("ANOFOX_MID_LDS") ? atoi(getenv("ANOFOX_MID_LDS")) : -1;
	int ldsx = 0;
	if (T == 1 && a.p >= 12 && 4 * (size_t)mid_lds_columns(a.p, weighted, T) * mid_lds_stride(2) * sizeof(double) <= 80 * 1024) ldsx = 2;
	if (T == 2 && a.p >= 27) ldsx = 1; // (since the row filter became one FMA per value: 4.4-4.6 against 3.9-4.3 TB/s at p = 27 .. 32)
	if (forced >= 0 && forced <= 2) ldsx = forced;
	const size_t lds_bytes = 4 * (size_t)mid_lds_columns(a.p, weighted, T) * mid_lds_stride(ldsx == 2 ? 2 : 1) * sizeof(double); // 4 waves' slices
	static const bool attr_set = [] {
#define ANOFOX_MID_ATTR(W, C, X) \
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_mid_kernel<T, W, C, X, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_mid_kernel<T, W, C, X, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
		ANOFOX_MID_ATTR(true, true, true); ANOFOX_MID_ATTR(true, true, false); ANOFOX_MID_ATTR(true, false, true); ANOFOX_MID_ATTR(true, false, false);
		ANOFOX_MID_ATTR(false, true, true); ANOFOX_MID_ATTR(false, true, false); ANOFOX_MID_ATTR(false, false, true); ANOFOX_MID_ATTR(false, false, false);
#undef ANOFOX_MID_ATTR
		return true;
	}();
	(void)attr_set;
#define ANOFOX_MID_LAUNCH2(W, C, X)                                                                              \
	do {                                                                                                         \
		if (ldsx == 2) hipLaunchKernelGGL((accumulate_mid_kernel<T, W, C, X, 2>), grid, block, lds_bytes, stream, a); \
		else if (ldsx == 1) hipLaunchKernelGGL((accumulate_mid_kernel<T, W, C, X, 1>), grid, block, lds_bytes, stream, a); \
		else hipLaunchKernelGGL((accumulate_mid_kernel<T, W, C, X, 0>), grid, block, 0, stream, a);             \
		if (a.seg_table) hipLaunchKernelGGL((accumulate_mid_segments_kernel<T, W, C, X>), seg_grid, block, 0, stream, a); \
	} while (0)
#define ANOFOX_MID_LAUNCH(W, C)                                                                                  \
	do {                                                                                                         \
		if (aux) ANOFOX_MID_LAUNCH2(W, C, true);                                                                 \
		else ANOFOX_MID_LAUNCH2(W, C, false);                                                                    \
	} while (0)
	if (weighted) {
		if (center) ANOFOX_MID_LAUNCH(true, true);
		else ANOFOX_MID_LAUNCH(true, false);
	} else {
		if (center) ANOFOX_MID_LAUNCH(false, true);
		else ANOFOX_MID_LAUNCH(false, false);
	}
#undef ANOFOX_MID_LAUNCH
#undef ANOFOX_MID_LAUNCH2
	return hipGetLastError();
}

} // namespace

bool accumulate_mid_supports(int p) { return p > kNarrowMaxP && p <= 32; }

namespace {
template <int T>
hipError_t launch_mid_segments_T(const WideArgs &a, hipStream_t stream) {
	const bool weighted = a.model == ANOFOX_HIP_MODEL_WLS;
	const bool center = a.fit_intercept != 0;
	const dim3 block(256), seg_grid((unsigned)((kSegMaxSegments + 3) / 4));
	const bool aux = a.p + 2 <= 16 * T;
#define ANOFOX_MID_SEG(W, C)                                                                                       \
	do {                                                                                                           \
		if (aux) hipLaunchKernelGGL((accumulate_mid_segments_kernel<T, W, C, true>), seg_grid, block, 0, stream, a); \
		else hipLaunchKernelGGL((accumulate_mid_segments_kernel<T, W, C, false>), seg_grid, block, 0, stream, a);  \
	} while (0)
	if (weighted) {
		if (center) ANOFOX_MID_SEG(true, true);
		else ANOFOX_MID_SEG(true, false);
	} else {
		if (center) ANOFOX_MID_SEG(false, true);
		else ANOFOX_MID_SEG(false, false);
	}
#undef ANOFOX_MID_SEG
	return hipGetLastError();
}
} // namespace

// only the row-segment kernel (accumulate_quad.hip registers its very large groups in the same table)
hipError_t launch_accumulate_mid_segments(const WideArgs &a, hipStream_t stream) {
	if (!a.seg_table) return hipSuccess;
	switch (wide_tiles(a.p)) {
	case 1: return launch_mid_segments_T<1>(a, stream);
	case 2: return launch_mid_segments_T<2>(a, stream);
	default: return hipErrorInvalidValue;
	}
}

// the full version on the redo list of a speculative kernel: the batch's grid, the segment kernel not launched again
hipError_t launch_accumulate_mid_redo(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	WideArgs b = a;
	b.from_redo_list = 1;
	b.seg_table = nullptr;
	return launch_accumulate_mid(b, stream);
}

// (r4) p = 34 .. 64, unweighted with an intercept: the speculative LDS-DMA kernel on 16 x 16 tiles.  OFF by default — measured
// SLOWER than accumulate_wide's workgroup-per-group kernel at every one of these widths (100 000 x 1000 rows, same box: p = 34
// 3.7 against 3.8 TB/s, p = 48 3.9 against 4.3, p = 64 2.9-3.1 against 3.9; docs/HISTORY.md): three or four column tiles cost 96 /
// 160 matrix cycles per row whatever is done about the loads, and one wavefront per 20-35 KB slice leaves 4-6 waves per CU to hide
// them.  Kept as a measurement switch (ANOFOX_TILE_SPEC=1) and as the test vehicle of lds_dma.h's partial-block handling.
bool accumulate_tile_supports(int p, bool weighted, bool center, bool no_fast_path) {
	static const bool on = getenv("ANOFOX_TILE_SPEC") && atoi(getenv("ANOFOX_TILE_SPEC")) == 1;
	static const int max_p = getenv("ANOFOX_TILE_SPEC_MAXP") ? atoi(getenv("ANOFOX_TILE_SPEC_MAXP")) : 64;
	return on && p >= 34 && p <= 64 && p <= max_p && !weighted && center && !no_fast_path;
}

hipError_t launch_accumulate_tile(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	switch (wide_tiles(a.p)) {
	case 3: return launch_tile_spec_T<3>(a, stream);
	case 4: return launch_tile_spec_T<4>(a, stream);
	default: return hipErrorInvalidValue;
	}
}

hipError_t launch_accumulate_mid(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	switch (wide_tiles(a.p)) {
	case 1: return launch_mid_T<1>(a, stream);
	case 2: return launch_mid_T<2>(a, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace anofox
