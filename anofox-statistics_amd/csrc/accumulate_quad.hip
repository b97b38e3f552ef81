// accumulate_quad.hip — moment accumulation for 8 < p <= 32 on v_mfma_f64_4x4x4_4b_f64: one WAVEFRONT per group, the
// Gram matrix of the p + 2 columns (x, y, ones) in 4 x 4 blocks.
//
// Same role and the same record as accumulate_mid.hip (reference: the row buffering + dense decomposition of
// src/aggregate_functions/ols_aggregate.cpp:120-186,249-296, row filter / constant test of
// crates/anofox-stats-core/src/models/ols.rs:59-87).  accumulate_mid pads every column block to 16: 18 columns (p = 16) cost
// two blocks = three 16 x 16 tiles, of which a third is padding and half of each diagonal tile a mirror image — and on this
// part the matrix cores ARE the bound of these widths (f64 VALU and MFMA time add up; profiles/r03_mid_paths.txt).  The small
// f64 shape runs at the same flop rate (csrc/tools/mfma_f64_4x4_rate: 69-71 TFLOP/s against 66 for 16x16x4, 16 cycles per
// instruction) and takes FOUR independent 4 x 4 x 4 products per instruction:
//   operands  A[b][i][k] on lane 16 k + 4 b + i,  B[b][k][j] on lane 16 k + 4 b + j,  D[b][i][j] on lane 16 i + 4 b + j
//   (csrc/tools/mfma_4x4_layout prints this from one-hot inputs).
// Let the four blocks b be four ROW QUADS of a 16-row step (row 4 k + b of the step in block b, K-step k) and let lane
// (k, b, c4) hold its row's values of the columns 4 g + c4, g = 0 .. NB - 1, one register per column group g.  Then register
// g IS the A operand of "column group g as rows" and the B operand of "column group g as columns" — as in the 16 x 16 kernels
// — and one instruction per pair g <= h accumulates the 4 x 4 block (g, h) of the Gram matrix over 16 rows, as four partial
// sums (one per row quad) that are added once per group.  NB (NB + 1) / 2 instructions of 16 cycles per 16 rows, NB =
// ceil((p + 2) / 4): p = 16: 15 x 16 = 240 cycles (accumulate_mid: 256 + the side sums on the vector unit), p = 24: 448 (768),
// p = 32: 720 (768 + side sums).  y and the column of ones are columns p and p + 1 of the matrix, so X'Wy, the column sums
// and the y moments come out of the same instructions.
//
// Rows arrive as in accumulate_mid's staged path: lane l loads ITS rows of every column (128-row blocks and 16-byte loads
// up to p = 16, 64-row blocks beyond), the row filter runs on those registers (one ballot per block), the block goes to a
// wave-private LDS slice column by column (stride = 8 mod 32 doubles: the 32 lanes of a half wave read rows 0..7 of four
// columns, 32 different banks) and every lane reads one double per column group and step.  No barrier.
#include <stddef.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "lds_dma.h"

// what-if builds (never shipped: results are wrong): bit 0 = no MFMA, 1 = no row filter, 2 = no constant-column test,
// 3 = a block's rows are not written to LDS, 4 = one step per block instead of 4 RL
#ifndef ANOFOX_QUAD_SKIP
#define ANOFOX_QUAD_SKIP 0
#endif
#ifndef ANOFOX_QUAD_NOUNROLL_NB
#define ANOFOX_QUAD_NOUNROLL_NB 8 // from this many column groups on the steps of a block are not unrolled in pairs (registers)
#endif

namespace anofox {

hipError_t launch_accumulate_mid_segments(const WideArgs &a, hipStream_t stream); // accumulate_mid.hip

namespace {

typedef double quad_dbl2u __attribute__((ext_vector_type(2), aligned(8)));
typedef double quad_dbl2a __attribute__((ext_vector_type(2)));
typedef const double __attribute__((address_space(1))) *quad_gptr_t;
typedef const quad_dbl2u __attribute__((address_space(1))) *quad_gptr2_t;

__host__ __device__ constexpr int quad_lds_stride(int RL) { return 64 * RL + 8; } // 72 / 136 doubles
__host__ __device__ constexpr int quad_blocks(int p) { return (p + 2 + 3) / 4; }  // NB: 4 x 4 column groups of x, y, ones
// a wave's slice: the data columns (x, y, w) of a block, and at the end of the group the Gram image + first row + flags
__host__ __device__ constexpr int quad_slice_doubles(int p, bool weighted, int RL) {
	const int data = (p + 1 + (weighted ? 1 : 0)) * quad_lds_stride(RL);
	const int R = 4 * quad_blocks(p);
	const int image = R * R + 2 * R;
	return data > image ? data : image;
}

__device__ __forceinline__ double quad_mask(double v, long long m) { return __longlong_as_double(__double_as_longlong(v) & m); }

__device__ __forceinline__ unsigned quad_spread8(unsigned x) { // bit i of the low byte -> bit 2 i
	x = (x | (x << 4)) & 0x0F0Fu;
	x = (x | (x << 2)) & 0x3333u;
	x = (x | (x << 1)) & 0x5555u;
	return x;
}

// The end of a group, shared by the loaders: the four row quads' partial sums, the Gram image through LDS, (SPEC) the check of what
// the speculation assumed, then accumulate_mid's record layout.
template <int NB, bool SPEC>
__device__ __forceinline__ bool quad_finish(const WideArgs &args, double *rec, int lane, double *buf, double (&acc)[NB * (NB + 1) / 2],
                                            const double (&first)[NB], double (&dmax)[NB], bool isx_prev, bool isx_last, int cnt,
                                            int64_t lo, int64_t hi) {
	constexpr int NPAIR = NB * (NB + 1) / 2;
	const int p = args.p;
	const int T = wide_tiles(p), P16 = 16 * T, NT = T * (T + 1) / 2;
	const int b = (lane >> 2) & 3;
	// ---- the record: the four row quads' partial sums, the Gram image through LDS, then accumulate_mid's layout ----
#pragma unroll
	for (int t = 0; t < NPAIR; ++t) {
		acc[t] += __shfl_xor(acc[t], 4, 64);
		acc[t] += __shfl_xor(acc[t], 8, 64);
	}
#pragma unroll
	for (int g = 0; g < (SPEC ? 0 : NB); ++g) { // the largest deviation of column 4 g + c4 over all rows: the 16 lanes that share c4
		double m = dmax[g];
		m = fmax(m, __shfl_xor(m, 4, 64));
		m = fmax(m, __shfl_xor(m, 8, 64));
		m = fmax(m, __shfl_xor(m, 16, 64));
		m = fmax(m, __shfl_xor(m, 32, 64));
		dmax[g] = m;
	}
	constexpr int R = 4 * NB;
	double *G = buf, *F = buf + R * R, *NC = F + R;
	{
		const int i = lane >> 4, j = lane & 3; // D[b][i][j] on lane 16 i + 4 b + j: the b = 0 lanes write
		int t = 0;
#pragma unroll
		for (int g = 0; g < NB; ++g) {
#pragma unroll
			for (int h = g; h < NB; ++h) {
				if (b == 0) {
					G[(4 * g + i) * R + 4 * h + j] = acc[t];
					if (h != g) G[(4 * h + j) * R + 4 * g + i] = acc[t];
				}
				++t;
			}
		}
		if (lane < 4) {
#pragma unroll
			for (int g = 0; g < NB; ++g) {
				F[4 * g + lane] = first[g];
				const bool isx = g < NB - 2 ? true : (g == NB - 2 ? isx_prev : isx_last);
				if (!SPEC) NC[4 * g + lane] = (isx && !(dmax[g] < 1e-10)) ? 1.0 : 0.0;
			}
		}
	}
	__builtin_amdgcn_wave_barrier();
	if (SPEC) {
		// what the speculation assumed, checked on the result (see quad_accumulate_rows): every moment finite, every x column
		// clearly constant (sum d^2 < 1e-20) or clearly not (sum d^2 >= n 1e-20)
		double zs = 0.0;
#pragma unroll
		for (int t = 0; t < NPAIR; ++t) zs = fma(acc[t], 0.0, zs);
		bool bad = isnan(zs);
		const double n_d = (double)(hi - lo);
		for (int j = lane; j < R; j += 64) {
			const double mjj = G[j * R + j];
			const bool moved = mjj >= n_d * 1e-20;
			if (j < p) bad = bad || (!moved && !(mjj < 1e-20));
			NC[j] = (j < p && moved) ? 1.0 : 0.0;
		}
		if (__ballot(bad) != 0ull) return false;
		cnt = (int)(hi - lo);
		__builtin_amdgcn_wave_barrier();
	}
	for (int tile = 0, I = 0; I < T; ++I) {
		for (int J = I; J < T; ++J, ++tile) {
			double *tp = rec + (int64_t)tile * 256; // tile-major, element (row, col) at row * 16 + col
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int e = lane + 64 * q, r = 16 * I + (e >> 4), c = 16 * J + (e & 15);
				tp[e] = (r < p && c < p) ? G[r * R + c] : 0.0;
			}
		}
	}
	double *vec = rec + (int64_t)NT * 256;
	for (int j = lane; j < P16; j += 64) {
		const bool in = j < p;
		vec[0 * P16 + j] = in ? G[j * R + p + 1] : 0.0; // sum w d_j
		vec[1 * P16 + j] = in ? G[j * R + p] : 0.0;     // sum w d_j dy
		vec[2 * P16 + j] = in ? F[j] : 0.0;             // x at the first valid row
		vec[3 * P16 + j] = in ? NC[j] : 0.0;            // not constant
	}
	double *sc = vec + 4 * P16;
	if (lane == 0) {
		sc[0] = G[p * R + p + 1];       // sum w dy
		sc[1] = G[p * R + p];           // sum w dy^2
		sc[2] = G[(p + 1) * R + p + 1]; // sum w
		sc[3] = (double)cnt;
		sc[4] = F[p];                   // y of the first valid row
	}
	return true;
}

// SPEC (r4): the speculative version for the common case — no weights, an intercept.  It takes every row for valid and the
// first row for the shift, so the per-row finiteness test, the per-step constant-column test (NB running maxima in
// registers) and every mask disappear from the loop; what it assumed is checked ON THE RESULT: every moment finite (a NaN /
// inf anywhere in a row poisons at least one sum) and every x column clearly constant (sum d^2 < 1e-20, which implies every
// |d| < 1e-10) or clearly not (sum d^2 >= n 1e-20, which rules out "every |d| < 1e-10") — the predicate of ols.rs:76-87
// decided exactly in both cases.  Anything else returns false and the group goes to the full version through the redo
// list (the scheme of accumulate_wide's speculative kernel).  The registers it frees are what lets NB = 8, 9 (p = 27 .. 34)
// run here at two waves per SIMD.
template <int NB, bool WEIGHTED, bool CENTER, int RL, bool SPEC = false>
__device__ __forceinline__ bool quad_accumulate_rows(const WideArgs &args, int64_t lo, int64_t hi, double *rec, int lane, double *buf) {
	static_assert(!SPEC || (CENTER && !WEIGHTED), "the speculative version exists for the unweighted fit with an intercept");
	constexpr int NCOL = 4 * NB; // load slots per block: x (p), y, (w) — p + 2 <= 4 NB
	constexpr int NPAIR = NB * (NB + 1) / 2;
	constexpr int RS = quad_lds_stride(RL);
	constexpr int BR = 64 * RL;
	const int p = args.p;
	const int ncol = p + 1 + (WEIGHTED ? 1 : 0);
	const int k = lane >> 4, b = (lane >> 2) & 3, c4 = lane & 3;
	const int rsub = 4 * k + b; // this lane's row of every 16-row step

	// This lane's columns: 4 g + c4.  NB = ceil((p + 2) / 4), so the columns of the groups g < NB - 1 are all <= p — x columns
	// or y, read from the slice at base_off + 4 g RS (a compile-time offset per group: no address registers) — and only
	// the LAST group can hold the column of ones (p + 1) or padding, which are constants and read nothing.
	const int base_off = c4 * RS + rsub;
	const int c_last = 4 * (NB - 1) + c4;
	const bool rd_last = c_last <= p;
	const int last_off = (rd_last ? c_last : p) * RS + rsub;
	const double fill_last = c_last == p + 1 ? 1.0 : 0.0;
	const bool isx_prev = 4 * (NB - 2) + c4 < p, isx_last = c_last < p; // x columns take part in the constant-column test
	double acc[NPAIR];
#pragma unroll
	for (int t = 0; t < NPAIR; ++t) acc[t] = 0.0;
	double first[NB], dmax[NB];
#pragma unroll
	for (int g = 0; g < NB; ++g) first[g] = dmax[g] = 0.0;
	bool have_first = false;
	int cnt = 0;

	double reg[NCOL][RL];
	// every block issues the same NCOL loads (a load behind a branch of its own makes the compiler wait for all outstanding
	// loads before the first use); the slots past the last column read y again (the line is in L1) and are not written to LDS
	auto colp = [&](int c) -> quad_gptr_t {
		const double *ptr = c < p ? args.x_table[c < kWideMaxP ? c : 0] : ((WEIGHTED && c == p + 1) ? args.w : args.y);
		return (quad_gptr_t)(uintptr_t)ptr;
	};
	auto issue = [&](int64_t blk) {
		if (blk + BR <= hi) {
			const int64_t row = blk + RL * lane;
#pragma unroll
			for (int c = 0; c < NCOL; ++c) {
				if (RL == 2) {
					const quad_dbl2u v = *reinterpret_cast<quad_gptr2_t>(colp(c) + row);
					reg[c][0] = v.x;
					reg[c][RL - 1] = v.y;
				} else {
					reg[c][0] = colp(c)[row];
				}
			}
		} else { // clamped; rows past the end fail the row filter below
#pragma unroll
			for (int c = 0; c < NCOL; ++c) {
#pragma unroll
				for (int e = 0; e < RL; ++e) {
					const int64_t row = blk + RL * lane + e;
					reg[c][e] = colp(c)[row < hi ? row : hi - 1];
				}
			}
		}
	};
	// one 16-row step: this lane's row is rsub of the step
	auto step = [&](int sidx, bool valid_all, unsigned rowmask) {
		const long long rm = valid_all ? -1ll : -(long long)((rowmask >> rsub) & 1u);
		double d[NB], a[NB];
		double wv = 1.0;
		if (WEIGHTED) {
			wv = buf[(p + 1) * RS + 16 * sidx + rsub];
			if (!valid_all) wv = quad_mask(wv, rm);
		}
#pragma unroll
		for (int g = 0; g < NB; ++g) {
			double v;
			if (g < NB - 1) {
				v = buf[base_off + 16 * sidx + 4 * g * RS];
			} else {
				const double raw = buf[last_off + 16 * sidx];
				v = rd_last ? raw : fill_last;
			}
			double dev = v - first[g];
			if (!valid_all) dev = quad_mask(dev, rm);
			// CENTER: deviations from the group's first valid row (first = 0 for the ones); otherwise raw values, masked
			d[g] = CENTER ? dev : (valid_all ? v : quad_mask(v, rm));
			// constant-column predicate of ols.rs:76-87: the largest |x - x_first| per lane and column, tested once per group
			if (!SPEC && !(ANOFOX_QUAD_SKIP & 4)) dmax[g] = fmax(dmax[g], fabs(dev));
			a[g] = WEIGHTED ? wv * d[g] : d[g];
		}
		int t = 0;
#pragma unroll
		for (int g = 0; g < NB; ++g) {
#pragma unroll
			for (int h = g; h < NB; ++h) {
				if (ANOFOX_QUAD_SKIP & 1) acc[t] += a[g] + d[h];
				else acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[g], d[h], acc[t], 0, 0, 0);
				++t;
			}
		}
	};
	auto block = [&](int64_t blk, int64_t blk_next) {
		// row filter on this lane's rows (ols.rs:59-66, wls.rs:76-86)
		// (finiteness of a whole row as ONE number: z = sum 0 * v is NaN iff some v is not finite — an FMA per value instead
		// of a class test and a mask AND per value; the slots past the last column hold y again)
		bool ok[RL];
#pragma unroll
		for (int e = 0; e < RL; ++e) {
			double z = 0.0;
			if (!SPEC && !(ANOFOX_QUAD_SKIP & 2)) {
#pragma unroll
				for (int c = 0; c < NCOL; ++c) z = fma(0.0, reg[c][e], z);
			}
			ok[e] = (blk + RL * lane + e < hi) && (z == 0.0);
			if (WEIGHTED) { // the weight sits in slot p + 1, one of the last four (p + 2 <= 4 NB <= p + 5)
				double wv = 1.0;
#pragma unroll
				for (int c = NCOL - 4; c < NCOL; ++c) wv = (c == p + 1) ? reg[c][e] : wv;
				ok[e] = ok[e] && (wv > 0.0);
			}
		}
#pragma unroll
		for (int c = 0; c < NCOL; ++c) {
			if (c < ncol && !(ANOFOX_QUAD_SKIP & 8)) {
				if (RL == 2) *reinterpret_cast<quad_dbl2a *>(buf + c * RS + 2 * lane) = quad_dbl2a{reg[c][0], reg[c][RL - 1]};
				else buf[c * RS + lane] = reg[c][0];
			}
		}
		if (blk_next < hi) issue(blk_next);
		__builtin_amdgcn_wave_barrier();
		const unsigned long long v0 = __ballot(ok[0]), v1 = RL == 2 ? __ballot(ok[RL - 1]) : v0;
		if ((v0 | v1) != 0ull) {
			if (!have_first) { // the first valid row of the group: every lane fetches its columns' values of that row
				const int f0 = __ffsll((long long)v0) - 1, f1 = __ffsll((long long)v1) - 1;
				int fr;
				if (RL == 2) {
					const int r0 = v0 ? 2 * f0 : 1 << 20, r1 = v1 ? 2 * f1 + 1 : 1 << 20;
					fr = r0 < r1 ? r0 : r1;
				} else {
					fr = f0;
				}
#pragma unroll
				for (int g = 0; g < NB - 1; ++g) first[g] = buf[c4 * RS + 4 * g * RS + fr];
				first[NB - 1] = rd_last ? buf[(rd_last ? c_last : p) * RS + fr] : 0.0; // (the constants are not shifted)
				have_first = true;
			}
			if ((v0 & v1) == ~0ull) {
				cnt += BR;
				if (NB >= ANOFOX_QUAD_NOUNROLL_NB) {
#pragma unroll 1
					for (int sidx = 0; sidx < 4 * RL; ++sidx) step(sidx, true, 0xFFFFu);
				} else {
#pragma unroll 2
					for (int sidx = 0; sidx < ((ANOFOX_QUAD_SKIP & 16) ? 1 : 4 * RL); ++sidx) step(sidx, true, 0xFFFFu);
				}
			} else {
				cnt += __popcll(v0) + (RL == 2 ? __popcll(v1) : 0);
#pragma unroll 1
				for (int sidx = 0; sidx < 4 * RL; ++sidx) {
					unsigned rowmask;
					if (RL == 2) rowmask = quad_spread8((unsigned)(v0 >> (8 * sidx)) & 0xFFu) | (quad_spread8((unsigned)(v1 >> (8 * sidx)) & 0xFFu) << 1);
					else rowmask = (unsigned)(v0 >> (16 * sidx)) & 0xFFFFu;
					if (rowmask == 0u) continue; // wave-uniform
					step(sidx, false, rowmask);
				}
			}
		}
		__builtin_amdgcn_wave_barrier(); // the reads above before the next block's writes
	};
	if (lo < hi) issue(lo);
	for (int64_t blk = lo; blk < hi; blk += BR) block(blk, blk + BR);

	return quad_finish<NB, SPEC>(args, rec, lane, buf, acc, first, dmax, isx_prev, isx_last, cnt, lo, hi);
}

// ---- (r4) the speculative version on LDS-DMA, NB = 8, 9 (p = 27 .. 33) -------------------------------------------------
// The full version and the register-staged speculative one do not fit two waves per SIMD from NB = 8 on (45 accumulators + 36
// staged columns: the compiler spills the STAGED LOADS one by one, each behind its own vmcnt(0)).  Here no row passes through a
// register on its way in: `global_load_lds_dword` moves 256 bytes = 32 rows of one column per wave-instruction straight into the
// wave's slice (M0 = destination, a scalar base per column + the lane's constant 4-byte offset: no vector instruction at all),
// block k + 1 lands in one half of a 64-row ring while the steps of block k read the other half.  The instructions are inline asm
// (M0 is written in the statement that uses it), so the compiler neither counts nor waits for them: the loop waits with
// vmcnt(number of columns) — one block stays in flight — before it reads a half.  What is left per 32-row block: 4 NB - 1 DMA
// instructions, 2 x (NB ds_read_b64 + NB subtractions + NB (NB + 1) / 2 v_mfma_f64_4x4x4) and nothing else.
// doubles per column of a ring of RING 32-row blocks: 32 RING + 8 (= 8 mod 32: conflict-free ds_read_b64, see above)
__host__ __device__ constexpr int quad_dma_stride(int ring) { return 32 * ring + 8; }
__host__ __device__ constexpr int quad_dma_slice_doubles(int p, int ring) {
	const int data = (p + 1) * quad_dma_stride(ring);
	const int R = 4 * quad_blocks(p);
	const int image = R * R + 2 * R;
	return data > image ? data : image;
}

// RING = 2: block k + 1 lands while block k is read (one block in flight per wavefront).  RING = 3: blocks k + 1 and k + 2 are in
// flight — the bytes in flight are what the HBM latency under load is paid with (one block per wavefront, eight wavefronts per
// CU: 17.8 MB on the chip at p = 33, 3.8 us at the measured 4.7 TB/s) — at the price of a slice half as large again.
template <int NB, int RING>
__device__ __forceinline__ bool quad_spec_dma_rows(const WideArgs &args, int64_t lo, int64_t hi, double *rec, int lane, double *buf) {
	constexpr int NPAIR = NB * (NB + 1) / 2;
	constexpr int RS = quad_dma_stride(RING);
	const int p = args.p;
	const int k = lane >> 4, b = (lane >> 2) & 3, c4 = lane & 3;
	const int rsub = 4 * k + b;
	const int base_off = c4 * RS + rsub;
	const int c_last = 4 * (NB - 1) + c4;
	const bool rd_last = c_last <= p;
	const int last_off = (rd_last ? c_last : p) * RS + rsub;
	const double fill_last = c_last == p + 1 ? 1.0 : 0.0;
	const bool isx_prev = 4 * (NB - 2) + c4 < p, isx_last = c_last < p;
	const unsigned lds0 = lds_dma_address(buf);
	double acc[NPAIR];
#pragma unroll
	for (int t = 0; t < NPAIR; ++t) acc[t] = 0.0;
	double first[NB], dmax[NB];
#pragma unroll
	for (int g = 0; g < NB; ++g) first[g] = dmax[g] = 0.0;

	// one 32-row block into half `h` of the ring: the column pointers are re-read from the kernel arguments (lds_dma.h); a partial
	// block clamps the lanes' offsets to its last row (the rows past the end are masked by the steps)
	const lds_dma_table_t tab = lds_dma_table((unsigned)offsetof(WideArgs, x_table)); // (y sits behind the last feature: host_api.hip)
	auto dma = [&](int64_t blk, int h) {
		lds_dma_block<RS * 8, 4 * NB - 1>(tab, p + 1, blk, lds_dma_offsets(lane, hi - blk), lds0 + (unsigned)h * 256u);
	};
	auto step = [&](const double *hb, int s, bool valid_all, unsigned rowmask) {
		const long long rm = valid_all ? -1ll : -(long long)((rowmask >> rsub) & 1u);
		double d[NB];
#pragma unroll
		for (int g = 0; g < NB; ++g) {
			double v;
			if (g < NB - 1) {
				v = hb[base_off + 16 * s + 4 * g * RS];
			} else {
				const double raw = hb[last_off + 16 * s];
				v = rd_last ? raw : fill_last;
			}
			double dev = v - first[g];
			if (!valid_all) dev = quad_mask(dev, rm);
			d[g] = dev;
		}
		int t = 0;
#pragma unroll
		for (int g = 0; g < NB; ++g) {
#pragma unroll
			for (int h = g; h < NB; ++h) {
				acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(d[g], d[h], acc[t], 0, 0, 0);
				++t;
			}
		}
	};
	// all but the youngest block have landed (the loads retire in order; ncol = p + 1 of them per block, 4 NB - 4 .. 4 NB - 1)
	auto wait_but_one_block = [&]() {
		switch (p + 1 - (4 * NB - 4)) { // wave-uniform
		case 0: lds_dma_wait_but<4 * NB - 4>(); break;
		case 1: lds_dma_wait_but<4 * NB - 3>(); break;
		case 2: lds_dma_wait_but<4 * NB - 2>(); break;
		default: lds_dma_wait_but<4 * NB - 1>(); break;
		}
	};
	int h = 0;
	dma(lo, 0);
	if (RING == 3 && lo + 32 < hi) {
		dma(lo + 32, 1);
		wait_but_one_block();
	} else {
		lds_dma_wait_all();
	}
	__builtin_amdgcn_wave_barrier();
	{ // the shift: the group's first row (the constants are not shifted)
#pragma unroll
		for (int g = 0; g < NB - 1; ++g) first[g] = buf[c4 * RS + 4 * g * RS];
		first[NB - 1] = rd_last ? buf[(rd_last ? c_last : p) * RS] : 0.0;
	}
	// full blocks in a loop without a branch around the matrix instructions, the partial last block after it (with the choice
	// inside the loop the compiler copies the accumulators at every join)
	int64_t blk = lo;
	if constexpr (RING == 2) {
		for (; blk + 32 <= hi; blk += 32, h ^= 1) {
			if (blk + 32 < hi) dma(blk + 32, h ^ 1); // lands while this block's steps run
			__builtin_amdgcn_wave_barrier();
			const double *hb = buf + 32 * h;
			step(hb, 0, true, 0xFFFFu);
			step(hb, 1, true, 0xFFFFu);
			__builtin_amdgcn_wave_barrier(); // the reads of this half before the DMA that refills it (next trip but one)
			lds_dma_wait_all();              // the next block has landed (it had this block's steps to do so)
		}
	} else {
		for (; blk + 32 <= hi; blk += 32, h = h == 2 ? 0 : h + 1) {
			const bool more = blk + 64 < hi;
			if (more) dma(blk + 64, h == 0 ? 2 : h - 1); // the third of the ring that the block before this one was read from
			__builtin_amdgcn_wave_barrier();
			const double *hb = buf + 32 * h;
			step(hb, 0, true, 0xFFFFu);
			step(hb, 1, true, 0xFFFFu);
			__builtin_amdgcn_wave_barrier();
			if (more) wait_but_one_block(); // block + 32 has landed, block + 64 may still be under way
			else lds_dma_wait_all();
		}
	}
	if (blk < hi) {
		const int64_t left = hi - blk;
		const unsigned m32 = (1u << (unsigned)left) - 1u; // 1 <= left < 32
		const double *hb = buf + 32 * h;
		step(hb, 0, false, m32 & 0xFFFFu);
		if (left > 16) step(hb, 1, false, m32 >> 16);
	}
	lds_dma_wait_all();
	return quad_finish<NB, true>(args, rec, lane, buf, acc, first, dmax, isx_prev, isx_last, 0, lo, hi);
}

// ---- (r4) the same on `global_load_lds_dwordx4`: 128 rows of a column per wave-instruction ---------------------------------
// What the dword form cannot give is the REQUEST SIZE: 256 bytes per column and instruction, 70 000 column streams in flight on
// the chip.  The streaming kernels of this library that reach 6 TB/s ask for 1 KB per column and instruction (accumulate_narrow:
// 16-byte loads, 128-row tiles; its 512-byte variant was 25 % slower, profiles/r03_hbm_read_variants.txt), the ones that ask for
// 512 bytes reach 5.2-5.4 (accumulate_quad up to p = 26), the ones that ask for 256 stay at 4.1-5.0 (this kernel's dword form,
// accumulate_wide) — and a third block in flight did not help the dword form (ANOFOX_QUAD_RING=3: slower at every width), so
// bytes in flight are not what is missing.  Here a block is 128 rows, ONE buffer (stride 136 = 8 mod 32 doubles per column: the
// slice of a 34-column group is 37 KB, four wavefronts per CU): load the block, wait, run its eight 16-row steps, next block —
// nothing overlaps inside a wavefront, the other three of the CU's wavefronts cover the wait.  The last rows of a group (fewer
// than 128) come in 32-row pieces through the dword form, whose clamped lane offsets touch nothing behind the group.
template <int NB>
__device__ __forceinline__ bool quad_spec_dma_rows_x4(const WideArgs &args, int64_t lo, int64_t hi, double *rec, int lane, double *buf) {
	constexpr int NPAIR = NB * (NB + 1) / 2;
	constexpr int BR = 128;
	constexpr int RS = quad_dma_stride(4);
	const int p = args.p;
	const int k = lane >> 4, b = (lane >> 2) & 3, c4 = lane & 3;
	const int rsub = 4 * k + b;
	const int base_off = c4 * RS + rsub;
	const int c_last = 4 * (NB - 1) + c4;
	const bool rd_last = c_last <= p;
	const int last_off = (rd_last ? c_last : p) * RS + rsub;
	const double fill_last = c_last == p + 1 ? 1.0 : 0.0;
	const bool isx_prev = 4 * (NB - 2) + c4 < p, isx_last = c_last < p;
	const unsigned lds0 = lds_dma_address(buf);
	double acc[NPAIR];
#pragma unroll
	for (int t = 0; t < NPAIR; ++t) acc[t] = 0.0;
	double first[NB], dmax[NB];
#pragma unroll
	for (int g = 0; g < NB; ++g) first[g] = dmax[g] = 0.0;
	const lds_dma_table_t tab = lds_dma_table((unsigned)offsetof(WideArgs, x_table));
	auto step = [&](int s, bool valid_all, unsigned rowmask) {
		const long long rm = valid_all ? -1ll : -(long long)((rowmask >> rsub) & 1u);
		double d[NB];
#pragma unroll
		for (int g = 0; g < NB; ++g) {
			double v;
			if (g < NB - 1) {
				v = buf[base_off + 16 * s + 4 * g * RS];
			} else {
				const double raw = buf[last_off + 16 * s];
				v = rd_last ? raw : fill_last;
			}
			double dev = v - first[g];
			if (!valid_all) dev = quad_mask(dev, rm);
			d[g] = dev;
		}
		int t = 0;
#pragma unroll
		for (int g = 0; g < NB; ++g) {
#pragma unroll
			for (int h = g; h < NB; ++h) {
				acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(d[g], d[h], acc[t], 0, 0, 0);
				++t;
			}
		}
	};
	auto fetch_first = [&]() { // the shift: the group's first row (the constants are not shifted)
#pragma unroll
		for (int g = 0; g < NB - 1; ++g) first[g] = buf[c4 * RS + 4 * g * RS];
		first[NB - 1] = rd_last ? buf[(rd_last ? c_last : p) * RS] : 0.0;
	};
	int64_t blk = lo;
	if (lo + BR <= hi) {
		// full blocks; the first one is peeled for the shift (a branch around the matrix instructions inside the loop makes the
		// compiler copy the accumulators at the join)
		lds_dma_block<RS * 8, 4 * NB - 1, true>(tab, p + 1, blk, (unsigned)lane * 16u, lds0);
		lds_dma_wait_all();
		__builtin_amdgcn_wave_barrier();
		fetch_first();
#pragma unroll
		for (int s = 0; s < BR / 16; ++s) step(s, true, 0xFFFFu);
		__builtin_amdgcn_wave_barrier(); // the reads of this block before the DMA that overwrites it
		for (blk += BR; blk + BR <= hi; blk += BR) {
			lds_dma_block<RS * 8, 4 * NB - 1, true>(tab, p + 1, blk, (unsigned)lane * 16u, lds0);
			lds_dma_wait_all();
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int s = 0; s < BR / 16; ++s) step(s, true, 0xFFFFu);
			__builtin_amdgcn_wave_barrier();
		}
	}
	if (blk < hi) { // the last 1 .. 127 rows: 32-row pieces through the dword form
		const int left = (int)(hi - blk);
		for (int sub = 0; 32 * sub < left; ++sub)
			lds_dma_block<RS * 8, 4 * NB - 1, false>(tab, p + 1, blk + 32 * sub, lds_dma_offsets(lane, hi - (blk + 32 * sub)), lds0 + (unsigned)sub * 256u);
		lds_dma_wait_all();
		__builtin_amdgcn_wave_barrier();
		if (blk == lo) fetch_first();
		for (int s = 0; 16 * s < left; ++s) {
			const int rows = left - 16 * s;
			step(s, false, rows >= 16 ? 0xFFFFu : (1u << (unsigned)rows) - 1u);
		}
	}
	lds_dma_wait_all();
	return quad_finish<NB, true>(args, rec, lane, buf, acc, first, dmax, isx_prev, isx_last, 0, lo, hi);
}

// ---- (r4) the fine ring: 16-row sub-blocks of EIGHT columns per `global_load_lds_dwordx4` -------------------------------------
// What the 32-row ring above keeps in flight per wavefront is one block — half its slice — and never more than the 63 vector-memory
// instructions a wavefront may have outstanding, 256 bytes each; with the slices of 36 .. 65 columns a CU holds six, then four
// wavefronts, and blocks in flight / memory latency is what those widths run at (profiles/r04_widths_n1000.md: 16.1 MB on the chip at
// p = 40 -> 4.2 TB/s, 13.4 MB at p = 50 -> 3.1 TB/s: 3.8-4.3 us either way).  Here eight lanes carry one column: an instruction moves
// 16 rows x 8 columns = 1 KB, each lane with its own 64-bit address (column pointer of its group of eight, two rows per lane), a
// sub-block is ceil((p + 1) / 8) instructions, and the slice is a ring of RING sub-blocks of which RING - 1 are in flight while one
// is read: three quarters of the slice at RING = 4 instead of one half, in an eighth of the instructions.
// LDS of a sub-block: chunk i (1 KB) = columns 8 i .. 8 i + 7, 128 bytes each (the hardware's M0 + 16 * lane), and inside a column
// the row PAIR pr sits at position pr ^ 4 * ((column >> 1) & 1) — the lane picks its source rows accordingly — so that the 32 lanes
// of a half-wave of the fragment read (ds_read_b64: 4 columns x 8 rows) touch every bank once.
// (The instruction's immediate offset cannot carry the row step of consecutive sub-blocks: the hardware adds it to the LDS address
// as well as to the source — the first version of this loader scattered its sub-blocks over the slice that way.)
// The last 1 .. 15 rows of a group: a lane whose pair would end behind the group loads the pair (hi - 2, hi - 1) instead; the lanes
// of the masked step that read row `left - 1` of an odd remainder find it in the second half of that pair.  (A group of one row
// goes to the full version.)
__host__ __device__ constexpr int quad_fine_chunks(int p) { return (p + 1 + 7) / 8; }
__host__ __device__ constexpr int quad_fine_slice_doubles(int p, int ring) {
	const int data = ring * quad_fine_chunks(p) * 128;
	const int R = 4 * quad_blocks(p);
	const int image = R * R + 2 * R;
	return data > image ? data : image;
}

template <int OFF> // 16 bytes from each lane's own address (+ OFF) to dst + 16 * lane
__device__ __forceinline__ void lds_dma_x4_lanes(const double *src, unsigned dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep)
	             : "v"(src), "s"(dst), "n"(OFF)
	             : "memory");
}

template <int NI, int RING>
__device__ __forceinline__ void quad_fine_wait(int younger) { // all but the `younger` most recent sub-blocks have landed (wave-uniform)
	static_assert((RING - 1) * NI <= 63, "vmcnt");
	switch (younger) {
	case 0: lds_dma_wait_but<0>(); break;
	case 1: lds_dma_wait_but<NI>(); break;
	case 2: lds_dma_wait_but<(RING > 2 ? 2 : 0) * NI>(); break;
	case 3: lds_dma_wait_but<(RING > 3 ? 3 : 0) * NI>(); break;
	case 4: lds_dma_wait_but<(RING > 4 ? 4 : 0) * NI>(); break;
	case 5: lds_dma_wait_but<(RING > 5 ? 5 : 0) * NI>(); break;
	case 6: lds_dma_wait_but<(RING > 6 ? 6 : 0) * NI>(); break;
	default: lds_dma_wait_but<(RING > 7 ? 7 : 0) * NI>(); break;
	}
}

template <int NB, int RING>
__device__ __forceinline__ bool quad_spec_fine_rows(const WideArgs &args, int64_t lo, int64_t hi, double *rec, int lane, double *buf) {
	static_assert(RING >= 2 && RING <= 8, "ring depth");
	constexpr int NPAIR = NB * (NB + 1) / 2;
	constexpr int NIMAX = (4 * NB - 1 + 7) / 8; // chunks of the widest design with NB blocks; the narrowest has NIMAX or NIMAX - 1
	if (hi - lo < 2) return false;
	const int p = args.p;
	const int ni = (p + 1 + 7) >> 3;
	const int k = lane >> 4, b = (lane >> 2) & 3, c4 = lane & 3;
	const int rsub = 4 * k + b;
	// the fragment read of column 4 g + c4, row rsub of a sub-block: 64 g + rd_off doubles
	const int rd_off = 16 * c4 + 2 * ((rsub >> 1) ^ (4 * (c4 >> 1))) + (rsub & 1);
	const int c_last = 4 * (NB - 1) + c4;
	const bool rd_last = c_last <= p;
	const int cl = rd_last ? c_last : p;
	const int last_off = (cl >> 3) * 128 + (cl & 7) * 16 + 2 * ((rsub >> 1) ^ (4 * ((cl >> 1) & 1))) + (rsub & 1);
	const double fill_last = c_last == p + 1 ? 1.0 : 0.0;
	const bool isx_prev = 4 * (NB - 2) + c4 < p, isx_last = c_last < p;
	const unsigned lds0 = lds_dma_address(buf);
	const unsigned sb_bytes = (unsigned)ni * 1024u;
	double acc[NPAIR];
#pragma unroll
	for (int t = 0; t < NPAIR; ++t) acc[t] = 0.0;
	double first[NB], dmax[NB];
#pragma unroll
	for (int g = 0; g < NB; ++g) first[g] = dmax[g] = 0.0;

	// the lane's sources: column 8 i + (lane >> 3) (columns past y repeat y: the same addresses, no traffic), row pair `pr`
	const int s8 = lane >> 3, pr = (lane & 7) ^ (4 * ((s8 >> 1) & 1));
	const lds_dma_table_t tab = lds_dma_table((unsigned)offsetof(WideArgs, x_table)); // (y sits behind the last feature: host_api.hip)
	const double *ptr[NIMAX];
#pragma unroll
	for (int i = 0; i < NIMAX; ++i) {
		int c = 8 * i + s8;
		c = c < p ? c : p;
		ptr[i] = tab[c] + lo + 2 * pr;
	}
	const int64_t n = hi - lo;
	const int nfull = (int)(n >> 4), left = (int)(n & 15), nsb = nfull + (left ? 1 : 0);
	// (the remainder) doubles to step back so that the lane's pair ends inside the group
	const int tail_back = 2 * pr + 1 >= left ? 2 * pr - (left - 2) : 0;

	// sub-block `sub` into its slot; the lanes' pointers move on by 16 rows.  (The instruction's immediate offset cannot carry the
	// row step: the hardware adds it to the LDS address as well.)
	auto dma = [&](int sub) {
		const unsigned dst = lds0 + (unsigned)(sub % RING) * sb_bytes;
		if (sub >= nfull) { // (wave-uniform) the remainder
#pragma unroll
			for (int i = 0; i < NIMAX; ++i)
				if (i < ni) lds_dma_x4_lanes<0>(ptr[i] - tail_back, dst + (unsigned)i * 1024u);
			return;
		}
#pragma unroll
		for (int i = 0; i < NIMAX; ++i) {
			if (i < ni) lds_dma_x4_lanes<0>(ptr[i], dst + (unsigned)i * 1024u);
			ptr[i] += 16;
		}
	};
	auto wait = [&](int younger) {
		if (ni == NIMAX) quad_fine_wait<NIMAX, RING>(younger);
		else quad_fine_wait<(NIMAX > 1 ? NIMAX - 1 : 1), RING>(younger);
	};
	auto step = [&](const double *hb, bool valid_all, long long rm, int fix) {
		double d[NB];
#pragma unroll
		for (int g = 0; g < NB; ++g) {
			double v;
			if (g < NB - 1) {
				v = hb[64 * g + rd_off + fix];
			} else {
				const double raw = hb[last_off + fix];
				v = rd_last ? raw : fill_last;
			}
			double dev = v - first[g];
			if (!valid_all) dev = quad_mask(dev, rm);
			d[g] = dev;
		}
		int t = 0;
#pragma unroll
		for (int g = 0; g < NB; ++g) {
#pragma unroll
			for (int h = g; h < NB; ++h) {
				acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(d[g], d[h], acc[t], 0, 0, 0);
				++t;
			}
		}
	};

	int issued = 0;
	for (; issued < RING - 1 && issued < nsb; ++issued) dma(issued);
	wait(issued - 1);
	__builtin_amdgcn_wave_barrier();
	{ // the shift: the group's first row (the constants are not shifted)
		const int f_off = 16 * c4 + 2 * (4 * (c4 >> 1));
#pragma unroll
		for (int g = 0; g < NB - 1; ++g) first[g] = buf[64 * g + f_off];
		first[NB - 1] = rd_last ? buf[(cl >> 3) * 128 + (cl & 7) * 16 + 2 * (4 * ((cl >> 1) & 1))] : 0.0;
	}
	for (int j = 0; j < nfull; ++j) {
		if (issued < nsb) { // (the slot read in the trip before this one)
			dma(issued);
			++issued;
		}
		wait(issued - 1 - j);
		__builtin_amdgcn_wave_barrier();
		step(buf + (size_t)(j % RING) * (size_t)(ni * 128), true, -1ll, 0);
		__builtin_amdgcn_wave_barrier(); // the reads of this slot before the DMA that refills it
	}
	if (left) {
		lds_dma_wait_all();
		__builtin_amdgcn_wave_barrier();
		const long long rm = rsub < left ? -1ll : 0ll;
		const int fix = ((left & 1) && rsub == left - 1) ? 1 : 0;
		step(buf + (size_t)(nfull % RING) * (size_t)(ni * 128), false, rm, fix);
	}
	lds_dma_wait_all();
	__builtin_amdgcn_wave_barrier();
	return quad_finish<NB, true>(args, rec, lane, buf, acc, first, dmax, isx_prev, isx_last, 0, lo, hi);
}

// MODE 0: the full version on every group; 1: the speculative version, give-ups (and empty groups) listed for the full one —
// the list borrows the refine queue and its counter word kWideRedoCounter, as accumulate_wide's does; 2: the full version on
// the listed groups (launched with the batch's grid: one scalar load and out for the wavefronts beyond the list).
template <int NB, bool WEIGHTED, bool CENTER, int RL, int WPS, int MODE = 0> // WPS: waves per SIMD the register budget is cut for
__global__ __launch_bounds__(256, WPS) void accumulate_quad_kernel(WideArgs args) {
	extern __shared__ double quad_lds[];
	const int lane = threadIdx.x & 63;
	// (four wavefronts per workgroup; the LDS-DMA kernel of the widest designs is launched with two, so that three workgroups' slices fit a CU)
	int64_t gl = (int64_t)blockIdx.x * (int64_t)(blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	if (MODE == 2) {
		if (gl >= args.refine_count[kWideRedoCounter]) return;
		gl = args.refine_list[gl];
	} else if (gl >= args.n_groups) {
		return;
	}
	const int T = wide_tiles(args.p);
	const int64_t lo = args.row_offsets[args.group_base + gl];
	const int64_t hi = group_row_end(args, args.group_base + gl);
	if (MODE != 2 && args.seg_table && hi - lo > args.seg_rows) {
		// (three column tiles, p = 33, 34: the host sized the table for accumulate_wide's workgroup-per-segment kernel)
		if (wide_register_big_group(args, gl, lo, hi, T, lane, T <= 2 ? kSegMaxBig : kWideSegMaxBig, T <= 2 ? kSegMaxSegments : kWideSegMaxSegments)) return;
	}
	double *rec = args.moments + gl * (int64_t)wide_record_len(T);
	if constexpr (MODE == 4) { // the speculative version on the fine LDS-DMA ring (the template's RL parameter is the ring depth)
		double *slice = quad_lds + (threadIdx.x >> 6) * quad_fine_slice_doubles(args.p, RL);
		if (hi > lo && quad_spec_fine_rows<NB, RL>(args, lo, hi, rec, lane, slice)) return;
		if (lane == 0) args.refine_list[atomicAdd(args.refine_count + kWideRedoCounter, 1)] = (int32_t)gl;
		return;
	}
	if constexpr (MODE == 3) { // the speculative version on LDS-DMA (its own slice size)
		// (MODE 3: the template's RL parameter is the ring depth)
		double *slice = quad_lds + (threadIdx.x >> 6) * quad_dma_slice_doubles(args.p, RL);
		if constexpr (RL == 4) { // (128-row blocks on the dwordx4 form)
			if (hi > lo && quad_spec_dma_rows_x4<NB>(args, lo, hi, rec, lane, slice)) return;
		} else {
			if (hi > lo && quad_spec_dma_rows<NB, RL>(args, lo, hi, rec, lane, slice)) return;
		}
		if (lane == 0) args.refine_list[atomicAdd(args.refine_count + kWideRedoCounter, 1)] = (int32_t)gl;
		return;
	}
	double *slice = quad_lds + (threadIdx.x >> 6) * quad_slice_doubles(args.p, WEIGHTED, RL);
	if constexpr (MODE == 1) {
		if (hi > lo && quad_accumulate_rows<NB, WEIGHTED, CENTER, RL, true>(args, lo, hi, rec, lane, slice)) return;
		if (lane == 0) args.refine_list[atomicAdd(args.refine_count + kWideRedoCounter, 1)] = (int32_t)gl;
	} else {
		quad_accumulate_rows<NB, WEIGHTED, CENTER, RL>(args, lo, hi, rec, lane, slice);
	}
}

template <int NB, int RL, int WPS>
hipError_t launch_quad_nb(const WideArgs &a, hipStream_t stream) {
	const bool weighted = a.model == ANOFOX_HIP_MODEL_WLS;
	const bool center = a.fit_intercept != 0;
	const dim3 grid((unsigned)((a.n_groups + 3) / 4)), block(256);
	const size_t lds_bytes = 4 * (size_t)quad_slice_doubles(a.p, weighted, RL) * sizeof(double);
	static const bool attr_set = [] {
#define ANOFOX_QUAD_ATTR(W, C, M) \
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_quad_kernel<NB, W, C, RL, WPS, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
		ANOFOX_QUAD_ATTR(true, true, 0); ANOFOX_QUAD_ATTR(true, false, 0); ANOFOX_QUAD_ATTR(false, true, 0); ANOFOX_QUAD_ATTR(false, false, 0);
		ANOFOX_QUAD_ATTR(false, true, 1); ANOFOX_QUAD_ATTR(false, true, 2);
#undef ANOFOX_QUAD_ATTR
		return true;
	}();
	(void)attr_set;
	// the unweighted fit with an intercept first runs the speculative kernel, then the full version on what it listed (the
	// counter is zeroed by the caller; ANOFOX_WIDE_FAST=0 / ANOFOX_QUAD_SPEC=0: the full version only)
	static const bool spec_on = !(getenv("ANOFOX_QUAD_SPEC") && atoi(getenv("ANOFOX_QUAD_SPEC")) == 0);
	// (NB = 3, 4 — p = 9 .. 14 — stay with the full version: measured 2-5 % faster there on two boxes, profiles/r04_widths_n1000.txt,
	// r04_final_widths_n1000.txt; the speculative one pays a second launch and wins only from p = 15 on)
	const bool spec = spec_on && !a.no_fast_path && !weighted && center && NB >= 5;
	if (weighted) {
		if (center) hipLaunchKernelGGL((accumulate_quad_kernel<NB, true, true, RL, WPS>), grid, block, lds_bytes, stream, a);
		else hipLaunchKernelGGL((accumulate_quad_kernel<NB, true, false, RL, WPS>), grid, block, lds_bytes, stream, a);
	} else {
		if (!center) hipLaunchKernelGGL((accumulate_quad_kernel<NB, false, false, RL, WPS>), grid, block, lds_bytes, stream, a);
		else if (spec) hipLaunchKernelGGL((accumulate_quad_kernel<NB, false, true, RL, WPS, 1>), grid, block, lds_bytes, stream, a);
		else hipLaunchKernelGGL((accumulate_quad_kernel<NB, false, true, RL, WPS>), grid, block, lds_bytes, stream, a);
	}
	hipError_t rc = hipGetLastError();
	if (rc != hipSuccess) return rc;
	// very large groups were registered for row splitting: accumulate_mid's segment kernel takes them (idle otherwise)
	if (a.seg_table && (rc = launch_accumulate_mid_segments(a, stream)) != hipSuccess) return rc;
	if (spec) hipLaunchKernelGGL((accumulate_quad_kernel<NB, false, true, RL, WPS, 2>), grid, block, lds_bytes, stream, a);
	return hipGetLastError();
}

// NB = 8, 9 (p = 27 .. 34): only the speculative kernel exists here (the full version's registers do not fit two waves per
// SIMD, see below); its give-ups go to accumulate_mid (p <= 32) or accumulate_wide (p = 33, 34: three column tiles, whose
// segment table also takes this kernel's very large groups).
template <int NB, int WPS, int RING>
hipError_t launch_quad_spec_only(const WideArgs &a, hipStream_t stream) {
	constexpr int RL = RING;
	// eight wavefronts per CU while their slices fit its 160 KB of LDS (RING = 2, p <= 34: 35 columns x 576 bytes x 8 = 161 280
	// bytes) in workgroups of four; beyond, the workgroup size that leaves the fewest bytes of the CU's LDS unused
	const size_t slice_bytes = (size_t)quad_dma_slice_doubles(a.p, RING) * sizeof(double);
	const int fit = (int)(((size_t)160 * 1024) / slice_bytes);
	const int waves = fit >= 8 ? 4 : (fit % 2 == 0 ? 2 : 1);
	const dim3 grid((unsigned)((a.n_groups + waves - 1) / waves)), block(64 * waves);
	const size_t lds_bytes = (size_t)waves * slice_bytes;
	static const bool attr_set = [] {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_quad_kernel<NB, false, true, RL, WPS, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		return true;
	}();
	(void)attr_set;
	hipLaunchKernelGGL((accumulate_quad_kernel<NB, false, true, RL, WPS, 3>), grid, block, lds_bytes, stream, a);
	hipError_t rc = hipGetLastError();
	if (rc != hipSuccess) return rc;
	if (wide_tiles(a.p) <= 2) {
		if (a.seg_table && (rc = launch_accumulate_mid_segments(a, stream)) != hipSuccess) return rc;
		return launch_accumulate_mid_redo(a, stream);
	}
	return launch_accumulate_wide_followup(a, stream);
}

// the fine ring (measurement switch ANOFOX_QUAD_FINE=3 | 4 = ring depth; OFF by default): NB = 8 .. 17, p = 27 .. 64.
// Measured on one box (scripts/quad_fine_ab.sh, 100 000 / 50 000 groups x 1000 rows, TB/s of the kernel, default | ring 4 | ring 3):
//   p = 33: 4.08 | 3.63 | 3.68      p = 40: 3.79 | 3.71 | 3.82      p = 50: 3.13 | 2.35 | 2.51      p = 64 (accumulate_wide<4>): 3.24 | 2.10 | 2.11
// Correct (the GPU suite passes with it on), and slower at every width although it keeps half as much again in flight per wavefront
// in an eighth of the instructions — so neither the bytes in flight per wavefront nor the instruction count is what holds these
// widths.  (The contiguous run per column and request halves to 128 bytes here; that is not it either: accumulate_wide staged with
// 256-byte instead of 128-byte runs measures the same, profiles/r04_wide_run256_ab.txt.)  NB >= 15 also spill here (20-296 bytes:
// 512 registers hold 120-153 accumulators and little else).
template <int NB, int WPS, int RING>
hipError_t launch_quad_spec_fine(const WideArgs &a, hipStream_t stream) {
	const size_t slice_bytes = (size_t)quad_fine_slice_doubles(a.p, RING) * sizeof(double);
	int fit = (int)(((size_t)160 * 1024) / slice_bytes);
	if (fit > 4 * WPS) fit = 4 * WPS;
	const int waves = fit >= 8 ? 4 : (fit % 4 == 0 ? 4 : (fit % 2 == 0 ? 2 : 1));
	const dim3 grid((unsigned)((a.n_groups + waves - 1) / waves)), block(64 * waves);
	const size_t lds_bytes = (size_t)waves * slice_bytes;
	static const bool attr_set = [] {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_quad_kernel<NB, false, true, RING, WPS, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		return true;
	}();
	(void)attr_set;
	hipLaunchKernelGGL((accumulate_quad_kernel<NB, false, true, RING, WPS, 4>), grid, block, lds_bytes, stream, a);
	hipError_t rc = hipGetLastError();
	if (rc != hipSuccess) return rc;
	if (wide_tiles(a.p) <= 2) {
		if (a.seg_table && (rc = launch_accumulate_mid_segments(a, stream)) != hipSuccess) return rc;
		return launch_accumulate_mid_redo(a, stream);
	}
	return launch_accumulate_wide_followup(a, stream);
}

template <int NB, int WPS>
hipError_t launch_quad_spec_fine_ring(const WideArgs &a, int ring, hipStream_t stream) {
	if (ring == 3) return launch_quad_spec_fine<NB, WPS, 3>(a, stream);
	return launch_quad_spec_fine<NB, WPS, 4>(a, stream);
}

int quad_fine_ring() {
	static const int ring = getenv("ANOFOX_QUAD_FINE") ? atoi(getenv("ANOFOX_QUAD_FINE")) : 0;
	return ring;
}

} // namespace

// The full version at p = 27 .. 32 (NB = 8, 9) stays with accumulate_mid: 36 / 45 accumulators + the staged block + the running
// maxima of the constant-column test do not fit the 256 registers of two waves per SIMD (168-456 bytes of spills: 2.9-3.0 TB/s
// against 4.3).  (Also measured and not kept: TWO wavefronts per group sharing the slice — each loads every other column and
// owns every other block pair, the block written to LDS centred and masked, two barriers per 64-row block: correct, 164-188
// registers, but 3.5-4.1 TB/s against accumulate_mid's 4.2-4.4 at p = 27 .. 32; and this kernel with the registers of ONE wave
// per SIMD: no spills, 3.5-4.0 TB/s.  profiles/r03_quad.txt.)  (r4) The SPECULATIVE version needs neither the maxima nor the
// masks and takes p = 27 .. 34 for the unweighted fit with an intercept.
bool accumulate_quad_supports(int p, bool weighted, bool center, bool no_fast_path) {
	static const bool spec_on = !(getenv("ANOFOX_QUAD_SPEC") && atoi(getenv("ANOFOX_QUAD_SPEC")) == 0);
	// (NB = 10, 11: p = 35 .. 42 — 55 / 66 accumulators, six wavefronts per CU; measured against accumulate_wide in docs/MEASUREMENTS.md)
	// (r4, NB = 12, 13: 224 / 252 registers, no scratch.  Against accumulate_wide on one box, scripts/quad50_ab.sh: level at p = 43, 44,
	// 6-10 % behind at p = 46, 48 (three column tiles are well filled there), 10 % ahead at p = 49, 50 — where four column tiles pad 51
	// columns to 64: 160 matrix cycles per row against 91.  So: up to 42, and 49, 50.  ANOFOX_QUAD_SPEC_MAXP=50 takes 43 .. 48 as well.)
	static const int spec_max_p = getenv("ANOFOX_QUAD_SPEC_MAXP") ? atoi(getenv("ANOFOX_QUAD_SPEC_MAXP")) : 0;
	if (p > kNarrowMaxP && p <= 26) return true;
	if (quad_fine_ring() > 0 && p > 26 && p <= 64) return spec_on && !weighted && center && !no_fast_path;
	const bool in_range = spec_max_p > 0 ? p <= spec_max_p : (p <= 42 || p == 49 || p == 50);
	return p > 26 && p <= 50 && in_range && spec_on && !weighted && center && !no_fast_path;
}

hipError_t launch_accumulate_quad(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	if (!accumulate_quad_supports(a.p, a.model == ANOFOX_HIP_MODEL_WLS, a.fit_intercept != 0, a.no_fast_path != 0)) return hipErrorInvalidValue;
	// rows per lane and block, by measurement (profiles/r03_quad.txt): 128-row blocks for p <= 11 and p = 15..17, 64-row
	// blocks elsewhere (from p = 18 the larger slice would leave one workgroup per CU).  A register budget for three waves
	// per SIMD was measured slower for every NB >= 4 (spills).  ANOFOX_QUAD_RL=1/2 forces one.
	static const int env_rl = getenv("ANOFOX_QUAD_RL") ? atoi(getenv("ANOFOX_QUAD_RL")) : 0;
	const int nb = quad_blocks(a.p);
	static const int ring = getenv("ANOFOX_QUAD_RING") ? atoi(getenv("ANOFOX_QUAD_RING")) : 2; // (measurement switch: 3 = three 32-row blocks, 4 = 128-row blocks on dwordx4)
	int rl = (a.p <= 11 || (a.p >= 15 && a.p <= 17)) ? 2 : 1;
	if (env_rl == 1 || (env_rl == 2 && a.p <= 18)) rl = env_rl;
	if (const int fine = quad_fine_ring(); fine > 0 && nb >= 8) {
		switch (nb) {
		case 8: return launch_quad_spec_fine_ring<8, 2>(a, fine, stream);
		case 9: return launch_quad_spec_fine_ring<9, 2>(a, fine, stream);
		case 10: return launch_quad_spec_fine_ring<10, 2>(a, fine, stream);
		case 11: return launch_quad_spec_fine_ring<11, 2>(a, fine, stream);
		case 12: return launch_quad_spec_fine_ring<12, 2>(a, fine, stream);
		case 13: return launch_quad_spec_fine_ring<13, 2>(a, fine, stream);
		case 14: return launch_quad_spec_fine_ring<14, 1>(a, fine, stream);
		case 15: return launch_quad_spec_fine_ring<15, 1>(a, fine, stream);
		case 16: return launch_quad_spec_fine_ring<16, 1>(a, fine, stream);
		case 17: return launch_quad_spec_fine_ring<17, 1>(a, fine, stream);
		default: return hipErrorInvalidValue;
		}
	}
	switch (nb) { // p = 9, 10 | 11..14 | 15..18 | 19..22 | 23..26 | 27..30 | 31..34
	case 3: return rl == 2 ? launch_quad_nb<3, 2, 3>(a, stream) : launch_quad_nb<3, 1, 3>(a, stream);
	case 4: return rl == 2 ? launch_quad_nb<4, 2, 2>(a, stream) : launch_quad_nb<4, 1, 2>(a, stream);
	case 5: return rl == 2 ? launch_quad_nb<5, 2, 2>(a, stream) : launch_quad_nb<5, 1, 2>(a, stream);
	case 6: return launch_quad_nb<6, 1, 2>(a, stream);
	case 7: return launch_quad_nb<7, 1, 2>(a, stream);
	case 8: return ring == 4 ? launch_quad_spec_only<8, 2, 4>(a, stream) : ring == 3 ? launch_quad_spec_only<8, 2, 3>(a, stream) : launch_quad_spec_only<8, 2, 2>(a, stream);
	case 9: return ring == 4 ? launch_quad_spec_only<9, 2, 4>(a, stream) : ring == 3 ? launch_quad_spec_only<9, 2, 3>(a, stream) : launch_quad_spec_only<9, 2, 2>(a, stream);
	case 10: return ring == 4 ? launch_quad_spec_only<10, 2, 4>(a, stream) : ring == 3 ? launch_quad_spec_only<10, 2, 3>(a, stream) : launch_quad_spec_only<10, 2, 2>(a, stream);
	case 11: return ring == 4 ? launch_quad_spec_only<11, 2, 4>(a, stream) : ring == 3 ? launch_quad_spec_only<11, 2, 3>(a, stream) : launch_quad_spec_only<11, 2, 2>(a, stream);
	case 12: return launch_quad_spec_only<12, 2, 2>(a, stream);
	case 13: return launch_quad_spec_only<13, 2, 2>(a, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace anofox
