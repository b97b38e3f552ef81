// accumulate_wide_impl.h — (compiled by accumulate_wide.hip and accumulate_wide_t{5,6,7,8}.hip, one tile count each: the
// four wave copies of two kernels per width take minutes to compile in one translation unit)
// moment accumulation for wide designs (8 < p <= 128) on the FP64 matrix cores.
//
// Same role as accumulate_narrow.hip (it replaces the reference's row buffering + dense decomposition,
// src/aggregate_functions/ols_aggregate.cpp:120-186,249-296 and crates/anofox-stats-core/src/models/ols.rs:59-87,
// 149-161), for designs whose (p+1)(p+2)/2 moments no longer fit a lane's registers.  The X'WX block is a
// symmetric rank-n update and goes to v_mfma_f64_16x16x4_f64; X'Wy, the column sums, y'Wy and the
// constant-column / finite-row predicates stay on the VALU (they are O(p) per row).
//
// Mapping: one 256-thread workgroup (4 wavefronts) per group.
//   * A chunk is 32 (16 for p > 96) consecutive rows of every column (x_1..x_p, y, [w]).  The four waves load it coalesced
//     (each load instruction: 8 columns x 128 contiguous bytes) one chunk ahead into registers and write it to
//     a double-buffered LDS image laid out [column][18 doubles] (conflict-free for the fragment reads below); the
//     image holds the values already shifted by the group's first valid row, with the rows that do not take part
//     zeroed (the staging lanes do both; the chunk that contains the first valid row, and chunks with an invalid
//     row, are repaired in place once the row masks of all four waves are known).
//   * A slab is 4 rows.  For the 16-column block I, lane l of a wave reads the fragment element
//     (row 4t + (l>>4), column 16I + (l&15)) with one ds_read_b64; the same register is the MFMA's A operand
//     for tile row I and the B operand for tile column I:  M[16I+i][16J+j] += sum_k w_k d[k][16I+i] d[k][16J+j].
//   * The T(T+1)/2 upper-triangular 16x16 tiles are dealt round-robin to the four waves (9 tiles = 72
//     accumulator registers each at p = 128), so every wave issues the same number of MFMAs per slab.
//   * Column block I is "owned" by wave I % 4, which also accumulates sum w d, sum w d dy and the
//     constant-column flags for those 16 columns; wave 0 accumulates the y moments.
//
// Roofline: FP64 MFMA (matrix-core) bound for p >= ~48: 2*256*4 flop per instruction, T(T+1)/2 instructions
// per 4 rows; HBM traffic 8(p+1) B per row is read once.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "lds_dma.h"

namespace anofox {

typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));
typedef double dbl4 __attribute__((ext_vector_type(4)));
// column base pointers are parked in LDS as integers and turned back into *global* pointers, so that the
// loads through them are global_load (a generic pointer read from memory would give flat_load, whose
// s_waitcnt vmcnt(0) lgkmcnt(0) serialises every load behind the previous one)
typedef const double __attribute__((address_space(1))) *gptr_t;
typedef const dbl2u __attribute__((address_space(1))) *gptr2_t;

// Diagnostic build only (-DANOFOX_SOLVE_STAMPS, csrc/Makefile target `diag`): wave 0 of workgroup 0 sums s_memtime
// deltas of the phases of its chunk loop: [0] row masks known  [1] next chunk's loads issued (+ repairs)  [2] the chunk's
// slabs (fragment reads + MFMAs + side sums)  [3] next chunk's loads landed  [4] staged into LDS  [5] barrier passed
// [6] chunks  [7] whole group.
#ifdef ANOFOX_SOLVE_STAMPS
static __device__ unsigned long long g_acc_stamps[8];
// (r4) milestones of one group's life, s_memtime since the function's entry: [0] setup + barrier  [1] first row known  [2] chunk 0 staged
// [3] steady trips done  [4] chunk loop done  [5] split tiles collected  [6] speculation checked  [7] record written.  The workgroup in
// the middle of the grid, so that the chip is loaded.
static __device__ unsigned long long g_acc_marks[8];
#define ACC_MARK_DECL const bool mk_on = blockIdx.x == gridDim.x / 2 && WAVE == 0 && (threadIdx.x & 63) == 0; const unsigned long long mk_begin = __builtin_amdgcn_s_memtime()
#define ACC_MARK(k) do { if (mk_on) g_acc_marks[k] = __builtin_amdgcn_s_memtime() - mk_begin; } while (0)
#define ACC_STAMP_DECL unsigned long long st_t = 0, st_acc[6] = {0, 0, 0, 0, 0, 0}, st_n = 0; const bool st_on = blockIdx.x == gridDim.x / 2 && WAVE == 0; const unsigned long long st_begin = __builtin_amdgcn_s_memtime()
#define ACC_STAMP_START() do { if (st_on) st_t = __builtin_amdgcn_s_memtime(); } while (0)
#define ACC_STAMP(k) do { if (st_on) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_t; st_t = now_; } } while (0)
#define ACC_STAMP_FLUSH() do { if (st_on && threadIdx.x == 0) { for (int k_ = 0; k_ < 6; ++k_) g_acc_stamps[k_] = st_acc[k_]; g_acc_stamps[6] = st_n; g_acc_stamps[7] = __builtin_amdgcn_s_memtime() - st_begin; } } while (0)
#else
#define ACC_MARK_DECL do { } while (0)
#define ACC_MARK(k) do { } while (0)
#define ACC_STAMP_DECL do { } while (0)
#define ACC_STAMP_START() do { } while (0)
#define ACC_STAMP(k) do { } while (0)
#define ACC_STAMP_FLUSH() do { } while (0)
#endif

namespace {

#ifndef ANOFOX_WIDE_WAVES
#define ANOFOX_WIDE_WAVES 4
#endif
constexpr int kWaves = ANOFOX_WIDE_WAVES;   // wavefronts per workgroup (4 or 8: a translation unit's choice)
constexpr int kThreads = 64 * kWaves;
constexpr int kWavesPerSimd = kWaves / 2; // two workgroups per CU either way
static_assert(kWaves == 4 || kWaves == 8, "4 or 8 wavefronts per workgroup");
// (kWideRedoCounter — the word of the refine counter block that counts the speculative kernels' give-ups — lives in common.h)
constexpr int kWideFastMinT = 3;    // (narrower designs go through accumulate_mid.hip)

__device__ __forceinline__ double readlane_d(double v, int src) {
	return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
	                        __builtin_amdgcn_readlane(__double2loint(v), src));
}

__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

// Row masks of a chunk.  Bit layout (what the staging ballots produce without any bit shuffling): lane (colsub, rp)
// stages rows 2 rp, 2 rp + 1, 16 + 2 rp, 17 + 2 rp; byte k of the mask holds those four row classes, bit rp within it.
__device__ __forceinline__ int row_bit(int row) { return ((row >> 4) << 4) | ((row & 1) << 3) | ((row >> 1) & 7); }
template <int CH>
__device__ __forceinline__ unsigned range_mask(int64_t left) { // rows [0, left) of a chunk (wave-uniform, partial chunks only)
	if (left >= CH) return CH == 32 ? 0xffffffffu : 0xffffu;
	unsigned m = 0;
	for (int r = 0; r < (int)left; ++r) m |= 1u << row_bit(r);
	return m;
}
__device__ __forceinline__ int first_row_of_mask(unsigned m) { // lowest row whose bit is set (m != 0)
	int best = 64;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const unsigned byte = (m >> (8 * k)) & 0xffu;
		const int r = 16 * (k >> 1) + (k & 1) + 2 * (__ffs((int)byte) - 1);
		best = (byte != 0u && r < best) ? r : best;
	}
	return best;
}

// which kernels stage 16 rows per barrier instead of 32: those that spill with the staging registers of 32 rows
// (p = 128 without intercept 14.2 -> 13.7 ms, p = 112 weighted 12.4 -> 12.0 ms per 8192 x 4096 rows)
#ifndef ANOFOX_WIDE_SHORT_CHUNK
#define ANOFOX_WIDE_SHORT_CHUNK(T, weighted, center) (((T) == 8 && ((weighted) || !(center))) || ((T) == 7 && (weighted)))
#endif
template <int T>
struct WideCfg {
	static constexpr int NT = T * (T + 1) / 2;             // upper-triangular tiles
	// (r4) Tiles to wavefronts.  Dealt round-robin, 6 tiles (T = 3) give two wavefronts two tiles and two wavefronts one — and
	// every chunk ends in a barrier, so the workgroup runs at the pace of the two-tile wavefronts; wavefront w of every workgroup
	// sits on SIMD w, so SIMDs 0 and 1 carry twice the matrix work of SIMDs 2 and 3 (the 60-65 % "matrix pipe busy" of the
	// round-4 counters is the average of 85 % and 42 %).  Neither a lighter staging side nor LDS-DMA changes that (measured:
	// ANOFOX_WIDE_DMA=1, -3 %).  So: the first 4 F = 4 floor(NT / 4) tiles go round-robin as before, and the R = NT mod 4 that
	// are left are SPLIT ALONG THE ROWS — a tile in halves between two wavefronts (even / odd slabs of every chunk), or in
	// quarters between all four — each wavefront keeping a partial sum that the tile's owner collects once, at the end of the
	// group.  Busiest wavefront per chunk: T = 3: 2 -> 1.5 tiles, T = 4: 3 -> 2.5, T = 5: 4 -> 3.75, T = 6: 6 -> 5.25 (T = 7, 8: R = 0).
#ifdef ANOFOX_WIDE_NOSPLIT // (measurement build: the round-robin deal of rounds 1-3, csrc/Makefile target `nosplit`)
	static constexpr bool kSplit = false;
#else
	static constexpr bool kSplit = kWaves == 4;
#endif
	static constexpr int F = kSplit ? NT / 4 : 0, R = kSplit ? NT % 4 : 0;
	static constexpr int TPW = kSplit ? F + (R > 0 ? 1 : 0) + (R == 3 ? 1 : 0) : (NT + kWaves - 1) / kWaves; // accumulator tiles per wave
	// remaining tile j = tile - 4 F: split in halves between wavefronts 2 j, 2 j + 1 (R = 2; R = 3: j < 2) or in quarters (R = 1; R = 3: j = 2)
	static constexpr bool quartered(int tile) { return kSplit && tile >= 4 * F && (R == 1 || (R == 3 && tile - 4 * F == 2)); }
	static constexpr bool halved(int tile) { return kSplit && tile >= 4 * F && !quartered(tile); }
	static constexpr int acc_index(int tile) { return !kSplit ? tile / kWaves : (tile < 4 * F ? tile / 4 : (R == 3 && tile - 4 * F == 2 ? F + 1 : F)); }
	// the wavefront that holds the tile at the end of the group (checks it, writes it to the record)
	static constexpr int owner(int tile) { return !kSplit ? tile % kWaves : (tile < 4 * F ? tile % 4 : (quartered(tile) ? 0 : 2 * (tile - 4 * F))); }
	// does wavefront `wave` issue the tile's matrix instruction in slab `slab` of a chunk
	static constexpr bool works(int tile, int wave, int slab) {
		if (!kSplit) return tile % kWaves == wave;
		if (tile < 4 * F) return tile % 4 == wave;
		if (quartered(tile)) return (slab & 3) == wave;
		return (wave >> 1) == tile - 4 * F && (slab & 1) == (wave & 1);
	}
	static constexpr int OWN = (T + kWaves - 1) / kWaves;  // column blocks owned per wave
	// rows staged per barrier: the per-chunk costs that are latency, not work (row masks through LDS, issuing the next
	// chunk's loads, the barrier) are as long as the MFMAs of 16 rows at T = 8, so chunks are 32 rows for every width
	// (except where 32 rows of staging registers spill: ANOFOX_WIDE_SHORT_CHUNK)
	// (r4) chunks the staging loads run ahead of the chunk being multiplied.  A chunk of 3 or 4 column tiles is 0.4-0.7 us of
	// matrix work; with the loads of chunk c + 1 issued behind the first slab of chunk c and needed behind its last slabs, they have
	// less than that to cross the memory system — a wavefront then waits on them in every chunk.  The staging registers of such a
	// chunk are few (16-24 per lane), so narrow designs keep two chunks in flight; from 5 tiles on a chunk is long enough.
	// Measured on one box (scripts/wide_depth_ab.sh, builds of `make variant`), kernel ms at 50 000 x 1000 x p = 48 / 56 / 64:
	// rounds 1-3 (round-robin deal, one chunk ahead) 5.59 / 7.81 / 8.43; split tiles 5.43 / 7.60 / 8.45; + two chunks ahead
	// 5.50 / 7.52 / 8.13; three 5.41 / 7.92 / 8.51; four 5.47 / 8.01 / 8.61 (registers).  Two it is: +2 ... 4 %, not the
	// +20 ... 33 % the arithmetic of the tile deal promised — the chunk takes its ~4 us whatever is done to its parts.
	// (The speculative version only: the full one — row masks, repairs, weights — has no registers to spare: 224-476 bytes of
	// scratch per lane with three chunks in flight.)
#ifndef ANOFOX_WIDE_DEPTH
#define ANOFOX_WIDE_DEPTH(T) ((T) <= 4 ? 2 : 1)
#endif
#ifndef ANOFOX_WIDE_RUN256
#define ANOFOX_WIDE_RUN256(T) 0
#endif
// (r4) the ends of a group in the speculative version with two chunks of loads in flight (ANOFOX_WIDE_TAIL2=0: as before; see kTail2)
#ifndef ANOFOX_WIDE_TAIL2
#define ANOFOX_WIDE_TAIL2 0
#endif
#ifndef ANOFOX_WIDE_STAGGER
#define ANOFOX_WIDE_STAGGER 0
#endif
// what-if builds of the speculative version's steady state (never shipped: the results are wrong; `make variant`): bit 0 = one v_fma_f64
// in place of every matrix instruction, 1 = no global loads (the staging registers keep the first chunk), 2 = no barrier per chunk,
// 3 = the fragments of a chunk's first slab are used for all of its slabs (one eighth of the LDS reads), 4 = the staged chunk is not written to LDS
#ifndef ANOFOX_WIDE_SKIP
#define ANOFOX_WIDE_SKIP 0
#endif
	static constexpr int depth(bool fast) { return kWaves == 4 && fast ? ANOFOX_WIDE_DEPTH(T) : 1; }
	static constexpr int chunk_rows(bool weighted, bool center = true) { return ANOFOX_WIDE_SHORT_CHUNK(T, weighted, center) ? 16 : 32; }
	static constexpr int stride(bool weighted, bool center = true) { return chunk_rows(weighted, center) + 2; } // doubles per column in the LDS image (+2 pad: conflict-free b64 reads)
};

// One wave's share of a chunk: the slabs of MFMAs + the VALU side sums.  WAVE is a compile-time constant so
// that the tile list unrolls into straight-line MFMAs.  `img` is the chunk's LDS image, already in the form the
// matrix cores consume: shifted by the group's first valid row (CENTER) and with every row that does not take part
// (non-finite value, w <= 0, past the end of the group) zeroed in ALL columns by the staging side — so a fragment
// goes from ds_read_b64 straight into the MFMA and the only VALU work left per slab is the side sums of the column
// blocks this wave owns.  (Shifting and masking inside the slab loop cost ~5 VALU instructions per block in each of
// the four waves.)
// `between(t)` is called once per slab, after the slab's MFMAs: the staging work of the NEXT chunk is dealt over the
// slabs of this one (iteration() below), so that a wave has no phase without MFMAs in flight.
// RAW ((r4) the LDS-DMA staging of the speculative version): the image holds the rows as they lie in memory and the shift by the
// group's first row (`fsub` per column block of this lane, `fy`) is applied to the fragment on its way to the matrix cores.
template <int T, int WAVE, bool WEIGHTED, bool CENTER, bool FAST, bool RAW = false, int CH = 0, typename Between>
__device__ __forceinline__ void compute_chunk(const double *img, const double *firstcol, int ycol, int lane, unsigned rowmask,
                                              dbl4 (&acc)[WideCfg<T>::TPW], double (&sx)[WideCfg<T>::OWN],
                                              double (&sxy)[WideCfg<T>::OWN], double (&dmax)[WideCfg<T>::OWN], unsigned &ncmask,
                                              double &sy, double &syy, double &sw, Between &&between, const double *fsub = nullptr, double fy = 0.0) {
	// (CH: rows per chunk when the caller's image is not the configuration's — the 64-row chunks of the LDS-DMA staging)
	constexpr int kChunkRows = CH ? CH : WideCfg<T>::chunk_rows(WEIGHTED, CENTER), kLdsStride = CH ? CH + 2 : WideCfg<T>::stride(WEIGHTED, CENTER), OWN = WideCfg<T>::OWN;
	const int k = lane >> 4;
	const int i = lane & 15;
	// without an intercept the image holds raw values; the constant-column test still compares with the first valid row
	double fown[OWN];
#pragma unroll
	for (int o = 0; o < OWN; ++o) fown[o] = (!CENTER && WAVE + kWaves * o < T) ? firstcol[16 * (WAVE + kWaves * o) + i] : 0.0;
	// Software pipeline over the chunk's slabs: the fragment reads of slab t + 1 are issued BEFORE the MFMAs of slab t
	// (two register sets, the loop is fully unrolled), so that a wave goes from one slab's MFMAs straight into the
	// next slab's — with the reads at the top of each slab the matrix pipe idles for an LDS round trip per slab
	// whenever the SIMD's other wave is not in its own MFMA phase (measured: 115 cycles per MFMA per SIMD against
	// 64-68 for back-to-back issue with two waves).
	constexpr int NS = kChunkRows / 4;
	double d[2][T], dy[2], w[2];
	auto read_slab = [&](int t, int s) {
		const int row = 4 * t + k;
#pragma unroll
		for (int I = 0; I < T; ++I) d[s][I] = RAW ? img[(16 * I + i) * kLdsStride + row] - fsub[I] : img[(16 * I + i) * kLdsStride + row];
		dy[s] = RAW ? img[ycol * kLdsStride + row] - fy : img[ycol * kLdsStride + row];
		w[s] = WEIGHTED ? img[(ycol + 1) * kLdsStride + row] : 1.0;
	};
	read_slab(0, 0);
#pragma unroll
	for (int t = 0; t < NS; ++t) {
		const int s = t & 1;
		const int row = 4 * t + k;
#if ANOFOX_WIDE_SKIP & 8
		if (FAST) {
			if (t + 1 < NS) {
#pragma unroll
				for (int I = 0; I < T; ++I) d[s ^ 1][I] = d[s][I];
				dy[s ^ 1] = dy[s];
				w[s ^ 1] = w[s];
			}
		} else
#endif
		if (t + 1 < NS) read_slab(t + 1, s ^ 1);
		double a[T];
#pragma unroll
		for (int I = 0; I < T; ++I) a[I] = WEIGHTED ? w[s] * d[s][I] : d[s][I];

		int tile = 0;
#pragma unroll
		for (int I = 0; I < T; ++I) {
#pragma unroll
			for (int J = I; J < T; ++J) {
				if (WideCfg<T>::works(tile, WAVE, t)) {
#if ANOFOX_WIDE_SKIP & 1
					if (FAST) acc[WideCfg<T>::acc_index(tile)][0] = fma(a[I], d[s][J], acc[WideCfg<T>::acc_index(tile)][0]);
					else
#endif
					acc[WideCfg<T>::acc_index(tile)] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], d[s][J], acc[WideCfg<T>::acc_index(tile)], 0, 0, 0);
				}
				++tile;
			}
		}
#pragma unroll
		for (int I = 0; I < T; ++I) {
			if (I % kWaves == WAVE) {
				// constant-column predicate of ols.rs:76-87: |x - x_first| < 1e-10 on every valid row (a zeroed row of a
				// shifted image gives 0; a raw image needs the row's validity)
				if (FAST) {
					// (the constant-column flags come out of the diagonal of the moments at the end)
				} else if (CENTER) {
					// shifted image: |x - x_first| itself, 0 on rows that do not take part — keep the largest (one v_max_f64;
					// the compare / select / or per slab it replaces sat in the MFMA waves' instruction stream)
					dmax[I / kWaves] = fmax(dmax[I / kWaves], fabs(d[s][I]));
				} else {
					const double dev = d[s][I] - fown[I / kWaves];
					const bool moved = !(fabs(dev) < 1e-10) && ((rowmask >> row_bit(row)) & 1u);
					ncmask |= moved ? (1u << I) : 0u;
				}
				sx[I / kWaves] += a[I];
				sxy[I / kWaves] = fma(a[I], dy[s], sxy[I / kWaves]);
			}
		}
		if (WAVE == 0) {
			const double wdy = WEIGHTED ? w[s] * dy[s] : dy[s];
			sy += wdy;
			syy = fma(wdy, dy[s], syy);
			if (WEIGHTED) sw += w[s]; // unweighted: the row count, taken from the masks
		}
		between(t);
	}
}

// The rows [lo, hi) of one group — or of one segment of a very large group, then with the group's first valid
// row handed in (`forced_first`: x per column, y at index 16 T) — into one moment record at `rec`, by a workgroup.
// WAVE (this wavefront's number within the workgroup) is a template parameter of the WHOLE row loop, not only of
// compute_chunk: with a per-chunk `switch (wave)` the four arms get different register assignments for the 72
// accumulator registers and the compiler copies all of them in and out around every chunk (2 x 36 v_mov_b64 behind
// the last MFMA of each 16-row chunk, ~20 % of the chunk's matrix-core time).  The barriers inside are executed the
// same number of times by every wave, from four copies of the loop.
//
// FAST (unweighted, with intercept, whole groups): the speculative version.  It takes every row of the group as
// valid and the group's first row as the shift, so the staging side neither tests values nor builds row masks, the
// chunk loop reads no masks and repairs nothing, and the constant-column test is not run per row (ablation of the
// full version at p = 128, 8192 x 4096 rows: staging stores 19 %, side sums 9 %, mask read 3.5 %, barrier 3 % of
// the kernel; fragment reads + MFMAs alone run at the 68 TFLOP/s two waves per SIMD can issue).  At the end it
// checks what it assumed: a non-finite value anywhere in the group makes the moments non-finite; a column is
// constant (|x - x_first| < 1e-10 on every row, ols.rs:76-87) if sum d^2 < 1e-20 and not constant if
// sum d^2 >= n 1e-20.  Anything else — a NaN / inf somewhere, a column in between — returns false, and the caller
// runs the full version on the group.  Returns true when the record at `rec` is complete.
template <int T, int WAVE, bool WEIGHTED, bool CENTER, bool FAST = false, int DMA = 0> // DMA: rows per chunk of the LDS-DMA staging (0: registers)
__device__ __forceinline__ bool wide_accumulate_rows_wave(const WideArgs &args, int64_t lo, int64_t hi, double *rec,
                                                          const double *forced_first) {
	static_assert(!FAST || (!WEIGHTED && CENTER), "the speculative version exists for the unweighted fit with an intercept");
	static_assert(DMA == 0 || FAST, "the LDS-DMA staging exists for the speculative version");
	using Cfg = WideCfg<T>;
	constexpr int kChunkRows = DMA ? DMA : Cfg::chunk_rows(WEIGHTED, CENTER), kLdsStride = DMA ? DMA + 2 : Cfg::stride(WEIGHTED, CENTER);
	constexpr bool kWideChunk = kChunkRows == 32;
	// (r4) kRun: the speculative version's lanes stage 32 rows of FOUR columns per load instruction (16 lanes x 16 bytes = a run of 256
	// bytes per column and request) instead of 16 rows of eight columns (128 bytes): registers `a` of a slot hold its first four columns,
	// `b` its last four, both the same 32 rows.  (The full version keeps the 16-row classes its row masks are built from.)
	constexpr bool kRun = FAST && DMA == 0 && kWideChunk && ANOFOX_WIDE_RUN256(T);
	// (r4) kTail2 (measurement build -DANOFOX_WIDE_TAIL2=1; OFF).  A load takes ~9 000 cycles to come back on the loaded chip and a chunk of
	// 3-4 column tiles ~3 500 (milestones of a group's life, profiles/r04_wide_group_life.md): at 1000 rows per group a quarter of a
	// workgroup's life goes into its two ends — 9 500 cycles for the first row (the shift) before chunk 0 is even asked for, and ~7 800
	// per chunk in the last four chunks, which the generic iteration loads ONE chunk ahead into the registers it has just stored.  With
	// kTail2 chunks 0 and 1 are requested first and the shift is taken from chunk 0's registers (the lane that holds row 0 of the
	// column), and the last chunks alternate between the two register sets of the steady state.  Correct (GPU suite), and NOT faster:
	// T = 3 level, T = 4 3-4 % slower (scripts/wide_tail_ab.sh) — a workgroup's idle ends are filled by the CU's other workgroups
	// already; the CU is bound by what its SIMDs issue (matrix + vector instructions add up to ~88 % of the steady chunk).
	constexpr bool kTail2 = FAST && DMA == 0 && Cfg::depth(FAST) == 2 && ANOFOX_WIDE_TAIL2;
	constexpr unsigned kFullMask = kWideChunk ? 0xffffffffu : 0xffffu;
	constexpr int P16 = 16 * T;
	const int p = args.p;
	const int ncol = p + 1 + (WEIGHTED ? 1 : 0);
	const int lane = threadIdx.x & 63;
	constexpr int wave = WAVE;
	const int64_t nrows = hi - lo;
	ACC_MARK_DECL;

	extern __shared__ double lds[];
	// layout: image[2][ncol_pad][STRIDE] | colbase[ncol_pad] (as pointers) | firstcol[ncol_pad] | rowmask partials [2][4]
	// image columns: x_0..x_{p-1} | zeros up to 16T | y | w | padding to a multiple of 8
	const int ncol_pad = wide_ncol_pad(p, WEIGHTED);
	const int ycol = P16;
	double *image = lds;
	unsigned long long *colbase = reinterpret_cast<unsigned long long *>(lds + 2 * ncol_pad * kLdsStride);
	double *firstcol = reinterpret_cast<double *>(colbase + ncol_pad); // value at the group's first valid row, by image column
	unsigned *maskslot = reinterpret_cast<unsigned *>(firstcol + ncol_pad);

	// source column c (x_0.., y, w) is staged by load slot c; slot -> image column: x in place, y/w after 16T
	for (int c = threadIdx.x; c < ncol_pad; c += kThreads) {
		const double *b = args.y; // unused slots read y again and are dropped at the store
		if (c < p) b = args.x_table[c];
		else if (WEIGHTED && c == p + 1) b = args.w;
		colbase[c] = reinterpret_cast<unsigned long long>(b + lo);
		double f = 0.0;
		if (forced_first) f = c < P16 ? forced_first[c] : (c == ycol ? forced_first[P16] : 0.0);
		firstcol[c] = f;
	}
	// zero the padding columns p .. 16T-1 of both buffers once; nothing writes them afterwards
	for (int idx = threadIdx.x; idx < 2 * (P16 - p) * kLdsStride; idx += kThreads) {
		const int bufi = idx / ((P16 - p) * kLdsStride);
		const int rem = idx - bufi * (P16 - p) * kLdsStride;
		image[bufi * ncol_pad * kLdsStride + p * kLdsStride + rem] = 0.0;
	}
	__syncthreads();
	ACC_MARK(0);

	// staging assignment: load instruction q of this wave covers columns 8*(wave + 4q) .. +7; lane -> (col, row pair).
	// Everything a lane needs to know about its load slots lives in registers (column pointer, LDS destination, flags):
	// with the pointers parked in LDS every load of a chunk waited for its own ds_read round trip, ~1000 cycles per
	// chunk of pure latency (phase stamps, scripts/dbg_acc_stamps.py).
	constexpr int kMaxLoads = (P16 + 2 + 7) / 8 / kWaves + 1;
	const int colsub = kRun ? lane >> 4 : lane >> 3;
	const int rp = kRun ? lane & 15 : lane & 7;
	const int n_loads_total = (ncol + 7) / 8; // load slots: 8 source columns each
	gptr_t colp[kMaxLoads]; // this lane's column of slot q, at the group's first row + 2 rp
	int dcol[kMaxLoads];    // element offset of (image column, row 2 rp) within an image
	gptr_t colpB[kRun ? kMaxLoads : 1]; // (kRun) the same for the lane's column of the slot's second half
	int dcolB[kRun ? kMaxLoads : 1];
	unsigned actbits = 0;   // bit q: the column exists (slots beyond ncol read y again and store into a spare column)
	unsigned wbits = 0;     // bit q: the column is the weight column
#pragma unroll
	for (int q = 0; q < kMaxLoads; ++q) {
		const int src = 8 * (wave + kWaves * q) + colsub;
		const bool active = src < ncol;
		int col = src < p ? src : ycol + (src - p); // x in place, y / w after 16 T
		if (!active) col = ycol + 2 + (colsub % 6); // spare columns 16T+2 .. 16T+7: nothing reads them
		dcol[q] = col * kLdsStride + 2 * rp;
		colp[q] = reinterpret_cast<gptr_t>(colbase[src < ncol_pad ? src : ncol_pad - 1]) + 2 * rp;
		if constexpr (kRun) {
			const int srcB = src + 4;
			int colB = srcB < p ? srcB : ycol + (srcB - p);
			if (srcB >= ncol) colB = ycol + 2 + ((colsub + 4) % 6);
			dcolB[q] = colB * kLdsStride + 2 * rp;
			colpB[q] = reinterpret_cast<gptr_t>(colbase[srcB < ncol_pad ? srcB : ncol_pad - 1]) + 2 * rp;
		}
		actbits |= active ? (1u << q) : 0u;
		wbits |= (WEIGHTED && src == p + 1) ? (1u << q) : 0u;
	}

	dbl4 acc[Cfg::TPW];
#pragma unroll
	for (int t = 0; t < Cfg::TPW; ++t) acc[t] = (dbl4){0.0, 0.0, 0.0, 0.0};
	double sx[Cfg::OWN], sxy[Cfg::OWN], dmax[Cfg::OWN];
#pragma unroll
	for (int o = 0; o < Cfg::OWN; ++o) sx[o] = sxy[o] = dmax[o] = 0.0;
	unsigned ncmask = 0;
	double sy = 0.0, syy = 0.0, sw = 0.0;
	bool have_first = forced_first != nullptr; // wave-uniform
	int cnt = 0;
	// the shift the staging side applies to the columns THIS lane stages (0 until the first valid row is known, 0 for
	// the weight column and when there is no intercept)
	double fq[kMaxLoads];
	double fqB[kRun ? kMaxLoads : 1]; // (kRun: forced_first does not come with the speculative version)
#pragma unroll
	for (int q = 0; q < (kRun ? kMaxLoads : 1); ++q) fqB[q] = 0.0;
#pragma unroll
	for (int q = 0; q < kMaxLoads; ++q) {
		fq[q] = 0.0;
		const int src = 8 * (wave + kWaves * q) + colsub;
		if (CENTER && forced_first && src <= p) fq[q] = src < p ? forced_first[src] : forced_first[P16];
	}

	if (FAST && DMA == 0 && !kTail2) { // the shift is the group's first row (nrows > 0: the caller's condition)
#pragma unroll
		for (int q = 0; q < kMaxLoads; ++q) {
			const int src = 8 * (wave + kWaves * q) + colsub;
			if (wave + kWaves * q < n_loads_total && src <= p) {
				const double f = colp[q][-2 * rp];
				fq[q] = f;
				if (rp == 0) firstcol[src < p ? src : ycol] = f;
			}
			if constexpr (kRun) {
				const int srcB = src + 4;
				if (wave + kWaves * q < n_loads_total && srcB <= p) {
					const double f = colpB[q][-2 * rp];
					fqB[q] = f;
					if (rp == 0) firstcol[srcB < p ? srcB : ycol] = f;
				}
			}
		}
		have_first = true;
	}
	// (the shifts may still be in flight as loads the compiler keeps track of: have it wait for them HERE — an empty asm
	// that reads them — and not with a vmcnt(0) at their first use inside the steady loop, where it would also wait
	// for the loop's own loads)
#pragma unroll
	for (int q = 0; q < kMaxLoads; ++q) asm volatile("" : "+v"(fq[q]));
#pragma unroll
	for (int q = 0; q < (kRun ? kMaxLoads : 0); ++q) asm volatile("" : "+v"(fqB[q]));
	ACC_MARK(1);
	const int64_t n_chunks = (nrows + kChunkRows - 1) / kChunkRows;
	// staging registers of one chunk: rows 2 rp, 2 rp + 1 (v0, v1) and 16 + 2 rp, 17 + 2 rp (v2, v3) of this lane's columns
	struct Stage {
		dbl2u a[kMaxLoads], b[kMaxLoads]; // a = rows 2 rp, 2 rp + 1; b = rows 16 + 2 rp, 17 + 2 rp (each one 16-byte load)
	};
	// issue the loads of load slot q of one chunk (global -> registers); consumed by stage_store_piece
	auto stage_load_piece = [&](int64_t chunk, Stage &sg, int q) {
		const int64_t c0 = chunk * kChunkRows;              // first row of the chunk within the group
		const bool full = c0 + kChunkRows <= nrows;         // wave-uniform: every row of the chunk exists
		{
			sg.a[q] = (dbl2u){0.0, 0.0};
			sg.b[q] = (dbl2u){0.0, 0.0};
			if (wave + kWaves * q < n_loads_total) { // wave-uniform
				const gptr_t b = colp[q] + c0;
				if constexpr (kRun) {
					const gptr_t b2 = colpB[q] + c0;
					if (full) {
						sg.a[q] = *reinterpret_cast<gptr2_t>(b);
						sg.b[q] = *reinterpret_cast<gptr2_t>(b2);
					} else {
						const int64_t r0 = c0 + 2 * rp;
						if (r0 < nrows) sg.a[q].x = b[0], sg.b[q].x = b2[0];
						if (r0 + 1 < nrows) sg.a[q].y = b[1], sg.b[q].y = b2[1];
					}
				} else if (full) {
					sg.a[q] = *reinterpret_cast<gptr2_t>(b);
					if (kWideChunk) sg.b[q] = *reinterpret_cast<gptr2_t>(b + 16);
				} else {
					const int64_t r0 = c0 + 2 * rp;
					if (r0 < nrows) sg.a[q].x = b[0];
					if (r0 + 1 < nrows) sg.a[q].y = b[1];
					if (kWideChunk && r0 + 16 < nrows) sg.b[q].x = b[16];
					if (kWideChunk && r0 + 17 < nrows) sg.b[q].y = b[17];
				}
			}
		}
	};
	auto stage_load = [&](int64_t chunk, Stage &sg) {
#pragma unroll
		for (int q = 0; q < kMaxLoads; ++q) stage_load_piece(chunk, sg, q);
	};
	// The same without a single branch, for the steady state of the chunk loop (full chunks only): slots that do not
	// exist load y and are stored into a spare column, like the columns that do not exist.  With the wave-uniform
	// branches of stage_load_piece in the loop the compiler cannot count the loads in flight and puts s_waitcnt
	// vmcnt(0) in front of every use.  It also loses count across the loop's back edge, so the steady loop keeps no
	// load in flight from one iteration to the next: all of a chunk's loads go out behind slab 0 and are consumed
	// behind the last slabs of the same iteration, each behind an exact vmcnt.  (Loads kept in flight across
	// iterations, with the loads and waits as volatile asm naming their own count, measured the same at p = 128 —
	// and broke as soon as the compiler had a reason to copy a register whose load had not landed.)
	auto stage_load_piece_steady = [&](int64_t chunk, Stage &sg, int q) {
		const gptr_t b = colp[q] + chunk * kChunkRows;
#if ANOFOX_WIDE_SKIP & 2
		if (FAST) return;
#endif
		sg.a[q] = *reinterpret_cast<gptr2_t>(b);
		if constexpr (kRun) sg.b[q] = *reinterpret_cast<gptr2_t>(colpB[q] + chunk * kChunkRows);
		else if (kWideChunk) sg.b[q] = *reinterpret_cast<gptr2_t>(b + 16);
	};

	// registers -> LDS image `buf` (shifted by fq; rows past the end of the group as zeros), plus this wave's partial
	// row-validity mask (ols.rs:59-66, wls.rs:76-86).  Straight-line code: lanes whose column does not exist store into
	// a spare column and count as valid.
	struct StoreState { // what the pieces of one stage_store share
		double z0, z1, z2, z3;
		bool wbad0, wbad1, wbad2, wbad3;
	};
	auto stage_store_begin = [&](StoreState &ss) {
		ss.z0 = ss.z1 = ss.z2 = ss.z3 = 0.0;
		ss.wbad0 = ss.wbad1 = ss.wbad2 = ss.wbad3 = false;
	};
	// load slot q of the chunk
	auto stage_store_piece = [&](int64_t chunk, int buf, const Stage &sg, StoreState &ss, int q) {
		double *img = image + buf * ncol_pad * kLdsStride;
		const int64_t left = nrows - chunk * kChunkRows; // rows of the group in this chunk (wave-uniform)
		// A row is valid when every value is finite (ols.rs:59-66): z_k = sum_q 0 * v_k[q] is NaN exactly when one of the
		// lane's values of row class k is not finite — one FMA per value and one compare per row class instead of a class
		// test and two scalar mask updates per value.  (Lanes of columns that do not exist hold y values: harmless.)
		if (wave + kWaves * q < n_loads_total) { // wave-uniform
			if (!FAST) {
				ss.z0 = fma(sg.a[q].x, 0.0, ss.z0);
				ss.z1 = fma(sg.a[q].y, 0.0, ss.z1);
				if (kWideChunk) {
					ss.z2 = fma(sg.b[q].x, 0.0, ss.z2);
					ss.z3 = fma(sg.b[q].y, 0.0, ss.z3);
				}
			}
			if (WEIGHTED) { // wls.rs:76-86: w > 0
				const bool isw = (wbits >> q) & 1u;
				ss.wbad0 = ss.wbad0 || (isw && !(sg.a[q].x > 0.0));
				ss.wbad1 = ss.wbad1 || (isw && !(sg.a[q].y > 0.0));
				ss.wbad2 = ss.wbad2 || (isw && !(sg.b[q].x > 0.0));
				ss.wbad3 = ss.wbad3 || (isw && !(sg.b[q].y > 0.0));
			}
			double *dst = img + dcol[q];
			if constexpr (kRun) {
				double *dstB = img + dcolB[q];
				const bool in0 = 2 * rp < left, in1 = 2 * rp + 1 < left;
				dst[0] = in0 ? sg.a[q].x - fq[q] : 0.0;
				dst[1] = in1 ? sg.a[q].y - fq[q] : 0.0;
				dstB[0] = in0 ? sg.b[q].x - fqB[q] : 0.0;
				dstB[1] = in1 ? sg.b[q].y - fqB[q] : 0.0;
			} else if (left >= kChunkRows) { // wave-uniform: every row of the chunk exists
				dst[0] = sg.a[q].x - fq[q];
				dst[1] = sg.a[q].y - fq[q];
				if (kWideChunk) {
					dst[16] = sg.b[q].x - fq[q];
					dst[17] = sg.b[q].y - fq[q];
				}
			} else { // the group's last chunk: rows past its end as zeros
				dst[0] = 2 * rp < left ? sg.a[q].x - fq[q] : 0.0;
				dst[1] = 2 * rp + 1 < left ? sg.a[q].y - fq[q] : 0.0;
				if (kWideChunk) {
					dst[16] = 2 * rp + 16 < left ? sg.b[q].x - fq[q] : 0.0;
					dst[17] = 2 * rp + 17 < left ? sg.b[q].y - fq[q] : 0.0;
				}
			}
		}
	};
	// (steady state: every slot, a full chunk; see stage_load_piece_steady)
	auto stage_store_piece_steady = [&](int buf, const Stage &sg, StoreState &ss, int q) {
		if (!FAST) {
			ss.z0 = fma(sg.a[q].x, 0.0, ss.z0);
			ss.z1 = fma(sg.a[q].y, 0.0, ss.z1);
			if (kWideChunk) {
				ss.z2 = fma(sg.b[q].x, 0.0, ss.z2);
				ss.z3 = fma(sg.b[q].y, 0.0, ss.z3);
			}
		}
		if (WEIGHTED) {
			const bool isw = (wbits >> q) & 1u;
			ss.wbad0 = ss.wbad0 || (isw && !(sg.a[q].x > 0.0));
			ss.wbad1 = ss.wbad1 || (isw && !(sg.a[q].y > 0.0));
			ss.wbad2 = ss.wbad2 || (isw && !(sg.b[q].x > 0.0));
			ss.wbad3 = ss.wbad3 || (isw && !(sg.b[q].y > 0.0));
		}
		double *dst = image + buf * ncol_pad * kLdsStride + dcol[q];
#if ANOFOX_WIDE_SKIP & 16
		if (FAST) { asm volatile("" ::"v"(sg.a[q].x), "v"(sg.a[q].y), "v"(sg.b[q].x), "v"(sg.b[q].y)); return; }
#endif
		dst[0] = sg.a[q].x - fq[q];
		dst[1] = sg.a[q].y - fq[q];
		if constexpr (kRun) {
			double *dstB = image + buf * ncol_pad * kLdsStride + dcolB[q];
			dstB[0] = sg.b[q].x - fqB[q];
			dstB[1] = sg.b[q].y - fqB[q];
		} else if (kWideChunk) {
			dst[16] = sg.b[q].x - fq[q];
			dst[17] = sg.b[q].y - fq[q];
		}
	};
	// this wave's partial row-validity mask (ols.rs:59-66, wls.rs:76-86) of the chunk, once every piece is through
	auto stage_store_finish = [&](int64_t chunk, int buf, const StoreState &ss) {
		if (FAST) return; // no row masks
		const int64_t left = nrows - chunk * kChunkRows;
		const bool ok0 = !isnan(ss.z0) && !ss.wbad0, ok1 = !isnan(ss.z1) && !ss.wbad1, ok2 = !isnan(ss.z2) && !ss.wbad2,
		           ok3 = !isnan(ss.z3) && !ss.wbad3;
		// fold the 8 column sub-groups: byte k of the mask = rows {2j, 2j+1, 16+2j, 17+2j}[k], j = bit (row_bit below)
		unsigned long long b0 = __ballot(ok0), b1 = __ballot(ok1);
		b0 &= b0 >> 32; b0 &= b0 >> 16; b0 &= b0 >> 8;
		b1 &= b1 >> 32; b1 &= b1 >> 16; b1 &= b1 >> 8;
		unsigned m = ((unsigned)b0 & 0xffu) | (((unsigned)b1 & 0xffu) << 8);
		if (kWideChunk) {
			unsigned long long b2 = __ballot(ok2), b3 = __ballot(ok3);
			b2 &= b2 >> 32; b2 &= b2 >> 16; b2 &= b2 >> 8;
			b3 &= b3 >> 32; b3 &= b3 >> 16; b3 &= b3 >> 8;
			m |= (((unsigned)b2 & 0xffu) << 16) | (((unsigned)b3 & 0xffu) << 24);
		}
		if (left < kChunkRows) m &= range_mask<kChunkRows>(left); // rows past the end of the group are invalid
		if (lane == 0) maskslot[buf * kWaves + wave] = m;
	};
	// registers -> LDS image `buf` (shifted by fq; rows past the end of the group as zeros), plus the partial row mask.
	// Straight-line code: lanes whose column does not exist store into a spare column and count as valid.
	auto stage_store = [&](int64_t chunk, int buf, const Stage &sg) {
		StoreState ss;
		stage_store_begin(ss);
#pragma unroll
		for (int q = 0; q < kMaxLoads; ++q) stage_store_piece(chunk, buf, sg, ss, q);
		stage_store_finish(chunk, buf, ss);
	};
	// Rare repairs of a staged image, each lane on the elements it staged itself (so reads precede writes in program
	// order): `shift` — the chunk was staged before the group's first valid row was known, subtract it now; and every
	// row that is in range but invalid (NaN / inf somewhere, w <= 0) becomes zero in all columns.
	auto repair_image = [&](int buf, unsigned rowmask, bool shift) {
		double *img = image + buf * ncol_pad * kLdsStride;
#pragma unroll
		for (int q = 0; q < kMaxLoads; ++q) {
			if (wave + kWaves * q < n_loads_total && ((actbits >> q) & 1u)) {
				double *dst = img + dcol[q];
				const double f = shift ? fq[q] : 0.0;
#pragma unroll
				for (int e = 0; e < (kWideChunk ? 4 : 2); ++e) {
					const int bit = 16 * (e >> 1) + 8 * (e & 1) + rp; // row_bit(2 rp + (e & 1) + 16 (e >> 1))
					const double cur = dst[(e & 1) + 16 * (e >> 1)];
					dst[(e & 1) + 16 * (e >> 1)] = ((rowmask >> bit) & 1u) ? cur - f : 0.0;
				}
			}
		}
	};

	// One chunk.  The loads run one chunk ahead and are stored to the other image once this chunk's MFMAs are issued.
	ACC_STAMP_DECL;
	// steady = std::true_type: chunks c and c + 1 are full (the branch-free staging pieces)
	// STEADY: the loads of chunk c + depth go into `ahead`, chunk c + 1 is stored from `landed` (the same registers when kDepth = 1)
	// (generic iterations: `ga` = how many chunks beyond c + 1 the chunk lies that is loaded into `ahead` once chunk c + 1 is stored from it)
	auto iteration = [&](auto steady, int64_t c, Stage &ahead, Stage &landed, int ga) {
		constexpr bool STEADY = decltype(steady)::value;
		ACC_STAMP_START();
		const int buf = (int)(c & 1);
		const double *img = image + buf * ncol_pad * kLdsStride;
		const int64_t left = nrows - c * kChunkRows;
		const unsigned rangemask = (STEADY || left >= kChunkRows) ? kFullMask : range_mask<kChunkRows>(left);
		unsigned rowmask = rangemask;
		if (!FAST) {
			rowmask = maskslot[buf * kWaves];
#pragma unroll
			for (int w = 1; w < kWaves; ++w) rowmask &= maskslot[buf * kWaves + w];
			rowmask = __builtin_amdgcn_readfirstlane(rowmask);
		}
		ACC_STAMP(0);
		const bool more = STEADY || c + 1 < n_chunks; // wave-uniform

		const bool found_first = !have_first && rowmask != 0u; // wave-uniform
		// (a chunk without any valid row is zeroed as well and goes through the MFMAs like every other chunk: skipping
		// its compute step would put the 72 accumulator registers behind a branch, and the compiler then copies all of
		// them at the join — 36 v_mov_b64 waiting on the last MFMA of EVERY chunk)
		if (!FAST && (found_first || rowmask != rangemask)) {
			if (found_first) {
				// the group's first valid row: remember its values (the shift when CENTER, the reference point of the
				// constant-column test, part of the record) and shift this chunk, which was staged unshifted
				const int r = first_row_of_mask(rowmask);
#pragma unroll
				for (int q = 0; q < kMaxLoads; ++q) {
					const int li = wave + kWaves * q;
					if (li < n_loads_total) {
						const int src = 8 * li + colsub;
						if (src <= p) { // x columns and y; the weight column is never shifted
							const int col = src < p ? src : ycol;
							const double f = img[col * kLdsStride + r];
							if (rp == 0) firstcol[col] = f;
							if (CENTER) fq[q] = f;
						}
					}
				}
			}
			repair_image(buf, rowmask, CENTER && found_first);
			__syncthreads();
		}
		have_first = have_first || found_first;

		cnt += __popc(rowmask);
		ACC_STAMP(1);
		// The next chunk is staged BETWEEN the slabs of this one.  Steady state: its loads go out behind slab 0, load slot
		// q is shifted and written to the other image behind one of the last slabs, the row mask follows behind the
		// last slab.  (Outside the steady state — a partial chunk ahead — the generic pieces do the same one chunk
		// further ahead, behind whatever waits the compiler chooses.)  With the staging as a phase of its own a wave
		// spent 35 % of a chunk in it: its ~100 vector / LDS instructions each wait for a gap in the MFMA stream of the
		// SIMD's other wave (phase stamps), while its own matrix work stands still.
		constexpr int NS = kChunkRows / 4;
		StoreState ss;
		stage_store_begin(ss);
		compute_chunk<T, WAVE, WEIGHTED, CENTER, FAST>(img, firstcol, ycol, lane, rowmask, acc, sx, sxy, dmax, ncmask, sy, syy, sw, [&](int t) {
			if (STEADY) {
				if (t == 0) {
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int q = 0; q < kMaxLoads; ++q) stage_load_piece_steady(c + Cfg::depth(FAST), ahead, q);
					__builtin_amdgcn_sched_barrier(0);
				}
#pragma unroll
				for (int q = 0; q < kMaxLoads; ++q) {
					// as late as the slabs allow (several slots per slab when there are fewer slabs than slots)
					constexpr int kPerSlab = (kMaxLoads + NS - 2) / (NS - 1);
					if (NS - 1 - (kMaxLoads - 1 - q) / kPerSlab == t) {
						// (fences for the instruction scheduler: left alone it gathers the subtractions of all slots
						// behind one wait for the last load)
						__builtin_amdgcn_sched_barrier(0);
						stage_store_piece_steady(buf ^ 1, landed, ss, q);
						__builtin_amdgcn_sched_barrier(0);
					}
				}
				return;
			}
			if (!more) return;
#pragma unroll
			for (int q = 0; q < kMaxLoads; ++q) {
				if ((q * NS) / kMaxLoads == t) { // slab behind which slot q is stored
					stage_store_piece(c + 1, buf ^ 1, ahead, ss, q);
					if (c + 1 + ga < n_chunks) stage_load_piece(c + 1 + ga, ahead, q);
				}
			}
		});
		ACC_STAMP(2);
		if (more) stage_store_finish(c + 1, buf ^ 1, ss);
		ACC_STAMP(4);
#if ANOFOX_WIDE_SKIP & 4
		if (!(FAST && STEADY))
#endif
		__syncthreads();
		ACC_STAMP(5);
#ifdef ANOFOX_SOLVE_STAMPS
		++st_n;
#endif
	};

	if constexpr (FAST && DMA != 0) {
		// ---- (r4) the speculative version staged by `global_load_lds_dword` (lds_dma.h): no row passes through a register on its way
		// into the image, no vector instruction shifts or stores it.  The x columns are dealt to the four wavefronts in contiguous
		// ranges (the pointer table of the kernel arguments is walked as in accumulate_quad's kernel), the last wavefront also brings
		// y; chunk c + 1 lands in the other image while the slabs of chunk c run, one barrier per chunk as before.  The image holds
		// RAW rows: the shift by the first row is one subtraction per fragment (compute_chunk RAW).  The rows of the last chunk
		// that lie behind the group are overwritten with the first row's values (difference 0) by the wavefront that loaded them.
		static_assert(kChunkRows == 32 || kChunkRows == 64, "32 or 64 rows per chunk");
		constexpr int QMAX = (P16 + kWaves - 1) / kWaves;
		const int q_cols = (p + kWaves - 1) / kWaves;
		const int c_begin = WAVE * q_cols;
		int n_mine = p - c_begin;
		n_mine = n_mine < 0 ? 0 : (n_mine > q_cols ? q_cols : n_mine);
		// the first row, by image column (x in place, y at 16 T); the loop's first barrier orders these writes before the reads
		for (int c = threadIdx.x; c <= p; c += kThreads) firstcol[c < p ? c : ycol] = c < p ? args.x_table[c][lo] : args.y[lo];
		const lds_dma_table_t tab = lds_dma_table((unsigned)offsetof(WideArgs, x_table)) + c_begin;
		const unsigned img0 = lds_dma_address(image);
		auto dma_chunk = [&](int64_t c, int buf) {
#pragma unroll
			for (int sub = 0; sub < kChunkRows / 32; ++sub) { // 256 bytes of a column per instruction: a 64-row chunk takes two per column
				const int64_t blk = lo + c * kChunkRows + 32 * sub;
				if (blk >= hi) break; // wave-uniform (the rows are filled by fill_tail)
				const unsigned voff = lds_dma_offsets(lane, hi - blk);
				if (n_mine > 0) lds_dma_block<kLdsStride * 8, QMAX>(tab, n_mine, blk, voff, img0 + (unsigned)((buf * ncol_pad + c_begin) * kLdsStride * 8 + sub * 256));
				if (WAVE == kWaves - 1) lds_dma1(voff, img0 + (unsigned)((buf * ncol_pad + ycol) * kLdsStride * 8 + sub * 256), args.y + blk);
			}
		};
		auto fill_tail = [&](int buf, int left) { // rows left .. 31 of the columns this wavefront loaded := the first row
			double *img = image + buf * ncol_pad * kLdsStride;
			const int nr = kChunkRows - left;
			for (int e = lane; e < n_mine * nr; e += 64) {
				const int cc = c_begin + e / nr, r = left + e % nr;
				img[cc * kLdsStride + r] = firstcol[cc];
			}
			if (WAVE == kWaves - 1)
				for (int r = left + lane; r < kChunkRows; r += 64) img[ycol * kLdsStride + r] = firstcol[ycol];
		};
		const int tail = (int)(nrows % kChunkRows);
		dma_chunk(0, 0);
		lds_dma_wait_all();
		__syncthreads(); // (firstcol complete, chunk 0 landed)
		if (n_chunks == 1 && tail) {
			fill_tail(0, tail);
			__syncthreads();
		}
		double fsub[T];
#pragma unroll
		for (int I = 0; I < T; ++I) fsub[I] = firstcol[16 * I + (lane & 15)];
		const double fy = firstcol[ycol];
		for (int64_t c = 0; c < n_chunks; ++c) {
			const int buf = (int)(c & 1);
			const bool more = c + 1 < n_chunks; // wave-uniform
			if (more) dma_chunk(c + 1, buf ^ 1);
			compute_chunk<T, WAVE, WEIGHTED, CENTER, FAST, true, kChunkRows>(image + buf * ncol_pad * kLdsStride, firstcol, ycol, lane, kFullMask, acc, sx, sxy, dmax, ncmask,
			                                                    sy, syy, sw, [](int) {}, fsub, fy);
			lds_dma_wait_all();
			if (more && c + 2 == n_chunks && tail) fill_tail(buf ^ 1, tail);
			__syncthreads();
		}
	} else {
	constexpr int D = Cfg::depth(FAST);
	Stage st[D];
	if constexpr (kTail2) {
		stage_load(0, st[0]);
		if (n_chunks > 1) stage_load(1, st[1]);
		// the shift = the group's first row (nrows > 0: the caller's condition): rows 0, 1 of a column sit in the lane with rp == 0
#pragma unroll
		for (int q = 0; q < kMaxLoads; ++q) {
			const int src = 8 * (wave + kWaves * q) + colsub;
			const double f = __shfl(st[0].a[q].x, lane & (kRun ? ~15 : ~7), 64);
			if (wave + kWaves * q < n_loads_total && src <= p) {
				fq[q] = f;
				if (rp == 0) firstcol[src < p ? src : ycol] = f;
			}
			if constexpr (kRun) {
				const int srcB = src + 4;
				const double fb = __shfl(st[0].b[q].x, lane & ~15, 64);
				if (wave + kWaves * q < n_loads_total && srcB <= p) {
					fqB[q] = fb;
					if (rp == 0) firstcol[srcB < p ? srcB : ycol] = fb;
				}
			}
		}
		have_first = true;
		stage_store(0, 0, st[0]);
	} else if (n_chunks > 0) {
		stage_load(0, st[0]);
		stage_store(0, 0, st[0]);
	}
	__syncthreads();
	ACC_MARK(2);
	{
		int64_t c = 0;
		const int64_t n_full = nrows / kChunkRows; // full chunks
		if constexpr (D == 1) {
			for (; c + 1 < n_full; ++c) iteration(std::true_type(), c, st[0], st[0], 1); // (nothing of chunk c + 1 is loaded yet)
		} else {
			// trips of D chunks, so that which registers hold which chunk is known at compile time: chunk c + j + 1 sits in st[(j + 1) % D]
			// when chunk c + j is multiplied, and the loads of chunk c + j + D go into st[j], whose chunk was stored one iteration ago
			if (c + 2 * D - 1 < n_full) {
				if constexpr (!kTail2) {
#pragma unroll
					for (int j = 1; j < D; ++j) stage_load(j, st[j]);
				}
				for (; c + 2 * D - 1 < n_full; c += D) {
#pragma unroll
					for (int j = 0; j < D; ++j) iteration(std::true_type(), c + j, st[j], st[(j + 1) % D], 1);
				}
			}
		}
		ACC_MARK(3);
		if constexpr (kTail2) {
			// chunk c is staged and chunk c + 1 sits in st[1] (from before the loop, or from its last trip); chunk c + 2 goes into st[0] now and
			// the remaining iterations alternate between the two sets
			if (c + 2 < n_chunks) stage_load(c + 2, st[0]);
			for (;;) {
				if (c >= n_chunks) break;
				iteration(std::false_type(), c, st[1], st[1], 2);
				++c;
				if (c >= n_chunks) break;
				iteration(std::false_type(), c, st[0], st[0], 2);
				++c;
			}
		} else {
			// (the generic iterations expect the next chunk in registers; after steady trips it is there already and is read once more)
			if (c + 1 < n_chunks) stage_load(c + 1, st[1 % D]);
			for (; c < n_chunks; ++c) iteration(std::false_type(), c, st[1 % D], st[1 % D], 1);
		}
	}
	}

	ACC_STAMP_FLUSH();
	ACC_MARK(4);
	// (r4) the tiles that were split along the rows: the partial sums go through the image (free since the loop's last barrier)
	// to the tile's owner, element for element (every wavefront holds a tile in the same lane / register layout)
	if constexpr (Cfg::R > 0) {
		int slot = 0; // 256 doubles per partial
		bool wrote = false;
#pragma unroll
		for (int tile = 4 * Cfg::F; tile < Cfg::NT; ++tile) {
			const int parts = Cfg::quartered(tile) ? 4 : 2, first = Cfg::owner(tile);
			if (WAVE > first && WAVE < first + parts) {
#pragma unroll
				for (int r = 0; r < 4; ++r) image[(slot + WAVE - first - 1) * 256 + 64 * r + lane] = acc[Cfg::acc_index(tile)][r];
				wrote = true;
			}
			slot += parts - 1;
		}
		(void)wrote;
		__syncthreads();
		slot = 0;
#pragma unroll
		for (int tile = 4 * Cfg::F; tile < Cfg::NT; ++tile) {
			const int parts = Cfg::quartered(tile) ? 4 : 2, first = Cfg::owner(tile);
			if (WAVE == first) {
#pragma unroll
				for (int k = 0; k < parts - 1; ++k)
#pragma unroll
					for (int r = 0; r < 4; ++r) acc[Cfg::acc_index(tile)][r] += image[(slot + k) * 256 + 64 * r + lane];
			}
			slot += parts - 1;
		}
		__syncthreads(); // (the callers rewrite the LDS)
	}
	ACC_MARK(5);
	unsigned fast_nc[Cfg::TPW]; // FAST: constant-column flags of the diagonal tiles this wave holds (bit r: element r)
	if (FAST) {
		// what the speculation assumed, checked on the result: every moment finite, every column clearly constant or
		// clearly not.  (n_chunks >= 1, so the loop's barriers ordered the firstcol writes before the reads below.)
		double zs = 0.0;
#pragma unroll
		for (int t = 0; t < Cfg::TPW; ++t) zs = fma(acc[t][0], 0.0, fma(acc[t][1], 0.0, fma(acc[t][2], 0.0, fma(acc[t][3], 0.0, zs))));
#pragma unroll
		for (int o = 0; o < Cfg::OWN; ++o) zs = fma(sx[o], 0.0, fma(sxy[o], 0.0, zs));
		zs = fma(sy, 0.0, fma(syy, 0.0, zs));
		bool bad = isnan(zs);
		const double n_d = (double)nrows;
		int tile = 0;
#pragma unroll
		for (int I = 0; I < T; ++I) {
#pragma unroll
			for (int J = I; J < T; ++J) {
				if (Cfg::owner(tile) == WAVE) {
					fast_nc[Cfg::acc_index(tile)] = 0u;
					if (J == I) {
						// diagonal element (j, j), j = lane & 15: held by the lane with lane >> 4 == j % 4, in element j / 4
						const int j = lane & 15;
						if ((lane >> 4) == (j & 3)) {
							double mjj = acc[Cfg::acc_index(tile)][0];
#pragma unroll
							for (int r = 1; r < 4; ++r) mjj = (j >> 2) == r ? acc[Cfg::acc_index(tile)][r] : mjj;
							const bool moved = mjj >= n_d * 1e-20;
							bad = bad || (!moved && !(mjj < 1e-20));
							fast_nc[Cfg::acc_index(tile)] = moved ? 1u : 0u;
						}
					}
				}
				++tile;
			}
		}
		const bool wave_bad = __ballot(bad) != 0ull;
		if (lane == 0) maskslot[wave] = wave_bad ? 1u : 0u;
		__syncthreads();
		unsigned any_bad = maskslot[0];
#pragma unroll
		for (int w = 1; w < kWaves; ++w) any_bad |= maskslot[w];
		__syncthreads(); // (the caller's full version rewrites the LDS)
		if (__builtin_amdgcn_readfirstlane(any_bad) != 0u) return false;
		cnt = (int)nrows;
	}
	ACC_MARK(6);
	// ---- write the moment record ----
	// tiles: tile-major, 256 doubles each, element (row, col) at row*16 + col
	{
		int tile = 0;
#pragma unroll
		for (int I = 0; I < T; ++I) {
#pragma unroll
			for (int J = I; J < T; ++J) {
				if (Cfg::owner(tile) == wave) {
					double *tp = rec + (int64_t)tile * 256;
#pragma unroll
					for (int r = 0; r < 4; ++r) tp[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[Cfg::acc_index(tile)][r];
				}
				++tile;
			}
		}
	}
	double *vec = rec + (int64_t)Cfg::NT * 256;
	if (FAST) {
		int tile = 0;
#pragma unroll
		for (int I = 0; I < T; ++I) {
#pragma unroll
			for (int J = I; J < T; ++J) {
				if (Cfg::owner(tile) == wave && J == I && (lane >> 4) == (lane & 3))
					vec[3 * P16 + 16 * I + (lane & 15)] = (double)fast_nc[Cfg::acc_index(tile)];
				++tile;
			}
		}
	}
	// column sums: reduce over the four k-groups (lanes l, l^16, l^32, l^48)
#pragma unroll
	for (int I = 0; I < T; ++I) {
		if (I % kWaves == wave) {
			double a = sx[I / kWaves], b = sxy[I / kWaves];
			a += shfl_xor_d(a, 16); a += shfl_xor_d(a, 32);
			b += shfl_xor_d(b, 16); b += shfl_xor_d(b, 32);
			unsigned nc = CENTER ? (dmax[I / kWaves] >= 1e-10 ? 1u : 0u) : ((ncmask >> I) & 1u);
			nc |= __shfl_xor((int)nc, 16, 64);
			nc |= __shfl_xor((int)nc, 32, 64);
			if (lane < 16) {
				vec[0 * P16 + 16 * I + lane] = a;
				vec[1 * P16 + 16 * I + lane] = b;
				vec[2 * P16 + 16 * I + lane] = firstcol[16 * I + lane]; // padding columns: 0
				if (!FAST) vec[3 * P16 + 16 * I + lane] = (double)nc;
			}
		}
	}
	if (wave == 0) {
		// every lane of a k-group holds the same partial: take lanes 0, 16, 32, 48
		double a = sy, b = syy, c2 = sw;
		a += shfl_xor_d(a, 16); a += shfl_xor_d(a, 32);
		b += shfl_xor_d(b, 16); b += shfl_xor_d(b, 32);
		c2 += shfl_xor_d(c2, 16); c2 += shfl_xor_d(c2, 32);
		if (!WEIGHTED) c2 = (double)cnt;
		if (lane == 0) {
			double *sc = vec + 4 * P16;
			sc[0] = a;
			sc[1] = b;
			sc[2] = c2;
			sc[3] = (double)cnt;
			sc[4] = firstcol[ycol];
		}
	}
	ACC_MARK(7);
	return true;
}

// FAST: the speculative version (every wave returns the same verdict); otherwise the full one (always true).
// The two never share a kernel: together they need more registers than there are (1110 spilled at T = 8).
template <int T, bool WEIGHTED, bool CENTER, bool FAST = false, int DMA = 0>
__device__ __forceinline__ bool wide_accumulate_rows(const WideArgs &args, int64_t lo, int64_t hi, double *rec,
                                                     const double *forced_first) {
	switch (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) {
	case 0: return wide_accumulate_rows_wave<T, 0, WEIGHTED, CENTER, FAST, DMA>(args, lo, hi, rec, forced_first);
	case 1: return wide_accumulate_rows_wave<T, 1, WEIGHTED, CENTER, FAST, DMA>(args, lo, hi, rec, forced_first);
	case 2: return wide_accumulate_rows_wave<T, 2, WEIGHTED, CENTER, FAST, DMA>(args, lo, hi, rec, forced_first);
#if ANOFOX_WIDE_WAVES == 8
	case 3: return wide_accumulate_rows_wave<T, 3, WEIGHTED, CENTER, FAST, DMA>(args, lo, hi, rec, forced_first);
	case 4: return wide_accumulate_rows_wave<T, 4, WEIGHTED, CENTER, FAST, DMA>(args, lo, hi, rec, forced_first);
	case 5: return wide_accumulate_rows_wave<T, 5, WEIGHTED, CENTER, FAST, DMA>(args, lo, hi, rec, forced_first);
	case 6: return wide_accumulate_rows_wave<T, 6, WEIGHTED, CENTER, FAST, DMA>(args, lo, hi, rec, forced_first);
	default: return wide_accumulate_rows_wave<T, 7, WEIGHTED, CENTER, FAST, DMA>(args, lo, hi, rec, forced_first);
#else
	default: return wide_accumulate_rows_wave<T, 3, WEIGHTED, CENTER, FAST, DMA>(args, lo, hi, rec, forced_first);
#endif
	}
}

// FAST: the speculative version on every group; the groups it gives up on (and empty ones) go to a list — borrowed
// from the refine queue, which the solve that follows starts to fill only later — for accumulate_wide_redo_kernel.
// (measurement builds: the register budget of the speculative version cut for more wavefronts per SIMD, `-DANOFOX_WIDE_FAST_WPS(T)=3`)
#ifndef ANOFOX_WIDE_FAST_WPS
#define ANOFOX_WIDE_FAST_WPS(T) kWavesPerSimd
#endif
template <int T, bool WEIGHTED, bool CENTER, bool FAST, int DMA = 0>
__global__ __launch_bounds__(kThreads, (FAST ? ANOFOX_WIDE_FAST_WPS(T) : kWavesPerSimd)) void accumulate_wide_kernel(WideArgs args) {
	const int64_t g = blockIdx.x;
	const int64_t lo = args.row_offsets[args.group_base + g];
	const int64_t hi = group_row_end(args, args.group_base + g);
	if (args.seg_table && hi - lo > args.seg_rows) {
		// four waves stream a group at ~25 GB/s: hand it to accumulate_wide_segments_kernel in pieces
		__shared__ int registered;
		if (threadIdx.x < 64) {
			const bool ok = wide_register_big_group(args, g, lo, hi, T, (int)threadIdx.x, kWideSegMaxBig, kWideSegMaxSegments);
			if (threadIdx.x == 0) registered = ok ? 1 : 0;
		}
		__syncthreads();
		if (registered) return; // (tables full: the workgroup accumulates the group itself)
	}
	double *rec = args.moments + g * (int64_t)wide_record_len(T);
	if constexpr (FAST) {
#if ANOFOX_WIDE_STAGGER
		// Groups of equal length keep the workgroups that share a CU in step: they start together, share the SIMDs evenly and reach their
		// latency-bound ends (first loads, last chunks, record) together, with nothing left to fill the matrix pipe.  The workgroups of the
		// grid's first wave therefore start a random fraction of one group's time apart (s_sleep, ~4 us per step; large grids only).
		if (gridDim.x >= 4096 && blockIdx.x < 1024) {
			unsigned steps = (unsigned)(((hi - lo) * (int64_t)(args.p + 1) * 8) / (6500 * 4)); // a group's time at ~6.5 GB/s per workgroup
			steps = steps > 64u ? 64u : steps;
			const unsigned k = steps ? ((blockIdx.x * 2654435761u) >> 20) % steps : 0u;
			for (unsigned i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(127);
		}
#endif
		if (hi > lo && wide_accumulate_rows<T, WEIGHTED, CENTER, true, DMA>(args, lo, hi, rec, nullptr)) return;
		if (threadIdx.x == 0) args.refine_list[atomicAdd(args.refine_count + kWideRedoCounter, 1)] = (int32_t)g;
	} else {
		wide_accumulate_rows<T, WEIGHTED, CENTER>(args, lo, hi, rec, nullptr);
	}
}

// the full version on the groups the speculative kernel listed: one workgroup per list entry (launched with one per
// group of the batch — a loop over the list inside the kernel costs the full version 118 spilled registers at T = 8)
template <int T, bool WEIGHTED, bool CENTER>
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void accumulate_wide_redo_kernel(WideArgs args) {
	if ((int)blockIdx.x >= args.refine_count[kWideRedoCounter]) return;
	const int64_t g = args.refine_list[blockIdx.x];
	wide_accumulate_rows<T, WEIGHTED, CENTER>(args, args.row_offsets[args.group_base + g], group_row_end(args, args.group_base + g),
	                                          args.moments + g * (int64_t)wide_record_len(T), nullptr);
}

// One workgroup per registered segment; every segment of a group uses the group's first valid row as its shift,
// so the workgroup that completes the last one merges by plain (ordered) sums.
template <int T, bool WEIGHTED, bool CENTER>
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void accumulate_wide_segments_kernel(WideArgs args) {
	constexpr int P16 = 16 * T;
	constexpr int NT = T * (T + 1) / 2;
	const int reclen = wide_record_len(T);
	const int v = blockIdx.x;
	SegHeader *h = wseg_header(args.seg_table);
	const int total = h->seg_total; // reservations never exceed the capacity
	if (v >= total) return;
	const SegEntry e = wseg_entries(args.seg_table, kWideSegMaxBig)[v];
	if (e.slot < 0) return; // reserved but unclaimed
	SegBigGroup *b = wseg_big(args.seg_table) + e.slot;
	const double *ff = wseg_first(args.seg_table, kWideSegMaxBig, kWideSegMaxSegments) + (size_t)e.slot * (P16 + 2);
	double *recs = wseg_records(args.seg_table, T, kWideSegMaxBig, kWideSegMaxSegments);
	wide_accumulate_rows<T, WEIGHTED, CENTER>(args, e.lo, e.hi, recs + (int64_t)v * reclen, ff);
	__shared__ int last;
	__threadfence(); // this segment's record before the counter
	__syncthreads();
	if (threadIdx.x == 0) last = (atomicAdd(&b->done, 1) == b->nseg - 1) ? 1 : 0;
	__syncthreads();
	if (!last) return;
	__threadfence(); // every other segment's record after the counter
	const double *src = recs + (int64_t)b->base * reclen;
	double *dst = args.moments + b->g * (int64_t)reclen;
	const int vec0 = NT * 256;
	for (int k = threadIdx.x; k < reclen; k += kThreads) {
		const bool is_first = (k >= vec0 + 2 * P16 && k < vec0 + 3 * P16) || k == vec0 + 4 * P16 + 4; // first x / first y: shared
		const bool is_flag = k >= vec0 + 3 * P16 && k < vec0 + 4 * P16;                                   // non-constant flags: OR
		double acc = 0.0;
		for (int t = 0; t < b->nseg; ++t) acc += src[(int64_t)t * reclen + k];
		if (is_first) acc = src[k];
		if (is_flag) acc = acc > 0.0 ? 1.0 : 0.0;
		dst[k] = acc;
	}
}

} // namespace

template <int T>
hipError_t launch_accumulate_wide_T(const WideArgs &a, hipStream_t stream) {
	const bool weighted = a.model == ANOFOX_HIP_MODEL_WLS;
	const bool center = a.fit_intercept != 0;
	const int ncol_pad = wide_ncol_pad(a.p, weighted);
	const size_t lds = (size_t)2 * ncol_pad * WideCfg<T>::stride(weighted, center) * sizeof(double) + (size_t)ncol_pad * (sizeof(double *) + sizeof(double)) + 64;
	const dim3 grid((unsigned)a.n_groups), block(kThreads);
	const dim3 seg_grid((unsigned)kWideSegMaxSegments); // idle unless some group exceeded seg_rows
	// launch_part (host_api.hip, several slabs): the idle-unless-needed kernels behind the main one run on the stream of the slab's
	// solve, so that they never queue on the accumulate stream behind the previous slab's solve (which holds every SIMD's registers)
	const bool main_part = a.launch_part != 2, rest_part = a.launch_part != 1;
#define ANOFOX_WIDE_LAUNCH(W, C)                                                                                  \
	do {                                                                                                          \
		if (main_part) hipLaunchKernelGGL((accumulate_wide_kernel<T, W, C, false>), grid, block, lds, stream, a); \
		if (rest_part && a.seg_table) hipLaunchKernelGGL((accumulate_wide_segments_kernel<T, W, C>), seg_grid, block, lds, stream, a); \
	} while (0)
	if (weighted) {
		if (center) ANOFOX_WIDE_LAUNCH(true, true);
		else ANOFOX_WIDE_LAUNCH(true, false);
	} else if (!center) {
		ANOFOX_WIDE_LAUNCH(false, false);
	} else if (a.no_fast_path || T < kWideFastMinT) {
		ANOFOX_WIDE_LAUNCH(false, true);
	} else {
		// speculative kernel, then the full version on whatever it listed (the counter is zeroed by the caller)
		// (r4) ANOFOX_WIDE_DMA=1: its LDS-DMA staging (3 and 4 column tiles: 35 <= p <= 64, the widths accumulate_quad does not take)
		static const int dma_rows = getenv("ANOFOX_WIDE_DMA") ? atoi(getenv("ANOFOX_WIDE_DMA")) : 0; // 1 / 32: 32-row chunks, 64: 64-row chunks
		constexpr bool kHasDma = T >= kWideFastMinT && T <= 4 && kWaves == 4;
		const size_t lds64 = (size_t)2 * ncol_pad * 66 * sizeof(double) + (size_t)ncol_pad * (sizeof(double *) + sizeof(double)) + 64;
		if (main_part && kHasDma && dma_rows == 64) {
			static const bool attr_set = [] {
				(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_wide_kernel<T, false, true, (T >= kWideFastMinT), (kHasDma ? 64 : 0)>),
				                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); // (the kernel has 4 bytes of static LDS)
				return true;
			}();
			(void)attr_set;
			hipLaunchKernelGGL((accumulate_wide_kernel<T, false, true, (T >= kWideFastMinT), (kHasDma ? 64 : 0)>), grid, block, lds64, stream, a);
		} else if (main_part && kHasDma && dma_rows != 0) {
			hipLaunchKernelGGL((accumulate_wide_kernel<T, false, true, (T >= kWideFastMinT), (kHasDma ? 32 : 0)>), grid, block, lds, stream, a);
		} else if (main_part) hipLaunchKernelGGL((accumulate_wide_kernel<T, false, true, (T >= kWideFastMinT)>), grid, block, lds, stream, a);
		if (rest_part && a.seg_table) hipLaunchKernelGGL((accumulate_wide_segments_kernel<T, false, true>), seg_grid, block, lds, stream, a);
		if (rest_part) hipLaunchKernelGGL((accumulate_wide_redo_kernel<T, false, true>), grid, block, lds, stream, a);
	}
#undef ANOFOX_WIDE_LAUNCH
	return hipGetLastError();
}

// What follows a speculative kernel that is NOT this file's (accumulate_quad.hip at p = 33, 34 registers its very large groups in
// the workgroup-per-segment table and lists its give-ups in the redo list): the segment kernel and the full version on the list.
template <int T>
hipError_t launch_accumulate_wide_followup_T(const WideArgs &a, hipStream_t stream) {
	const int ncol_pad = wide_ncol_pad(a.p, false);
	const size_t lds = (size_t)2 * ncol_pad * WideCfg<T>::stride(false, true) * sizeof(double) + (size_t)ncol_pad * (sizeof(double *) + sizeof(double)) + 64;
	const dim3 grid((unsigned)a.n_groups), block(kThreads), seg_grid((unsigned)kWideSegMaxSegments);
	if (a.seg_table) hipLaunchKernelGGL((accumulate_wide_segments_kernel<T, false, true>), seg_grid, block, lds, stream, a);
	hipLaunchKernelGGL((accumulate_wide_redo_kernel<T, false, true>), grid, block, lds, stream, a);
	return hipGetLastError();
}

} // namespace anofox
