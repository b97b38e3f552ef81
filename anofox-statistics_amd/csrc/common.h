// common.h — shared definitions between the HIP kernels and the host side of libanofox_stats_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/anofox_stats_hip.h"

namespace anofox {

// Register-resident ("narrow") path: one wavefront streams one group, every lane keeps the whole
// (p+1)(p+2)/2 moment triangle in VGPRs.  Above this the LDS/MFMA ("wide") path takes over.
constexpr int kNarrowMaxP = 8;

// bytes of the t critical value memo (device_math.h: TcritSlot[kTcritSlots]) at the end of a workspace
constexpr size_t kTcritTableBytes = 4096;

// Per-group moment record written by the accumulate kernel and read by the solve kernel.
// z = (x_1 .. x_p, y), Z = p + 1 columns, shifted by the group's first valid row ("first") when an
// intercept is fitted (shift 0 otherwise):  d = z - shift.
//   [0, Z)            s_a   = sum w d_a
//   [Z, Z+ZZ)         q_ab  = sum w d_a d_b, a <= b, row-major upper triangle
//   [Z+ZZ]            sw    = sum w           (w == 1 for OLS / ridge)
//   [KRED, KRED+Z)    first = z at the first valid row (x_first drives the constant-column test,
//                             crates/anofox-stats-core/src/models/ols.rs:76-87)
//   [KRED+Z]          cnt   = number of valid rows
//   [KRED+Z+1]        mask  = bit j set iff |x_j - x_j,first| >= 1e-10 on some valid row
template <int P>
struct MomentLayout {
	static constexpr int Z = P + 1;
	static constexpr int ZZ = Z * (Z + 1) / 2;
	static constexpr int OFF_S = 0;
	static constexpr int OFF_Q = Z;
	static constexpr int OFF_SW = Z + ZZ;
	static constexpr int KRED = Z + ZZ + 1; // entries that need a cross-lane reduction
	static constexpr int OFF_FIRST = KRED;
	static constexpr int OFF_CNT = KRED + Z;
	static constexpr int OFF_MASK = KRED + Z + 1;
	static constexpr int REC = KRED + Z + 2;
	__host__ __device__ static constexpr int q_index(int a, int b) { // a <= b
		return OFF_Q + a * Z - a * (a - 1) / 2 + (b - a);
	}
};

// Per queued group, the refinement passes hand the solve {sum w r^2, sum w r, X'Wr [p], sum w (y - ybar)^2}: the last
// entry is the centred second moment of y summed over the rows about the mean — what glmnet's lambda = n alpha / sd_y
// needs when the moments are uncentred (no intercept) and qyy - sy^2 / sw cancels.
inline __host__ __device__ int refine_vec_len(int p) { return p + 3; }
// (digits of sd_y lost to that cancellation ~ log10(qyy / cyy): beyond this ratio the group is queued for refinement)
constexpr double kGlmnetCancelRatio = 1e4;

// A-priori error of coefficient j from the Cholesky factor of the moment matrix: ~ eps / (smallest pivot ratio) *
// sqrt(tss / diag_j) — the size coefficient j would have if its column alone explained y, times the rounding the
// factorisation amplifies.  A coefficient whose own contribution to y is tiny next to the others' (a small column norm
// with a small coefficient) is wrong by far more than eps cond^2 relative to ITSELF: the deep narrow sweep's three misses
// (1.5 .. 2.5e-9 at pivot ratios 1.1 .. 1.6e-3, just above the pivot test) are within 3 x of this estimate.  The primary
// solves queue the group for the refinement passes when the estimate exceeds 1e-10 of max(|b_j|, 1e-3 max|b|) — the
// scale the parity bar of 1e-9 is measured on.  Benchmark-like designs (comparable contributions, pivot ratios ~ 1) sit
// five orders of magnitude below it.
constexpr double kCoefBoundEps = 1.1e-16, kCoefBoundTol = 1e-10;
inline __host__ __device__ bool coef_bound_weak(double beta, double beta_max, double diag0, double tss, double min_ratio) {
	const double scale = fmax(fabs(beta), 1e-3 * beta_max);
	return kCoefBoundEps * sqrt(tss / diag0) > kCoefBoundTol * min_ratio * scale;
}

inline __host__ __device__ int moment_record_len(int p) {
	const int Z = p + 1;
	return Z + Z * (Z + 1) / 2 + 1 + Z + 2;
}

struct BatchArgs {
	const int64_t *row_offsets; // [G+1]
	const double *y;            // [N]
	const double *x[kNarrowMaxP];
	const double *w;            // [N] or nullptr
	int64_t n_groups;
	int64_t n_rows;
	int p;
	// options
	int model; // AnofoxHipModel
	int fit_intercept;
	int compute_inference;
	int lambda_scaling;
	int hc_type; // AnofoxHcType; acted on by launch_hc_narrow only
	double confidence_level;
	double alpha;
	// workspace / outputs
	double *moments;      // [G * REC]
	double *core;         // [G * (p+6)]
	double *inference;    // [G * (5p+2)] or nullptr
	int32_t *refine_list; // [G]   groups whose RSS must be recomputed from residuals
	int32_t *refine_count; // [1]
	double *refine_vec;   // [G * refine_vec_len(p)]  {sum w r^2, sum w r, X'Wr, sum w (y - ybar)^2} of the queued groups
	void *tcrit_table;    // TcritSlot[kTcritSlots] (device_math.h), zeroed per call
	const int64_t *rule_counts; // optional [G]: the count the "< 2 rows -> NULL" rule looks at (default: rows of the group)
	// row splitting of very large groups (accumulate_narrow.hip): groups with more than seg_rows rows are cut into
	// segments of seg_rows rows, one wavefront each, and merged; seg_table == nullptr disables it
	void *seg_table;
	int64_t seg_rows;
	// optional [G]: group g owns rows [row_offsets[g], row_ends[g]) instead of [row_offsets[g], row_offsets[g+1]) —
	// row ranges may then overlap (window frames fitted as a batch of "virtual groups", frames.hip)
	const int64_t *row_ends;
};
inline __host__ __device__ int64_t group_row_end(const BatchArgs &a, int64_t g) { return a.row_ends ? a.row_ends[g] : a.row_offsets[g + 1]; }

// Segment bookkeeping of the narrow accumulate kernel.  seg_rows >= ceil(n_rows / kSegTargetWaves), so fewer than
// kSegTargetWaves groups can exceed it and their segments number fewer than 2 kSegTargetWaves: fixed-size tables.
constexpr int kSegTargetWaves = 2048;
constexpr int kSegMaxBig = kSegTargetWaves + 8;
constexpr int kSegMaxSegments = 2 * kSegTargetWaves + 16;
constexpr int64_t kSegMinRows = 8192;
struct SegHeader {
	int32_t seg_total; // segments registered (may exceed what was written if the caller understated n_rows)
	int32_t big_total; // groups registered
	int32_t pad[14];
};
struct SegBigGroup {
	int64_t g;
	int32_t base; // first segment
	int32_t nseg;
	int32_t done; // segments finished (the wave that finishes the last one merges)
	int32_t pad;
};
struct SegEntry {
	int64_t lo, hi; // rows
	int32_t slot;   // index into the big-group table
	int32_t pad;
};
inline __host__ __device__ size_t seg_table_bytes(int p) {
	return sizeof(SegHeader) + sizeof(SegBigGroup) * kSegMaxBig + sizeof(SegEntry) * kSegMaxSegments +
	       sizeof(double) * (size_t)kSegMaxSegments * (size_t)moment_record_len(p);
}
inline __host__ __device__ int64_t seg_rows_for(int64_t n_rows) {
	int64_t s = (n_rows + kSegTargetWaves - 1) / kSegTargetWaves;
	s = (s + 127) / 128 * 128;
	return s < kSegMinRows ? kSegMinRows : s;
}

// ---- wide path (8 < p <= kWideMaxP): FP64-MFMA accumulation, LDS Cholesky ----
constexpr int kWideMaxP = 128;

inline __host__ __device__ int wide_tiles(int p) { return (p + 15) / 16; }
// moment record of the wide path: [NT tiles x 256] | sx[P16] | sxy[P16] | first[P16] | nonconst[P16] | scalars[8]
//   tile (I <= J) number I*T - I(I-1)/2 + (J-I), element (r, c) at r*16 + c  =  M[16I + r][16J + c]
//   scalars: sy, syy, sw, cnt, first_y
inline __host__ __device__ int wide_record_len(int T) { return T * (T + 1) / 2 * 256 + 4 * 16 * T + 8; }
// columns of the accumulate kernel's LDS image: 16T (x, zero padded) + y + w, rounded up to 8
inline __host__ __device__ int wide_ncol_pad(int p, bool weighted) {
	(void)weighted;
	return (16 * wide_tiles(p) + 2 + 7) & ~7;
}

struct WideArgs {
	const int64_t *row_offsets; // [G_total + 1]
	const double *y;
	const double *x_table[kWideMaxP]; // column pointers, by value in the kernel arguments
	const double *w;
	int64_t group_base; // first group of this launch
	int64_t n_groups;   // groups in this launch
	int p;
	int model;
	int fit_intercept;
	int compute_inference;
	int lambda_scaling;
	int hc_type; // AnofoxHcType; acted on by launch_hc_wide only
	double confidence_level;
	double alpha;
	double *moments;      // [n_groups * wide_record_len(T)] (this launch)
	double *core;         // [G_total * (p+6)]
	double *inference;    // [G_total * (5p+2)] or nullptr
	int32_t *refine_list; // [G_total]
	int32_t *refine_count;
	double *refine_vec;   // [G_total * refine_vec_len(p)]
	void *tcrit_table;    // TcritSlot[kTcritSlots] (device_math.h), zeroed per call
	const int64_t *rule_counts; // optional [G_total], see BatchArgs
	double *hc_df;        // [n_groups of this launch] scratch of launch_hc_wide: residual df, NaN = group skipped
	// row splitting of very large groups, as in BatchArgs (here every segment is accumulated with the GROUP's first
	// valid row as its shift, found when the group is registered, so that merging is a plain sum)
	void *seg_table;
	int64_t seg_rows;
	const int64_t *row_ends; // optional [G_total], see BatchArgs
	int no_fast_path;        // accumulate_wide / accumulate_quad: 1 = skip the speculative version (A/B switch ANOFOX_WIDE_FAST=0, tests)
	int from_redo_list;      // accumulate_mid: 1 = wavefront k takes group refine_list[k], k < refine_count[kWideRedoCounter]
	int launch_part;         // launch_accumulate_wide: 0 = everything, 1 = the main kernel only, 2 = only what follows it (segment + redo kernels)
};
// word of the refine counter block that counts the give-ups of the speculative accumulate kernels (accumulate_wide_impl.h,
// accumulate_quad.hip); the list itself borrows refine_list, which the solve that follows starts to fill only later
constexpr int kWideRedoCounter = 8;
inline __host__ __device__ int64_t group_row_end(const WideArgs &a, int64_t g) { return a.row_ends ? a.row_ends[g] : a.row_offsets[g + 1]; }

// wide-record segment table: SegHeader | SegBigGroup[kSegMaxBig] | SegEntry[kSegMaxSegments] |
//   first[kSegMaxBig][16 T + 2] (x at the group's first valid row, then y) | records[kSegMaxSegments][record_len]
inline __host__ __device__ size_t wide_seg_table_bytes(int T, int max_big, int max_seg) {
	return sizeof(SegHeader) + sizeof(SegBigGroup) * (size_t)max_big + sizeof(SegEntry) * (size_t)max_seg +
	       sizeof(double) * ((size_t)max_big * (size_t)(16 * T + 2) + (size_t)max_seg * (size_t)wide_record_len(T));
}
// workgroup-per-segment path (accumulate_wide.hip): a quarter of the segments fill the chip (4 waves each), and the
// segment records are up to 74 KB
constexpr int kWideSegTarget = 512;
constexpr int kWideSegMaxBig = kWideSegTarget + 8;
constexpr int kWideSegMaxSegments = 2 * kWideSegTarget + 16;
inline __host__ __device__ int64_t wide_seg_rows_for(int64_t n_rows) {
	int64_t s = (n_rows + kWideSegTarget - 1) / kWideSegTarget;
	s = (s + 127) / 128 * 128;
	return s < kSegMinRows ? kSegMinRows : s;
}


#ifdef __HIPCC__
// Reserve `n` consecutive entries of a fixed-capacity device table whose fill count is *counter; -1 when they do not
// fit (the caller then keeps the work for itself).  Never over-commits, so readers may trust every index < *counter.
__device__ __forceinline__ int reserve_table_entries(int32_t *counter, int n, int capacity) {
	int old = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	while (true) {
		if (old + n > capacity) return -1;
		if (__hip_atomic_compare_exchange_strong(counter, &old, old + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
			return old;
	}
}
#endif

// ---- segment tables of the wide-record paths (accumulate_mid.hip, accumulate_wide.hip): device-side accessors and
// the registration of a very large group.  max_big / max_seg are the capacities the table was laid out with.
#ifdef __HIPCC__
__device__ __forceinline__ SegHeader *wseg_header(void *t) { return static_cast<SegHeader *>(t); }
__device__ __forceinline__ SegBigGroup *wseg_big(void *t) { return reinterpret_cast<SegBigGroup *>(wseg_header(t) + 1); }
__device__ __forceinline__ SegEntry *wseg_entries(void *t, int max_big) { return reinterpret_cast<SegEntry *>(wseg_big(t) + max_big); }
__device__ __forceinline__ double *wseg_first(void *t, int max_big, int max_seg) {
	return reinterpret_cast<double *>(wseg_entries(t, max_big) + max_seg);
}
__device__ __forceinline__ double *wseg_records(void *t, int T, int max_big, int max_seg) {
	return wseg_first(t, max_big, max_seg) + (size_t)max_big * (16 * T + 2);
}

// Called by ONE wavefront for a group it will not stream itself: find the group's first valid row (the shift and
// the reference point of the constant-column test: ols.rs:59-87), store it, cut the rows into segments.
// Returns false when the tables are full (the caller then accumulates the group itself).
__device__ inline bool wide_register_big_group(const WideArgs &args, int64_t gl, int64_t lo, int64_t hi, int T, int lane,
                                               int max_big, int max_seg) {
	const int p = args.p;
	const bool weighted = args.model == ANOFOX_HIP_MODEL_WLS;
	SegHeader *h = wseg_header(args.seg_table);
	const int64_t S = args.seg_rows;
	const int nseg = (int)((hi - lo + S - 1) / S);
	int slot = -1, base = -1;
	if (lane == 0) {
		base = reserve_table_entries(&h->seg_total, nseg, max_seg);
		if (base >= 0) slot = reserve_table_entries(&h->big_total, 1, max_big);
	}
	slot = __builtin_amdgcn_readfirstlane(slot);
	base = __builtin_amdgcn_readfirstlane(base);
	if (slot < 0 || base < 0) {
		// (segments reserved without a group slot stay unclaimed: mark them empty)
		if (base >= 0)
			for (int k = lane; k < nseg; k += 64) {
				SegEntry e;
				e.lo = e.hi = lo; e.slot = -1; e.pad = 0;
				wseg_entries(args.seg_table, max_big)[base + k] = e;
			}
		return false;
	}
	int64_t rfirst = -1;
	for (int64_t b0 = lo; b0 < hi && rfirst < 0; b0 += 64) {
		const int64_t r = b0 + lane < hi ? b0 + lane : hi - 1;
		bool ok = (b0 + lane < hi) && isfinite(args.y[r]);
		for (int j = 0; j < p; ++j) ok = ok && isfinite(args.x_table[j][r]);
		if (weighted) {
			const double w = args.w[r];
			ok = ok && isfinite(w) && (w > 0.0);
		}
		const unsigned long long b = __ballot(ok);
		if (b != 0ull) rfirst = b0 + (__ffsll((long long)b) - 1);
	}
	double *ff = wseg_first(args.seg_table, max_big, max_seg) + (size_t)slot * (16 * T + 2);
	for (int j = lane; j <= 16 * T; j += 64) {
		double v = 0.0;
		if (rfirst >= 0) v = j < p ? args.x_table[j][rfirst] : (j == 16 * T ? args.y[rfirst] : 0.0);
		ff[j] = v;
	}
	if (lane == 0) {
		SegBigGroup b;
		b.g = gl; b.base = base; b.nseg = nseg; b.done = 0; b.pad = 0;
		wseg_big(args.seg_table)[slot] = b;
	}
	for (int k = lane; k < nseg; k += 64) {
		SegEntry e;
		e.lo = lo + k * S;
		e.hi = (e.lo + S < hi) ? e.lo + S : hi;
		e.slot = slot; e.pad = 0;
		wseg_entries(args.seg_table, max_big)[base + k] = e;
	}
	return true;
}
#endif

// per-row predictions (predict.hip), any p <= kWideMaxP
struct PredictArgs {
	const int64_t *row_offsets;
	const double *x_table[kWideMaxP];
	const double *core; // [G * (p+6)] fit records
	double *pred;       // [N * 3] {yhat, yhat_lower, yhat_upper}
	double *margin;     // [G] scratch: half-width of the interval per group
	int64_t n_groups;
	int p;
	double confidence_level;
	void *tcrit_table;
	// rows beyond the first seg_rows of a group are handed to extra wavefronts (PredictSegTable, filled by the group's
	// own wave); nullptr disables it
	void *seg_table;
	int64_t seg_rows;
	double avg_rows; // rows per group on average (0 = unknown): small groups share a wavefront
};
struct PredictSegEntry {
	int64_t g, lo, hi;
};
struct PredictSegTable {
	int32_t count;
	int32_t pad[15];
	PredictSegEntry entries[kSegTargetWaves + 16]; // sum over groups of (ceil(n_g / seg_rows) - 1) <= n_rows / seg_rows <= 2048
};
#ifdef __HIPCC__
// Called by a group's own wavefront in the per-row kernels (predict, HC): keeps the first seg_rows rows, hands the
// rest to extra wavefronts through the table.  Returns the end of the rows the caller keeps.
__device__ __forceinline__ int64_t register_overflow_rows(void *table, int64_t seg_rows, int64_t g, int64_t lo, int64_t hi, int lane) {
	if (!table || hi - lo <= seg_rows) return hi;
	PredictSegTable *t = static_cast<PredictSegTable *>(table);
	const int extra = (int)((hi - lo - 1) / seg_rows);
	int base = -1;
	if (lane == 0) base = reserve_table_entries(&t->count, extra, kSegTargetWaves + 16);
	base = __builtin_amdgcn_readfirstlane(base);
	if (base < 0) return hi; // table full (n_rows understated by the caller): this wave keeps every row
	for (int k = lane; k < extra; k += 64) {
		PredictSegEntry e;
		e.g = g;
		e.lo = lo + (k + 1) * seg_rows;
		e.hi = e.lo + seg_rows < hi ? e.lo + seg_rows : hi;
		t->entries[base + k] = e;
	}
	return lo + seg_rows;
}
#endif
hipError_t launch_predict(const PredictArgs &a, hipStream_t stream);
// out[g] = { rss, aic, bic } from the fit records (predict.hip)
hipError_t launch_information_criteria(const double *core, int64_t n_groups, int p, int fit_intercept, int wls, double *out,
                                       hipStream_t stream);

// expanding-window fit + predict (window_narrow.hip), p <= kNarrowMaxP
struct WindowArgs {
	const int64_t *row_offsets;
	const double *y;
	const double *x[kNarrowMaxP];
	const double *w;
	double *pred; // [N * 3]
	int64_t n_groups;
	int p;
	int model;
	int fit_intercept;
	int lambda_scaling;
	double alpha;
	const double *tcrit; // [tcrit_cap + 1], see tcrit_table_kernel
	int tcrit_cap;
	// ROWS BETWEEN frame_start PRECEDING AND frame_end PRECEDING; negative = FOLLOWING; frame_start = kFrameUnbounded =
	// UNBOUNDED PRECEDING, frame_end = -kFrameUnbounded = UNBOUNDED FOLLOWING
	int64_t frame_start;
	int64_t frame_end;
	double avg_rows; // rows per partition on average (0 = unknown)
	// Output rows whose fit was ill-conditioned (smallest Cholesky pivot ratio < 1e-3, or rss / tss < 1e-7: the moment
	// identity has cancelled) are appended here; the host refits them through the virtual-group path (frames.hip),
	// whose refinement passes restore QR-level accuracy.  nullptr = no flagging.
	int32_t *flag_list;
	int32_t *flag_count;
	int32_t flag_cap;
};
constexpr int kWindowTcritCap = 65536;
constexpr int64_t kFrameUnbounded = ANOFOX_HIP_FRAME_UNBOUNDED;
hipError_t launch_tcrit_table(double *table, int cap, double prob, hipStream_t stream);
hipError_t launch_window_predict(const WindowArgs &a, hipStream_t stream);

// residual diagnostics (residuals_narrow.hip), p <= kNarrowMaxP
struct ResidualArgs {
	const int64_t *row_offsets;
	const double *y;
	const double *y_hat;
	const double *x[kNarrowMaxP];
	const double *rse; // one per group, NaN = none; may be null
	double *out;       // [N * 4]
	double *group_out; // [G * 2]
	int64_t n_groups;
	int p;
	int include_studentized;
	int drop_nan_rows;
};
hipError_t launch_residuals_narrow(const ResidualArgs &a, hipStream_t stream);
// residuals_wide.hip: 9 <= p <= 128, one workgroup per group; d_x_table = DEVICE array of the p column pointers
constexpr int kResidualsMaxP = kWideMaxP; // 128
hipError_t launch_residuals_wide(const ResidualArgs &a, const double *const *d_x_table, hipStream_t stream); // 9 .. 128 features

// ---- streaming ingest (ingest.hip): row chunks in arrival order -> per-slot moment records, p <= kNarrowMaxP ----
// A "slot" is one aggregate state (one GROUP BY key of one hash table).  The state keeps the narrow path's moment
// record (MomentLayout<P>) per slot plus the number of rows the aggregate's Update accepted, i.e. what the
// reference buffers as whole rows (src/aggregate_functions/ols_aggregate.cpp:19-42,120-186) reduced to O(p^2).
constexpr int64_t kIngestChunkRows = 1 << 22; // rows folded per pass (bounds the sort buffers); (r4) 2^20 -> 2^22: with rows in random state
                                              // order a state's run in a pass is 4 x longer, its record read and written once per run
constexpr int64_t kIngestStageRows = 1 << 20; // rows per host staging buffer of update_host (PCIe-bound whatever the pass size)
constexpr int kIngestPieceRows = 2048;        // a run longer than this within one chunk is cut into pieces
constexpr int kIngestMaxBig = (int)(kIngestChunkRows / kIngestPieceRows) + 8;
constexpr int kIngestMaxPieces = 2 * (int)(kIngestChunkRows / kIngestPieceRows) + 16;
struct IngestArgs {
	// one chunk of rows in arrival order (device pointers)
	const uint32_t *slot;  // [n] state index of each row
	const double *y;       // [n]
	const double *x;       // [n * p] row-major (DuckDB's LIST(DOUBLE) child)
	const double *w;       // [n] or nullptr
	const uint8_t *valid;  // [n] or nullptr: 0 = the aggregate skips the row (NULL y / NULL x list / NULL weight)
	int64_t n;
	int p;
	int weighted;
	int center; // fit_intercept
	// state
	double *moments;   // [n_slots * moment_record_len(p)]
	int64_t *n_accum;  // [n_slots] rows accepted by Update (the "< 2 rows -> NULL" rule looks at this)
	int64_t n_slots;
	// scratch of one pass
	uint32_t *keys_in, *keys_out, *rows_out; // [n]
	int32_t *run_start, *run_end;            // [n_slots]
	uint32_t *run_list;                      // [n]
	int32_t *counters;                       // [0] runs in this chunk (zeroed per pass)  [1] sticky: a slot index was out of range
	void *piece_table;                       // ingest_piece_table_bytes(p)
	void *sort_temp;
	size_t sort_temp_bytes;
};
size_t ingest_piece_table_bytes(int p);
size_t ingest_sort_temp_bytes(int64_t n);
hipError_t launch_ingest_chunk(const IngestArgs &a, hipStream_t stream);
// merge slot src[i] into slot dst[i] (dst's rows come first, as in OlsAggCombine: ols_aggregate.cpp:189-234) and
// empty src[i]; the dst indices of one call must be distinct
// preserve != 0: the sources keep their records (DuckDB's AggregateCombineType::PRESERVE_INPUT: window segment trees)
hipError_t launch_ingest_combine(double *moments, int64_t *n_accum, int64_t n_slots, const uint32_t *src, const uint32_t *dst,
                                 int64_t n_pairs, int p, int center, int preserve, hipStream_t stream);
hipError_t launch_ingest_gather_slots(const double *moments, const int64_t *n_accum, const uint32_t *list, int64_t n_list, int p, double *out_m,
                                      int64_t *out_n, hipStream_t st);
hipError_t launch_ingest_scatter_slots(double *moments, int64_t *n_accum, const uint32_t *list, int64_t n_list, int p, const double *in_m,
                                       const int64_t *in_n, hipStream_t st);
hipError_t launch_ingest_clear_slots(double *moments, int64_t *n_accum, const uint32_t *list, int64_t n_list, int p, hipStream_t st);

// ---- row log of a streaming aggregate state (rowlog.hip): the rows kept for the refit of queued groups ----
struct RowLogSlab {
	double *x;       // [cap * p] row-major
	double *y;       // [cap]
	double *w;       // [cap] or nullptr
	uint32_t *slot;  // [cap]
	uint8_t *valid;  // [cap]
	int64_t first_row, rows, cap;
	int32_t on_host; // 1: page-locked host memory (the spill beyond the HBM budget), read by the kernels over PCIe
	int32_t pad;
};
// (r4) copies of the rows of preserved Combine sources, labelled with their targets (rowlog.hip)
int64_t rowlog_dup_tiles(int64_t rows);
hipError_t launch_rowlog_dup_count(const RowLogSlab *h_slabs, int n_slabs, const uint32_t *usrc, const int32_t *uoff, const uint32_t *utgt, int m,
                                   int64_t *tile_cnt, int64_t n_tiles, hipStream_t st);
hipError_t launch_rowlog_dup_fill(const RowLogSlab *h_slabs, const int64_t *src_rows, int n_slabs, int p, int weighted, const uint32_t *usrc,
                                  const int32_t *uoff, const uint32_t *utgt, int m, const int64_t *tile_base, const RowLogSlab &dst, int64_t dst_at,
                                  hipStream_t st);
size_t rowlog_sort_temp_bytes(int64_t n);
hipError_t launch_rowlog_sort_slots(const int32_t *in, int32_t *out, int64_t n, void *temp, size_t temp_bytes, hipStream_t st);
hipError_t launch_rowlog_iota(int32_t *v, int64_t n, hipStream_t st); // v[i] = i
hipError_t launch_rowlog_dense(const int32_t *sorted_slots, int64_t k_n, int32_t *dense, int64_t n_slots, hipStream_t st);
bool rowlog_key_bits(int64_t log_rows, int64_t k_n, unsigned *row_bits, unsigned *end_bit); // false: they do not fit one key
hipError_t launch_rowlog_select(bool fill, const uint32_t *slot, const uint8_t *valid, int64_t n, int64_t base_row, const int32_t *dense,
                                int64_t n_slots, unsigned long long *counter, uint64_t *keys, unsigned row_bits, hipStream_t st);
hipError_t launch_rowlog_sort_keys(const uint64_t *in, uint64_t *out, int64_t m, unsigned end_bit, void *temp, size_t temp_bytes, hipStream_t st);
hipError_t launch_rowlog_gather(const uint64_t *keys, int64_t m, int64_t k_n, const RowLogSlab *d_slabs, int n_slabs, int p, int weighted,
                                double *y, double *x_cols, size_t col_stride, double *w, int64_t *offs, unsigned row_bits, hipStream_t st);
// dst row of sorted_slots[k]: the slot itself, or pos[slot] when pos is given (subset Finalize: rows of the caller's list)
hipError_t launch_rowlog_scatter(const double *src, const int32_t *sorted_slots, int64_t k_n, int len, double *dst, const int32_t *pos,
                                 hipStream_t st);
// pos[list[k]] = k, -1 elsewhere
hipError_t launch_rowlog_positions(const uint32_t *list, int64_t n_list, int32_t *pos, int64_t n_slots, hipStream_t st);
// out[i] = sel[queue[i]] for i < *count (the slots behind the subset indices a solve queued); grid-stride, count on the device
hipError_t launch_rowlog_map_queue(const int32_t *queue, const int32_t *count, const uint32_t *sel, int32_t *out, hipStream_t st);
// logged rows of the slots marked in mark[] (one byte per slot) are invalidated (Destroy of their aggregate states)
hipError_t launch_rowlog_invalidate(uint8_t *mark, int64_t n_slots, const uint32_t *list, int64_t n_list, const RowLogSlab *h_slabs, int n_slabs,
                                    hipStream_t st);
// records of the slots list[0 .. *count) (both on the device): every field NaN, status ANOFOX_HIP_STATUS_UNREFINED
hipError_t launch_rowlog_flag_unrefined(const int32_t *list, const int32_t *count, int64_t n_slots, int p, double *core, double *inf,
                                        hipStream_t st);
hipError_t launch_rowlog_remap(uint32_t *remap, int64_t n_slots, const uint32_t *src, const uint32_t *dst, int64_t n_pairs,
                               const RowLogSlab *h_slabs, int n_slabs, hipStream_t st);

// ---- window frames as a batch of virtual groups (frames.hip): any p <= kWideMaxP, any frame ----
struct FrameArgs {
	const double *y;
	const double *x_table[kWideMaxP];
	int64_t n_frames;
	int p;
	int fit_intercept;
	double confidence_level;
	const int64_t *lo, *hi;   // frame e = rows [lo[e], hi[e]); the row predicted is hi[e] - 1
	const int64_t *ynn;       // [n_rows + 1] prefix count of rows whose y is not NaN; nullptr = count each frame directly
	int64_t *rule_counts;     // [n_frames] out: training rows when MORE than p + [intercept] exist, else 0 (-> NULL)
	const double *core;       // [n_frames * (p + 6)] fit records of the frames
	const double *tcrit;      // [tcrit_cap + 1], launch_tcrit_table
	int tcrit_cap;
	double *pred;             // [n_frames * 3]
	const int32_t *list;      // optional: frame e of this batch is output row list[e] (pred index), else e
};
size_t frames_scan_temp_bytes(int64_t n_rows);
hipError_t launch_frames_ynn(const double *y, int64_t n_rows, int64_t *ynn, void *temp, size_t temp_bytes, hipStream_t stream);
// ROWS BETWEEN start_preceding PRECEDING AND end_preceding PRECEDING -> lo / hi per row (clipped to the partition)
// (list != nullptr: bounds of the rows list[0 .. n_list) only, written to lo[k] / hi[k])
hipError_t launch_frames_from_rows_spec(const int64_t *row_offsets, int64_t n_groups, int64_t n_rows, int64_t start_preceding,
                                        int64_t end_preceding, int64_t *lo, int64_t *hi, hipStream_t stream,
                                        const int32_t *list = nullptr, int64_t n_list = 0);
hipError_t launch_frames_rule(const FrameArgs &a, hipStream_t stream);
hipError_t launch_frames_predict(const FrameArgs &a, hipStream_t stream);

hipError_t launch_accumulate_wide(const WideArgs &a, hipStream_t stream);
hipError_t launch_solve_wide(const WideArgs &a, int mode, hipStream_t stream);
hipError_t launch_inference_wide_finish(const WideArgs &a, hipStream_t stream); // after the last solve_wide mode
// solve_tiles.hip: the primary solve (mode 0) with one wavefront per group and the matrix in registers, 32 < p <= 128
bool solve_tiles_supports(int p);
hipError_t launch_solve_tiles(const WideArgs &a, hipStream_t stream);
hipError_t launch_residual_grad_wide(const WideArgs &a, hipStream_t stream);
// refit_dd.hip: the queued groups' records once more, from the rows in double-double (standard errors, R's aliasing rule)
hipError_t launch_refit_dd_wide(const WideArgs &a, hipStream_t stream);
hipError_t launch_refit_dd_narrow(const BatchArgs &a, hipStream_t stream);
// accumulate_mid.hip: wave-per-group accumulation into the same records for 8 < p <= 32
bool accumulate_mid_supports(int p);
hipError_t launch_accumulate_mid(const WideArgs &a, hipStream_t stream);
hipError_t launch_accumulate_mid_segments(const WideArgs &a, hipStream_t stream);
// accumulate_quad.hip: the same on 4 x 4 blocks of v_mfma_f64_4x4x4_4b_f64 (no padding to 16 columns)
bool accumulate_quad_supports(int p, bool weighted, bool center, bool no_fast_path);
hipError_t launch_accumulate_quad(const WideArgs &a, hipStream_t stream);
// accumulate_wide.hip: the segment kernel + the full version on the redo list, behind accumulate_quad's speculative kernel at p = 33, 34
// (r4) accumulate_prefix.hip: the moment records of frames that extend one another (expanding windows), frames_per_block per workgroup
hipError_t launch_accumulate_prefix(const WideArgs &a, int frames_per_block, hipStream_t stream);
hipError_t launch_accumulate_wide_followup(const WideArgs &a, hipStream_t stream);
// accumulate_mid.hip: the full version on the redo list (p = 27 .. 32)
hipError_t launch_accumulate_mid_redo(const WideArgs &a, hipStream_t stream);
// accumulate_mid.hip: the speculative LDS-DMA kernel for three and four column tiles (p = 34 .. 64, no weights, an intercept)
bool accumulate_tile_supports(int p, bool weighted, bool center, bool no_fast_path);
hipError_t launch_accumulate_tile(const WideArgs &a, hipStream_t stream);
// solve_mid.hip: lane-per-group solve on the same records for 8 < p <= 32 (same modes as launch_solve_wide)
bool solve_mid_supports(int p);
hipError_t launch_solve_mid(const WideArgs &a, int mode, hipStream_t stream);
// HC0..HC3 standard errors over the finished fits of this launch's groups (rewrites se/t/p/ci)
hipError_t launch_hc_wide(const WideArgs &a, hipStream_t stream);

// launchers implemented in the .hip translation units
hipError_t launch_accumulate_narrow(const BatchArgs &a, hipStream_t stream);
hipError_t launch_accumulate_narrow_list(const BatchArgs &a, const int32_t *list, const int32_t *count, hipStream_t stream);
// accumulate_small.hip: several small groups per wavefront; groups too long for it are appended to big_list / big_count
int accumulate_small_segment_width(double avg_rows); // 0 = not worth it
hipError_t launch_accumulate_small(const BatchArgs &a, int segw, int32_t *big_list, int32_t *big_count, hipStream_t stream);
// primary solve of every group (queues the groups that need refinement), then ONE launch that takes every queued
// group through `steps` iterative-refinement updates and the final statistics from the directly summed RSS
hipError_t launch_solve_narrow(const BatchArgs &a, hipStream_t stream);
hipError_t launch_refine_fused_narrow(const BatchArgs &a, int steps, hipStream_t stream);
// vif_narrow.hip: out[g] = { vif[p], status }; rows < min_rows -> NULL (status 100)
hipError_t launch_vif_narrow(const double *moments, const int64_t *row_offsets, int64_t n_groups, int p, int64_t min_rows,
                             double *out, hipStream_t stream);
hipError_t launch_vif_from_core(const double *core, const int64_t *row_offsets, int64_t n_groups, int q, int j, int p,
                                int64_t min_rows, double *out, hipStream_t stream);
// hc_narrow.hip: HC0..HC3 standard errors over the finished fit (rewrites se/t/p/ci of the inference records)
size_t hc_prep_bytes(int64_t n_groups, int p); // scratch the pass needs (prep records)
hipError_t launch_hc_narrow(const BatchArgs &a, double *prep, void *overflow_table, hipStream_t stream); // table: PredictSegTable, count zeroed

} // namespace anofox
