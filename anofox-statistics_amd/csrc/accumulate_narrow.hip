// accumulate_narrow.hip — the HBM-streaming kernel of the grouped least-squares path (p <= 8).
//
// Replaces, for a whole batch of groups at once, what the reference does per group on the CPU:
// buffer every row (src/aggregate_functions/ols_aggregate.cpp:120-186), copy the buffers twice
// (crates/anofox-stats-ffi/src/lib.rs:128-132, crates/anofox-stats-core/src/models/ols.rs:149-152),
// filter non-finite rows (ols.rs:59-66, wls.rs:76-86), test columns for constancy (ols.rs:76-87) and
// hand the dense design to a QR/SVD (ols.rs:155-161).  Here each group's columns are read from HBM
// exactly once and reduced to the O(p^2) moment record of common.h.
//
// Mapping: one 64-lane wavefront per group, four groups per 256-thread workgroup.  A tile is 128
// consecutive rows; lane l owns rows 2l and 2l+1 of the tile and loads them with one 16-byte access per
// column (the wave reads 1 KiB contiguous per column per tile).  Every lane keeps the whole moment
// triangle (s, q, sw: 55 f64 at p = 8) in VGPRs; rows are shifted by the group's first valid row so the
// one-pass accumulation is as well conditioned as a centred one.  The per-lane partials are summed with a
// transposing butterfly (v_permlane32_swap / v_permlane16_swap, then 4 shuffle steps) that leaves moment
// k on lane k, so the record is written with one coalesced store.
//
// Very large groups: a single wavefront streams at only ~5 GB/s, so a group with more than seg_rows rows
// (>= 1/2048 of the batch) is cut into segments, one wavefront each (accumulate_segments_kernel, a second launch
// that is idle otherwise); the wave that finishes a group's last segment merges the segment records, moving
// each to the group's shift.  1 x 50M x 8 went from 698 ms to the HBM-bound time.
//
// (Measured and removed in round 3: a resident grid whose waves draw groups from a counter and run one loop over the tiles of
// all their groups, requesting the next group's first tile before the current group's reduction — 12.06-12.09 ms against
// 11.96-12.05 ms per 1M x 1000 x 8 step; the group-start latency is not what holds the kernel at 6.0 TB/s.  A draw per
// group (a million atomics on one address) cost 2.4 ms by itself.  profiles/r03_hbm_read_variants.txt.)
//
// Roofline: HBM-bound.  Algorithmic bytes per row 8(p+1) (+8 with weights); 63 f64 VALU ops per row at
// p = 8 (~20 % of the f64 vector rate at the HBM-bound row rate).
#include <stdlib.h>

#include "common.h"
#include "wave_reduce.h"

namespace anofox {

typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));

#ifndef ANOFOX_NARROW_NT
#define ANOFOX_NARROW_NT false
#endif
// 16-byte streaming load; NT = non-temporal (measured: no gain over the default policy, kept for experiments)
template <bool NT>
__device__ __forceinline__ dbl2u load2(const double *p) {
	if (NT) return __builtin_nontemporal_load(reinterpret_cast<const dbl2u *>(p));
	return *reinterpret_cast<const dbl2u *>(p);
}

// the loads of one 128-row tile, in flight or landed: rows 2 lane and 2 lane + 1 of every column
template <int Z>
struct NarrowTile {
	double n0[Z], n1[Z], nw0, nw1;
};

template <int P, bool WEIGHTED>
__device__ __forceinline__ void narrow_load_tile(const BatchArgs &args, int64_t base, int64_t hi, int lane, NarrowTile<P + 1> &t) {
	const int64_t r0 = base + 2 * lane;
	// (measured in round 3 and taken back: loading a group's partial last tile like a full one wherever the arrays reach that
	// far — the rows past the group's end are masked — is neutral in time and reads 2.3 % more from HBM at n = 1000)
	if (base + 128 <= hi) { // full tile: one 16-byte load per column
#pragma unroll
		for (int j = 0; j < P; ++j) {
			const dbl2u v = load2<ANOFOX_NARROW_NT>(args.x[j] + r0);
			t.n0[j] = v.x;
			t.n1[j] = v.y;
		}
		{
			const dbl2u v = load2<ANOFOX_NARROW_NT>(args.y + r0);
			t.n0[P] = v.x;
			t.n1[P] = v.y;
		}
		if (WEIGHTED) {
			const dbl2u v = load2<ANOFOX_NARROW_NT>(args.w + r0);
			t.nw0 = v.x;
			t.nw1 = v.y;
		}
	} else { // ragged tail: unconditional 8-byte loads from clamped (valid) rows; a guarded load per element
		// would sit behind its own branch and wait.  Rows past the end are masked by in0 / in1 in the caller.
		const int64_t c0 = r0 < hi ? r0 : hi - 1, c1 = r0 + 1 < hi ? r0 + 1 : hi - 1;
#pragma unroll
		for (int j = 0; j < P; ++j) {
			t.n0[j] = args.x[j][c0];
			t.n1[j] = args.x[j][c1];
		}
		t.n0[P] = args.y[c0];
		t.n1[P] = args.y[c1];
		if (WEIGHTED) {
			t.nw0 = args.w[c0];
			t.nw1 = args.w[c1];
		}
	}
}

// The per-lane partial moments of one group while its tiles stream through.
template <int P>
struct NarrowAcc {
	static constexpr int Z = P + 1;
	static constexpr int ZZ = Z * (Z + 1) / 2;
	double s[Z];
	double q[ZZ];
	double sw;
	double first[Z]; // wave-uniform: z at the first valid row
	bool have_first;
	int cnt;
	unsigned mask;
	// CENTER: the constant-column test (|x - x_first| < 1e-10 on every valid row, ols.rs:76-87) keeps the largest
	// |d| per column and lane — d is already there, and 0 on rows that do not take part — and is decided once per
	// group; a ballot per column and tile (the version without intercept below) cost 8 vector + 4 scalar
	// instructions per column and tile, a fifth of the kernel's instructions
	double dmax[P];
};

template <int P>
__device__ __forceinline__ void narrow_acc_init(NarrowAcc<P> &c) {
	constexpr int Z = P + 1, ZZ = Z * (Z + 1) / 2;
#pragma unroll
	for (int a = 0; a < Z; ++a) c.s[a] = c.first[a] = 0.0;
#pragma unroll
	for (int k = 0; k < ZZ; ++k) c.q[k] = 0.0;
	c.sw = 0.0;
	c.have_first = false;
	c.cnt = 0;
	c.mask = 0;
#pragma unroll
	for (int j = 0; j < P; ++j) c.dmax[j] = 0.0;
}

// One 128-row tile (this lane: rows r0 and r0 + 1, values z0 / z1, weights w0 / w1) into the partial moments.
template <int P, bool WEIGHTED, bool CENTER>
__device__ __forceinline__ void narrow_tile_compute(NarrowAcc<P> &c, const double (&z0)[P + 1], const double (&z1)[P + 1], double w0, double w1,
                                                    int64_t r0, int64_t hi) {
	constexpr int Z = P + 1;
	const bool in0 = r0 < hi, in1 = r0 + 1 < hi;
	// row filter: everything finite (and w > 0), ols.rs:59-66 / wls.rs:76-86
	bool v0 = in0, v1 = in1;
#pragma unroll
	for (int a = 0; a < Z; ++a) {
		v0 = v0 && isfinite(z0[a]);
		v1 = v1 && isfinite(z1[a]);
	}
	if (WEIGHTED) {
		v0 = v0 && (w0 > 0.0) && isfinite(w0);
		v1 = v1 && (w1 > 0.0) && isfinite(w1);
	}

	const unsigned long long b0 = __ballot(v0);
	const unsigned long long b1 = __ballot(v1);
	const unsigned long long bany = b0 | b1;
	if (bany == 0ull) return; // no valid row in this tile (wave-uniform)

	if (!c.have_first) {
		const int fl = __ffsll((long long)bany) - 1; // lowest lane with a valid row = lowest row index
#pragma unroll
		for (int a = 0; a < Z; ++a) c.first[a] = readlane_f64(v0 ? z0[a] : z1[a], fl);
		c.have_first = true;
	}
	c.cnt += __popcll(b0) + __popcll(b1);

	// constant-column test against the first valid row: |x - x_first| >= 1e-10 anywhere -> not constant
	if (!CENTER) {
#pragma unroll
		for (int j = 0; j < P; ++j) {
			const unsigned long long nc = __ballot((v0 && !(fabs(z0[j] - c.first[j]) < 1e-10)) ||
			                                       (v1 && !(fabs(z1[j] - c.first[j]) < 1e-10)));
			c.mask |= (nc != 0ull) ? (1u << j) : 0u;
		}
	}

	double d0[Z], d1[Z];
#pragma unroll
	for (int a = 0; a < Z; ++a) {
		const double sh = CENTER ? c.first[a] : 0.0;
		d0[a] = v0 ? z0[a] - sh : 0.0;
		d1[a] = v1 ? z1[a] - sh : 0.0;
	}
	if (CENTER) {
#pragma unroll
		for (int j = 0; j < P; ++j) c.dmax[j] = fmax(c.dmax[j], fmax(fabs(d0[j]), fabs(d1[j])));
	}
	const double ww0 = v0 ? w0 : 0.0;
	const double ww1 = v1 ? w1 : 0.0;
	c.sw += ww0 + ww1;
#pragma unroll
	for (int a = 0; a < Z; ++a) {
		const double wd0 = WEIGHTED ? ww0 * d0[a] : d0[a];
		const double wd1 = WEIGHTED ? ww1 * d1[a] : d1[a];
		c.s[a] += wd0 + wd1;
#pragma unroll
		for (int b = a; b < Z; ++b) {
			const int k = a * Z - a * (a - 1) / 2 + (b - a);
			c.q[k] = fma(wd0, d0[b], c.q[k]);
			c.q[k] = fma(wd1, d1[b], c.q[k]);
		}
	}
}

// The group's record: cross-lane sums (transposing butterfly, moment k lands on lane k), then the wave-uniform extras.
template <int P, bool CENTER>
__device__ __forceinline__ void narrow_acc_finish(NarrowAcc<P> &c, double *rec, int lane) {
	using L = MomentLayout<P>;
	constexpr int Z = L::Z;
	constexpr int ZZ = L::ZZ;
	if (CENTER) {
#pragma unroll
		for (int j = 0; j < P; ++j) c.mask |= (__ballot(!(c.dmax[j] < 1e-10)) != 0ull) ? (1u << j) : 0u;
	}
	double v[64];
#pragma unroll
	for (int k = 0; k < 64; ++k) v[k] = 0.0;
#pragma unroll
	for (int a = 0; a < Z; ++a) v[L::OFF_S + a] = c.s[a];
#pragma unroll
	for (int k = 0; k < ZZ; ++k) v[L::OFF_Q + k] = c.q[k];
	v[L::OFF_SW] = c.sw;

	transpose_reduce64(v, lane);

	if (lane < L::KRED) rec[lane] = v[0];

	// wave-uniform extras: first[], cnt, mask
	double e = 0.0;
#pragma unroll
	for (int a = 0; a < Z; ++a) e = (lane == a) ? c.first[a] : e;
	e = (lane == Z) ? (double)c.cnt : e;
	e = (lane == Z + 1) ? (double)c.mask : e;
	if (lane < Z + 2) rec[L::KRED + lane] = e;
}

// The rows [lo, hi) of one group (or of one segment of a very large group) -> one moment record at `rec`.
template <int P, bool WEIGHTED, bool CENTER, bool PREFETCH>
__device__ __forceinline__ void accumulate_rows(const BatchArgs &args, int64_t lo, int64_t hi, double *rec, int lane) {
	constexpr int Z = P + 1;
	NarrowAcc<P> c;
	narrow_acc_init<P>(c);
	NarrowTile<Z> t;
	// the loads of tile t + 1 are issued before the arithmetic of tile t (PREFETCH), so that a wave always has
	// one tile of loads in flight; all loads of a tile sit in one arm of the (wave-uniform) full / ragged branch
	if (PREFETCH && lo < hi) narrow_load_tile<P, WEIGHTED>(args, lo, hi, lane, t);
	for (int64_t base = lo; base < hi; base += 128) {
		if (!PREFETCH) narrow_load_tile<P, WEIGHTED>(args, base, hi, lane, t);
		double z0[Z], z1[Z];
		double w0 = 1.0, w1 = 1.0;
#pragma unroll
		for (int a = 0; a < Z; ++a) {
			z0[a] = t.n0[a];
			z1[a] = t.n1[a];
		}
		if (WEIGHTED) {
			w0 = t.nw0;
			w1 = t.nw1;
		}
		if (PREFETCH && base + 128 < hi) narrow_load_tile<P, WEIGHTED>(args, base + 128, hi, lane, t);
		narrow_tile_compute<P, WEIGHTED, CENTER>(c, z0, z1, w0, w1, base + 2 * lane, hi);
	}
	narrow_acc_finish<P, CENTER>(c, rec, lane);
}

__device__ __forceinline__ SegHeader *seg_header(void *t) { return static_cast<SegHeader *>(t); }
__device__ __forceinline__ SegBigGroup *seg_big(void *t) { return reinterpret_cast<SegBigGroup *>(seg_header(t) + 1); }
__device__ __forceinline__ SegEntry *seg_entries(void *t) { return reinterpret_cast<SegEntry *>(seg_big(t) + kSegMaxBig); }
__device__ __forceinline__ double *seg_records(void *t) { return reinterpret_cast<double *>(seg_entries(t) + kSegMaxSegments); }

// A very large group is handed to accumulate_segments_kernel in pieces; false: it stays with the calling wavefront.
__device__ __forceinline__ bool narrow_register_big_group(const BatchArgs &args, int64_t g, int64_t lo, int64_t hi, int lane) {
	{
		// a single wavefront streams at ~5 GB/s: hand the group to accumulate_segments_kernel in pieces (unless the
		// tables are full, which only happens when the caller understated n_rows: then it stays with this wave)
		SegHeader *h = seg_header(args.seg_table);
		const int64_t S = args.seg_rows;
		const int nseg = (int)((hi - lo + S - 1) / S);
		int slot = -1, base = -1;
		if (lane == 0) {
			base = reserve_table_entries(&h->seg_total, nseg, kSegMaxSegments);
			if (base >= 0) slot = reserve_table_entries(&h->big_total, 1, kSegMaxBig);
		}
		slot = __builtin_amdgcn_readfirstlane(slot);
		base = __builtin_amdgcn_readfirstlane(base);
		if (base >= 0) {
			if (slot >= 0 && lane == 0) {
				SegBigGroup b;
				b.g = g; b.base = base; b.nseg = nseg; b.done = 0; b.pad = 0;
				seg_big(args.seg_table)[slot] = b;
			}
			for (int k = lane; k < nseg; k += 64) {
				SegEntry e;
				e.lo = lo + k * S;
				e.hi = (e.lo + S < hi) ? e.lo + S : hi;
				e.slot = slot; e.pad = 0;
				if (slot < 0) e.hi = e.lo; // reserved without a group slot: empty, unclaimed
				seg_entries(args.seg_table)[base + k] = e;
			}
			if (slot >= 0) return true;
		}
	}
	return false;
}

// One group per wavefront: accumulate it, or register it for row splitting when it is very large.
template <int P, bool WEIGHTED, bool CENTER, bool PF>
__device__ __forceinline__ void accumulate_group(const BatchArgs &args, int64_t g, int lane) {
	using L = MomentLayout<P>;
	const int64_t lo = args.row_offsets[g];
	const int64_t hi = group_row_end(args, g);
	if (args.seg_table && hi - lo > args.seg_rows && narrow_register_big_group(args, g, lo, hi, lane)) return;
	accumulate_rows<P, WEIGHTED, CENTER, PF>(args, lo, hi, args.moments + g * (int64_t)L::REC, lane);
}

template <int P, bool WEIGHTED, bool CENTER, bool PF>
__global__ __launch_bounds__(256) void accumulate_narrow_kernel(BatchArgs args) {
	const int lane = threadIdx.x & 63;
	const int64_t g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (int64_t)blockIdx.x * 4;
	if (g >= args.n_groups) return;
	accumulate_group<P, WEIGHTED, CENTER, PF>(args, g, lane);
}

// The same for an explicit list of groups (the ones accumulate_small.hip leaves out): persistent grid.
template <int P, bool WEIGHTED, bool CENTER, bool PF>
__global__ __launch_bounds__(256) void accumulate_narrow_list_kernel(BatchArgs args, const int32_t *list, const int32_t *count) {
	const int lane = threadIdx.x & 63;
	const int n = *count;
	const int n_waves = (int)gridDim.x * 4;
	for (int v = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); v < n; v += n_waves)
		accumulate_group<P, WEIGHTED, CENTER, PF>(args, list[v], lane);
}

// Merge the segment records of one group into its moment record: lane k owns moment k.  Segment b was shifted by
// ITS first valid row; with delta = first_b - first_group:  s += s_b + sw_b delta,
// q_ij += q_b,ij + delta_i s_b,j + delta_j s_b,i + sw_b delta_i delta_j  (no shift without an intercept).
template <int P, bool CENTER>
__device__ void merge_segments(const double *seg_rec, int nseg, double *rec, int lane) {
	using L = MomentLayout<P>;
	constexpr int Z = L::Z;
	// this lane's moment: s_i (lane < Z), q_ij, or sw
	int mi = 0, mj = 0;
	const bool is_s = lane < Z, is_q = lane >= L::OFF_Q && lane < L::OFF_SW, is_sw = lane == L::OFF_SW;
	if (is_s) mi = mj = lane;
	if (is_q) {
		int k = lane - L::OFF_Q;
		for (int a = 0; a < Z; ++a) {
			if (k < Z - a) { mi = a; mj = a + k; break; }
			k -= Z - a;
		}
	}
	double total = 0.0, cnt = 0.0, ai = 0.0, aj = 0.0;
	double a_first = 0.0; // lane a < Z keeps first_group[a]
	unsigned mask = 0;
	bool have = false;
	for (int t = 0; t < nseg; ++t) {
		const double *rb = seg_rec + (int64_t)t * L::REC;
		const double cb = rb[L::OFF_CNT];
		if (!(cb > 0.0)) continue; // wave-uniform
		if (!have) {
			ai = rb[L::OFF_FIRST + mi];
			aj = rb[L::OFF_FIRST + mj];
			a_first = lane < Z ? rb[L::OFF_FIRST + lane] : 0.0;
			have = true;
		}
		const double swb = rb[L::OFF_SW];
		const double di = CENTER ? rb[L::OFF_FIRST + mi] - ai : 0.0;
		const double dj = CENTER ? rb[L::OFF_FIRST + mj] - aj : 0.0;
		double v = 0.0;
		if (is_s) v = rb[L::OFF_S + mi] + swb * di;
		else if (is_q) v = rb[lane] + di * rb[L::OFF_S + mj] + dj * rb[L::OFF_S + mi] + swb * di * dj;
		else if (is_sw) v = swb;
		total += v;
		cnt += cb;
		// constant-column test against the GROUP's first row: varies inside the segment, or the segment sits elsewhere
		unsigned m = (unsigned)rb[L::OFF_MASK];
		const bool moved = lane < P && !(fabs(rb[L::OFF_FIRST + lane] - a_first) < 1e-10);
		m |= (unsigned)__ballot(moved);
		mask |= m;
	}
	if (lane < L::KRED) rec[lane] = total;
	double e = a_first;
	e = (lane == Z) ? cnt : e;
	e = (lane == Z + 1) ? (double)(mask & ((1u << P) - 1u)) : e;
	if (lane < Z + 2) rec[L::KRED + lane] = e;
}

// One wavefront per registered segment; the wave that completes a group's last segment merges them.
template <int P, bool WEIGHTED, bool CENTER, bool PF>
__global__ __launch_bounds__(256) void accumulate_segments_kernel(BatchArgs args) {
	using L = MomentLayout<P>;
	const int lane = threadIdx.x & 63;
	const int v = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (int)blockIdx.x * 4;
	SegHeader *h = seg_header(args.seg_table);
	const int total = h->seg_total; // reservations never exceed the capacity
	if (v >= total) return;
	const SegEntry e = seg_entries(args.seg_table)[v];
	if (e.slot < 0) return; // reserved but unclaimed
	double *recs = seg_records(args.seg_table);
	accumulate_rows<P, WEIGHTED, CENTER, PF>(args, e.lo, e.hi, recs + (int64_t)v * L::REC, lane);
	__threadfence(); // this segment's record before the counter
	SegBigGroup *b = seg_big(args.seg_table) + e.slot;
	int old = 0;
	if (lane == 0) old = atomicAdd(&b->done, 1);
	old = __builtin_amdgcn_readfirstlane(old);
	if (old != b->nseg - 1) return;
	__threadfence(); // every other segment's record after the counter
	merge_segments<P, CENTER>(recs + (int64_t)b->base * L::REC, b->nseg, args.moments + b->g * (int64_t)L::REC, lane);
}

template <int P, bool PF>
static hipError_t launch_pn(const BatchArgs &a, hipStream_t stream) {
	const dim3 block(256);
	const dim3 grid((unsigned)((a.n_groups + 3) / 4));
	const dim3 seg_grid((unsigned)((kSegMaxSegments + 3) / 4)); // idle unless some group exceeded seg_rows
	const bool weighted = a.model == ANOFOX_HIP_MODEL_WLS;
	const bool center = a.fit_intercept != 0;
#define ANOFOX_ACC_LAUNCH(W, C)                                                                               \
	do {                                                                                                      \
		hipLaunchKernelGGL((accumulate_narrow_kernel<P, W, C, PF>), grid, block, 0, stream, a);               \
		if (a.seg_table) hipLaunchKernelGGL((accumulate_segments_kernel<P, W, C, PF>), seg_grid, block, 0, stream, a); \
	} while (0)
	if (weighted) {
		if (center) ANOFOX_ACC_LAUNCH(true, true);
		else ANOFOX_ACC_LAUNCH(true, false);
	} else {
		if (center) ANOFOX_ACC_LAUNCH(false, true);
		else ANOFOX_ACC_LAUNCH(false, false);
	}
#undef ANOFOX_ACC_LAUNCH
	return hipGetLastError();
}

template <int P, bool PF>
static hipError_t launch_list_pn(const BatchArgs &a, const int32_t *list, const int32_t *count, hipStream_t stream) {
	const dim3 block(256), grid(2048);
	const dim3 seg_grid((unsigned)((kSegMaxSegments + 3) / 4));
	const bool weighted = a.model == ANOFOX_HIP_MODEL_WLS;
	const bool center = a.fit_intercept != 0;
#define ANOFOX_ACC_LIST_LAUNCH(W, C)                                                                          \
	do {                                                                                                      \
		hipLaunchKernelGGL((accumulate_narrow_list_kernel<P, W, C, PF>), grid, block, 0, stream, a, list, count); \
		if (a.seg_table) hipLaunchKernelGGL((accumulate_segments_kernel<P, W, C, PF>), seg_grid, block, 0, stream, a); \
	} while (0)
	if (weighted) {
		if (center) ANOFOX_ACC_LIST_LAUNCH(true, true);
		else ANOFOX_ACC_LIST_LAUNCH(true, false);
	} else {
		if (center) ANOFOX_ACC_LIST_LAUNCH(false, true);
		else ANOFOX_ACC_LIST_LAUNCH(false, false);
	}
#undef ANOFOX_ACC_LIST_LAUNCH
	return hipGetLastError();
}

template <int P>
static hipError_t launch_p(const BatchArgs &a, const int32_t *list, const int32_t *count, hipStream_t stream) {
	// ANOFOX_ACC_PF=0 issues a tile's loads at the top of its own iteration instead of one tile ahead (A/B measurements)
	static const int pf = [] { const char *e = getenv("ANOFOX_ACC_PF"); return e ? atoi(e) : 1; }();
	if (list) return pf ? launch_list_pn<P, true>(a, list, count, stream) : launch_list_pn<P, false>(a, list, count, stream);
	return pf ? launch_pn<P, true>(a, stream) : launch_pn<P, false>(a, stream);
}

static hipError_t launch_any(const BatchArgs &a, const int32_t *list, const int32_t *count, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	switch (a.p) {
	case 1: return launch_p<1>(a, list, count, stream);
	case 2: return launch_p<2>(a, list, count, stream);
	case 3: return launch_p<3>(a, list, count, stream);
	case 4: return launch_p<4>(a, list, count, stream);
	case 5: return launch_p<5>(a, list, count, stream);
	case 6: return launch_p<6>(a, list, count, stream);
	case 7: return launch_p<7>(a, list, count, stream);
	case 8: return launch_p<8>(a, list, count, stream);
	default: return hipErrorInvalidValue;
	}
}

hipError_t launch_accumulate_narrow(const BatchArgs &a, hipStream_t stream) { return launch_any(a, nullptr, nullptr, stream); }

// only the groups in list[0 .. *count) (device memory), e.g. the ones accumulate_small.hip left out
hipError_t launch_accumulate_narrow_list(const BatchArgs &a, const int32_t *list, const int32_t *count, hipStream_t stream) {
	return launch_any(a, list, count, stream);
}

} // namespace anofox
