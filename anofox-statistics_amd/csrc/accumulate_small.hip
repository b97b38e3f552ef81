// accumulate_small.hip — moment accumulation for SMALL groups (p <= 8): several groups per wavefront.
//
// accumulate_narrow.hip gives every group a whole wavefront and a 128-row tile: with 20 rows per group 84 % of the
// lanes idle and the kernel falls to 0.7 TB/s.  GROUP BY workloads with tens of rows per key are the common case
// (the reference's own benchmark has 100), so when the batch averages at most 128 rows per group this kernel runs
// instead: a wavefront is cut into segments of SEGW = 8, 16 or 32 lanes, one group per segment, every lane owns two
// rows of its group per pass (a pass = 2 SEGW rows).  Same record, same semantics (row filter ols.rs:59-66 /
// wls.rs:76-86, constant-column test ols.rs:76-87, shift by the first valid row):
//   * loads are unconditional 8-byte loads from clamped rows (segments start and end at different rows);
//   * ballots are taken wave-wide and every lane looks at its own segment's bits;
//   * the transposing butterfly of accumulate_narrow.hip stops at the segment width: after log2(SEGW) fold steps
//     every lane holds 64 / SEGW fully reduced moments of its group;
//   * groups that need more than kSmallMaxPasses passes are not accumulated here: their numbers go to a list that
//     the one-wave-per-group kernel then works off (which also splits the very large ones).
#include "common.h"

namespace anofox {

namespace {

constexpr int kSmallMaxPasses = 4;

__device__ __forceinline__ double sm_fold16(double a, double b) {
	auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
	auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
	return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// lanes with (lane & M) == 0 end up with a summed over the pair {l, l ^ M}, the others with b
template <int M>
__device__ __forceinline__ double sm_fold_shfl(double a, double b, int lane) {
	const bool upper = (lane & M) != 0;
	const double keep = upper ? b : a;
	const double send = upper ? a : b;
	return keep + __shfl_xor(send, M, 64);
}

template <int P, bool WEIGHTED, bool CENTER, int SEGW>
__global__ __launch_bounds__(256) void accumulate_small_kernel(BatchArgs args, int32_t *big_list, int32_t *big_count) {
	using L = MomentLayout<P>;
	constexpr int Z = L::Z;
	constexpr int ZZ = L::ZZ;
	constexpr int GPW = 64 / SEGW;     // groups per wavefront
	constexpr int PASS = 2 * SEGW;     // rows per pass and group
	constexpr unsigned long long SEGMASK = (SEGW == 32) ? 0xFFFFFFFFull : (SEGW == 16 ? 0xFFFFull : 0xFFull);

	const int lane = threadIdx.x & 63;
	const int seg = lane / SEGW, sl = lane % SEGW;
	const int64_t wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (int64_t)blockIdx.x * 4;
	const int64_t g = wave_id * GPW + seg;
	const bool live_g = g < args.n_groups;
	int64_t lo = 0, hi = 0;
	if (live_g) {
		lo = args.row_offsets[g];
		hi = args.row_offsets[g + 1];
	}
	bool mine = live_g;
	if (live_g && hi - lo > (int64_t)kSmallMaxPasses * PASS) { // too long for this kernel: hand over
		if (sl == 0) big_list[atomicAdd(big_count, 1)] = (int32_t)g;
		mine = false;
		hi = lo;
	}
	// wave-uniform number of passes
	int npass = mine ? (int)((hi - lo + PASS - 1) / PASS) : 0;
#pragma unroll
	for (int m = 32; m >= SEGW; m >>= 1) npass = max(npass, __shfl_xor(npass, m, 64));
	npass = __builtin_amdgcn_readfirstlane(npass);
	if (__ballot(live_g) == 0ull) return;

	double s[Z], q[ZZ], first[Z];
	double sw = 0.0;
#pragma unroll
	for (int a = 0; a < Z; ++a) s[a] = first[a] = 0.0;
#pragma unroll
	for (int k = 0; k < ZZ; ++k) q[k] = 0.0;
	bool have_first = false;
	int cnt = 0;
	unsigned mask = 0;
	const int sshift = seg * SEGW;

	for (int pass = 0; pass < npass; ++pass) {
		const int64_t r0 = lo + (int64_t)pass * PASS + 2 * sl;
		const bool in0 = r0 < hi, in1 = r0 + 1 < hi;
		const int64_t last = hi > lo ? hi - 1 : 0; // (empty segments read row 0 of the batch: masked)
		const int64_t c0 = in0 ? r0 : last, c1 = in1 ? r0 + 1 : last;
		double z0[Z], z1[Z], w0 = 1.0, w1 = 1.0;
#pragma unroll
		for (int j = 0; j < P; ++j) {
			z0[j] = args.x[j][c0];
			z1[j] = args.x[j][c1];
		}
		z0[P] = args.y[c0];
		z1[P] = args.y[c1];
		if (WEIGHTED) {
			w0 = args.w[c0];
			w1 = args.w[c1];
		}
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
		const unsigned long long b0 = (__ballot(v0) >> sshift) & SEGMASK; // this segment's rows
		const unsigned long long b1 = (__ballot(v1) >> sshift) & SEGMASK;
		const unsigned long long bany = b0 | b1;
		// the first valid row of a group: the lowest lane of its segment with a valid row
		const bool need_first = !have_first && bany != 0ull;
		if (__ballot(need_first) != 0ull) {
			const int fl = sshift + (bany != 0ull ? __ffsll((long long)bany) - 1 : 0);
#pragma unroll
			for (int a = 0; a < Z; ++a) {
				const double cand = __shfl(v0 ? z0[a] : z1[a], fl, 64);
				first[a] = need_first ? cand : first[a];
			}
			have_first = have_first || need_first;
		}
		cnt += __popcll(b0) + __popcll(b1);
		// constant-column test against the first valid row: |x - x_first| >= 1e-10 anywhere -> not constant
#pragma unroll
		for (int j = 0; j < P; ++j) {
			const unsigned long long nc = (__ballot((v0 && !(fabs(z0[j] - first[j]) < 1e-10)) ||
			                                        (v1 && !(fabs(z1[j] - first[j]) < 1e-10))) >> sshift) & SEGMASK;
			mask |= (nc != 0ull) ? (1u << j) : 0u;
		}
		double d0[Z], d1[Z];
#pragma unroll
		for (int a = 0; a < Z; ++a) {
			const double sh = CENTER ? first[a] : 0.0;
			d0[a] = v0 ? z0[a] - sh : 0.0;
			d1[a] = v1 ? z1[a] - sh : 0.0;
		}
		const double ww0 = v0 ? w0 : 0.0;
		const double ww1 = v1 ? w1 : 0.0;
		sw += ww0 + ww1;
#pragma unroll
		for (int a = 0; a < Z; ++a) {
			const double wd0 = WEIGHTED ? ww0 * d0[a] : d0[a];
			const double wd1 = WEIGHTED ? ww1 * d1[a] : d1[a];
			s[a] += wd0 + wd1;
#pragma unroll
			for (int b = a; b < Z; ++b) {
				const int k = a * Z - a * (a - 1) / 2 + (b - a);
				q[k] = fma(wd0, d0[b], q[k]);
				q[k] = fma(wd1, d1[b], q[k]);
			}
		}
	}

	// ---- reduction inside the segment: the butterfly of accumulate_narrow.hip without its widest steps; with
	// NV = 64 / SEGW, moment sl + SEGW * t of the group ends up in v[t * SEGW] of lane sl ----
	double v[64];
#pragma unroll
	for (int k = 0; k < 64; ++k) v[k] = 0.0;
#pragma unroll
	for (int a = 0; a < Z; ++a) v[L::OFF_S + a] = s[a];
#pragma unroll
	for (int k = 0; k < ZZ; ++k) v[L::OFF_Q + k] = q[k];
	v[L::OFF_SW] = sw;
#pragma unroll
	for (int t = 0; t < GPW; ++t) {
		const int o = t * SEGW; // sub-array [o, o + SEGW)
		if (SEGW == 32) {
#pragma unroll
			for (int i = 0; i < 16; ++i) v[o + i] = sm_fold16(v[o + i], v[o + i + 16]);
		}
		if (SEGW >= 16) {
#pragma unroll
			for (int i = 0; i < 8; ++i) v[o + i] = sm_fold_shfl<8>(v[o + i], v[o + i + 8], lane);
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) v[o + i] = sm_fold_shfl<4>(v[o + i], v[o + i + 4], lane);
#pragma unroll
		for (int i = 0; i < 2; ++i) v[o + i] = sm_fold_shfl<2>(v[o + i], v[o + i + 2], lane);
		v[o] = sm_fold_shfl<1>(v[o], v[o + 1], lane);
	}
	if (mine) {
		double *rec = args.moments + g * (int64_t)L::REC;
#pragma unroll
		for (int t = 0; t < GPW; ++t) {
			const int k = sl + SEGW * t;
			if (k < L::KRED) rec[k] = v[t * SEGW];
		}
#pragma unroll
		for (int x0 = 0; x0 < Z + 2; x0 += SEGW) { // first[], cnt, mask: more entries than an 8-lane segment has lanes
			const int x = x0 + sl;
			double e = 0.0;
#pragma unroll
			for (int a = 0; a < Z; ++a) e = (x == a) ? first[a] : e;
			e = (x == Z) ? (double)cnt : e;
			e = (x == Z + 1) ? (double)mask : e;
			if (x < Z + 2) rec[L::KRED + x] = e;
		}
	}
}

template <int P, int SEGW>
hipError_t launch_small_ps(const BatchArgs &a, int32_t *big_list, int32_t *big_count, hipStream_t stream) {
	constexpr int GPW = 64 / SEGW;
	const int64_t waves = (a.n_groups + GPW - 1) / GPW;
	const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
	const bool weighted = a.model == ANOFOX_HIP_MODEL_WLS;
	const bool center = a.fit_intercept != 0;
	if (weighted) {
		if (center) hipLaunchKernelGGL((accumulate_small_kernel<P, true, true, SEGW>), grid, block, 0, stream, a, big_list, big_count);
		else hipLaunchKernelGGL((accumulate_small_kernel<P, true, false, SEGW>), grid, block, 0, stream, a, big_list, big_count);
	} else {
		if (center) hipLaunchKernelGGL((accumulate_small_kernel<P, false, true, SEGW>), grid, block, 0, stream, a, big_list, big_count);
		else hipLaunchKernelGGL((accumulate_small_kernel<P, false, false, SEGW>), grid, block, 0, stream, a, big_list, big_count);
	}
	return hipGetLastError();
}

template <int P>
hipError_t launch_small_p(const BatchArgs &a, int segw, int32_t *big_list, int32_t *big_count, hipStream_t stream) {
	if (segw == 8) return launch_small_ps<P, 8>(a, big_list, big_count, stream);
	return segw == 16 ? launch_small_ps<P, 16>(a, big_list, big_count, stream) : launch_small_ps<P, 32>(a, big_list, big_count, stream);
}

} // namespace

// segment width for a batch that averages `avg_rows` rows per group; 0 = use the one-wave-per-group kernel
int accumulate_small_segment_width(double avg_rows) {
	if (avg_rows <= 16.0) return 8;
	if (avg_rows <= 32.0) return 16;
	if (avg_rows <= 128.0) return 32;
	return 0;
}

hipError_t launch_accumulate_small(const BatchArgs &a, int segw, int32_t *big_list, int32_t *big_count, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	switch (a.p) {
	case 1: return launch_small_p<1>(a, segw, big_list, big_count, stream);
	case 2: return launch_small_p<2>(a, segw, big_list, big_count, stream);
	case 3: return launch_small_p<3>(a, segw, big_list, big_count, stream);
	case 4: return launch_small_p<4>(a, segw, big_list, big_count, stream);
	case 5: return launch_small_p<5>(a, segw, big_list, big_count, stream);
	case 6: return launch_small_p<6>(a, segw, big_list, big_count, stream);
	case 7: return launch_small_p<7>(a, segw, big_list, big_count, stream);
	case 8: return launch_small_p<8>(a, segw, big_list, big_count, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace anofox
