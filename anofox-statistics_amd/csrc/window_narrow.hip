// window_narrow.hip — expanding-window fit + predict for the `*_fit_predict` window functions (p <= 8).
//
// The reference implements `anofox_stats_{ols,ridge,wls}_fit_predict(y, x [, w] [, opts]) OVER (...)` as a window
// aggregate (src/window_functions/ols_fit_predict.cpp): the state buffers the frame's training rows (y not NULL,
// :164-190) and remembers the x of the LAST row of the frame (:157-162); Finalize refits from scratch for every
// output row — NULL unless more than p + [intercept] training rows (:257-262) — and predicts that x with
// anofox_predict_with_interval (:296-308).  For the frames of the reference's published benchmark
// (ROWS BETWEEN UNBOUNDED PRECEDING AND CURRENT ROW / 1 PRECEDING, examples/performance_1m_groups/benchmark_ols.sql)
// that is O(n) refits of O(n) rows per partition.
//
// Here: out[e] = prediction for x_e from the fit on rows 0..e of its partition, for every row e, in ONE pass:
// one wavefront per partition, lanes = consecutive rows; each lane forms its row's moment contribution, a wave
// inclusive scan (6 shuffle steps per value) plus a running carry turns them into prefix moments, and every
// lane then solves its own prefix problem in registers (same semantics as solve_narrow.hip: row filter,
// constant / aliased columns, intercept-only shortcut, minimum observations).  A frame ending b PRECEDING is the
// same output written b rows further down (the x that is predicted is the LAST row of the frame, not the current
// row: ols_fit_predict.cpp:157-162).  Loads and stores are fully coalesced; the kernel is f64-VALU bound (one
// p x p solve per row).
//
// Rolling frames (ROWS BETWEEN a PRECEDING AND b PRECEDING, a finite) are summed directly: lane = output row,
// a wave-uniform loop walks the a - b + 1 frame offsets and every lane accumulates the moments of ITS frame,
// shifted by the first training row of that frame.  Differences of prefix moments would be cheaper but cancel
// (a 10-row frame at row 1000 of a trending regressor loses ~9 digits); the loads of neighbouring lanes are
// consecutive rows and hit L1/L2 after the first touch, so the direct sum costs O(frame) cached loads per row.
#include "common.h"
#include "device_math.h"
#include <cstdlib>

namespace anofox {

namespace {

constexpr double kAliasTolX = 1e-11;

// table[0] = normal quantile z at prob, table[i] = Student-t quantile at prob for df = i (1 <= i <= cap)
__global__ __launch_bounds__(256) void tcrit_table_kernel(double *table, int cap, double prob) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i > cap) return;
	const bool valid = prob > 0.5 && prob < 1.0;
	table[i] = !valid ? __builtin_nan("") : (i == 0 ? normcdfinv(prob) : dm_t_quantile_upper(prob, (double)i));
}

// critical value for an integer df: table lookup; beyond the table the Cornish-Fisher series
// (Abramowitz & Stegun 26.7.5) is exact to < 1e-15 relative (df > 65536)
__device__ __forceinline__ double window_tcrit(const WindowArgs &a, double df) {
	const int i = (int)df;
	if (i <= a.tcrit_cap) return a.tcrit[i];
	const double z = a.tcrit[0], z2 = z * z, r = 1.0 / df;
	const double g1 = z * (z2 + 1.0) * 0.25;
	const double g2 = z * ((5.0 * z2 + 16.0) * z2 + 3.0) * (1.0 / 96.0);
	const double g3 = z * (((3.0 * z2 + 19.0) * z2 + 17.0) * z2 - 15.0) * (1.0 / 384.0);
	return z + r * (g1 + r * (g2 + r * g3));
}

template <int P>
struct PrefixFit {
	double coef[P]; // NaN = dropped
	double intercept;
	double rse;
	double nobs;
	bool ok;
	bool suspect; // ill-conditioned or cancelled: to be refitted with refinement (WindowArgs::flag_list)
};

// Called by every lane of the wavefront (convergent); lanes with `flag` append their row: one atomic per wavefront.
__device__ __forceinline__ void window_flag_row(const WindowArgs &a, bool flag, int64_t row) {
	if (!a.flag_list) return;
	const unsigned long long b = __ballot(flag);
	if (b == 0ull) return;
	const int lane = threadIdx.x & 63;
	const int leader = __ffsll((long long)b) - 1;
	int base = 0;
	if (lane == leader) base = atomicAdd(a.flag_count, (int)__popcll(b));
	base = __shfl(base, leader, 64);
	const int k = base + (int)__popcll(b & ((1ull << lane) - 1ull));
	if (flag && k < a.flag_cap) a.flag_list[k] = (int32_t)row;
}

// Fit from the moments of one prefix; rec uses the MomentLayout<P> of the accumulate kernel.
template <int P>
__device__ __forceinline__ void fit_from_moments(const double (&rec)[MomentLayout<P>::REC], int model, bool icpt, double alpha,
                                                 int lambda_scaling, PrefixFit<P> &out) {
	using L = MomentLayout<P>;
	const double nanv = __builtin_nan("");
	out.ok = false;
	out.suspect = false;
	out.intercept = out.rse = out.nobs = nanv;
#pragma unroll
	for (int j = 0; j < P; ++j) out.coef[j] = nanv;
	if (model == ANOFOX_HIP_MODEL_RIDGE && alpha < 0.0) return;       // ridge.rs:38-40
	const double cnt = rec[L::OFF_CNT];
	if (!(cnt > 0.0)) return;                                          // ols.rs:68-70
	const double sw = rec[L::OFF_SW];
	const unsigned mask = (unsigned)rec[L::OFF_MASK];
	const int p_eff = __popc(mask);
	const double sy = rec[L::OFF_S + P];
	const double qyy = rec[L::q_index(P, P)];
	const double cyy_c = qyy - sy * sy / sw;
	const double ymean = (icpt ? rec[L::OFF_FIRST + P] : 0.0) + sy / sw;
	if (p_eff == 0) {                                                   // ols.rs:101-130, wls.rs:119-150
		if (!icpt) return;
		out.ok = true;
		out.intercept = ymean;
		out.rse = (model == ANOFOX_HIP_MODEL_WLS) ? sqrt(cyy_c / sw) : sqrt(cyy_c / (cnt - 1.0));
		out.nobs = cnt;
		return;
	}
	if (cnt < (double)(p_eff + (icpt ? 1 : 0))) return;                 // ols.rs:132-139
	double A[P][P], c[P];
	bool active[P];
#pragma unroll
	for (int i = 0; i < P; ++i) {
		active[i] = (mask >> i) & 1u;
		const double si = rec[L::OFF_S + i];
#pragma unroll
		for (int j = 0; j <= i; ++j) {
			const double qij = rec[L::q_index(j, i)];
			A[i][j] = icpt ? qij - si * rec[L::OFF_S + j] / sw : qij;
		}
		const double qiy = rec[L::q_index(i, P)];
		c[i] = icpt ? qiy - si * sy / sw : qiy;
	}
	const double tss = icpt ? cyy_c : qyy;
	double lam = 0.0;
	if (model == ANOFOX_HIP_MODEL_RIDGE) {
		lam = alpha;
		if (lambda_scaling == ANOFOX_LAMBDA_SCALING_GLMNET) lam = cnt * alpha / sqrt(cyy_c / cnt);
#pragma unroll
		for (int i = 0; i < P; ++i) A[i][i] += lam;
	}
	double diag0[P];
#pragma unroll
	for (int j = 0; j < P; ++j) diag0[j] = A[j][j];
	bool small_pivot = false; // some accepted pivot below 1e-3 of its diagonal entry (no division: this runs per row)
#pragma unroll
	for (int j = 0; j < P; ++j) {
		double d = A[j][j];
#pragma unroll
		for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
		const bool ok = active[j] && (d > kAliasTolX * diag0[j]) && (d > 0.0);
		active[j] = ok;
		small_pivot = small_pivot || (ok && d < 1e-3 * diag0[j]);
		const double ljj = ok ? sqrt(d) : 1.0;
		A[j][j] = ljj;
		const double inv = 1.0 / ljj;
#pragma unroll
		for (int i = j + 1; i < P; ++i) {
			double t = A[i][j];
#pragma unroll
			for (int k = 0; k < j; ++k) t -= A[i][k] * A[j][k];
			A[i][j] = ok ? t * inv : 0.0;
		}
	}
	int rank = 0;
#pragma unroll
	for (int j = 0; j < P; ++j) rank += active[j] ? 1 : 0;
	double zf[P], beta[P];
	double zz = 0.0, bc = 0.0, bb = 0.0;
#pragma unroll
	for (int i = 0; i < P; ++i) {
		double t = c[i];
#pragma unroll
		for (int k = 0; k < i; ++k) t -= A[i][k] * zf[k];
		zf[i] = active[i] ? t / A[i][i] : 0.0;
		zz += zf[i] * zf[i];
	}
#pragma unroll
	for (int i = P - 1; i >= 0; --i) {
		double t = zf[i];
#pragma unroll
		for (int k = i + 1; k < P; ++k) t -= A[k][i] * beta[k];
		beta[i] = active[i] ? t / A[i][i] : 0.0;
		bc += beta[i] * c[i];
		bb += beta[i] * beta[i];
	}
	double rss = (model == ANOFOX_HIP_MODEL_RIDGE) ? tss - bc - lam * bb : tss - zz;
	// the same triggers as the fit path's refinement queue (solve_narrow.hip): the normal equations have squared a
	// large condition number, or rss = tss - |z|^2 has cancelled
	out.suspect = small_pivot || !(rss > 1e-7 * tss);
	if (rss < 0.0) rss = 0.0; // exact fits: the moment identity can round below zero (flagged above)
	const double df = cnt - (double)(rank + (icpt ? 1 : 0));
	double b0 = nanv;
	if (icpt) {
		b0 = ymean;
#pragma unroll
		for (int i = 0; i < P; ++i) b0 -= beta[i] * (rec[L::OFF_FIRST + i] + rec[L::OFF_S + i] / sw);
	}
	out.ok = true;
	out.intercept = b0;
#pragma unroll
	for (int i = 0; i < P; ++i) out.coef[i] = active[i] ? beta[i] : nanv;
	out.rse = sqrt(rss / df);
	out.nobs = cnt;
}

// inclusive prefix within segments of SEGW consecutive lanes (sl = lane % SEGW)
template <int SEGW>
__device__ __forceinline__ double scan_incl(double v, int sl) {
#pragma unroll
	for (int d = 1; d < SEGW; d <<= 1) {
		const double u = __shfl_up(v, d, 64);
		v += (sl >= d) ? u : 0.0;
	}
	return v;
}

template <int SEGW>
__device__ __forceinline__ unsigned scan_or(unsigned v, int sl) {
#pragma unroll
	for (int d = 1; d < SEGW; d <<= 1) {
		const unsigned u = (unsigned)__shfl_up((int)v, d, 64);
		v |= (sl >= d) ? u : 0u;
	}
	return v;
}

__device__ __forceinline__ double rl64(double v, int src) {
	return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
	                        __builtin_amdgcn_readlane(__double2loint(v), src));
}

// the value held by lane `src` (wave-uniform for whole-wave segments, else per segment)
template <int SEGW>
__device__ __forceinline__ double seg_pick(double v, int src) {
	if (SEGW == 64) return rl64(v, __builtin_amdgcn_readfirstlane(src));
	return __shfl(v, src, 64);
}

// Partitions are mapped to segments of SEGW lanes (64 = one partition per wavefront; 8 / 16 for batches of short
// partitions, which would leave most of a wavefront idle): g = the lane's partition, sl = its lane within the segment.
template <int SEGW>
struct SegMap {
	int lane, sl, seg_base;
	int64_t g;
	unsigned long long segmask;
	__device__ __forceinline__ SegMap() {
		lane = threadIdx.x & 63;
		sl = lane & (SEGW - 1);
		seg_base = lane - sl;
		const int64_t wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (int64_t)blockIdx.x * 4;
		g = wave * (64 / SEGW) + lane / SEGW;
		segmask = SEGW == 64 ? ~0ull : (((1ull << (SEGW & 63)) - 1ull) << seg_base);
	}
};

template <int P, bool WEIGHTED, bool CENTER, int SEGW>
__global__ __launch_bounds__(256) void expanding_predict_kernel(WindowArgs args) {
	using L = MomentLayout<P>;
	constexpr int Z = L::Z;
	constexpr int ZZ = L::ZZ;
	const SegMap<SEGW> sm;
	const int lane = sm.lane, sl = sm.sl;
	const bool glive = sm.g < args.n_groups;
	const int64_t lo = glive ? args.row_offsets[sm.g] : 0, hi = glive ? args.row_offsets[sm.g + 1] : 0;
	const bool icpt = CENTER;
	const double nanv = __builtin_nan("");

	// running totals of the rows of all previous tiles (uniform within the segment)
	double car_s[Z], car_q[ZZ], car_w = 0.0, car_n = 0.0, car_ny = 0.0;
	unsigned car_mask = 0;
#pragma unroll
	for (int a = 0; a < Z; ++a) car_s[a] = 0.0;
#pragma unroll
	for (int k = 0; k < ZZ; ++k) car_q[k] = 0.0;
	double first[Z];
#pragma unroll
	for (int a = 0; a < Z; ++a) first[a] = 0.0;
	bool have_first = false;
	double fin[3] = {nanv, nanv, nanv};
	bool fin_suspect = false;

	for (int64_t base = lo; __ballot(base < hi) != 0ull; base += SEGW) { // until every segment of the wave is done
		const int64_t r = base + sl;
		const bool in = r < hi;
		const int64_t rc = in ? r : (hi > 0 ? hi - 1 : 0); // clamped: unconditional loads
		double z[Z];
#pragma unroll
		for (int j = 0; j < P; ++j) z[j] = args.x[j][rc];
		z[P] = args.y[rc];
		double w = 1.0;
		if (WEIGHTED) w = args.w[rc];
		bool xfinite = true;
#pragma unroll
		for (int j = 0; j < P; ++j) xfinite = xfinite && isfinite(z[j]);
		bool valid = in && xfinite && isfinite(z[P]);
		if (WEIGHTED) valid = valid && (w > 0.0) && isfinite(w);
		const unsigned long long bv = __ballot(valid) & sm.segmask;
		// the window aggregate counts every row whose y is not NULL (NaN here) as a training row for its
		// "enough rows" rule, before the fit drops the non-finite ones (ols_fit_predict.cpp:164-190,257-262)
		const unsigned long long by = __ballot(in && !isnan(z[P])) & sm.segmask;
		const double n_y = car_ny + (double)__popcll(by & ((2ull << lane) - 1ull));
		{
			const bool take = !have_first && bv != 0ull;
			const int fl = bv != 0ull ? __ffsll((long long)bv) - 1 : lane;
#pragma unroll
			for (int a = 0; a < Z; ++a) {
				const double zf = seg_pick<SEGW>(z[a], fl);
				first[a] = take ? zf : first[a];
			}
			have_first = have_first || take;
		}
		// this row's contribution (zero when it does not train), then inclusive prefix over the tile + carry
		double rec[L::REC];
		const double ww = valid ? w : 0.0;
		double d[Z];
		unsigned m = 0;
#pragma unroll
		for (int a = 0; a < Z; ++a) {
			const double dev = valid ? z[a] - first[a] : 0.0;
			d[a] = CENTER ? dev : (valid ? z[a] : 0.0);
			if (a < P) m |= !(fabs(dev) < 1e-10) ? (1u << a) : 0u;
		}
#pragma unroll
		for (int a = 0; a < Z; ++a) {
			const double wd = WEIGHTED ? ww * d[a] : d[a];
			rec[L::OFF_S + a] = car_s[a] + scan_incl<SEGW>(wd, sl);
#pragma unroll
			for (int b = a; b < Z; ++b) {
				const int k = a * Z - a * (a - 1) / 2 + (b - a);
				rec[L::OFF_Q + k] = car_q[k] + scan_incl<SEGW>(wd * d[b], sl);
			}
		}
		rec[L::OFF_SW] = car_w + scan_incl<SEGW>(ww, sl);
		const double cnt = car_n + (double)__popcll(bv & ((2ull << lane) - 1ull));
		rec[L::OFF_CNT] = cnt;
		const unsigned mask = car_mask | scan_or<SEGW>(m, sl);
		rec[L::OFF_MASK] = (double)mask;
#pragma unroll
		for (int a = 0; a < Z; ++a) rec[L::OFF_FIRST + a] = first[a];

		// carry for the next tile = the inclusive prefix of the segment's last lane
		const int last = sm.seg_base + SEGW - 1;
#pragma unroll
		for (int a = 0; a < Z; ++a) car_s[a] = seg_pick<SEGW>(rec[L::OFF_S + a], last);
#pragma unroll
		for (int k = 0; k < ZZ; ++k) car_q[k] = seg_pick<SEGW>(rec[L::OFF_Q + k], last);
		car_w = seg_pick<SEGW>(rec[L::OFF_SW], last);
		car_n += (double)__popcll(bv);
		car_ny += (double)__popcll(by);
		car_mask = SEGW == 64 ? (unsigned)__builtin_amdgcn_readlane((int)mask, 63) : (unsigned)__shfl((int)mask, last, 64);

		// Finalize of the window aggregate for the frame ending at this row
		double yhat = nanv, ylo = nanv, yhi = nanv;
		bool suspect = false;
		// (a NaN in this row's x only matters where the fit kept the column: the prediction skips NaN coefficients and is NULL
		// when IT is not finite — lib.rs:2264-2349; an intercept-only fit predicts whatever x holds)
		if (in && n_y > (double)(P + (icpt ? 1 : 0))) {       // ols_fit_predict.cpp:257-262 (strictly more)
			PrefixFit<P> f;
			fit_from_moments<P>(rec, args.model, icpt, args.alpha, args.lambda_scaling, f);
			suspect = f.ok && f.suspect;
			if (f.ok) {
				double v = isnan(f.intercept) ? 0.0 : f.intercept;
#pragma unroll
				for (int j = 0; j < P; ++j) v = fma(isnan(f.coef[j]) ? 0.0 : f.coef[j], isnan(f.coef[j]) ? 0.0 : z[j], v);
				if (isfinite(v)) {
					yhat = ylo = yhi = v;
					// anofox_predict_with_interval: lib.rs:2306-2347
					if (!(isnan(f.rse) || f.rse <= 0.0 || f.nobs <= (double)(P + 1))) {
						const double df = f.nobs - (double)(P + (icpt ? 1 : 0));
						if (df > 0.0) {
							const double tcrit = window_tcrit(args, df);
							if (!isnan(tcrit)) {
								const double margin = tcrit * f.rse * sqrt(1.0 + 1.0 / f.nobs);
								ylo = v - margin;
								yhi = v + margin;
							}
						}
					}
				}
			}
		}
		const int64_t b = args.frame_end; // frame ends b rows before the current row: row r + b gets this result
		window_flag_row(args, suspect && in && r + b < hi && r + b >= lo, r + b);
		if (in) {
			if (r + b < hi && r + b >= lo) { // b < 0 (FOLLOWING): the frame of row r + b ends here
				double *out = args.pred + (r + b) * 3;
				out[0] = yhat;
				out[1] = ylo;
				out[2] = yhi;
			}
			if (r - lo < b) { // empty frame: the aggregate state was never initialised (ols_fit_predict.cpp:253-256)
				double *out = args.pred + r * 3;
				out[0] = out[1] = out[2] = nanv;
			}
		}
		if (b < 0) { // wave-uniform: remember the result of the partition's last row (segment-wide)
			const bool has_last = base < hi && hi - 1 < base + SEGW;
			const int src = sm.seg_base + (has_last ? (int)(hi - 1 - base) : 0);
			const double f0 = seg_pick<SEGW>(yhat, src), f1 = seg_pick<SEGW>(ylo, src), f2 = seg_pick<SEGW>(yhi, src);
			fin[0] = has_last ? f0 : fin[0];
			fin[1] = has_last ? f1 : fin[1];
			fin[2] = has_last ? f2 : fin[2];
			const double fs = seg_pick<SEGW>(suspect ? 1.0 : 0.0, src);
			fin_suspect = has_last ? (fs != 0.0) : fin_suspect;
		}
	}
	if (args.frame_end < 0 && hi > lo) { // frames reaching past the partition's end stop at its last row: same result
		int64_t t0 = hi + args.frame_end;
		if (t0 < lo) t0 = lo;
		for (int64_t t = t0 + sl; t < hi; t += SEGW) {
			double *out = args.pred + t * 3;
			out[0] = fin[0];
			out[1] = fin[1];
			out[2] = fin[2];
		}
		if (fin_suspect) // (segment-uniform; every lane takes part in the ballots of the same number of trips)
			for (int64_t t = t0; __ballot(t < hi) != 0ull; t += SEGW) window_flag_row(args, t + sl < hi, t + sl);
	}
}

// Shared tail of both kernels: Finalize of the window aggregate for one frame (moments in rec), predicting z.
template <int P>
__device__ __forceinline__ void predict_from_moments(const WindowArgs &args, const double (&rec)[MomentLayout<P>::REC], bool icpt,
                                                     const double (&z)[P + 1], double &yhat, double &ylo, double &yhi, bool &suspect) {
	PrefixFit<P> f;
	fit_from_moments<P>(rec, args.model, icpt, args.alpha, args.lambda_scaling, f);
	if (!f.ok) return;
	suspect = f.suspect;
	double v = isnan(f.intercept) ? 0.0 : f.intercept;
#pragma unroll
	for (int j = 0; j < P; ++j) v = fma(isnan(f.coef[j]) ? 0.0 : f.coef[j], isnan(f.coef[j]) ? 0.0 : z[j], v);
	if (!isfinite(v)) return;
	yhat = ylo = yhi = v;
	// anofox_predict_with_interval: lib.rs:2306-2347
	if (isnan(f.rse) || f.rse <= 0.0 || f.nobs <= (double)(P + 1)) return;
	const double df = f.nobs - (double)(P + (icpt ? 1 : 0));
	if (!(df > 0.0)) return;
	const double tcrit = window_tcrit(args, df);
	if (isnan(tcrit)) return;
	const double margin = tcrit * f.rse * sqrt(1.0 + 1.0 / f.nobs);
	ylo = v - margin;
	yhi = v + margin;
}

template <int P, bool WEIGHTED, bool CENTER, int SEGW>
__global__ __launch_bounds__(256) void rolling_predict_kernel(WindowArgs args) {
	using L = MomentLayout<P>;
	constexpr int Z = L::Z;
	constexpr int ZZ = L::ZZ;
	const SegMap<SEGW> sm;
	const int sl = sm.sl;
	const bool glive = sm.g < args.n_groups;
	const int64_t lo = glive ? args.row_offsets[sm.g] : 0, hi = glive ? args.row_offsets[sm.g + 1] : 0;
	const int64_t fa = args.frame_start, fb = args.frame_end; // fa >= fb; negative = FOLLOWING, +-kFrameUnbounded
	const bool icpt = CENTER;
	const double nanv = __builtin_nan("");

	for (int64_t base = lo; __ballot(base < hi) != 0ull; base += SEGW) { // until every segment of the wave is done
		const int64_t e = base + sl;
		const bool in = e < hi;
		double s[Z], q[ZZ], first[Z], z[Z];
#pragma unroll
		for (int a = 0; a < Z; ++a) s[a] = first[a] = z[a] = 0.0;
#pragma unroll
		for (int k = 0; k < ZZ; ++k) q[k] = 0.0;
		double sw = 0.0, cnt = 0.0, n_y = 0.0;
		unsigned mask = 0;
		bool have_first = false, live = false, xfinite = false;

		// this lane's frame = offsets kl..kh (row e - k), clipped to the partition; the loop walks the union over the wave
		const int64_t kh = in ? (fa < e - lo ? fa : e - lo) : INT64_MIN + 1;
		const int64_t kl = in ? (fb > e - (hi - 1) ? fb : e - (hi - 1)) : INT64_MAX;
		// union over the segment's rows in closed form, then over the segments of the wave
		int64_t k_hi = INT64_MIN + 1, k_lo = INT64_MAX;
		if (base < hi) {
			const int64_t e_last = base + (SEGW - 1) < hi - 1 ? base + (SEGW - 1) : hi - 1;
			k_hi = fa < e_last - lo ? fa : e_last - lo;
			k_lo = fb > base - (hi - 1) ? fb : base - (hi - 1);
		}
#pragma unroll
		for (int m = 32; m >= SEGW; m >>= 1) {
			const int64_t oh = __shfl_xor(k_hi, m, 64), ol = __shfl_xor(k_lo, m, 64);
			k_hi = oh > k_hi ? oh : k_hi;
			k_lo = ol < k_lo ? ol : k_lo;
		}
		for (int64_t k = k_hi; k >= k_lo; --k) {
			const int64_t r = e - k;
			live = k <= kh && k >= kl;
			const int64_t rc = r < lo ? lo : (r >= hi ? (hi > 0 ? hi - 1 : 0) : r); // clamped: unconditional loads
#pragma unroll
			for (int j = 0; j < P; ++j) z[j] = args.x[j][rc];
			z[P] = args.y[rc];
			double w = 1.0;
			if (WEIGHTED) w = args.w[rc];
			xfinite = true;
#pragma unroll
			for (int j = 0; j < P; ++j) xfinite = xfinite && isfinite(z[j]);
			bool valid = live && xfinite && isfinite(z[P]);
			if (WEIGHTED) valid = valid && (w > 0.0) && isfinite(w);
			n_y += (live && !isnan(z[P])) ? 1.0 : 0.0; // ols_fit_predict.cpp:164-190: y not NULL makes a training row
			const bool take = valid && !have_first;
#pragma unroll
			for (int a = 0; a < Z; ++a) first[a] = take ? z[a] : first[a];
			have_first = have_first || valid;
			const double ww = valid ? w : 0.0;
			double d[Z];
#pragma unroll
			for (int a = 0; a < Z; ++a) {
				const double dev = valid ? z[a] - first[a] : 0.0;
				d[a] = CENTER ? dev : (valid ? z[a] : 0.0);
				if (a < P) mask |= !(fabs(dev) < 1e-10) ? (1u << a) : 0u;
			}
#pragma unroll
			for (int a = 0; a < Z; ++a) {
				const double wd = WEIGHTED ? ww * d[a] : d[a];
				s[a] += wd;
#pragma unroll
				for (int b = a; b < Z; ++b) {
					const int kk = a * Z - a * (a - 1) / 2 + (b - a);
					q[kk] = fma(wd, d[b], q[kk]);
				}
			}
			sw += ww;
			cnt += valid ? 1.0 : 0.0;
		}
		// the x to predict is the LAST row of the frame (offset kl): read it again (an L1 hit) instead of tracking it
		const bool any_live = in && kh >= kl;
		{
			const int64_t rl = any_live ? e - kl : (hi > 0 ? hi - 1 : 0);
#pragma unroll
			for (int j = 0; j < P; ++j) z[j] = args.x[j][rl];
		}
		double yhat = nanv, ylo = nanv, yhi = nanv;
		bool suspect = false;
		if (any_live && n_y > (double)(P + (icpt ? 1 : 0))) { // ols_fit_predict.cpp:253-262 (x is judged by the prediction itself)
			double rec[L::REC];
#pragma unroll
			for (int a = 0; a < Z; ++a) { rec[L::OFF_S + a] = s[a]; rec[L::OFF_FIRST + a] = first[a]; }
#pragma unroll
			for (int k = 0; k < ZZ; ++k) rec[L::OFF_Q + k] = q[k];
			rec[L::OFF_SW] = sw;
			rec[L::OFF_CNT] = cnt;
			rec[L::OFF_MASK] = (double)mask;
			predict_from_moments<P>(args, rec, icpt, z, yhat, ylo, yhi, suspect);
		}
		if (in) {
			double *out = args.pred + e * 3;
			out[0] = yhat;
			out[1] = ylo;
			out[2] = yhi;
		}
		window_flag_row(args, suspect && in, e);
	}
}

template <int P, int SEGW>
hipError_t launch_window_ps(const WindowArgs &a, hipStream_t stream) {
	const int64_t waves = (a.n_groups + (64 / SEGW) - 1) / (64 / SEGW);
	const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
	const bool weighted = a.model == ANOFOX_HIP_MODEL_WLS;
	const bool center = a.fit_intercept != 0;
	const bool rolling = a.frame_start != kFrameUnbounded;
#define ANOFOX_WINDOW_LAUNCH(KERNEL)                                                                             \
	do {                                                                                                         \
		if (weighted) {                                                                                          \
			if (center) hipLaunchKernelGGL((KERNEL<P, true, true, SEGW>), grid, block, 0, stream, a);            \
			else hipLaunchKernelGGL((KERNEL<P, true, false, SEGW>), grid, block, 0, stream, a);                  \
		} else {                                                                                                 \
			if (center) hipLaunchKernelGGL((KERNEL<P, false, true, SEGW>), grid, block, 0, stream, a);           \
			else hipLaunchKernelGGL((KERNEL<P, false, false, SEGW>), grid, block, 0, stream, a);                 \
		}                                                                                                        \
	} while (0)
	if (rolling) ANOFOX_WINDOW_LAUNCH(rolling_predict_kernel);
	else ANOFOX_WINDOW_LAUNCH(expanding_predict_kernel);
#undef ANOFOX_WINDOW_LAUNCH
	return hipGetLastError();
}

template <int P>
hipError_t launch_window_p(const WindowArgs &a, hipStream_t stream) {
	// Narrow segments waste fewer lanes on a partition's last tile and scan in fewer steps (3 instead of 6), so the
	// expanding kernel prefers 8 lanes per partition whenever there are enough partitions to fill the machine that way
	// (1M x 100 x 3: 3.5 ms vs 4.7 ms with a wavefront each; 4M x 20 x 3: 3.3 ms vs 9.1 ms); few long partitions keep
	// whole wavefronts.  The rolling kernel has no scans: narrow segments only for short partitions.
	int segw = 64;
	if (a.avg_rows > 0.0) {
		if (a.frame_start == kFrameUnbounded) {
			if (a.avg_rows <= 8.0 || a.n_groups >= 32768) segw = 8;
			else if (a.avg_rows <= 16.0 || a.n_groups >= 16384) segw = 16;
		} else {
			if (a.avg_rows <= 24.0) segw = 8;
			else if (a.avg_rows <= 48.0) segw = 16;
		}
	}
	if (const char *e = getenv("ANOFOX_WIN_SEGW")) segw = atoi(e);
	if (segw == 8) return launch_window_ps<P, 8>(a, stream);
	if (segw == 16) return launch_window_ps<P, 16>(a, stream);
	return launch_window_ps<P, 64>(a, stream);
}

} // namespace

hipError_t launch_tcrit_table(double *table, int cap, double prob, hipStream_t stream) {
	hipLaunchKernelGGL(tcrit_table_kernel, dim3((unsigned)((cap + 256) / 256)), dim3(256), 0, stream, table, cap, prob);
	return hipGetLastError();
}

hipError_t launch_window_predict(const WindowArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	if (a.frame_start < a.frame_end || a.frame_start == -kFrameUnbounded || a.frame_end == kFrameUnbounded) return hipErrorInvalidValue;
	switch (a.p) {
	case 1: return launch_window_p<1>(a, stream);
	case 2: return launch_window_p<2>(a, stream);
	case 3: return launch_window_p<3>(a, stream);
	case 4: return launch_window_p<4>(a, stream);
	case 5: return launch_window_p<5>(a, stream);
	case 6: return launch_window_p<6>(a, stream);
	case 7: return launch_window_p<7>(a, stream);
	case 8: return launch_window_p<8>(a, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace anofox
