// solve_mid.hip — per-group solve and diagnostics for moderately wide designs (8 < p <= 32), one LANE per group.
//
// Same contract and the same arithmetic rules as solve_narrow.hip / solve_wide.hip (the reference's pre-checks
// and shortcuts of crates/anofox-stats-core/src/models/ols.rs:68-139, ridge.rs:38-40, wls.rs:119-157; the
// regressor's closed forms; NaN re-expansion ols.rs:167-171,191-206) on the tile-major
// moment records of the wide accumulate kernel.  The workgroup-per-group solve of solve_wide.hip pays ~25
// barriers and a dozen serialized phases per group whatever its size — 58 us per group at p = 16 — which made
// p = 9 seven times slower than p = 8.  Here every lane factors its own (p x p) matrix with plain loops; the
// matrix lives in the lane's private memory (scratch: lane-interleaved, so the 64 lanes of a wave touch
// consecutive words), the per-group work is O(p^3 / 3) fused multiply-adds and there is no synchronisation at all.
// Queued groups (small pivot ratio or RSS/TSS < 1e-7) take the same refinement passes: MODE 1 = one step of
// iterative refinement from residual_grad_wide_kernel's gradient, MODE 2 = final statistics from the summed RSS.
#include "common.h"
#include "device_math.h"

namespace anofox {

namespace {

constexpr int kMidMaxP = 32;
constexpr double kAliasTolM = 1e-11;
constexpr double kRefineTolM = 1e-7;
constexpr double kPivotWarnM = 1e-3;
enum { MODE_PRIMARY = 0, MODE_UPDATE = 1, MODE_FINAL = 2 };

__device__ __forceinline__ double nan64m() { return __builtin_nan(""); }
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; } // j <= i

template <int MODE>
__device__ void solve_mid_one(const WideArgs &args, int64_t gl) {
	const int p = args.p;
	const int T = wide_tiles(p), P16 = 16 * T, NT = T * (T + 1) / 2;
	const bool icpt = args.fit_intercept != 0;
	const int model = args.model;
	const int64_t g = args.group_base + gl;
	const double *rec = args.moments + gl * (int64_t)wide_record_len(T);
	const double *vec = rec + (int64_t)NT * 256;
	const double *sx = vec, *sxy = vec + P16, *fx = vec + 2 * P16, *nonconst = vec + 3 * P16;
	const double *sc = vec + 4 * P16;
	double *core = args.core + g * (int64_t)(p + 6);
	double *inf = (args.inference && args.compute_inference) ? args.inference + g * (int64_t)(5 * p + 2) : nullptr;
	const double *rvec = args.refine_vec + g * (int64_t)refine_vec_len(p); // {sum w r^2, sum w r, X'Wr, centred yy}
	const int64_t nrows = args.rule_counts ? args.rule_counts[g] : args.row_offsets[g + 1] - args.row_offsets[g];

	auto write_null = [&](int status, bool core_too) {
		if (core_too)
			for (int k = 0; k < p + 6; ++k) core[k] = (k == p + 5) ? (double)status : nan64m();
		if (inf)
			for (int k = 0; k < 5 * p + 2; ++k) inf[k] = nan64m();
	};

	const double sy = sc[0], syy = sc[1], sw = sc[2], cnt = sc[3], first_y = sc[4];
	if (nrows < 2) { write_null(ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS, true); return; }                          // ols_aggregate.cpp:263-267
	if (model == ANOFOX_HIP_MODEL_RIDGE && args.alpha < 0.0) { write_null(ANOFOX_ERROR_INVALID_ALPHA, true); return; } // ridge.rs:38-40
	if (!(cnt > 0.0)) { write_null(ANOFOX_ERROR_NO_VALID_DATA, true); return; }                                 // ols.rs:68-70

	unsigned active = 0;
	for (int j = 0; j < p; ++j) active |= (nonconst[j] != 0.0) ? (1u << j) : 0u;
	const int peff = __popc(active);
	const double cyy_c = syy - sy * sy / sw;
	const double ymean = (icpt ? first_y : 0.0) + sy / sw;
	if (peff == 0) { // ols.rs:101-130, wls.rs:119-150
		if (!icpt) { write_null(ANOFOX_ERROR_INSUFFICIENT_DATA, true); return; }
		write_null(0, false); // inference: None
		for (int k = 0; k < p; ++k) core[k] = nan64m();
		core[p] = ymean;
		core[p + 1] = 0.0;
		core[p + 2] = 0.0;
		core[p + 3] = (model == ANOFOX_HIP_MODEL_WLS) ? sqrt(cyy_c / sw) : sqrt(cyy_c / (cnt - 1.0));
		core[p + 4] = cnt;
		core[p + 5] = 0.0;
		return;
	}
	if (cnt < (double)(peff + (icpt ? 1 : 0))) { write_null(ANOFOX_ERROR_INSUFFICIENT_DATA, true); return; } // ols.rs:132-139

	// ridge penalty: `lam` goes into the primary factor; the refinement modes factor with and aim at `lam_rows`, glmnet's lambda with sd_y re-summed
	// over the rows about the mean (uncentred moments of a nearly constant y cancel; such groups are queued)
	double lam = 0.0, lam_rows = 0.0;
	bool glmnet_cancels = false;
	if (model == ANOFOX_HIP_MODEL_RIDGE) {
		lam = lam_rows = args.alpha;
		if (args.lambda_scaling == ANOFOX_LAMBDA_SCALING_GLMNET) {
			lam = lam_rows = cnt * args.alpha / sqrt(cyy_c / cnt);
			glmnet_cancels = !icpt && !(cyy_c * kGlmnetCancelRatio > syy);
			if (MODE != MODE_PRIMARY && !icpt) lam_rows = cnt * args.alpha / sqrt(rvec[p + 2] / cnt);
		}
	}
	const double tss = icpt ? cyy_c : syy;

	// centred moment matrix (lower triangle, packed) and right-hand side
	double A[kMidMaxP * (kMidMaxP + 1) / 2];
	double c[kMidMaxP], diag0[kMidMaxP];
	const double inv_sw = 1.0 / sw;
	for (int i = 0; i < p; ++i) {
		const int J = i >> 4, ci = i & 15;
		const double si = sx[i];
		for (int j = 0; j <= i; ++j) {
			const int I = j >> 4;
			const int tile = I * T - I * (I - 1) / 2 + (J - I);
			double v = rec[(int64_t)tile * 256 + (j & 15) * 16 + ci]; // M[16I + r][16J + c], r = j & 15, c = i & 15
			if (icpt) v -= si * sx[j] * inv_sw;
			if (i == j) v += (MODE == MODE_PRIMARY) ? lam : lam_rows; // the refinement modes factor with the re-summed lambda
			A[tri(i, j)] = v;
		}
		c[i] = icpt ? sxy[i] - si * sy * inv_sw : sxy[i];
		diag0[i] = A[tri(i, i)];
	}

	// Cholesky (left-looking, in place), deactivating constant and aliased columns
	double min_ratio = 1.0;
	bool band = false; // (r4) a column dropped with a pivot that is not clearly zero: queued, the refit decides (solve_narrow.hip)
	for (int j = 0; j < p; ++j) {
		double d = A[tri(j, j)];
		for (int k = 0; k < j; ++k) d -= A[tri(j, k)] * A[tri(j, k)];
		const bool ok = ((active >> j) & 1u) && (d > kAliasTolM * diag0[j]) && (d > 0.0);
		band = band || (((active >> j) & 1u) && !ok && d > 1e-13 * diag0[j]);
		if (!ok) active &= ~(1u << j);
		if (ok) min_ratio = fmin(min_ratio, d / diag0[j]);
		const double ljj = ok ? sqrt(d) : 1.0;
		const double inv = 1.0 / ljj;
		A[tri(j, j)] = ljj;
		for (int i = j + 1; i < p; ++i) {
			double t = A[tri(i, j)];
			for (int k = 0; k < j; ++k) t -= A[tri(i, k)] * A[tri(j, k)];
			A[tri(i, j)] = ok ? t * inv : 0.0;
		}
		if (!ok)
			for (int k = 0; k < j; ++k) A[tri(j, k)] = 0.0;
	}
	const int rank = __popc(active);

	// L zf = rhs, L' x = zf
	auto solve_llt = [&](const double *rhs, double *zf, double *x) {
		for (int i = 0; i < p; ++i) {
			double t = rhs[i];
			for (int k = 0; k < i; ++k) t -= A[tri(i, k)] * zf[k];
			zf[i] = ((active >> i) & 1u) ? t / A[tri(i, i)] : 0.0;
		}
		for (int i = p - 1; i >= 0; --i) {
			double t = zf[i];
			for (int k = i + 1; k < p; ++k) t -= A[tri(k, i)] * x[k];
			x[i] = ((active >> i) & 1u) ? t / A[tri(i, i)] : 0.0;
		}
	};

	double beta[kMidMaxP], zf[kMidMaxP];
	double rss;
	bool refine = false;
	if (MODE == MODE_PRIMARY) {
		solve_llt(c, zf, beta);
		double zz = 0.0, bc = 0.0, bb = 0.0;
		for (int i = 0; i < p; ++i) {
			zz += zf[i] * zf[i];
			bc += beta[i] * c[i];
			bb += beta[i] * beta[i];
		}
		rss = (model == ANOFOX_HIP_MODEL_RIDGE) ? tss - bc - lam * bb : tss - zz;
		refine = !(rss > kRefineTolM * tss) || (min_ratio < kPivotWarnM) || glmnet_cancels || band;
		double bmax = 0.0;
		for (int i = 0; i < p; ++i) bmax = fmax(bmax, ((active >> i) & 1u) ? fabs(beta[i]) : 0.0);
		for (int i = 0; i < p; ++i) refine = refine || (((active >> i) & 1u) && coef_bound_weak(beta[i], bmax, diag0[i], tss, min_ratio));
	} else {
		for (int i = 0; i < p; ++i) beta[i] = ((active >> i) & 1u) ? core[i] : 0.0; // residual_grad used exactly these
		rss = rvec[0];
	}
	if (MODE == MODE_UPDATE) {
		// gradient of the (penalised) objective at beta, in centred coordinates
		const double gs = rvec[1];
		double gc[kMidMaxP], delta[kMidMaxP];
		for (int i = 0; i < p; ++i) {
			double gi = rvec[2 + i];
			if (icpt) gi -= (sx[i] * inv_sw) * gs;
			gc[i] = ((active >> i) & 1u) ? gi - lam_rows * beta[i] : 0.0;
		}
		solve_llt(gc, zf, delta);
		for (int i = 0; i < p; ++i) beta[i] += delta[i];
	}

	double b0 = nan64m();
	if (icpt) {
		b0 = ymean;
		for (int i = 0; i < p; ++i) b0 -= beta[i] * (fx[i] + sx[i] * inv_sw);
	}
	if (MODE == MODE_UPDATE) { // only the coefficients change in this pass
		for (int j = 0; j < p; ++j) core[j] = ((active >> j) & 1u) ? beta[j] : nan64m();
		core[p] = b0;
		return;
	}
	const int n_par = rank + (icpt ? 1 : 0);
	const double df = cnt - (double)n_par;
	const double dfm = (double)rank;
	const double r2 = 1.0 - rss / tss;
	const double fstat = ((tss - rss) / dfm) / (rss / df);
	for (int j = 0; j < p; ++j) core[j] = ((active >> j) & 1u) ? beta[j] : nan64m();
	core[p] = b0;
	core[p + 1] = r2;
	core[p + 2] = 1.0 - (1.0 - r2) * (cnt - (icpt ? 1.0 : 0.0)) / df;
	core[p + 3] = sqrt(rss / df);
	core[p + 4] = cnt;
	core[p + 5] = 0.0;

	if (inf) {
		const double sigma2 = rss / df;
		const double tcrit = dm_tcrit_cached(static_cast<TcritSlot *>(args.tcrit_table), 0.5 * (1.0 + args.confidence_level), df);
		// diag of (L L')^-1 through the columns of L^-1
		for (int j = 0; j < p; ++j) {
			double se = nan64m(), tv = nan64m(), pv = nan64m(), lo = nan64m(), hi = nan64m();
			if ((active >> j) & 1u) {
				double wcol[kMidMaxP];
				double dj = 0.0;
				for (int i = j; i < p; ++i) {
					double t = (i == j) ? 1.0 : 0.0;
					for (int k = j; k < i; ++k) t -= A[tri(i, k)] * wcol[k];
					wcol[i] = ((active >> i) & 1u) ? t / A[tri(i, i)] : 0.0;
					dj += wcol[i] * wcol[i];
				}
				se = sqrt(sigma2 * dj);
				tv = beta[j] / se;
				pv = dm_t_two_sided_p(tv, df);
				lo = beta[j] - tcrit * se;
				hi = beta[j] + tcrit * se;
			}
			inf[j] = se;
			inf[p + j] = tv;
			inf[2 * p + j] = pv;
			inf[3 * p + j] = lo;
			inf[4 * p + j] = hi;
		}
		inf[5 * p] = fstat;
		inf[5 * p + 1] = dm_f_sf(fstat, dfm, df);
	}
	if (MODE == MODE_PRIMARY && refine) {
		const int slot = atomicAdd(args.refine_count, 1);
		args.refine_list[slot] = (int32_t)gl;
	}
}

__global__ __launch_bounds__(64) void solve_mid_kernel(WideArgs args) {
	const int64_t gl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (gl >= args.n_groups) return;
	solve_mid_one<MODE_PRIMARY>(args, gl);
}

template <int MODE>
__global__ __launch_bounds__(64) void solve_mid_refine_kernel(WideArgs args) {
	const int n = *args.refine_count;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) solve_mid_one<MODE>(args, args.refine_list[i]);
}

} // namespace

bool solve_mid_supports(int p) { return p > kNarrowMaxP && p <= kMidMaxP; }

hipError_t launch_solve_mid(const WideArgs &a, int mode, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	if (!solve_mid_supports(a.p)) return hipErrorInvalidValue;
	if (mode == MODE_PRIMARY) {
		hipLaunchKernelGGL(solve_mid_kernel, dim3((unsigned)((a.n_groups + 63) / 64)), dim3(64), 0, stream, a);
	} else if (mode == MODE_UPDATE) {
		hipLaunchKernelGGL((solve_mid_refine_kernel<MODE_UPDATE>), dim3(256), dim3(64), 0, stream, a);
	} else {
		hipLaunchKernelGGL((solve_mid_refine_kernel<MODE_FINAL>), dim3(256), dim3(64), 0, stream, a);
	}
	return hipGetLastError();
}

} // namespace anofox
