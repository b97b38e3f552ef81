// solve_narrow.hip — per-group solve and diagnostics (p <= 8), one lane per group.
//
// Consumes the moment records of accumulate_narrow.hip and produces the reference's result STRUCT fields
// (src/aggregate_functions/ols_aggregate.cpp:74-96) as dense records.  Restates, per group:
//   - the model's pre-checks and shortcuts: crates/anofox-stats-core/src/models/ols.rs:68-139,
//     ridge.rs:38-40,104-146, wls.rs:119-157
//   - the regressor (anofox-regression OlsRegressor / RidgeRegressor / WlsRegressor, call sites
//     ols.rs:155-161, ridge.rs:156-164, wls.rs:174-182): here a Cholesky factorisation of the centred
//     (shifted) moment matrix with aliased-pivot detection, two triangular solves, and the closed forms for
//     R^2, adjusted R^2, sigma, SE, t, p, CI and F
//   - NaN re-expansion at dropped columns: ols.rs:167-171,191-206
//
// Normal equations square the condition number, and RSS = Syy - b'Sxy cancels when R^2 -> 1.  Groups where
// either matters (smallest Cholesky pivot ratio < 1e-3, or RSS/TSS < 1e-7) are queued on the device; for
// those groups only, refine_fused_kernel re-reads the rows (one wavefront per queued group) and forms the
// residuals r = y - b0 - x'b, their weighted sum of squares and the gradient X'Wr directly from the data, and
// the solve runs again in
//   MODE 1: one step of iterative refinement  b += (X'WX)^-1 X'Wr   (twice), then
//   MODE 2: final statistics from the directly summed RSS.
// This restores the accuracy of a QR on the design (the reference's algorithm class) for the queued groups.
#include "common.h"
#include "device_math.h"
#include "dd_arith.h"

namespace anofox {

static_assert(sizeof(TcritSlot) * kTcritSlots == kTcritTableBytes, "t memo does not fill its workspace slice");

namespace {

constexpr double kAliasTol = 1e-11;   // pivot / original diagonal below this => column aliased (collinear)
constexpr double kAliasBand = 1e-13;  // ... and above this: not rounding noise — the refit applies the reference's rule
constexpr double kRefineTol = 1e-7;   // RSS / TSS below this => recompute RSS from residuals
constexpr double kPivotWarn = 1e-3;   // smallest pivot ratio below this => iterative refinement

__device__ __forceinline__ double nan64() { return __builtin_nan(""); }

enum { MODE_PRIMARY = 0, MODE_UPDATE = 1, MODE_FINAL = 2 };

template <int P, int MODE>
__device__ void solve_one(const BatchArgs &args, int64_t g) {
	using L = MomentLayout<P>;
	constexpr int Z = L::Z;
	const int p = P;
	const bool icpt = args.fit_intercept != 0;
	const int model = args.model;

	double *core = args.core + g * (int64_t)(p + 6);
	double *inf = (args.inference && args.compute_inference) ? args.inference + g * (int64_t)(5 * p + 2) : nullptr;
	const double *rv = args.refine_vec + g * (int64_t)refine_vec_len(p); // [rss, sum w r, X'Wr, centred yy] from residual_grad_wave

	int status = ANOFOX_ERROR_SUCCESS;
	double coef[P];
	double se[P], tv[P], pv[P], cl[P], cu[P];
#pragma unroll
	for (int j = 0; j < P; ++j) coef[j] = se[j] = tv[j] = pv[j] = cl[j] = cu[j] = nan64();
	double intercept = nan64(), r2 = nan64(), adj = nan64(), rse = nan64(), fstat = nan64(), fp = nan64();
	double nobs = nan64();
	bool has_inf = false;
	bool refine = false;

	const double *rec = args.moments + g * (int64_t)L::REC;
	const int64_t nrows = args.rule_counts ? args.rule_counts[g] : args.row_offsets[g + 1] - args.row_offsets[g];

	do {
		if (nrows < 2) { status = ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS; break; }           // ols_aggregate.cpp:263-267
		if (model == ANOFOX_HIP_MODEL_RIDGE && args.alpha < 0.0) { status = ANOFOX_ERROR_INVALID_ALPHA; break; } // ridge.rs:38-40
		const double cnt = rec[L::OFF_CNT];
		if (!(cnt > 0.0)) { status = ANOFOX_ERROR_NO_VALID_DATA; break; }                 // ols.rs:68-70
		const double sw = rec[L::OFF_SW];
		const unsigned mask = (unsigned)rec[L::OFF_MASK];
		const int p_eff = __popc(mask);

		double s[Z], first[Z];
#pragma unroll
		for (int a = 0; a < Z; ++a) { s[a] = rec[L::OFF_S + a]; first[a] = rec[L::OFF_FIRST + a]; }
		const double qyy = rec[L::q_index(P, P)];
		// centred second moment of y about its (weighted) mean; the accumulate kernel shifts only when an
		// intercept is fitted, the identity holds either way
		const double cyy_centred = qyy - s[P] * s[P] / sw;
		const double ymean = (icpt ? first[P] : 0.0) + s[P] / sw;

		if (p_eff == 0) { // ols.rs:101-130, wls.rs:119-150
			if (!icpt) { status = ANOFOX_ERROR_INSUFFICIENT_DATA; break; }
			intercept = ymean;
			r2 = 0.0;
			adj = 0.0;
			rse = (model == ANOFOX_HIP_MODEL_WLS) ? sqrt(cyy_centred / sw) : sqrt(cyy_centred / (cnt - 1.0));
			nobs = cnt;
			break; // inference: None
		}
		if (cnt < (double)(p_eff + (icpt ? 1 : 0))) { status = ANOFOX_ERROR_INSUFFICIENT_DATA; break; } // ols.rs:132-139

		// moment matrix of the kept columns: centred when an intercept is fitted, raw otherwise
		double A[P][P]; // lower triangle used
		double c[P];
		bool active[P];
#pragma unroll
		for (int i = 0; i < P; ++i) {
			active[i] = (mask >> i) & 1u;
#pragma unroll
			for (int j = 0; j <= i; ++j) {
				const double qij = rec[L::q_index(j, i)];
				A[i][j] = icpt ? qij - s[i] * s[j] / sw : qij;
			}
			const double qiy = rec[L::q_index(i, P)];
			c[i] = icpt ? qiy - s[i] * s[P] / sw : qiy;
		}
		const double tss = icpt ? cyy_centred : qyy;

		double lam = 0.0, lam_rows = 0.0; // the penalty in the factor / the penalty the refinement aims at
		bool glmnet_cancels = false;
		if (model == ANOFOX_HIP_MODEL_RIDGE) {
			lam = args.alpha;
			if (args.lambda_scaling == ANOFOX_LAMBDA_SCALING_GLMNET) {
				// sd_y from the moments; a queued group's passes over the rows re-sum it about the mean (uncentred moments of
				// a nearly constant y cancel), and the refinement modes factor with and aim at that lambda (the standard errors
				// come from the same matrix: two nearly equal y values gave a lambda 3e-5 off and standard errors 1.6e-5 off)
				lam = cnt * args.alpha / sqrt(cyy_centred / cnt);
				glmnet_cancels = !icpt && !(cyy_centred * kGlmnetCancelRatio > qyy);
				if (MODE != MODE_PRIMARY && !icpt) lam_rows = cnt * args.alpha / sqrt(rv[p + 2] / cnt);
				else lam_rows = lam;
			} else {
				lam_rows = lam;
			}
#pragma unroll
			for (int i = 0; i < P; ++i) A[i][i] += (MODE == MODE_PRIMARY) ? lam : lam_rows; // the refinement modes factor with the re-summed lambda
		}

		// Cholesky (left-looking, in place), deactivating constant and aliased columns
		double diag0[P];
#pragma unroll
		for (int j = 0; j < P; ++j) diag0[j] = A[j][j];
		double min_ratio = 1.0;
		// (r4) a non-constant column dropped with a pivot above the rounding noise of the moments (1e-13 .. 1e-11 of the diagonal: sin of
		// its angle to the earlier columns 3e-7 .. 3e-6) may be one the reference's rule (remaining norm >= 1e-7 of the column's norm)
		// keeps: the group is queued, and the double-double refit decides with that rule (refit_dd.hip).  Exact copies and dummy-variable
		// traps leave a pivot of rounding noise (1e-16 .. 1e-13 of the diagonal) and are NOT queued: a batch in which every group carries
		// one must not pay the refinement passes for it (tests/test_gpu_parity.py::test_exactly_aliased_columns_are_not_queued).
		bool band = false;
#pragma unroll
		for (int j = 0; j < P; ++j) {
			double d = A[j][j];
#pragma unroll
			for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
			const bool ok = active[j] && (d > kAliasTol * diag0[j]) && (d > 0.0);
			band = band || (active[j] && !ok && d > kAliasBand * diag0[j]);
			active[j] = ok;
			if (ok) min_ratio = fmin(min_ratio, d / diag0[j]);
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
			if (!ok) {
#pragma unroll
				for (int k = 0; k < j; ++k) A[j][k] = 0.0;
			}
		}
		int rank = 0;
#pragma unroll
		for (int j = 0; j < P; ++j) rank += active[j] ? 1 : 0;

		// L zf = rhs, L' x = zf
		auto solve_llt = [&](const double (&rhs)[P], double (&zf)[P], double (&x)[P]) {
#pragma unroll
			for (int i = 0; i < P; ++i) {
				double t = rhs[i];
#pragma unroll
				for (int k = 0; k < i; ++k) t -= A[i][k] * zf[k];
				zf[i] = active[i] ? t / A[i][i] : 0.0;
			}
#pragma unroll
			for (int i = P - 1; i >= 0; --i) {
				double t = zf[i];
#pragma unroll
				for (int k = i + 1; k < P; ++k) t -= A[k][i] * x[k];
				x[i] = active[i] ? t / A[i][i] : 0.0;
			}
		};

		double beta[P];
		double rss;
		if (MODE == MODE_PRIMARY) {
			double zf[P];
			solve_llt(c, zf, beta);
			double zz = 0.0, bc = 0.0, bb = 0.0;
#pragma unroll
			for (int i = 0; i < P; ++i) { zz += zf[i] * zf[i]; bc += beta[i] * c[i]; bb += beta[i] * beta[i]; }
			rss = (model == ANOFOX_HIP_MODEL_RIDGE) ? tss - bc - lam * bb : tss - zz;
			refine = !(rss > kRefineTol * tss) || (min_ratio < kPivotWarn) || glmnet_cancels || band;
			double bmax = 0.0;
#pragma unroll
			for (int i = 0; i < P; ++i) bmax = fmax(bmax, active[i] ? fabs(beta[i]) : 0.0);
#pragma unroll
			for (int i = 0; i < P; ++i) refine = refine || (active[i] && coef_bound_weak(beta[i], bmax, diag0[i], tss, min_ratio));
		} else {
			// current coefficients come from the record; residual_grad_wave used exactly these
#pragma unroll
			for (int i = 0; i < P; ++i) {
				const double b = core[i];
				beta[i] = active[i] ? b : 0.0;
			}
			rss = rv[0];
		}

		if (MODE == MODE_UPDATE) {
			// gradient of the (penalised) objective at beta, in centred coordinates
			const double gs = rv[1];
			double gc[P], u[P], delta[P];
#pragma unroll
			for (int i = 0; i < P; ++i) {
				double gi = rv[2 + i];
				if (icpt) gi -= (s[i] / sw) * gs;
				gc[i] = active[i] ? gi - lam_rows * beta[i] : 0.0;
			}
			solve_llt(gc, u, delta);
#pragma unroll
			for (int i = 0; i < P; ++i) beta[i] += delta[i];
		}

		const int n_par = rank + (icpt ? 1 : 0);
		const double df = cnt - (double)n_par;
		const double dfm = (double)rank;

		double b0 = 0.0;
		if (icpt) {
			b0 = ymean;
#pragma unroll
			for (int i = 0; i < P; ++i) b0 -= beta[i] * (first[i] + s[i] / sw);
			intercept = b0;
		}
#pragma unroll
		for (int i = 0; i < P; ++i) coef[i] = active[i] ? beta[i] : nan64();
		if (MODE == MODE_UPDATE) { // only the coefficients change in this pass
#pragma unroll
			for (int j = 0; j < P; ++j) core[j] = coef[j];
			core[p] = intercept;
			return;
		}
		r2 = 1.0 - rss / tss;
		adj = 1.0 - (1.0 - r2) * (cnt - (icpt ? 1.0 : 0.0)) / df;
		rse = sqrt(rss / df);
		nobs = cnt;
		fstat = ((tss - rss) / dfm) / (rss / df);

		if (inf) {
			has_inf = true;
			fp = dm_f_sf(fstat, dfm, df);
			const double sigma2 = rss / df;
			const double tcrit = dm_tcrit_cached(static_cast<TcritSlot *>(args.tcrit_table), 0.5 * (1.0 + args.confidence_level), df);
			// diag of (L L')^-1 through the columns of L^-1
#pragma unroll
			for (int j = 0; j < P; ++j) {
				double wcol[P];
				double dj = 0.0;
#pragma unroll
				for (int i = j; i < P; ++i) {
					double t = (i == j) ? 1.0 : 0.0;
#pragma unroll
					for (int k = j; k < i; ++k) t -= A[i][k] * wcol[k];
					wcol[i] = active[i] ? t / A[i][i] : 0.0;
					dj += wcol[i] * wcol[i];
				}
				if (active[j]) {
					se[j] = sqrt(sigma2 * dj);
					tv[j] = beta[j] / se[j];
					pv[j] = dm_t_two_sided_p(tv[j], df);
					cl[j] = beta[j] - tcrit * se[j];
					cu[j] = beta[j] + tcrit * se[j];
				}
			}
		}
	} while (false);

	if (MODE == MODE_UPDATE) return; // queued groups always have status 0; nothing else to write

	if (status != ANOFOX_ERROR_SUCCESS) {
#pragma unroll
		for (int j = 0; j < P; ++j) coef[j] = nan64();
		intercept = r2 = adj = rse = nobs = nan64();
	}
#pragma unroll
	for (int j = 0; j < P; ++j) core[j] = coef[j];
	core[p] = intercept;
	core[p + 1] = r2;
	core[p + 2] = adj;
	core[p + 3] = rse;
	core[p + 4] = nobs;
	core[p + 5] = (double)status;
	if (inf) {
		if (!has_inf) {
#pragma unroll
			for (int j = 0; j < P; ++j) se[j] = tv[j] = pv[j] = cl[j] = cu[j] = nan64();
			fstat = fp = nan64();
		}
#pragma unroll
		for (int j = 0; j < P; ++j) {
			inf[j] = se[j];
			inf[p + j] = tv[j];
			inf[2 * p + j] = pv[j];
			inf[3 * p + j] = cl[j];
			inf[4 * p + j] = cu[j];
		}
		inf[5 * p] = fstat;
		inf[5 * p + 1] = fp;
	}
	if (MODE == MODE_PRIMARY && refine && status == ANOFOX_ERROR_SUCCESS) {
		const int slot = atomicAdd(args.refine_count, 1);
		args.refine_list[slot] = (int32_t)g;
	}
}

template <int P>
__global__ __launch_bounds__(64) void solve_narrow_kernel(BatchArgs args) {
	const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= args.n_groups) return;
	solve_one<P, MODE_PRIMARY>(args, g);
}

// One wavefront per queued group, straight from the data with the record's current coefficients:
//   refine_vec[g] = { sum w r^2, sum w r, sum w r (x_j - shift_j) ..., sum w (y - ybar)^2 },  r = y - b0 - x'b
// over the valid rows (same row filter as the accumulate kernel); shift = first valid row when an intercept
// is fitted (the shift of the moment record), 0 otherwise.  The residual and the gradient sums are formed in
// double-double arithmetic (dd_arith.h), as on the wide path: with the residual in working precision the update stalls
// at cond(X) eps — designs without an intercept whose columns sit far from zero reach cond 1e7 at p <= 8 (3 of 240 000
// cases of the deep narrow sweep were 1.5 .. 2.5e-9 off).  Only queued groups pay for it.
__device__ __forceinline__ void residual_grad_wave(const BatchArgs &args, int64_t g, int lane) {
#pragma clang fp contract(off)
	const int p = args.p;
	const bool weighted = args.model == ANOFOX_HIP_MODEL_WLS;
	const int Z = p + 1;
	const int off_first = Z + Z * (Z + 1) / 2 + 1;
	const int rec_len = moment_record_len(p);
	const double *core = args.core + g * (int64_t)(p + 6);
	const double *rec = args.moments + g * (int64_t)rec_len;
	// acc: [0] sum w r, [1 + j] sum w r (x_j - shift_j), as (hi, lo) pairs; rss and the centred yy in working precision
	double b[kNarrowMaxP], sh[kNarrowMaxP], acc_h[kNarrowMaxP + 1], acc_l[kNarrowMaxP + 1];
	// mean of y over the valid rows, from the record (sum / weight, plus the shift an intercept fit accumulates about)
	const double ybar = rec[p] / rec[Z + Z * (Z + 1) / 2] + (args.fit_intercept ? rec[off_first + p] : 0.0);
#pragma unroll
	for (int j = 0; j < kNarrowMaxP; ++j) {
		b[j] = sh[j] = 0.0;
		if (j < p) {
			const double bj = core[j];
			b[j] = isnan(bj) ? 0.0 : bj; // dropped / aliased columns do not enter the fit
			sh[j] = args.fit_intercept ? rec[off_first + j] : 0.0;
		}
	}
#pragma unroll
	for (int k = 0; k < kNarrowMaxP + 1; ++k) acc_h[k] = acc_l[k] = 0.0;
	double rss = 0.0, cyy = 0.0;
	const double b0 = args.fit_intercept ? core[p] : 0.0;
	const int64_t lo = args.row_offsets[g], hi = group_row_end(args, g);
	for (int64_t r = lo + lane; r < hi; r += 64) {
		const double yv = args.y[r];
		bool ok = isfinite(yv);
		double fh = b0, fl = 0.0; // fit = fh + fl
		double xv[kNarrowMaxP];
#pragma unroll
		for (int j = 0; j < kNarrowMaxP; ++j) {
			xv[j] = 0.0;
			if (j < p) {
				xv[j] = args.x[j][r];
				ok = ok && isfinite(xv[j]);
				dd_fit_term(fh, fl, b[j], xv[j]);
			}
		}
		double wv = 1.0;
		if (weighted) {
			wv = args.w[r];
			ok = ok && (wv > 0.0) && isfinite(wv);
		}
		if (ok) {
			double e, wh, wl;
			dd_weighted_residual(yv, fh, fl, wv, e, wh, wl); // wh + wl = w (y - fit) to twice the working precision
			rss = fma(wh, e, rss);
			dd_add(acc_h[0], acc_l[0], wh, wl);
#pragma unroll
			for (int j = 0; j < kNarrowMaxP; ++j)
				if (j < p) dd_add_scaled_diff(acc_h[1 + j], acc_l[1 + j], wh, wl, xv[j], sh[j]);
			const double dy = yv - ybar;
			cyy = fma(wv * dy, dy, cyy);
		}
	}
	for (int m = 32; m >= 1; m >>= 1) {
		rss += __shfl_xor(rss, m, 64);
		cyy += __shfl_xor(cyy, m, 64);
#pragma unroll
		for (int k = 0; k < kNarrowMaxP + 1; ++k)
			if (k < p + 1) dd_add(acc_h[k], acc_l[k], __shfl_xor(acc_h[k], m, 64), __shfl_xor(acc_l[k], m, 64));
	}
	double *out = args.refine_vec + g * (int64_t)refine_vec_len(p);
	double mine = rss;
#pragma unroll
	for (int k = 0; k < kNarrowMaxP + 1; ++k) mine = (lane == 1 + k) ? acc_h[k] + acc_l[k] : mine;
	if (lane == p + 2) mine = cyy;
	if (lane < p + 3) out[lane] = mine;
}

// The whole refinement of the queued groups in ONE launch: a wavefront takes a queued group through
//   kRefineSteps x (residual + gradient pass over its rows, iterative-refinement update)  and the final
//   residual pass + statistics,
// the row passes on all 64 lanes, the small solves on lane 0.  Groups are independent, so nothing has to be
// synchronised across waves; six dependent launches (~35 us of dispatch latency each on an otherwise empty
// queue) become one.
template <int P>
__global__ __launch_bounds__(64) void refine_fused_kernel(BatchArgs args, int steps) {
	const int lane = threadIdx.x & 63;
	const int n = *args.refine_count;
	for (int i = blockIdx.x; i < n; i += gridDim.x) {
		const int64_t g = args.refine_list[i];
		for (int it = 0; it <= steps; ++it) {
			// Producer and consumer are lanes of the SAME wavefront: a workgroup-scope fence orders the stores before the
			// loads without the agent-scope L2 write-back / invalidate of __threadfence(), which costs microseconds when
			// the L2 is full of another kernel's dirty lines (7 801 queued window frames: 0.89 ms -> measured below)
			residual_grad_wave(args, g, lane);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // refine_vec[g] written by lanes 0..p+2, read by lane 0
			if (lane == 0) {
				if (it < steps) solve_one<P, MODE_UPDATE>(args, g);
				else solve_one<P, MODE_FINAL>(args, g);
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // the record's coefficients written by lane 0, read by every lane in the next pass
		}
	}
}

template <int P>
hipError_t launch_solve_p(const BatchArgs &a, hipStream_t stream) {
	const unsigned grid = (unsigned)((a.n_groups + 63) / 64);
	hipLaunchKernelGGL((solve_narrow_kernel<P>), dim3(grid), dim3(64), 0, stream, a);
	return hipGetLastError();
}

} // namespace

hipError_t launch_solve_narrow(const BatchArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	switch (a.p) {
	case 1: return launch_solve_p<1>(a, stream);
	case 2: return launch_solve_p<2>(a, stream);
	case 3: return launch_solve_p<3>(a, stream);
	case 4: return launch_solve_p<4>(a, stream);
	case 5: return launch_solve_p<5>(a, stream);
	case 6: return launch_solve_p<6>(a, stream);
	case 7: return launch_solve_p<7>(a, stream);
	case 8: return launch_solve_p<8>(a, stream);
	default: return hipErrorInvalidValue;
	}
}

template <int P>
hipError_t launch_refine_p(const BatchArgs &a, int steps, hipStream_t stream) {
	// one wavefront per queued group and trip: the grid covers small batches entirely (the refits of flagged window
	// frames queue nearly every group of theirs: 7 801 groups on 1 024 wavefronts took 0.9 ms, eight groups in a row each)
	const unsigned grid = a.n_groups < 1024 ? 1024u : (a.n_groups > 16384 ? 16384u : (unsigned)a.n_groups);
	hipLaunchKernelGGL((refine_fused_kernel<P>), dim3(grid), dim3(64), 0, stream, a, steps);
	return hipGetLastError();
}

hipError_t launch_refine_fused_narrow(const BatchArgs &a, int steps, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	switch (a.p) {
	case 1: return launch_refine_p<1>(a, steps, stream);
	case 2: return launch_refine_p<2>(a, steps, stream);
	case 3: return launch_refine_p<3>(a, steps, stream);
	case 4: return launch_refine_p<4>(a, steps, stream);
	case 5: return launch_refine_p<5>(a, steps, stream);
	case 6: return launch_refine_p<6>(a, steps, stream);
	case 7: return launch_refine_p<7>(a, steps, stream);
	case 8: return launch_refine_p<8>(a, steps, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace anofox
