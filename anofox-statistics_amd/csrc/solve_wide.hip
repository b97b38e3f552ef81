// solve_wide.hip — per-group solve and diagnostics for wide designs (8 < p <= 128): one 256-thread workgroup
// per group, the p x p moment matrix in LDS.
//
// Same contract as solve_narrow.hip (the reference's pre-checks and shortcuts of
// crates/anofox-stats-core/src/models/ols.rs:68-139, ridge.rs:38-40, wls.rs:119-157; the regressor's closed
// forms; NaN re-expansion ols.rs:167-171,191-206), with the Cholesky factorisation,
// the triangular solves and the diagonal of the inverse done cooperatively in LDS:
//   lower triangle  : L (in place, right-looking, constant / aliased columns deactivated in place)
//   upper triangle  : L^-1 transposed (only when inference is requested), one thread per column
// Queued groups (small pivot ratio or RSS/TSS < 1e-7) get the same iterative-refinement passes as in
// solve_narrow.hip: MODE 1 = b += (X'WX)^-1 X'Wr from residual_grad_wide_kernel, MODE 2 = final statistics.
#include "common.h"
#include "device_math.h"

namespace anofox {

namespace {

constexpr double kAliasTolW = 1e-11;
constexpr double kRefineTolW = 1e-7;
constexpr double kPivotWarnW = 1e-3;
enum { MODE_PRIMARY = 0, MODE_UPDATE = 1, MODE_FINAL = 2 };

__device__ __forceinline__ double nan64w() { return __builtin_nan(""); }

template <int MODE>
__global__ __launch_bounds__(256) void solve_wide_kernel(WideArgs args) {
	constexpr bool REFINE = MODE != MODE_PRIMARY;
	extern __shared__ double sm[];
	const int p = args.p;
	const int T = wide_tiles(p);
	const int P16 = 16 * T;
	const int NT = T * (T + 1) / 2;
	const int LD = p | 1; // odd leading dimension: walks down a column hit distinct banks
	const int tid = threadIdx.x;
	const bool icpt = args.fit_intercept != 0;
	const int model = args.model;

	// LDS: A[p][LD] | cv[p] | zf[p] | sv[p] | fx[p] | diag0[p] | ldiag[p] | active[p], live[p] (int) | red[8]
	double *A = sm;
	double *cv = A + (size_t)p * LD; // right-hand side c, later the coefficients
	double *zf = cv + p;             // forward-solve result
	double *sv = zf + p;             // column sums of the shifted data
	double *fx = sv + p;             // x at the first valid row
	double *diag0 = fx + p;
	double *ldiag = diag0 + p;       // diagonal of L
	int *active = reinterpret_cast<int *>(ldiag + p); // not constant (from the accumulate kernel)
	int *live = active + p;                           // active and not aliased
	double *red = reinterpret_cast<double *>(live + p); // 2p ints = 8p bytes: stays 8-byte aligned

	const int n_items = REFINE ? *args.refine_count : (int)args.n_groups;
	for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
		const int64_t gl = REFINE ? (int64_t)args.refine_list[item] : (int64_t)item; // group within this launch
		const int64_t g = args.group_base + gl;                                       // global group
		const double *rec = args.moments + gl * (int64_t)wide_record_len(T);
		const double *vec = rec + (int64_t)NT * 256;
		const double *sc = vec + 4 * P16;
		double *core = args.core + g * (int64_t)(p + 6);
		double *inf = (args.inference && args.compute_inference) ? args.inference + g * (int64_t)(5 * p + 2) : nullptr;
		const int64_t nrows = args.row_offsets[g + 1] - args.row_offsets[g];

		// a record whose fit failed (or has no inference block): everything NaN, status in the last slot
		auto write_null = [&](int status, bool core_too) {
			if (core_too)
				for (int k = tid; k < p + 6; k += 256) core[k] = (k == p + 5) ? (double)status : nan64w();
			if (inf)
				for (int k = tid; k < 5 * p + 2; k += 256) inf[k] = nan64w();
		};

		__syncthreads(); // the previous item's LDS contents are dead
		const double sy = sc[0], syy = sc[1], sw = sc[2], cnt = sc[3], first_y = sc[4];
		int status = ANOFOX_ERROR_SUCCESS;
		if (nrows < 2) status = ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS;                                       // ols_aggregate.cpp:263-267
		else if (model == ANOFOX_HIP_MODEL_RIDGE && args.alpha < 0.0) status = ANOFOX_ERROR_INVALID_ALPHA;  // ridge.rs:38-40
		else if (!(cnt > 0.0)) status = ANOFOX_ERROR_NO_VALID_DATA;                                         // ols.rs:68-70
		if (status != ANOFOX_ERROR_SUCCESS) {
			write_null(status, true);
			continue;
		}

		if (tid < 8) red[tid] = 0.0;
		__syncthreads();
		int mine = 0;
		for (int j = tid; j < p; j += 256) {
			const int a = vec[3 * P16 + j] != 0.0 ? 1 : 0;
			active[j] = a;
			sv[j] = vec[0 * P16 + j];
			fx[j] = vec[2 * P16 + j];
			mine += a;
		}
		if (mine) atomicAdd(&red[0], (double)mine);
		__syncthreads();
		const int peff = (int)red[0];

		const double cyy_c = syy - sy * sy / sw;
		const double ymean = (icpt ? first_y : 0.0) + sy / sw;
		if (peff == 0) { // ols.rs:101-130, wls.rs:119-150
			if (!icpt) {
				write_null(ANOFOX_ERROR_INSUFFICIENT_DATA, true);
			} else {
				write_null(0, false); // inference: None
				for (int k = tid; k < p + 6; k += 256) {
					double v = nan64w();
					if (k == p) v = ymean;
					else if (k == p + 1 || k == p + 2 || k == p + 5) v = 0.0;
					else if (k == p + 3) v = (model == ANOFOX_HIP_MODEL_WLS) ? sqrt(cyy_c / sw) : sqrt(cyy_c / (cnt - 1.0));
					else if (k == p + 4) v = cnt;
					core[k] = v;
				}
			}
			continue;
		}
		if (cnt < (double)(peff + (icpt ? 1 : 0))) { // ols.rs:132-139
			write_null(ANOFOX_ERROR_INSUFFICIENT_DATA, true);
			continue;
		}

		double lam = 0.0;
		if (model == ANOFOX_HIP_MODEL_RIDGE) {
			lam = args.alpha;
			if (args.lambda_scaling == ANOFOX_LAMBDA_SCALING_GLMNET) lam = cnt * args.alpha / sqrt(cyy_c / cnt);
		}
		const double tss = icpt ? cyy_c : syy;

		// moment matrix (lower triangle), centred when an intercept is fitted
		for (int idx = tid; idx < p * p; idx += 256) {
			const int i = idx / p, j = idx - i * p;
			if (j > i) continue;
			const int I = j >> 4, J = i >> 4; // M[j][i] lives in the upper-triangular tile (I <= J)
			const int tile = I * T - I * (I - 1) / 2 + (J - I);
			double v = rec[(int64_t)tile * 256 + (j & 15) * 16 + (i & 15)];
			if (icpt) v -= sv[i] * sv[j] / sw;
			if (i == j) v += lam;
			A[(size_t)i * LD + j] = v;
		}
		for (int j = tid; j < p; j += 256) {
			const double q = vec[1 * P16 + j];
			cv[j] = icpt ? q - sv[j] * sy / sw : q;
		}
		__syncthreads();
		for (int j = tid; j < p; j += 256) diag0[j] = A[(size_t)j * LD + j];
		__syncthreads();

		// ---- Cholesky, right-looking; the diagonal of L goes to ldiag, A[j][j] keeps the pivot ----
		const int ti = tid >> 4, tk = tid & 15;
		double min_ratio = 1.0;
		for (int j = 0; j < p; ++j) {
			const double d = A[(size_t)j * LD + j];
			const bool ok = active[j] && (d > kAliasTolW * diag0[j]) && (d > 0.0);
			if (ok) min_ratio = fmin(min_ratio, d / diag0[j]);
			const double inv = ok ? 1.0 / sqrt(d) : 0.0;
			for (int i = j + 1 + tid; i < p; i += 256) A[(size_t)i * LD + j] *= inv; // aliased / constant: column := 0
			if (tid == 0) {
				ldiag[j] = ok ? sqrt(d) : 1.0;
				live[j] = ok ? 1 : 0;
			}
			__syncthreads();
			if (ok) {
				for (int i = j + 1 + ti; i < p; i += 16) {
					const double lij = A[(size_t)i * LD + j];
					for (int k = j + 1 + tk; k <= i; k += 16) A[(size_t)i * LD + k] -= lij * A[(size_t)k * LD + j];
				}
			}
			__syncthreads();
		}

		// cv := (L L')^-1 cv
		auto tri_solve = [&]() {
			for (int j = 0; j < p; ++j) { // forward: L zf = cv
				const bool lj = live[j] != 0;
				const double zj = lj ? cv[j] / ldiag[j] : 0.0;
				if (tid == 0) zf[j] = zj;
				if (lj) for (int i = j + 1 + tid; i < p; i += 256) cv[i] -= A[(size_t)i * LD + j] * zj;
				__syncthreads();
			}
			for (int j = p - 1; j >= 0; --j) { // back: L' x = zf, x -> cv
				const bool lj = live[j] != 0;
				const double bj = lj ? zf[j] / ldiag[j] : 0.0;
				if (tid == 0) cv[j] = bj;
				if (lj) for (int k = tid; k < j; k += 256) zf[k] -= A[(size_t)j * LD + k] * bj;
				__syncthreads();
			}
		};
		const double *rvec = args.refine_vec + g * (int64_t)(p + 2); // {sum w r^2, sum w r, X'Wr}
		if (MODE == MODE_PRIMARY) {
			tri_solve(); // cv = beta
		} else if (MODE == MODE_UPDATE) {
			// gradient of the (penalised) objective at the record's coefficients, centred coordinates
			const double gs = rvec[1];
			for (int j = tid; j < p; j += 256) {
				double gj = rvec[2 + j];
				if (icpt) gj -= (sv[j] / sw) * gs;
				const double bcur = live[j] ? core[j] : 0.0;
				cv[j] = live[j] ? gj - lam * bcur : 0.0;
			}
			__syncthreads();
			tri_solve(); // cv = delta
			for (int j = tid; j < p; j += 256) cv[j] += live[j] ? core[j] : 0.0;
			__syncthreads();
		} else {
			for (int j = tid; j < p; j += 256) cv[j] = live[j] ? core[j] : 0.0;
			__syncthreads();
		}

		// block sums: rank, b'c, b'b, mean correction of the intercept
		double rk = 0.0, bc = 0.0, bb = 0.0, xb = 0.0;
		for (int j = tid; j < p; j += 256) {
			if (live[j]) {
				rk += 1.0;
				const double q = vec[1 * P16 + j];
				const double cj = icpt ? q - sv[j] * sy / sw : q;
				const double bj = cv[j];
				bc += bj * cj;
				bb += bj * bj;
				xb += bj * ((icpt ? fx[j] : 0.0) + sv[j] / sw);
			}
		}
		atomicAdd(&red[2], rk);
		atomicAdd(&red[3], bc);
		atomicAdd(&red[4], bb);
		atomicAdd(&red[5], xb);
		__syncthreads();
		const int rank = (int)red[2];
		if (MODE == MODE_UPDATE) { // only the coefficients change in this pass
			for (int k = tid; k <= p; k += 256) {
				if (k < p) core[k] = live[k] ? cv[k] : nan64w();
				else core[k] = icpt ? ymean - red[5] : nan64w();
			}
			continue;
		}
		double rss;
		if (MODE == MODE_FINAL) rss = rvec[0];
		else rss = tss - red[3] - lam * red[4]; // = Syy - b'Sxy (- lam |b|^2 for ridge)
		const bool refine = (MODE == MODE_PRIMARY) && (!(rss > kRefineTolW * tss) || min_ratio < kPivotWarnW);
		const int n_par = rank + (icpt ? 1 : 0);
		const double df = cnt - (double)n_par;
		const double dfm = (double)rank;
		const double r2 = 1.0 - rss / tss;
		const double fstat = ((tss - rss) / dfm) / (rss / df);

		for (int k = tid; k < p + 6; k += 256) {
			double v;
			if (k < p) v = live[k] ? cv[k] : nan64w();
			else if (k == p) v = icpt ? ymean - red[5] : nan64w();
			else if (k == p + 1) v = r2;
			else if (k == p + 2) v = 1.0 - (1.0 - r2) * (cnt - (icpt ? 1.0 : 0.0)) / df;
			else if (k == p + 3) v = sqrt(rss / df);
			else if (k == p + 4) v = cnt;
			else v = 0.0;
			core[k] = v;
		}
		if (tid == 0 && refine) {
			const int slot = atomicAdd(args.refine_count, 1);
			args.refine_list[slot] = (int32_t)gl;
		}

		if (inf) {
			if (tid == 0) red[6] = dm_t_quantile_upper(0.5 * (1.0 + args.confidence_level), df);
			__syncthreads();
			const double tcrit = red[6];
			const double sigma2 = rss / df;
			// column j of L^-1, stored transposed in the upper triangle (W[i][j] -> A[j][i], i > j); one thread per column
			for (int j = tid; j < p; j += 256) {
				double se = nan64w(), tval = nan64w(), pval = nan64w(), lo = nan64w(), hi = nan64w();
				if (live[j]) {
					const double wjj = 1.0 / ldiag[j];
					double dj = wjj * wjj;
					for (int i = j + 1; i < p; ++i) {
						double t = 0.0;
						if (live[i]) {
							t = A[(size_t)i * LD + j] * wjj;
							for (int k = j + 1; k < i; ++k) t += A[(size_t)i * LD + k] * A[(size_t)j * LD + k];
							t = -t / ldiag[i];
						}
						A[(size_t)j * LD + i] = t;
						dj += t * t;
					}
					const double b = cv[j];
					se = sqrt(sigma2 * dj);
					tval = b / se;
					pval = dm_t_two_sided_p(tval, df);
					lo = b - tcrit * se;
					hi = b + tcrit * se;
				}
				inf[j] = se;
				inf[p + j] = tval;
				inf[2 * p + j] = pval;
				inf[3 * p + j] = lo;
				inf[4 * p + j] = hi;
			}
			if (tid == 0) {
				inf[5 * p] = fstat;
				inf[5 * p + 1] = dm_f_sf(fstat, dfm, df);
			}
		}
	}
}

// One workgroup per queued group, straight from the data with the record's current coefficients:
//   refine_vec[g] = { sum w r^2, sum w r, sum w r (x_j - shift_j) ... },  r = y - b0 - x'b  over the valid rows.
__global__ __launch_bounds__(256) void residual_grad_wide_kernel(WideArgs args) {
	__shared__ double bsh[kWideMaxP];
	__shared__ double we[256];
	__shared__ double part[8];
	const int p = args.p;
	const int T = wide_tiles(p);
	const int P16 = 16 * T;
	const int NT = T * (T + 1) / 2;
	const int tid = threadIdx.x;
	const bool weighted = args.model == ANOFOX_HIP_MODEL_WLS;
	const int n = *args.refine_count;
	for (int item = blockIdx.x; item < n; item += gridDim.x) {
		const int64_t gl = args.refine_list[item];
		const int64_t g = args.group_base + gl;
		const double *core = args.core + g * (int64_t)(p + 6);
		const double *vec = args.moments + gl * (int64_t)wide_record_len(T) + (int64_t)NT * 256;
		__syncthreads();
		for (int j = tid; j < p; j += 256) {
			const double bj = core[j];
			bsh[j] = isnan(bj) ? 0.0 : bj;
		}
		__syncthreads();
		const double b0 = args.fit_intercept ? core[p] : 0.0;
		const double shift = (tid < p && args.fit_intercept) ? vec[2 * P16 + tid] : 0.0; // x at the first valid row
		const double *mycol = tid < p ? args.x_table[tid] : nullptr;
		const int64_t lo = args.row_offsets[g], hi = args.row_offsets[g + 1];
		double rss = 0.0, gs = 0.0, gj = 0.0;
		for (int64_t r0 = lo; r0 < hi; r0 += 256) {
			const int64_t r = r0 + tid;
			double wev = 0.0;
			if (r < hi) {
				const double yv = args.y[r];
				bool ok = isfinite(yv);
				double fit = b0;
				for (int j = 0; j < p; ++j) {
					const double xv = args.x_table[j][r];
					ok = ok && isfinite(xv);
					fit = fma(bsh[j], xv, fit);
				}
				double wv = 1.0;
				if (weighted) {
					wv = args.w[r];
					ok = ok && (wv > 0.0) && isfinite(wv);
				}
				if (ok) {
					const double e = yv - fit;
					wev = wv * e;
					rss = fma(wev, e, rss);
					gs += wev;
				}
			}
			we[tid] = wev;
			__syncthreads();
			if (mycol) {
				const int64_t m = (hi - r0 < 256) ? hi - r0 : 256;
				for (int64_t t = 0; t < m; ++t) {
					const double wt = we[t];
					if (wt != 0.0) gj = fma(wt, mycol[r0 + t] - shift, gj); // invalid rows carry 0 and are skipped
				}
			}
			__syncthreads();
		}
		for (int m = 32; m >= 1; m >>= 1) {
			rss += __shfl_xor(rss, m, 64);
			gs += __shfl_xor(gs, m, 64);
		}
		if ((tid & 63) == 0) {
			part[tid >> 6] = rss;
			part[4 + (tid >> 6)] = gs;
		}
		__syncthreads();
		double *out = args.refine_vec + g * (int64_t)(p + 2);
		if (tid == 0) {
			out[0] = part[0] + part[1] + part[2] + part[3];
			out[1] = part[4] + part[5] + part[6] + part[7];
		}
		if (tid < p) out[2 + tid] = gj;
	}
}

size_t solve_wide_lds_bytes(int p) {
	const int LD = p | 1;
	return ((size_t)p * LD + 6 * (size_t)p + 8) * sizeof(double) + (2 * (size_t)p + 2) * sizeof(int);
}

} // namespace

hipError_t launch_solve_wide(const WideArgs &a, int mode, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	const size_t lds = solve_wide_lds_bytes(a.p);
	static bool attr_set = false;
	if (!attr_set) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_wide_kernel<MODE_PRIMARY>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_wide_kernel<MODE_UPDATE>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_wide_kernel<MODE_FINAL>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		attr_set = true;
	}
	if (mode == MODE_PRIMARY) {
		const unsigned grid = (unsigned)(a.n_groups < 65535 * 16 ? a.n_groups : 65535 * 16);
		hipLaunchKernelGGL((solve_wide_kernel<MODE_PRIMARY>), dim3(grid), dim3(256), lds, stream, a);
	} else if (mode == MODE_UPDATE) {
		hipLaunchKernelGGL((solve_wide_kernel<MODE_UPDATE>), dim3(256), dim3(256), lds, stream, a);
	} else {
		hipLaunchKernelGGL((solve_wide_kernel<MODE_FINAL>), dim3(256), dim3(256), lds, stream, a);
	}
	return hipGetLastError();
}

hipError_t launch_residual_grad_wide(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(residual_grad_wide_kernel, dim3(512), dim3(256), 0, stream, a);
	return hipGetLastError();
}

} // namespace anofox
