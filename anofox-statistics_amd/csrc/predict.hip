// predict.hip — per-row predictions with the reference's simplified interval, for every row of every group.
//
// Second half of the reference's `*_fit_predict_agg` Finalize (src/aggregate_functions/ols_predict_aggregate.cpp:
// 373-420): after the group's fit, each row gets anofox_predict_with_interval
// (crates/anofox-stats-ffi/src/lib.rs:2264-2349):
//   yhat = intercept (0 when NaN) + sum over the non-NaN coefficients of coef_j * x_j
//   interval yhat -/+ t_{(1+c)/2, df} * rse * sqrt(1 + 1/n),  df = n - p - [intercept]
//   no interval (bounds = yhat) when rse is NaN or <= 0, n <= p + 1, df == 0 or the critical value is NaN;
//   a non-finite yhat is SQL NULL (ols_predict_aggregate.cpp:404-412) -> all three outputs NaN.
// Groups whose fit is NULL (status != 0) give NaN for all their rows.
//
// Mapping: one wavefront per group, lanes stride the rows; the group's coefficients sit in LDS.
// HBM-bound: 8p B/row read, 24 B/row written.
#include "common.h"
#include <cstdlib>
#include "device_math.h"

namespace anofox {

namespace {

// Half-width of the interval per group (0 = bounds equal yhat), one thread per group.  Kept out of the row
// kernels: the t quantile's call tree needs ~240 VGPRs, the row kernels ~40 — separating them keeps the
// streaming kernels at full occupancy.
__global__ __launch_bounds__(256) void predict_margin_kernel(PredictArgs args) {
	const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= args.n_groups) return;
	const int p = args.p;
	const double *core = args.core + g * (int64_t)(p + 6);
	const double icpt = core[p];
	const double rse = core[p + 3];
	const double nobs = core[p + 4];
	const bool is_null = core[p + 5] != 0.0;
	const bool has_icpt = !isnan(icpt);
	double margin = 0.0;
	if (!is_null && !(isnan(rse) || rse <= 0.0 || nobs <= (double)(p + 1))) {
		const double df = has_icpt ? nobs - (double)(p + 1) : nobs - (double)p;
		const double c = args.confidence_level;
		if (df > 0.0 && c > 0.0 && c < 1.0) { // anofox_t_critical: NaN outside (0, 1) -> no interval
			const double tcrit = dm_tcrit_cached(static_cast<TcritSlot *>(args.tcrit_table), 0.5 * (1.0 + c), df);
			if (!isnan(tcrit)) margin = tcrit * rse * sqrt(1.0 + 1.0 / nobs);
		}
	}
	args.margin[g] = margin;
}

__device__ __forceinline__ void predict_rows_generic(const PredictArgs &args, int64_t g, int64_t lo, int64_t hi, double (*coef_s)[kWideMaxP],
                                                     int (*dead_s)[kWideMaxP], int lane, int wv) {
	const int p = args.p;
	const double *core = args.core + g * (int64_t)(p + 6);
	for (int j = lane; j < p; j += 64) {
		const double c = core[j];
		const bool dead = isnan(c);
		coef_s[wv][j] = dead ? 0.0 : c;
		dead_s[wv][j] = dead ? 1 : 0;
	}
	const double icpt = core[p];
	const bool is_null = core[p + 5] != 0.0;
	const double b0 = isnan(icpt) ? 0.0 : icpt;
	const double margin = args.margin[g]; // 0 => bounds equal yhat
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // the wave's own LDS writes above
	__builtin_amdgcn_wave_barrier();
	const double nanv = __builtin_nan("");
	for (int64_t r = lo + lane; r < hi; r += 64) {
		double yhat = b0;
		for (int j = 0; j < p; ++j) {
			if (!dead_s[wv][j]) yhat = fma(coef_s[wv][j], args.x_table[j][r], yhat);
		}
		const bool ok = !is_null && isfinite(yhat);
		double *out = args.pred + r * 3;
		out[0] = ok ? yhat : nanv;
		out[1] = ok ? yhat - margin : nanv;
		out[2] = ok ? yhat + margin : nanv;
	}
}

// SEGMENTS = false: wave per group (its first seg_rows rows); true: wave per registered overflow segment
template <bool SEGMENTS>
__global__ __launch_bounds__(256) void predict_kernel(PredictArgs args) {
	__shared__ double coef_s[4][kWideMaxP];
	__shared__ int dead_s[4][kWideMaxP];
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int64_t v = (int64_t)blockIdx.x * 4 + wv;
	int64_t g, lo, hi;
	if (SEGMENTS) {
		const PredictSegTable *t = static_cast<const PredictSegTable *>(args.seg_table);
		if (v >= t->count) return; // reservations never exceed the capacity
		g = t->entries[v].g; lo = t->entries[v].lo; hi = t->entries[v].hi;
	} else {
		if (v >= args.n_groups) return;
		g = v; lo = args.row_offsets[g];
		hi = register_overflow_rows(args.seg_table, args.seg_rows, g, lo, args.row_offsets[g + 1], lane);
	}
	predict_rows_generic(args, g, lo, hi, coef_s, dead_s, lane, wv);
}

// p <= 8: coefficients in registers, 128-row tiles with one 16-byte load per column and lane (rows 2l, 2l+1),
// all loads of a tile in flight together; 48 contiguous output bytes per lane.
typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));

template <int P, bool SEGMENTS>
__global__ __launch_bounds__(256) void predict_narrow_kernel(PredictArgs args) {
	const int lane = threadIdx.x & 63;
	const int64_t v = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (int64_t)blockIdx.x * 4;
	int64_t g, lo, hi;
	if (SEGMENTS) {
		const PredictSegTable *t = static_cast<const PredictSegTable *>(args.seg_table);
		if (v >= t->count) return; // reservations never exceed the capacity
		g = t->entries[v].g; lo = t->entries[v].lo; hi = t->entries[v].hi;
	} else {
		if (v >= args.n_groups) return;
		g = v; lo = args.row_offsets[g];
		hi = register_overflow_rows(args.seg_table, args.seg_rows, g, lo, args.row_offsets[g + 1], lane);
	}
	const double *core = args.core + g * (int64_t)(P + 6);
	double coef[P];
	bool dead[P];
#pragma unroll
	for (int j = 0; j < P; ++j) {
		const double c = core[j];
		dead[j] = isnan(c);
		coef[j] = dead[j] ? 0.0 : c;
	}
	const double icpt = core[P];
	const bool is_null = core[P + 5] != 0.0;
	const double b0 = isnan(icpt) ? 0.0 : icpt;
	const double margin = args.margin[g];
	const double nanv = __builtin_nan("");
	for (int64_t base = lo; base < hi; base += 128) {
		const int64_t r0 = base + 2 * lane;
		double x0[P], x1[P];
		if (base + 128 <= hi) {
#pragma unroll
			for (int j = 0; j < P; ++j) {
				const dbl2u v = *reinterpret_cast<const dbl2u *>(args.x_table[j] + r0);
				x0[j] = v.x;
				x1[j] = v.y;
			}
		} else {
			// ragged tail: unconditional loads from clamped (valid) rows — a guarded load per element would put
			// every load behind its own branch and wait; rows past the end are dropped at the store
			const int64_t c0 = r0 < hi ? r0 : hi - 1, c1 = r0 + 1 < hi ? r0 + 1 : hi - 1;
#pragma unroll
			for (int j = 0; j < P; ++j) {
				x0[j] = args.x_table[j][c0];
				x1[j] = args.x_table[j][c1];
			}
		}
		double y0 = b0, y1 = b0;
#pragma unroll
		for (int j = 0; j < P; ++j) { // NaN coefficients are skipped whatever x holds (lib.rs:2300-2304)
			y0 = fma(coef[j], dead[j] ? 0.0 : x0[j], y0);
			y1 = fma(coef[j], dead[j] ? 0.0 : x1[j], y1);
		}
		const bool ok0 = !is_null && isfinite(y0), ok1 = !is_null && isfinite(y1);
		double *out = args.pred + r0 * 3;
		if (r0 + 1 < hi) {
			dbl2u a, b, c;
			a.x = ok0 ? y0 : nanv;
			a.y = ok0 ? y0 - margin : nanv;
			b.x = ok0 ? y0 + margin : nanv;
			b.y = ok1 ? y1 : nanv;
			c.x = ok1 ? y1 - margin : nanv;
			c.y = ok1 ? y1 + margin : nanv;
			*reinterpret_cast<dbl2u *>(out) = a;
			*reinterpret_cast<dbl2u *>(out + 2) = b;
			*reinterpret_cast<dbl2u *>(out + 4) = c;
		} else if (r0 < hi) {
			out[0] = ok0 ? y0 : nanv;
			out[1] = ok0 ? y0 - margin : nanv;
			out[2] = ok0 ? y0 + margin : nanv;
		}
	}
}

// Small groups: several groups per wavefront (segments of SEGW lanes, one group each, lanes stride the rows) — with
// 20 rows per group the 128-row tiles above leave 84 % of the lanes idle.
template <int P, int SEGW>
__global__ __launch_bounds__(256) void predict_small_kernel(PredictArgs args) {
	constexpr int GPW = 64 / SEGW;
	const int lane = threadIdx.x & 63;
	const int seg = lane / SEGW, sl = lane % SEGW;
	const int64_t wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (int64_t)blockIdx.x * 4;
	const int64_t g = wave_id * GPW + seg;
	const bool live = g < args.n_groups;
	const int64_t gc = live ? g : 0;
	int64_t lo = 0, hi = 0;
	if (live) {
		lo = args.row_offsets[g];
		hi = args.row_offsets[g + 1];
	}
	if (args.seg_table && hi - lo > args.seg_rows) { // very large group: its first lane hands the tail to extra wavefronts
		PredictSegTable *t = static_cast<PredictSegTable *>(args.seg_table);
		const int64_t S = args.seg_rows;
		const int extra = (int)((hi - lo - 1) / S);
		int base = -1;
		if (sl == 0) {
			base = reserve_table_entries(&t->count, extra, kSegTargetWaves + 16);
			for (int k = 0; base >= 0 && k < extra; ++k) {
				PredictSegEntry e;
				e.g = g;
				e.lo = lo + (k + 1) * S;
				e.hi = e.lo + S < hi ? e.lo + S : hi;
				t->entries[base + k] = e;
			}
		}
		base = __shfl(base, seg * SEGW, 64);
		if (base >= 0) hi = lo + S;
	}
	const double *core = args.core + gc * (int64_t)(P + 6);
	double coef[P];
	bool dead[P];
#pragma unroll
	for (int j = 0; j < P; ++j) {
		const double c = core[j];
		dead[j] = isnan(c);
		coef[j] = dead[j] ? 0.0 : c;
	}
	const double icpt = core[P];
	const bool is_null = core[P + 5] != 0.0;
	const double b0 = isnan(icpt) ? 0.0 : icpt;
	const double margin = args.margin[gc];
	const double nanv = __builtin_nan("");
	int64_t nmax = hi - lo;
#pragma unroll
	for (int m = 32; m >= SEGW; m >>= 1) {
		const int64_t o = __shfl_xor(nmax, m, 64);
		nmax = o > nmax ? o : nmax;
	}
	for (int64_t off = sl; off < nmax; off += SEGW) {
		const int64_t r = lo + off;
		const bool in = r < hi;
		const int64_t rc = in ? r : (hi > lo ? hi - 1 : 0); // clamped: loads stay unconditional
		double yhat = b0;
#pragma unroll
		for (int j = 0; j < P; ++j) { // NaN coefficients are skipped whatever x holds (lib.rs:2300-2304)
			const double xv = args.x_table[j][rc];
			yhat = fma(coef[j], dead[j] ? 0.0 : xv, yhat);
		}
		const bool ok = !is_null && isfinite(yhat);
		if (in) {
			double *out = args.pred + r * 3;
			out[0] = ok ? yhat : nanv;
			out[1] = ok ? yhat - margin : nanv;
			out[2] = ok ? yhat + margin : nanv;
		}
	}
}

template <int P, int SEGW>
void launch_predict_small(const PredictArgs &a, hipStream_t stream) {
	constexpr int GPW = 64 / SEGW;
	const int64_t waves = (a.n_groups + GPW - 1) / GPW;
	hipLaunchKernelGGL((predict_small_kernel<P, SEGW>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, a);
}

template <int P>
hipError_t launch_predict_p(const PredictArgs &a, hipStream_t stream) {
	int segw = 0;
	if (a.avg_rows > 0.0 && a.avg_rows <= 12.0) segw = 8;
	else if (a.avg_rows > 0.0 && a.avg_rows <= 24.0) segw = 16;
	else if (a.avg_rows > 0.0 && a.avg_rows <= 64.0) segw = 32;
	if (const char *e = getenv("ANOFOX_PRED_SEGW")) segw = atoi(e);
	if (segw == 4) launch_predict_small<P, 4>(a, stream);
	else if (segw == 8) launch_predict_small<P, 8>(a, stream);
	else if (segw == 16) launch_predict_small<P, 16>(a, stream);
	else if (segw == 32) launch_predict_small<P, 32>(a, stream);
	else hipLaunchKernelGGL((predict_narrow_kernel<P, false>), dim3((unsigned)((a.n_groups + 3) / 4)), dim3(256), 0, stream, a);
	if (a.seg_table) // idle unless some group exceeded seg_rows
		hipLaunchKernelGGL((predict_narrow_kernel<P, true>), dim3((unsigned)((kSegTargetWaves + 16 + 3) / 4)), dim3(256), 0, stream, a);
	return hipGetLastError();
}

} // namespace

namespace {

// Information criteria of every fit of a batch, from the fit records alone (information_criteria.rs:15-33,67-85):
//   k   = estimated parameters = coefficients that are not NaN (+ 1 with an intercept)
//   rss = residual_std_error^2 (n - k)         (the record's sigma is sqrt(rss / (n - k)), ols.rs:183)
//   aic = n ln(rss / n) + 2 k,   bic = n ln(rss / n) + k ln n;   rss == 0 -> -inf;   NULL group or n == k -> NaN
// out[g] = { rss, aic, bic }.  One thread per group.
__global__ __launch_bounds__(256) void information_criteria_kernel(const double *core, int64_t n_groups, int p, int icpt, int wls,
                                                                   double *out) {
	const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_groups) return;
	const double *c = core + g * (int64_t)(p + 6);
	const double nanv = __builtin_nan("");
	double rss = nanv, aic = nanv, bic = nanv;
	if (c[p + 5] == 0.0) {
		int k = icpt ? 1 : 0;
		for (int j = 0; j < p; ++j) k += isnan(c[j]) ? 0 : 1;
		const double n = c[p + 4], rse = c[p + 3];
		// the intercept-only shortcut reports sqrt(sum w (y - ybar)^2 / sum w) for WLS (wls.rs:126-135): the weight
		// total is not in the record, so rss cannot be recovered there
		const bool wls_icpt_only = wls && k == (icpt ? 1 : 0);
		if (n > (double)k && !wls_icpt_only && !isnan(rse)) {
			rss = rse * rse * (n - (double)k);
			if (rss == 0.0) {
				aic = bic = -__builtin_inf();
			} else {
				const double base = n * log(rss / n);
				aic = base + 2.0 * (double)k;
				bic = base + (double)k * log(n);
			}
		}
	}
	out[3 * g] = rss;
	out[3 * g + 1] = aic;
	out[3 * g + 2] = bic;
}

} // namespace

hipError_t launch_information_criteria(const double *core, int64_t n_groups, int p, int fit_intercept, int wls, double *out,
                                       hipStream_t stream) {
	if (n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(information_criteria_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, stream, core, n_groups, p,
	                   fit_intercept, wls, out);
	return hipGetLastError();
}

hipError_t launch_predict(const PredictArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(predict_margin_kernel, dim3((unsigned)((a.n_groups + 255) / 256)), dim3(256), 0, stream, a);
	switch (a.p) {
	case 1: return launch_predict_p<1>(a, stream);
	case 2: return launch_predict_p<2>(a, stream);
	case 3: return launch_predict_p<3>(a, stream);
	case 4: return launch_predict_p<4>(a, stream);
	case 5: return launch_predict_p<5>(a, stream);
	case 6: return launch_predict_p<6>(a, stream);
	case 7: return launch_predict_p<7>(a, stream);
	case 8: return launch_predict_p<8>(a, stream);
	default: break;
	}
	hipLaunchKernelGGL(predict_kernel<false>, dim3((unsigned)((a.n_groups + 3) / 4)), dim3(256), 0, stream, a);
	if (a.seg_table) hipLaunchKernelGGL(predict_kernel<true>, dim3((unsigned)((kSegTargetWaves + 16 + 3) / 4)), dim3(256), 0, stream, a);
	return hipGetLastError();
}

} // namespace anofox
