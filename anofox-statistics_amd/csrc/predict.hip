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
#include "device_math.h"

namespace anofox {

namespace {

__global__ __launch_bounds__(256) void predict_kernel(PredictArgs args) {
	__shared__ double coef_s[4][kWideMaxP];
	__shared__ int dead_s[4][kWideMaxP];
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int64_t g = (int64_t)blockIdx.x * 4 + wv;
	if (g >= args.n_groups) return;
	const int p = args.p;
	const double *core = args.core + g * (int64_t)(p + 6);
	for (int j = lane; j < p; j += 64) {
		const double c = core[j];
		const bool dead = isnan(c);
		coef_s[wv][j] = dead ? 0.0 : c;
		dead_s[wv][j] = dead ? 1 : 0;
	}
	const double icpt = core[p];
	const double rse = core[p + 3];
	const double nobs = core[p + 4];
	const bool is_null = core[p + 5] != 0.0;
	const bool has_icpt = !isnan(icpt);
	const double b0 = has_icpt ? icpt : 0.0;
	double margin = 0.0; // 0 => bounds equal yhat
	if (!is_null && !(isnan(rse) || rse <= 0.0 || nobs <= (double)(p + 1))) {
		const double df = has_icpt ? nobs - (double)(p + 1) : nobs - (double)p;
		const double c = args.confidence_level;
		if (df > 0.0 && c > 0.0 && c < 1.0) { // anofox_t_critical: NaN outside (0, 1) -> no interval
			const double tcrit = dm_tcrit_cached(static_cast<TcritSlot *>(args.tcrit_table), 0.5 * (1.0 + c), df);
			if (!isnan(tcrit)) margin = tcrit * rse * sqrt(1.0 + 1.0 / nobs);
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // the wave's own LDS writes above
	__builtin_amdgcn_wave_barrier();
	const int64_t lo = args.row_offsets[g], hi = args.row_offsets[g + 1];
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

} // namespace

hipError_t launch_predict(const PredictArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(predict_kernel, dim3((unsigned)((a.n_groups + 3) / 4)), dim3(256), 0, stream, a);
	return hipGetLastError();
}

} // namespace anofox
