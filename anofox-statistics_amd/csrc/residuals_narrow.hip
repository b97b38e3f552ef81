// residuals_narrow.hip — raw / standardized / studentized residuals and leverage for up to 8 features, one
// wavefront per group.
//
// Reference: compute_residuals (crates/anofox-stats-core/src/diagnostics/residuals.rs:30-77) and its helper
// compute_studentized_residuals (residuals.rs:82-145):
//     raw_i          = y_i - yhat_i
//     standardized_i = raw_i / s                      (s given and > 0; raw_i when s <= 0; absent without s)
//     leverage_i     = x~_i' (X~'X~)^-1 x~_i          (X~ = [1, X]; absent without X, or when X~'X~ is singular)
//     studentized_i  = raw_i / (s sqrt(max(1 - h_i, 1e-10)))     (needs s and the leverage)
// Callers: anofox_compute_residuals (crates/anofox-stats-ffi/src/lib.rs:1787-1894), residuals_diagnostics_agg
// (src/aggregate_functions/residuals_diagnostics_aggregate.cpp:154-163 drops rows whose y or yhat is NaN; :232
// passes s = NaN, so the aggregate yields raw + leverage only), the scalar residuals_diagnostics
// (src/scalar_functions/residuals_diagnostics.cpp:75-192).
//
// The reference inverts the (p+1) x (p+1) raw cross-product matrix by Gauss-Jordan elimination.  With the
// intercept column eliminated first that is h_i = 1/n + d_i' A^-1 d_i with d_i = x_i - mean and A the centred
// Gram matrix, which is what this kernel evaluates (A = L L', h_i = 1/n + |L^-1 d_i|^2): the centred form does not
// lose digits to a large mean.  Rank-deficient X: the reference reports "no leverage" when a pivot falls below
// 1e-14 in absolute value, i.e. only when the cancellation happens to be exact, and otherwise divides by rounding
// noise; here a pivot below 1e-11 of its column's centred sum of squares always means "no leverage".
//
// Two passes over the group's rows: moments (shifted by the group's first used row, summed per lane, butterfly
// reduction so that every lane holds the totals), then the per-row outputs — 32 contiguous bytes per row.
#include "common.h"
#include "wave_reduce.h"

namespace anofox {

namespace {

constexpr double kLeverageAliasTol = 1e-11;

// a value every lane already agrees on, moved to scalar registers
__device__ inline double wave_uniform(double v) {
	const long long b = __double_as_longlong(v);
	const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(b & 0xffffffffu));
	const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)b >> 32));
	return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int P>
__global__ __launch_bounds__(256, 3) void residuals_narrow_kernel(ResidualArgs args) {
	constexpr int NQ = P * (P + 1) / 2;
	const int lane = threadIdx.x & 63;
	const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (g >= args.n_groups) return;
	const int64_t lo = args.row_offsets[g], hi = args.row_offsets[g + 1];
	const bool drop = args.drop_nan_rows != 0;
	const double nanv = __builtin_nan("");
	const double s = args.rse ? args.rse[g] : nanv;
	const bool has_s = !isnan(s);

	double Linv[NQ > 0 ? NQ : 1]; // packed lower triangle of L^-1, row-major
	double mean[P > 0 ? P : 1];
	double inv_n = 0.0;
	bool has_lev = false;
	double cnt = 0.0;

	if (P > 0 && args.include_studentized) {
		// the group's first used row centres the sums
		int64_t first = -1;
		for (int64_t base = lo; base < hi && first < 0; base += 64) {
			const int64_t r = base + lane;
			bool used = r < hi;
			if (used && drop) used = !isnan(args.y[r]) && !isnan(args.y_hat[r]);
			const unsigned long long m = __ballot(used);
			if (m) first = base + __builtin_ctzll(m);
		}
		double c[P > 0 ? P : 1], sum[P > 0 ? P : 1], q[NQ > 0 ? NQ : 1];
#pragma unroll
		for (int j = 0; j < P; ++j) {
			c[j] = first >= 0 ? args.x[j][first] : 0.0;
			sum[j] = 0.0;
		}
#pragma unroll
		for (int k = 0; k < NQ; ++k) q[k] = 0.0;
		for (int64_t r = lo + lane; r < hi; r += 64) {
			bool used = true;
			if (drop) used = !isnan(args.y[r]) && !isnan(args.y_hat[r]);
			double d[P > 0 ? P : 1];
#pragma unroll
			for (int j = 0; j < P; ++j) d[j] = args.x[j][r] - c[j];
			if (used) {
				cnt += 1.0;
#pragma unroll
				for (int j = 0; j < P; ++j) {
					sum[j] += d[j];
#pragma unroll
					for (int k = 0; k <= j; ++k) q[j * (j + 1) / 2 + k] = fma(d[j], d[k], q[j * (j + 1) / 2 + k]);
				}
			}
		}
		{ // transposing butterfly: total k lands on lane k, then one broadcast per total
			double v[64];
#pragma unroll
			for (int k = 0; k < 64; ++k) v[k] = 0.0;
#pragma unroll
			for (int k = 0; k < NQ; ++k) v[k] = q[k];
#pragma unroll
			for (int j = 0; j < P; ++j) v[NQ + j] = sum[j];
			v[NQ + P] = cnt;
			transpose_reduce64(v, lane);
#pragma unroll
			for (int k = 0; k < NQ; ++k) q[k] = readlane_f64(v[0], k);
#pragma unroll
			for (int j = 0; j < P; ++j) sum[j] = readlane_f64(v[0], NQ + j);
			cnt = readlane_f64(v[0], NQ + P);
		}
		if (cnt > 0.0) {
			inv_n = 1.0 / cnt;
			// centred Gram matrix, Cholesky factor in place, then L^-1
			bool ok = true;
#pragma unroll
			for (int j = 0; j < P; ++j) {
				mean[j] = c[j] + sum[j] * inv_n;
#pragma unroll
				for (int k = 0; k <= j; ++k) q[j * (j + 1) / 2 + k] -= sum[j] * sum[k] * inv_n;
			}
			double diag0[P > 0 ? P : 1], rdiag[P > 0 ? P : 1];
#pragma unroll
			for (int j = 0; j < P; ++j) diag0[j] = q[j * (j + 1) / 2 + j];
#pragma unroll
			for (int j = 0; j < P; ++j) {
				double dj = q[j * (j + 1) / 2 + j];
#pragma unroll
				for (int k = 0; k < j; ++k) dj -= q[j * (j + 1) / 2 + k] * q[j * (j + 1) / 2 + k];
				if (!(dj > kLeverageAliasTol * diag0[j]) || !(dj > 0.0)) { // NaN moments fall through here as well
					ok = false;
					dj = 1.0;
				}
				const double rl = 1.0 / sqrt(dj); // one division per column; the rest multiply by it
				rdiag[j] = rl;
#pragma unroll
				for (int i = j + 1; i < P; ++i) {
					double v = q[i * (i + 1) / 2 + j];
#pragma unroll
					for (int k = 0; k < j; ++k) v -= q[i * (i + 1) / 2 + k] * q[j * (j + 1) / 2 + k];
					q[i * (i + 1) / 2 + j] = v * rl;
				}
			}
#pragma unroll
			for (int j = P - 1; j >= 0; --j) { // L^-1 in place (off-diagonal part; its diagonal is rdiag), last column first
#pragma unroll
				for (int i = P - 1; i > j; --i) { // descending i: q[k][j], k < i, still hold L
					double v = rdiag[i] * q[i * (i + 1) / 2 + j];
#pragma unroll
					for (int k = j + 1; k < i; ++k) v = fma(q[i * (i + 1) / 2 + k], q[k * (k + 1) / 2 + j], v);
					q[i * (i + 1) / 2 + j] = -v * rdiag[j];
				}
			}
#pragma unroll
			for (int i = 0; i < P; ++i) {
#pragma unroll
				for (int k = 0; k <= i; ++k) Linv[i * (i + 1) / 2 + k] = k == i ? rdiag[i] : q[i * (i + 1) / 2 + k];
			}
			// a NaN / inf feature value in a used row poisons every leverage of the group in the reference (its
			// elimination carries the NaN through); the NaN moments end in !ok here, restore that outcome
			bool poisoned = false;
#pragma unroll
			for (int j = 0; j < P; ++j) poisoned |= !isfinite(sum[j]) || !isfinite(diag0[j]);
			has_lev = ok || poisoned;
			if (poisoned) {
#pragma unroll
				for (int k = 0; k < NQ; ++k) Linv[k] = nanv;
			}
		}
	}

	if (P > 0 && has_lev) { // the per-row pass keeps the factor in scalar registers
		inv_n = wave_uniform(inv_n);
#pragma unroll
		for (int j = 0; j < P; ++j) mean[j] = wave_uniform(mean[j]);
#pragma unroll
		for (int k = 0; k < NQ; ++k) Linv[k] = wave_uniform(Linv[k]);
	}
	const bool has_stud = has_lev && has_s;
	double n_used = 0.0;
	for (int64_t r = lo + lane; r < hi; r += 64) {
		const double yv = args.y[r], yh = args.y_hat[r];
		const bool used = !drop || (!isnan(yv) && !isnan(yh));
		double z[P > 0 ? P : 1];
#pragma unroll
		for (int j = 0; j < P; ++j) z[j] = has_lev ? args.x[j][r] - mean[j] : 0.0;
		const double raw = yv - yh;
		double lev = nanv, stud = nanv, stdz = nanv;
		if (has_lev) {
			double h = inv_n;
#pragma unroll
			for (int i = 0; i < P; ++i) {
				double t = 0.0;
#pragma unroll
				for (int k = 0; k <= i; ++k) t = fma(Linv[i * (i + 1) / 2 + k], z[k], t);
				h = fma(t, t, h);
			}
			lev = h;
			if (has_stud) stud = raw / (s * sqrt(fmax(1.0 - h, 1e-10)));
		}
		if (has_s) stdz = s > 0.0 ? raw / s : raw;
		double *out = args.out + r * 4;
		out[0] = used ? raw : nanv;
		out[1] = used ? stdz : nanv;
		out[2] = used ? stud : nanv;
		out[3] = used ? lev : nanv;
		n_used += used ? 1.0 : 0.0;
	}
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) n_used += __shfl_xor(n_used, m, 64);
	if (lane == 0) {
		args.group_out[g * 2] = n_used;
		args.group_out[g * 2 + 1] = (double)((has_s ? ANOFOX_HIP_RESIDUALS_HAS_STANDARDIZED : 0) |
		                                      (has_stud ? ANOFOX_HIP_RESIDUALS_HAS_STUDENTIZED : 0) |
		                                      (has_lev ? ANOFOX_HIP_RESIDUALS_HAS_LEVERAGE : 0));
	}
}

template <int P>
hipError_t launch_p(const ResidualArgs &a, hipStream_t stream) {
	hipLaunchKernelGGL((residuals_narrow_kernel<P>), dim3((unsigned)((a.n_groups + 3) / 4)), dim3(256), 0, stream, a);
	return hipGetLastError();
}

} // namespace

hipError_t launch_residuals_narrow(const ResidualArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	switch (a.p) {
	case 0: return launch_p<0>(a, stream);
	case 1: return launch_p<1>(a, stream);
	case 2: return launch_p<2>(a, stream);
	case 3: return launch_p<3>(a, stream);
	case 4: return launch_p<4>(a, stream);
	case 5: return launch_p<5>(a, stream);
	case 6: return launch_p<6>(a, stream);
	case 7: return launch_p<7>(a, stream);
	case 8: return launch_p<8>(a, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace anofox
