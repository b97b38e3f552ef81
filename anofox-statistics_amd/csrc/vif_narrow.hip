// vif_narrow.hip — variance inflation factors for up to 8 features, one lane per group.
//
// Reference: compute_vif (crates/anofox-stats-core/src/diagnostics/vif.rs:23-98) regresses every feature j on
// all the others with fit_ols (intercept, no inference) — p complete OLS fits per group — and maps the R^2:
//     fit failed -> inf,   R^2 >= 0.9999 -> inf,   R^2 < 0 -> 1,   else 1 / (1 - R^2).
// Callers: anofox_compute_vif (crates/anofox-stats-ffi/src/lib.rs:1688-1739), vif_agg
// (src/aggregate_functions/vif_aggregate.cpp:144-185), the scalar vif (src/scalar_functions/vif.cpp).
//
// All p regressions of a group share one Gram matrix: the accumulate kernel of the fit path is run once over
// the p features (its y slot carries x_0 again; only the x block of the record is used), and each lane then
// performs the p sub-fits on the centred moments with the solve's rules — constant "other" columns dropped by
// the |x - x_first| < 1e-10 test (ols.rs:76-87), intercept-only shortcut -> R^2 = 0 (ols.rs:101-130),
// fewer rows than parameters -> error (ols.rs:132-139), aliased columns by the pivot test of solve_narrow.hip.
#include "common.h"

namespace anofox {

namespace {

constexpr double kVifAliasTol = 1e-11;

template <int P>
__global__ __launch_bounds__(64) void vif_narrow_kernel(const double *moments, const int64_t *row_offsets, int64_t n_groups,
                                                         int64_t min_rows, double *out_all) {
	using L = MomentLayout<P>;
	const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_groups) return;
	const double *rec = moments + g * (int64_t)L::REC;
	double *out = out_all + g * (int64_t)(P + 1);
	const double nanv = __builtin_nan("");
	const double infv = __builtin_inf();
	const int64_t nrows = row_offsets[g + 1] - row_offsets[g];
	if (nrows < min_rows) { // vif_aggregate.cpp:154: fewer than 3 buffered rows -> NULL
#pragma unroll
		for (int j = 0; j < P; ++j) out[j] = nanv;
		out[P] = (double)ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS;
		return;
	}
	out[P] = 0.0;
	if (P == 1) { out[0] = 1.0; return; } // vif.rs:30-33
	const double cnt = rec[L::OFF_CNT];
	if (!(cnt > 0.0)) { // every sub-fit ends in NoValidData (ols.rs:68-70) -> inf (vif.rs:91-94)
#pragma unroll
		for (int j = 0; j < P; ++j) out[j] = infv;
		return;
	}
	const unsigned mask = (unsigned)rec[L::OFF_MASK];
	double S[P][P]; // centred second moments, lower triangle
#pragma unroll
	for (int i = 0; i < P; ++i) {
#pragma unroll
		for (int k = 0; k <= i; ++k) S[i][k] = rec[L::q_index(k, i)] - rec[L::OFF_S + i] * rec[L::OFF_S + k] / cnt;
	}
#pragma unroll
	for (int j = 0; j < P; ++j) {
		constexpr int Q = P > 1 ? P - 1 : 1;
		const int p_eff = __popc(mask & ~(1u << j));
		double v;
		if (p_eff == 0) {
			v = 1.0; // intercept-only model: R^2 = 0
		} else if (cnt < (double)(p_eff + 1)) {
			v = infv; // InsufficientData
		} else {
			double A[Q][Q], c[Q];
			bool act[Q];
#pragma unroll
			for (int a = 0; a < Q; ++a) {
				const int ia = a < j ? a : a + 1;
				act[a] = (mask >> ia) & 1u;
				c[a] = ia > j ? S[ia][j] : S[j][ia];
#pragma unroll
				for (int b = 0; b <= a; ++b) {
					const int ib = b < j ? b : b + 1;
					A[a][b] = S[ia][ib];
				}
			}
			double zz = 0.0;
			double zf[Q];
#pragma unroll
			for (int k = 0; k < Q; ++k) {
				double d = A[k][k];
				const double d0 = d;
#pragma unroll
				for (int m = 0; m < k; ++m) d -= A[k][m] * A[k][m];
				const bool ok = act[k] && (d > kVifAliasTol * d0) && (d > 0.0);
				act[k] = ok;
				const double lkk = ok ? sqrt(d) : 1.0;
				const double inv = 1.0 / lkk;
#pragma unroll
				for (int i = k + 1; i < Q; ++i) {
					double t = A[i][k];
#pragma unroll
					for (int m = 0; m < k; ++m) t -= A[i][m] * A[k][m];
					A[i][k] = ok ? t * inv : 0.0;
				}
				if (!ok) {
#pragma unroll
					for (int m = 0; m < k; ++m) A[k][m] = 0.0;
				}
				double t = c[k];
#pragma unroll
				for (int m = 0; m < k; ++m) t -= A[k][m] * zf[m];
				zf[k] = ok ? t * inv : 0.0;
				zz = fma(zf[k], zf[k], zz);
			}
			const double tss = S[j][j];
			const double r2 = 1.0 - (tss - zz) / tss;
			v = r2 >= 0.9999 ? infv : (r2 < 0.0 ? 1.0 : 1.0 / (1.0 - r2)); // vif.rs:80-88
		}
		out[j] = v;
	}
}

// Wide designs: the fit of x_j on the others ran through the grouped fit path; core = its records (q = p - 1 features).
__global__ __launch_bounds__(256) void vif_from_core_kernel(const double *core, const int64_t *row_offsets, int64_t n_groups, int q,
                                                             int j, int p, int64_t min_rows, double *out_all) {
	const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_groups) return;
	double *out = out_all + g * (int64_t)(p + 1);
	const double *rec = core + g * (int64_t)(q + 6);
	if (row_offsets[g + 1] - row_offsets[g] < min_rows) {
		out[j] = __builtin_nan("");
		out[p] = (double)ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS;
		return;
	}
	out[p] = 0.0;
	const double r2 = rec[q + 1];
	double v;
	if (rec[q + 5] != 0.0) v = __builtin_inf(); // failed fit (vif.rs:91-94)
	else v = r2 >= 0.9999 ? __builtin_inf() : (r2 < 0.0 ? 1.0 : 1.0 / (1.0 - r2));
	out[j] = v;
}

} // namespace

hipError_t launch_vif_from_core(const double *core, const int64_t *row_offsets, int64_t n_groups, int q, int j, int p,
                                int64_t min_rows, double *out, hipStream_t stream) {
	if (n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(vif_from_core_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, stream, core, row_offsets,
	                   n_groups, q, j, p, min_rows, out);
	return hipGetLastError();
}

hipError_t launch_vif_narrow(const double *moments, const int64_t *row_offsets, int64_t n_groups, int p, int64_t min_rows,
                             double *out, hipStream_t stream) {
	if (n_groups <= 0) return hipSuccess;
	const dim3 grid((unsigned)((n_groups + 63) / 64)), block(64);
	switch (p) {
	case 1: hipLaunchKernelGGL((vif_narrow_kernel<1>), grid, block, 0, stream, moments, row_offsets, n_groups, min_rows, out); break;
	case 2: hipLaunchKernelGGL((vif_narrow_kernel<2>), grid, block, 0, stream, moments, row_offsets, n_groups, min_rows, out); break;
	case 3: hipLaunchKernelGGL((vif_narrow_kernel<3>), grid, block, 0, stream, moments, row_offsets, n_groups, min_rows, out); break;
	case 4: hipLaunchKernelGGL((vif_narrow_kernel<4>), grid, block, 0, stream, moments, row_offsets, n_groups, min_rows, out); break;
	case 5: hipLaunchKernelGGL((vif_narrow_kernel<5>), grid, block, 0, stream, moments, row_offsets, n_groups, min_rows, out); break;
	case 6: hipLaunchKernelGGL((vif_narrow_kernel<6>), grid, block, 0, stream, moments, row_offsets, n_groups, min_rows, out); break;
	case 7: hipLaunchKernelGGL((vif_narrow_kernel<7>), grid, block, 0, stream, moments, row_offsets, n_groups, min_rows, out); break;
	case 8: hipLaunchKernelGGL((vif_narrow_kernel<8>), grid, block, 0, stream, moments, row_offsets, n_groups, min_rows, out); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

} // namespace anofox
