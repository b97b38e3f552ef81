// hc_narrow.hip — heteroscedasticity-consistent standard errors HC0..HC3 (p <= 8), one wavefront per group.
//
// Reference behaviour: with `hc_type` set, OLS and WLS replace std_errors / t_values / p_values / ci_lower /
// ci_upper by their HC versions and keep the classical F statistic
// (crates/anofox-stats-core/src/models/ols.rs:209-231, wls.rs:230-252; ridge has no HC branch, ridge.rs).
// The estimator itself is anofox-regression's inference::compute_hc_inference, a crate that is not vendored
// in the reference tree, and the reference's tests assert only "finite, positive, differs from classical"
// (ols.rs:402-453, test/sql/regression/test_map_options.test:91-131): PARITY UNPINNED.  This kernel follows
// the published definition (MacKinnon & White 1985; R sandwich::vcovHC) on the sqrt(w)-scaled design:
//     V = B (sum_i omega_i a_i a_i') B,   B = (A'A)^-1,   h_i = a_i' B a_i,
//     omega_i = e_i^2 | e_i^2 n/(n-k) | e_i^2/(1-h_i) | e_i^2/(1-h_i)^2        (HC0 | HC1 | HC2 | HC3)
// with t quantiles and p-values at n-k degrees of freedom.
//
// Only diag(V) of the slopes is needed.  With the centred moment matrix S = L L' of the solve
// (solve_narrow.hip) and c_i = x_i - xbar (c_i = x_i without intercept):
//     h_i = w_i (1/sum(w) + |L^-1 c_i|^2),   V_jj = sum_i omega_i w_i (L^-T L^-1 c_i)_j^2
// so the pass is one more stream over the rows with p accumulators per lane.  Two kernels:
//   hc_prepare_kernel  one lane per group: L^-1 from the moment record with the active set the solve decided
//                      on (NaN coefficients mark dropped and aliased columns), plus xbar, b, b0 and the scalars
//                      of the group -> a "prep" record;
//   hc_narrow_kernel   one wavefront per group: the prep record is wave-uniform (scalar loads), the rows
//                      stream through lane-strided loads; writes the HC standard errors;
//   hc_finish_kernel   one lane per group: t, p-value and interval from those errors.
// Rows beyond a group's first seg_rows go to extra wavefronts (second launch of hc_narrow_kernel, idle otherwise);
// the partial V_jj of such a group are summed with atomics in its prep record.
#include "common.h"
#include "device_math.h"

namespace anofox {

namespace {

// prep record: [LT) L^-1 packed row-major lower | xbar[P] | b[P] | b0, h0, n/(n-k), df, valid, active mask
template <int P>
struct HcPrep {
	static constexpr int LT = P * (P + 1) / 2;
	static constexpr int OFF_XBAR = LT;
	static constexpr int OFF_B = LT + P;
	static constexpr int OFF_B0 = LT + 2 * P;
	static constexpr int OFF_H0 = OFF_B0 + 1;
	static constexpr int OFF_HC1 = OFF_B0 + 2;
	static constexpr int OFF_DF = OFF_B0 + 3;
	static constexpr int OFF_VALID = OFF_B0 + 4;
	static constexpr int OFF_MASK = OFF_B0 + 5;
	static constexpr int OFF_VSUM = OFF_B0 + 6; // [P] V_jj of groups whose rows were split over several wavefronts (atomic sums)
	static constexpr int REC = OFF_B0 + 6 + P;
	__host__ __device__ static constexpr int li(int i, int j) { return i * (i + 1) / 2 + j; } // i >= j
};

template <int P>
__global__ __launch_bounds__(64) void hc_prepare_kernel(BatchArgs args, double *prep_all) {
	using L = MomentLayout<P>;
	using H = HcPrep<P>;
	const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= args.n_groups) return;
	const bool icpt = args.fit_intercept != 0;
	const double *core = args.core + g * (int64_t)(P + 6);
	const double *rec = args.moments + g * (int64_t)L::REC;
	double *prep = prep_all + g * (int64_t)H::REC;

	bool active[P];
	int rank = 0;
	unsigned mask = 0;
	const bool fitted = core[P + 5] == 0.0; // NULL groups keep their NaN inference record
#pragma unroll
	for (int j = 0; j < P; ++j) {
		active[j] = fitted && !isnan(core[j]);
		rank += active[j] ? 1 : 0;
		mask |= active[j] ? (1u << j) : 0u;
	}
	prep[H::OFF_VALID] = rank > 0 ? 1.0 : 0.0; // rank 0: intercept-only fit, inference is None (ols.rs:101-130)
	prep[H::OFF_MASK] = (double)mask;
#pragma unroll
	for (int j = 0; j < P; ++j) prep[H::OFF_VSUM + j] = 0.0;
	if (rank == 0) return;

	const double sw = rec[L::OFF_SW];
	const double cnt = rec[L::OFF_CNT];
	double s[P];
#pragma unroll
	for (int j = 0; j < P; ++j) s[j] = rec[L::OFF_S + j];

	// Cholesky of the centred moment matrix on the active set, then its inverse factor
	double A[P][P];
#pragma unroll
	for (int i = 0; i < P; ++i) {
#pragma unroll
		for (int j = 0; j <= i; ++j) {
			const double qij = rec[L::q_index(j, i)];
			A[i][j] = icpt ? qij - s[i] * s[j] / sw : qij;
		}
	}
#pragma unroll
	for (int j = 0; j < P; ++j) {
		double d = A[j][j];
#pragma unroll
		for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
		const bool ok = active[j];
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
#pragma unroll
	for (int j = 0; j < P; ++j) {
		double col[P]; // column j of L^-1, zero at inactive positions
#pragma unroll
		for (int i = j; i < P; ++i) {
			double t = (i == j) ? 1.0 : 0.0;
#pragma unroll
			for (int k = j; k < i; ++k) t -= A[i][k] * col[k];
			col[i] = (active[i] && active[j]) ? t / A[i][i] : 0.0;
			prep[H::li(i, j)] = col[i];
		}
	}
#pragma unroll
	for (int j = 0; j < P; ++j) {
		prep[H::OFF_XBAR + j] = icpt ? rec[L::OFF_FIRST + j] + s[j] / sw : 0.0;
		prep[H::OFF_B + j] = active[j] ? core[j] : 0.0;
	}
	const double df = cnt - (double)(rank + (icpt ? 1 : 0));
	prep[H::OFF_B0] = icpt ? core[P] : 0.0;
	prep[H::OFF_H0] = icpt ? 1.0 / sw : 0.0;
	prep[H::OFF_HC1] = cnt / df;
	prep[H::OFF_DF] = df;
}

// SEGMENTS = false: wave per group (its first seg_rows rows); true: wave per registered overflow segment
template <int P, bool WEIGHTED, bool SEGMENTS>
__global__ __launch_bounds__(256) void hc_narrow_kernel(BatchArgs args, const double *__restrict__ prep_all, double *prep_sums,
                                                        void *overflow) {
	// prep_all and prep_sums are the same buffer: the kernel only READS the record fields through prep_all (uniform,
	// scalar loads) and only ADDS to the OFF_VSUM slots through prep_sums, which it never reads
	using H = HcPrep<P>;
	const int lane = threadIdx.x & 63;
	const int v = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
	int g;
	int64_t lo, hi;
	bool split;
	if (SEGMENTS) {
		const PredictSegTable *t = static_cast<const PredictSegTable *>(overflow);
		if (v >= t->count) return; // reservations never exceed the capacity
		g = (int)t->entries[v].g; lo = t->entries[v].lo; hi = t->entries[v].hi;
		split = true;
	} else {
		if (v >= args.n_groups) return;
		g = v; lo = args.row_offsets[g]; hi = args.row_offsets[g + 1];
		split = overflow && hi - lo > args.seg_rows;
	}
	const double *__restrict__ prep = prep_all + (int64_t)g * H::REC;
	if (prep[H::OFF_VALID] == 0.0) return;
	if (!SEGMENTS) hi = register_overflow_rows(overflow, args.seg_rows, g, lo, hi, lane);
	const int hc = args.hc_type;

	double Li[H::LT], xbar[P], b[P];
#pragma unroll
	for (int k = 0; k < H::LT; ++k) Li[k] = prep[k];
#pragma unroll
	for (int j = 0; j < P; ++j) {
		xbar[j] = prep[H::OFF_XBAR + j];
		b[j] = prep[H::OFF_B + j];
	}
	const double b0 = prep[H::OFF_B0], h0 = prep[H::OFF_H0], hc1 = prep[H::OFF_HC1];
	const unsigned mask = (unsigned)prep[H::OFF_MASK];

	double acc[P];
#pragma unroll
	for (int j = 0; j < P; ++j) acc[j] = 0.0;

	for (int64_t base = lo; base < hi; base += 64) {
		const bool in = base + lane < hi;
		const int64_t r = in ? base + lane : hi - 1; // clamped: loads stay unconditional
		const double yv = args.y[r];
		bool ok = in && isfinite(yv);
		double c[P];
		double fit = b0;
#pragma unroll
		for (int j = 0; j < P; ++j) {
			const double xv = args.x[j][r];
			ok = ok && isfinite(xv);
			fit = fma(b[j], xv, fit);
			c[j] = xv - xbar[j];
		}
		double wv = 1.0;
		if (WEIGHTED) {
			wv = args.w[r];
			ok = ok && (wv > 0.0) && isfinite(wv);
		}
		// z = L^-1 c, h = w (h0 + |z|^2), u = L^-T z
		double z[P], q = 0.0;
#pragma unroll
		for (int i = 0; i < P; ++i) {
			double t = 0.0;
#pragma unroll
			for (int k = 0; k <= i; ++k) t = fma(Li[H::li(i, k)], c[k], t);
			z[i] = t;
			q = fma(t, t, q);
		}
		const double e = yv - fit;
		const double h = wv * (h0 + q);
		double om = wv * wv * e * e; // w_i e_i^2 of the scaled residual, times the w_i of the scaled row
		if (hc == ANOFOX_HC_HC1) om *= hc1;
		else if (hc == ANOFOX_HC_HC2) om /= (1.0 - h);
		else if (hc == ANOFOX_HC_HC3) om /= (1.0 - h) * (1.0 - h);
		om = ok ? om : 0.0;
#pragma unroll
		for (int j = 0; j < P; ++j) {
			double u = 0.0;
#pragma unroll
			for (int i = j; i < P; ++i) u = fma(Li[H::li(i, j)], z[i], u);
			u = ok ? u : 0.0; // rows outside the fit may carry NaN / inf
			acc[j] = fma(om, u * u, acc[j]);
		}
	}
#pragma unroll
	for (int j = 0; j < P; ++j)
		for (int m = 32; m >= 1; m >>= 1) acc[j] += __shfl_xor(acc[j], m, 64);

	double vmine = 0.0;
#pragma unroll
	for (int j = 0; j < P; ++j) vmine = (lane == j) ? acc[j] : vmine;
	if (lane < P && ((mask >> lane) & 1u)) {
		if (split) atomicAdd(prep_sums + (int64_t)g * H::REC + H::OFF_VSUM + lane, vmine); // finished in hc_finish_kernel
		else args.inference[(int64_t)g * (5 * P + 2) + lane] = sqrt(vmine);
	}
}

// t, p and the interval from the HC errors, one lane per group (the special functions are out-of-line calls:
// kept out of the streaming kernel so that they do not set its register budget)
template <int P>
__global__ __launch_bounds__(64) void hc_finish_kernel(BatchArgs args, const double *__restrict__ prep_all) {
	using H = HcPrep<P>;
	const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= args.n_groups) return;
	const double *prep = prep_all + g * (int64_t)H::REC;
	if (prep[H::OFF_VALID] == 0.0) return;
	const double df = prep[H::OFF_DF];
	const unsigned mask = (unsigned)prep[H::OFF_MASK];
	const double tcrit = dm_tcrit_cached(static_cast<TcritSlot *>(args.tcrit_table), 0.5 * (1.0 + args.confidence_level), df);
	double *inf = args.inference + g * (int64_t)(5 * P + 2);
	const bool split = args.seg_table && args.row_offsets[g + 1] - args.row_offsets[g] > args.seg_rows;
	for (int j = 0; j < P; ++j) {
		if (!((mask >> j) & 1u)) continue;
		if (split) inf[j] = sqrt(prep[H::OFF_VSUM + j]);
		const double se = inf[j];
		const double bj = prep[H::OFF_B + j];
		const double tv = bj / se;
		inf[P + j] = tv;
		inf[2 * P + j] = dm_t_two_sided_p(tv, df);
		inf[3 * P + j] = bj - tcrit * se;
		inf[4 * P + j] = bj + tcrit * se;
	}
}

template <int P>
hipError_t launch_hc_p(const BatchArgs &a, double *prep, void *overflow, hipStream_t stream) {
	hipLaunchKernelGGL((hc_prepare_kernel<P>), dim3((unsigned)((a.n_groups + 63) / 64)), dim3(64), 0, stream, a, prep);
	const dim3 grid((unsigned)((a.n_groups + 3) / 4)), seg_grid((unsigned)((kSegTargetWaves + 16 + 3) / 4)), block(256);
	if (a.model == ANOFOX_HIP_MODEL_WLS) {
		hipLaunchKernelGGL((hc_narrow_kernel<P, true, false>), grid, block, 0, stream, a, prep, prep, overflow);
		if (overflow) hipLaunchKernelGGL((hc_narrow_kernel<P, true, true>), seg_grid, block, 0, stream, a, prep, prep, overflow);
	} else {
		hipLaunchKernelGGL((hc_narrow_kernel<P, false, false>), grid, block, 0, stream, a, prep, prep, overflow);
		if (overflow) hipLaunchKernelGGL((hc_narrow_kernel<P, false, true>), seg_grid, block, 0, stream, a, prep, prep, overflow);
	}
	hipLaunchKernelGGL((hc_finish_kernel<P>), dim3((unsigned)((a.n_groups + 63) / 64)), dim3(64), 0, stream, a, prep);
	return hipGetLastError();
}

} // namespace

size_t hc_prep_bytes(int64_t n_groups, int p) {
	return (size_t)n_groups * (size_t)(p * (p + 1) / 2 + 3 * p + 6) * sizeof(double);
}

hipError_t launch_hc_narrow(const BatchArgs &a, double *prep, void *overflow, hipStream_t stream) {
	if (a.n_groups <= 0 || !a.inference) return hipSuccess;
	if (a.n_groups > (int64_t)0x7fffffff / 4) return hipErrorInvalidValue;
	switch (a.p) {
	case 1: return launch_hc_p<1>(a, prep, overflow, stream);
	case 2: return launch_hc_p<2>(a, prep, overflow, stream);
	case 3: return launch_hc_p<3>(a, prep, overflow, stream);
	case 4: return launch_hc_p<4>(a, prep, overflow, stream);
	case 5: return launch_hc_p<5>(a, prep, overflow, stream);
	case 6: return launch_hc_p<6>(a, prep, overflow, stream);
	case 7: return launch_hc_p<7>(a, prep, overflow, stream);
	case 8: return launch_hc_p<8>(a, prep, overflow, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace anofox
