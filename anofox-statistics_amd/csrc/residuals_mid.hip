// residuals_mid.hip — residual diagnostics (raw / standardized / studentized residuals, leverage) for 9..32 features:
// one 256-thread workgroup per group.  Same definitions, flags and conventions as residuals_narrow.hip (which see for
// the reference citations: crates/anofox-stats-core/src/diagnostics/residuals.rs:30-145,
// src/aggregate_functions/residuals_diagnostics_aggregate.cpp:154-163,213-286); what changes is where the
// (p+1)(p+2)/2 moments live — not in one lane's registers any more:
//   moments   rows are staged 64 at a time into an LDS tile [64][p + 1] (d = x - x_first of the used rows, plus a
//             column u = 1 for used rows); thread t owns up to three entries (j, k) of the packed triangle of the
//             augmented Gram matrix (d, u)'(d, u) — which holds sum d_j d_k, the column sums (row u) and the row count;
//   factor    Cholesky of the centred Gram matrix in LDS, all threads, two barriers per column (p <= 32);
//   rows      one thread per row: forward substitution t = L^-1 (x - mean) with t in registers and L read from LDS
//             (every thread reads the same element: a broadcast), h = 1/n + |t|^2.
#include "common.h"

namespace anofox {

namespace {

constexpr double kLeverageAliasTolMid = 1e-11;
constexpr int kMidMaxP = 32;
constexpr int kTileRows = 64;

__global__ __launch_bounds__(256) void residuals_mid_kernel(ResidualArgs args, const double *const *x_table) {
	const int p = args.p, Z = p + 1, NE = Z * (Z + 1) / 2;
	const int tid = threadIdx.x;
	const int64_t g = blockIdx.x;
	const int64_t lo = args.row_offsets[g], hi = args.row_offsets[g + 1];
	const bool drop = args.drop_nan_rows != 0;
	const double nanv = __builtin_nan("");
	const double s = args.rse ? args.rse[g] : nanv;
	const bool has_s = !isnan(s);

	__shared__ double tile[kTileRows][kMidMaxP + 2];   // d_0 .. d_{p-1}, u  (stride 34: rows of a column 34 apart)
	__shared__ double M[(kMidMaxP + 1) * (kMidMaxP + 2) / 2]; // packed lower triangle of the augmented moments, row-major
	__shared__ double A[kMidMaxP][kMidMaxP + 1];       // centred Gram matrix -> L (strict lower part), rdiag separately
	__shared__ double cvec[kMidMaxP], meanv[kMidMaxP], rdiag[kMidMaxP], diag0[kMidMaxP];
	__shared__ const double *xcol[kMidMaxP];
	__shared__ int first_row, flags; // flags: 1 = factor ok

	if (tid < p) xcol[tid] = x_table[tid];
	if (tid == 0) { first_row = 0x7fffffff; flags = 0; }
	__syncthreads();

	bool has_lev = false, poisoned = false;
	double inv_n = 0.0;
	if (args.include_studentized) {
		// the group's first used row centres the sums
		for (int64_t base = lo; base < hi; base += 256) {
			const int64_t r = base + tid;
			bool used = r < hi;
			if (used && drop) used = !isnan(args.y[r]) && !isnan(args.y_hat[r]);
			if (used) atomicMin(&first_row, (int)(r - lo));
			__syncthreads();
			if (first_row != 0x7fffffff) break; // uniform: read after the barrier
			__syncthreads();
		}
		__syncthreads();
		const int64_t first = first_row == 0x7fffffff ? -1 : lo + first_row;
		if (tid < p) cvec[tid] = first >= 0 ? xcol[tid][first] : 0.0;
		// this thread's entries of the packed triangle
		int ej[3], ek[3];
		double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
		for (int m = 0; m < 3; ++m) {
			const int e = tid + 256 * m;
			int j = 0;
			while ((j + 1) * (j + 2) / 2 <= e && j < Z - 1) ++j; // row j holds entries j(j+1)/2 .. j(j+1)/2 + j
			ej[m] = j;
			ek[m] = e < NE ? e - j * (j + 1) / 2 : 0;
			if (e >= NE) ej[m] = 0;
		}
		__syncthreads();
		for (int64_t base = lo; base < hi; base += kTileRows) {
			// stage 64 rows: element idx -> (column idx / 64, row idx % 64): 256 contiguous bytes per column per wave
			for (int idx = tid; idx < kTileRows * Z; idx += 256) {
				const int col = idx / kTileRows, row = idx - col * kTileRows;
				const int64_t r = base + row;
				bool used = r < hi;
				if (used && drop) used = !isnan(args.y[r]) && !isnan(args.y_hat[r]);
				double v = 0.0;
				if (used) v = col < p ? xcol[col][r] - cvec[col] : 1.0;
				tile[row][col] = v;
			}
			__syncthreads();
#pragma unroll
			for (int m = 0; m < 3; ++m) {
				if (tid + 256 * m < NE) {
					double a = acc[m];
					for (int row = 0; row < kTileRows; ++row) a = fma(tile[row][ej[m]], tile[row][ek[m]], a);
					acc[m] = a;
				}
			}
			__syncthreads();
		}
#pragma unroll
		for (int m = 0; m < 3; ++m)
			if (tid + 256 * m < NE) M[tid + 256 * m] = acc[m];
		__syncthreads();
		const double cnt = M[p * (p + 1) / 2 + p]; // (u, u)
		if (cnt > 0.0) {
			inv_n = 1.0 / cnt;
			// centred Gram matrix; the column sums are row u of the augmented triangle
			for (int idx = tid; idx < p * p; idx += 256) {
				const int i = idx / p, k = idx - i * p;
				if (k <= i) {
					const double si = M[p * (p + 1) / 2 + i], sk = M[p * (p + 1) / 2 + k];
					A[i][k] = M[i * (i + 1) / 2 + k] - si * sk * inv_n;
				}
			}
			if (tid < p) meanv[tid] = cvec[tid] + M[p * (p + 1) / 2 + tid] * inv_n;
			__syncthreads();
			if (tid < p) diag0[tid] = A[tid][tid];
			if (tid == 0) flags = 1;
			__syncthreads();
			// right-looking Cholesky, two barriers per column
			for (int j = 0; j < p; ++j) {
				double dj = A[j][j];
				const bool okj = (dj > kLeverageAliasTolMid * diag0[j]) && (dj > 0.0); // NaN moments fall through here as well
				if (!okj) dj = 1.0;
				const double rl = 1.0 / sqrt(dj);
				if (tid == 0) {
					rdiag[j] = rl;
					if (!okj) flags = 0;
				}
				__syncthreads();
				if (tid > j && tid < p) A[tid][j] *= rl;
				__syncthreads();
				const int rem = p - j - 1; // trailing block: rows / columns j + 1 .. p - 1
				for (int idx = tid; idx < rem * rem; idx += 256) {
					const int i = j + 1 + idx / rem, k = j + 1 + idx % rem;
					if (k <= i) A[i][k] -= A[i][j] * A[k][j];
				}
				// (the next column's barrier orders these updates before its reads)
				__syncthreads();
			}
			// a NaN / inf feature value in a used row poisons every leverage of the group in the reference
			bool bad = false;
			for (int j = 0; j < p; ++j) bad |= !isfinite(M[p * (p + 1) / 2 + j]) || !isfinite(diag0[j]);
			poisoned = bad;
			has_lev = (flags != 0) || poisoned;
		}
	}
	__syncthreads();

	const bool has_stud = has_lev && has_s;
	double n_used = 0.0;
	for (int64_t r = lo + tid; r < hi; r += 256) {
		const double yv = args.y[r], yh = args.y_hat[r];
		const bool used = !drop || (!isnan(yv) && !isnan(yh));
		const double raw = yv - yh;
		double lev = nanv, stud = nanv, stdz = nanv;
		if (has_lev) {
			double h = inv_n;
			double t[kMidMaxP];
#pragma unroll
			for (int i = 0; i < kMidMaxP; ++i) {
				if (i < p) {
					double v = xcol[i][r] - meanv[i];
#pragma unroll
					for (int k = 0; k < i; ++k) v = fma(-A[i][k], t[k], v);
					t[i] = v * rdiag[i];
					h = fma(t[i], t[i], h);
				}
			}
			lev = poisoned ? nanv : h;
			if (has_stud) stud = raw / (s * sqrt(fmax(1.0 - lev, 1e-10)));
		}
		if (has_s) stdz = s > 0.0 ? raw / s : raw;
		double *out = args.out + r * 4;
		out[0] = used ? raw : nanv;
		out[1] = used ? stdz : nanv;
		out[2] = used ? stud : nanv;
		out[3] = used ? lev : nanv;
		n_used += used ? 1.0 : 0.0;
	}
	// block sum of n_used
	__shared__ double red[4];
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) n_used += __shfl_xor(n_used, m, 64);
	if ((tid & 63) == 0) red[tid >> 6] = n_used;
	__syncthreads();
	if (tid == 0) {
		args.group_out[g * 2] = red[0] + red[1] + red[2] + red[3];
		args.group_out[g * 2 + 1] = (double)((has_s ? ANOFOX_HIP_RESIDUALS_HAS_STANDARDIZED : 0) |
		                                      (has_stud ? ANOFOX_HIP_RESIDUALS_HAS_STUDENTIZED : 0) |
		                                      (has_lev ? ANOFOX_HIP_RESIDUALS_HAS_LEVERAGE : 0));
	}
}

} // namespace

// x_table: DEVICE array of p column pointers (the argument struct of the narrow kernel holds only 8)
hipError_t launch_residuals_mid(const ResidualArgs &a, const double *const *d_x_table, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	if (a.p <= kNarrowMaxP || a.p > kMidMaxP) return hipErrorInvalidValue;
	hipLaunchKernelGGL(residuals_mid_kernel, dim3((unsigned)a.n_groups), dim3(256), 0, stream, a, d_x_table);
	return hipGetLastError();
}

} // namespace anofox
