// residuals_wide.hip — residual diagnostics (raw / standardized / studentized residuals, leverage) for 9..128
// features: one 256-thread workgroup per group.  Same definitions, flags and conventions as residuals_narrow.hip
// (which see for the reference citations: crates/anofox-stats-core/src/diagnostics/residuals.rs:30-145,
// src/aggregate_functions/residuals_diagnostics_aggregate.cpp:154-163,213-286); what changes is where the
// (p+1)(p+2)/2 moments live — not in one lane's registers any more:
//   moments   rows are staged 32 at a time into an LDS tile [32][p + 2] (d = x - x_first of the used rows, plus a
//             column u = 1 for used rows); thread t owns the entries t, t + 256, ... (up to 33) of the packed triangle
//             of the augmented Gram matrix (d, u)'(d, u);
//   factor    the centred Gram matrix replaces the x part of that triangle in place (packed, row-major), then a
//             right-looking Cholesky, all threads, barriers per column;
//   rows      one thread per row: forward substitution t = L^-1 (x - mean), h = 1/n + |t|^2, with t in the thread's
//             private (scratch) memory — 128 values do not fit the registers next to the loop.
// A diagnostics function, not a hot path: the moments are summed on the vector units (O(n p^2 / 256) per thread).
// (It replaced round 2's first kernel for 9..32 features — 64-row tiles, three fixed entries per thread, a full
// square for the factor: 20 000 groups x 500 rows, p = 9 / 16 / 24 / 32: 4.47 / 7.0 / 10.7 / 13.6 ms there, 1.66 / 3.0 /
// 5.1 / 8.3 ms here; p = 64: 27.7 ms, p = 128 (5000 x 1000 rows): 89 ms.)
#include "common.h"

namespace anofox {

namespace {

constexpr double kLeverageAliasTolWide = 1e-11;
constexpr int kResWideMaxP = kWideMaxP; // 128
constexpr int kResTileRows = 32;
constexpr int kResMaxOwn = ((kResWideMaxP + 1) * (kResWideMaxP + 2) / 2 + 255) / 256; // packed entries per thread

__device__ __forceinline__ int res_tri(int i, int k) { return i * (i + 1) / 2 + k; } // k <= i

__global__ __launch_bounds__(256) void residuals_wide_kernel(ResidualArgs args, const double *const *x_table) {
	const int p = args.p, Z = p + 1, NE = Z * (Z + 1) / 2, TS = Z + 1; // TS: tile row stride (odd or even: either way no 2^k)
	const int tid = threadIdx.x;
	const int64_t g = blockIdx.x;
	const int64_t lo = args.row_offsets[g], hi = args.row_offsets[g + 1];
	const bool drop = args.drop_nan_rows != 0;
	const double nanv = __builtin_nan("");
	const double s = args.rse ? args.rse[g] : nanv;
	const bool has_s = !isnan(s);

	extern __shared__ double sm[];
	double *tile = sm;                              // [kResTileRows][TS]
	double *M = tile + kResTileRows * TS;           // packed lower triangle of the augmented moments, row-major; then L
	double *cvec = M + NE;                          // [p] x at the first used row
	double *meanv = cvec + p;                       // [p]
	double *rdiag = meanv + p;                      // [p] 1 / L_jj
	double *diag0 = rdiag + p;                      // [p]
	double *red = diag0 + p;                        // [4]
	const double **xcol = reinterpret_cast<const double **>(red + 4); // [p]
	int *ints = reinterpret_cast<int *>(xcol + p);  // first_row, flags (1 = factor ok)

	for (int j = tid; j < p; j += 256) xcol[j] = x_table[j];
	if (tid == 0) { ints[0] = 0x7fffffff; ints[1] = 0; }
	__syncthreads();

	bool has_lev = false, poisoned = false;
	double inv_n = 0.0;
	if (args.include_studentized) {
		// the group's first used row centres the sums
		for (int64_t base = lo; base < hi; base += 256) {
			const int64_t r = base + tid;
			bool used = r < hi;
			if (used && drop) used = !isnan(args.y[r]) && !isnan(args.y_hat[r]);
			if (used) atomicMin(&ints[0], (int)(r - lo));
			__syncthreads();
			if (ints[0] != 0x7fffffff) break; // uniform: read after the barrier
			__syncthreads();
		}
		__syncthreads();
		const int64_t first = ints[0] == 0x7fffffff ? -1 : lo + ints[0];
		for (int j = tid; j < p; j += 256) cvec[j] = first >= 0 ? xcol[j][first] : 0.0;
		// this thread's entries of the packed triangle: e = tid + 256 m -> (row ej, column ek)
		short ej[kResMaxOwn], ek[kResMaxOwn];
		double acc[kResMaxOwn];
		const int n_own = (NE - tid + 255) / 256; // entries tid, tid + 256, ... below NE
#pragma unroll 1
		for (int m = 0; m < kResMaxOwn; ++m) {
			const int e = tid + 256 * m;
			int j = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
			while (j * (j + 1) / 2 > e) --j;
			while ((j + 1) * (j + 2) / 2 <= e) ++j;
			ej[m] = (short)(e < NE ? j : 0);
			ek[m] = (short)(e < NE ? e - j * (j + 1) / 2 : 0);
			acc[m] = 0.0;
		}
		__syncthreads();
		for (int64_t base = lo; base < hi; base += kResTileRows) {
			// stage 32 rows: element idx -> (column idx / 32, row idx % 32): 256 contiguous bytes per column
			for (int idx = tid; idx < kResTileRows * Z; idx += 256) {
				const int col = idx / kResTileRows, row = idx - col * kResTileRows;
				const int64_t r = base + row;
				bool used = r < hi;
				if (used && drop) used = !isnan(args.y[r]) && !isnan(args.y_hat[r]);
				double v = 0.0;
				if (used) v = col < p ? xcol[col][r] - cvec[col] : 1.0;
				tile[row * TS + col] = v;
			}
			__syncthreads();
#pragma unroll 1
			for (int m = 0; m < n_own; ++m) {
				double a = acc[m];
				const int cj = ej[m], ck = ek[m];
#pragma unroll 8
				for (int row = 0; row < kResTileRows; ++row) a = fma(tile[row * TS + cj], tile[row * TS + ck], a);
				acc[m] = a;
			}
			__syncthreads();
		}
#pragma unroll 1
		for (int m = 0; m < n_own; ++m) M[tid + 256 * m] = acc[m];
		__syncthreads();
		const double cnt = M[res_tri(p, p)]; // (u, u)
		if (cnt > 0.0) {
			inv_n = 1.0 / cnt;
			// a NaN / inf feature value in a used row poisons every leverage of the group in the reference
			// (the column sums are row u of the augmented triangle; checked before they are consumed below)
			bool bad = false;
			for (int j = 0; j < p; ++j) bad |= !isfinite(M[res_tri(p, j)]);
			for (int j = tid; j < p; j += 256) meanv[j] = cvec[j] + M[res_tri(p, j)] * inv_n;
			__syncthreads();
			// centred Gram matrix in place (rows 0 .. p - 1 of the triangle)
			for (int e = tid; e < p * (p + 1) / 2; e += 256) {
				int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
				while (i * (i + 1) / 2 > e) --i;
				while ((i + 1) * (i + 2) / 2 <= e) ++i;
				const int k = e - i * (i + 1) / 2;
				M[e] -= M[res_tri(p, i)] * M[res_tri(p, k)] * inv_n;
			}
			__syncthreads();
			for (int j = tid; j < p; j += 256) diag0[j] = M[res_tri(j, j)];
			if (tid == 0) ints[1] = 1;
			__syncthreads();
			for (int j = 0; j < p; ++j) bad |= !isfinite(diag0[j]);
			// right-looking Cholesky in the packed triangle
			for (int j = 0; j < p; ++j) {
				double dj = M[res_tri(j, j)];
				const bool okj = (dj > kLeverageAliasTolWide * diag0[j]) && (dj > 0.0); // NaN moments fall through here as well
				if (!okj) dj = 1.0;
				const double rl = 1.0 / sqrt(dj);
				__syncthreads(); // every thread has read the pivot
				if (tid == 0) {
					rdiag[j] = rl;
					if (!okj) ints[1] = 0;
				}
				for (int i = j + 1 + tid; i < p; i += 256) M[res_tri(i, j)] *= rl;
				__syncthreads();
				const int rem = p - j - 1; // trailing block: rows / columns j + 1 .. p - 1
				for (int idx = tid; idx < rem * rem; idx += 256) {
					const int i = j + 1 + idx / rem, k = j + 1 + idx % rem;
					if (k <= i) M[res_tri(i, k)] -= M[res_tri(i, j)] * M[res_tri(k, j)];
				}
				__syncthreads();
			}
			poisoned = bad;
			has_lev = (ints[1] != 0) || poisoned;
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
			double t[kResWideMaxP]; // private memory (dynamically indexed)
#pragma unroll 1
			for (int i = 0; i < p; ++i) {
				double v = xcol[i][r] - meanv[i];
				const double *Li = M + res_tri(i, 0);
#pragma unroll 4
				for (int k = 0; k < i; ++k) v = fma(-Li[k], t[k], v);
				const double ti = v * rdiag[i];
				t[i] = ti;
				h = fma(ti, ti, h);
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
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) n_used += __shfl_xor(n_used, m, 64);
	__syncthreads();
	if ((tid & 63) == 0) red[tid >> 6] = n_used;
	__syncthreads();
	if (tid == 0) {
		args.group_out[g * 2] = red[0] + red[1] + red[2] + red[3];
		args.group_out[g * 2 + 1] = (double)((has_s ? ANOFOX_HIP_RESIDUALS_HAS_STANDARDIZED : 0) |
		                                      (has_stud ? ANOFOX_HIP_RESIDUALS_HAS_STUDENTIZED : 0) |
		                                      (has_lev ? ANOFOX_HIP_RESIDUALS_HAS_LEVERAGE : 0));
	}
}

size_t residuals_wide_lds_bytes(int p) {
	const size_t Z = (size_t)p + 1, NE = Z * (Z + 1) / 2, TS = Z + 1;
	return (kResTileRows * TS + NE + 4 * (size_t)p + 4) * sizeof(double) + (size_t)p * sizeof(double *) + 16;
}

} // namespace

// x_table: DEVICE array of p column pointers
hipError_t launch_residuals_wide(const ResidualArgs &a, const double *const *d_x_table, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	if (a.p <= kNarrowMaxP || a.p > kResWideMaxP) return hipErrorInvalidValue;
	const size_t lds = residuals_wide_lds_bytes(a.p);
	static bool attr_done = false; // (idempotent; a race sets it twice at worst)
	if (!attr_done) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&residuals_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		attr_done = true;
	}
	hipLaunchKernelGGL(residuals_wide_kernel, dim3((unsigned)a.n_groups), dim3(256), lds, stream, a, d_x_table);
	return hipGetLastError();
}

} // namespace anofox
