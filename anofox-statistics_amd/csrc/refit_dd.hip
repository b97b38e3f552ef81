// refit_dd.hip — the LAST word on the groups a wide solve queued: the whole fit once more from the rows, in double-double.
//
// Why.  The fit path keeps O(p^2) moments per group and solves normal equations; for the groups whose moments cannot carry the
// answer (small Cholesky pivots, nearly exact fits, nearly square designs — the on-device refinement queue, DESIGN.md "refine")
// residual passes repair the COEFFICIENTS, but two things stay out of reach of a double-precision moment matrix:
//   * diag((X'WX)^-1), hence the standard errors: formed from moments it carries cond(X)^2 eps whatever the refinement does
//     (round 3's deep sweeps: three standard errors of very wide no-intercept designs 1.1 .. 1.8e-6 off, contract 1e-6);
//   * the aliasing decision: the reference's algorithm class (Householder QR, R's rule "remaining norm < 1e-7 of the column's
//     norm", test/data/ols_tests/expected/perfect_collinearity.json) sees a remaining norm to ~1e-16 of the column's; a pivot
//     formed from double-precision moments is noise below ~1e-13 of the diagonal, so the solves used 1e-11 (sin(theta) < 3.2e-6)
//     and disagreed with the checker on the band 1e-7 < sin(theta) < 3.2e-6 (NaN pattern differs).
// Here the moments of (1, x - x_first, y - y_first) are summed over the group's rows in double-double (error-free products, ~32
// digits), centred, factored and inverted in double-double: a Cholesky pivot then IS R's squared remaining norm to ~1e-25 of the
// column's, the 1e-7 rule is applied as the oracle applies it (column order, the norm of the column of the decomposed design),
// and coefficients, RSS, R^2 and diag of the inverse are exact to working precision.  Only queued groups pay: one 256-thread
// workgroup per queued group, the packed triangle (<= 8515 entries x 16 bytes) in LDS.
//
// Reference semantics restated (crates/anofox-stats-core/src/models/{ols,ridge,wls}.rs; oracle/anofox_oracle.c:300-600):
// row filter ols.rs:59-66 / wls.rs:76-86, constant columns and first valid row taken from the moment record (ols.rs:76-87),
// statistics of SURVEY.md Appendix B.7, ridge = (Xc'Xc + lambda I) beta = Xc'yc with glmnet scaling lambda n / sd_y.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "device_math.h"
#include "dd_arith.h"

namespace anofox {
namespace {

#pragma clang fp contract(off)

__device__ __forceinline__ double nan64w() { return __builtin_nan(""); }

// what the kernel needs of a launch, whichever path (narrow / wide) queued the groups
struct RefitArgs {
	const int64_t *row_offsets, *row_ends; // group g owns rows [row_offsets[g], row_ends ? row_ends[g] : row_offsets[g + 1])
	const double *y, *w;
	const double *x[kWideMaxP];
	int64_t group_base;
	int p, model, fit_intercept, lambda_scaling, hc_type;
	double confidence_level, alpha;
	double *core, *inference;
	const int32_t *refine_list, *refine_count;
	void *tcrit_table;
	int max_items; // more queued groups than this: the kernel leaves all of them as the refinement passes left them
};

constexpr int kDdThreads = 256;
constexpr int kDdMaxM = kWideMaxP + 2;                         // ones, x (<= 128), y
constexpr int kDdMaxTri = kDdMaxM * (kDdMaxM + 1) / 2;         // 8515
constexpr int kDdPerThread = (kDdMaxTri + kDdThreads - 1) / kDdThreads; // 34
constexpr int kDdTileRows = 8;

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; } // j <= i

struct DdLds {
	dd *M;          // packed lower triangle of the (m x m) moment matrix, then the centred matrix, then L, then W = L^-1
	double *tile;   // [kDdTileRows][m]: one row tile, z = (1, d_1 .., dy) per row
	double *wrow;   // [kDdTileRows] weight of the row (0 = row does not take part)
	dd *zv;         // [m] the y row of L
	dd *beta;       // [m]
	dd *dg;         // [m] diag((L L')^-1), or the sandwich's diagonal with HC errors
	double *hc_c;   // [2 m] HC pass: the centred row (hi, lo pairs)
	double *hc_t;   // [2 m] HC pass: t = W c
	double *first;  // [m] shift of every variable (0 for the ones)
	int *col;       // [m] original feature of variable v (1 .. pe), -1 otherwise
	int *live;      // [m] 1 = accepted pivot
	double *red;    // [kDdThreads / 64 * 4 + 8] block reductions
	int *flag;      // [kDdTileRows] row validity of the tile being loaded
	int *nc;        // [p] feature is not constant over the valid rows (ols.rs:76-87)
	double *fx;     // [p + 1] the first valid row: x, then y
};

__device__ __forceinline__ dd block_sum_dd(dd v, double *red, int tid) {
	// (a plain tree: 4 waves)
	for (int m = 32; m >= 1; m >>= 1) {
		dd o;
		o.h = __shfl_xor(v.h, m, 64);
		o.l = __shfl_xor(v.l, m, 64);
		v = v + o;
	}
	__syncthreads();
	if ((tid & 63) == 0) {
		red[2 * (tid >> 6)] = v.h;
		red[2 * (tid >> 6) + 1] = v.l;
	}
	__syncthreads();
	dd t = dd{red[0], red[1]};
	for (int w = 1; w < kDdThreads / 64; ++w) t = t + dd{red[2 * w], red[2 * w + 1]};
	return t;
}

__global__ __launch_bounds__(kDdThreads) void refit_dd_kernel(RefitArgs args) {
	extern __shared__ double dd_lds[];
	const int p = args.p;
	const int tid = threadIdx.x;
	const int model = args.model;
	const bool icpt = args.fit_intercept != 0;
	const bool weighted = model == ANOFOX_HIP_MODEL_WLS;
	const bool ridge = model == ANOFOX_HIP_MODEL_RIDGE;
	DdLds l;
	{
		double *q = dd_lds;
		l.M = reinterpret_cast<dd *>(q); q += 2 * kDdMaxTri;
		l.tile = q; q += kDdTileRows * kDdMaxM;
		l.wrow = q; q += kDdTileRows;
		l.zv = reinterpret_cast<dd *>(q); q += 2 * kDdMaxM;
		l.beta = reinterpret_cast<dd *>(q); q += 2 * kDdMaxM;
		l.dg = reinterpret_cast<dd *>(q); q += 2 * kDdMaxM;
		l.hc_c = q; q += 2 * kDdMaxM;
		l.hc_t = q; q += 2 * kDdMaxM;
		l.first = q; q += kDdMaxM;
		l.red = q; q += 16;
		l.col = reinterpret_cast<int *>(q); q += (kDdMaxM + 1) / 2;
		l.live = reinterpret_cast<int *>(q); q += (kDdMaxM + 1) / 2;
		l.flag = reinterpret_cast<int *>(q); q += kDdTileRows;
		l.nc = reinterpret_cast<int *>(q); q += (kWideMaxP + 1) / 2;
		l.fx = q;
	}
	const int n_items = *args.refine_count;
	// A workgroup per queued group and dozens of barriers per row tile: right for the handful of hard groups of an ordinary
	// batch, hopeless for a batch that is ALL exact fits (1e8 noise-free window frames would take hours where the residual
	// passes take a second).  Beyond the cap the groups keep what those passes gave them (round 3's behaviour).
	if (n_items > args.max_items) return;
	for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
		const int64_t gl = args.refine_list[item];
		const int64_t g = args.group_base + gl;
		double *core = args.core + g * (int64_t)(p + 6);
		double *inf = args.inference ? args.inference + g * (int64_t)(5 * p + 2) : nullptr;
		if (core[p + 5] != 0.0) continue; // (queued groups carry a fit; anything else keeps its record)
		const int64_t lo = args.row_offsets[g], hi = args.row_ends ? args.row_ends[g] : args.row_offsets[g + 1];
		// row validity of one tile (ols.rs:59-66, wls.rs:76-86: every feature, y and a positive weight finite — also the
		// constant features); l.wrow = the row's weight, 0 for rows that do not take part
		auto tile_flags = [&](int64_t r0) {
			__syncthreads();
			if (tid < kDdTileRows) l.flag[tid] = 1;
			__syncthreads();
			for (int e = tid; e < kDdTileRows * (p + 2); e += kDdThreads) {
				const int rr = e / (p + 2), c = e - rr * (p + 2);
				const int64_t r = r0 + rr;
				if (r >= hi) continue;
				double v;
				if (c < p) v = args.x[c][r];
				else if (c == p) v = args.y[r];
				else v = weighted ? args.w[r] : 1.0;
				const bool ok = isfinite(v) && (c <= p || v > 0.0);
				if (!ok) l.flag[rr] = 0; // (benign race: every writer writes 0)
				if (c == p + 1) l.wrow[rr] = v;
			}
			__syncthreads();
			if (tid < kDdTileRows) {
				const bool in = r0 + tid < hi && l.flag[tid];
				if (!in) l.wrow[tid] = 0.0;
			}
			__syncthreads();
		};
		// ---- pass 0: the first valid row and the constant-column predicate |x - x_first| < 1e-10 on every valid row (ols.rs:76-87) ----
		int64_t rfirst = -1;
		for (int64_t r0 = lo; r0 < hi && rfirst < 0; r0 += kDdTileRows) {
			tile_flags(r0);
			for (int rr = 0; rr < kDdTileRows; ++rr)
				if (l.wrow[rr] != 0.0) { rfirst = r0 + rr; break; } // (uniform)
		}
		if (rfirst < 0) continue;
		__syncthreads();
		for (int c = tid; c <= p; c += kDdThreads) {
			l.fx[c] = c < p ? args.x[c][rfirst] : args.y[rfirst];
			if (c < p) l.nc[c] = 0;
		}
		__syncthreads();
		double cnt0 = 0.0;
		for (int64_t r0 = lo; r0 < hi; r0 += kDdTileRows) {
			tile_flags(r0);
			for (int rr = 0; rr < kDdTileRows; ++rr) cnt0 += l.wrow[rr] != 0.0 ? 1.0 : 0.0;
			for (int e = tid; e < kDdTileRows * p; e += kDdThreads) {
				const int rr = e / p, c = e - rr * p;
				if (l.wrow[rr] == 0.0) continue;
				if (!(fabs(args.x[c][r0 + rr] - l.fx[c]) < 1e-10)) l.nc[c] = 1; // (benign race)
			}
		}
		__syncthreads();
		if (cnt0 != core[p + 4]) continue; // (must agree with the fit's own row count; uniform)
		// ---- the variables: 0 = ones, 1 .. pe = the non-constant features in their order, pe + 1 = y ----
		if (tid == 0) {
			int pe = 0;
			l.col[0] = -1;
			l.first[0] = 0.0;
			for (int j = 0; j < p; ++j) {
				if (l.nc[j]) {
					++pe;
					l.col[pe] = j;
					l.first[pe] = icpt && !weighted ? l.fx[j] : 0.0; // (weighted: the rows are SCALED below, not shifted)
				}
			}
			l.col[pe + 1] = -1;
			l.first[pe + 1] = icpt && !weighted ? l.fx[p] : 0.0;
			l.flag[kDdTileRows - 1] = pe; // (hand-over below)
		}
		__syncthreads();
		const int pe = l.flag[kDdTileRows - 1];
		const bool hc_active = args.hc_type != ANOFOX_HC_NONE && args.inference && !ridge; // (ridge.rs has no HC branch)
		const int m = pe + 2, yv = pe + 1;
		const int E = m * (m + 1) / 2;
		__syncthreads();
		if (pe == 0) continue;
		// this thread's entries of the packed triangle
		int eij[kDdPerThread]; // (row << 8) | column
		dd acc[kDdPerThread];
#pragma unroll
		for (int k = 0; k < kDdPerThread; ++k) {
			const int e = tid + k * kDdThreads;
			int i = 0;
			if (e < E) { // row of entry e: the largest i with i (i + 1) / 2 <= e
				i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
				while ((i + 1) * (i + 2) / 2 <= e) ++i;
				while (i * (i + 1) / 2 > e) --i;
			}
			eij[k] = (i << 8) | (e < E ? e - i * (i + 1) / 2 : 0);
			acc[k] = dd{0.0, 0.0};
		}
		// ---- the moments over the valid rows, tile by tile ----
		double cnt_rows = 0.0;
		// one tile of rows into l.tile, 0 for rows that do not take part, and l.wrow.  Unweighted: z = (1, x - first .., y - first).
		// Weighted: z = (s, fl(s x) .., fl(s y)) with s = fl(sqrt(w)) — the rows of the sqrt(w)-SCALED design as the reference's
		// algorithm class forms them in working precision (wls.rs:171-176 hands the weights to the solver; the oracle scales the
		// rows, anofox_oracle.c "sw = sqrt(w)").  The exact weighted problem, sum w z z', differs from that one by a relative
		// perturbation eps of every entry of the design: cond(X) eps in the coefficients — 1.1e-9 on an exactly determined
		// 30 x 30 system with cond 2.4e8 (deep sweep, seed 92215), where the oracle agrees with the exact solution of the SCALED
		// system to 2e-13.  Parity is with the reference's formulation, so the scaled rows are what is summed here, exactly.
		auto load_tile = [&](int64_t r0) {
			tile_flags(r0);
			for (int e = tid; e < kDdTileRows * m; e += kDdThreads) {
				const int rr = e / m, v = e - rr * m;
				const int64_t r = r0 + rr;
				double z = 0.0;
				if (r < hi && l.wrow[rr] != 0.0) {
					const double raw = v == 0 ? 1.0 : (v == yv ? args.y[r] : args.x[l.col[v]][r]);
					if (weighted) z = sqrt(l.wrow[rr]) * raw;
					else z = v == 0 ? 1.0 : raw - l.first[v];
				}
				l.tile[rr * m + v] = z;
			}
			__syncthreads();
		};
		for (int64_t r0 = lo; r0 < hi; r0 += kDdTileRows) {
			load_tile(r0);
			for (int rr = 0; rr < kDdTileRows; ++rr) {
				const double wv = l.wrow[rr];
				if (wv == 0.0) continue; // (uniform)
				cnt_rows += 1.0;
				const double *zr = l.tile + rr * m;
#pragma unroll
				for (int k = 0; k < kDdPerThread; ++k) {
					if (tid + k * kDdThreads < E) {
						// (the product z_i z_j exactly; the weight is in the rows, SYMMETRICALLY — rounding w z_i and multiplying by
						// z_j perturbs every entry of the matrix on its own: cond(X)^2 eps, seen as 3e-9 on a weighted 31-column design)
						acc[k] = acc[k] + dd_prod(zr[eij[k] >> 8], zr[eij[k] & 255]);
					}
				}
			}
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < kDdPerThread; ++k)
			if (tid + k * kDdThreads < E) l.M[tid + k * kDdThreads] = acc[k];
		__syncthreads();
		const double cnt = cnt_rows;
		if (cnt < 2.0 || cnt != cnt0) continue; // (uniform)
		const dd sw = l.M[tri(0, 0)];
		// ---- uncentred squared norms of the columns of the decomposed design (the oracle's `full2`), then centring ----
		// with an intercept the record's shift f is in the variables: sum w x^2 = q + 2 f s + f^2 sw
		__syncthreads();
		for (int v = 1 + tid; v <= yv; v += kDdThreads) {
			const dd q = l.M[tri(v, v)], s = l.M[tri(v, 0)];
			const double f = l.first[v];
			dd full = q + dd_mul_d(s, 2.0 * f) + dd_mul_d(sw, f) * dd_make(f);
			l.beta[v] = full; // (scratch until the back substitution)
		}
		__syncthreads();
		dd tss_raw_y = l.beta[yv]; // sum w y^2 (no intercept: the total sum of squares)
		// centring in a race-free order: column 0 (the sums) is read-only here, every thread rewrites only its own entries (j >= 1)
		if (icpt) {
#pragma unroll
			for (int k = 0; k < kDdPerThread; ++k) {
				const int e = tid + k * kDdThreads;
				if (e < E && (eij[k] & 255) >= 1) l.M[e] = l.M[e] - l.M[tri(eij[k] >> 8, 0)] * l.M[tri(eij[k] & 255, 0)] / sw;
			}
		}
		__syncthreads();
		const dd cyy = l.M[tri(yv, yv)]; // centred (intercept) or raw sum w dy^2
		const dd tss = icpt ? cyy : tss_raw_y;
		// ridge penalty (ridge.rs; glmnet scaling: lambda n / sd_y with the population sd of y)
		double lam = 0.0;
		if (ridge) {
			lam = args.alpha;
			if (args.lambda_scaling == ANOFOX_LAMBDA_SCALING_GLMNET) {
				const dd cyy_c = icpt ? cyy : (l.M[tri(yv, yv)] - l.M[tri(yv, 0)] * l.M[tri(yv, 0)] / sw);
				lam = cnt * args.alpha / sqrt(dd_to_double(cyy_c) / cnt);
			}
			for (int v = 1 + tid; v <= pe; v += kDdThreads) l.M[tri(v, v)] = l.M[tri(v, v)] + dd_make(lam);
		}
		__syncthreads();
		// the norm the aliasing rule compares with: OLS / WLS the uncentred column (the oracle decomposes [1 | X]), ridge the
		// column of the centred, augmented design
		for (int v = 1 + tid; v <= pe; v += kDdThreads) {
			if (ridge) l.beta[v] = l.M[tri(v, v)];
			else if (!icpt) l.beta[v] = l.M[tri(v, v)];
		}
		__syncthreads();
		// ---- Cholesky of variables 1 .. pe with the y row carried along, left-looking, R's aliasing rule ----
		int accepted = icpt && !ridge ? 1 : 0; // columns of the decomposed design accepted so far (the ones come first)
		const double rows_of_design = ridge ? cnt + (double)pe : cnt;
		for (int j = 1; j <= pe; ++j) {
			// pivot = C_jj - sum_k L_jk^2 over accepted k < j
			dd part = dd{0.0, 0.0};
			for (int k = 1 + tid; k < j; k += kDdThreads)
				if (l.live[k]) part = part + l.M[tri(j, k)] * l.M[tri(j, k)];
			const dd ssum = block_sum_dd(part, l.red, tid);
			const dd piv = l.M[tri(j, j)] - ssum;
			const dd full = l.beta[j];
			const bool ok = (double)accepted < rows_of_design && piv.h > 0.0 && (piv.h > 1e-14 * full.h);
			__syncthreads();
			if (tid == 0) l.live[j] = ok ? 1 : 0;
			if (!ok) {
				__syncthreads();
				continue; // (uniform)
			}
			++accepted;
			const dd ljj = dd_sqrt(piv);
			// column j below the diagonal (rows j + 1 .. yv)
			for (int i = j + 1 + tid; i <= yv; i += kDdThreads) {
				dd s = l.M[tri(i, j)];
				for (int k = 1; k < j; ++k)
					if (l.live[k]) s = s - l.M[tri(i, k)] * l.M[tri(j, k)];
				l.M[tri(i, j)] = s / ljj;
			}
			if (tid == 0) l.M[tri(j, j)] = ljj;
			__syncthreads();
		}
		__syncthreads();
		// z = the y row of L; rss (OLS / WLS) = C_yy - |z|^2
		dd zz = dd{0.0, 0.0};
		int rank_part = 0;
		for (int j = 1 + tid; j <= pe; j += kDdThreads) {
			const dd z = l.live[j] ? l.M[tri(yv, j)] : dd{0.0, 0.0};
			l.zv[j] = z;
			if (l.live[j]) {
				zz = zz + z * z;
				++rank_part;
			}
		}
		const dd zz_t = block_sum_dd(zz, l.red, tid);
		const dd rk_t = block_sum_dd(dd_make((double)rank_part), l.red, tid);
		const int rank = (int)(rk_t.h + 0.5);
		// ---- W = L^-1 in place (column by column from the right), over the accepted variables ----
		for (int j = pe; j >= 1; --j) {
			__syncthreads();
			if (!l.live[j]) continue; // (uniform)
			const dd wjj = dd_make(1.0) / l.M[tri(j, j)];
			// W(i, j) = - (sum_{k = j+1 .. i} W(i, k) L(k, j)) wjj for i > j: the trailing block is already inverted; column j of L is
			// read by every row, so the new column is written after a barrier (pe <= 128 < 256: one row per thread)
			const int i = j + 1 + tid;
			dd s = dd{0.0, 0.0};
			if (i <= pe && l.live[i]) {
				for (int k = j + 1; k <= i; ++k)
					if (l.live[k]) s = s + l.M[tri(i, k)] * l.M[tri(k, j)];
				s = -(s * wjj);
			}
			__syncthreads();
			if (i <= pe) l.M[tri(i, j)] = s;
			if (tid == 0) l.M[tri(j, j)] = wjj;
		}
		__syncthreads();
		// beta_j = sum_{i >= j} W(i, j) z_i ; diag_j = sum_{i >= j} W(i, j)^2
		for (int j = 1 + tid; j <= pe; j += kDdThreads) {
			dd b = dd{0.0, 0.0}, d = dd{0.0, 0.0};
			if (l.live[j]) {
				for (int i = j; i <= pe; ++i)
					if (l.live[i]) {
						const dd wv = l.M[tri(i, j)];
						b = b + wv * l.zv[i];
						d = d + wv * wv;
					}
			}
			l.beta[j] = b;
			l.dg[j] = d; // (NOT into zv: the other wavefronts still read z — with more than 64 variables that race gave garbage coefficients)
		}
		__syncthreads();
		// ---- statistics (solve_wide.hip's conventions for ridge) ----
		dd bb = dd{0.0, 0.0}, xb = dd{0.0, 0.0};
		for (int j = 1 + tid; j <= pe; j += kDdThreads)
			if (l.live[j]) {
				bb = bb + l.beta[j] * l.beta[j];
				// mean of the variable in original units: first + s / sw
				const dd mu = dd_make(l.first[j]) + (icpt ? l.M[tri(j, 0)] / sw : dd{0.0, 0.0});
				xb = xb + l.beta[j] * mu;
			}
		const dd bb_t = block_sum_dd(bb, l.red, tid);
		const dd xb_t = block_sum_dd(xb, l.red, tid);
		dd rss = cyy - zz_t;
		if (!icpt) rss = tss_raw_y - zz_t;
		if (ridge) rss = rss - dd_mul_d(bb_t, lam);
		double rss_d = dd_to_double(rss) > 0.0 ? dd_to_double(rss) : 0.0;
		const double tss_d = dd_to_double(tss);
		const dd ymean = dd_make(l.first[yv]) + (icpt ? l.M[tri(yv, 0)] / sw : dd{0.0, 0.0});
		const double b0 = icpt ? dd_to_double(ymean - xb_t) : nan64w();
		const int n_par = rank + (icpt ? 1 : 0);
		const double df = cnt - (double)n_par;
		// no residual degrees of freedom: the reference's sigma = sqrt(rss / 0) is +inf because ITS rss is positive rounding noise;
		// an exact 0 here would turn that into 0 / 0 = NaN (and adjusted r^2, F likewise): keep the reference's pattern
		if (df <= 0.0 && rss_d == 0.0) rss_d = 2.2250738585072014e-308;
		const double dfm = (double)rank;
		const double r2 = 1.0 - rss_d / tss_d;
		const double sigma2 = rss_d / df;
		const double fstat = ((tss_d - rss_d) / dfm) / (rss_d / df);
		__syncthreads();
		for (int k = tid; k < p + 4; k += kDdThreads) {
			double v;
			if (k < p) v = nan64w();
			else if (k == p) v = b0;
			else if (k == p + 1) v = r2;
			else if (k == p + 2) v = 1.0 - (1.0 - r2) * (cnt - (icpt ? 1.0 : 0.0)) / df;
			else v = sqrt(sigma2);
			core[k] = v;
		}
		if (inf)
			for (int k = tid; k < 5 * p + 2; k += kDdThreads) inf[k] = nan64w();
		__syncthreads();
		for (int j = 1 + tid; j <= pe; j += kDdThreads)
			if (l.live[j]) core[l.col[j]] = dd_to_double(l.beta[j]);
		if (inf && hc_active) {
			// ---- HC0 .. HC3 (ols.rs:209-245, wls.rs:230-252; the estimator of solve_wide.hip's hc_wide_kernel): one more pass over
			// the rows with S^-1 = W'W applied in double-double, in the variables of the tile (z_0 = s = 1 or sqrt(w)): c = z - xbar z_0
			// (z without an intercept), u = S^-1 c, h = z_0^2 / sum w + c'u, e = (y - fit) z_0, omega = e^2 [n / df | 1 / (1 - h) |
			// 1 / (1 - h)^2], V_jj = sum omega u_j^2 — MacKinnon-White on the sqrt(w)-scaled design, as the oracle applies it.
			const int hc = args.hc_type;
			const double hc1 = cnt / df, h0 = icpt ? 1.0 / dd_to_double(sw) : 0.0;
			// the centred variables: c_j = z_j - s_j / sw (z is shifted by the first row already)
			for (int v = 1 + tid; v <= yv; v += kDdThreads) l.zv[v] = icpt ? l.M[tri(v, 0)] / sw : dd{0.0, 0.0};
			dd vacc = dd{0.0, 0.0};
			const int j = 1 + tid; // this thread's variable (pe <= 128 < 256 threads)
			const bool mine = j <= pe && l.live[j];
			for (int64_t r0 = lo; r0 < hi; r0 += kDdTileRows) {
				load_tile(r0);
				for (int rr = 0; rr < kDdTileRows; ++rr) {
					const double wv = l.wrow[rr];
					if (wv == 0.0) continue; // (uniform)
					const double *zr = l.tile + rr * m;
					// c, then t = W c, are published through two small LDS arrays (every thread needs all of them)
					dd cj = dd{0.0, 0.0};
					if (j <= pe) cj = dd_make(zr[j]) - dd_mul_d(l.zv[j], zr[0]);
					__syncthreads();
					if (j <= pe) {
						l.hc_c[2 * j] = cj.h;
						l.hc_c[2 * j + 1] = cj.l;
					}
					__syncthreads();
					dd ti = dd{0.0, 0.0};
					if (mine)
						for (int k = 1; k <= j; ++k)
							if (l.live[k]) ti = ti + l.M[tri(j, k)] * dd{l.hc_c[2 * k], l.hc_c[2 * k + 1]};
					__syncthreads();
					if (j <= pe) {
						l.hc_t[2 * j] = ti.h;
						l.hc_t[2 * j + 1] = ti.l;
					}
					__syncthreads();
					dd uj = dd{0.0, 0.0};
					if (mine)
						for (int i = j; i <= pe; ++i)
							if (l.live[i]) uj = uj + l.M[tri(i, j)] * dd{l.hc_t[2 * i], l.hc_t[2 * i + 1]};
					// c'u and b'c
					const dd hp = block_sum_dd(mine ? cj * uj : dd{0.0, 0.0}, l.red, tid);
					const dd ep = block_sum_dd(mine ? l.beta[j] * cj : dd{0.0, 0.0}, l.red, tid);
					const dd ycen = dd_make(zr[yv]) - dd_mul_d(l.zv[yv], zr[0]); // (y - ybar) z_0 (y z_0 without an intercept)
					const double e = dd_to_double(ycen - ep);
					const double lev = zr[0] * zr[0] * h0 + dd_to_double(hp);
					double om = e * e;
					if (hc == ANOFOX_HC_HC1) om *= hc1;
					else if (hc == ANOFOX_HC_HC2) om /= (1.0 - lev);
					else if (hc == ANOFOX_HC_HC3) om /= (1.0 - lev) * (1.0 - lev);
					if (mine && isfinite(e) && isfinite(lev)) vacc = vacc + dd_mul_d(uj * uj, om); // (hc_wide_kernel's rule for a row that counts)
				}
			}
			__syncthreads();
			if (j <= pe) l.dg[j] = mine ? vacc : dd{0.0, 0.0}; // the sandwich's diagonal: the standard errors' squares
			__syncthreads();
		}
		if (inf) {
			if (tid == 0) l.red[8] = dm_tcrit_cached(static_cast<TcritSlot *>(args.tcrit_table), 0.5 * (1.0 + args.confidence_level), df);
			__syncthreads();
			const double tcrit = l.red[8];
			for (int j = 1 + tid; j <= pe; j += kDdThreads) {
				if (!l.live[j]) continue;
				const int c = l.col[j];
				const double b = dd_to_double(l.beta[j]);
				const double se = hc_active ? sqrt(dd_to_double(l.dg[j])) : sqrt(sigma2 * dd_to_double(l.dg[j]));
				const double tv = b / se;
				inf[c] = se;
				inf[p + c] = tv;
				inf[2 * p + c] = dm_t_two_sided_p(tv, df);
				inf[3 * p + c] = b - tcrit * se;
				inf[4 * p + c] = b + tcrit * se;
			}
			if (tid == 0) {
				inf[5 * p] = fstat;
				inf[5 * p + 1] = dm_f_sf(fstat, dfm, df);
			}
		}
	}
}

} // namespace

size_t refit_dd_lds_bytes() {
	return sizeof(double) * (2 * (size_t)kDdMaxTri + (size_t)kDdTileRows * kDdMaxM + kDdTileRows + 10 * (size_t)kDdMaxM + kDdMaxM + 16 +
	                         2 * ((kDdMaxM + 1) / 2) + kDdTileRows + (kWideMaxP + 1) / 2 + kWideMaxP + 1);
}

namespace {
// ~1-2 ms per wide group and workgroup, ~0.3 ms per narrow one; 256-512 workgroups in flight: a cap's worth costs 40-50 ms
int refit_dd_cap(int p) {
	static const int forced = getenv("ANOFOX_REFIT_DD_MAX") ? atoi(getenv("ANOFOX_REFIT_DD_MAX")) : -1;
	if (forced >= 0) return forced;
	return p <= kNarrowMaxP ? 65536 : 8192;
}
hipError_t launch_refit_dd(const RefitArgs &r, hipStream_t stream) {
	static const bool attr_set = [] {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&refit_dd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		return true;
	}();
	(void)attr_set;
	hipLaunchKernelGGL(refit_dd_kernel, dim3(256), dim3(kDdThreads), refit_dd_lds_bytes(), stream, r);
	return hipGetLastError();
}
} // namespace

// after every other kernel of the fit: the queued groups' records once more, in double-double
hipError_t launch_refit_dd_wide(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	RefitArgs r;
	memset(&r, 0, sizeof r);
	r.row_offsets = a.row_offsets;
	r.row_ends = a.row_ends;
	r.y = a.y;
	r.w = a.w;
	for (int j = 0; j < a.p; ++j) r.x[j] = a.x_table[j];
	r.group_base = a.group_base;
	r.p = a.p; r.model = a.model; r.fit_intercept = a.fit_intercept; r.lambda_scaling = a.lambda_scaling; r.hc_type = a.hc_type;
	r.confidence_level = a.confidence_level; r.alpha = a.alpha;
	r.core = a.core; r.inference = a.inference;
	r.refine_list = a.refine_list; r.refine_count = a.refine_count;
	r.tcrit_table = a.tcrit_table;
	r.max_items = refit_dd_cap(a.p);
	return launch_refit_dd(r, stream);
}

// the same behind the narrow path (p <= 8): its refinement queue, its records
hipError_t launch_refit_dd_narrow(const BatchArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	RefitArgs r;
	memset(&r, 0, sizeof r);
	r.row_offsets = a.row_offsets;
	r.row_ends = a.row_ends;
	r.y = a.y;
	r.w = a.w;
	for (int j = 0; j < a.p; ++j) r.x[j] = a.x[j];
	r.group_base = 0;
	r.p = a.p; r.model = a.model; r.fit_intercept = a.fit_intercept; r.lambda_scaling = a.lambda_scaling; r.hc_type = a.hc_type;
	r.confidence_level = a.confidence_level; r.alpha = a.alpha;
	r.core = a.core; r.inference = a.inference;
	r.refine_list = a.refine_list; r.refine_count = a.refine_count;
	r.tcrit_table = a.tcrit_table;
	r.max_items = refit_dd_cap(a.p);
	return launch_refit_dd(r, stream);
}

} // namespace anofox
