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
#include <stdlib.h>

#include "common.h"
#include <mutex>
#include "device_math.h"
#include "dd_arith.h"

namespace anofox {

typedef double sw_dbl4 __attribute__((ext_vector_type(4)));

namespace {

// Diagnostic build only (-DANOFOX_SOLVE_STAMPS, csrc/Makefile target `diag`): workgroup 0 records s_memtime at
// the phase boundaries of its first group into a buffer nothing else reads.  Not compiled into the product.
#ifdef ANOFOX_SOLVE_STAMPS
__device__ unsigned long long g_solve_stamps[32];
#define SOLVE_STAMP(k)                                                                  \
	do {                                                                                \
		if (blockIdx.x == 0 && threadIdx.x == 0 && item == 0) g_solve_stamps[k] = __builtin_amdgcn_s_memtime(); \
	} while (0)
#else
#define SOLVE_STAMP(k) do { } while (0)
#endif

constexpr double kAliasTolW = 1e-11;
constexpr double kRefineTolW = 1e-7;
constexpr double kPivotWarnW = 1e-3;
enum { MODE_PRIMARY = 0, MODE_UPDATE = 1, MODE_FINAL = 2 };

__device__ __forceinline__ double nan64w() { return __builtin_nan(""); }

// LDS layout of one group (doubles): A[(P16+1) x LD] | sv | fx | diag0 | ldiag | linv | zv | bv | red[16]
//   then ints: active[P16] | live[P16]
// A rows 0..P16-1 are the (zero padded) x columns, row P16 is the y row of the augmented matrix
//   [ Sxx  Sxy ]      Cholesky of the leading block leaves  z = L^-1 Sxy  in the y row, so the forward solve
//   [ Sxy' Syy ]      is free and RSS = Syy - |z|^2.
// Lower triangle: L.  Upper triangle: W = L^-1 stored transposed (W[i][j] at A[j][i], i > j), diag of W in linv.
struct WideLds {
	int P16, LD, T;
	double *A, *sv, *fx, *diag0, *ldiag, *linv, *zv, *bv, *red;
	int *active, *live;
};

__device__ __forceinline__ WideLds carve_lds(double *sm, int p) {
	WideLds l;
	l.T = wide_tiles(p);
	l.P16 = 16 * l.T;
	l.LD = l.P16 + 1;
	l.A = sm;
	l.sv = l.A + (size_t)(l.P16 + 1) * l.LD;
	l.fx = l.sv + l.P16;
	l.diag0 = l.fx + l.P16;
	l.ldiag = l.diag0 + l.P16;
	l.linv = l.ldiag + l.P16;
	l.zv = l.linv + l.P16;
	l.bv = l.zv + l.P16;
	l.red = l.bv + l.P16;
	l.active = reinterpret_cast<int *>(l.red + 16);
	l.live = l.active + l.P16;
	return l;
}

__device__ __forceinline__ double rl_f64(double v, int src_lane) {
	return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane),
	                        __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}

// NTHR = threads of the workgroup that solves one group: 256 (four waves share the group), or 64 for the narrower
// designs (p <= 64), where one wave per group and several groups per CU beat four waves that mostly wait for the
// single-wave phases of the factorisation.
// sum over the threads of a workgroup; every thread gets the result.  `slot` = 4 doubles of LDS scratch.
template <int NTHR>
__device__ __forceinline__ double block_sum(double v, double *slot, int tid) {
	for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
	if (NTHR == 64) {
		__syncthreads(); // (callers count on the barrier between what they wrote to LDS before and read after)
		return v;
	}
	if ((tid & 63) == 0) slot[tid >> 6] = v;
	__syncthreads();
	double tot = slot[0];
#pragma unroll
	for (int w = 1; w < NTHR / 64; ++w) tot += slot[w];
	__syncthreads();
	return tot;
}

// t -> (a, b) with b <= a, t = a(a+1)/2 + b
__device__ __forceinline__ void tri_decode(int t, int &a, int &b) {
	a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
	while (a * (a + 1) / 2 > t) --a;
	while ((a + 1) * (a + 2) / 2 <= t) ++a;
	b = t - a * (a + 1) / 2;
}

__device__ __forceinline__ void wave_lds_sync() {
	// orders the LDS traffic of the lanes of ONE wavefront (used where a single wave works on a 16x16 block)
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// W[kk][jj] for kk >= jj
__device__ __forceinline__ double getW(const WideLds &l, int kk, int jj) {
	return kk > jj ? l.A[(size_t)jj * l.LD + kk] : (kk == jj ? l.linv[kk] : 0.0);
}

// Blocked right-looking Cholesky of the augmented matrix in LDS (block size 16), deactivating constant and
// aliased columns in place.  Returns (to every thread) the smallest accepted pivot ratio.
template <int NTHR>
__device__ double blocked_cholesky(const WideLds &l, int tid) {
	const int P16 = l.P16, LD = l.LD, T = l.T;
	double *A = l.A;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	if (tid == 0) l.red[7] = 1.0;
	for (int kb = 0; kb < T; ++kb) {
		const int k0 = 16 * kb;
		__syncthreads();
#ifdef ANOFOX_SOLVE_STAMPS
		if (blockIdx.x == 0 && tid == 0 && kb == 0) g_solve_stamps[8] = __builtin_amdgcn_s_memtime();
#endif
		if (wave == 0) {
			// (a) factor the 16x16 diagonal block in registers: lane r (mod 16) owns row r; column j is finished
			// left-looking with the already final columns k < j, whose row-j entries are broadcast by readlane.
			const int r = lane & 15;
			double arow[16]; // row r of the block; column j becomes L[r][j] at step j (right-looking: the
			                 // 15-j updates of a step are independent, the only chain is pivot -> rsqrt -> scale)
#pragma unroll
			for (int c = 0; c < 16; ++c) arow[c] = A[(size_t)(k0 + r) * LD + k0 + c];
			double min_ratio = l.red[7];
			const double d0 = l.diag0[k0 + r];
			const double d0inv = 1.0 / d0;
			const int act = l.active[k0 + r];
			// Pivot j: the pivot is wave-uniform, but it is deliberately laundered into a vector register: with a scalar
			// condition the compiler emits ~8 scalar branches per pivot, with a vector one plain selects.
			struct Pivot {
				bool ok;
				double inv, ljj, ratio;
			};
			auto pivot = [&](int j) {
				double d = rl_f64(arow[j], j);
				double dj0 = rl_f64(d0, j);
				int actj = __builtin_amdgcn_readlane(act, j);
				asm volatile("" : "+v"(d), "+v"(dj0), "+v"(actj));
				Pivot pv;
				pv.ok = (actj != 0) && (d > kAliasTolW * dj0) && (d > 0.0);
				const double inv0 = rsqrt(pv.ok ? d : 1.0);
				pv.inv = pv.ok ? inv0 : 0.0;
				pv.ljj = pv.ok ? d * inv0 : 1.0;
				pv.ratio = d * rl_f64(d0inv, j);
				return pv;
			};
			// The chain pivot -> rsqrt -> scale of step j + 1 needs only the FIRST update of step j (column j + 1): it is
			// started right behind that update, and the other 14 - j updates of step j run while its rsqrt is under way.
			Pivot pv = pivot(0);
#pragma unroll
			for (int j = 0; j < 16; ++j) {
				const bool ok = pv.ok;
				const double inv = pv.inv, ljj = pv.ljj;
				min_ratio = ok ? fmin(min_ratio, pv.ratio) : min_ratio;
				const double lrj = (r > j) ? arow[j] * inv : 0.0; // aliased / constant: column := 0
				arow[j] = (r == j) ? ljj : lrj;
				if (j + 1 < 16) {
					arow[j + 1] -= lrj * rl_f64(lrj, j + 1); // L[r][j] * L[j + 1][j]
					pv = pivot(j + 1);
				}
#pragma unroll
				for (int c = j + 2; c < 16; ++c) arow[c] -= lrj * rl_f64(lrj, c); // L[r][j] * L[c][j]
				if (lane == j) {
					l.ldiag[k0 + j] = ljj;
					l.linv[k0 + j] = inv;
					l.live[k0 + j] = ok ? 1 : 0;
				}
			}
			double (&lrow)[16] = arow;
			if (lane < 16) {
#pragma unroll
				for (int c = 0; c < 16; ++c)
					if (c < r) A[(size_t)(k0 + r) * LD + k0 + c] = lrow[c];
			}
			if (lane == 0) l.red[7] = min_ratio;
		}
		__syncthreads();
#ifdef ANOFOX_SOLVE_STAMPS
		if (blockIdx.x == 0 && tid == 0 && kb == 0) g_solve_stamps[9] = __builtin_amdgcn_s_memtime();
#endif
		// (b) panel: rows below the block (and the y row): X := X L_kk^-T, one thread per row
		const int n_below = P16 - k0 - 16 + 1;
		for (int rb = tid; rb < n_below; rb += NTHR) {
			const int i = k0 + 16 + rb;
			double x[16];
#pragma unroll
			for (int c = 0; c < 16; ++c) x[c] = A[(size_t)i * LD + k0 + c];
#pragma unroll
			for (int c = 0; c < 16; ++c) {
				double sacc = x[c];
#pragma unroll
				for (int m = 0; m < c; ++m) sacc -= x[m] * A[(size_t)(k0 + c) * LD + k0 + m];
				x[c] = sacc * l.linv[k0 + c];
			}
#pragma unroll
			for (int c = 0; c < 16; ++c) A[(size_t)i * LD + k0 + c] = x[c];
		}
		__syncthreads();
#ifdef ANOFOX_SOLVE_STAMPS
		if (blockIdx.x == 0 && tid == 0 && kb == 0) g_solve_stamps[10] = __builtin_amdgcn_s_memtime();
#endif
		// (c) trailing update with the 16 new columns on the matrix cores: one 16x16 tile of the lower triangle per
		// wave and trip, C -= L_i L_c' as four v_mfma_f64_16x16x4 (the 4x4 register tiles this replaces read eight
		// LDS operands per 16 FMAs and were LDS-bound).  Diagonal tiles are updated in full: their upper halves land
		// in the upper triangle, which nothing reads before blocked_tri_inverse clears it.
		const int mrem = P16 - k0 - 16; // remaining x rows (multiple of 16)
		const int nt = mrem >> 2;
		{
			const int nb = mrem >> 4;
			const int ntri = nb * (nb + 1) / 2;
			const int fr = lane & 15, fk = lane >> 4;
			for (int t = wave; t < ntri; t += NTHR / 64) {
				int a16, b16;
				tri_decode(t, a16, b16);
				const int i0 = k0 + 16 + 16 * a16, c0 = k0 + 16 + 16 * b16;
				sw_dbl4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const double aop = A[(size_t)(i0 + fr) * LD + k0 + 4 * q + fk];
					const double bop = A[(size_t)(c0 + fr) * LD + k0 + 4 * q + fk];
					acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
				}
#pragma unroll
				for (int r = 0; r < 4; ++r) A[(size_t)(i0 + fk + 4 * r) * LD + c0 + fr] -= acc[r];
			}
		}
		// y row: 1 x 4 tiles
		for (int b4 = tid; b4 < nt; b4 += NTHR) {
			const int c0 = k0 + 16 + 4 * b4;
			double acc[4] = {0.0, 0.0, 0.0, 0.0};
			for (int m = 0; m < 16; ++m) {
				const double xa = A[(size_t)P16 * LD + k0 + m];
#pragma unroll
				for (int cc = 0; cc < 4; ++cc) acc[cc] = fma(xa, A[(size_t)(c0 + cc) * LD + k0 + m], acc[cc]);
			}
#pragma unroll
			for (int cc = 0; cc < 4; ++cc) A[(size_t)P16 * LD + c0 + cc] -= acc[cc];
		}
#ifdef ANOFOX_SOLVE_STAMPS
		__syncthreads();
		if (blockIdx.x == 0 && tid == 0 && kb == 0) g_solve_stamps[11] = __builtin_amdgcn_s_memtime();
#endif
	}
	__syncthreads();
	return l.red[7];
}

// W = L^-1, right-looking by 16-row blocks, solving L W = I:  block row kb of W is finished by a triangular
// solve with L_kk (one thread per column), then the rows below receive the rank-16 update
// RHS[i][j] -= sum_m L[i][k0+m] W[k0+m][j] in 4x4 register tiles.  W[i][j], i > j, lives at A[j][i].
template <int NTHR>
__device__ void blocked_tri_inverse(const WideLds &l, int tid) {
	const int P16 = l.P16, LD = l.LD, T = l.T;
	double *A = l.A;
	// RHS = I: clear the strictly upper triangle (the transposed strictly lower part of the RHS)
	for (int idx = tid; idx < P16 * P16; idx += NTHR) {
		const int j = idx / P16, i = idx - j * P16;
		if (i > j) A[(size_t)j * LD + i] = 0.0;
	}
	__syncthreads();
	for (int kb = 0; kb < T; ++kb) {
		const int k0 = 16 * kb;
		// (i) block row kb: x = L_kk^-1 rhs, one thread per column j <= k0 + 15
		for (int j = tid; j < k0 + 16; j += NTHR) {
			double x[16];
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const double raw = A[(size_t)j * LD + k0 + r];         // accumulated RHS (columns left of the block)
				x[r] = (k0 + r > j) ? raw : ((k0 + r == j) ? 1.0 : 0.0); // identity inside the block
			}
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				double sacc = x[r];
#pragma unroll
				for (int m = 0; m < r; ++m) sacc -= A[(size_t)(k0 + r) * LD + k0 + m] * x[m];
				x[r] = sacc * l.linv[k0 + r];
			}
#pragma unroll
			for (int r = 0; r < 16; ++r)
				if (k0 + r > j) A[(size_t)j * LD + k0 + r] = x[r];
		}
		__syncthreads();
		// (ii) rows below: 16x16 tiles over (P16 - k0 - 16) x (k0 + 16) on the matrix cores, RHS -= L_i W_kb
		{
			const int nrb = (P16 - k0 - 16) >> 4, ncb = (k0 + 16) >> 4;
			const int lane = tid & 63, wave = tid >> 6;
			const int fr = lane & 15, fk = lane >> 4;
			for (int t = wave; t < nrb * ncb; t += NTHR / 64) {
				const int a16 = t / ncb, b16 = t - a16 * ncb;
				const int i0 = k0 + 16 + 16 * a16, j0 = 16 * b16;
				const bool inblock = j0 >= k0; // columns of the current block: W's block is lower triangular
				sw_dbl4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const int kk = k0 + 4 * q + fk, jj = j0 + fr;
					const double aop = A[(size_t)(i0 + fr) * LD + kk];
					const double raw = A[(size_t)jj * LD + kk];
					const double bop = (!inblock || kk > jj) ? raw : ((kk == jj) ? l.linv[jj] : 0.0);
					acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
				}
#pragma unroll
				for (int r = 0; r < 4; ++r) A[(size_t)(j0 + fr) * LD + i0 + fk + 4 * r] -= acc[r];
			}
		}
		__syncthreads();
	}
}

// beta = L^-T z by 16-column blocks from the bottom: wave 0 finishes the 16 unknowns of a block with readlane
// broadcasts, then every thread removes their contribution from the rows above.  zv is consumed, bv := beta.
template <int NTHR>
__device__ void blocked_back_solve(const WideLds &l, int tid) {
	const int LD = l.LD, T = l.T;
	const double *A = l.A;
	const int lane = tid & 63;
	for (int kb = T - 1; kb >= 0; --kb) {
		const int k0 = 16 * kb;
		if (tid < 64) {
			const int r = lane & 15;
			double zr = l.zv[k0 + r];
			const double inv = l.linv[k0 + r];
			double col[16]; // L[k0+m][k0+r], m > r: column r of the block = coefficients of beta_m in equation r
#pragma unroll
			for (int m = 0; m < 16; ++m) col[m] = A[(size_t)(k0 + m) * LD + k0 + r];
			double br = 0.0;
#pragma unroll
			for (int m = 15; m >= 0; --m) {
				const double bm = rl_f64(zr * inv, m); // beta_m once every later unknown has been removed from z_m
				if (r == m) br = bm;
				zr -= (r < m) ? col[m] * bm : 0.0;
			}
			if (lane < 16) l.bv[k0 + r] = br;
		}
		__syncthreads();
		for (int i = tid; i < k0; i += NTHR) {
			double zi = l.zv[i];
#pragma unroll
			for (int m = 0; m < 16; ++m) zi -= A[(size_t)(k0 + m) * LD + i] * l.bv[k0 + m];
			l.zv[i] = zi;
		}
		__syncthreads();
	}
}

// Augmented moment matrix -> LDS, lower triangle; centred when an intercept is fitted; padding rows / columns zero.
// The record is tile-major (256 contiguous doubles per 16x16 tile): thread t reads element t of each tile.
// Needs l.sv (column sums) in place.
template <int NTHR>
__device__ void load_moment_matrix(const WideLds &l, const double *rec, int p, bool icpt, double lam, double sw, double sy, int tid) {
	const int T = l.T, P16 = l.P16, LD = l.LD;
	const int NT = T * (T + 1) / 2;
	const double *vec = rec + (int64_t)NT * 256;
	double *A = l.A;
	const double inv_sw = 1.0 / sw;
	// four tiles per trip so that four independent 2 KiB loads are in flight (the record comes from HBM / L2); a
	// 64-thread workgroup takes the four quarters of a tile one after the other
	for (int t0 = 0; t0 < NT; t0 += 4) {
#pragma unroll 1
		for (int e0 = 0; e0 < 256; e0 += NTHR) {
			const int el = e0 + tid;
			const int tr = el >> 4, tc = el & 15; // element (tr, tc) of an upper-triangular tile = M[16I+tr][16J+tc]
			double v4[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) v4[u] = rec[(int64_t)((t0 + u < NT) ? t0 + u : NT - 1) * 256 + el];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int tile = t0 + u;
				if (tile >= NT) break;
				int I = 0;
				while ((I + 1) * T - (I + 1) * I / 2 <= tile) ++I; // first tile of block row I+1 is past `tile`
				const int J = I + (tile - (I * T - I * (I - 1) / 2));
				const int jj = 16 * I + tr, ii = 16 * J + tc; // jj <= ii except inside diagonal tiles
				if (jj > ii) continue;
				double v = 0.0;
				if (ii < p) {
					v = v4[u];
					if (icpt) v -= l.sv[ii] * l.sv[jj] * inv_sw;
					if (ii == jj) v += lam;
				}
				A[(size_t)ii * LD + jj] = v;
			}
		}
	}
	for (int j = tid; j < P16; j += NTHR) { // y row: centred Sxy
		double v = 0.0;
		if (j < p) {
			const double q = vec[1 * P16 + j];
			v = icpt ? q - l.sv[j] * sy / sw : q;
		}
		A[(size_t)P16 * LD + j] = v;
	}
}

// OCC = workgroups per CU the register budget is cut for: 2 where the LDS matrix leaves room for two (p <= 88; the
// spills that costs are cheaper than an idle half of the CU: solve -26 % at p = 64), 1 for the widest designs, whose
// 134 KB matrix fills the CU's LDS anyway (there the tighter budget only adds spills: +22..47 %).
template <int MODE, int OCC, int NTHR>
__global__ __launch_bounds__(NTHR, OCC) void solve_wide_kernel(WideArgs args) {
	extern __shared__ double sm[];
	const int p = args.p;
	const int tid = threadIdx.x;
	const bool icpt = args.fit_intercept != 0;
	const int model = args.model;
	const WideLds l = carve_lds(sm, p);
	const int T = l.T, P16 = l.P16, LD = l.LD;
	const int NT = T * (T + 1) / 2;
	double *A = l.A;

	const int n_items = (MODE == MODE_PRIMARY) ? (int)args.n_groups : *args.refine_count;
	for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
		const int64_t gl = (MODE == MODE_PRIMARY) ? (int64_t)item : (int64_t)args.refine_list[item]; // group within this launch
		const int64_t g = args.group_base + gl;                                                       // global group
		const double *rec = args.moments + gl * (int64_t)wide_record_len(T);
		const double *vec = rec + (int64_t)NT * 256;
		const double *sc = vec + 4 * P16;
		double *core = args.core + g * (int64_t)(p + 6);
		double *inf = (args.inference && args.compute_inference) ? args.inference + g * (int64_t)(5 * p + 2) : nullptr;
		const double *rvec = args.refine_vec + g * (int64_t)refine_vec_len(p); // {sum w r^2, sum w r, X'Wr, centred yy}
		const int64_t nrows = args.rule_counts ? args.rule_counts[g] : args.row_offsets[g + 1] - args.row_offsets[g];

		// a record whose fit failed (or has no inference block): everything NaN, status in the last slot
		auto write_null = [&](int status, bool core_too) {
			if (core_too)
				for (int k = tid; k < p + 6; k += NTHR) core[k] = (k == p + 5) ? (double)status : nan64w();
			if (inf)
				for (int k = tid; k < 5 * p + 2; k += NTHR) inf[k] = nan64w();
		};

		__syncthreads(); // the previous item's LDS contents are dead
		SOLVE_STAMP(0);
		const double sy = sc[0], syy = sc[1], sw = sc[2], cnt = sc[3], first_y = sc[4];
		int status = ANOFOX_ERROR_SUCCESS;
		if (nrows < 2) status = ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS;                                       // ols_aggregate.cpp:263-267
		else if (model == ANOFOX_HIP_MODEL_RIDGE && args.alpha < 0.0) status = ANOFOX_ERROR_INVALID_ALPHA;  // ridge.rs:38-40
		else if (!(cnt > 0.0)) status = ANOFOX_ERROR_NO_VALID_DATA;                                         // ols.rs:68-70
		if (status != ANOFOX_ERROR_SUCCESS) {
			write_null(status, true);
			continue;
		}

		int mine = 0;
		for (int j = tid; j < P16; j += NTHR) {
			const int a = (j < p && vec[3 * P16 + j] != 0.0) ? 1 : 0;
			l.active[j] = a;
			l.sv[j] = j < p ? vec[0 * P16 + j] : 0.0;
			l.fx[j] = j < p ? vec[2 * P16 + j] : 0.0;
			mine += a;
		}
		const int peff = (int)block_sum<NTHR>((double)mine, l.red + 12, tid);

		const double cyy_c = syy - sy * sy / sw;
		const double ymean = (icpt ? first_y : 0.0) + sy / sw;
		if (peff == 0) { // ols.rs:101-130, wls.rs:119-150
			if (!icpt) {
				write_null(ANOFOX_ERROR_INSUFFICIENT_DATA, true);
			} else {
				write_null(0, false); // inference: None
				for (int k = tid; k < p + 6; k += NTHR) {
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

		// ridge penalty: `lam` goes into the primary factor; the refinement modes factor with and aim at `lam_rows`, glmnet's lambda with sd_y
		// re-summed over the rows about the mean (uncentred moments of a nearly constant y cancel; such groups are queued)
		double lam = 0.0, lam_rows = 0.0;
		bool glmnet_cancels = false;
		if (model == ANOFOX_HIP_MODEL_RIDGE) {
			lam = lam_rows = args.alpha;
			if (args.lambda_scaling == ANOFOX_LAMBDA_SCALING_GLMNET) {
				lam = lam_rows = cnt * args.alpha / sqrt(cyy_c / cnt);
				glmnet_cancels = !icpt && !(cyy_c * kGlmnetCancelRatio > syy);
				if (MODE != MODE_PRIMARY && !icpt) lam_rows = cnt * args.alpha / sqrt(rvec[p + 2] / cnt);
			}
		}
		const double tss = icpt ? cyy_c : syy;

		// (the refinement modes factor the matrix with the re-summed lambda: the update is then an exact Newton step and the
		// standard errors come from the matrix the coefficients solve)
		load_moment_matrix<NTHR>(l, rec, p, icpt, MODE == MODE_PRIMARY ? lam : lam_rows, sw, sy, tid);
		__syncthreads();
		for (int j = tid; j < P16; j += NTHR) l.diag0[j] = j < p ? A[(size_t)j * LD + j] : 1.0;
		// blocked_cholesky starts with a barrier
		SOLVE_STAMP(1);
		const double min_ratio = blocked_cholesky<NTHR>(l, tid);
		SOLVE_STAMP(2);
		// W = L^-1 is needed for the standard errors (diag of the inverse) and for the refinement passes; the
		// plain primary solve only needs beta = L^-T z
		const bool need_w = (MODE != MODE_PRIMARY) || (inf != nullptr);
		for (int j = tid; j < P16; j += NTHR) l.zv[j] = A[(size_t)P16 * LD + j]; // z = y row
		__syncthreads();
		if (!need_w) {
			blocked_back_solve<NTHR>(l, tid);
			SOLVE_STAMP(3);
		} else {
			blocked_tri_inverse<NTHR>(l, tid);
			SOLVE_STAMP(3);
			if (MODE == MODE_UPDATE) {
				// gradient of the (penalised) objective at the record's coefficients, centred: zv := W gc
				const double gs = rvec[1];
				for (int j = tid; j < P16; j += NTHR) {
					double gj = 0.0;
					if (j < p && l.live[j]) {
						gj = rvec[2 + j];
						if (icpt) gj -= (l.sv[j] / sw) * gs;
						gj -= lam_rows * core[j];
					}
					l.bv[j] = gj;
				}
				__syncthreads();
				for (int i = tid; i < P16; i += NTHR) {
					double u = 0.0;
					for (int j = 0; j <= i; ++j) u = fma(getW(l, i, j), l.bv[j], u);
					l.zv[i] = u;
				}
				__syncthreads();
			}
			// bv_j = sum_{i >= j} W[i][j] zv_i  (= beta, or the refinement step delta);  diag_j = sum_i W[i][j]^2
			for (int j = tid; j < P16; j += NTHR) {
				const double wjj = l.linv[j];
				double bj = wjj * l.zv[j], dj = wjj * wjj;
				for (int i = j + 1; i < P16; ++i) {
					const double wv = A[(size_t)j * LD + i];
					bj = fma(wv, l.zv[i], bj);
					dj = fma(wv, wv, dj);
				}
				if (MODE == MODE_PRIMARY) l.bv[j] = bj;
				else if (MODE == MODE_UPDATE) l.bv[j] = (j < p && l.live[j] ? core[j] : 0.0) + bj;
				else l.bv[j] = (j < p && l.live[j]) ? core[j] : 0.0;
				l.diag0[j] = dj; // diag0 is dead after the factorisation: reuse for diag((LL')^-1)
			}
			__syncthreads();
		}
		SOLVE_STAMP(4);
		// block sums: rank, b'c, b'b, mean correction of the intercept, |z|^2
		double rk = 0.0, bc = 0.0, bb = 0.0, xb = 0.0, zz = 0.0;
		for (int j = tid; j < p; j += NTHR) {
			if (l.live[j]) {
				rk += 1.0;
				const double q = vec[1 * P16 + j];
				const double cj = icpt ? q - l.sv[j] * sy / sw : q;
				const double bj = l.bv[j];
				bc += bj * cj;
				bb += bj * bj;
				xb += bj * ((icpt ? l.fx[j] : 0.0) + l.sv[j] / sw);
				const double zj = A[(size_t)P16 * LD + j];
				zz += zj * zj;
			}
		}
		const double rk_t = block_sum<NTHR>(rk, l.red + 12, tid);
		const double bc_t = block_sum<NTHR>(bc, l.red + 12, tid);
		const double bb_t = block_sum<NTHR>(bb, l.red + 12, tid);
		const double xb_t = block_sum<NTHR>(xb, l.red + 12, tid);
		const double zz_t = block_sum<NTHR>(zz, l.red + 12, tid);
		const int rank = (int)rk_t;
		if (MODE == MODE_UPDATE) { // only the coefficients change in this pass
			for (int k = tid; k <= p; k += NTHR) {
				if (k < p) core[k] = l.live[k] ? l.bv[k] : nan64w();
				else core[k] = icpt ? ymean - xb_t : nan64w();
			}
			continue;
		}
		double rss;
		if (MODE == MODE_FINAL) rss = rvec[0];
		else if (model == ANOFOX_HIP_MODEL_RIDGE) rss = tss - bc_t - lam * bb_t;
		else rss = tss - zz_t; // Syy - |L^-1 Sxy|^2
		const int n_par = rank + (icpt ? 1 : 0);
		const double df = cnt - (double)n_par;
		// (nearly square designs as well: see solve_tiles_impl.h)
		const bool refine = (MODE == MODE_PRIMARY) && (!(rss > kRefineTolW * tss) || min_ratio < kPivotWarnW || df < 0.25 * (double)rank || glmnet_cancels);
		const double dfm = (double)rank;
		const double r2 = 1.0 - rss / tss;
		const double fstat = ((tss - rss) / dfm) / (rss / df);

		for (int k = tid; k < p + 6; k += NTHR) {
			double v;
			if (k < p) v = l.live[k] ? l.bv[k] : nan64w();
			else if (k == p) v = icpt ? ymean - xb_t : nan64w();
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
		SOLVE_STAMP(5);

		if (inf) {
			// standard errors and the F statistic; t, p, the interval and the F p-value are filled in by
			// inference_wide_finish_kernel (the special functions' ~240-VGPR call tree stays out of this kernel, which
			// then fits 2-4 workgroups per CU instead of one).  df travels in the F p-value's slot.
			const double sigma2 = rss / df;
			for (int j = tid; j < p; j += NTHR) inf[j] = l.live[j] ? sqrt(sigma2 * l.diag0[j]) : nan64w();
			if (tid == 0) {
				inf[5 * p] = fstat;
				inf[5 * p + 1] = df;
			}
		}
		SOLVE_STAMP(6);
	}
}

// t, p, interval and F p-value of the classical inference from the standard errors solve_wide_kernel left behind:
// one 128-thread workgroup per group (lib.rs:188-254 fills the same five arrays from anofox-regression's result).
__global__ __launch_bounds__(128) void inference_wide_finish_kernel(WideArgs args) {
	__shared__ double sh[4];
	__shared__ int cnt_sh[2];
	const int p = args.p;
	const int tid = threadIdx.x;
	const int64_t g = args.group_base + blockIdx.x;
	double *inf = args.inference + g * (int64_t)(5 * p + 2);
	const double *core = args.core + g * (int64_t)(p + 6);
	const double df = inf[5 * p + 1]; // left there by the solve; NaN = no inference for this group
	if (isnan(df)) return;            // (uniform over the workgroup)
	int mine = 0;
	for (int j = tid; j < p; j += 128) mine += isnan(core[j]) ? 0 : 1;
	for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m, 64);
	if ((tid & 63) == 0) cnt_sh[tid >> 6] = mine;
	if (tid == 0) sh[0] = dm_tcrit_cached(static_cast<TcritSlot *>(args.tcrit_table), 0.5 * (1.0 + args.confidence_level), df);
	__syncthreads();
	const double tcrit = sh[0];
	const double dfm = (double)(cnt_sh[0] + cnt_sh[1]);
	const double fstat = inf[5 * p];
	for (int j = tid; j < p; j += 128) {
		const double b = core[j], se = inf[j];
		double tval = nan64w(), pval = nan64w(), lo = nan64w(), hi = nan64w();
		if (!isnan(b) && !isnan(se)) {
			tval = b / se;
			pval = dm_t_two_sided_p(tval, df);
			lo = b - tcrit * se;
			hi = b + tcrit * se;
		}
		inf[p + j] = tval;
		inf[2 * p + j] = pval;
		inf[3 * p + j] = lo;
		inf[4 * p + j] = hi;
	}
	if (tid == 0) inf[5 * p + 1] = dm_f_sf(fstat, dfm, df); // every thread read df before the barrier above
}

// (error-free transformations: dd_arith.h; no FMA contraction from here to the end of the residual kernel)
#pragma clang fp contract(off)

// One workgroup per queued group, straight from the data with the record's current coefficients:
//   refine_vec[g] = { sum w r^2, sum w r, sum w r (x_j - shift_j) ..., sum w (y - ybar)^2 },  r = y - b0 - x'b  over the valid rows.
// The residual and the two gradient sums are formed in double-double arithmetic (compensated dot products): with the
// residual in working precision the refinement stalls at cond(X) * 1e-13 — 1e-9 .. 5e-9 on nearly square designs of
// 40 .. 128 columns (cond 1e3 .. 2e4), measured in round 2 — while an extended-precision residual takes the same
// update to cond * eps in one step (the oracle refines the same way, in long double).  Only queued groups pay for it.
__global__ __launch_bounds__(256) void residual_grad_wide_kernel(WideArgs args) {
	__shared__ double bsh[kWideMaxP];
	__shared__ double weh[256], wel[256];
	__shared__ double part[16];
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
		const double *sc = vec + 4 * P16;                                                 // {sy, syy, sw, cnt, first_y}
		const double ybar = sc[0] / sc[2] + (args.fit_intercept ? sc[4] : 0.0);           // mean of y over the valid rows
		double cyy = 0.0;
		const double *mycol = tid < p ? args.x_table[tid] : nullptr;
		const int64_t lo = args.row_offsets[g], hi = group_row_end(args, g);
		double rss = 0.0, gs_h = 0.0, gs_l = 0.0, gj_h = 0.0, gj_l = 0.0;
		for (int64_t r0 = lo; r0 < hi; r0 += 256) {
			const int64_t r = r0 + tid;
			double wh = 0.0, wl = 0.0;
			if (r < hi) {
				const double yv = args.y[r];
				bool ok = isfinite(yv);
				double fh = b0, fl = 0.0; // fit = fh + fl
				for (int j = 0; j < p; ++j) {
					const double xv = args.x_table[j][r];
					ok = ok && isfinite(xv);
					dd_fit_term(fh, fl, bsh[j], xv);
				}
				double wv = 1.0;
				if (weighted) {
					wv = args.w[r];
					ok = ok && (wv > 0.0) && isfinite(wv);
				}
				if (ok) {
					double e;
					dd_weighted_residual(yv, fh, fl, wv, e, wh, wl); // wh + wl = w (y - fit) to twice the working precision
					rss = fma(wh, e, rss);
					dd_add(gs_h, gs_l, wh, wl);
					const double dy = yv - ybar;
					cyy = fma(wv * dy, dy, cyy);
				}
			}
			weh[tid] = wh;
			wel[tid] = wl;
			__syncthreads();
			if (mycol) {
				const int64_t m = (hi - r0 < 256) ? hi - r0 : 256;
				for (int64_t t = 0; t < m; ++t) {
					const double wt = weh[t];
					if (wt != 0.0) // invalid rows carry 0 and are skipped
						dd_add_scaled_diff(gj_h, gj_l, wt, wel[t], mycol[r0 + t], shift);
				}
			}
			__syncthreads();
		}
		for (int m = 32; m >= 1; m >>= 1) {
			rss += __shfl_xor(rss, m, 64);
			cyy += __shfl_xor(cyy, m, 64);
			dd_add(gs_h, gs_l, __shfl_xor(gs_h, m, 64), __shfl_xor(gs_l, m, 64));
		}
		if ((tid & 63) == 0) {
			part[12 + (tid >> 6)] = cyy;
			part[tid >> 6] = rss;
			part[4 + (tid >> 6)] = gs_h;
			part[8 + (tid >> 6)] = gs_l;
		}
		__syncthreads();
		double *out = args.refine_vec + g * (int64_t)refine_vec_len(p);
		if (tid == 0) {
			out[0] = part[0] + part[1] + part[2] + part[3];
			out[p + 2] = part[12] + part[13] + part[14] + part[15];
			double h = part[4], l = part[8];
			for (int w2 = 1; w2 < 4; ++w2) dd_add(h, l, part[4 + w2], part[8 + w2]);
			out[1] = h + l;
		}
		if (tid < p) out[2 + tid] = gj_h + gj_l;
	}
}

#pragma clang fp contract(fast)

size_t solve_wide_lds_bytes(int p) {
	const int T = wide_tiles(p), P16 = 16 * T, LD = P16 + 1;
	const size_t dbl = (size_t)(P16 + 1) * LD + 7 * (size_t)P16 + 16;
	return dbl * sizeof(double) + 2 * (size_t)P16 * sizeof(int);
}

// Heteroscedasticity-consistent standard errors for wide designs (same estimator and references as
// hc_narrow.hip).  One workgroup per group: the centred moment matrix is factored again exactly as in the solve
// (same inputs, same code => same active set), W = L^-1 is expanded to the full symmetric S^-1 = W'W in LDS, and
// the rows stream through in chunks of 16, one chunk per wavefront at a time (hc_rows): U = C S^-1 (16 x p) on
// the FP64 matrix cores — v_mfma_f64_16x16x4_f64, A fragment = 16 rows x 4 features of the centred chunk
// (loaded from global memory in fragment layout, one chunk ahead), B fragment = 4 x 16 block of S^-1 (one
// ds_read_b64 per lane) — then h = w (1/sum(w) + c'u), V_jj += omega u_j^2 from the accumulator registers.
// No barrier inside the row pass.  It costs n p^2 FMAs per group, twice the flops of the accumulate kernel.
typedef const double __attribute__((address_space(1))) *hc_gptr_t;

typedef double hc_dbl4 __attribute__((ext_vector_type(4)));

// Row pass of hc_wide_kernel for one wavefront: chunks of 16 rows, chunk c of the group goes to wave c mod 4.
// Everything of a chunk stays inside the wave (no barriers): the A fragments of U = C S^-1 are loaded straight
// from global memory in fragment layout (lane (kk, lj): row lj, feature 4 ks + kk), one chunk ahead; S^-1 comes
// from LDS; the same rows are read a second time in accumulator layout (row kk + 4 i, feature 16 J + lj, L1 / L2
// hits) for the row sums c'u and b'c, which are then 16-lane reductions.  vacc[J] = this lane's share of V for
// column 16 J + lj.
template <int TT>
__device__ __forceinline__ void hc_rows(const WideArgs &args, const WideLds &l, const unsigned long long *colptr, int64_t lo,
                                        int64_t hi, int wave, int kk, int lj, int p, bool weighted, int hc, double ycen, double h0,
                                        double hc1, double (&vacc)[kWideMaxP / 16]) {
	constexpr int KS = 4 * TT; // k-steps of 4 features
	const int LD = l.LD;
	const double *A = l.A;
	const hc_gptr_t yp = (hc_gptr_t)(uintptr_t)args.y;
	const hc_gptr_t wp = (hc_gptr_t)(uintptr_t)args.w;
	double bD[TT], xbD[TT];
	hc_gptr_t colD[TT];
#pragma unroll
	for (int J = 0; J < TT; ++J) {
		bD[J] = l.zv[16 * J + lj];
		xbD[J] = l.bv[16 * J + lj];
		colD[J] = (hc_gptr_t)colptr[16 * J + lj];
	}
	const int64_t first = lo + 16 * (int64_t)wave;
	for (int64_t R0 = first; R0 < hi; R0 += 64) {
		const bool a_in = R0 + lj < hi;
		const int64_t ra = a_in ? R0 + lj : hi - 1; // clamped: loads stay unconditional
		// the rows in accumulator layout, plus y and w of rows kk + 4 i (consumed after the products)
		double cD[TT][4], yv[4], wv[4];
		bool rin[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			rin[i] = R0 + kk + 4 * i < hi;
			const int64_t r = rin[i] ? R0 + kk + 4 * i : hi - 1;
#pragma unroll
			for (int J = 0; J < TT; ++J) cD[J][i] = colD[J][r];
			yv[i] = yp[r];
			wv[i] = weighted ? wp[r] : 1.0;
		}
		hc_dbl4 U[TT];
#pragma unroll
		for (int J = 0; J < TT; ++J) U[J] = hc_dbl4{0.0, 0.0, 0.0, 0.0};
		// A fragments four k-steps ahead of their use (a rolled loop keeps the live set small)
		double an[4];
#pragma unroll
		for (int q = 0; q < 4; ++q) an[q] = ((hc_gptr_t)colptr[4 * q + kk])[ra];
#pragma unroll 1
		for (int k4 = 0; k4 < KS; k4 += 4) {
			double ac[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) { // centred, padding features zero, rows past the end marked invalid (NaN)
				const int j = 4 * (k4 + q) + kk;
				const double v = j < p ? an[q] - l.bv[j] : 0.0;
				ac[q] = a_in ? v : nan64w();
			}
			if (k4 + 4 < KS) {
#pragma unroll
				for (int q = 0; q < 4; ++q) an[q] = ((hc_gptr_t)colptr[4 * (k4 + 4 + q) + kk])[ra];
			}
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const double *srow = A + (size_t)(4 * (k4 + q) + kk) * LD + lj;
#pragma unroll
				for (int J = 0; J < TT; ++J) U[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[q], srow[16 * J], U[J], 0, 0, 0);
			}
		}
		double hp[4] = {0.0, 0.0, 0.0, 0.0}, ep[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
		for (int J = 0; J < TT; ++J) {
			const bool real = 16 * J + lj < p;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const double c = real ? cD[J][i] - xbD[J] : 0.0;
				hp[i] = fma(c, U[J][i], hp[i]);
				ep[i] = fma(bD[J], c, ep[i]);
			}
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			for (int m = 8; m >= 1; m >>= 1) { // the 16 lanes that share the rows kk + 4 i
				hp[i] += __shfl_xor(hp[i], m, 64);
				ep[i] += __shfl_xor(ep[i], m, 64);
			}
			const double e = yv[i] - ycen - ep[i];
			const double lev = wv[i] * (h0 + hp[i]);
			double om = wv[i] * wv[i] * e * e; // w_i e_i^2 of the scaled residual, times the w_i of the scaled row
			if (hc == ANOFOX_HC_HC1) om *= hc1;
			else if (hc == ANOFOX_HC_HC2) om /= (1.0 - lev);
			else if (hc == ANOFOX_HC_HC3) om /= (1.0 - lev) * (1.0 - lev);
			// rows outside the fit (non-finite x / y, w <= 0) carry NaN / inf through e or lev
			const bool ok = rin[i] && isfinite(e) && isfinite(lev) && (!weighted || ((wv[i] > 0.0) && isfinite(wv[i])));
			om = ok ? om : 0.0;
#pragma unroll
			for (int J = 0; J < TT; ++J) {
				const double uu = ok ? U[J][i] : 0.0;
				vacc[J] = fma(om, uu * uu, vacc[J]);
			}
		}
	}
}

__global__ __launch_bounds__(256) void hc_wide_kernel(WideArgs args) {
	extern __shared__ double sm[];
	const int p = args.p;
	const int tid = threadIdx.x;
	const bool icpt = args.fit_intercept != 0;
	const bool weighted = args.model == ANOFOX_HIP_MODEL_WLS;
	const int hc = args.hc_type;
	const WideLds l = carve_lds(sm, p);
	const int T = l.T, P16 = l.P16, LD = l.LD;
	const int NT = T * (T + 1) / 2;
	double *A = l.A;
	unsigned long long *colptr = reinterpret_cast<unsigned long long *>(l.live + P16); // [P16] feature column addresses
	double *vsum = reinterpret_cast<double *>(colptr + P16);                             // [4][P16] per-wave V

	const int lane = tid & 63, wave = tid >> 6;
	const int kk = lane >> 4, lj = lane & 15; // MFMA fragment coordinates of this lane
	for (int j = tid; j < P16; j += 256) colptr[j] = (unsigned long long)(uintptr_t)args.x_table[j < p ? j : p - 1];

	for (int64_t gl = blockIdx.x; gl < args.n_groups; gl += gridDim.x) {
		const int64_t g = args.group_base + gl;
		const double *rec = args.moments + gl * (int64_t)wide_record_len(T);
		const double *vec = rec + (int64_t)NT * 256;
		const double *sc = vec + 4 * P16;
		const double *core = args.core + g * (int64_t)(p + 6);
		double *inf = args.inference + g * (int64_t)(5 * p + 2);
		__syncthreads(); // the previous group's LDS contents are dead
		if (tid == 0) args.hc_df[gl] = nan64w(); // overwritten below when HC errors are produced
		if (core[p + 5] != 0.0) continue; // NULL group: the inference record is already NaN
		const double sy = sc[0], sw = sc[2], cnt = sc[3];
		double mine = 0.0;
		for (int j = tid; j < P16; j += 256) {
			l.active[j] = (j < p && vec[3 * P16 + j] != 0.0) ? 1 : 0;
			l.sv[j] = j < p ? vec[0 * P16 + j] : 0.0;
			l.fx[j] = j < p ? vec[2 * P16 + j] : 0.0;
			mine += (j < p && !isnan(core[j])) ? 1.0 : 0.0;
		}
		const int rank = (int)block_sum<256>(mine, l.red + 12, tid);
		if (rank == 0) continue; // intercept-only fit: inference is None (ols.rs:101-130)

		load_moment_matrix<256>(l, rec, p, icpt, 0.0, sw, sy, tid);
		__syncthreads();
		for (int j = tid; j < P16; j += 256) l.diag0[j] = j < p ? A[(size_t)j * LD + j] : 1.0;
		(void)blocked_cholesky<256>(l, tid);
		blocked_tri_inverse<256>(l, tid);
		// S^-1 = W'W: entry (i, j), i >= j, = sum_{k >= i} W[k][i] W[k][j]; W[k][i] (k > i) sits at A[i][k], so
		// these are dot products of row tails of the upper triangle; the results go to the (dead) lower triangle
		for (int idx = tid; idx < P16 * (P16 + 1) / 2; idx += 256) {
			int i, j;
			tri_decode(idx, i, j);
			double acc = (i == j) ? l.linv[i] * l.linv[i] : l.linv[i] * A[(size_t)j * LD + i];
			for (int k = i + 1; k < P16; ++k) acc = fma(A[(size_t)i * LD + k], A[(size_t)j * LD + k], acc);
			A[(size_t)i * LD + j] = acc;
		}
		__syncthreads();
		for (int idx = tid; idx < P16 * P16; idx += 256) { // mirror into the upper triangle
			const int i = idx / P16, j = idx - i * P16;
			if (i > j) A[(size_t)j * LD + i] = A[(size_t)i * LD + j];
		}
		// coefficients (0 at dropped columns) and column means
		double yc = 0.0;
		for (int j = tid; j < P16; j += 256) {
			const double bj = (j < p && l.live[j]) ? core[j] : 0.0;
			const double xb = (icpt && j < p) ? l.fx[j] + l.sv[j] / sw : 0.0;
			l.zv[j] = bj;
			l.bv[j] = xb;
			yc = fma(bj, xb, yc);
		}
		const double ycen = (icpt ? core[p] : 0.0) + block_sum<256>(yc, l.red + 12, tid); // fitted value at x = xbar
		const double df = cnt - (double)(rank + (icpt ? 1 : 0));
		const double hc1 = cnt / df;
		const double h0 = icpt ? 1.0 / sw : 0.0;

		const int64_t lo = args.row_offsets[g], hi = args.row_offsets[g + 1];
		__syncthreads(); // bv / zv / S^-1 / colptr visible
		double vacc[kWideMaxP / 16];
#pragma unroll
		for (int J = 0; J < kWideMaxP / 16; ++J) vacc[J] = 0.0;
		switch (T) {
		case 1: hc_rows<1>(args, l, colptr, lo, hi, wave, kk, lj, p, weighted, hc, ycen, h0, hc1, vacc); break;
		case 2: hc_rows<2>(args, l, colptr, lo, hi, wave, kk, lj, p, weighted, hc, ycen, h0, hc1, vacc); break;
		case 3: hc_rows<3>(args, l, colptr, lo, hi, wave, kk, lj, p, weighted, hc, ycen, h0, hc1, vacc); break;
		case 4: hc_rows<4>(args, l, colptr, lo, hi, wave, kk, lj, p, weighted, hc, ycen, h0, hc1, vacc); break;
		case 5: hc_rows<5>(args, l, colptr, lo, hi, wave, kk, lj, p, weighted, hc, ycen, h0, hc1, vacc); break;
		case 6: hc_rows<6>(args, l, colptr, lo, hi, wave, kk, lj, p, weighted, hc, ycen, h0, hc1, vacc); break;
		case 7: hc_rows<7>(args, l, colptr, lo, hi, wave, kk, lj, p, weighted, hc, ycen, h0, hc1, vacc); break;
		default: hc_rows<8>(args, l, colptr, lo, hi, wave, kk, lj, p, weighted, hc, ycen, h0, hc1, vacc); break;
		}
#pragma unroll
		for (int J = 0; J < kWideMaxP / 16; ++J) {
			vacc[J] += __shfl_xor(vacc[J], 16, 64);
			vacc[J] += __shfl_xor(vacc[J], 32, 64); // the four row groups of a wavefront
			if (J < T && lane < 16) vsum[wave * P16 + 16 * J + lj] = vacc[J];
		}
		__syncthreads();
		// standard errors only; t, p and the interval follow in hc_wide_finish_kernel (the special functions are
		// out-of-line calls and would cap this kernel's register budget)
		for (int j = tid; j < p; j += 256) {
			if (!l.live[j]) continue;
			inf[j] = sqrt(vsum[j] + vsum[P16 + j] + vsum[2 * P16 + j] + vsum[3 * P16 + j]);
		}
		if (tid == 0) args.hc_df[gl] = df;
	}
}

// one thread per (group, coefficient): t, p-value and interval from the HC standard error
__global__ __launch_bounds__(256) void hc_wide_finish_kernel(WideArgs args) {
	const int p = args.p;
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= args.n_groups * (int64_t)p) return;
	const int64_t gl = e / p;
	const int j = (int)(e - gl * p);
	const double df = args.hc_df[gl];
	if (isnan(df)) return; // no HC errors were computed for this group
	const int64_t g = args.group_base + gl;
	double *inf = args.inference + g * (int64_t)(5 * p + 2);
	const double b = args.core[g * (int64_t)(p + 6) + j];
	if (isnan(b)) return; // dropped / aliased column
	const double se = inf[j];
	const double tcrit = dm_tcrit_cached(static_cast<TcritSlot *>(args.tcrit_table), 0.5 * (1.0 + args.confidence_level), df);
	const double tval = b / se;
	inf[p + j] = tval;
	inf[2 * p + j] = dm_t_two_sided_p(tval, df);
	inf[3 * p + j] = b - tcrit * se;
	inf[4 * p + j] = b + tcrit * se;
}

size_t hc_wide_lds_bytes(int p) {
	const int P16 = 16 * wide_tiles(p);
	return solve_wide_lds_bytes(p) + (size_t)(5 * P16) * sizeof(double);
}

} // namespace

hipError_t launch_inference_wide_finish(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0 || !a.inference || !a.compute_inference) return hipSuccess;
	hipLaunchKernelGGL(inference_wide_finish_kernel, dim3((unsigned)a.n_groups), dim3(128), 0, stream, a);
	return hipGetLastError();
}

template <int OCC, int NTHR>
hipError_t launch_solve_wide_occ(const WideArgs &a, int mode, size_t lds, hipStream_t stream) {
	static std::once_flag attr_once; // contexts of several host threads launch concurrently
	std::call_once(attr_once, [] {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_wide_kernel<MODE_PRIMARY, OCC, NTHR>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_wide_kernel<MODE_UPDATE, OCC, NTHR>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_wide_kernel<MODE_FINAL, OCC, NTHR>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	});
	const unsigned few = 256 * (256 / NTHR); // grid of the refinement modes (they loop over the queue)
	if (mode == MODE_PRIMARY) {
		const unsigned grid = (unsigned)(a.n_groups < 65535 * 16 ? a.n_groups : 65535 * 16);
		hipLaunchKernelGGL((solve_wide_kernel<MODE_PRIMARY, OCC, NTHR>), dim3(grid), dim3(NTHR), lds, stream, a);
	} else if (mode == MODE_UPDATE) {
		hipLaunchKernelGGL((solve_wide_kernel<MODE_UPDATE, OCC, NTHR>), dim3(few), dim3(NTHR), lds, stream, a);
	} else {
		hipLaunchKernelGGL((solve_wide_kernel<MODE_FINAL, OCC, NTHR>), dim3(few), dim3(NTHR), lds, stream, a);
	}
	return hipGetLastError();
}

hipError_t launch_solve_wide(const WideArgs &a, int mode, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	// the primary solve of 32 < p <= 128 runs with one wavefront per group and the matrix in registers (solve_tiles.hip);
	// the refinement modes — a handful of queued groups — stay here.  ANOFOX_SOLVE_TILES=0: this file's kernels only.
	static const bool tiles_on = !(getenv("ANOFOX_SOLVE_TILES") && atoi(getenv("ANOFOX_SOLVE_TILES")) == 0);
	if (tiles_on && mode == MODE_PRIMARY && solve_tiles_supports(a.p)) return launch_solve_tiles(a, stream);
	const size_t lds = solve_wide_lds_bytes(a.p);
	// p <= 96 (two or more groups' matrices fit a CU's LDS): one wave per group.  50 000 x 1000 rows, solve without /
	// with inference: p = 33 5.78 / 9.94 -> 2.18 / 3.30 ms, p = 64 7.20 / 12.4 -> 2.92 / 4.56 ms; 20 000 groups: p = 80
	// 3.79 / 6.73 -> 3.32 / 5.03 ms, p = 96 4.41 / 7.65 -> 4.14 / 6.27 ms; slower from p = 97 (one group per CU:
	// p = 112 6.24 / 8.63 -> 9.87 / 15.1 ms; two waves per group there: 7.34 / 10.4 ms).  With the register budget
	// cut for two waves per SIMD (ANOFOX_SOLVE_WAVE=2) the spills cost more than the occupancy gains (p = 33:
	// 2.47 / 4.67 ms).  ANOFOX_SOLVE_WAVE=0: four waves per group everywhere.
	static const int wave_mode = getenv("ANOFOX_SOLVE_WAVE") ? atoi(getenv("ANOFOX_SOLVE_WAVE")) : 1;
	static const int wave_max_t = getenv("ANOFOX_SOLVE_WAVE_T") ? atoi(getenv("ANOFOX_SOLVE_WAVE_T")) : 6;
	if (wave_mode && wide_tiles(a.p) <= wave_max_t)
		return wave_mode == 2 ? launch_solve_wide_occ<2, 64>(a, mode, lds, stream) : launch_solve_wide_occ<1, 64>(a, mode, lds, stream);
	return 2 * lds <= (size_t)160 * 1024 ? launch_solve_wide_occ<2, 256>(a, mode, lds, stream) : launch_solve_wide_occ<1, 256>(a, mode, lds, stream);
}

#ifdef ANOFOX_SOLVE_STAMPS
extern "C" __attribute__((visibility("default"))) int anofox_hip_diag_solve_stamps(unsigned long long *out32) {
	return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_solve_stamps), 32 * sizeof(unsigned long long));
}
#endif

hipError_t launch_hc_wide(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0 || !a.inference) return hipSuccess;
	if (!a.hc_df) return hipErrorInvalidValue;
	const size_t lds = hc_wide_lds_bytes(a.p);
	if (lds > 160 * 1024) return hipErrorInvalidValue;
	static std::once_flag attr_once;
	std::call_once(attr_once, [] {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hc_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	});
	const unsigned grid = (unsigned)(a.n_groups < 65535 * 16 ? a.n_groups : 65535 * 16);
	hipLaunchKernelGGL(hc_wide_kernel, dim3(grid), dim3(256), lds, stream, a);
	const int64_t elems = a.n_groups * (int64_t)a.p;
	hipLaunchKernelGGL(hc_wide_finish_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_residual_grad_wide(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(residual_grad_wide_kernel, dim3(512), dim3(256), 0, stream, a);
	return hipGetLastError();
}

} // namespace anofox
