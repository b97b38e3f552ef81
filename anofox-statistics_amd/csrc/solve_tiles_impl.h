// solve_tiles_impl.h — primary solve (and diagnostics) of one group by ONE wavefront for 32 < p <= 128, with the
// whole moment matrix held in registers as 16 x 16 tiles in the accumulator layout of v_mfma_f64_16x16x4_f64.
//
// Same contract as solve_wide.hip's MODE_PRIMARY (the reference's pre-checks and shortcuts of
// crates/anofox-stats-core/src/models/ols.rs:68-139, ridge.rs:38-40, wls.rs:119-157; the regressor's closed forms,
// SURVEY.md Appendix B.7; NaN re-expansion ols.rs:167-171,191-206) and the same record in / records out, so the
// refinement passes, the HC kernel and inference_wide_finish_kernel of solve_wide.hip follow it unchanged.
//
// Why: the workgroup-per-group solve keeps the (p + 1)^2 matrix in LDS — 134 KB at p = 128, ONE group in flight per
// CU, 281 k cycles per group of which 107 k are single-wave 16 x 16 diagonal blocks — and was 21.5 of the 97 ms of a
// 50 000 x 4096 x 128 step.  Here a group needs no LDS for its matrix and no barrier, so every SIMD of a CU works on a
// group of its own.
//
// Layout.  Lane (q, n) = lane 16 q + n holds, in register r of a tile, element [q + 4 r][n] ("C layout": what the
// matrix core leaves in its accumulators, and what accumulate_wide writes to the record, so a tile loads as four
// coalesced 512-byte rows).  Two facts make that layout sufficient for everything:
//   * the B operand of k-step s of a product N M is register s of C-layout(M)      (B[k][n], k = q + 4 s)
//   * the A operand of k-step s of a product N M is register s of C-layout(N')     (A[m][k] = N'[k][m])
// so with the UPPER triangle stored (tile (i, j), i <= j, U = L'):
//   trailing update   S_ij -= U_ki' U_kj        A = -U_ki, B = U_kj         registers as they are
//   panel             U_kj  = X S_kj            A = C-layout(X'), B = S_kj   X = L_kk^-1 from the diagonal step
//   inverse           R_ij -= U_ki' W_kj, R_ik = -U_ki' X, W_kj = X R_kj     (W = L^-1 by forward substitution on I)
// Slot (i, j) holds S_ij until step i, U_ij during step i, then R_ji until step j, where it becomes W_ji and is
// consumed at once (diag((LL')^-1) += W^2, beta += W' z): T (T + 1) / 2 tile slots, nothing else.
//
// Diagonal step (the only serial part): the 16 x 16 block goes through 2.4 KB of wave-private LDS into row layout
// (lane r owns row r, as in solve_wide.hip), lanes 16..31 start as the identity and lanes 32.. as the y column; the 16
// right-looking pivot steps apply the same instruction to all of them, so X = L_kk^-1 and z_k = L_kk^-1 c_k cost
// nothing extra.  Constant / aliased columns get inv = 0: column, row of X and z_j vanish (solve_wide.hip's rule).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace anofox {
namespace tiles {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr double kAliasTol = 1e-11;  // solve_wide.hip: kAliasTolW
constexpr double kRefineTol = 1e-7;  // kRefineTolW
constexpr double kPivotWarn = 1e-3;  // kPivotWarnW

__device__ __forceinline__ double nan64() { return __builtin_nan(""); }

__device__ __forceinline__ double rl(double v, int src_lane) {
	return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane),
	                        __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}

__device__ __forceinline__ void wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// sum over the four lanes (q = 0..3) that share a column n; every lane gets the result
__device__ __forceinline__ double sum_q(double v) {
	v += __shfl_xor(v, 16, 64);
	v += __shfl_xor(v, 32, 64);
	return v;
}
__device__ __forceinline__ double sum_wave(double v) {
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
	return v;
}

__device__ __forceinline__ d4 mfma4(const d4 &a, const d4 &b, d4 c) {
#pragma unroll
	for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], c, 0, 0, 0);
	return c;
}

// acc -= v[lane C] * v: the broadcast goes through one fixed scalar register pair and is consumed at once.  (Written
// with the readlane builtins the compiler issues all fifteen broadcasts of a pivot step first — thirty scalar registers
// per step, hundreds spilled to vector lanes once it overlaps consecutive steps.)
// FIRST: the block may directly follow the instruction that wrote v.  gfx950 needs a wait state between a vector
// instruction writing a register and a v_readlane reading it; the compiler inserts that s_nop for its own readlanes
// but cannot see into an asm block (without it the broadcast picked up a stale low half: errors of 1e-7).
template <int C, bool FIRST>
__device__ __forceinline__ void bcast_fnma_c(double &acc, double v) {
	const int lo = __double2loint(v), hi = __double2hiint(v);
	if (FIRST)
		asm volatile("s_nop 1\n\tv_readlane_b32 s98, %2, %4\n\tv_readlane_b32 s99, %3, %4\n\tv_fma_f64 %0, -s[98:99], %1, %0"
		             : "+v"(acc)
		             : "v"(v), "v"(lo), "v"(hi), "n"(C)
		             : "s98", "s99");
	else
		asm volatile("v_readlane_b32 s98, %2, %4\n\tv_readlane_b32 s99, %3, %4\n\tv_fma_f64 %0, -s[98:99], %1, %0"
		             : "+v"(acc)
		             : "v"(v), "v"(lo), "v"(hi), "n"(C)
		             : "s98", "s99");
}

// compile-time loop: f(integral_constant<int, B>), ..., f(integral_constant<int, E - 1>)
template <int B, int E, class F>
__device__ __forceinline__ void sfor(F &&f) {
	if constexpr (B < E) {
		f(std::integral_constant<int, B>{});
		sfor<B + 1, E>(f);
	}
}

// 1 / sqrt(d) for d > 0: v_rsq_f64 (about 2^-26) plus one cubic step
__device__ __forceinline__ double rsqrt_pos(double d) {
	const double y = __builtin_amdgcn_rsq(d);
	const double e = fma(-d * y, y, 1.0);
	return fma(y * e, fma(0.375, e, 0.5), y);
}

template <int T>
__device__ __forceinline__ constexpr int slot(int i, int j) { // i <= j
	return i * T - i * (i - 1) / 2 + (j - i);
}

// (r4) tiles parked in the wave's LDS at T = 8: the LAST block column and the top of the one before it.  Every slot is touched
// once per block step, so a parked one costs 8 ds_read_b64 + 8 ds_write_b64 per step where a spilled register costs scratch
// (HBM / L2) traffic: 35 registers were still spilled after the partial sums had moved to LDS.
template <int T, bool PARKED>
__host__ __device__ constexpr int parked_tiles() { return (PARKED && T == 8) ? 10 : 0; } // (11: no scratch at all, but 41.6 KB per wave: three waves per CU instead of four)
template <int T, bool PARKED>
__host__ __device__ constexpr int parked_index(int t) { // -1: the slot lives in registers
	for (int i = 0; i < parked_tiles<T, PARKED>(); ++i) { // the last block column from the top, then the one before it
		const int col = i < T ? T - 1 : T - 2, row = i < T ? i : i - T;
		if (t == slot<T>(row, col)) return i;
	}
	return -1;
}

constexpr int kDsLd = 18; // row stride of the diagonal-block image (doubles): 16-byte aligned rows
constexpr int kXsLd = 17;

// PARKED = false: everything in registers, as in round 3 (ANOFOX_SOLVE_PARK=0: the A/B switch of the measurement)
template <int T, int WPE, bool PARKED = true>
__global__ __launch_bounds__(64, WPE) void solve_tiles_kernel(WideArgs args) {
	constexpr int P16 = 16 * T;
	constexpr int NT = T * (T + 1) / 2;
	// (r4) PARK: at seven and eight column tiles the 28 / 36 tile slots (224 / 288 registers) plus the per-block partial sums —
	// c (the y column), beta and diag of the inverse: three doubles per lane and block, 42-48 registers — did not fit the 512
	// registers of one wave per SIMD: 59 registers spilled at T = 8, i.e. 1.9 GB read + 0.7 GB written of SCRATCH per 13 785-group
	// launch against 1.1 GB of records (profiles/hbm_traffic.json, r03).  Those three arrays are touched once or twice per
	// (block step, block) pair, so they live in the wave's own LDS instead ([block][lane]: conflict-free, 12 KB per wave of the
	// 160 KB four waves share), and the column sums are re-read from the record where the statistics need them.
	constexpr bool PARK = PARKED && T >= 7;
	constexpr int NPT = parked_tiles<T, PARKED>();
	__shared__ __attribute__((aligned(16))) double lds[17 * kDsLd + 16 * kXsLd + 16 + 2 * P16 + (PARK ? 3 * T * 64 : 0) + NPT * 256];
	double *Ds = lds;                 // [17][18]: the diagonal block (rows 0..15) and the y column (row 16)
	double *Xs = Ds + 17 * kDsLd;     // [16][17]: X = L_kk^-1
	double *zs = Xs + 16 * kXsLd;     // [16]: z_k
	double *d0s = zs + 16;            // [P16]: the diagonal before the factorisation
	double *acts = d0s + P16;         // [P16]: 1.0 = column takes part
	double *park = acts + P16;        // PARK: [3][T][64] c partial sums, beta partial sums, diag partial sums
	double *ptile = park + (PARK ? 3 * T * 64 : 0); // [NPT][4][64]: the parked tile slots, register r of lane l at [r][l]

	const int p = args.p;
	const int lane = threadIdx.x;
	const int q = lane >> 4, n = lane & 15;
	const bool icpt = args.fit_intercept != 0;
	const int model = args.model;

	// one group per workgroup (the launcher's grid is the group count): as a grid-stride loop the compiler hoisted every
	// group-invariant mask and address out of it and kept them in (spilled) registers
	const int64_t gl = blockIdx.x;
	if (gl >= args.n_groups) return;
	{
		const int64_t g = args.group_base + gl;
		const double *rec = args.moments + gl * (int64_t)wide_record_len(T);
		const double *vec = rec + (int64_t)NT * 256;
		const double *sc = vec + 4 * P16;
		double *core = args.core + g * (int64_t)(p + 6);
		double *inf = (args.inference && args.compute_inference) ? args.inference + g * (int64_t)(5 * p + 2) : nullptr;
		const int64_t nrows = args.rule_counts ? args.rule_counts[g] : args.row_offsets[g + 1] - args.row_offsets[g];

		auto write_null = [&](int status, bool core_too) {
			if (core_too)
				for (int k = lane; k < p + 6; k += 64) core[k] = (k == p + 5) ? (double)status : nan64();
			if (inf)
				for (int k = lane; k < 5 * p + 2; k += 64) inf[k] = nan64();
		};

		const double sy = sc[0], syy = sc[1], sw = sc[2], cnt = sc[3], first_y = sc[4];
		int status = ANOFOX_ERROR_SUCCESS;
		if (nrows < 2) status = ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS;                                       // ols_aggregate.cpp:263-267
		else if (model == ANOFOX_HIP_MODEL_RIDGE && args.alpha < 0.0) status = ANOFOX_ERROR_INVALID_ALPHA;  // ridge.rs:38-40
		else if (!(cnt > 0.0)) status = ANOFOX_ERROR_NO_VALID_DATA;                                         // ols.rs:68-70
		if (status != ANOFOX_ERROR_SUCCESS) {
			write_null(status, true);
			return;
		}

		// per column block J: this lane's column 16 J + n
		double scol[T]; // column sums
		int peff_l = 0;
		sfor<0, T>([&](auto J_) __attribute__((always_inline)) {
			constexpr int J = decltype(J_)::value;
			const int col = 16 * J + n;
			scol[J] = col < p ? vec[0 * P16 + col] : 0.0;
			const bool a = col < p && vec[3 * P16 + col] != 0.0;
			peff_l += (a && q == 0) ? 1 : 0;
		});
		const int peff = (int)sum_wave((double)peff_l);

		const double cyy_c = syy - sy * sy / sw;
		const double ymean = (icpt ? first_y : 0.0) + sy / sw;
		if (peff == 0) { // ols.rs:101-130, wls.rs:119-150
			if (!icpt) {
				write_null(ANOFOX_ERROR_INSUFFICIENT_DATA, true);
			} else {
				write_null(0, false); // inference: None
				for (int k = lane; k < p + 6; k += 64) {
					double v = nan64();
					if (k == p) v = ymean;
					else if (k == p + 1 || k == p + 2 || k == p + 5) v = 0.0;
					else if (k == p + 3) v = (model == ANOFOX_HIP_MODEL_WLS) ? sqrt(cyy_c / sw) : sqrt(cyy_c / (cnt - 1.0));
					else if (k == p + 4) v = cnt;
					core[k] = v;
				}
			}
			return;
		}
		if (cnt < (double)(peff + (icpt ? 1 : 0))) { // ols.rs:132-139
			write_null(ANOFOX_ERROR_INSUFFICIENT_DATA, true);
			return;
		}

		double lam = 0.0;
		bool glmnet_cancels = false; // sd_y from uncentred moments of a nearly constant y: the refinement re-sums it over the rows
		if (model == ANOFOX_HIP_MODEL_RIDGE) {
			lam = args.alpha;
			if (args.lambda_scaling == ANOFOX_LAMBDA_SCALING_GLMNET) {
				lam = cnt * args.alpha / sqrt(cyy_c / cnt);
				glmnet_cancels = !icpt && !(cyy_c * kGlmnetCancelRatio > syy);
			}
		}
		const double tss = icpt ? cyy_c : syy;
		const double inv_sw = 1.0 / sw;

		// ---- the upper triangle, centred (intercept), ridge term on the diagonal, padding rows / columns zero ----
		// (every index into tile[] is a constant expression — sfor, not a loop — so that the array is split into
		// registers before any unrolling; as an unrolled loop nest the 36 tiles of T = 8 stayed in scratch memory)
		d4 tile[NT];
		// slot t, wherever it lives (t is a constant expression everywhere)
		auto tget = [&](auto t_) __attribute__((always_inline)) -> d4 {
			constexpr int t = decltype(t_)::value;
			constexpr int pi = parked_index<T, PARKED>(t);
			if constexpr (pi >= 0) {
				d4 v;
#pragma unroll
				for (int r = 0; r < 4; ++r) v[r] = ptile[(pi * 4 + r) * 64 + lane];
				return v;
			} else {
				return tile[t];
			}
		};
		auto tset = [&](auto t_, const d4 &v) __attribute__((always_inline)) {
			constexpr int t = decltype(t_)::value;
			constexpr int pi = parked_index<T, PARKED>(t);
			if constexpr (pi >= 0) {
#pragma unroll
				for (int r = 0; r < 4; ++r) ptile[(pi * 4 + r) * 64 + lane] = v[r];
			} else {
				tile[t] = v;
			}
		};
#define TS(i, j) std::integral_constant<int, slot<T>(i, j)> {}
		wave_lds_sync(); // the previous group's reads of d0s / acts are done
		// (loaded and centred one tile row at a time, the next row's loads issued before this row's arithmetic: all 144
		// loads of T = 8 at once need more than the 256 architectural registers a load can target and spilled)
		auto load_row = [&](auto I_) __attribute__((always_inline)) {
			constexpr int I = decltype(I_)::value;
			sfor<I, T>([&](auto J_) __attribute__((always_inline)) {
				constexpr int t = slot<T>(I, decltype(J_)::value);
				d4 v;
#pragma unroll
				for (int r = 0; r < 4; ++r) v[r] = rec[(int64_t)t * 256 + 64 * r + lane];
				tset(std::integral_constant<int, t>{}, v);
			});
		};
		load_row(std::integral_constant<int, 0>{});
		sfor<0, T>([&](auto I_) __attribute__((always_inline)) {
			constexpr int I = decltype(I_)::value;
			if constexpr (I + 1 < T) load_row(std::integral_constant<int, I + 1>{});
			__builtin_amdgcn_sched_barrier(0);
			double srow[4];
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int a = 16 * I + q + 4 * r;
				srow[r] = (icpt && a < p) ? vec[0 * P16 + a] * inv_sw : 0.0;
			}
			sfor<I, T>([&](auto J_) __attribute__((always_inline)) {
				constexpr int J = decltype(J_)::value;
				d4 tl = tget(TS(I, J));
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					double v = fma(-srow[r], scol[J], tl[r]);
					if (I == J && q + 4 * r == n) v += lam;
					if (J == T - 1) { // the only block with padding columns (and, on the diagonal, padding rows)
						const bool in = (16 * J + n < p) && (I < T - 1 || 16 * I + q + 4 * r < p);
						v = in ? v : 0.0;
					}
					tl[r] = v;
				}
				tset(TS(I, J), tl);
			});
			// the diagonal before the factorisation and the activity flags of block I, for the pivot tests
			if (q == 0) {
				const int col = 16 * I + n;
				double dv = 1.0, av = 0.0;
				if (col < p) {
					dv = rec[(int64_t)slot<T>(I, I) * 256 + 17 * n];
					if (icpt) dv = fma(-scol[I] * inv_sw, scol[I], dv);
					dv += lam;
					av = vec[3 * P16 + col] != 0.0 ? 1.0 : 0.0;
				}
				d0s[col] = dv;
				acts[col] = av;
			}
		});
		// y column (centred Sxy), kept as partial sums over q: c_J[n] = sum_q cpart[J](q, n)
		double cpart[PARK ? 1 : T], bacc[PARK ? 1 : T], dacc[PARK ? 1 : T];
		sfor<0, T>([&](auto J_) __attribute__((always_inline)) {
			constexpr int J = decltype(J_)::value;
			const int col = 16 * J + n;
			double v = 0.0;
			if (q == 0 && col < p) {
				v = vec[1 * P16 + col];
				if (icpt) v = fma(-scol[J] * inv_sw, sy, v);
			}
			if constexpr (PARK) {
				park[(0 * T + J) * 64 + lane] = v;
			} else {
				cpart[J] = v;
				bacc[J] = 0.0;
				dacc[J] = 0.0;
			}
		});
		wave_lds_sync();

		double min_ratio = 1.0, zz_l = 0.0;
		bool band = false; // (r4) a column dropped with a pivot in 1e-13 .. 1e-11 of its diagonal: queued, the refit decides (solve_narrow.hip)
		unsigned live_bits[T];

		sfor<0, T>([&](auto k_) __attribute__((always_inline)) {
			constexpr int k = decltype(k_)::value;
			// (the lane's coordinates are re-derived from an opaque copy in every step: otherwise the compiler hoists the
			// step-invariant masks, identity columns and LDS addresses of all T steps out of the loops and keeps ~200
			// registers of them alive — spilled to scratch memory at T = 8)
			int lane_k = lane;
			asm volatile("" : "+v"(lane_k));
			const int q = lane_k >> 4, n = lane_k & 15, lane = lane_k;
			(void)lane;
			// ---- (1) diagonal block and y column -> row layout ----
			{
				const d4 dk = tget(TS(k, k));
#pragma unroll
				for (int r = 0; r < 4; ++r) Ds[(q + 4 * r) * kDsLd + n] = dk[r];
				double cpk;
				if constexpr (PARK) cpk = park[(0 * T + k) * 64 + lane];
				else cpk = cpart[k];
				const double ck = sum_q(cpk);
				if (q == 0) Ds[16 * kDsLd + n] = ck;
			}
			wave_lds_sync();
			double reg[16];
			{
				const int row = lane < 16 ? lane : 16;
#pragma unroll
				for (int c = 0; c < 16; ++c) reg[c] = Ds[row * kDsLd + c];
				if (q == 1) {
#pragma unroll
					for (int c = 0; c < 16; ++c) reg[c] = (c == n) ? 1.0 : 0.0;
				}
			}
			// pivot j is accepted iff its column takes part and d > 1e-11 diag0 and d > 0 (solve_wide.hip's test): one
			// threshold per column
			const double d0 = d0s[16 * k + n];
			const double thr = (acts[16 * k + n] != 0.0 && d0 == d0) ? fmax(kAliasTol * d0, 0.0) : __builtin_inf();
			// ---- (2) 16 right-looking pivot steps: lanes 0..15 rows of the block, 16..31 identity -> X, 32.. y -> z ----
			sfor<0, 16>([&](auto j_) __attribute__((always_inline)) {
				constexpr int j = decltype(j_)::value;
				double d = rl(reg[j], j);
				double th = rl(thr, j);
				// (the pivot is wave-uniform; kept in vector registers so that the test becomes a select, not a branch)
				asm volatile("" : "+v"(d), "+v"(th));
				const bool ok = d > th;
				band = band || (!ok && d * 1e2 > th); // (th = 1e-11 diag0, +inf for a column that takes no part)
				const double inv = ok ? rsqrt_pos(ok ? d : 1.0) : 0.0;
				const double lj = reg[j] * inv;
				reg[j] = lj;
				sfor<j + 1, 16>([&](auto c_) __attribute__((always_inline)) {
					constexpr int c = decltype(c_)::value;
					bcast_fnma_c<c, c == j + 1>(reg[c], lj);
				});
			});
			// lane r < 16 now holds L[r][r] (0 = column dropped) in reg[r]: live flags and pivot ratios from that
			{
				double dg = 0.0;
#pragma unroll
				for (int c = 0; c < 16; ++c) dg = (lane == c) ? reg[c] : dg;
				const bool lv = lane < 16 && dg != 0.0;
				live_bits[k] = (unsigned)(__ballot(lv) & 0xffffull);
				min_ratio = lv ? fmin(min_ratio, dg * dg / d0) : min_ratio;
			}
			// ---- (3) X and z back to tile layout ----
			if (q == 1) {
#pragma unroll
				for (int r = 0; r < 16; ++r) Xs[r * kXsLd + n] = reg[r];
			}
			if (lane == 32) {
#pragma unroll
				for (int r = 0; r < 16; ++r) zs[r] = reg[r];
			}
			wave_lds_sync();
			d4 Xc, XT; // C-layout(X), C-layout(X')
			double zq[4];
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				Xc[r] = Xs[(q + 4 * r) * kXsLd + n];
				XT[r] = Xs[n * kXsLd + q + 4 * r];
				zq[r] = zs[q + 4 * r];
				if (n == 0) zz_l = fma(zq[r], zq[r], zz_l);
			}
			const d4 zero = {0.0, 0.0, 0.0, 0.0};
			// ---- (4) block row k of W = L^-1 is final: W_kj = X R_kj (j < k), W_kk = X; consume it ----
			sfor<0, k + 1>([&](auto j_) __attribute__((always_inline)) {
				constexpr int j = decltype(j_)::value;
				d4 w;
				if constexpr (j < k) {
					w = mfma4(XT, tget(TS(j, k)), zero);
					tset(TS(j, k), w); // B operand of this step's R updates
				} else {
					w = Xc;
				}
				if constexpr (PARK) {
					double db = 0.0, dd2 = 0.0;
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						dd2 = fma(w[r], w[r], dd2);
						db = fma(w[r], zq[r], db);
					}
					// (block j's first contribution comes at step k = j)
					double *bp = park + (1 * T + j) * 64 + lane, *dp = park + (2 * T + j) * 64 + lane;
					*bp = (j == k) ? db : *bp + db;
					*dp = (j == k) ? dd2 : *dp + dd2;
				} else {
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						dacc[j] = fma(w[r], w[r], dacc[j]);
						bacc[j] = fma(w[r], zq[r], bacc[j]);
					}
				}
			});
			// ---- (5) panel: U_kj = X S_kj, and its share of the y column ----
			sfor<k + 1, T>([&](auto j_) __attribute__((always_inline)) {
				constexpr int j = decltype(j_)::value;
				const d4 u = mfma4(XT, tget(TS(k, j)), zero);
				tset(TS(k, j), u);
				if constexpr (PARK) {
					double dc = 0.0;
#pragma unroll
					for (int r = 0; r < 4; ++r) dc = fma(-u[r], zq[r], dc);
					park[(0 * T + j) * 64 + lane] += dc;
				} else {
#pragma unroll
					for (int r = 0; r < 4; ++r) cpart[j] = fma(-u[r], zq[r], cpart[j]);
				}
			});
			// ---- (6) trailing updates: S_ij -= U_ki' U_kj, R_ij -= U_ki' W_kj, and R_ik = -U_ki' X is born ----
			sfor<k + 1, T>([&](auto i_) __attribute__((always_inline)) {
				constexpr int i = decltype(i_)::value;
				d4 nu;
				const d4 uki = tget(TS(k, i));
#pragma unroll
				for (int r = 0; r < 4; ++r) nu[r] = -uki[r];
				sfor<i, T>([&](auto j_) __attribute__((always_inline)) {
					constexpr int j = decltype(j_)::value;
					tset(TS(i, j), mfma4(nu, tget(TS(k, j)), tget(TS(i, j))));
				});
				sfor<0, k>([&](auto j_) __attribute__((always_inline)) {
					constexpr int j = decltype(j_)::value;
					tset(TS(j, i), mfma4(nu, tget(TS(j, k)), tget(TS(j, i))));
				});
				tset(TS(k, i), mfma4(nu, Xc, zero)); // slot (k, i): U_ki is dead, R_ik takes its place
			});
		});

#undef TS
		// ---- coefficients, diag((LL')^-1), statistics ----
		double rk = 0.0, bc = 0.0, bb = 0.0, xb = 0.0;
		double beta[T], dinv[T];
		bool live[T];
		sfor<0, T>([&](auto J_) __attribute__((always_inline)) {
			constexpr int J = decltype(J_)::value;
			if constexpr (PARK) {
				beta[J] = sum_q(park[(1 * T + J) * 64 + lane]);
				dinv[J] = sum_q(park[(2 * T + J) * 64 + lane]);
			} else {
				beta[J] = sum_q(bacc[J]);
				dinv[J] = sum_q(dacc[J]);
			}
			const int col = 16 * J + n;
			live[J] = col < p && ((live_bits[J] >> n) & 1u) != 0u;
			if (q == 0 && live[J]) {
				rk += 1.0;
				const double qv = vec[1 * P16 + col];
				const double sc_j = PARK ? vec[0 * P16 + col] : scol[J]; // (PARK: the column sum from the record again, not from a register kept for it)
				const double cj = icpt ? qv - sc_j * sy / sw : qv;
				const double fx = icpt ? vec[2 * P16 + col] : 0.0;
				bc = fma(beta[J], cj, bc);
				bb = fma(beta[J], beta[J], bb);
				xb = fma(beta[J], fx + sc_j / sw, xb);
			}
		});
		const double rk_t = sum_wave(rk), bc_t = sum_wave(bc), bb_t = sum_wave(bb), xb_t = sum_wave(xb), zz_t = sum_wave(zz_l);
		const int rank = (int)rk_t;
		double rss;
		if (model == ANOFOX_HIP_MODEL_RIDGE) rss = tss - bc_t - lam * bb_t;
		else rss = tss - zz_t; // Syy - |L^-1 Sxy|^2
		double mr = min_ratio; // smallest accepted pivot ratio of the group
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1) mr = fmin(mr, __shfl_xor(mr, m, 64));
		const int n_par = rank + (icpt ? 1 : 0);
		const double df = cnt - (double)n_par;
		// Nearly square designs (fewer residual degrees of freedom than a quarter of the columns) are ill conditioned
		// whatever the column scales, and a ridge penalty hides that from the pivot test (it lifts every pivot): they take
		// the refinement passes as well.  (Deep fuzz sweep, ridge p = 127, n = 129: 4.6e-9 without.)
		bool refine = !(rss > kRefineTol * tss) || mr < kPivotWarn || df < 0.25 * (double)rank || glmnet_cancels || __any(band);
		{ // a coefficient whose a-priori error exceeds the parity scale (coef_bound_weak, common.h)
			double bmax = 0.0;
			sfor<0, T>([&](auto J_) __attribute__((always_inline)) {
				constexpr int J = decltype(J_)::value;
				bmax = fmax(bmax, (q == 0 && live[J]) ? fabs(beta[J]) : 0.0);
			});
#pragma unroll
			for (int m = 32; m >= 1; m >>= 1) bmax = fmax(bmax, __shfl_xor(bmax, m, 64));
			bool weak = false;
			sfor<0, T>([&](auto J_) __attribute__((always_inline)) {
				constexpr int J = decltype(J_)::value;
				if (q == 0 && live[J]) weak = weak || coef_bound_weak(beta[J], bmax, d0s[16 * J + n], tss, mr);
			});
			refine = refine || __any(weak);
		}
		const double dfm = (double)rank;
		const double r2 = 1.0 - rss / tss;
		const double fstat = ((tss - rss) / dfm) / (rss / df);
		const double sigma2 = rss / df;
		sfor<0, T>([&](auto J_) __attribute__((always_inline)) {
			constexpr int J = decltype(J_)::value;
			const int col = 16 * J + n;
			if (q == 0 && col < p) {
				core[col] = live[J] ? beta[J] : nan64();
				if (inf) inf[col] = live[J] ? sqrt(sigma2 * dinv[J]) : nan64();
			}
		});
		if (lane < 6) {
			double v;
			if (lane == 0) v = icpt ? ymean - xb_t : nan64();
			else if (lane == 1) v = r2;
			else if (lane == 2) v = 1.0 - (1.0 - r2) * (cnt - (icpt ? 1.0 : 0.0)) / df;
			else if (lane == 3) v = sqrt(rss / df);
			else if (lane == 4) v = cnt;
			else v = 0.0;
			core[p + lane] = v;
		}
		if (lane == 0) {
			if (refine) {
				const int sl = atomicAdd(args.refine_count, 1);
				args.refine_list[sl] = (int32_t)gl;
			}
			if (inf) { // t, p, interval and the F p-value follow in inference_wide_finish_kernel; df travels in the last slot
				inf[5 * p] = fstat;
				inf[5 * p + 1] = df;
			}
		}
	}
}

} // namespace tiles

template <int T, int WPE>
hipError_t launch_solve_tiles_T(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups > (int64_t)0x7fffffff) return hipErrorInvalidValue; // (a slab of run_wide_batch holds at most 2^30 / 2.6 KB groups)
	static const bool park_on = !(getenv("ANOFOX_SOLVE_PARK") && atoi(getenv("ANOFOX_SOLVE_PARK")) == 0);
	bool launched = false;
	if constexpr (T >= 7) {
		if (!park_on) {
			hipLaunchKernelGGL((tiles::solve_tiles_kernel<T, WPE, false>), dim3((unsigned)a.n_groups), dim3(64), 0, stream, a);
			launched = true;
		}
	}
	if (!launched) hipLaunchKernelGGL((tiles::solve_tiles_kernel<T, WPE>), dim3((unsigned)a.n_groups), dim3(64), 0, stream, a);
	return hipGetLastError();
}

} // namespace anofox
