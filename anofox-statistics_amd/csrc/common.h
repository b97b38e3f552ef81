// common.h — shared definitions between the HIP kernels and the host side of libanofox_stats_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/anofox_stats_hip.h"

namespace anofox {

// Register-resident ("narrow") path: one wavefront streams one group, every lane keeps the whole
// (p+1)(p+2)/2 moment triangle in VGPRs.  Above this the LDS/MFMA ("wide") path takes over.
constexpr int kNarrowMaxP = 8;

// Per-group moment record written by the accumulate kernel and read by the solve kernel.
// z = (x_1 .. x_p, y), Z = p + 1 columns, shifted by the group's first valid row ("first") when an
// intercept is fitted (shift 0 otherwise):  d = z - shift.
//   [0, Z)            s_a   = sum w d_a
//   [Z, Z+ZZ)         q_ab  = sum w d_a d_b, a <= b, row-major upper triangle
//   [Z+ZZ]            sw    = sum w           (w == 1 for OLS / ridge)
//   [KRED, KRED+Z)    first = z at the first valid row (x_first drives the constant-column test,
//                             crates/anofox-stats-core/src/models/ols.rs:76-87)
//   [KRED+Z]          cnt   = number of valid rows
//   [KRED+Z+1]        mask  = bit j set iff |x_j - x_j,first| >= 1e-10 on some valid row
template <int P>
struct MomentLayout {
	static constexpr int Z = P + 1;
	static constexpr int ZZ = Z * (Z + 1) / 2;
	static constexpr int OFF_S = 0;
	static constexpr int OFF_Q = Z;
	static constexpr int OFF_SW = Z + ZZ;
	static constexpr int KRED = Z + ZZ + 1; // entries that need a cross-lane reduction
	static constexpr int OFF_FIRST = KRED;
	static constexpr int OFF_CNT = KRED + Z;
	static constexpr int OFF_MASK = KRED + Z + 1;
	static constexpr int REC = KRED + Z + 2;
	__host__ __device__ static constexpr int q_index(int a, int b) { // a <= b
		return OFF_Q + a * Z - a * (a - 1) / 2 + (b - a);
	}
};

inline __host__ __device__ int moment_record_len(int p) {
	const int Z = p + 1;
	return Z + Z * (Z + 1) / 2 + 1 + Z + 2;
}

struct BatchArgs {
	const int64_t *row_offsets; // [G+1]
	const double *y;            // [N]
	const double *x[kNarrowMaxP];
	const double *w;            // [N] or nullptr
	int64_t n_groups;
	int64_t n_rows;
	int p;
	// options
	int model; // AnofoxHipModel
	int fit_intercept;
	int compute_inference;
	int lambda_scaling;
	double confidence_level;
	double alpha;
	// workspace / outputs
	double *moments;      // [G * REC]
	double *core;         // [G * (p+6)]
	double *inference;    // [G * (5p+2)] or nullptr
	int32_t *refine_list; // [G]   groups whose RSS must be recomputed from residuals
	int32_t *refine_count; // [1]
	double *rss_direct;   // [G]
};

// launchers implemented in the .hip translation units
hipError_t launch_accumulate_narrow(const BatchArgs &a, hipStream_t stream);
hipError_t launch_solve_narrow(const BatchArgs &a, bool refine_pass, hipStream_t stream);
hipError_t launch_residual_rss(const BatchArgs &a, hipStream_t stream);

} // namespace anofox
