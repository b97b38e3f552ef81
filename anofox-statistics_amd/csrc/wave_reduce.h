// wave_reduce.h — cross-lane helpers for 64-lane wavefronts: the transposing butterfly that sums up to 64 per-lane
// partials so that total k lands on lane k (63 two-register exchanges instead of 6 per value).
#pragma once
#include <hip/hip_runtime.h>

namespace anofox {

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
	return __hiloint2double(hi, lo);
}

// After the call: lanes [0,32) hold a[l] + a[l+32] (both halves' partials of `a`), lanes [32,64) hold
// the same for `b`.
__device__ __forceinline__ double fold32(double a, double b) {
	auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
	auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
	return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// Even 16-lane rows end up with a[l] + a[l+16], odd rows with b[l-16] + b[l].
__device__ __forceinline__ double fold16(double a, double b) {
	auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
	auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
	return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// Lanes with (lane & M) == 0 end up with a summed over the pair {l, l^M}, the others with b.
template <int M>
__device__ __forceinline__ double fold_shfl(double a, double b, int lane) {
	const bool upper = (lane & M) != 0;
	const double keep = upper ? b : a;
	const double send = upper ? a : b;
	return keep + __shfl_xor(send, M, 64);
}

// v[0..63] per-lane partial sums in, lane k holds the wave total of v[k] in v[0] out.
__device__ __forceinline__ void transpose_reduce64(double (&v)[64], int lane) {
#pragma unroll
	for (int i = 0; i < 32; ++i) v[i] = fold32(v[i], v[i + 32]);
#pragma unroll
	for (int i = 0; i < 16; ++i) v[i] = fold16(v[i], v[i + 16]);
#pragma unroll
	for (int i = 0; i < 8; ++i) v[i] = fold_shfl<8>(v[i], v[i + 8], lane);
#pragma unroll
	for (int i = 0; i < 4; ++i) v[i] = fold_shfl<4>(v[i], v[i + 4], lane);
#pragma unroll
	for (int i = 0; i < 2; ++i) v[i] = fold_shfl<2>(v[i], v[i + 2], lane);
	v[0] = fold_shfl<1>(v[0], v[1], lane);
}

} // namespace anofox
