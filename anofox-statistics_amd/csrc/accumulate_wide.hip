// accumulate_wide.hip — moment accumulation for wide designs (8 < p <= 128) on the FP64 matrix cores: the launcher,
// and the kernels of 1..4 column tiles.  The kernels themselves are in accumulate_wide_impl.h; 5..8 tiles are compiled
// by accumulate_wide_t5.hip .. accumulate_wide_t8.hip (one translation unit each, so that make -j builds them side by side).
#include "accumulate_wide_impl.h"

namespace anofox {

template hipError_t launch_accumulate_wide_T<1>(const WideArgs &, hipStream_t);
template hipError_t launch_accumulate_wide_T<2>(const WideArgs &, hipStream_t);
template hipError_t launch_accumulate_wide_T<3>(const WideArgs &, hipStream_t);
template hipError_t launch_accumulate_wide_T<4>(const WideArgs &, hipStream_t);
extern template hipError_t launch_accumulate_wide_T<5>(const WideArgs &, hipStream_t);
extern template hipError_t launch_accumulate_wide_T<6>(const WideArgs &, hipStream_t);
extern template hipError_t launch_accumulate_wide_T<7>(const WideArgs &, hipStream_t);
extern template hipError_t launch_accumulate_wide_T<8>(const WideArgs &, hipStream_t);

hipError_t launch_accumulate_wide(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	switch (wide_tiles(a.p)) {
	case 1: return launch_accumulate_wide_T<1>(a, stream);
	case 2: return launch_accumulate_wide_T<2>(a, stream);
	case 3: return launch_accumulate_wide_T<3>(a, stream);
	case 4: return launch_accumulate_wide_T<4>(a, stream);
	case 5: return launch_accumulate_wide_T<5>(a, stream);
	case 6: return launch_accumulate_wide_T<6>(a, stream);
	case 7: return launch_accumulate_wide_T<7>(a, stream);
	case 8: return launch_accumulate_wide_T<8>(a, stream);
	default: return hipErrorInvalidValue;
	}
}

template hipError_t launch_accumulate_wide_followup_T<3>(const WideArgs &, hipStream_t);
template hipError_t launch_accumulate_wide_followup_T<4>(const WideArgs &, hipStream_t);

hipError_t launch_accumulate_wide_followup(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	switch (wide_tiles(a.p)) {
	case 3: return launch_accumulate_wide_followup_T<3>(a, stream);
	case 4: return launch_accumulate_wide_followup_T<4>(a, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace anofox

#ifdef ANOFOX_SOLVE_STAMPS
// diagnostic build: the phase stamps and milestones of the kernels of 1 .. 4 column tiles (this translation unit's copies)
extern "C" __attribute__((visibility("default"))) int anofox_hip_diag_acc_stamps_t4(unsigned long long *out16) {
	hipError_t rc = hipMemcpyFromSymbol(out16, HIP_SYMBOL(anofox::g_acc_stamps), 8 * sizeof(unsigned long long));
	if (rc != hipSuccess) return (int)rc;
	return (int)hipMemcpyFromSymbol(out16 + 8, HIP_SYMBOL(anofox::g_acc_marks), 8 * sizeof(unsigned long long));
}
#endif
