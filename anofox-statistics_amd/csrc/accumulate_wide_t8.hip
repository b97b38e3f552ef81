// accumulate_wide_t8.hip — the wide accumulation kernels of 8 column tiles (112 < p <= 128); see accumulate_wide_impl.h
#include "accumulate_wide_impl.h"

namespace anofox {
template hipError_t launch_accumulate_wide_T<8>(const WideArgs &, hipStream_t);
} // namespace anofox

#ifdef ANOFOX_SOLVE_STAMPS
// diagnostic build: the phase stamps of the p = 128 kernels (this translation unit's copy of g_acc_stamps)
extern "C" __attribute__((visibility("default"))) int anofox_hip_diag_acc_stamps(unsigned long long *out8) {
	return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(anofox::g_acc_stamps), 8 * sizeof(unsigned long long));
}
#endif
