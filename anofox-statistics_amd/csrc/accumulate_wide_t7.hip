// accumulate_wide_t7.hip — the wide accumulation kernels of 7 column tiles (96 < p <= 112); see accumulate_wide_impl.h
#include "accumulate_wide_impl.h"

namespace anofox {
template hipError_t launch_accumulate_wide_T<7>(const WideArgs &, hipStream_t);
} // namespace anofox
