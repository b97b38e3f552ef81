// accumulate_wide_t6.hip — the wide accumulation kernels of 6 column tiles (80 < p <= 96); see accumulate_wide_impl.h
#include "accumulate_wide_impl.h"

namespace anofox {
template hipError_t launch_accumulate_wide_T<6>(const WideArgs &, hipStream_t);
} // namespace anofox
