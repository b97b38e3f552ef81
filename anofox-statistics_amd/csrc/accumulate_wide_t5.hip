// accumulate_wide_t5.hip — the wide accumulation kernels of 5 column tiles (64 < p <= 80); see accumulate_wide_impl.h
#include "accumulate_wide_impl.h"

namespace anofox {
template hipError_t launch_accumulate_wide_T<5>(const WideArgs &, hipStream_t);
} // namespace anofox
