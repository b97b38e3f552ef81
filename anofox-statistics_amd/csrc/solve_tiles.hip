// solve_tiles.hip — instantiations and launcher of the one-wavefront-per-group primary solve (solve_tiles_impl.h).
#include "solve_tiles_impl.h"

namespace anofox {

bool solve_tiles_supports(int p) {
	const int T = wide_tiles(p);
	return T >= 1 && T <= 8;
}

// the second template argument is the occupancy (waves per SIMD) the register budget is cut for: 36 tiles are 288
// vector registers per lane at T = 8, so one wave per SIMD there; the narrow ones fit two to five
hipError_t launch_solve_tiles(const WideArgs &a, hipStream_t stream) {
	if (a.n_groups <= 0) return hipSuccess;
	switch (wide_tiles(a.p)) {
	case 1: return launch_solve_tiles_T<1, 4>(a, stream);  // 68 registers (2 .. 6 waves per SIMD measure the same: 0.25 ms per 100 000 groups)
	case 2: return launch_solve_tiles_T<2, 3>(a, stream);  // 104
	case 3: return launch_solve_tiles_T<3, 3>(a, stream);  // 140
	case 4: return launch_solve_tiles_T<4, 3>(a, stream);  // 184 (168 with a few spills: faster than two waves per SIMD)
	case 5: return launch_solve_tiles_T<5, 2>(a, stream);  // 246
	case 6: return launch_solve_tiles_T<6, 2>(a, stream);  // 435 at one wave per SIMD; cut to 256 the spills cost less than the second wave gains
	case 7: return launch_solve_tiles_T<7, 1>(a, stream);  // 512, 18 spilled
	case 8: return launch_solve_tiles_T<8, 1>(a, stream);  // 512, 59 spilled
	default: return hipErrorInvalidValue;
	}
}

} // namespace anofox
