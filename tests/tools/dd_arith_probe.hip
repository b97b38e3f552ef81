// dd_arith_probe.hip — device-code probe for csrc/dd_arith.h (tests/test_sanitizers_cpu.py::test_dd_arith_device_code):
// cross-compiled to gfx950 assembly (no GPU needed), the loop body of this kernel must contain exactly ONE v_fma_f64 —
// two_prod's — per term.  Built with -DDD_UNGUARDED (the formulas without `#pragma clang fp contract(off)`, the state of the
// code before the fix) hipcc fuses the product into two_sum's additions and the count goes up: the defect the guard removes.
#include <hip/hip_runtime.h>

#ifdef DD_UNGUARDED
namespace anofox {
__device__ __forceinline__ void two_sum(double a, double b, double &s, double &e) { s = a + b; const double bb = s - a; e = (a - (s - bb)) + (b - bb); }
__device__ __forceinline__ void two_prod(double a, double b, double &p, double &e) { p = a * b; e = fma(a, b, -p); }
__device__ __forceinline__ void dd_fit_term(double &fh, double &fl, double b, double x) {
	double ph, pl, sh, sl;
	two_prod(b, x, ph, pl);
	two_sum(fh, ph, sh, sl);
	fh = sh;
	fl += pl + sl;
}
}
#else
#include "dd_arith.h"
#endif

extern "C" __global__ void dd_probe(const double *b, const double *x, double *out, int p) {
	double fh = out[0], fl = 0.0;
#pragma unroll 1
	for (int j = 0; j < p; ++j) anofox::dd_fit_term(fh, fl, b[j], x[j]);
	out[0] = fh;
	out[1] = fl;
}
