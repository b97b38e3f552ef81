// dd_arith.h — error-free transformations and double-double accumulation for the residual passes of the refinement
// (solve_narrow.hip: residual_grad_wave; solve_wide.hip: residual_grad_wide_kernel).  Round-to-nearest, no fast-math.
//
// hipcc's default -ffp-contract=fast-honor-pragmas fuses a product with the sum that consumes it ACROSS statements; inside
// two_sum that turns s = a + RN(x b) into fma(x, b, a), whose rounding error the rest of two_sum does not recover, and a
// "compensated" residual then carries working-precision noise (r3: a 15 x 15 system of the deep fuzz sweep wandered between
// 4e-10 and 1.4e-8 whatever the number of updates, and converges to 3e-12 in two without the fusion).  Every function here
// switches contraction off for its own body — an add without the `contract` flag is not fused with a product from the
// caller — and the kernels that call them do the same for theirs; the fma() calls are the only fused operations.
#pragma once
#include <hip/hip_runtime.h>

namespace anofox {

// s + e = a + b exactly (Knuth)
__host__ __device__ __forceinline__ void two_sum(double a, double b, double &s, double &e) {
#pragma clang fp contract(off)
	s = a + b;
	const double bb = s - a;
	e = (a - (s - bb)) + (b - bb);
}

// p + e = a b exactly
__host__ __device__ __forceinline__ void two_prod(double a, double b, double &p, double &e) {
#pragma clang fp contract(off)
	p = a * b;
	e = fma(a, b, -p);
}

// (hi, lo) += (ahi, alo), kept as an unevaluated sum with |lo| <= ulp(hi)
__host__ __device__ __forceinline__ void dd_add(double &hi, double &lo, double ahi, double alo) {
#pragma clang fp contract(off)
	double s, e;
	two_sum(hi, ahi, s, e);
	e += lo + alo;
	hi = s + e;
	lo = e - (hi - s);
}

// (hi, lo) += (wh + wl) (x - shift), the difference taken exactly
__host__ __device__ __forceinline__ void dd_add_scaled_diff(double &hi, double &lo, double wh, double wl, double x, double shift) {
#pragma clang fp contract(off)
	double dh, dl, ph, pl;
	two_sum(x, -shift, dh, dl);
	two_prod(wh, dh, ph, pl);
	pl = fma(wl, dh, pl);
	pl = fma(wh, dl, pl);
	dd_add(hi, lo, ph, pl);
}

// One row's residual in double-double: (e, e_l) = y - b0 - sum_j b[j] x[j]; fit accumulated as (fh, fl) by the caller
// through dd_fit_term.
__host__ __device__ __forceinline__ void dd_fit_term(double &fh, double &fl, double b, double x) {
#pragma clang fp contract(off)
	double ph, pl, sh, sl;
	two_prod(b, x, ph, pl);
	two_sum(fh, ph, sh, sl);
	fh = sh;
	fl += pl + sl;
}

// (wh, wl) = w (y - (fh + fl)) to twice the working precision; e = the residual rounded to working precision
__host__ __device__ __forceinline__ void dd_weighted_residual(double y, double fh, double fl, double w, double &e, double &wh, double &wl) {
#pragma clang fp contract(off)
	double eh, el;
	two_sum(y, -fh, eh, el);
	el -= fl;
	e = eh + el;
	const double e_l = el - (e - eh);
	double ph, pl;
	two_prod(w, e, ph, pl);
	pl = fma(w, e_l, pl);
	wh = ph + pl;
	wl = pl - (wh - ph);
}

// ---- a double-double number and its arithmetic (refit_dd.hip): ~32 significant digits, round-to-nearest, no contraction ----
struct dd {
	double h, l;
};
__host__ __device__ __forceinline__ dd dd_make(double h, double l = 0.0) { return dd{h, l}; }
__host__ __device__ __forceinline__ dd dd_renorm(double s, double e) {
#pragma clang fp contract(off)
	const double h = s + e;
	return dd{h, e - (h - s)};
}
__host__ __device__ __forceinline__ dd operator+(dd a, dd b) {
#pragma clang fp contract(off)
	const double s = a.h + b.h;
	const double bb = s - a.h;
	double e = (a.h - (s - bb)) + (b.h - bb);
	e += a.l + b.l;
	return dd_renorm(s, e);
}
__host__ __device__ __forceinline__ dd operator-(dd a) { return dd{-a.h, -a.l}; }
__host__ __device__ __forceinline__ dd operator-(dd a, dd b) { return a + (-b); }
__host__ __device__ __forceinline__ dd operator*(dd a, dd b) {
#pragma clang fp contract(off)
	const double p = a.h * b.h;
	double e = fma(a.h, b.h, -p);
	e = fma(a.h, b.l, e);
	e = fma(a.l, b.h, e);
	return dd_renorm(p, e);
}
__host__ __device__ __forceinline__ dd dd_mul_d(dd a, double b) {
#pragma clang fp contract(off)
	const double p = a.h * b;
	double e = fma(a.h, b, -p);
	e = fma(a.l, b, e);
	return dd_renorm(p, e);
}
__host__ __device__ __forceinline__ dd dd_prod(double a, double b) { // a b exactly
	const double p = a * b;
	return dd{p, fma(a, b, -p)};
}
__host__ __device__ __forceinline__ dd operator/(dd a, dd b) {
#pragma clang fp contract(off)
	const double q1 = a.h / b.h;
	dd r = a - dd_mul_d(b, q1);
	const double q2 = r.h / b.h;
	r = r - dd_mul_d(b, q2);
	const double q3 = r.h / b.h;
	return dd_renorm(q1, q2) + dd_make(q3);
}
__host__ __device__ __forceinline__ dd dd_sqrt(dd a) {
#pragma clang fp contract(off)
	if (!(a.h > 0.0)) return dd{a.h == 0.0 ? 0.0 : __builtin_nan(""), 0.0};
	const double x = sqrt(a.h);
	const dd r = a - dd_prod(x, x);
	return dd_renorm(x, r.h / (2.0 * x));
}
__host__ __device__ __forceinline__ double dd_to_double(dd a) { return a.h + a.l; }


} // namespace anofox
