// dd_arith_check.hip — CPU check of csrc/dd_arith.h (tests/test_sanitizers_cpu.py::test_dd_arith_host).
// Compiled by hipcc for the HOST with FMA available (-mfma) and hipcc's default -ffp-contract=fast-honor-pragmas: the
// error-free transformations must stay exact although the compiler is allowed to fuse a product with the sum that
// consumes it everywhere else.  The check replays one row of the refinement's residual pass — fit = b0 + sum b_j x_j,
// e = y - fit, g += e (x_j - shift_j) — and compares with __float128 arithmetic (113-bit significand: exact for these
// operand ranges up to a relative 1e-33).  With -DDD_UNGUARDED the same formulas are compiled WITHOUT the guard (the
// state of the code before the fix): that build is expected to fail the check, which shows the check sees the defect.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#ifdef DD_UNGUARDED
namespace anofox {
static inline void two_sum(double a, double b, double &s, double &e) { s = a + b; const double bb = s - a; e = (a - (s - bb)) + (b - bb); }
static inline void two_prod(double a, double b, double &p, double &e) { p = a * b; e = fma(a, b, -p); }
static inline void dd_fit_term(double &fh, double &fl, double b, double x) {
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

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static double uniform() { // xorshift64*, (0, 1)
	rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
	return ((rng_state * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0) + 1e-17;
}

int main() {
	const int p = 14, trials = 20000;
	double worst = 0.0;
	for (int t = 0; t < trials; ++t) {
		double b[p], x[p];
		__float128 exact = 0;
		for (int j = 0; j < p; ++j) {
			b[j] = (uniform() - 0.5) * 200.0;
			x[j] = (uniform() - 0.5) * 20.0;
		}
		// y close to the fit: the residual is the small difference of large terms, as near an exact fit
		double fh = 3.0 * (uniform() - 0.5), fl = 0.0;
		exact = (__float128)fh;
		for (int j = 0; j < p; ++j) {
			anofox::dd_fit_term(fh, fl, b[j], x[j]);
			exact += (__float128)b[j] * (__float128)x[j];
		}
		const __float128 got = (__float128)fh + (__float128)fl;
		__float128 d = got - exact;
		if (d < 0) d = -d;
		__float128 scale = exact < 0 ? -exact : exact;
		if (scale < 1) scale = 1;
		const double rel = (double)(d / scale);
		if (rel > worst) worst = rel;
	}
	printf("dd_fit_term: worst relative error of (fh + fl) against __float128 over %d rows of %d terms: %.3e\n", trials, p, worst);
	// a double-double sum of 14 products is exact to ~ p * 2^-106 = 2e-31; working precision would be 1e-16
	if (!(worst < 1e-28)) {
		printf("FAILED: the compensated dot product lost its low part (FMA contraction inside two_sum?)\n");
		return 1;
	}
#ifndef DD_UNGUARDED
	// the double-double type of refit_dd.hip: sums, products, quotients and square roots against __float128
	{
		using anofox::dd;
		auto rnd = [&]() { // a double-double of magnitude 1e-8 .. 1e8 with a full low part
			const double h = (uniform() - 0.5) * pow(10.0, 16.0 * uniform() - 8.0);
			const double l = h * (uniform() - 0.5) * 1e-16;
			return anofox::dd_renorm(h, l);
		};
		auto q = [](dd a) { return (__float128)a.h + (__float128)a.l; };
		auto rel = [](__float128 got, __float128 want) {
			__float128 d = got - want;
			if (d < 0) d = -d;
			__float128 sc = want < 0 ? -want : want;
			return (double)(d / sc);
		};
		double w_add = 0, w_mul = 0, w_div = 0, w_sqrt = 0, w_chain = 0;
		for (int t = 0; t < 200000; ++t) {
			const dd a = rnd(), b = rnd();
			// (a sum may cancel: measure it against the larger operand)
			{
				__float128 want = q(a) + q(b), got = q(a + b), d = got - want;
				if (d < 0) d = -d;
				__float128 sc = q(a) < 0 ? -q(a) : q(a), sb = q(b) < 0 ? -q(b) : q(b);
				if (sb > sc) sc = sb;
				const double r = (double)(d / sc);
				if (r > w_add) w_add = r;
			}
			w_mul = fmax(w_mul, rel(q(a * b), q(a) * q(b)));
			w_div = fmax(w_div, rel(q(a / b), q(a) / q(b)));
			const dd aa = a.h < 0 ? -a : a;
			const dd s = anofox::dd_sqrt(aa);
			w_sqrt = fmax(w_sqrt, rel(q(s) * q(s), q(aa)));
		}
		// a Cholesky-like chain: c - sum l_k^2 with 100 terms, the shape of a pivot
		for (int t = 0; t < 2000; ++t) {
			dd c = anofox::dd_make(0.0);
			__float128 want = 0;
			for (int k = 0; k < 100; ++k) {
				const dd lk = rnd();
				c = c + lk * lk;
				want += q(lk) * q(lk);
			}
			w_chain = fmax(w_chain, rel(q(c), want));
		}
		printf("dd type: worst relative errors  add %.2e  mul %.2e  div %.2e  sqrt %.2e  sum of 100 squares %.2e\n", w_add, w_mul, w_div, w_sqrt, w_chain);
		if (!(w_add < 1e-30 && w_mul < 1e-30 && w_div < 1e-29 && w_sqrt < 1e-29 && w_chain < 1e-29)) {
			printf("FAILED: double-double arithmetic below twice the working precision\n");
			return 1;
		}
	}
#endif
	printf("ok\n");
	return 0;
}
