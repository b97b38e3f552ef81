// host_math.h — scalar Student-t quantile for the host-side helpers (anofox_t_critical,
// anofox_predict_with_interval).  Same formulas as device_math.h; the reference takes the value from
// statrs::StudentsT::inverse_cdf (crates/anofox-stats-ffi/src/lib.rs:2217-2231).
#pragma once
#include <math.h>

namespace anofox {
namespace hostmath {

inline double betacf(double a, double b, double x) {
	const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
	double am = 1.0, bm = 1.0, az = 1.0, bz = 1.0 - qab * x / qap;
	const int cap = 10000 + (int)(4.0 * sqrt(a > b ? a : b)); // O(sqrt(max(a, b))) terms near the mode
	for (int m = 1; m <= cap; ++m) {
		const double em = m, tem = em + em;
		double d = em * (b - em) * x / ((qam + tem) * (a + tem));
		const double ap = az + d * am, bp = bz + d * bm;
		d = -(a + em) * (qab + em) * x / ((a + tem) * (qap + tem));
		const double app = ap + d * az, bpp = bp + d * bz;
		const double aold = az;
		am = ap / bpp;
		bm = bp / bpp;
		az = app / bpp;
		bz = 1.0;
		if (fabs(az - aold) <= 4e-16 * fabs(az)) break;
	}
	return az;
}

inline double betainc(double a, double b, double x) {
	if (x <= 0.0) return 0.0;
	if (x >= 1.0) return 1.0;
	const double lbt = lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log1p(-x);
	if (x < (a + 1.0) / (a + b + 2.0)) return exp(lbt) * betacf(a, b, x) / a;
	return 1.0 - exp(lbt) * betacf(b, a, 1.0 - x) / b;
}

inline double t_upper(double t, double df) { return 0.5 * betainc(0.5 * df, 0.5, df / (df + t * t)); }

// quantile at prob in (0.5, 1)
inline double t_quantile_upper(double prob, double df) {
	const double tail = 1.0 - prob;
	if (df > 1e5) {
		// Cornish-Fisher series around the normal quantile (Abramowitz & Stegun 26.7.5), exact to < 1e-16 here;
		// z by bisection on the normal tail
		double zl = 0.0, zh = 40.0;
		for (int i = 0; i < 200; ++i) {
			const double zm = 0.5 * (zl + zh);
			if (0.5 * erfc(zm * M_SQRT1_2) > tail) zl = zm; else zh = zm;
		}
		const double z = 0.5 * (zl + zh), z2 = z * z, r = 1.0 / df;
		const double g1 = z * (z2 + 1.0) * 0.25;
		const double g2 = z * ((5.0 * z2 + 16.0) * z2 + 3.0) * (1.0 / 96.0);
		const double g3 = z * (((3.0 * z2 + 19.0) * z2 + 17.0) * z2 - 15.0) * (1.0 / 384.0);
		const double g4 = z * ((((79.0 * z2 + 776.0) * z2 + 1482.0) * z2 - 1920.0) * z2 - 945.0) * (1.0 / 92160.0);
		return z + r * (g1 + r * (g2 + r * (g3 + r * g4)));
	}
	double lo = 0.0, hi = 1.0;
	for (int i = 0; i < 1100 && t_upper(hi, df) > tail; ++i) { lo = hi; hi *= 2.0; }
	const double lnc = lgamma(0.5 * (df + 1.0)) - lgamma(0.5 * df) - 0.5 * log(df * M_PI);
	double t = 0.5 * (lo + hi);
	for (int it = 0; it < 200; ++it) {
		const double u = t_upper(t, df);
		if (u > tail) lo = t; else hi = t;
		const double pdf = exp(lnc - 0.5 * (df + 1.0) * log1p(t * t / df));
		double tn = t + (u - tail) / pdf;
		if (!(tn > lo && tn < hi)) tn = 0.5 * (lo + hi);
		if (fabs(tn - t) <= 1e-15 * fabs(tn)) return tn;
		t = tn;
	}
	return t;
}

} // namespace hostmath
} // namespace anofox
