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
	for (int m = 1; m <= 10000; ++m) {
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
