// device_math.h — f64 distribution functions evaluated on the GPU by the per-group
// solve kernels: Student-t two-sided p-value, F survival function, Student-t quantile.
//
// The reference takes these from statrs 0.18 (StudentsT / FisherSnedecor; call sites
// crates/anofox-stats-ffi/src/lib.rs:26,2223-2230 and, through anofox-regression, the
// p_values / conf_interval_* / f_pvalue fields read at crates/anofox-stats-core/src/models/ols.rs:189-261).
// Here: regularised incomplete beta by the modified-Lentz continued fraction, which keeps
// full relative accuracy in the far tails (p-values of 1e-85 in
// tests/golden/inference_tests/expected/multiple_inference.json).
#pragma once
#include <hip/hip_runtime.h>

namespace anofox {

// Continued fraction of the incomplete beta function, evaluated with the forward (Wallis) recurrence on the
// convergents, renormalised every step: two divisions per iteration (the modified-Lentz form needs six, and
// f64 division is a ~15-instruction dependent chain on the VALU).
__device__ __forceinline__ double dm_betacf(double a, double b, double x) {
	const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
	double am = 1.0, bm = 1.0, az = 1.0;
	double bz = 1.0 - qab * x / qap;
	// the fraction needs O(sqrt(max(a, b))) terms near the mode (37 924 at a = 5e8: groups of 10^9 rows)
	const int cap = 3000 + (int)(4.0 * sqrt(fmax(a, b)));
	for (int m = 1; m <= cap; ++m) {
		const double em = (double)m;
		const double tem = em + em;
		const double a2 = a + tem;
		// d_even = em (b - em) x / ((qam + tem)(a + tem)),  d_odd = -(a + em)(qab + em) x / ((a + tem)(qap + tem))
		const double rden = 1.0 / ((qam + tem) * a2 * (qap + tem));
		const double de = em * (b - em) * x * (qap + tem) * rden;
		const double dod = -(a + em) * (qab + em) * x * (qam + tem) * rden;
		const double ap = az + de * am;
		const double bp = bz + de * bm;
		const double app = ap + dod * az;
		const double bpp = bp + dod * bz;
		const double r = 1.0 / bpp;
		const double aold = az;
		am = ap * r;
		bm = bp * r;
		az = app * r;
		bz = 1.0;
		if (fabs(az - aold) <= 4e-16 * fabs(az)) break;
	}
	return az;
}

// ln Gamma(a + b) - ln Gamma(a).  For huge a (df / 2 of a group with millions of rows) the difference of two
// lgamma values of size a ln a cancels ~9 digits; there the Stirling series in 1 / a is used instead:
//   b ln a + sum_k (-1)^(k+1) (B_{k+1}(b) - B_{k+1}) / (k (k+1) a^k)      (Bernoulli polynomials)
__device__ __forceinline__ double dm_lgamma_diff(double a, double b) {
	if (!(a > 1e6 && b * b < 1e-2 * a)) return lgamma(a + b) - lgamma(a);
	const double r = 1.0 / a, b2 = b * b;
	const double c1 = (b2 - b) * 0.5;
	const double c2 = -((b2 - 1.5 * b + 0.5) * b) * (1.0 / 6.0);
	const double c3 = ((b2 - 2.0 * b + 1.0) * b2) * (1.0 / 12.0);
	const double c4 = -((((b - 2.5) * b + 5.0 / 3.0) * b2 - 1.0 / 6.0) * b) * (1.0 / 20.0);
	const double c5 = ((((b - 3.0) * b + 2.5) * b2 - 0.5) * b2) * (1.0 / 30.0);
	return b * log(a) + r * (c1 + r * (c2 + r * (c3 + r * (c4 + r * c5))));
}

// I_x(a, b)
static __device__ __attribute__((noinline)) double dm_betainc(double a, double b, double x) {
	if (isnan(a) || isnan(b) || isnan(x)) return __builtin_nan("");
	if (x <= 0.0) return 0.0;
	if (x >= 1.0) return 1.0;
	const double lbt = dm_lgamma_diff(a, b) - lgamma(b) + a * log(x) + b * log1p(-x);
	if (x < (a + 1.0) / (a + b + 2.0)) return exp(lbt) * dm_betacf(a, b, x) / a;
	return 1.0 - exp(lbt) * dm_betacf(b, a, 1.0 - x) / b;
}

// 2 P(T_df > |t|)
__device__ __forceinline__ double dm_t_two_sided_p(double t, double df) {
	if (isnan(t) || !(df > 0.0)) return __builtin_nan("");
	if (isinf(t)) return 0.0;
	return dm_betainc(0.5 * df, 0.5, df / (df + t * t));
}

// P(F_{d1,d2} > f)
__device__ __forceinline__ double dm_f_sf(double f, double d1, double d2) {
	if (isnan(f) || !(d1 > 0.0) || !(d2 > 0.0)) return __builtin_nan("");
	if (f <= 0.0) return 1.0;
	if (isinf(f)) return 0.0;
	return dm_betainc(0.5 * d2, 0.5 * d1, d2 / (d2 + d1 * f));
}

// upper-tail probability P(T_df > t) for t >= 0
__device__ __forceinline__ double dm_t_upper(double t, double df) {
	return 0.5 * dm_betainc(0.5 * df, 0.5, df / (df + t * t));
}

// Student-t quantile for prob in (0.5, 1), robust version: bracketing + safeguarded Newton on the upper tail.
static __device__ __attribute__((noinline)) double dm_t_quantile_upper_slow(double prob, double df) {
	const double tail = 1.0 - prob; // target upper-tail mass
	double lo = 0.0, hi = 1.0;
	for (int i = 0; i < 1100 && dm_t_upper(hi, df) > tail; ++i) {
		lo = hi;
		hi *= 2.0;
	}
	const double lnc = lgamma(0.5 * (df + 1.0)) - lgamma(0.5 * df) - 0.5 * log(df * 3.14159265358979323846);
	double t = 0.5 * (lo + hi);
	for (int it = 0; it < 100; ++it) {
		const double u = dm_t_upper(t, df);
		if (u > tail) lo = t;
		else hi = t;
		const double pdf = exp(lnc - 0.5 * (df + 1.0) * log1p(t * t / df));
		double tn = t + (u - tail) / pdf; // d(upper)/dt = -pdf
		if (!(tn > lo && tn < hi)) tn = 0.5 * (lo + hi);
		if (fabs(tn - t) <= 1e-15 * fabs(tn)) {
			t = tn;
			break;
		}
		t = tn;
	}
	return t;
}

// Student-t quantile for prob in (0.5, 1): Cornish-Fisher start from the normal quantile
// (Abramowitz & Stegun 26.7.5), polished by Newton steps on the exact tail; closed forms for df = 1, 2.
static __device__ __attribute__((noinline)) double dm_t_quantile_upper(double prob, double df) {
	if (!(prob > 0.5 && prob < 1.0) || !(df > 0.0)) {
		if (prob == 0.5) return 0.0;
		return __builtin_nan("");
	}
	if (df == 1.0) return tan(3.14159265358979323846 * (prob - 0.5));
	if (df == 2.0) return (2.0 * prob - 1.0) / sqrt(2.0 * prob * (1.0 - prob));
	const double tail = 1.0 - prob;
	const double z = normcdfinv(prob);
	const double z2 = z * z;
	const double r = 1.0 / df;
	const double g1 = z * (z2 + 1.0) * 0.25;
	const double g2 = z * ((5.0 * z2 + 16.0) * z2 + 3.0) * (1.0 / 96.0);
	const double g3 = z * (((3.0 * z2 + 19.0) * z2 + 17.0) * z2 - 15.0) * (1.0 / 384.0);
	const double g4 = z * ((((79.0 * z2 + 776.0) * z2 + 1482.0) * z2 - 1920.0) * z2 - 945.0) * (1.0 / 92160.0);
	double t = z + r * (g1 + r * (g2 + r * (g3 + r * g4)));
	if (df > 1e5 && t > 0.0) return t; // the series is exact to < 1e-16 here; the incomplete beta costs O(sqrt(df)) terms
	if (!(t > 0.0)) return dm_t_quantile_upper_slow(prob, df);
	const double lnc = lgamma(0.5 * (df + 1.0)) - lgamma(0.5 * df) - 0.5 * log(df * 3.14159265358979323846);
	for (int it = 0; it < 12; ++it) {
		const double u = dm_t_upper(t, df);
		const double pdf = exp(lnc - 0.5 * (df + 1.0) * log1p(t * t * r));
		const double tn = t + (u - tail) / pdf;
		if (!(tn > 0.0) || !isfinite(tn)) return dm_t_quantile_upper_slow(prob, df);
		const bool done = fabs(tn - t) <= 1e-14 * fabs(tn);
		t = tn;
		if (done) return t;
	}
	return dm_t_quantile_upper_slow(prob, df);
}

// The critical value depends only on (confidence level, df): groups of one batch usually share df, so the
// value is memoised in a small open-addressed table in device memory (zeroed at the start of every batch call).
// A slot is written once: it is claimed by a compare-and-swap of its key from 0 to kTcritBusy, the value is
// stored, then the key is published with release order; readers that find their key (acquire) may read the
// value.  A df that finds its probe window taken by other keys is simply computed every time.
struct TcritSlot {
	unsigned long long key; // bits of df (0 = empty, 1 = being written)
	double value;
};
constexpr int kTcritSlots = 256; // kTcritTableBytes / sizeof(TcritSlot)
constexpr int kTcritProbes = 4;
constexpr unsigned long long kTcritBusy = 1ull;

static __device__ __forceinline__ double dm_tcrit_cached(TcritSlot *table, double prob, double df) {
	if (!(df > 0.0)) return __builtin_nan("");
	const unsigned long long key = (unsigned long long)__double_as_longlong(df);
	const int home = (int)((key * 0x9E3779B97F4A7C15ull) >> 56); // top 8 bits
	TcritSlot *empty = nullptr;
	for (int i = 0; i < kTcritProbes; ++i) {
		TcritSlot *slot = table + ((home + i) & (kTcritSlots - 1));
		const unsigned long long seen = __hip_atomic_load(&slot->key, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
		if (seen == key) return __hip_atomic_load(&slot->value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (seen == 0ull) { empty = slot; break; }
	}
	const double t = dm_t_quantile_upper(prob, df);
	// one lane per wavefront publishes, and only into a slot that is still empty: a batch whose groups share df
	// would otherwise start with ~10^5 lanes queueing on the same compare-and-swap
	const unsigned long long missed = __ballot(1);
	const bool leader = (int)__lane_id() == __ffsll((unsigned long long)missed) - 1;
	if (empty && leader && __hip_atomic_load(&empty->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) {
		unsigned long long expect = 0ull;
		if (__hip_atomic_compare_exchange_strong(&empty->key, &expect, kTcritBusy, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
		                                         __HIP_MEMORY_SCOPE_AGENT)) {
			__hip_atomic_store(&empty->value, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&empty->key, key, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	return t;
}

} // namespace anofox
