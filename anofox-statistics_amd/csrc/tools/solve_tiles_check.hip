// solve_tiles_check.hip — unit check of solve_tiles_kernel against a host long-double Cholesky on synthetic records.
//   solve_tiles_check <T> <groups> [corr]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <random>

#include "../solve_tiles_impl.h"

using namespace anofox;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#ifndef STC_WPE
#define STC_WPE 1
#endif
template <int T>
int run(int G, double corr) {
	constexpr int WPE_ = STC_WPE;
	const int P16 = 16 * T, p = P16 - (T == 8 ? 0 : 3), NT = T * (T + 1) / 2;
	const int reclen = wide_record_len(T);
	const int n = 400;
	std::mt19937_64 rng(7);
	std::uniform_real_distribution<double> U(-10, 10);
	std::normal_distribution<double> N01(0, 1);
	std::vector<double> rec((size_t)G * reclen, 0.0);
	std::vector<long double> refb((size_t)G * p), refrss(G);
	std::vector<int64_t> offs(G + 1);
	for (int g = 0; g <= G; ++g) offs[g] = (int64_t)g * n;
	for (int g = 0; g < G; ++g) {
		std::vector<double> X((size_t)n * p), y(n), beta(p);
		for (auto &b : beta) b = U(rng) * 0.5;
		for (int i = 0; i < n; ++i) {
			double common = U(rng);
			double acc = U(rng);
			for (int j = 0; j < p; ++j) { X[(size_t)i * p + j] = (1 - corr) * U(rng) + corr * common; acc += beta[j] * X[(size_t)i * p + j]; }
			y[i] = acc + 2.0 * N01(rng);
		}
		// shifted moments (shift = first row), as accumulate_wide writes them
		double *r = rec.data() + (size_t)g * reclen;
		double *vec = r + (size_t)NT * 256;
		std::vector<double> d((size_t)n * P16, 0.0), dy(n);
		for (int i = 0; i < n; ++i) { for (int j = 0; j < p; ++j) d[(size_t)i * P16 + j] = X[(size_t)i * p + j] - X[j]; dy[i] = y[i] - y[0]; }
		for (int I = 0; I < T; ++I) for (int J = I; J < T; ++J) {
			double *tp = r + (size_t)(I * T - I * (I - 1) / 2 + (J - I)) * 256;
			for (int a = 0; a < 16; ++a) for (int b = 0; b < 16; ++b) {
				long double s = 0; for (int i = 0; i < n; ++i) s += (long double)d[(size_t)i * P16 + 16 * I + a] * d[(size_t)i * P16 + 16 * J + b];
				tp[a * 16 + b] = (double)s;
			}
		}
		long double sy = 0, syy = 0;
		for (int i = 0; i < n; ++i) { sy += dy[i]; syy += (long double)dy[i] * dy[i]; }
		for (int j = 0; j < P16; ++j) {
			long double sx = 0, sxy = 0; for (int i = 0; i < n; ++i) { sx += d[(size_t)i * P16 + j]; sxy += (long double)d[(size_t)i * P16 + j] * dy[i]; }
			vec[0 * P16 + j] = (double)sx; vec[1 * P16 + j] = (double)sxy; vec[2 * P16 + j] = j < p ? X[j] : 0.0; vec[3 * P16 + j] = j < p ? 1.0 : 0.0;
		}
		double *sc = vec + 4 * P16; sc[0] = (double)sy; sc[1] = (double)syy; sc[2] = n; sc[3] = n; sc[4] = y[0];
		// reference: long double normal equations on the centred data
		std::vector<long double> S((size_t)p * p), c(p), mx(p, 0); long double my = 0;
		for (int i = 0; i < n; ++i) { for (int j = 0; j < p; ++j) mx[j] += X[(size_t)i * p + j]; my += y[i]; }
		for (auto &m : mx) m /= n; my /= n;
		for (int a = 0; a < p; ++a) { for (int b = 0; b <= a; ++b) { long double s = 0; for (int i = 0; i < n; ++i) s += (X[(size_t)i * p + a] - mx[a]) * (X[(size_t)i * p + b] - mx[b]); S[(size_t)a * p + b] = S[(size_t)b * p + a] = s; }
			long double s = 0; for (int i = 0; i < n; ++i) s += (X[(size_t)i * p + a] - mx[a]) * (y[i] - my); c[a] = s; }
		for (int j = 0; j < p; ++j) { for (int k = 0; k < j; ++k) { } }
		std::vector<long double> L(S);
		for (int j = 0; j < p; ++j) { long double dd = L[(size_t)j * p + j]; for (int k = 0; k < j; ++k) dd -= L[(size_t)j * p + k] * L[(size_t)j * p + k]; dd = sqrtl(dd); L[(size_t)j * p + j] = dd;
			for (int i = j + 1; i < p; ++i) { long double s = L[(size_t)i * p + j]; for (int k = 0; k < j; ++k) s -= L[(size_t)i * p + k] * L[(size_t)j * p + k]; L[(size_t)i * p + j] = s / dd; } }
		std::vector<long double> z(p), b(p);
		for (int i = 0; i < p; ++i) { long double s = c[i]; for (int k = 0; k < i; ++k) s -= L[(size_t)i * p + k] * z[k]; z[i] = s / L[(size_t)i * p + i]; }
		for (int i = p - 1; i >= 0; --i) { long double s = z[i]; for (int k = i + 1; k < p; ++k) s -= L[(size_t)k * p + i] * b[k]; b[i] = s / L[(size_t)i * p + i]; }
		long double rss = 0; for (int i = 0; i < n; ++i) { long double e = y[i] - my; for (int j = 0; j < p; ++j) e -= b[j] * (X[(size_t)i * p + j] - mx[j]); rss += e * e; }
		for (int j = 0; j < p; ++j) refb[(size_t)g * p + j] = b[j];
		refrss[g] = rss;
	}
	WideArgs a; memset(&a, 0, sizeof a);
	double *d_rec, *d_core, *d_inf; int64_t *d_off; int32_t *d_list, *d_cnt;
	CHECK(hipMalloc(&d_rec, rec.size() * 8)); CHECK(hipMemcpy(d_rec, rec.data(), rec.size() * 8, hipMemcpyHostToDevice));
	CHECK(hipMalloc(&d_core, (size_t)G * (p + 6) * 8)); CHECK(hipMalloc(&d_inf, (size_t)G * (5 * p + 2) * 8));
	CHECK(hipMalloc(&d_off, (G + 1) * 8)); CHECK(hipMemcpy(d_off, offs.data(), (G + 1) * 8, hipMemcpyHostToDevice));
	CHECK(hipMalloc(&d_list, G * 4)); CHECK(hipMalloc(&d_cnt, 64)); CHECK(hipMemset(d_cnt, 0, 64));
	a.row_offsets = d_off; a.group_base = 0; a.n_groups = G; a.p = p; a.model = ANOFOX_HIP_MODEL_OLS; a.fit_intercept = 1; a.compute_inference = 1;
	a.confidence_level = 0.95; a.alpha = 0; a.moments = d_rec; a.core = d_core; a.inference = d_inf; a.refine_list = d_list; a.refine_count = d_cnt;
	hipLaunchKernelGGL((tiles::solve_tiles_kernel<T, WPE_>), dim3(G), dim3(64), 0, 0, a);
	CHECK(hipDeviceSynchronize());
	std::vector<double> core((size_t)G * (p + 6));
	CHECK(hipMemcpy(core.data(), d_core, core.size() * 8, hipMemcpyDeviceToHost));
	int cnt = 0; CHECK(hipMemcpy(&cnt, d_cnt, 4, hipMemcpyDeviceToHost));
	double worst_b = 0, worst_s = 0;
	for (int g = 0; g < G; ++g) {
		long double sc = 0; for (int j = 0; j < p; ++j) sc = fmaxl(sc, fabsl(refb[(size_t)g * p + j]));
		for (int j = 0; j < p; ++j) worst_b = fmax(worst_b, (double)(fabsl(core[(size_t)g * (p + 6) + j] - refb[(size_t)g * p + j]) / sc));
		const double sig = sqrt((double)refrss[g] / (n - p - 1));
		worst_s = fmax(worst_s, fabs(core[(size_t)g * (p + 6) + p + 3] / sig - 1.0));
	}
	if (getenv("STC_BENCH")) { // timing: the same records replicated to STC_BENCH groups
		const int GB = atoi(getenv("STC_BENCH"));
		double *d_big, *d_core2, *d_inf2; int64_t *d_off2;
		CHECK(hipMalloc(&d_big, (size_t)GB * reclen * 8)); CHECK(hipMalloc(&d_core2, (size_t)GB * (p + 6) * 8)); CHECK(hipMalloc(&d_inf2, (size_t)GB * (5 * p + 2) * 8));
		CHECK(hipMalloc(&d_off2, ((size_t)GB + 1) * 8));
		std::vector<int64_t> o2(GB + 1); for (int g = 0; g <= GB; ++g) o2[g] = (int64_t)g * n;
		CHECK(hipMemcpy(d_off2, o2.data(), o2.size() * 8, hipMemcpyHostToDevice));
		for (int g = 0; g < GB; ++g) CHECK(hipMemcpyAsync(d_big + (size_t)g * reclen, d_rec + (size_t)(g % G) * reclen, (size_t)reclen * 8, hipMemcpyDeviceToDevice, 0));
		WideArgs b = a; b.moments = d_big; b.core = d_core2; b.inference = d_inf2; b.row_offsets = d_off2; b.n_groups = GB;
		hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((tiles::solve_tiles_kernel<T, WPE_>), dim3(GB), dim3(64), 0, 0, b);
		CHECK(hipEventRecord(e0, 0));
		for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((tiles::solve_tiles_kernel<T, WPE_>), dim3(GB), dim3(64), 0, 0, b);
		CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize());
		float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
		printf("T=%d WPE=%d: %d groups in %.3f ms per launch\n", T, WPE_, GB, ms / 5);
	}
	printf("T=%d p=%d G=%d corr=%.2f: max coef err (rel to max|b|) %.3e, sigma rel err %.3e, status0 %.0f, queued %d\n", T, p, G, corr, worst_b, worst_s, core[p + 5], cnt);
	return 0;
}

int main(int argc, char **argv) {
	const int T = argc > 1 ? atoi(argv[1]) : 3;
	const int G = argc > 2 ? atoi(argv[2]) : 8;
	const double corr = argc > 3 ? atof(argv[3]) : 0.0;
	switch (T) {
	case 1: return run<1>(G, corr);
	case 2: return run<2>(G, corr);
	case 3: return run<3>(G, corr);
	case 4: return run<4>(G, corr);
	case 5: return run<5>(G, corr);
	case 6: return run<6>(G, corr);
	case 7: return run<7>(G, corr);
	default: return run<8>(G, corr);
	}
}
