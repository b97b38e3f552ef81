// native_bench.hip — torch-free driver of the C ABI for profiling (rocprofv3 --pmc) and as a minimal
// example of a native caller: fills device-resident grouped columns with a counter-based generator of the
// benchmark's distribution, then calls anofox_hip_fit_batch_device `steps` times.
//   native_bench <groups> <rows_per_group> <features> <ols|ridge|wls> <steps> [inference]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "../../../include/anofox_stats_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ inline unsigned long long mix64(unsigned long long z) {
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
__device__ inline double u01(unsigned long long h) { return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

struct Cols { double *x[128]; double *y; double *w; };

__global__ void fill_kernel(Cols c, int p, long long n_per, long long n_rows) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_rows) return;
	const long long g = i / n_per;
	const unsigned long long gs = mix64(42ull + 0xD1B54A32D192ED03ull * (unsigned long long)g);
	double acc = u01(mix64(gs ^ 0xFFFFull)) * 20.0 - 10.0;
	for (int j = 0; j < p; ++j) {
		const double bj = u01(mix64(gs ^ (0xFFFF0000ull + j))) * 10.0 - 5.0;
		const double xv = u01(mix64(gs ^ ((unsigned long long)i * 0xAEF17502108EF2D9ull + 1 + j))) * 20.0 - 10.0;
		c.x[j][i] = xv;
		acc += bj * xv;
	}
	const double u1 = u01(mix64(gs ^ ((unsigned long long)i * 0xAEF17502108EF2D9ull + 100)));
	const double u2 = u01(mix64(gs ^ ((unsigned long long)i * 0xAEF17502108EF2D9ull + 101)));
	c.y[i] = acc + 2.0 * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
	if (c.w) c.w[i] = u01(mix64(gs ^ ((unsigned long long)i * 0xAEF17502108EF2D9ull + 102))) + 0.5;
}

__global__ void offsets_kernel(long long *off, long long G, long long n_per) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i <= G) off[i] = i * n_per;
}

int main(int argc, char **argv) {
	const long long G = argc > 1 ? atoll(argv[1]) : 100000;
	const long long n = argc > 2 ? atoll(argv[2]) : 1000;
	const int p = argc > 3 ? atoi(argv[3]) : 8;
	const char *model = argc > 4 ? argv[4] : "ols";
	const int steps = argc > 5 ? atoi(argv[5]) : 5;
	const bool inference = argc > 6 && (!strcmp(argv[6], "inference") || !strncmp(argv[6], "hc", 2));
	const int hc = (argc > 6 && !strncmp(argv[6], "hc", 2)) ? 1 + atoi(argv[6] + 2) : 0; // hc0..hc3
	if (p < 1 || p > 128) { fprintf(stderr, "features must be 1..128\n"); return 2; }
	const long long N = G * n;
	const bool weighted = !strcmp(model, "wls");

	Cols c;
	memset(&c, 0, sizeof c);
	for (int j = 0; j < p; ++j) CHECK(hipMalloc(&c.x[j], N * sizeof(double)));
	CHECK(hipMalloc(&c.y, N * sizeof(double)));
	if (weighted) CHECK(hipMalloc(&c.w, N * sizeof(double)));
	long long *off = nullptr;
	CHECK(hipMalloc(&off, (G + 1) * sizeof(long long)));
	hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, c, p, n, N);
	hipLaunchKernelGGL(offsets_kernel, dim3((unsigned)((G + 256) / 256)), dim3(256), 0, 0, off, G, n);
	CHECK(hipDeviceSynchronize());

	double *core = nullptr, *inf = nullptr;
	CHECK(hipMalloc(&core, G * (p + 6) * sizeof(double)));
	if (inference) CHECK(hipMalloc(&inf, G * (5 * p + 2) * sizeof(double)));

	AnofoxError err;
	AnofoxHipContext *ctx = nullptr;
	if (!anofox_hip_context_create(-1, &ctx, &err)) { fprintf(stderr, "context: %s\n", err.message); return 1; }
	AnofoxHipBatchOptions opt;
	memset(&opt, 0, sizeof opt);
	opt.model = weighted ? ANOFOX_HIP_MODEL_WLS : (!strcmp(model, "ridge") ? ANOFOX_HIP_MODEL_RIDGE : ANOFOX_HIP_MODEL_OLS);
	opt.fit_intercept = !(getenv("NB_NO_INTERCEPT") && atoi(getenv("NB_NO_INTERCEPT"))); // NB_NO_INTERCEPT=1: regression through the origin
	opt.compute_inference = inference;
	opt.confidence_level = 0.95;
	opt.alpha = 1.0;
	opt.solver = ANOFOX_SOLVER_SVD;
	opt.hc_type = (AnofoxHcType)hc;
	const double *xc[128];
	for (int j = 0; j < p; ++j) xc[j] = c.x[j];

	auto run = [&]() {
		if (!anofox_hip_fit_batch_device(ctx, G, (size_t)p, N, (const int64_t *)off, c.y, xc, c.w, opt, core, inf, &err)) {
			fprintf(stderr, "fit: %s\n", err.message);
			exit(1);
		}
	};
	run();
	anofox_hip_context_synchronize(ctx, &err);
	anofox_hip_context_enable_timing(ctx, true, &err);
	const auto t0 = std::chrono::steady_clock::now();
	for (int s = 0; s < steps; ++s) run();
	anofox_hip_context_synchronize(ctx, &err);
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	AnofoxHipKernelTimes kt;
	anofox_hip_context_collect_timing(ctx, &kt, &err);

	std::vector<double> h((size_t)(p + 6));
	CHECK(hipMemcpy(h.data(), core, h.size() * sizeof(double), hipMemcpyDeviceToHost));
	const double bytes = (double)G * (8.0 * n * (p + 1 + (weighted ? 1 : 0)) + 8.0 * (p + 6) + (inference ? 8.0 * (5 * p + 2) : 0.0));
	// per step: a wide batch runs in several launches (slabs of groups), all of them counted
	const double acc_ms = steps > 0 ? kt.accumulate_ms / steps : 0.0;
	printf("{\"groups\": %lld, \"rows\": %lld, \"features\": %d, \"model\": \"%s\", \"steps\": %d, \"ms_per_step\": %.4f, "
	       "\"fits_per_s\": %.1f, \"accumulate_ms\": %.4f, \"solve_ms\": %.4f, \"accumulate_GBps\": %.1f, "
	       "\"group0_intercept\": %.12g, \"group0_r2\": %.12g, \"group0_status\": %g}\n",
	       G, n, p, model, steps, sec / steps * 1e3, G * steps / sec, acc_ms,
	       steps > 0 ? kt.solve_ms / steps : 0.0, acc_ms > 0 ? bytes / (acc_ms * 1e-3) / 1e9 : 0.0, h[p],
	       h[p + 1], h[p + 5]);
	anofox_hip_context_destroy(ctx);
	return 0;
}
