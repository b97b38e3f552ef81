// mid_pattern_probe.hip — what limits accumulate_mid (8 < p <= 32)?  One wavefront per 1000-row group streams NC column
// arrays with the kernel's access shapes and does a stand-in amount of work per 16 rows; no correctness, only rates.
//   RL      1: 8-byte loads, 64 rows per block     2: 16-byte loads, 128 rows per block
//   MF      dependent v_mfma_f64_16x16x4_f64 per 16 rows (0 = none; the real kernel issues 4 T (T + 1) / 2)
//   VA      f64 FMAs per loaded value (stand-in for the row filter / centring / side sums)
//   waves per SIMD are capped with dynamic LDS
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mid_pattern_probe.hip -o mid_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef double dbl4 __attribute__((ext_vector_type(4)));
constexpr int kMaxCols = 34;
struct Cols {
	const double *c[kMaxCols];
};

template <int NC, int RL, int MF, int VA>
__global__ __launch_bounds__(256) void probe(Cols cols, long long rows_per_group, long long n_groups, double *out) {
	const long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	if (g >= n_groups) return;
	const long long lo = g * rows_per_group, hi = lo + rows_per_group;
	constexpr int BR = 64 * RL;
	double nx[NC][RL];
	auto issue = [&](long long b) {
#pragma unroll
		for (int j = 0; j < NC; ++j) {
			if (RL == 2) {
				const long long r = b + 2 * lane < hi - 1 ? b + 2 * lane : hi - 2;
				const dbl2 v = *reinterpret_cast<const dbl2 *>(cols.c[j] + r);
				nx[j][0] = v.x;
				nx[j][RL - 1] = v.y;
			} else {
				const long long r = b + lane < hi ? b + lane : hi - 1;
				nx[j][0] = cols.c[j][r];
			}
		}
	};
	issue(lo);
	dbl4 acc = {0, 0, 0, 0};
	double s[4] = {0, 0, 0, 0};
	for (long long b = lo; b < hi; b += BR) {
		double v[NC][RL];
#pragma unroll
		for (int j = 0; j < NC; ++j)
#pragma unroll
			for (int e = 0; e < RL; ++e) v[j][e] = nx[j][e];
		if (b + BR < hi) issue(b + BR);
#pragma unroll
		for (int j = 0; j < NC; ++j)
#pragma unroll
			for (int e = 0; e < RL; ++e)
#pragma unroll
				for (int k = 0; k < VA; ++k) s[(j + k) & 3] = fma(v[j][e], v[(j + 1) % NC][e], s[(j + k) & 3]);
		// MF dependent MFMAs per 16 rows: 4 RL steps per block
#pragma unroll
		for (int st = 0; st < 4 * RL * MF; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[st % NC][0], v[(st + 1) % NC][0], acc, 0, 0, 0);
	}
	const double r = s[0] + s[1] + s[2] + s[3] + acc[0] + acc[1] + acc[2] + acc[3];
	if (r == 123.456) out[0] = r;
}

__global__ void fill_random(unsigned long long *a, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull;
		x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
		a[i] = (x & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull; // a double in [1, 2)
	}
}

static double *g_out;
static Cols g_cols;
static long long g_rows = 1000, g_groups;

template <int NC, int RL, int MF, int VA>
static void run(int waves_per_simd, const char *what) {
	const void *fn = reinterpret_cast<const void *>(&probe<NC, RL, MF, VA>);
	(void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	// a 256-thread workgroup = 1 wave per SIMD: W workgroups per CU -> LDS of 160 KB / W (minus a little)
	const size_t lds = waves_per_simd >= 8 ? 0 : (size_t)(160 * 1024 / waves_per_simd) - 512;
	const unsigned wgs = (unsigned)((g_groups + 3) / 4);
	hipEvent_t a, b;
	(void)hipEventCreate(&a);
	(void)hipEventCreate(&b);
	hipLaunchKernelGGL((probe<NC, RL, MF, VA>), dim3(wgs), dim3(256), lds, 0, g_cols, g_rows, g_groups, g_out);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(a);
	for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<NC, RL, MF, VA>), dim3(wgs), dim3(256), lds, 0, g_cols, g_rows, g_groups, g_out);
	(void)hipEventRecord(b);
	(void)hipEventSynchronize(b);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, a, b);
	const double bytes = (double)g_groups * g_rows * NC * 8;
	printf("NC %2d  RL %d  MFMA/16rows %d  FMA/value %d  waves/SIMD <= %d  %-28s %7.3f ms  %6.3f TB/s\n", NC, RL, MF, VA, waves_per_simd, what, ms / 5,
	       bytes / (ms / 5 * 1e-3) / 1e12);
	fflush(stdout);
}

int main(int argc, char **argv) {
	g_groups = argc > 1 ? atoll(argv[1]) : 100000;
	const size_t n = (size_t)g_groups * g_rows;
	double *buf;
	if (hipMalloc(&buf, n * kMaxCols * sizeof(double)) != hipSuccess || hipMalloc(&g_out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipLaunchKernelGGL(fill_random, dim3(65536), dim3(256), 0, 0, (unsigned long long *)buf, n * kMaxCols);
	(void)hipDeviceSynchronize();
	for (int j = 0; j < kMaxCols; ++j) g_cols.c[j] = buf + (size_t)j * n;
	for (int w : {2, 3, 4}) {
		run<9, 2, 0, 6>(w, "narrow-like");
		run<17, 2, 0, 1>(w, "17 cols, loads only");
		run<17, 2, 4, 4>(w, "17 cols + 4 MFMA + 17 FMA/step");
		run<17, 2, 4, 9>(w, "17 cols + 4 MFMA + 38 FMA/step");
		run<17, 2, 4, 18>(w, "17 cols + 4 MFMA + 76 FMA/step");
		run<17, 2, 4, 36>(w, "17 cols + 4 MFMA + 153 FMA/step");
		run<17, 2, 0, 18>(w, "17 cols + 76 FMA/step, no MFMA");
		run<17, 2, 8, 1>(w, "17 cols + 8 MFMA");
		run<33, 1, 12, 9>(w, "33 cols 8B + 12 MFMA + 74 FMA/step");
	}
	return 0;
}
