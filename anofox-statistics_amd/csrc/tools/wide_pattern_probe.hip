// wide_pattern_probe.hip — what the memory system delivers for accumulate_wide's ACCESS PATTERN alone (round 4).
//
// A columnar table of `ncol` f64 columns, groups of n consecutive rows, one 256-thread workgroup per group (as accumulate_wide), the
// rows read in chunks and only summed (two vector adds per 16-byte load, no LDS, no matrix instruction).  Variants:
//   ROWS = 16: a load instruction covers 16 rows x 8 columns (128-byte runs per column; accumulate_wide's staging), chunk = 32 rows
//   ROWS = 32: 32 rows x 4 columns (256-byte runs), chunk = 32 rows
//   ROWS = 128: 128 rows x 1 column (1 KB runs), chunk = 128 rows
// DEPTH chunks of loads are in flight per wavefront; BARRIER = a workgroup barrier per chunk; WPS = workgroups per CU (= wavefronts per SIMD:
// the register budget, and an LDS request of 160 KB / WPS that nothing uses, so that no more than WPS workgroups share a CU).
// usage: wide_pattern_probe [groups] [rows] [ncol]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));

template <int ROWS, int DEPTH, bool BARRIER, int WPS, int MAXSLOTS>
__global__ __launch_bounds__(256, WPS) void probe_kernel(const double *table, int64_t col_stride, int n, int ncol, double *out) {
	constexpr int CPI = 128 / ROWS;                 // columns per load instruction
	constexpr int CHUNK = ROWS == 128 ? 128 : 32;   // rows per chunk
	constexpr int IPC = CHUNK / ROWS;               // instructions per column set and chunk
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int colsub = lane / (64 / CPI), rp = lane % (64 / CPI);
	const int64_t g = blockIdx.x;
	const int nslots = (ncol + CPI - 1) / CPI;
	// (ncol is a multiple of the columns per instruction: every lane's column exists)
	const double *base = table + (int64_t)(CPI * wave + colsub) * col_stride + g * n + 2 * rp;
	const int64_t slot_stride = (int64_t)4 * CPI * col_stride;
	dbl2u r[DEPTH][MAXSLOTS][IPC];
	double sum = 0.0;
	const int nchunks = n / CHUNK; // (full chunks only)
	auto load = [&](int c, int d) {
#pragma unroll
		for (int q = 0; q < MAXSLOTS; ++q)
#pragma unroll
			for (int i = 0; i < IPC; ++i)
				if (wave + 4 * q < nslots) r[d][q][i] = *reinterpret_cast<const dbl2u *>(base + q * slot_stride + (int64_t)c * CHUNK + i * ROWS);
				else r[d][q][i] = (dbl2u){0.0, 0.0};
	};
	auto eat = [&](int d) {
#pragma unroll
		for (int q = 0; q < MAXSLOTS; ++q)
#pragma unroll
			for (int i = 0; i < IPC; ++i) sum += r[d][q][i].x + r[d][q][i].y;
	};
	int c = 0;
	if (nchunks >= 2 * DEPTH) {
#pragma unroll
		for (int d = 0; d < DEPTH; ++d) load(d, d);
		for (; c + 2 * DEPTH <= nchunks; c += DEPTH) {
#pragma unroll
			for (int d = 0; d < DEPTH; ++d) {
				eat(d);
				load(c + DEPTH + d, d);
				if (BARRIER) __syncthreads();
			}
		}
#pragma unroll
		for (int d = 0; d < DEPTH; ++d) eat(d);
		c += DEPTH;
	}
	for (; c < nchunks; ++c) {
		load(c, 0);
		eat(0);
	}
	out[g * 256 + threadIdx.x] = sum;
}

template <int ROWS, int DEPTH, bool BARRIER, int WPS>
void run(const char *name, const double *table, int64_t col_stride, int G, int n, int ncol, double *out) {
	constexpr int CPI = 128 / ROWS;
	constexpr int MAXSLOTS = ROWS == 16 ? 2 : (ROWS == 32 ? 4 : 16); // up to 64 columns over four wavefronts
	if ((ncol + CPI - 1) / CPI > 4 * MAXSLOTS || ncol % CPI) { printf("%s: columns\n", name); return; }
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const size_t lds = (size_t)160 * 1024 / WPS - 1024;
	CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_kernel<ROWS, DEPTH, BARRIER, WPS, MAXSLOTS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe_kernel<ROWS, DEPTH, BARRIER, WPS, MAXSLOTS>), dim3(G), dim3(256), lds, 0, table, col_stride, n, ncol, out);
	CHECK(hipDeviceSynchronize());
	const int reps = 5;
	CHECK(hipEventRecord(e0));
	for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe_kernel<ROWS, DEPTH, BARRIER, WPS, MAXSLOTS>), dim3(G), dim3(256), lds, 0, table, col_stride, n, ncol, out);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	ms /= reps;
	const int chunk = ROWS == 128 ? 128 : 32;
	const double bytes = (double)G * (n / chunk * chunk) * ncol * 8.0;
	printf("%-64s %8.3f ms  %6.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
	fflush(stdout);
}

int main(int argc, char **argv) {
	const int G = argc > 1 ? atoi(argv[1]) : 50000, n = argc > 2 ? atoi(argv[2]) : 1000, ncol = argc > 3 ? atoi(argv[3]) : 64;
	const int64_t col_stride = (int64_t)G * n;
	double *table, *out;
	CHECK(hipMalloc(&table, (size_t)col_stride * ncol * sizeof(double) + 4096));
	CHECK(hipMalloc(&out, (size_t)G * 256 * sizeof(double)));
	CHECK(hipMemset(table, 0, (size_t)col_stride * ncol * sizeof(double) + 4096));
	printf("groups %d x rows %d x columns %d (%.1f GB), one 256-thread workgroup per group\n", G, n, ncol, (double)col_stride * ncol * 8 / 1e9);
	run<16, 1, false, 2>("16 rows x 8 cols / instr (128 B runs), depth 1, 2 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<16, 2, false, 2>("16 rows x 8 cols / instr (128 B runs), depth 2, 2 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<16, 2, true, 2>("16 rows x 8 cols / instr, depth 2, barrier per chunk, 2 wg/CU", table, col_stride, G, n, ncol, out);
	run<16, 2, false, 4>("16 rows x 8 cols / instr, depth 2, 4 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<16, 2, false, 3>("16 rows x 8 cols / instr, depth 2, 3 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<16, 2, false, 8>("16 rows x 8 cols / instr, depth 2, 8 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<16, 4, false, 2>("16 rows x 8 cols / instr, depth 4, 2 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<16, 4, false, 4>("16 rows x 8 cols / instr, depth 4, 4 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<32, 2, false, 2>("32 rows x 4 cols / instr (256 B runs), depth 2, 2 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<32, 4, false, 4>("32 rows x 4 cols / instr (256 B runs), depth 4, 4 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<128, 1, false, 2>("128 rows x 1 col / instr (1 KB runs), depth 1, 2 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<128, 1, false, 4>("128 rows x 1 col / instr (1 KB runs), depth 1, 4 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<128, 2, false, 2>("128 rows x 1 col / instr (1 KB runs), depth 2, 2 workgroups/CU", table, col_stride, G, n, ncol, out);
	run<128, 1, true, 2>("128 rows x 1 col / instr, depth 1, barrier per chunk, 2 wg/CU", table, col_stride, G, n, ncol, out);
	return 0;
}
