// mfma_f64_rate.hip — measures the issue rate of v_mfma_f64_16x16x4_f64 on the device (cycles per instruction
// per SIMD and chip TFLOP/s), the ceiling used for the wide path's roofline.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double dbl4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void rate_kernel(double *out, int iters, long long *cycles) {
	dbl4 acc[NACC];
	for (int i = 0; i < NACC; ++i) acc[i] = (dbl4){0, 0, 0, 0};
	double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
	const long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
	}
	const long long t1 = __builtin_amdgcn_s_memtime();
	double s = 0;
	for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// The same with operands that CHANGE from one instruction to the next (eight random doubles per lane, cycled): the
// accumulate kernel feeds the matrix cores random data, and switching activity counts against the power limit.
// PATTERN: which registers feed consecutive instructions
//   0  A and B differ from instruction to instruction        1  the same A for all, B differs
//   2  A == B (one register for both operands), differing     3  consecutive PAIRS share A, B differs
//   4  the accumulate kernel's order at p = 128, wave 0: tiles (0,0) (0,4) (1,1) (1,5) (2,3) (3,3) (3,7) (4,7) (6,6)
template <int NACC, int PATTERN>
__global__ __launch_bounds__(256) void rate_kernel_random(double *out, int iters, long long *cycles, const double *src) {
	dbl4 acc[NACC];
	for (int i = 0; i < NACC; ++i) acc[i] = (dbl4){0, 0, 0, 0};
	double a[8], b[8];
	for (int i = 0; i < 8; ++i) {
		a[i] = src[(threadIdx.x * 16 + i) & 4095];
		b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095];
	}
	const long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it += 8) { // operand registers picked at compile time: the loop body is MFMAs only
#pragma unroll
		for (int k = 0; k < 8; ++k) {
#pragma unroll
			for (int i = 0; i < NACC; ++i)
			{
				constexpr int TI[9] = {0, 0, 1, 1, 2, 3, 3, 4, 6}, TJ[9] = {0, 4, 1, 5, 3, 3, 7, 7, 6};
				double x = a[(i + k) & 7], y = b[(i * 3 + k) & 7];
				if (PATTERN == 1) x = a[0];
				if (PATTERN == 2) y = x;
				if (PATTERN == 3) x = a[((i >> 1) + k) & 7];
				if (PATTERN == 4) { x = a[TI[i % 9]]; y = a[TJ[i % 9]]; }
				acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
			}
		}
	}
	const long long t1 = __builtin_amdgcn_s_memtime();
	double s = 0;
	for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// How much matrix-core time does ordinary vector work cost?  9 MFMAs (distinct operands) plus NV independent
// v_fma_f64 per loop trip, in the same wave.
template <int NV>
__global__ __launch_bounds__(256) void rate_kernel_mixed(double *out, int iters, const double *src) {
	dbl4 acc[9];
	for (int i = 0; i < 9; ++i) acc[i] = (dbl4){0, 0, 0, 0};
	double a[8], b[8], v[8];
	for (int i = 0; i < 8; ++i) {
		a[i] = src[(threadIdx.x * 16 + i) & 4095];
		b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095];
		v[i] = a[i] * 0.001;
	}
	for (int it = 0; it < iters; it += 8) {
#pragma unroll
		for (int k = 0; k < 8; ++k) {
#pragma unroll
			for (int i = 0; i < 9; ++i) {
				acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(i + k) & 7], b[(i * 3 + k) & 7], acc[i], 0, 0, 0);
#pragma unroll
				for (int j = 0; j < NV / 9 + ((i < NV % 9) ? 1 : 0); ++j) v[(i + j) & 7] = fma(v[(i + j) & 7], 0.999999, a[(i + j + k) & 7]);
			}
		}
	}
	double s = 0;
	for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	for (int i = 0; i < 8; ++i) s += v[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
void run_mixed(int blocks) {
	double *out, *src;
	(void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
	(void)hipMalloc(&src, 4096 * sizeof(double));
	double h[4096];
	unsigned long long x = 88172645463325252ull;
	for (int i = 0; i < 4096; ++i) {
		x ^= x << 13; x ^= x >> 7; x ^= x << 17;
		h[i] = ((double)(x >> 11) / 9007199254740992.0) * 20.0 - 10.0;
	}
	(void)hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
	const int iters = 20000;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	hipLaunchKernelGGL(rate_kernel_mixed<NV>, dim3(blocks), dim3(256), 0, 0, out, 104, src);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(rate_kernel_mixed<NV>, dim3(blocks), dim3(256), 0, 0, out, iters, src);
	(void)hipEventRecord(e1);
	(void)hipDeviceSynchronize();
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	const double flops = ((double)blocks * 4) * (double)iters * 9 * 2.0 * 16 * 16 * 4;
	printf("%d waves/SIMD, 9 MFMA + %3d v_fma_f64 per trip   chip %.2f TFLOP/s (MFMA only)   %.3f ms\n", blocks / 256, NV, flops / (ms * 1e-3) / 1e12, ms);
	(void)hipFree(out);
	(void)hipFree(src);
}

template <int NACC, int PATTERN>
void run_random(int blocks, int threads, const char *label) {
	double *out, *src;
	long long *cyc;
	(void)hipMalloc(&out, (size_t)blocks * threads * sizeof(double));
	(void)hipMalloc(&src, 4096 * sizeof(double));
	(void)hipMalloc(&cyc, sizeof(long long));
	double h[4096];
	unsigned long long x = 88172645463325252ull;
	for (int i = 0; i < 4096; ++i) {
		x ^= x << 13; x ^= x >> 7; x ^= x << 17;
		h[i] = ((double)(x >> 11) / 9007199254740992.0) * 20.0 - 10.0;
	}
	(void)hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
	const int iters = 20000; // a multiple of 8: the operand index (i + it) & 7 is resolved by unrolling
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	hipLaunchKernelGGL((rate_kernel_random<NACC, PATTERN>), dim3(blocks), dim3(threads), 0, 0, out, 104, cyc, src);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL((rate_kernel_random<NACC, PATTERN>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc, src);
	(void)hipEventRecord(e1);
	(void)hipDeviceSynchronize();
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	const double flops = ((double)blocks * threads / 64) * (double)iters * NACC * 2.0 * 16 * 16 * 4;
	printf("%-28s acc=%2d  RANDOM operands                      chip %.2f TFLOP/s   %.3f ms\n", label, NACC, flops / (ms * 1e-3) / 1e12, ms);
	(void)hipFree(out);
	(void)hipFree(src);
	(void)hipFree(cyc);
}

template <int NACC>
void run(int blocks, int threads, const char *label) {
	double *out;
	long long *cyc;
	(void)hipMalloc(&out, (size_t)blocks * threads * sizeof(double));
	(void)hipMalloc(&cyc, sizeof(long long));
	const int iters = 20000;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	hipLaunchKernelGGL(rate_kernel<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(rate_kernel<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
	(void)hipEventRecord(e1);
	(void)hipDeviceSynchronize();
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	long long c;
	(void)hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost);
	const double n_mfma_wave = (double)iters * NACC;
	const double waves = (double)blocks * threads / 64;
	const double flops = waves * n_mfma_wave * 2.0 * 16 * 16 * 4;
	printf("%-28s acc=%2d  cycles/MFMA/wave (s_memtime) %.1f   chip %.2f TFLOP/s   %.3f ms\n", label, NACC,
	       (double)c / n_mfma_wave, flops / (ms * 1e-3) / 1e12, ms);
	(void)hipFree(out);
	(void)hipFree(cyc);
}

int main() {
	run<1>(256, 256, "1 wave/SIMD, dependent");
	run<4>(256, 256, "1 wave/SIMD, 4 acc");
	run<9>(256, 256, "1 wave/SIMD, 9 acc");
	run<9>(512, 256, "2 waves/SIMD, 9 acc");
	run<4>(1024, 256, "4 waves/SIMD, 4 acc");
	run_random<9, 0>(256, 256, "1 wave/SIMD, 9 acc");
	run_random<9, 0>(512, 256, "2 waves/SIMD, 9 acc");
	run_random<4, 0>(1024, 256, "4 waves/SIMD, 4 acc");
	run_random<9, 1>(512, 256, "2 w/SIMD same A");
	run_random<9, 2>(512, 256, "2 w/SIMD A == B");
	run_random<9, 3>(512, 256, "2 w/SIMD pairs share A");
	run_random<9, 4>(512, 256, "2 w/SIMD kernel tile order");
	run_mixed<0>(512);
	run_mixed<9>(512);
	run_mixed<18>(512);
	run_mixed<36>(512);
	run_mixed<72>(512);
	run_mixed<36>(256);
	run_mixed<36>(1024);
	return 0;
}
