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
	return 0;
}
