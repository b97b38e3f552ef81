// mfma_f64_4x4_rate.hip — issue rate of v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 products per instruction, one
// result per lane) next to v_mfma_f64_16x16x4_f64: is the small shape a cheaper way to the few remainder columns of a
// design whose width is just past a multiple of 16?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double dbl4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k4(double *out, int iters, const double *src) {
	double acc[NACC];
	for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
	double a[8], b[8];
	for (int i = 0; i < 8; ++i) {
		a[i] = src[(threadIdx.x * 16 + i) & 4095];
		b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095];
	}
	for (int it = 0; it < iters; it += 8) {
#pragma unroll
		for (int k = 0; k < 8; ++k)
#pragma unroll
			for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[(i + k) & 7], b[(3 * i + k) & 7], acc[i], 0, 0, 0);
	}
	double s = 0;
	for (int i = 0; i < NACC; ++i) s += acc[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(double *out, int iters, const double *src) {
	dbl4 acc[NACC];
	for (int i = 0; i < NACC; ++i) acc[i] = (dbl4){0, 0, 0, 0};
	double a[8], b[8];
	for (int i = 0; i < 8; ++i) {
		a[i] = src[(threadIdx.x * 16 + i) & 4095];
		b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095];
	}
	for (int it = 0; it < iters; it += 8) {
#pragma unroll
		for (int k = 0; k < 8; ++k)
#pragma unroll
			for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(i + k) & 7], b[(3 * i + k) & 7], acc[i], 0, 0, 0);
	}
	double s = 0;
	for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
	double *out, *src;
	const int blocks = 256 * 2, iters = 4096;
	(void)hipMalloc(&out, (size_t)blocks * 256 * 8);
	(void)hipMalloc(&src, 4096 * 8);
	double h[4096];
	for (int i = 0; i < 4096; ++i) h[i] = 1.0 + (i * 2654435761u % 1000) * 1e-3;
	(void)hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	auto run = [&](const char *what, auto launch, double flops_per_inst, int nacc) {
		launch();
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(e0);
		launch();
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		float ms = 0;
		(void)hipEventElapsedTime(&ms, e0, e1);
		const double insts = (double)blocks * 4 * iters * nacc; // per wave
		printf("%-34s %8.3f ms  %7.2f G wave-instructions/s  %6.2f TFLOP/s\n", what, ms, insts / (ms * 1e-3) / 1e9, insts * flops_per_inst / (ms * 1e-3) / 1e12);
	};
	run("16x16x4, 8 accumulators, 2 waves/SIMD", [&] { hipLaunchKernelGGL((k16<8>), dim3(blocks), dim3(256), 0, 0, out, iters, src); }, 2048.0, 8);
	run("4x4x4 (4 blocks), 8 accumulators", [&] { hipLaunchKernelGGL((k4<8>), dim3(blocks), dim3(256), 0, 0, out, iters, src); }, 512.0, 8);
	run("4x4x4 (4 blocks), 16 accumulators", [&] { hipLaunchKernelGGL((k4<16>), dim3(blocks), dim3(256), 0, 0, out, iters, src); }, 512.0, 16);
	return 0;
}
