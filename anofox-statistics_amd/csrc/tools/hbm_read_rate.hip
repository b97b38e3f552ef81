// hbm_read_rate.hip — the read bandwidth a plain streaming kernel reaches on the device: the achievable ceiling next
// to which the accumulate kernels' TB/s are read (the 8 TB/s in the roofline is the datasheet figure).
//   variant A: one array, every lane reads 16 B per load, grid-stride, UNROLL independent loads in flight;
//   variant B: nine arrays (the layout of the p = 8 fit: x_1..x_8, y), a wavefront reads the same 128 rows of each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int UNROLL>
__global__ __launch_bounds__(256) void read_one(const dbl2 *__restrict__ a, size_t n2, double *out) {
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	double s = 0.0;
	for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
		dbl2 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) v[u] = a[i + u * stride];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y;
	}
	for (; i < n2; i += stride) s += a[i].x + a[i].y;
	if (s == 123.456) out[0] = s;
}

struct Cols {
	const dbl2 *c[9];
};

// each wavefront takes tiles of 128 rows (64 lanes x 16 B) from all nine columns, next tile's loads issued first
__global__ __launch_bounds__(256) void read_nine(Cols cols, size_t rows2, double *out) {
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const int lane = threadIdx.x & 63;
	double s = 0.0;
	for (size_t t = wave * 64; t < rows2; t += nwaves * 64) {
		const size_t i = t + lane;
		if (i < rows2) {
			dbl2 v[9];
#pragma unroll
			for (int j = 0; j < 9; ++j) v[j] = cols.c[j][i];
#pragma unroll
			for (int j = 0; j < 9; ++j) s += v[j].x + v[j].y;
		}
	}
	if (s == 123.456) out[0] = s;
}

int main(int argc, char **argv) {
	const size_t gib = argc > 1 ? (size_t)atoll(argv[1]) : 36; // total bytes read per launch, GiB
	const size_t bytes = gib << 30;
	char *buf;
	double *out;
	if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void)hipMemset(buf, 0, bytes);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	auto time = [&](const char *label, auto launch) {
		launch();
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(e0);
		for (int r = 0; r < 5; ++r) launch();
		(void)hipEventRecord(e1);
		(void)hipDeviceSynchronize();
		float ms;
		(void)hipEventElapsedTime(&ms, e0, e1);
		printf("%-44s %7.3f ms per pass  %6.3f TB/s\n", label, ms / 5, (double)bytes / (ms / 5 * 1e-3) / 1e12);
	};
	const size_t n2 = bytes / 16;
	for (int blocks : {2048, 8192, 32768}) {
		char label[96];
		snprintf(label, sizeof label, "one array, 16 B/lane, 4 in flight, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL(read_one<4>, dim3(blocks), dim3(256), 0, 0, (const dbl2 *)buf, n2, out); });
		snprintf(label, sizeof label, "one array, 16 B/lane, 8 in flight, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL(read_one<8>, dim3(blocks), dim3(256), 0, 0, (const dbl2 *)buf, n2, out); });
	}
	Cols cols;
	const size_t rows2 = n2 / 9;
	for (int j = 0; j < 9; ++j) cols.c[j] = (const dbl2 *)buf + (size_t)j * rows2;
	const size_t bytes9 = rows2 * 9 * 16;
	for (int blocks : {2048, 8192, 32768}) {
		char label[96];
		snprintf(label, sizeof label, "nine columns, 128-row tiles, %d WGs", blocks);
		launch_again:
		hipLaunchKernelGGL(read_nine, dim3(blocks), dim3(256), 0, 0, cols, rows2, out);
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(e0);
		for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(read_nine, dim3(blocks), dim3(256), 0, 0, cols, rows2, out);
		(void)hipEventRecord(e1);
		(void)hipDeviceSynchronize();
		float ms;
		(void)hipEventElapsedTime(&ms, e0, e1);
		printf("%-44s %7.3f ms per pass  %6.3f TB/s\n", label, ms / 5, (double)bytes9 / (ms / 5 * 1e-3) / 1e12);
		(void)&&launch_again;
	}
	(void)hipFree(buf);
	(void)hipFree(out);
	return 0;
}
