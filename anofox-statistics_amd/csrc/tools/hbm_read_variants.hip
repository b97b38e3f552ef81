// hbm_read_variants.hip — does any access pattern read faster than the grid-stride kernel of hbm_read_rate.hip?
// (experiment: contiguous region per workgroup, one eighth of the array per XCD, non-temporal loads, 32 B per lane)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_stride(const dbl2 *__restrict__ a, size_t n2, double *out) {
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	double s = 0.0;
	for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
		dbl2 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y;
	}
	if (s == 123.456) out[0] = s;
}

// workgroup b streams its own contiguous region
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_chunked(const dbl2 *__restrict__ a, size_t n2, double *out) {
	const size_t per = n2 / gridDim.x;
	const dbl2 *p = a + (size_t)blockIdx.x * per;
	double s = 0.0;
	for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < per; i += UNROLL * 256) {
		dbl2 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y;
	}
	if (s == 123.456) out[0] = s;
}

// one eighth of the array per XCD (workgroup b runs on XCD b % 8), grid-stride inside the eighth
template <int UNROLL>
__global__ __launch_bounds__(256) void read_xcd(const dbl2 *__restrict__ a, size_t n2, double *out) {
	const size_t eighth = n2 / 8;
	const dbl2 *p = a + (size_t)(blockIdx.x & 7) * eighth;
	const size_t stride = (size_t)(gridDim.x >> 3) * blockDim.x;
	size_t i = (size_t)(blockIdx.x >> 3) * blockDim.x + threadIdx.x;
	double s = 0.0;
	for (; i + (UNROLL - 1) * stride < eighth; i += UNROLL * stride) {
		dbl2 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y;
	}
	if (s == 123.456) out[0] = s;
}

struct Cols {
	const dbl2 *c[9];
};
// the layout of the p = 8 fit: nine arrays, a wavefront reads TILES consecutive 128-row tiles of each (one group = 8 tiles)
template <bool NT, int TILES>
__global__ __launch_bounds__(256) void read_nine(Cols cols, size_t rows2, double *out) {
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const int lane = threadIdx.x & 63;
	double s = 0.0;
	for (size_t t = wave * 64 * TILES; t + 64 * TILES <= rows2; t += nwaves * 64 * TILES) {
#pragma unroll 1
		for (int k = 0; k < TILES; ++k) {
			const size_t i = t + 64 * k + lane;
			dbl2 v[9];
#pragma unroll
			for (int j = 0; j < 9; ++j) v[j] = NT ? __builtin_nontemporal_load(cols.c[j] + i) : cols.c[j][i];
#pragma unroll
			for (int j = 0; j < 9; ++j) s += v[j].x + v[j].y;
		}
	}
	if (s == 123.456) out[0] = s;
}

// the launch shape of accumulate_narrow_kernel: a wavefront reads ONE group (8 tiles of 128 rows from nine columns) and
// ends; four groups per workgroup, one workgroup per four groups
template <bool NT>
__global__ __launch_bounds__(256) void read_nine_once(Cols cols, size_t rows2, double *out) {
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	double s = 0.0;
	const size_t t = wave * 64 * 8;
	if (t + 64 * 8 > rows2) return;
#pragma unroll 1
	for (int k = 0; k < 8; ++k) {
		const size_t i = t + 64 * k + lane;
		dbl2 v[9];
#pragma unroll
		for (int j = 0; j < 9; ++j) v[j] = NT ? __builtin_nontemporal_load(cols.c[j] + i) : cols.c[j][i];
#pragma unroll
		for (int j = 0; j < 9; ++j) s += v[j].x + v[j].y;
	}
	if (s == 123.456) out[0] = s;
}

__global__ void fill_random(unsigned long long *a, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull;
		x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
		a[i] = (x & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull; // a double in [1, 2)
	}
}

// read_nine_once plus what else accumulate_narrow_kernel does to memory: the group's bounds from an offsets array and a
// 77-double record written per group (WRITE), and COMPUTE dependent FMAs per loaded value between the tiles
template <bool NT, int WRITE, int COMPUTE>
__global__ __launch_bounds__(256) void read_nine_like(Cols cols, const long long *offs, size_t n_groups, double *recs, double *out) {
	const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	if (g >= n_groups) return;
	const size_t lo = (size_t)offs[g] >> 1, hi = (size_t)offs[g + 1] >> 1; // in 16-byte units
	double s = 0.0, acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	dbl2 nx[9];
#pragma unroll
	for (int j = 0; j < 9; ++j) nx[j] = NT ? __builtin_nontemporal_load(cols.c[j] + lo + lane) : cols.c[j][lo + lane];
	for (size_t t = lo; t < hi; t += 64) {
		dbl2 v[9];
#pragma unroll
		for (int j = 0; j < 9; ++j) v[j] = nx[j];
		if (t + 64 < hi) {
#pragma unroll
			for (int j = 0; j < 9; ++j) nx[j] = NT ? __builtin_nontemporal_load(cols.c[j] + t + 64 + lane) : cols.c[j][t + 64 + lane];
		}
#pragma unroll
		for (int j = 0; j < 9; ++j) {
			double a = v[j].x, b = v[j].y;
#pragma unroll
			for (int c = 0; c < COMPUTE; ++c) acc[(j + c) % 9] = fma(a, b, acc[(j + c) % 9]);
			s += a + b;
		}
	}
#pragma unroll
	for (int j = 0; j < 9; ++j) s += acc[j];
	if (WRITE == 1) {
		if (lane < 64) recs[g * 77 + lane] = s;
		if (lane < 13) recs[g * 77 + 64 + lane] = s;
	} else if (WRITE == 2) { // non-temporal stores
		if (lane < 64) __builtin_nontemporal_store(s, recs + g * 77 + lane);
		if (lane < 13) __builtin_nontemporal_store(s, recs + g * 77 + 64 + lane);
	} else if (WRITE == 3) { // records padded to 80 doubles = five 128-byte lines, written as 16 bytes per lane by 40 lanes
		if (lane < 40) *reinterpret_cast<dbl2 *>(recs + g * 80 + 2 * lane) = (dbl2){s, s};
	} else if (WRITE == 4) { // the same, non-temporal
		if (lane < 40) __builtin_nontemporal_store((dbl2){s, s}, reinterpret_cast<dbl2 *>(recs + g * 80 + 2 * lane));
	}
	if (s == 123.456) out[0] = s;
}

// Several consecutive groups per wavefront (wave w of a workgroup: groups base + w NG .. base + w NG + NG - 1, i.e. NG x 8 KB
// contiguous per column), records either written per group (WRITE 1) or staged in LDS and flushed by the workgroup as one
// contiguous 4 NG x 77-double block at its end (WRITE 5): fewer, larger writes between the read streams.
template <bool NT, int NG, int WRITE>
__global__ __launch_bounds__(256) void read_nine_multi(Cols cols, const long long *offs, size_t n_groups, double *recs, double *out) {
	extern __shared__ double stage[]; // [4 NG][77] when WRITE == 5 (the launch passes at least 70 KB to fix the occupancy)
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const size_t base = (size_t)blockIdx.x * 4 * NG;
	double s = 0.0;
	for (int i = 0; i < NG; ++i) {
		const size_t g = base + (size_t)wave * NG + i;
		if (g >= n_groups) break;
		const size_t lo = (size_t)offs[g] >> 1, hi = (size_t)offs[g + 1] >> 1;
		dbl2 nx[9];
#pragma unroll
		for (int j = 0; j < 9; ++j) nx[j] = NT ? __builtin_nontemporal_load(cols.c[j] + lo + lane) : cols.c[j][lo + lane];
		double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
		for (size_t t = lo; t < hi; t += 64) {
			dbl2 v[9];
#pragma unroll
			for (int j = 0; j < 9; ++j) v[j] = nx[j];
			if (t + 64 < hi) {
#pragma unroll
				for (int j = 0; j < 9; ++j) nx[j] = NT ? __builtin_nontemporal_load(cols.c[j] + t + 64 + lane) : cols.c[j][t + 64 + lane];
			}
#pragma unroll
			for (int j = 0; j < 9; ++j) {
#pragma unroll
				for (int c = 0; c < 6; ++c) acc[(j + c) % 9] = fma(v[j].x, v[j].y, acc[(j + c) % 9]);
			}
		}
		double r = 0.0;
#pragma unroll
		for (int j = 0; j < 9; ++j) r += acc[j];
		s += r;
		if (WRITE == 1) {
			recs[g * 77 + lane] = r;
			if (lane < 13) recs[g * 77 + 64 + lane] = r;
		} else if (WRITE == 5) {
			double *d = stage + (size_t)(wave * NG + i) * 77;
			d[lane] = r;
			if (lane < 13) d[64 + lane] = r;
		}
	}
	if (WRITE == 5) {
		__syncthreads();
		const size_t n_rec = (base + 4 * NG <= n_groups) ? 4 * NG : (n_groups > base ? n_groups - base : 0);
		double *dst = recs + base * 77;
		for (size_t k = threadIdx.x; k < n_rec * 77; k += 256) __builtin_nontemporal_store(stage[k], dst + k);
	}
	if (s == 123.456) out[0] = s;
}

int main(int argc, char **argv) {
	const size_t gib = argc > 1 ? (size_t)atoll(argv[1]) : 36;
	const size_t bytes = gib << 30;
	char *buf;
	double *out;
	if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void)hipMemset(buf, 0, bytes);
	if (argc > 2 && atoi(argv[2])) { // random doubles instead of zeros
		hipLaunchKernelGGL(fill_random, dim3(65536), dim3(256), 0, 0, (unsigned long long *)buf, bytes / 8);
		(void)hipDeviceSynchronize();
		printf("buffer filled with random doubles\n");
	}
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
		printf("%-52s %7.3f ms per pass  %6.3f TB/s\n", label, ms / 5, (double)bytes / (ms / 5 * 1e-3) / 1e12);
	};
	const size_t n2 = bytes / 16;
	const dbl2 *a = (const dbl2 *)buf;
	for (int blocks : {4096, 32768, 131072}) {
		char label[96];
		snprintf(label, sizeof label, "grid-stride, 8 in flight, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_stride<8, false>), dim3(blocks), dim3(256), 0, 0, a, n2, out); });
		snprintf(label, sizeof label, "grid-stride, 16 in flight, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_stride<16, false>), dim3(blocks), dim3(256), 0, 0, a, n2, out); });
		snprintf(label, sizeof label, "grid-stride, non-temporal, 8 in flight, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_stride<8, true>), dim3(blocks), dim3(256), 0, 0, a, n2, out); });
		snprintf(label, sizeof label, "contiguous region per WG, 8 in flight, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_chunked<8, false>), dim3(blocks), dim3(256), 0, 0, a, n2, out); });
		snprintf(label, sizeof label, "contiguous region per WG, non-temporal, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_chunked<8, true>), dim3(blocks), dim3(256), 0, 0, a, n2, out); });
		snprintf(label, sizeof label, "one eighth per XCD, 8 in flight, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_xcd<8>), dim3(blocks), dim3(256), 0, 0, a, n2, out); });
	}
	Cols cols;
	const size_t rows2 = n2 / 9;
	for (int j = 0; j < 9; ++j) cols.c[j] = a + (size_t)j * rows2;
	// the occupancy of accumulate_narrow_kernel (212 VGPRs: two waves per SIMD), forced with 70 KB of dynamic LDS per workgroup
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&read_nine<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&read_nine<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
	{
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&read_nine_once<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&read_nine_once<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
		const unsigned wgs = (unsigned)(rows2 / (64 * 8) / 4);
		char label[96];
		snprintf(label, sizeof label, "nine columns, one group per wave then exit, 2 waves/SIMD, %u WGs", wgs);
		time(label, [&] { hipLaunchKernelGGL((read_nine_once<false>), dim3(wgs), dim3(256), 70 * 1024, 0, cols, rows2, out); });
		snprintf(label, sizeof label, "same, non-temporal");
		time(label, [&] { hipLaunchKernelGGL((read_nine_once<true>), dim3(wgs), dim3(256), 70 * 1024, 0, cols, rows2, out); });
		snprintf(label, sizeof label, "same, non-temporal, default occupancy");
		time(label, [&] { hipLaunchKernelGGL((read_nine_once<true>), dim3(wgs), dim3(256), 0, 0, cols, rows2, out); });
	}
	{
		const size_t n_groups = rows2 / 512; // 1024 rows (512 16-byte units) per group
		long long *offs;
		double *recs;
		(void)hipMalloc(&offs, (n_groups + 1) * sizeof(long long));
		(void)hipMalloc(&recs, n_groups * 80 * sizeof(double));
		{
			long long *h = (long long *)malloc((n_groups + 1) * sizeof(long long));
			for (size_t g = 0; g <= n_groups; ++g) h[g] = (long long)(g * 1024);
			(void)hipMemcpy(offs, h, (n_groups + 1) * sizeof(long long), hipMemcpyHostToDevice);
			free(h);
		}
		const unsigned wgs = (unsigned)((n_groups + 3) / 4);
		const double scale = (double)(n_groups * 1024 * 72) / (double)bytes; // bytes actually read vs the label's total
		printf("(group-shaped kernels read %.3f of the buffer: multiply their TB/s by that)\n", scale);
#define LIKE(NTv, Wv, Cv, lds, text)                                                                                     \
	do {                                                                                                                 \
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&read_nine_like<NTv, Wv, Cv>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); \
		time(text, [&] { hipLaunchKernelGGL((read_nine_like<NTv, Wv, Cv>), dim3(wgs), dim3(256), lds, 0, cols, offs, n_groups, recs, out); }); \
	} while (0)
		LIKE(false, 0, 0, 70 * 1024, "group-shaped, prefetch 1, plain, 2 waves/SIMD");
		LIKE(true, 0, 0, 70 * 1024, "group-shaped, prefetch 1, NT, 2 waves/SIMD");
		LIKE(true, 1, 0, 70 * 1024, "group-shaped, NT + 77-double record writes");
		LIKE(true, 2, 0, 70 * 1024, "group-shaped, NT + non-temporal record writes");
		LIKE(true, 3, 0, 70 * 1024, "group-shaped, NT + 80-double aligned record writes");
		LIKE(true, 4, 0, 70 * 1024, "group-shaped, NT + aligned non-temporal record writes");
		LIKE(true, 4, 6, 70 * 1024, "group-shaped, NT + aligned NT writes + 108 FMA per row pair");
		LIKE(false, 1, 6, 70 * 1024, "group-shaped, plain + 77-double writes + 108 FMA");
		LIKE(false, 4, 6, 70 * 1024, "group-shaped, plain loads + aligned NT writes + 108 FMA");
#define MULTI(NTv, NGv, Wv, text)                                                                                        \
	do {                                                                                                                 \
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&read_nine_multi<NTv, NGv, Wv>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); \
		const unsigned wg = (unsigned)((n_groups + 4 * NGv - 1) / (4 * NGv));                                            \
		time(text, [&] { hipLaunchKernelGGL((read_nine_multi<NTv, NGv, Wv>), dim3(wg), dim3(256), 70 * 1024, 0, cols, offs, n_groups, recs, out); }); \
	} while (0)
		MULTI(false, 1, 1, "multi: 1 group/wave, plain, write per group + FMA");
		MULTI(false, 1, 5, "multi: 1 group/wave, plain, staged flush + FMA");
		MULTI(false, 4, 1, "multi: 4 groups/wave, plain, write per group + FMA");
		MULTI(false, 4, 5, "multi: 4 groups/wave, plain, staged flush + FMA");
		MULTI(false, 8, 5, "multi: 8 groups/wave, plain, staged flush + FMA");
		MULTI(false, 16, 5, "multi: 16 groups/wave, plain, staged flush + FMA");
		MULTI(false, 8, 0, "multi: 8 groups/wave, plain, no record + FMA");
		MULTI(true, 1, 1, "multi: 1 group/wave, NT, write per group + FMA");
		MULTI(true, 4, 5, "multi: 4 groups/wave, NT, staged flush + FMA");
		MULTI(true, 8, 5, "multi: 8 groups/wave, NT, staged flush + FMA");
		MULTI(true, 16, 5, "multi: 16 groups/wave, NT, staged flush + FMA");
		MULTI(true, 8, 0, "multi: 8 groups/wave, NT, no record + FMA");
	}
	for (int blocks : {8192, 65536}) {
		char label[96];
		snprintf(label, sizeof label, "nine columns, 8 tiles, 2 waves per SIMD, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_nine<false, 8>), dim3(blocks), dim3(256), 70 * 1024, 0, cols, rows2, out); });
		snprintf(label, sizeof label, "nine columns, 8 tiles, 2 waves per SIMD, non-temporal, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_nine<true, 8>), dim3(blocks), dim3(256), 70 * 1024, 0, cols, rows2, out); });
	}
	for (int blocks : {8192, 32768}) {
		char label[96];
		snprintf(label, sizeof label, "nine columns, 1 tile per turn, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_nine<false, 1>), dim3(blocks), dim3(256), 0, 0, cols, rows2, out); });
		snprintf(label, sizeof label, "nine columns, 1 tile per turn, non-temporal, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_nine<true, 1>), dim3(blocks), dim3(256), 0, 0, cols, rows2, out); });
		snprintf(label, sizeof label, "nine columns, 8 tiles per turn, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_nine<false, 8>), dim3(blocks), dim3(256), 0, 0, cols, rows2, out); });
		snprintf(label, sizeof label, "nine columns, 8 tiles per turn, non-temporal, %d WGs", blocks);
		time(label, [&] { hipLaunchKernelGGL((read_nine<true, 8>), dim3(blocks), dim3(256), 0, 0, cols, rows2, out); });
	}
	return 0;
}
