// lds_dma_x4_probe.hip — where `global_load_lds_dwordx4` (gfx950) puts a lane's 16 bytes: LDS address = M0 + 16 * lane ?
// With all 64 lanes and with the lower 32 only (EXEC masked).  Prints the first mismatch or "ok".  (Measurement tool.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void probe(const double *src, double *out, int half) {
	extern __shared__ double lds[];
	for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -1.0;
	__syncthreads();
	unsigned voff = threadIdx.x * 16u;
	unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds) + 64u; // 8 doubles in
	unsigned keep;
	if (half) {
		unsigned long long ex;
		asm volatile("s_mov_b64 %0, exec\n\ts_lshr_b64 exec, exec, 32\n\t"
		             "s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\ts_mov_b32 m0, %1\n\t"
		             "s_mov_b64 exec, %0"
		             : "=&s"(ex), "=&s"(keep)
		             : "v"(voff), "s"(dst), "s"(src)
		             : "memory", "scc");
	} else {
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
		             : "=&s"(keep)
		             : "v"(voff), "s"(dst), "s"(src)
		             : "memory");
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

int main() {
	std::vector<double> h(1024);
	for (int i = 0; i < 1024; ++i) h[i] = 1000.0 + i;
	double *d_src, *d_out;
	hipMalloc(&d_src, 1024 * 8);
	hipMalloc(&d_out, 512 * 8);
	hipMemcpy(d_src, h.data(), 1024 * 8, hipMemcpyHostToDevice);
	int bad = 0;
	for (int half = 0; half < 2; ++half) {
		hipLaunchKernelGGL(probe, dim3(1), dim3(64), 512 * 8, 0, d_src, d_out, half);
		std::vector<double> o(512);
		hipMemcpy(o.data(), d_out, 512 * 8, hipMemcpyDeviceToHost);
		const int n = half ? 64 : 128; // doubles moved
		for (int i = 0; i < 512; ++i) {
			const double want = (i >= 8 && i < 8 + n) ? 1000.0 + (i - 8) : -1.0;
			if (o[i] != want) {
				if (bad < 10) printf("half=%d: lds[%d] = %g, expected %g\n", half, i, o[i], want);
				++bad;
			}
		}
		printf("half=%d: %s\n", half, bad ? "MISMATCH" : "ok (LDS address = M0 + 16 * lane, inactive lanes write nothing)");
	}
	return bad ? 1 : 0;
}
