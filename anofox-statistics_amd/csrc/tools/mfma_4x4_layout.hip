// mfma_4x4_layout.hip — which lane holds which operand / result element of v_mfma_f64_4x4x4_4b_f64?  One-hot inputs:
// a = 1 on lane LA only, b = 1 on lane LB only; the lanes whose result is non-zero are printed for every (LA, LB).
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(unsigned long long *mask) { // mask[LA * 64 + LB] = lanes with a non-zero result
	const int lane = threadIdx.x;
	for (int la = 0; la < 64; ++la)
		for (int lb = 0; lb < 64; ++lb) {
			const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
			const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
			const unsigned long long m = __ballot(d != 0.0);
			if (lane == 0) mask[la * 64 + lb] = m;
		}
}

int main() {
	unsigned long long *d, h[4096];
	(void)hipMalloc(&d, sizeof h);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
	(void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	for (int la = 0; la < 64; ++la) {
		printf("A lane %2d:", la);
		for (int lb = 0; lb < 64; ++lb)
			if (h[la * 64 + lb]) {
				int out = -1, cnt = 0;
				for (int l = 0; l < 64; ++l)
					if ((h[la * 64 + lb] >> l) & 1) { out = l; ++cnt; }
				printf(" B%d->D%d%s", lb, out, cnt > 1 ? "*" : "");
			}
		printf("\n");
	}
	return 0;
}
