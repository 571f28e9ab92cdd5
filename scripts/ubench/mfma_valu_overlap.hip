// micro-benchmark: do a dependent v_mfma_f32_32x32x16_bf16 chain in one wave and plain VALU work in ANOTHER wave of the
// same SIMD overlap on gfx950?  Workgroup = 512 threads = 8 waves = 2 per SIMD (wave w and w + 4 share SIMD w & 3),
// one workgroup per CU (LDS-limited).  mode 0: waves 0-3 run MFMA chains, waves 4-7 exit.  mode 1: waves 4-7 run the
// VALU loop (the top-4 insertion network of match.hip), waves 0-3 exit.  mode 2: both.  mode 3: every wave alternates
// a 24-MFMA phase and a VALU phase (what k_match_sweep does), two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void topk4(float (&ts)[4], int (&ti)[4], float s, int idx) {
#pragma unroll
	for (int r = 3; r >= 0; --r) {
		const bool up = r > 0 && s > ts[r - 1];
		const float ns = up ? ts[r - 1] : s; const int ni = up ? ti[r - 1] : idx;
		const bool here = s > ts[r];
		ts[r] = here ? ns : ts[r]; ti[r] = here ? ni : ti[r];
	}
}

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters, int valu_per_iter, unsigned long long* clk) {
	const unsigned long long c0 = clock64(), w0 = wall_clock64();
	__shared__ float pad[30000];       // 120 KB: one workgroup per CU
	const int wave = threadIdx.x >> 6;
	pad[threadIdx.x] = 0.f;
	bf16x8 a, b;
	for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
	f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
	float ts[4] = {-1e30f, -1e30f, -1e30f, -1e30f}; int ti[4] = {-1, -1, -1, -1};
	float x = (float)threadIdx.x * 0.37f;
	const bool do_mfma = MODE == 3 || ((MODE == 0 || MODE == 2) && wave < 4);
	const bool do_valu = MODE == 3 || ((MODE == 1 || MODE == 2) && wave >= 4);
	for (int it = 0; it < iters; ++it) {
		if (do_mfma) {
#pragma unroll
			for (int k = 0; k < 24; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
		}
		if (do_valu) {
			for (int v = 0; v < valu_per_iter; ++v) {   // 15 VALU instructions per round (the insertion network, unconditional)
				x = x * 1.0001f + 0.5f;
				topk4(ts, ti, x, it * 64 + v);
			}
		}
	}
	float s = ts[0] + ts[1] + ts[2] + ts[3] + (float)(ti[0] + ti[1] + ti[2] + ti[3]);
	for (int i = 0; i < 16; ++i) s += acc[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s + pad[threadIdx.x];
	if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
template <int MODE> float run(int valu_per_iter, const char* what, int iters = 2000) {
	const int blocks = 256, threads = 512;
	unsigned long long* clk; hipMalloc(&clk, 16); hipMemset(clk, 0, 16);
	float* out; hipMalloc(&out, sizeof(float) * blocks * threads);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 10, valu_per_iter, (unsigned long long*)nullptr);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, valu_per_iter, clk);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
	printf("%-70s %8.3f ms  = %7.1f ns per iteration   shader clock %.2f GHz\n", what, ms, ms * 1e6 / iters, (double)h[0] / (double)(h[1] ? h[1] : 1) * 0.1);
	hipFree(out);
	return ms;
}
int main() {
	for (int v : {12, 24, 48}) {
		char buf[128];
		printf("-- VALU rounds per iteration: %d (x ~17 instructions), 24 MFMAs per iteration\n", v);
		snprintf(buf, sizeof buf, "mode 0: MFMA chain in waves 0-3 only"); const float m = run<0>(v, buf);
		snprintf(buf, sizeof buf, "mode 1: VALU loop in waves 4-7 only"); const float a = run<1>(v, buf);
		snprintf(buf, sizeof buf, "mode 2: MFMA waves + VALU waves on the same SIMDs"); const float c = run<2>(v, buf);
		snprintf(buf, sizeof buf, "mode 3: every wave alternates MFMA phase / VALU phase, 2 per SIMD"); const float d = run<3>(v, buf);
		printf("   sum %.3f  max %.3f  measured both %.3f   |  alternating: serial 2 x (m + a) = %.3f, measured %.3f\n", m + a, m > a ? m : a, c, 2 * (m + a), d);
	}
	printf("-- sustained (long kernels): does the clock hold?\n");
	run<0>(12, "mode 0, 4 MFMA waves per CU, 100000 iterations", 100000);
	run<3>(12, "mode 3, 8 alternating waves per CU, 12 VALU rounds, 40000 iterations", 40000);
	run<3>(4, "mode 3, 8 alternating waves per CU, 4 VALU rounds, 60000 iterations", 60000);
	return 0;
}
