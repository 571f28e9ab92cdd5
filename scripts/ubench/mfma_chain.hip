// micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 chains on gfx950 (how fast can one accumulator be fed?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int CHAINS>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
	bf16x8 a, b;
	for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
	f32x16 acc[CHAINS];
	for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int k = 0; k < 24 / CHAINS; ++k)
#pragma unroll
			for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
	}
	float s = 0.f;
	for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS> void run(int blocks, int threads, const char* what) {
	float* out; hipMalloc(&out, sizeof(float) * blocks * threads);
	const int iters = 2000;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, 10);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double waves = (double)blocks * threads / 64, mf = waves * iters * 24;
	const double tflops = mf * 32768.0 / (ms * 1e-3) / 1e12;
	// waves per SIMD assuming an even spread over 1024 SIMDs
	printf("%-44s %8.3f ms  %8.1f TFLOP/s  ns per MFMA per wave %.2f\n", what, ms, tflops, ms * 1e6 / (iters * 24.0));
	hipFree(out);
}
int main() {
	run<1>(1024, 64, "1 wave/SIMD, 1 dependent chain");
	run<2>(1024, 64, "1 wave/SIMD, 2 independent chains");
	run<1>(1024, 128, "2 waves/SIMD, 1 chain each");
	run<1>(1024, 192, "3 waves/SIMD, 1 chain each");
	run<1>(1024, 256, "4 waves/SIMD, 1 chain each");
	run<2>(1024, 256, "4 waves/SIMD, 2 chains each");
	return 0;
}
