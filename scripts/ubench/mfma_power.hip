// micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate and shader clock on gfx950 as a function of the operand DATA
// (constant operands vs random bf16 operands cycling through 8 A and 8 B fragments) on ~50 ms kernels -- the matrix pipe's
// power draw depends on how many operand bits toggle, and the clock follows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(const u32x4* __restrict__ src, float* out, int iters, unsigned long long* clk) {
	const unsigned long long c0 = clock64(), w0 = wall_clock64();
	u32x4 a[8], b[8];
	for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x & 63) * 16 + i]; b[i] = src[(threadIdx.x & 63) * 16 + 8 + i]; }
	f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int kb = 0; kb < 8; ++kb) {
			acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kb]), __builtin_bit_cast(bf16x8, b[kb]), acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kb]), __builtin_bit_cast(bf16x8, b[(kb + 3) & 7]), acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(kb + 5) & 7]), __builtin_bit_cast(bf16x8, b[kb]), acc, 0, 0, 0);
		}
	}
	float s = 0.f; for (int i = 0; i < 16; ++i) s += acc[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
static void run(const u32x4* src, int blocks, int iters, const char* what) {
	float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
	unsigned long long* clk; hipMalloc(&clk, 16);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, src, out, 10, clk); hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, src, out, iters, clk);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
	const double mf = (double)blocks * 4 * iters * 24;
	printf("%-64s %8.2f ms  %7.1f TFLOP/s  shader clock %.2f GHz\n", what, ms, mf * 32768.0 / (ms * 1e-3) / 1e12, (double)h[0] / (double)h[1] * 0.1);
	hipFree(out); hipFree(clk);
}
int main() {
	const size_t n = 64 * 16;      // 16 fragments of 16 bytes per lane
	u32x4* h = (u32x4*)malloc(n * sizeof(u32x4)); u32x4* d_const; u32x4* d_rand; u32x4* d_desc;
	hipMalloc(&d_const, n * sizeof(u32x4)); hipMalloc(&d_rand, n * sizeof(u32x4)); hipMalloc(&d_desc, n * sizeof(u32x4));
	for (size_t i = 0; i < n; ++i) h[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};           // all 1.0
	hipMemcpy(d_const, h, n * sizeof(u32x4), hipMemcpyHostToDevice);
	srand(1);
	auto rbf = [] { const unsigned m = rand() & 0x7f, e = 120 + (rand() & 7), s = rand() & 1; return (s << 15) | (e << 7) | m; };   // random sign / exponent / mantissa
	for (size_t i = 0; i < n; ++i) { unsigned w[4]; for (int q = 0; q < 4; ++q) w[q] = rbf() | (rbf() << 16); h[i] = u32x4{w[0], w[1], w[2], w[3]}; }
	hipMemcpy(d_rand, h, n * sizeof(u32x4), hipMemcpyHostToDevice);
	auto pbf = [] { const unsigned m = rand() & 0x7f, e = 128 + (rand() & 3); return (e << 7) | m; };                              // positive, narrow exponent range (descriptor-like)
	for (size_t i = 0; i < n; ++i) { unsigned w[4]; for (int q = 0; q < 4; ++q) w[q] = pbf() | (pbf() << 16); h[i] = u32x4{w[0], w[1], w[2], w[3]}; }
	hipMemcpy(d_desc, h, n * sizeof(u32x4), hipMemcpyHostToDevice);
	for (int wg : {1, 3}) {
		char buf[128];
		snprintf(buf, sizeof buf, "%d workgroup(s) of 4 waves per CU, constant operands", wg); run(d_const, 256 * wg, 120000 / wg, buf);
		snprintf(buf, sizeof buf, "%d workgroup(s) of 4 waves per CU, descriptor-like positive operands", wg); run(d_desc, 256 * wg, 120000 / wg, buf);
		snprintf(buf, sizeof buf, "%d workgroup(s) of 4 waves per CU, random-sign operands", wg); run(d_rand, 256 * wg, 120000 / wg, buf);
	}
	return 0;
}
