// Does a wave64 ds_add_f32 whose lanes hit the same LDS address apply the additions in ASCENDING LANE ORDER?
// (fp32 addition does not associate: the order is visible in the bits.)  Random values of widely varying magnitude, random
// lane -> address maps (1 .. 64 distinct addresses over 1 .. 4 banks' worth of strides), partial exec masks; the LDS result is
// compared with the sequential sum in lane order and, for contrast, in descending order.
//   hipcc --offload-arch=gfx950 -O2 -o lds_atomic_order lds_atomic_order.hip && ./lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

__global__ void k(const float* val, const int* addr, const unsigned char* act, float* out, int trials, int rounds) {
	__shared__ float s[256];
	const int lane = threadIdx.x;
	for (int t = blockIdx.x; t < trials; t += gridDim.x) {
		for (int i = lane; i < 256; i += 64) s[i] = 0.f;
		__syncthreads();
		for (int r = 0; r < rounds; ++r) {
			const long long e = ((long long)t * rounds + r) * 64 + lane;
			if (act[e]) atomicAdd(&s[addr[e]], val[e]);       // ds_add_f32 (no return)
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
		}
		__syncthreads();
		for (int i = lane; i < 256; i += 64) out[(long long)t * 256 + i] = s[i];
		__syncthreads();
	}
}

int main() {
	const int trials = 20000, rounds = 6;
	const size_t n = (size_t)trials * rounds * 64;
	std::vector<float> val(n); std::vector<int> addr(n); std::vector<unsigned char> act(n);
	srand(12345);
	for (int t = 0; t < trials; ++t) {
		const int naddr = 1 + rand() % 64, stride = 1 << (rand() % 3), base = rand() % 8;
		const int mode = rand() % 4;
		for (int r = 0; r < rounds; ++r)
			for (int l = 0; l < 64; ++l) {
				const size_t e = ((size_t)t * rounds + r) * 64 + l;
				const int a = mode == 0 ? (rand() % naddr) : mode == 1 ? (l / (64 / naddr > 0 ? 64 / naddr : 1)) % naddr : mode == 2 ? (l % naddr) : ((l * 7 + r) % naddr);
				addr[e] = (base + a * stride) & 255;
				const int ex = (t % 50 == 0) ? -(135 + rand() % 14) : rand() % 24 - 12;     // every 50th trial: sums in the denormal range
				val[e] = ldexpf((float)(rand() % 16777216 + 1) / 16777216.f, ex);       // positive, 24 significant bits, 2^-12 .. 2^12
				act[e] = (rand() % 10) != 0;
			}
	}
	float *dv, *dout; int* da; unsigned char* dc;
	hipMalloc(&dv, n * 4); hipMalloc(&da, n * 4); hipMalloc(&dc, n); hipMalloc(&dout, (size_t)trials * 256 * 4);
	hipMemcpy(dv, val.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(da, addr.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, act.data(), n, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, dv, da, dc, dout, trials, rounds);
	std::vector<float> out((size_t)trials * 256);
	if (hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
	long asc_bad = 0, desc_same = 0, cells = 0, multi = 0, den_cells = 0, den_bad = 0;
	for (int t = 0; t < trials; ++t) {
		float up[256], dn[256]; int cnt[256];
		memset(up, 0, sizeof up); memset(dn, 0, sizeof dn); memset(cnt, 0, sizeof cnt);
		for (int r = 0; r < rounds; ++r) {
			for (int l = 0; l < 64; ++l) { const size_t e = ((size_t)t * rounds + r) * 64 + l; if (act[e]) { volatile float v = up[addr[e]] + val[e]; up[addr[e]] = v; ++cnt[addr[e]]; } }
			for (int l = 63; l >= 0; --l) { const size_t e = ((size_t)t * rounds + r) * 64 + l; if (act[e]) { volatile float v = dn[addr[e]] + val[e]; dn[addr[e]] = v; } }
		}
		for (int i = 0; i < 256; ++i) {
			if (!cnt[i]) continue;
			++cells; multi += cnt[i] > rounds;
			if (t % 50 == 0) { ++den_cells; den_bad += memcmp(&up[i], &out[(size_t)t * 256 + i], 4) != 0; continue; }
			if (memcmp(&up[i], &out[(size_t)t * 256 + i], 4) != 0) ++asc_bad;
			if (memcmp(&dn[i], &out[(size_t)t * 256 + i], 4) == 0) ++desc_same;
		}
	}
	printf("denormal-range cells %ld, differ from the host's gradual-underflow sums: %ld\n", den_cells, den_bad);
	printf("cells %ld (with same-address conflicts inside an instruction: %ld): differ from ASCENDING lane order: %ld ; equal to descending order: %ld\n", cells, multi, asc_bad, desc_same);
	return asc_bad ? 1 : 0;
}
