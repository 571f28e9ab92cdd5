// pyramid.hip -- source image -> working image -> octave bases -> fused scale-space kernel.
//
// Replaces, for a whole batch of images per launch:
//   resize<float>/resize_bilinear          lib/imgproc.cc:22-80,319-326
//   rgb2grey                               lib/imgproc.cc:237-249
//   GaussianBlur::blur<float> (6 sigmas)   feature/gaussian.hh:30-91, feature/dog.cc:53-57
//   GaussianPyramid::cal_mag_ort           feature/dog.cc:60-94 (+ fast_atan :22-37)
//   DOGSpace::diff                         feature/dog.cc:116-129
//
// Numerics: fp32 multiply and add kept separate and in the reference's order (this TU is built
// with -ffp-contract=off), so every plane is bit-identical to the CPU path.
#include "internal.hpp"
#include "devmath.hpp"

namespace {

// lib/imgproc.cc:32-44: source index and weight of destination index d
__device__ __forceinline__ void resize_coord(int d, float inv_f, int srcn, int& s, float& r) {
	float rr = ((float)d + 0.5f) * inv_f - 0.5f;
	int ss = (int)floorf(rr);
	rr -= (float)ss;
	if (ss < 0) { ss = 0; rr = 0.f; }
	else if (ss + 1 >= srcn) { ss = srcn - 2; rr = 1.f; }
	s = ss; r = rr;
}

__device__ __forceinline__ float bilerp(float p00, float p01, float p10, float p11, float rx, float irx, float ry, float iry) {
	// lib/imgproc.cc:74-75 (x = row weight, y = column weight there)
	return rx * (p11 * ry + p10 * iry) + irx * (p01 * ry + p00 * iry);
}

// ---- K1: source (H x W x 3) -> working image (wh x ww x 3), feature/feature.cc:33-35 ----
__global__ void __launch_bounds__(256) k_resize_to_work(SiftPlan p) {
	const int img = blockIdx.z;
	const int row = blockIdx.y;
	const int col = blockIdx.x * 256 + threadIdx.x;
	if (col >= p.ww) return;
	const float fx = (float)p.wh / (float)p.sh, fy = (float)p.ww / (float)p.sw;
	const float ifx = 1.f / fx, ify = 1.f / fy;
	int sx, sy; float rx, ry;
	resize_coord(row, ifx, p.sh, sx, rx);
	resize_coord(col, ify, p.sw, sy, ry);
	const float irx = 1.0f - rx, iry = 1.0f - ry;
	const float* src = p.srcs[img];
	const float* p0 = src + ((long long)sx * p.sw + sy) * 3;
	const float* p1 = p0 + (long long)p.sw * 3;
	float* dst = p.work + (((long long)img * p.wh + row) * p.ww + col) * 3;
#pragma unroll
	for (int c = 0; c < 3; ++c)
		dst[c] = bilerp(p0[c], p0[3 + c], p1[c], p1[3 + c], rx, irx, ry, iry);
}

// ---- K2: working image -> grey base of every octave (feature/dog.cc:96-114, :48-51) ----
__global__ void __launch_bounds__(256) k_octave_grey(SiftPlan p) {
	const int img = blockIdx.z;
	const int o = blockIdx.y;
	const OctDesc od = p.oct[o];
	const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
	if (idx >= od.plane) return;
	const int row = (int)(idx / od.w), col = (int)(idx % od.w);
	const float* work = p.work + (long long)img * p.wh * p.ww * 3;
	float r, g, b;
	if (o == 0) {
		const float* q = work + idx * 3;
		r = q[0]; g = q[1]; b = q[2];
	} else {
		const float fx = (float)od.h / (float)p.wh, fy = (float)od.w / (float)p.ww;
		const float ifx = 1.f / fx, ify = 1.f / fy;
		int sx, sy; float rx, ry;
		resize_coord(row, ifx, p.wh, sx, rx);
		resize_coord(col, ify, p.ww, sy, ry);
		const float irx = 1.0f - rx, iry = 1.0f - ry;
		const float* p0 = work + ((long long)sx * p.ww + sy) * 3;
		const float* p1 = p0 + (long long)p.ww * 3;
		r = bilerp(p0[0], p0[3], p1[0], p1[3], rx, irx, ry, iry);
		g = bilerp(p0[1], p0[4], p1[1], p1[4], rx, irx, ry, iry);
		b = bilerp(p0[2], p0[5], p1[2], p1[5], rx, irx, ry, iry);
	}
	float* ws = p.ws + (long long)img * p.ws_stride;
	ws[plane_off_grey(od) + idx] = (r + g + b) / 3.f;   // lib/imgproc.cc:245
}

// ---- K3: fused scale space -----------------------------------------------------------
// One 256-thread workgroup owns a TW x TH tile of one octave and walks the 6 sigmas.  Per
// sigma: separable blur of the *unblurred* grey tile (column pass, then row pass, replicate
// borders -- feature/gaussian.hh:43-89) staged through LDS with register-blocked sliding
// windows, then |DoG|, gradient magnitude and orientation straight to HBM.  The Gaussian planes
// themselves only ever exist in LDS (two ping-pong buffers).
constexpr int TW = OP_PYR_TW, TH = OP_PYR_TH;
constexpr int GR = TH + 2, GC = TW + 2;       // blurred region incl. the 1-px gradient halo
constexpr int PG = GC + 1;                    // pitch of G buffers (odd: lanes along rows are conflict-free)
constexpr int PV = GR + 1;                    // pitch of the transposed column-pass buffer

// register-blocked passes for kernel half-width C (taps = 2C+1)
template <int C, int RV>
__device__ __forceinline__ void vpass_blocked(const float* __restrict__ In, float* __restrict__ VT,
		const float* __restrict__ kern /* center at [0] */, int NC, int pin, int halo, int tid) {
	// thread -> (column c, strip of RV rows)
	const int strips = (GR + RV - 1) / RV;
	if (tid >= NC * strips) return;
	const int c = tid % NC, st = tid / NC;
	const int r0 = st * RV;
	float win[RV + 2 * C];
#pragma unroll
	for (int i = 0; i < RV + 2 * C; ++i) {
		int rr = r0 + i + (halo - C);           // In row of G row (r0 + i - C)
		rr = rr < GR + 2 * halo ? rr : GR + 2 * halo - 1;   // tail strip over-reads stay in bounds
		win[i] = In[rr * pin + c];
	}
	float kw[2 * C + 1];
#pragma unroll
	for (int k = 0; k < 2 * C + 1; ++k) kw[k] = kern[k - C];
#pragma unroll
	for (int r = 0; r < RV; ++r) {
		float tmp = 0.f;
#pragma unroll
		for (int k = 0; k < 2 * C + 1; ++k) tmp += win[r + k] * kw[k];
		if (r0 + r < GR) VT[c * PV + r0 + r] = tmp;
	}
}

template <int C, int RH>
__device__ __forceinline__ void hpass_blocked(const float* __restrict__ VT, float* __restrict__ G,
		const float* __restrict__ kern, int NC, int halo, int tid) {
	const int strips = (GC + RH - 1) / RH;
	if (tid >= GR * strips) return;
	const int r = tid % GR, st = tid / GR;
	const int g0 = st * RH;
	float win[RH + 2 * C];
#pragma unroll
	for (int i = 0; i < RH + 2 * C; ++i) {
		int cc = g0 + i + (halo - C);
		cc = cc < NC ? cc : NC - 1;
		win[i] = VT[cc * PV + r];
	}
	float kw[2 * C + 1];
#pragma unroll
	for (int k = 0; k < 2 * C + 1; ++k) kw[k] = kern[k - C];
#pragma unroll
	for (int g = 0; g < RH; ++g) {
		float tmp = 0.f;
#pragma unroll
		for (int k = 0; k < 2 * C + 1; ++k) tmp += win[g + k] * kw[k];
		if (g0 + g < GC) G[r * PG + g0 + g] = tmp;
	}
}

// generic (any half-width) fall-backs, one output per thread-iteration
__device__ __forceinline__ void vpass_generic(const float* In, float* VT, const float* kern, int C,
		int NC, int pin, int halo, int tid) {
	for (int e = tid; e < NC * GR; e += 256) {
		const int c = e % NC, r = e / NC;
		float tmp = 0.f;
		for (int k = -C; k <= C; ++k) tmp += In[(r + halo + k) * pin + c] * kern[k];
		VT[c * PV + r] = tmp;
	}
}
__device__ __forceinline__ void hpass_generic(const float* VT, float* G, const float* kern, int C,
		int halo, int tid) {
	for (int e = tid; e < GR * GC; e += 256) {
		const int r = e % GR, g = e / GR;
		float tmp = 0.f;
		for (int k = -C; k <= C; ++k) tmp += VT[(g + halo + k) * PV + r] * kern[k];
		G[r * PG + g] = tmp;
	}
}

__global__ void __launch_bounds__(256) k_pyramid(SiftPlan p) {
	extern __shared__ __attribute__((aligned(16))) float smem[];
	const int img = blockIdx.y;
	const int tile = blockIdx.x;
	int o = 0;
	while (o + 1 < p.noct && tile >= p.oct[o + 1].tile_begin) ++o;
	const OctDesc od = p.oct[o];
	const int t = tile - od.tile_begin;
	const int tx = t % od.tiles_x, ty = t / od.tiles_x;
	const int x0 = tx * TW, y0 = ty * TH;
	const int halo = p.halo;
	const int NR = GR + 2 * halo, NC = GC + 2 * halo, pin = NC;
	float* In = smem;                       // NR x NC
	float* VT = In + NR * pin;              // NC x PV (column-pass result, transposed)
	float* Ga = VT + NC * PV;               // GR x PG
	float* Gb = Ga + GR * PG;
	const int tid = threadIdx.x;
	const int ns = p.nscale;
	float* ws = p.ws + (long long)img * p.ws_stride;
	const float* grey = ws + plane_off_grey(od);

	// stage the grey tile (+halo), replicate-clamped: clamping the *load* reproduces the
	// reference's border padding in both passes
	for (int e = tid; e < NR * NC; e += 256) {
		const int r = e / NC, c = e % NC;
		int yy = y0 - 1 - halo + r, xx = x0 - 1 - halo + c;
		yy = yy < 0 ? 0 : (yy > od.h - 1 ? od.h - 1 : yy);
		xx = xx < 0 ? 0 : (xx > od.w - 1 ? od.w - 1 : xx);
		In[r * pin + c] = grey[(long long)yy * od.w + xx];
	}
	__syncthreads();

	float* Gcur = Ga;
	float* Gprev = Gb;
	for (int s = 1; s < ns; ++s) {
		const int C = p.kcenter[s];
		const float* kern = &p.kern[s][OP_MAX_KCENTER];
		if (halo == 6 && C == 3) vpass_blocked<3, 12>(In, VT, kern, NC, pin, halo, tid);
		else if (halo == 6 && C == 6) vpass_blocked<6, 12>(In, VT, kern, NC, pin, halo, tid);
		else vpass_generic(In, VT, kern, C, NC, pin, halo, tid);
		__syncthreads();
		if (halo == 6 && C == 3) hpass_blocked<3, 10>(VT, Gcur, kern, NC, halo, tid);
		else if (halo == 6 && C == 6) hpass_blocked<6, 10>(VT, Gcur, kern, NC, halo, tid);
		else hpass_generic(VT, Gcur, kern, C, halo, tid);
		__syncthreads();

		float* dog = ws + plane_off_dog(od, s - 1);
		const bool want_grad = (s <= ns - 3);
		float* mag = ws + plane_off_mag(od, ns, want_grad ? s : 1);
		float* ort = ws + plane_off_ort(od, ns, want_grad ? s : 1);
#pragma unroll 2
		for (int e = tid; e < TW * TH; e += 256) {
			const int ly = e / TW, lx = e % TW;
			const int y = y0 + ly, x = x0 + lx;
			if (y >= od.h || x >= od.w) continue;
			const int gy = ly + 1, gx = lx + 1;
			const float cur = Gcur[gy * PG + gx];
			const float prev = (s == 1) ? In[(gy + halo) * pin + gx + halo] : Gprev[gy * PG + gx];
			const long long gi = (long long)y * od.w + x;
			dog[gi] = fabsf(prev - cur);                     // feature/dog.cc:126
			if (want_grad) {
				float m = 0.f, a = (float)3.14159265358979323846;
				if (x >= 1 && x <= od.w - 2 && y >= 1 && y <= od.h - 2) {   // feature/dog.cc:76-90
					const float dy = Gcur[(gy + 1) * PG + gx] - Gcur[(gy - 1) * PG + gx];
					const float dx = Gcur[gy * PG + gx + 1] - Gcur[gy * PG + gx - 1];
					m = opdev::hypotf_glibc(dx, dy);
					a = opdev::fast_atan_plus_pi(dy, dx);
				}
				mag[gi] = m;
				ort[gi] = a;
			}
		}
		float* tmp = Gprev; Gprev = Gcur; Gcur = tmp;
		// no barrier needed here: the next column pass only writes VT (its readers are past the
		// barrier above) and the next row pass writes the buffer that was Gprev only after the
		// barrier that follows the column pass.
	}
}

__global__ void k_debug_math(int which, const float* x, const float* y, int n, float* out) {
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float r;
	switch (which) {
		case 0: r = opdev::expf_glibc(x[i]); break;
		case 1: r = opdev::cosf_glibc(x[i]); break;
		case 2: r = opdev::sinf_glibc(x[i]); break;
		case 3: r = opdev::hypotf_glibc(x[i], y[i]); break;
		default: r = opdev::fast_atan_plus_pi(y[i], x[i]); break;
	}
	out[i] = r;
}

}	// namespace

size_t pyramid_lds_bytes(int halo) {
	const int NR = GR + 2 * halo, NC = GC + 2 * halo;
	return sizeof(float) * ((size_t)NR * NC + (size_t)NC * PV + 2 * (size_t)GR * PG);
}

hipError_t launch_resize_to_work(const SiftPlan& p, hipStream_t st) {
	dim3 grid((p.ww + 255) / 256, p.wh, p.n);
	hipLaunchKernelGGL(k_resize_to_work, grid, dim3(256), 0, st, p);
	return hipGetLastError();
}

hipError_t launch_octave_grey(const SiftPlan& p, hipStream_t st) {
	dim3 grid((unsigned)((p.oct[0].plane + 255) / 256), p.noct, p.n);
	hipLaunchKernelGGL(k_octave_grey, grid, dim3(256), 0, st, p);
	return hipGetLastError();
}

hipError_t launch_pyramid(const SiftPlan& p, hipStream_t st) {
	static bool attr_set = false;
	size_t lds = pyramid_lds_bytes(p.halo);
	if (!attr_set) {
		hipError_t e = hipFuncSetAttribute((const void*)k_pyramid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
		if (e != hipSuccess) return e;
		attr_set = true;
	}
	dim3 grid(p.total_tiles, p.n);
	hipLaunchKernelGGL(k_pyramid, grid, dim3(256), lds, st, p);
	return hipGetLastError();
}

hipError_t launch_debug_math(int which, const float* x, const float* y, int n, float* out, hipStream_t st) {
	hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, st, which, x, y, n, out);
	return hipGetLastError();
}
