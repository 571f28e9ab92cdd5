// pyramid.hip -- source image -> grey octave bases -> fused scale-space + extrema-scan kernel.
//
// Replaces, for a whole batch of images per launch:
//   resize<float>/resize_bilinear          lib/imgproc.cc:22-80,319-326 (working image and octaves)
//   read_img's byte conversion             lib/imgio.cc:54-56,75-77     (OP_U8 sources)
//   rgb2grey                               lib/imgproc.cc:237-249
//   GaussianBlur::blur<float> (6 sigmas)   feature/gaussian.hh:30-91, feature/dog.cc:53-57
//   DOGSpace::diff                         feature/dog.cc:116-129
//   ExtremaDetector::get_local_raw_extrema feature/extrema.cc:170-216
// GaussianPyramid::cal_mag_ort (feature/dog.cc:60-94) is not materialised: the orientation and
// descriptor kernels evaluate it on the Gaussian planes for the samples they read; the debug
// dump computes the planes with k_magort_plane.
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

// ---- K1: source (H x W x 3) -> grey base of every octave, in one pass ------------------------
// The reference resizes the source to the working image (feature/feature.cc:33-35), then resizes
// that *working RGB image* once per octave (feature/dog.cc:105-110) and greys each result
// (lib/imgproc.cc:237-249).  Here a workgroup computes a WT x WR tile of the working image (+1
// row / column for the bilinear taps) from the source straight into LDS, greys it for octave 0
// and emits every pixel of octaves 1.. whose bilinear footprint starts inside the tile (each
// octave pixel has exactly one such tile).  The working image never goes to HBM (the staged dump
// asks for it with write_work); every value is computed by the reference's formulas, so the grey
// planes are bit-identical to the two-pass result.
// 64 x 14: the (WR + 1) x (WT + 1) = 975 tile elements are 3.8 per thread -- four rounds of 256 lanes with 5 % of them idle; at
// 64 x 16 (1105 elements) the fifth round ran 81 lanes of 256, and its 12 more load registers cost two resident wavefronts per
// SIMD (82 -> 70 VGPRs: 5 -> 7).  Measured 12 / 14 / 16 / 20 / 22 rows: 0.1742 / 0.1634 / 0.1713 / 0.1770 / 0.1764 ms
// (profiles/r06_grey_tile_rows.txt).  (Walking only the run of octave candidates that belong to the tile -- the rectangle
// below is conservative: 10 x 35 candidates for 7 x 32 pixels at octave 1 -- was measured too: the threads that find the run's
// ends between the two barriers cost more than the round of 256 lanes they save, 0.1683 -> 0.1736 ms.)
#ifndef OP_GREY_WR
#define OP_GREY_WR 14
#endif
constexpr int WT = 64, WR = OP_GREY_WR;       // working-image tile
constexpr int WP = WT + 1 + 2;                // LDS pitch (WT + 1 columns used)

// six consecutive source elements from i (element-aligned only) -- fp32 as they are: one 16-byte and one 8-byte load; or
// decoder bytes converted like read_img (lib/imgio.cc:54-56,75-77)
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
// (the sources arrive through a pointer table, i.e. as generic pointers: they are device memory, and said to be -- a flat
// load counts on the LDS counter too, and the LDS reads of the coordinate tables would wait for every source load before them)
#define OP_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ void src_run6(const float* s, long long i, const float*, float (&t)[6]) {
	const OP_GLOBAL float* g = (const OP_GLOBAL float*)s + i;
	const f32x4_a4 a = *(const OP_GLOBAL f32x4_a4*)g; const f32x2_a4 b = *(const OP_GLOBAL f32x2_a4*)(g + 4);
	t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; t[4] = b.x; t[5] = b.y;
}
__device__ __forceinline__ void src_run6(const unsigned char* s, long long i, const float* lut, float (&t)[6]) {
#pragma unroll
	for (int k = 0; k < 6; ++k) t[k] = lut[((const OP_GLOBAL unsigned char*)s)[i + k]];
}

// s / 3.f (lib/imgproc.cc:245) as (float)((double)s * (1.0 / 3.0)): the double product is within 2^-52 of s / 3, and s / 3
// is never that close to a rounding boundary of fp32 (3 * midpoint is an odd 26..27-bit integer multiple of the grid, no fp32
// number), so this IS the correctly rounded quotient -- three VALU operations instead of the ten of an IEEE fp32 division.
__device__ __forceinline__ float third(float s) { return (float)((double)s * (1.0 / 3.0)); }

constexpr int TR = 24, TC = 72;               // per-workgroup coordinate tables (rows, columns)

#ifdef OP_GREY_WAVES            // A/B knob (scripts/build_variant.sh): resident wavefronts per SIMD the register allocation is held to
#define OP_GREY_ATTR __attribute__((amdgpu_waves_per_eu(OP_GREY_WAVES, OP_GREY_WAVES)))
#else
#define OP_GREY_ATTR
#endif
template <typename SrcT>
__global__ void __launch_bounds__(256) OP_GREY_ATTR k_grey_octaves(SiftPlan p, int write_work) {
	__shared__ float s_rgb[3][(WR + 1) * WP];
	__shared__ float s_lut[256];
	// resize_coord() is separable: the (source index, weight) of every row and of every column of the tile is computed
	// once per workgroup and read back per pixel (the kernel is bound by VALU issue, and the two resize_coord calls
	// were a third of a pixel's instructions)
	__shared__ int s_ri[TR], s_ci[TC];
	__shared__ float s_rw[TR], s_cw[TC];
	__shared__ long long s_ro[TR];            // working tile: element offset of the source row
	// Workgroup b runs on XCD b % 8 (observed; a speed assumption only), and neighbouring tiles read the same source
	// rows / columns at their seams: every XCD takes a CONTIGUOUS eighth of the row-major tile order, so a seam's second
	// reader finds the lines in its own L2 instead of fetching them again through another one
	const unsigned ntx = (unsigned)(p.ww + WT - 1) / WT, nty = (unsigned)(p.wh + WR - 1) / WR;
	const unsigned ntile = ntx * nty * (unsigned)p.n, per = (ntile + 7u) >> 3;
	const unsigned lin = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
	if (lin >= ntile) return;
	const int img = (int)(lin / (ntx * nty));
	const unsigned trem = lin - (unsigned)img * (ntx * nty);
	const int tx0 = (int)(trem % ntx) * WT, ty0 = (int)(trem / ntx) * WR;
	const int tid = threadIdx.x;
	const SrcT* src = (const SrcT*)p.srcs[img];
	// the step's counters (raw / refined / oriented per image, total) start at zero: cleared here, by the first kernel
	// of the step, instead of by a fill launch of their own (nothing reads or adds to them before this kernel has ended)
	if (blockIdx.x == 0 && p.zero)
		for (int i = tid; i < p.zero_n; i += 256) p.zero[i] = 0;
	if (sizeof(SrcT) == 1) s_lut[tid] = (float)((double)(float)tid / 255.0);      // (float)byte / 255.0: float -> double, IEEE division, round to float
	// working-image tile: lib/imgproc.cc:22-80 on the source
	{
		const float fx = (float)p.wh / (float)p.sh, fy = (float)p.ww / (float)p.sw;
		const float ifx = 1.f / fx, ify = 1.f / fy;
		if (tid < WR + 1) {
			int sx = -1; float rx = 0.f;
			if (ty0 + tid < p.wh) resize_coord(ty0 + tid, ifx, p.sh, sx, rx);
			s_ri[tid] = sx; s_rw[tid] = rx; s_ro[tid] = (long long)sx * p.sw * 3;
		} else if (tid >= 64 && tid < 64 + WT + 1) {
			const int c = tid - 64;
			int sy = -1; float ry = 0.f;
			if (tx0 + c < p.ww) resize_coord(tx0 + c, ify, p.sw, sy, ry);
			s_ci[c] = sy < 0 ? -1 : sy * 3; s_cw[c] = ry;
		}
		__syncthreads();
		// Every thread owns NE elements of the (WR + 1) x (WT + 1) tile.  All of their source runs are requested first -- the
		// 2 x 2 taps of one working pixel are two runs of 6 consecutive source elements (two RGB pixels of row sx and of
		// row sx + 1), fetched as one 16-byte and one 8-byte load each; neighbouring lanes' runs are 3-6 elements apart, so a
		// wavefront's load covers a contiguous kilobyte -- and only then interpolated: 20 loads in flight per thread instead
		// of 4 (the kernel streams the source once and does little else; what it needs is outstanding bytes)
		constexpr int NE = ((WR + 1) * (WT + 1) + 255) / 256;
		float t0[NE][6], t1[NE][6];
		bool live[NE];
#pragma unroll
		for (int k = 0; k < NE; ++k) {                   // no branch around the loads: an element without taps reads the image's first run and drops it
			const int e = tid + 256 * k < (WR + 1) * (WT + 1) ? tid + 256 * k : 0;
			const int r = e / (WT + 1), c = e % (WT + 1);
			const int sx = s_ri[r], sy3 = s_ci[c];
			live[k] = sx >= 0 && sy3 >= 0;
			const long long i0 = live[k] ? s_ro[r] + sy3 : 0, i1 = i0 + (long long)p.sw * 3;
			src_run6(src, i0, s_lut, t0[k]); src_run6(src, i1, s_lut, t1[k]);
		}
#pragma unroll
		for (int k = 0; k < NE; ++k) {
			const int e = tid + 256 * k;
			if (e >= (WR + 1) * (WT + 1)) break;
			const int r = e / (WT + 1), c = e % (WT + 1);
			float v0 = 0.f, v1 = 0.f, v2 = 0.f;
			if (live[k]) {
				const float rx = s_rw[r], ry = s_cw[c];
				const float irx = 1.0f - rx, iry = 1.0f - ry;
				v0 = bilerp(t0[k][0], t0[k][3], t1[k][0], t1[k][3], rx, irx, ry, iry);
				v1 = bilerp(t0[k][1], t0[k][4], t1[k][1], t1[k][4], rx, irx, ry, iry);
				v2 = bilerp(t0[k][2], t0[k][5], t1[k][2], t1[k][5], rx, irx, ry, iry);
				if (write_work && r < WR && c < WT) {
					float* dst = p.work + (((long long)img * p.wh + ty0 + r) * p.ww + tx0 + c) * 3;
					dst[0] = v0; dst[1] = v1; dst[2] = v2;
				}
			}
			s_rgb[0][r * WP + c] = v0; s_rgb[1][r * WP + c] = v1; s_rgb[2][r * WP + c] = v2;
		}
	}
	__syncthreads();
	float* ws = p.ws + (long long)img * p.ws_stride;
	// octave 0: grey of the working tile (lib/imgproc.cc:245); thread = column, rows dealt to the four waves
	{
		const int c = tid & (WT - 1), col = tx0 + c;
		float* g0 = ws + plane_off_grey(p.oct[0]) + (long long)ty0 * p.ww + col;
		if (col < p.ww)
			for (int r = tid >> 6; r < WR && ty0 + r < p.wh; r += 4)
				g0[(long long)r * p.ww] = third(s_rgb[0][r * WP + c] + s_rgb[1][r * WP + c] + s_rgb[2][r * WP + c]);
	}
	// octaves 1..: pixels whose top-left tap (sx, sy) lies in this tile
	for (int o = 1; o < p.noct; ++o) {
		const OctDesc od = p.oct[o];
		const float fx = (float)od.h / (float)p.wh, fy = (float)od.w / (float)p.ww;
		const float ifx = 1.f / fx, ify = 1.f / fy;
		// conservative candidate rectangle (membership is decided exactly below)
		int r_lo = (int)floorf(((float)ty0 + 0.5f) * fx - 0.5f) - 1, r_hi = (int)ceilf(((float)(ty0 + WR) + 0.5f) * fx - 0.5f) + 2;
		int c_lo = (int)floorf(((float)tx0 + 0.5f) * fy - 0.5f) - 1, c_hi = (int)ceilf(((float)(tx0 + WT) + 0.5f) * fy - 0.5f) + 2;
		r_lo = r_lo < 0 ? 0 : r_lo; c_lo = c_lo < 0 ? 0 : c_lo;
		r_hi = r_hi > od.h ? od.h : r_hi; c_hi = c_hi > od.w ? od.w : c_hi;
		const int nr = r_hi - r_lo, nc = c_hi - c_lo;
		if (nr <= 0 || nc <= 0) continue;                   // uniform over the workgroup
		const bool tables = nr <= TR && nc <= TC;           // always, as long as an octave is not larger than the working image
		__syncthreads();                                    // the previous user of the tables is done
		if (tables) {
			if (tid < nr) {
				int sx; float rx;
				resize_coord(r_lo + tid, ifx, p.wh, sx, rx);
				const int lr = sx - ty0;
				s_ri[tid] = (lr >= 0 && lr < WR) ? lr : -1; s_rw[tid] = rx;
			} else if (tid >= 64 && tid - 64 < nc) {
				int sy; float ry;
				resize_coord(c_lo + tid - 64, ify, p.ww, sy, ry);
				const int lc = sy - tx0;
				s_ci[tid - 64] = (lc >= 0 && lc < WT) ? lc : -1; s_cw[tid - 64] = ry;
			}
		}
		__syncthreads();
		float* go = ws + plane_off_grey(od);
		const float inv_nc = 1.0f / (float)nc;            // e / nc through a float reciprocal (e < 2^16): an integer division costs ~40 VALU
		for (int e = tid; e < nr * nc; e += 256) {
			int q = (int)(((float)e + 0.5f) * inv_nc), rem = e - q * nc;
			if (rem < 0) { --q; rem += nc; } else if (rem >= nc) { ++q; rem -= nc; }
			const int dr = r_lo + q, dc = c_lo + rem;
			int lr, lc; float rx, ry;
			if (tables) { lr = s_ri[q]; lc = s_ci[rem]; rx = s_rw[q]; ry = s_cw[rem]; }
			else {
				int sx, sy;
				resize_coord(dr, ifx, p.wh, sx, rx);
				resize_coord(dc, ify, p.ww, sy, ry);
				lr = sx - ty0; lc = sy - tx0;
				if (lr >= WR) lr = -1;
				if (lc >= WT) lc = -1;
			}
			if (lr < 0 || lc < 0) continue;
			const float irx = 1.0f - rx, iry = 1.0f - ry;
			const int b = lr * WP + lc;
			const float r = bilerp(s_rgb[0][b], s_rgb[0][b + 1], s_rgb[0][b + WP], s_rgb[0][b + WP + 1], rx, irx, ry, iry);
			const float g = bilerp(s_rgb[1][b], s_rgb[1][b + 1], s_rgb[1][b + WP], s_rgb[1][b + WP + 1], rx, irx, ry, iry);
			const float bl = bilerp(s_rgb[2][b], s_rgb[2][b + 1], s_rgb[2][b + WP], s_rgb[2][b + WP + 1], rx, irx, ry, iry);
			go[(long long)dr * od.w + dc] = third(r + g + bl);   // lib/imgproc.cc:245
		}
	}
}

// ---- K3: fused scale space + extrema scan: tile geometry (kernel and its description below) ----
constexpr int TW = OP_PYR_TW, TH = OP_PYR_TH;
constexpr int GR = TH + 2, GC = TW + 2;       // blurred region incl. the 1-px gradient halo
constexpr int PG = TH == 16 ? 75 : GC + 1;    // pitch of G / DoG buffers: with RH = 6 the row pass's (row, strip) lanes hit 32 distinct banks
constexpr int PV = GR + (GR % 2 == 0 ? 1 : 2);  // pitch of the transposed column-pass buffer (odd -> conflict-free)
constexpr int NPX = TW * TH / 256;            // tile pixels owned by one thread (lx = tid & 63, ly = (tid >> 6) + 4 k)
constexpr int NHALO = 2 * GC + 2 * TH;        // ring elements of the G / DoG region around the tile
static_assert(TW == 64 && TH % 4 == 0 && NHALO <= 256, "tile shape");
// register blocking of the separable passes: strips of RV rows (column pass) / RH columns (row pass)
constexpr int RV = TH == 32 ? 12 : (TH == 24 ? 9 : 6);
constexpr int RH = TH == 32 ? 10 : (TH == 24 ? 8 : 6);   // TH 16: RH * PV = 114 = 18 (mod 32): strips of 18 rows tile the banks
static_assert(((GR + RV - 1) / RV) * (GC + 12) <= 256 && ((GC + RH - 1) / RH) * GR <= 256, "blocking must fit 256 threads at halo 6");

// rows of the transposed column-pass buffer: NC used + padding for the row pass's tail strip
__host__ __device__ constexpr int vt_rows(int halo) { return ((GC + RH - 1) / RH) * RH + 2 * halo; }

// N outputs y[r] = sum_k win[r+k] * kw[k], taps accumulated in order k = 0..2C with separate
// multiply and add (the reference's  tmp += line[i+k] * kernel[k],  gaussian.hh:63-64), two
// outputs per v_pk_mul_f32 / v_pk_add_f32.  The operand pair of outputs (2j, 2j+1) at tap k is
// (win[2j+k], win[2j+k+1]): even offsets come from the aligned pairs WA, odd ones from a copy of
// the window shifted by one element (WB), so that EVERY multiply-add is packed.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int C, int N>
__device__ __forceinline__ void packed_taps(const float (&win)[N + 2 * C], const float (&kw)[2 * C + 1], float (&out)[N]) {
	static_assert(N % 2 == 0, "packed_taps: even output count");
	constexpr int W = N + 2 * C;             // even
	f32x2 WA[W / 2], WB[W / 2];
#pragma unroll
	for (int i = 0; i < W / 2; ++i) { WA[i] = f32x2{win[2 * i], win[2 * i + 1]}; WB[i] = f32x2{win[2 * i + 1], 2 * i + 2 < W ? win[2 * i + 2] : 0.f}; }
	f32x2 acc[N / 2];
#pragma unroll
	for (int j = 0; j < N / 2; ++j) acc[j] = f32x2{0.f, 0.f};
#pragma unroll
	for (int k = 0; k < 2 * C + 1; ++k) {
		const f32x2 kk = f32x2{kw[k], kw[k]};
#pragma unroll
		for (int j = 0; j < N / 2; ++j) {
			const f32x2 w = (k % 2 == 0) ? WA[j + k / 2] : WB[j + (k - 1) / 2];
			const f32x2 prod = w * kk;
			acc[j] = acc[j] + prod;
		}
	}
#pragma unroll
	for (int j = 0; j < N / 2; ++j) { out[2 * j] = acc[j].x; out[2 * j + 1] = acc[j].y; }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's
// outstanding GLOBAL memory operations (s_waitcnt vmcnt(0)); the scale-space kernel streams its
// results to HBM and never reads them back, so waiting for those stores at each of its 19
// barriers only adds their write latency to every sigma step.
__device__ __forceinline__ void lds_barrier() {
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

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
		if (strips * RV != GR) rr = rr < GR + 2 * halo ? rr : GR + 2 * halo - 1;   // tail strip over-reads stay in bounds
		win[i] = In[rr * pin + c];
	}
	float kw[2 * C + 1];
#pragma unroll
	for (int k = 0; k < 2 * C + 1; ++k) kw[k] = kern[k - C];
	if constexpr (RV % 2 == 0) {
		float o[RV];
		packed_taps<C, RV>(win, kw, o);
#pragma unroll
		for (int r = 0; r < RV; ++r) if (r0 + r < GR) VT[c * PV + r0 + r] = o[r];
	} else {
#pragma unroll
		for (int r = 0; r < RV; ++r) {
			float tmp = 0.f;
#pragma unroll
			for (int k = 0; k < 2 * C + 1; ++k) tmp += win[r + k] * kw[k];
			if (r0 + r < GR) VT[c * PV + r0 + r] = tmp;
		}
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
		const int cc = g0 + i + (halo - C);     // tail strip over-reads land in VT's padding rows
		win[i] = VT[cc * PV + r];
	}
	float kw[2 * C + 1];
#pragma unroll
	for (int k = 0; k < 2 * C + 1; ++k) kw[k] = kern[k - C];
	if constexpr (RH % 2 == 0) {
		float o[RH];
		packed_taps<C, RH>(win, kw, o);
#pragma unroll
		for (int g = 0; g < RH; ++g) if (g0 + g < GC) G[r * PG + g0 + g] = o[g];
	} else {
#pragma unroll
		for (int g = 0; g < RH; ++g) {
			float tmp = 0.f;
#pragma unroll
			for (int k = 0; k < 2 * C + 1; ++k) tmp += win[g + k] * kw[k];
			if (g0 + g < GC) G[r * PG + g0 + g] = tmp;
		}
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

// Raw-extrema test of one tile pixel against the three DoG layers held in LDS
// (ExtremaDetector::get_local_raw_extrema, feature/extrema.cc:170-216)
__device__ __forceinline__ bool is_raw_extremum(const float* __restrict__ Dm, const float* __restrict__ D0,
		const float* __restrict__ Dp, int gy, int gx, float center, float judge) {
	bool mx = true, mn = true;
	const float cmp1 = center - judge, cmp2 = center + judge;
#pragma unroll
	for (int di = -1; di <= 1; ++di)
#pragma unroll
		for (int dj = -1; dj <= 1; ++dj) {
			if (di == 0 && dj == 0) continue;
			const float v = D0[(gy + di) * PG + gx + dj];
			if (v >= cmp1) mx = false;
			if (v <= cmp2) mn = false;
		}
	if (!mx && !mn) return false;
#pragma unroll
	for (int di = -1; di <= 1; ++di)
#pragma unroll
		for (int dj = -1; dj <= 1; ++dj) {
			const float v = Dm[(gy + di) * PG + gx + dj], u = Dp[(gy + di) * PG + gx + dj];
			if (v >= cmp1 || u >= cmp1) mx = false;
			if (v <= cmp2 || u <= cmp2) mn = false;
		}
	return mx || mn;
}

// K3: fused scale space + extrema scan.  One 256-thread workgroup owns a TW x TH tile of one
// octave and walks the sigmas.  Per sigma: separable blur of the *unblurred* grey tile (column
// pass, then row pass, replicate borders -- feature/gaussian.hh:43-89) through LDS with
// register-blocked sliding windows.  Then every thread handles its NPX fixed tile pixels (lanes
// along x: coalesced HBM rows) plus one element of the 1-px ring: |DoG| against the previous
// Gaussian value it kept in registers (feature/dog.cc:126) goes into a 3-deep LDS ring of DoG
// layers and to HBM (the sub-pixel refinement reads it), the Gaussian value goes to HBM when its
// gradients will be needed.  As soon as three consecutive DoG layers sit in LDS the middle one is
// scanned for raw extrema (feature/extrema.cc:170-216) -- nothing is re-read from HBM.
// HALO_CT > 0: halo known at compile time (6 for the shipped Gaussian bank: all index arithmetic is constant-folded);
// HALO_CT == 0: any halo (p.halo).
template <int HALO_CT>
__global__ void __launch_bounds__(256) k_pyramid(SiftPlan p, int* __restrict__ raw, int* __restrict__ raw_count, int cap) {
	extern __shared__ __attribute__((aligned(16))) float smem[];
	const int img = blockIdx.y;
	const int tile = blockIdx.x;
	int o = 0;
	while (o + 1 < p.noct && tile >= p.oct[o + 1].tile_begin) ++o;
	const OctDesc od = p.oct[o];
	const int t = tile - od.tile_begin;
	const int tx = t % od.tiles_x, ty = t / od.tiles_x;
	const int x0 = tx * TW, y0 = ty * TH;
	const int halo = HALO_CT > 0 ? HALO_CT : p.halo;
	const int NR = GR + 2 * halo, NC = GC + 2 * halo, pin = NC;
	float* In = smem;                       // NR x NC
	float* VT = In + NR * pin;              // vt_rows x PV (column-pass result, transposed)
	float* G = VT + vt_rows(halo) * PV;     // GR x PG current Gaussian
	float* Dr = G + GR * PG;                // 3 x GR x PG ring of DoG layers
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

	// this thread's fixed pixels
	const int lx = tid & 63, ly0 = tid >> 6;
	const int x = x0 + lx;
	const int lds0 = (ly0 + 1) * PG + lx + 1;                 // LDS offset of pixel k = lds0 + 4 k PG
	const unsigned gi0 = (unsigned)((y0 + ly0) * od.w + x);     // plane offset of pixel k = gi0 + 4 k w
	int nvalid = 0;                                             // pixels k < nvalid are inside the image
	if (x < od.w) { const int rows = od.h - (y0 + ly0); nvalid = rows <= 0 ? 0 : (rows + 3) / 4; nvalid = nvalid > NPX ? NPX : nvalid; }
	const bool xin = x >= 1 && x <= od.w - 2;                   // extrema.cc:212
	// one ring element: top row, bottom row, left column, right column of the GR x GC region
	int hoff = -1;
	if (tid < GC) hoff = tid;
	else if (tid < 2 * GC) hoff = (GR - 1) * PG + (tid - GC);
	else if (tid < 2 * GC + TH) hoff = (1 + tid - 2 * GC) * PG;
	else if (tid < NHALO) hoff = (1 + tid - 2 * GC - TH) * PG + GC - 1;
	lds_barrier();

	float prev[NPX], dprev[NPX], prev_h = 0.f;
#pragma unroll
	for (int k = 0; k < NPX; ++k) {
		prev[k] = In[(ly0 + 4 * k + 1 + halo) * pin + lx + 1 + halo];   // data[0]: the unblurred grey (dog.cc:53)
		dprev[k] = 0.f;
	}
	if (hoff >= 0) prev_h = In[(hoff / PG + halo) * pin + hoff % PG + halo];

	for (int s = 1; s < ns; ++s) {
		const int C = p.kcenter[s];
		const float* kern = &p.kern[s][OP_MAX_KCENTER];
		if (halo == 6 && C == 3) vpass_blocked<3, RV>(In, VT, kern, NC, pin, halo, tid);
		else if (halo == 6 && C == 6) vpass_blocked<6, RV>(In, VT, kern, NC, pin, halo, tid);
		else vpass_generic(In, VT, kern, C, NC, pin, halo, tid);
		lds_barrier();
		if (halo == 6 && C == 3) hpass_blocked<3, RH>(VT, G, kern, NC, halo, tid);
		else if (halo == 6 && C == 6) hpass_blocked<6, RH>(VT, G, kern, NC, halo, tid);
		else hpass_generic(VT, G, kern, C, halo, tid);
		lds_barrier();

		const int d = s - 1;                                 // DoG layer produced by this sigma
		float* Dcur = Dr + (d % 3) * (GR * PG);
		float* gauss = ws + plane_off_gauss(od, ns, s);      // the Gaussian stack is what reaches HBM (internal.hpp); |DoG| stays in LDS
		float dcen[NPX];                                     // layer d-1 at this thread's pixels (scan centre)
#pragma unroll
		for (int k = 0; k < NPX; ++k) {
			const float cur = G[lds0 + 4 * k * PG];
			const float dv = fabsf(prev[k] - cur);           // feature/dog.cc:126
			Dcur[lds0 + 4 * k * PG] = dv;
			if (k < nvalid) {
				const unsigned gi = gi0 + (unsigned)(4 * k) * (unsigned)od.w;
				gauss[gi] = cur;
			}
			prev[k] = cur;
			dcen[k] = dprev[k]; dprev[k] = dv;
		}
		if (hoff >= 0) {
			const float cur = G[hoff];
			Dcur[hoff] = fabsf(prev_h - cur);
			prev_h = cur;
		}
		lds_barrier();
		if (d >= 2 && xin) {                                 // layers d-2, d-1, d in LDS: scan d-1 (extrema.cc:42)
			const int j = d - 1;
			const float* D0 = Dr + (j % 3) * (GR * PG);
			const float* Dm = Dr + ((j + 2) % 3) * (GR * PG);   // layer j-1
			const float* Dp = Dcur;                              // layer j+1
#pragma unroll
			for (int k = 0; k < NPX; ++k) {
				const int y = y0 + ly0 + 4 * k;
				if (k >= nvalid || y < 1 || y > od.h - 2) continue;              // extrema.cc:212
				const float center = dcen[k];
				if (center < p.pre_color_thres) continue;                          // :179
				if (!is_raw_extremum(Dm, D0, Dp, ly0 + 4 * k + 1, lx + 1, center, p.judge_thres)) continue;
				const int slot = atomicAdd(&raw_count[img], 1);
				if (slot < cap) {
					int* q = raw + ((long long)img * cap + slot) * 4;
					q[0] = x; q[1] = y; q[2] = o; q[3] = j;
				}
			}
		}
		// no barrier needed here: the next column pass only writes VT (its readers are past the
		// barriers above); the next row pass overwrites G, whose readers (the loop above) are
		// separated from it by the barrier after the column pass; the next DoG layer overwrites
		// ring slot (d+1)%3 = layer d-2, last read by this scan, two barriers earlier.
	}
}

// ---- K3r: the same stage as K3, row-streaming form for the shipped Gaussian bank ---------------
// A 256-thread workgroup owns a band of RW_OWN columns x p.rw_seg rows of one octave and walks down the
// rows two at a time; nothing but two rows of column-pass results and four rows of DoG live in LDS.
//   column pass: thread = column.  It keeps the 14 grey rows around the current row pair in
//     registers (a sliding window fed by one coalesced load per row) and accumulates all six sigmas
//     of both rows from it.  Two sigmas share one packed accumulator (v_pk_mul_f32 / v_pk_add_f32
//     with the window element broadcast by op_sel and the two taps in an SGPR pair), so every
//     multiply and every add of the reference's  tmp += line[i+k] * kernel[k]  (gaussian.hh:63-64)
//     is one half of a packed instruction, in the reference's order; no LDS reads, no shifted copies.
//   row pass: thread = two adjacent columns of one of the two rows; the (sigma a, sigma b) pairs
//     written by the column pass are read back as 64-bit LDS words and used as packed operands as
//     they are.  All six Gaussian values of a pixel end up in ONE thread: |DoG| (dog.cc:126) is
//     register arithmetic (it only feeds the scan), the six Gaussian planes leave as 8-byte stores on coalesced rows.
//   scan: |DoG| rows go through a 4-row LDS ring.  Pixels passing the PRE_COLOR_THRES gate
//     (extrema.cc:179) are compacted into an LDS queue and the 26-neighbour test (extrema.cc:181-207)
//     runs one queue entry per thread, instead of every wave paying for its rarest lane.
// Replicate borders (gaussian.hh:43-57,70-84) come from clamping the grey loads, as in K3.
constexpr int RW_OWN = OP_RW_OWN;     // columns owned by a band
constexpr int RW_H = RW_OWN + 4;      // row-pass columns: x0 - 2 .. x0 + 241 (one ring column each side is used)
constexpr int RW_QCAP = 1024;         // scan queue entries per row pair (overflow is handled in place)
#ifndef OP_RW_RAWCAP                   // debug knobs of tests/test_gpu_sift.py (a variant build): a tiny LDS list / the unpackable path
#define OP_RW_RAWCAP 192
#endif
#ifndef OP_RW_PACKABLE_BELOW
#define OP_RW_PACKABLE_BELOW 8192
#endif
constexpr int RW_RAWCAP = OP_RW_RAWCAP;   // raw extrema a workgroup collects in LDS (x | y << 13 | layer << 26); more go straight to the image's list

// wave64 inclusive add-scan on DPP (row_shr within the four rows of 16 lanes, then row_bcast:15 / :31 across rows);
// call in wave-uniform control flow
__device__ __forceinline__ int wave_scan_add_i(int v) {
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
	return v;
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// mask = 2 * mask + !(c < thr): one comparison into vcc, one add-with-carry
__device__ __forceinline__ unsigned gate_bit(unsigned m, float c, float thr) {
	asm("v_cmp_nlt_f32_e64 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(c), "s"(thr) : "vcc");
	return m;
}
// (a.x, b.x) / (a.y, b.y): one instruction each
__device__ __forceinline__ f32x2 pk_lo(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_hi(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// (plain stores: non-temporal ones were measured 25 % slower here -- the rows start at arbitrary 4-byte
// offsets, and L2 no longer merges the partial lines at wave and band seams before they reach HBM)

// 26-neighbour test on the DoG ring (extrema.cc:181-207): slot = ring row of the centre
__device__ __forceinline__ float max3f(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float min3f(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// The reference clears `max` on any neighbour v >= center - judge and `min` on any v <= center + judge: with the largest
// and the smallest of the 26 neighbours (v_max3 / v_min3 skip NaNs like the element-wise comparisons do) that is two
// comparisons, written negated so that a NaN on either side leaves the flag set as the reference's loop does.
__device__ __forceinline__ bool ring_extremum(const float (*sD)[6][RW_H], int slot, int L, int hc, float judge) {
	const float center = sD[slot][L][hc];
	const float cmp1 = center - judge, cmp2 = center + judge;
	float v[26]; int n = 0;
#pragma unroll
	for (int di = -1; di <= 1; ++di) {
		const int sl = (slot + di) & 3;
#pragma unroll
		for (int dl = -1; dl <= 1; ++dl)
#pragma unroll
			for (int dj = -1; dj <= 1; ++dj) {
				if (di == 0 && dl == 0 && dj == 0) continue;
				v[n++] = sD[sl][L + dl][hc + dj];
			}
	}
	float mx = max3f(v[0], v[1], v[2]), mn = min3f(v[0], v[1], v[2]);
#pragma unroll
	for (int i = 3; i + 1 < 26; i += 2) { mx = max3f(mx, v[i], v[i + 1]); mn = min3f(mn, v[i], v[i + 1]); }
	mx = max3f(mx, v[25], v[25]); mn = min3f(mn, v[25], v[25]);
	return !(mx >= cmp1) || !(mn <= cmp2);
}

__global__ void __launch_bounds__(256) k_pyramid_rows(SiftPlan p, int* __restrict__ raw, int* __restrict__ raw_count, int cap) {
	__shared__ f32x2 sV[3][2][256];          // column-pass results [sigma pair][row of the pair][column]
	__shared__ float sGrey[2][256];
#define OP_RING_ROWS 4
	__shared__ float sD[OP_RING_ROWS][6][RW_H];         // |DoG| ring [row & 3][layer][row-pass column]
	__shared__ unsigned short sQ[RW_QCAP];
	__shared__ int sQn[2];
	// Raw extrema found by this workgroup: collected here and appended to the image's list ONCE, at the end.  A global
	// atomic per found extremum returns a value, and waiting for it (vmcnt(0)) waits for every store the wavefront has in
	// flight as well -- two of five wavefront-steps found an extremum and paid a write round trip for it.
	__shared__ unsigned sRaw[RW_RAWCAP];
	__shared__ int sRawN[2];                                   // [0] entries, [1] this workgroup's first slot in the image's list
	const int tid = threadIdx.x;
	// work item; consecutive items (neighbouring bands share cache lines at their seams, neighbouring
	// segments their halo rows) are handed to one XCD, i.e. one L2
	const unsigned lin = blockIdx.x, per = gridDim.x >> 3;
	const unsigned swz = lin < per * 8 ? (lin & 7) * per + (lin >> 3) : lin;
	const int img = (int)(swz / (unsigned)p.rw_items);
	int item = (int)(swz % (unsigned)p.rw_items);
	int o = 0;
	for (;;) {
		const int cnt = p.oct[o].rw_nb * p.oct[o].rw_nseg;
		if (item < cnt || o + 1 == p.noct) break;
		item -= cnt; ++o;
	}
	const OctDesc od = p.oct[o];
	const int x0 = (item % od.rw_nb) * RW_OWN, y0 = (item / od.rw_nb) * p.rw_seg;
	const int rows_own = od.h - y0 < p.rw_seg ? od.h - y0 : p.rw_seg;
	const int nsteps = (rows_own + 3) >> 1;                       // row pairs (y0-1, y0), ... covering y0-1 .. y0+rows_own
	float* ws = p.ws + (long long)img * p.ws_stride;
	const float* grey = ws + plane_off_grey(od);

	f32x2 KP0[4], KP1[7], KP2[7];                                // uniform: SGPR pairs
#pragma unroll
	for (int d = 0; d < 7; ++d) {
		if (d < 4) KP0[d] = f32x2{p.kpair[0][d][0], p.kpair[0][d][1]};
		KP1[d] = f32x2{p.kpair[1][d][0], p.kpair[1][d][1]};
		KP2[d] = f32x2{p.kpair[2][d][0], p.kpair[2][d][1]};
	}

	// column-pass role: column x0 - 8 + tid, clamped
	int xc = x0 - 8 + tid; xc = xc < 0 ? 0 : (xc > od.w - 1 ? od.w - 1 : xc);
	// buffer addressing: the plane's descriptor and the row offset are uniform (SGPRs), the column a 32-bit per-thread
	// byte offset -- no 64-bit VALU address arithmetic, and no address registers held across the loop
	const unsigned plane_bytes = (unsigned)od.plane * 4u;
	const __amdgpu_buffer_rsrc_t r_grey = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(grey), 0, plane_bytes, 0x00020000);
	const unsigned xc4 = (unsigned)xc * 4u;
	auto grow = [&](int y) -> float {
		y = y < 0 ? 0 : (y > od.h - 1 ? od.h - 1 : y);
		return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_grey, xc4, (unsigned)y * (unsigned)od.w * 4u, 0));
	};
	// The two grey rows a step fetches are loaded by hand (buffer_load_dword as written, waited for as written).  vmcnt counts
	// loads AND stores on this target, in issue order, and the number of store instructions a wavefront issues per step is
	// not a compile-time constant (0, 6 or 12: rows outside the segment store nothing, a band's odd last column takes the
	// 4-byte form as well), so for a compiler-visible load the wait-count pass must assume NO store lies between the load
	// and its use: it put `s_waitcnt vmcnt(2)` into the column pass, i.e. waited for every Gaussian-plane store of the
	// previous step to be acknowledged -- a write round trip per step on the dominant kernel's critical path (the column
	// pass, 132 packed instructions, took 3900 cycles in the phase trace against 1800 for the row pass with as many).
	// Here the wavefront knows how many stores it has issued since the loads (a wave-uniform count) and waits for exactly
	// the loads: `s_waitcnt vmcnt(<stores of this step>)`, at the end of the step.
	typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
	u32x4v rg;
	{
		const unsigned long long ga = (unsigned long long)grey;
		rg.x = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ga);
		rg.y = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32)) & 0xFFFFu;     // stride 0
		rg.z = (unsigned)__builtin_amdgcn_readfirstlane((int)plane_bytes);
		rg.w = 0x00020000u;
	}
	auto grow_issue = [&](int y, float& dst) {
		y = y < 0 ? 0 : (y > od.h - 1 ? od.h - 1 : y);
		const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)y * (unsigned)od.w * 4u));
		asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(xc4), "s"(rg), "s"(so) : "memory");
	};
	const __amdgpu_buffer_rsrc_t r_gau = __builtin_amdgcn_make_buffer_rsrc(ws + plane_off_gauss(od, 7, 1), 0, 6u * plane_bytes, 0x00020000);
	f32x2 win[7];                                                 // grey rows r-6 .. r+7 of the current pair (r, r+1)
#pragma unroll
	for (int i = 0; i < 7; ++i) win[i] = f32x2{grow(y0 - 7 + 2 * i), grow(y0 - 6 + 2 * i)};
	// the window has LANDED before the loop: a load still pending at the loop header would put a wait for it -- and for the
	// stores issued since -- in front of its use in EVERY iteration
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
	for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(win[i]));

	// row-pass role: row rr of the pair, columns h, h+1 (x = x0 - 2 + h)
	const int rr = tid >> 7, j = tid & 127, h = 2 * j;
	const bool hact = j < RW_H / 2;
	const int x = x0 - 2 + h;
	const bool own = j >= 1 && j <= RW_OWN / 2;                   // both columns owned by this band
	const bool st0 = own && x < od.w, st1 = own && x + 1 < od.w;
	const bool sc0 = st0 && x >= 1 && x <= od.w - 2, sc1 = st1 && x + 1 <= od.w - 2;   // extrema.cc:212
	const unsigned allow = (sc0 ? 0x0Fu : 0u) | (sc1 ? 0xF0u : 0u);
	unsigned pm = 0;                                              // rr == 1: the gate mask of the row produced in the previous pair
	if (tid < 2) { sQn[tid] = 0; sRawN[tid] = 0; }
	const bool packable = od.w < OP_RW_PACKABLE_BELOW && od.h < OP_RW_PACKABLE_BELOW;
	auto emit_raw = [&](int ex, int ey, int eL) {
		int e = RW_RAWCAP;
		if (packable) e = atomicAdd(&sRawN[0], 1);
		if (e < RW_RAWCAP) sRaw[e] = (unsigned)ex | ((unsigned)ey << 13) | ((unsigned)eL << 26);
		else {
			const int sl = atomicAdd(&raw_count[img], 1);
			if (sl < cap) { int* q = raw + ((long long)img * cap + sl) * 4; q[0] = ex; q[1] = ey; q[2] = o; q[3] = eL; }
		}
	};
	// store instructions this wavefront issues in a step whose row lies inside the segment: six 8-byte stores if any lane
	// owns a column pair, six 4-byte stores if any lane owns a band's odd last column (each block is skipped when no lane takes it)
	const int kstores = (__ballot(st1) != 0ULL ? 6 : 0) + (__ballot(st0 && !st1) != 0ULL ? 6 : 0);

	for (int t = 0; t < nsteps; ++t) {
		const int r = y0 - 1 + 2 * t;
		float nx0, nx1;                                           // the next pair's two new rows, in flight during this one
		grow_issue(r + 8, nx0); grow_issue(r + 9, nx1);
		{	// ---- column pass: rows r (window elements 0..12) and r+1 (1..13)
			// (the chains start from their first product: the reference's `tmp = 0; tmp += ...` (gaussian.hh:60-64) differs
			// from it only in the sign of a zero sum, which no later stage can observe -- every consumer subtracts or compares)
			f32x2 a0[3], a1[3];
#define OP_WMUL(i, kp) ((((i) & 1) ? __builtin_shufflevector(win[(i) >> 1], win[(i) >> 1], 1, 1) : __builtin_shufflevector(win[(i) >> 1], win[(i) >> 1], 0, 0)) * (kp))
#pragma unroll
			for (int k = 0; k < 13; ++k) {
				const int d = k < 6 ? 6 - k : k - 6;
				if (k == 0) { a0[1] = OP_WMUL(k, KP1[d]); a1[1] = OP_WMUL(k + 1, KP1[d]); a0[2] = OP_WMUL(k, KP2[d]); a1[2] = OP_WMUL(k + 1, KP2[d]); }
				else {
					a0[1] = a0[1] + OP_WMUL(k, KP1[d]); a1[1] = a1[1] + OP_WMUL(k + 1, KP1[d]);
					a0[2] = a0[2] + OP_WMUL(k, KP2[d]); a1[2] = a1[2] + OP_WMUL(k + 1, KP2[d]);
				}
				if (k == 3) { a0[0] = OP_WMUL(k, KP0[3]); a1[0] = OP_WMUL(k + 1, KP0[3]); }
				else if (k > 3 && k <= 9) {
					const int d0 = k < 6 ? 6 - k : k - 6;       // distance from the centre: 3..0..3
					a0[0] = a0[0] + OP_WMUL(k, KP0[d0]); a1[0] = a1[0] + OP_WMUL(k + 1, KP0[d0]);
				}
			}
#undef OP_WMUL
#pragma unroll
			for (int pl = 0; pl < 3; ++pl) { sV[pl][0][tid] = a0[pl]; sV[pl][1][tid] = a1[pl]; }
			sGrey[0][tid] = win[3].x; sGrey[1][tid] = win[3].y;
		}
		lds_barrier();

		// ---- row pass + DoG for (row r + rr; columns h, h+1)
		float dcur[2][6];
		const int y = r + rr;
		if (hact) {
			f32x2 gA[3], gB[3];
			{
				f32x2 w[10];        // columns h+2 .. h+11: 16-byte aligned, so the window comes as five ds_read_b128
#pragma unroll
				for (int i = 0; i < 10; ++i) w[i] = sV[0][rr][h + 2 + i];
				f32x2 a = w[1] * KP0[3], b = w[2] * KP0[3];
#pragma unroll
				for (int k = 1; k < 7; ++k) {
					const int d = k < 3 ? 3 - k : k - 3;
					a = a + w[k + 1] * KP0[d]; b = b + w[k + 2] * KP0[d];
				}
				gA[0] = a; gB[0] = b;
			}
#pragma unroll
			for (int pl = 1; pl < 3; ++pl) {
				f32x2 w[14];
#pragma unroll
				for (int i = 0; i < 14; ++i) w[i] = sV[pl][rr][h + i];
				f32x2 a = w[0] * (pl == 1 ? KP1[6] : KP2[6]), b = w[1] * (pl == 1 ? KP1[6] : KP2[6]);
#pragma unroll
				for (int k = 1; k < 13; ++k) {
					const int d = k < 6 ? 6 - k : k - 6;
					const f32x2 kp = pl == 1 ? KP1[d] : KP2[d];
					a = a + w[k] * kp; b = b + w[k + 1] * kp;
				}
				gA[pl] = a; gB[pl] = b;
			}
			// Gaussian stack of the two pixels as (pixel A, pixel B) pairs per scale: P[0] = grey (dog.cc:53), P[1..6].
			// The accumulators hold (sigma a, sigma b) per pixel; one v_pk_mov_b32 per scale transposes them, and the
			// pairs are then the operands of the packed |DoG| subtraction, of the ring writes and of the stores as they are.
			f32x2 P[7];
			P[0] = *(const f32x2*)&sGrey[rr][h + 6];
#pragma unroll
			for (int pl = 0; pl < 3; ++pl) { P[2 * pl + 1] = pk_lo(gA[pl], gB[pl]); P[2 * pl + 2] = pk_hi(gA[pl], gB[pl]); }
			f32x2 D[6];
#pragma unroll
			for (int l = 0; l < 6; ++l) {                         // dog.cc:126
				const f32x2 d = P[l] - P[l + 1];
				D[l] = f32x2{fabsf(d.x), fabsf(d.y)};
				dcur[0][l] = D[l].x; dcur[1][l] = D[l].y;
			}
			const int slot = (2 * t + rr) & (OP_RING_ROWS - 1);
#pragma unroll
			for (int l = 0; l < 6; ++l) *(f32x2*)&sD[slot][l][h] = D[l];
			if (st0 && y >= y0 && y < y0 + rows_own) {
				const unsigned bo = ((unsigned)y * (unsigned)od.w + (unsigned)x) * 4u;
				if (st1) {
#pragma unroll
					for (int s = 1; s <= 6; ++s) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, P[s]), r_gau, bo, (unsigned)(s - 1) * plane_bytes, 0);
				} else {
#pragma unroll
					for (int s = 1; s <= 6; ++s) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, P[s].x), r_gau, bo, (unsigned)(s - 1) * plane_bytes, 0);
				}
			}
		} else {
#pragma unroll
			for (int e = 0; e < 2; ++e)
#pragma unroll
				for (int l = 0; l < 6; ++l) dcur[e][l] = 0.f;
		}

		// ---- gate: rr == 0 scans its current row r (row r+1 is written by the other half in this
		// pair), rr == 1 scans the row it produced in the previous pair (r-1)
		// Every thread gates the 8 candidates of ITS row (2 columns x DoG layers 1..4, extrema.cc:179) into a bit mask:
		// v_cmp + v_addc (mask = 2 mask + pass) per candidate.  The rr == 0 waves queue their mask at once (row r: its
		// lower neighbour r+1 is being written by the other half in this very pair), the rr == 1 waves queue the mask they
		// made in the previous pair (row r-1) and keep the new one -- a row's validity is a wave-uniform test.
		unsigned mine = 0;                                        // candidates that did not fit the queue
		const int rru = __builtin_amdgcn_readfirstlane(rr);
		const int ysc = rru == 0 ? r : r - 1;
		{
			unsigned cm = 0;
#pragma unroll
			for (int b = 7; b >= 0; --b) cm = gate_bit(cm, dcur[b >> 2][(b & 3) + 1], p.pre_color_thres);
			cm &= allow;
			unsigned mask;
			if (rru == 0) mask = cm; else { mask = pm; pm = cm; }
			if (!(ysc >= y0 && ysc < y0 + rows_own && ysc >= 1 && ysc <= od.h - 2)) mask = 0;     // extrema.cc:212
			// queue slots: one wave prefix sum (DPP) and ONE LDS atomic per wave -- a per-lane atomicAdd with lane-varying
			// operands is serialised by the compiler into a readlane loop over the active lanes
			const int cnt = __popc(mask);
			const int incl = wave_scan_add_i(cnt);
			const int wtot = __builtin_amdgcn_readlane(incl, 63);
			int wbase = 0;
			if (wtot) {
				if ((tid & 63) == 63) wbase = atomicAdd(&sQn[t & 1], wtot);
				wbase = __builtin_amdgcn_readlane(wbase, 63);
			}
			if (mask) {
				int base = wbase + incl - cnt;
				for (unsigned m = mask; m; m &= m - 1) {
					const int b = __ffs(m) - 1;
					if (base < RW_QCAP) sQ[base] = (unsigned short)((h + (b >> 2)) | (((b & 3) + 1) << 8) | (rr << 11));
					else mine |= 1u << b;
					++base;
				}
			}
		}
		lds_barrier();

		// ---- scan the queue: one entry per thread
		if (tid == 0) sQn[(t + 1) & 1] = 0;
		{
			int n = sQn[t & 1]; n = n > RW_QCAP ? RW_QCAP : n;
			for (int i = tid; i < n; i += 256) {
				const unsigned code = sQ[i];
				const int hc = code & 255, L = (code >> 8) & 7, qr = code >> 11;
				const int slot = (2 * t + (qr ? -1 : 0)) & 3;
				if (ring_extremum(sD, slot, L, hc, p.judge_thres)) emit_raw(x0 - 2 + hc, r + (qr ? -1 : 0), L);
			}
			for (unsigned m = mine; m; m &= m - 1) {
				const int b = __ffs(m) - 1, hc = h + (b >> 2), L = (b & 3) + 1;
				const int slot = (2 * t + (rr ? -1 : 0)) & 3;
				if (ring_extremum(sD, slot, L, hc, p.judge_thres)) emit_raw(x0 - 2 + hc, ysc, L);
			}
		}
		// the two rows fetched at the top of the step have landed: everything this wavefront issued since are its stores
		{
			const int yrow = r + __builtin_amdgcn_readfirstlane(rr);
			const int k = (yrow >= y0 && yrow < y0 + rows_own) ? kstores : 0;
			// (the loaded registers are INPUTS of the wait and the value it produces -- a zero -- is OR-ed into them afterwards:
			// every use of the rows then depends on the wait, and the compiler has no reason to copy the registers before it,
			// which a read-write operand invited: it copied them, still in flight, into the operand's own register)
			unsigned z;
			if (k == 0) asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, 0" : "=v"(z) : "v"(nx0), "v"(nx1) : "memory");
			else if (k == 6) asm volatile("s_waitcnt vmcnt(6)\n\tv_mov_b32 %0, 0" : "=v"(z) : "v"(nx0), "v"(nx1) : "memory");
			else asm volatile("s_waitcnt vmcnt(12)\n\tv_mov_b32 %0, 0" : "=v"(z) : "v"(nx0), "v"(nx1) : "memory");
			nx0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, nx0) | z);
			nx1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, nx1) | z);
		}
		const f32x2 nxt = f32x2{nx0, nx1};
		// slide the window by one row pair
#pragma unroll
		for (int i = 0; i < 6; ++i) win[i] = win[i + 1];
		win[6] = nxt;
	}
	// append this workgroup's raw extrema to the image's list: one atomic for all of them
	lds_barrier();
	int nraw = sRawN[0]; nraw = nraw < RW_RAWCAP ? nraw : RW_RAWCAP;
	if (nraw > 0) {
		if (tid == 0) sRawN[1] = atomicAdd(&raw_count[img], nraw);
		__syncthreads();
		const int base = sRawN[1];
		for (int i = tid; i < nraw; i += 256) {
			const int sl = base + i;
			if (sl < cap) {
				const unsigned e = sRaw[i];
				*(int4*)(raw + ((long long)img * cap + sl) * 4) = make_int4((int)(e & 8191u), (int)((e >> 13) & 8191u), o, (int)(e >> 26));
			}
		}
	}
}

// debug / staged dump: GaussianPyramid::cal_mag_ort (feature/dog.cc:60-94) of one Gaussian plane
__global__ void __launch_bounds__(256) k_magort_plane(SiftPlan p, int img, int o, int s, float* mag, float* ort) {
	const OctDesc od = p.oct[o];
	const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
	if (idx >= od.plane) return;
	const int y = (int)(idx / od.w), x = (int)(idx % od.w);
	const float* g = p.ws + (long long)img * p.ws_stride + plane_off_gauss(od, p.nscale, s);
	float m = 0.f, a = (float)3.14159265358979323846;
	if (x >= 1 && x <= od.w - 2 && y >= 1 && y <= od.h - 2) {
		const float dy = g[idx + od.w] - g[idx - od.w];
		const float dx = g[idx + 1] - g[idx - 1];
		m = opdev::hypotf_glibc(dx, dy);
		a = opdev::fast_atan_plus_pi(dy, dx);
	}
	mag[idx] = m; ort[idx] = a;
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
	return sizeof(float) * ((size_t)NR * NC + (size_t)vt_rows(halo) * PV + 4 * (size_t)GR * PG);
}

hipError_t launch_grey_octaves(const SiftPlan& p, bool write_work, hipStream_t st) {
	const unsigned ntile = (unsigned)((p.ww + WT - 1) / WT) * (unsigned)((p.wh + WR - 1) / WR) * (unsigned)p.n;
	dim3 grid(((ntile + 7u) >> 3) << 3);              // eight per-XCD queues of equal length (the kernel drops the padding)
	if (p.src_u8) hipLaunchKernelGGL(k_grey_octaves<unsigned char>, grid, dim3(256), 0, st, p, write_work ? 1 : 0);
	else hipLaunchKernelGGL(k_grey_octaves<float>, grid, dim3(256), 0, st, p, write_work ? 1 : 0);
	return hipGetLastError();
}

hipError_t launch_magort_plane(const SiftPlan& p, int img, int oct, int s, float* mag, float* ort, hipStream_t st) {
	const long long n = p.oct[oct].plane;
	hipLaunchKernelGGL(k_magort_plane, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, img, oct, s, mag, ort);
	return hipGetLastError();
}

hipError_t launch_pyramid(const SiftPlan& p, int* raw, int* raw_count, int cap, hipStream_t st) {
	size_t lds = pyramid_lds_bytes(p.halo);
	{	// per-function attribute; idempotent, so concurrent first calls from several host threads are harmless
		hipError_t e = hipFuncSetAttribute((const void*)k_pyramid<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
		if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_pyramid<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
		if (e != hipSuccess) return e;
	}
	if (p.rows_ok) {           // the shipped Gaussian bank: row-streaming kernel; any other bank: tiles
		const unsigned blocks = (unsigned)p.n * (unsigned)p.rw_items;
		hipLaunchKernelGGL(k_pyramid_rows, dim3(blocks), dim3(256), 0, st, p, raw, raw_count, cap);
		return hipGetLastError();
	}
	dim3 grid(p.total_tiles, p.n);
	if (p.halo == 6) hipLaunchKernelGGL(k_pyramid<6>, grid, dim3(256), lds, st, p, raw, raw_count, cap);
	else hipLaunchKernelGGL(k_pyramid<0>, grid, dim3(256), lds, st, p, raw, raw_count, cap);
	return hipGetLastError();
}

hipError_t launch_debug_math(int which, const float* x, const float* y, int n, float* out, hipStream_t st) {
	hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, st, which, x, y, n, out);
	return hipGetLastError();
}
