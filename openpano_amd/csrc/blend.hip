// blend.hip -- final warp + blend on the device.
//
// Replaces, for the whole bundle per launch:
//   ConnectedImages::blend            stitch/stitcher_image.cc:116-155 (inverse map per canvas pixel,
//                                     stitch/projection.hh:14-71, Homography::trans homography.hh:53-58)
//   LinearBlender::run                stitch/blender.cc:24-96 (both LAZY_READ branches)
//   MultiBandBlender::run             stitch/multiband.cc:19-151 (+ GaussianBlur::blur<WeightedPixel>,
//                                     feature/gaussian.hh:30-91)
//   interpolate                       lib/imgproc.cc:135-156
//   CylinderProject::project          stitch/warp.cc:25-44
// and keeps on the host, in fp64 with the host libm exactly like the reference, the O(n)
// geometry: calc_inverse_homo / update_proj_range / get_final_resolution
// (stitcher_image.cc:36-114) and the bounds of CylinderProject::project(Shape2D&) (warp.cc:46-67).
//
// The blender API of the reference takes the coordinate map as an opaque std::function
// (blender.hh:52-56), which a device cannot call; the seam is therefore one level up, at
// ConnectedImages::blend, and the map is passed as PODs (projection method, proj_range.min,
// resolution, homo_inv per image).
//
// Numerics: coordinates in fp64 exactly in the reference's operation order (this TU is built
// with -ffp-contract=off).  The map's transcendentals are SEPARABLE: proj2homo takes sin / cos of a value that
// depends on the canvas column only and tan of one that depends on the row only (stitcher_image.cc:144-145:
// c = Vec2D(t.x, t.y) * resolution + proj_range.min), and CylinderProject's tan / cos (warp.cc:19-23) take a
// per-column argument.  They are therefore tabulated once per canvas by the HOST libm -- W + H evaluations of the
// very functions the reference calls, instead of W x H evaluations of a device libm that can differ from glibc in
// the last ulp (rounds 1-4: masks equal up to 2e-5 of the pixels, colours within 1e-4) -- and the kernels read the
// tables: every coordinate is the reference's double, bit for bit.  Colour arithmetic is fp32 in the reference's
// order, so the canvas is bit-equal for every projection (tests assert array equality).
#include "internal.hpp"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

struct op_canvas {
	float* data = nullptr;   // device, h x w x 3
	int h = 0, w = 0;
	int device = 0;
};

namespace {

struct BlendImg {
	const float* data; int h, w;   // ImageRef::height()/width(): bounds, weights, centre
	int mh, mw;                    // rows / cols of the pixel buffer (differ after a cylinder pre-warp, see op_blend_image)
	int x0, y0, x1, y1;        // ROI on the canvas, inclusive (BlenderBase::Range, blender.hh:19-27)
	double hinv[9];
	long long roi_off;         // multiband: offset (pixels) of this image's ROI planes
	int rw, rh;
};

struct BlendGeom { int method; double minx, miny, resx, resy; };

// stitch/projection.hh:29-31,38-40,66-68 for canvas pixel (i, j).  Flat: the coordinates themselves.  Cylindrical /
// spherical: colsc[j] = (sin, cos) of the column's x, rowt[i] = y (cylindrical) or tan(y) (spherical), from the host
// libm (trig_tables below); the tables have one entry past the canvas (the multiband first level visits it).
struct BlendTrig { const double2* colsc; const double* rowt; int w1, h1; };
__device__ __forceinline__ void proj2homo(const BlendGeom& g, const BlendTrig& t, int i, int j, double& hx, double& hy, double& hz) {
	if (g.method == 0) { hx = (double)j * g.resx + g.minx; hy = (double)i * g.resy + g.miny; hz = 1.0; }
	else {
		const double2 sc = t.colsc[j < t.w1 ? j : t.w1 - 1];
		hx = sc.x; hz = sc.y; hy = t.rowt[i < t.h1 ? i : t.h1 - 1];
	}
}

// the lambda of ConnectedImages::blend (stitcher_image.cc:143-151) after proj2homo
__device__ __forceinline__ void space_to_image(const BlendImg& im, double hx, double hy, double hz, double& ox, double& oy) {
	const double* d = im.hinv;
	const double rx = d[0] * hx + d[1] * hy + d[2] * hz;
	const double ry = d[3] * hx + d[4] * hy + d[5] * hz;
	const double rz = d[6] * hx + d[7] * hy + d[8] * hz;
	if (rz < 0) { ox = -10; oy = -10; return; }
	const double denom = 1.0 / rz;
	ox = rx * denom + im.w * 0.5;
	oy = ry * denom + im.h * 0.5;
}

// interpolate (lib/imgproc.cc:135-156); false = Color::NO
__device__ __forceinline__ bool interpolate(const float* __restrict__ img, int rows, int cols, float r, float c, float (&out)[3]) {
	const int fr = (int)floorf(r), fc = (int)floorf(c);
	if (fr < 0 || fc < 0 || fc + 1 >= cols || fr + 1 >= rows) return false;
	r -= (float)fr; c -= (float)fc;
	const float* p00 = img + ((long long)fr * cols + fc) * 3;
	const float* p10 = p00 + (long long)cols * 3;
	float a0 = p00[0], a1 = p00[1], a2 = p00[2];
	if (a0 < 0) return false;
	float b0 = p10[0], b1 = p10[1], b2 = p10[2];
	if (b0 < 0) return false;
	float c0 = p10[3], c1 = p10[4], c2 = p10[5];
	if (c0 < 0) return false;
	float d0 = p00[3], d1 = p00[4], d2 = p00[5];
	if (d0 < 0) return false;
	float w = (1 - r) * (1 - c);
	float r0 = 0.f + a0 * w, r1 = 0.f + a1 * w, r2 = 0.f + a2 * w;
	w = r * (1 - c);
	r0 += b0 * w; r1 += b1 * w; r2 += b2 * w;
	w = r * c;
	r0 += c0 * w; r1 += c1 * w; r2 += c2 * w;
	w = (1 - r) * c;
	r0 += d0 * w; r1 += d1 * w; r2 += d2 * w;
	out[0] = r0; out[1] = r1; out[2] = r2;
	return true;
}

// The canvas-pixel kernels below walk "every image whose ROI holds this pixel, in index order".  Testing all n ROIs per
// pixel made them scalar-bound (38 images: 38 descriptor loads and 152 compares per pixel for the two or three that
// cover it): a workgroup -- a 64 x 4 pixel tile -- first marks the images whose ROI meets its TILE (one image per lane,
// one ballot per 64 images), then every pixel walks only those, with the exact per-pixel test of the reference.
constexpr int COVER_WORDS = 64;          // images per round of the walk: 64 x 64
__device__ __forceinline__ void tile_cover(const BlendImg* __restrict__ imgs, int k0, int n, int i0, int j0, int excl, unsigned long long* s_cover) {
	const int k1 = n - k0 < COVER_WORDS * 64 ? n : k0 + COVER_WORDS * 64;
	__syncthreads();                                    // the previous round's list is no longer read
	for (int k = k0 + (int)threadIdx.x; k < ((k1 - k0 + 63) & ~63) + k0; k += 256) {
		bool hit = false;
		if (k < k1) {
			const BlendImg& im = imgs[k];
			hit = im.x0 <= j0 + 63 && im.x1 - excl >= j0 && im.y0 <= i0 + 3 && im.y1 - excl >= i0;
		}
		const unsigned long long b = __ballot(hit);
		if ((threadIdx.x & 63) == 0) s_cover[(k - k0) >> 6] = b;
	}
	__syncthreads();
}
// visit(k) for every marked image of the round starting at k0, ascending; the list is wave-uniform (scalar loop control)
template <typename F>
__device__ __forceinline__ void walk_cover(const unsigned long long* s_cover, int k0, int n, F&& visit) {
	const int words = ((n - k0 < COVER_WORDS * 64 ? n - k0 : COVER_WORDS * 64) + 63) >> 6;
	for (int wd = 0; wd < words; ++wd) {
		const unsigned long long mw = s_cover[wd];
		unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)mw), hi = __builtin_amdgcn_readfirstlane((unsigned)(mw >> 32));
		while (lo) { const int b = __builtin_ctz(lo); lo &= lo - 1; visit(k0 + wd * 64 + b); }
		while (hi) { const int b = __builtin_ctz(hi); hi &= hi - 1; visit(k0 + wd * 64 + 32 + b); }
	}
}

// ---- LinearBlender::run (blender.cc:24-96): thread per canvas pixel, images in index order ----
__global__ void __launch_bounds__(256) k_blend_linear(BlendGeom g, BlendTrig trig, const BlendImg* __restrict__ imgs, int n,
		float* __restrict__ out, int H, int W, int ordered_input, int lazy) {
	__shared__ unsigned long long s_cover[COVER_WORDS];
	const int j = blockIdx.x * 64 + (threadIdx.x & 63);
	const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
	const bool live = i < H && j < W;
	double hx, hy, hz;
	proj2homo(g, trig, i, j, hx, hy, hz);
	float s0 = 0.f, s1 = 0.f, s2 = 0.f, wsum = 0.f;
	for (int k0 = 0; k0 < n; k0 += COVER_WORDS * 64) {
		tile_cover(imgs, k0, n, blockIdx.y * 4, blockIdx.x * 64, lazy ? 1 : 0, s_cover);
		if (!live) continue;
		walk_cover(s_cover, k0, n, [&](int k) {
			const BlendImg& im = imgs[k];
			// non-lazy: Range::contain, inclusive (blender.cc:84); lazy: loops exclude max (blender.cc:49-51)
			const bool in = lazy ? (i >= im.y0 && i < im.y1 && j >= im.x0 && j < im.x1)
			                     : (i >= im.y0 && i <= im.y1 && j >= im.x0 && j <= im.x1);
			if (!in) return;
			double ox, oy;
			space_to_image(im, hx, hy, hz, ox, oy);
			if (ox < 0 || ox >= im.w || oy < 0 || oy >= im.h) return;      // ImageToAdd::map_coor (blender.hh:39-44)
			const float r = (float)oy, c = (float)ox;
			float col[3];
			if (!interpolate(im.data, im.mh, im.mw, r, c, col)) return;
			if (col[0] < 0) return;
			float w = (float)(0.5 - fabs((double)(c / (float)im.w) - 0.5));
			if (!ordered_input) w = (float)((double)w * (0.5 - fabs((double)(r / (float)im.h) - 0.5)));
			s0 += col[0] * w; s1 += col[1] * w; s2 += col[2] * w;
			wsum += w;
		});
	}
	if (!live) return;
	float* row = out + ((long long)i * W + j) * 3;
	if (lazy) {
		if (wsum != 0.f) { row[0] = s0 / wsum; row[1] = s1 / wsum; row[2] = s2 / wsum; }   // blender.cc:68-70
		else { row[0] = -1.f; row[1] = -1.f; row[2] = -1.f; }
	} else {
		if (wsum > 0) {            // Vector::operator/(T p) = *this * (1.0 / p), lib/geometry.hh:123-124
			const float inv = (float)(1.0 / (double)wsum);
			row[0] = s0 * inv; row[1] = s1 * inv; row[2] = s2 * inv;
		} else { row[0] = -1.f; row[1] = -1.f; row[2] = -1.f; }
	}
}

// ---- create_first_level + update_weight_map (multiband.cc:19-56,125-143) in ONE pass, thread per canvas pixel:
// proj2homo once per pixel (it does not depend on the image), then every image whose ROI covers the pixel in index
// order: its level-0 WeightedPixel is written with weight 0 while the winner of the winner-takes-all map -- the first
// image with the strictly largest weight, as the reference's `if (w > max)` walk finds it -- is tracked in registers;
// the winner's weight is then set to 1 with one 4-byte store.  The ROI planes are written once and never read back
// (the two-kernel form re-read every weight and rewrote it: 0.49 GB of the 1.33 GB the two kernels moved), and the
// target canvas / its "seen" mask are initialised here too (fill(target, Color::NO), multiband.cc:60-61).
__global__ void __launch_bounds__(256) k_mb_first_fused(BlendGeom g, BlendTrig trig, const BlendImg* __restrict__ imgs, int n,
		float4* __restrict__ cur, unsigned char* __restrict__ mask, float* __restrict__ out, unsigned char* __restrict__ tmask, int H, int W) {
	const int j = blockIdx.x * 64 + (threadIdx.x & 63);
	const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
	// The target is as large as the largest bottom-right ROI coordinate (blender.cc:21) while ROIs are inclusive
	// (blender.hh:19-27): an ROI's last column / row can lie ONE pixel outside the target.  The reference still builds
	// that pixel of the image's level 0 (it feeds the blurs) but its weight-map walk never visits it, so it keeps its
	// own weight; the grid therefore covers (H + 1) x (W + 1) and only pixels inside the target take part in the map.
	__shared__ unsigned long long s_cover[COVER_WORDS];
	const bool live = !(i > H || j > W);
	const bool inside = i < H && j < W;
	double hx, hy, hz;
	proj2homo(g, trig, i, j, hx, hy, hz);
	float mx = 0.f; long long maxe = -1;
	for (int k0 = 0; k0 < n; k0 += COVER_WORDS * 64) {
		tile_cover(imgs, k0, n, blockIdx.y * 4, blockIdx.x * 64, 0, s_cover);
		if (!live) continue;
		walk_cover(s_cover, k0, n, [&](int k) {
			const BlendImg& im = imgs[k];
			if (!(i >= im.y0 && i <= im.y1 && j >= im.x0 && j <= im.x1)) return;
			const long long e = im.roi_off + (long long)(i - im.y0) * im.rw + (j - im.x0);
			double ox, oy;
			space_to_image(im, hx, hy, hz, ox, oy);
			float col[3];
			bool ok = interpolate(im.data, im.mh, im.mw, (float)oy, (float)ox, col);
			if (ok) { float mn = fminf(col[0], fminf(col[1], col[2])); if (mn < 0) ok = false; }
			float4 px = make_float4(0.f, 0.f, 0.f, 0.f);
			if (ok) {
				const double x = ox / (double)im.w - 0.5, y = oy / (double)im.h - 0.5;
				const double v = (0.5 - fabs(x)) * (0.5 - fabs(y));
				const float w = (float)((v > 0.0 ? v : 0.0) + 1e-6);
				px = make_float4(col[0], col[1], col[2], inside ? 0.f : w);
				if (w > mx) { mx = w; maxe = e; }                     // multiband.cc:133-137
			}
			cur[e] = px;
			mask[e] = ok ? 0 : 1;
		});
	}
	if (!inside) return;
	if (maxe >= 0) ((float*)&cur[maxe])[3] = 1.f;
	const long long pe = (long long)i * W + j;
	out[pe * 3] = -1.f; out[pe * 3 + 1] = -1.f; out[pe * 3 + 2] = -1.f;
	tmask[pe] = 0;
}

// ---- GaussianBlur::blur<WeightedPixel> (feature/gaussian.hh:30-91): column pass then row pass,
// replicate borders, sequential fp32 multiply-add in tap order on all 4 channels ----
struct BlurTaps { int center; float k[2 * OP_MAX_KCENTER + 1]; };

// CT > 0: half-width known at compile time (6 and 9 for the shipped window factor): the taps are
// unrolled and all loads of a pixel are in flight together; CT == 0: any half-width.
template <bool COLS, int CT>
__global__ void __launch_bounds__(256) k_mb_blur(const BlendImg* __restrict__ imgs, BlurTaps taps,
		const float4* __restrict__ src, float4* __restrict__ dst) {
	const BlendImg& im = imgs[blockIdx.y];
	const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
	if (e >= (long long)im.rw * im.rh) return;
	const int i = (int)(e / im.rw), j = (int)(e % im.rw);
	const float4* base = src + im.roi_off;
	float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
	if (CT > 0) {
		float4 v[2 * (CT > 0 ? CT : 1) + 1];
#pragma unroll
		for (int k = -CT; k <= CT; ++k) {
			if (COLS) { int ii = i + k; ii = ii < 0 ? 0 : (ii > im.rh - 1 ? im.rh - 1 : ii); v[k + CT] = base[(long long)ii * im.rw + j]; }
			else { int jj = j + k; jj = jj < 0 ? 0 : (jj > im.rw - 1 ? im.rw - 1 : jj); v[k + CT] = base[(long long)i * im.rw + jj]; }
		}
#pragma unroll
		for (int k = 0; k <= 2 * CT; ++k) {
			const float kv = taps.k[k];
			t.w += v[k].w * kv; t.x += v[k].x * kv; t.y += v[k].y * kv; t.z += v[k].z * kv;
		}
	} else {
		const int C = taps.center;
		for (int k = -C; k <= C; ++k) {
			float4 v;
			if (COLS) { int ii = i + k; ii = ii < 0 ? 0 : (ii > im.rh - 1 ? im.rh - 1 : ii); v = base[(long long)ii * im.rw + j]; }
			else { int jj = j + k; jj = jj < 0 ? 0 : (jj > im.rw - 1 ? im.rw - 1 : jj); v = base[(long long)i * im.rw + jj]; }
			const float kv = taps.k[k + C];
			t.w += v.w * kv; t.x += v.x * kv; t.y += v.y * kv; t.z += v.z * kv;
		}
	}
	dst[im.roi_off + e] = t;
}

// ---- the same blur, both passes in ONE kernel (the shipped window factor: half-widths 6 and 9).
// The two-pass form above moves every WeightedPixel plane through HBM twice per level and re-reads each
// source pixel 2C+1 times from cache (measured: 1.23 GB of traffic per launch for 0.44 GB algorithmic).
// Here a workgroup owns a band of 256 - 2C output columns and walks down a segment of rows:
//   column pass  thread = column (band + C halo columns either side, clamped = replicate border); the
//                2C+1 source rows around the current row live in registers as a rotating window (one
//                coalesced 16-byte load per thread and row), the reference's  tmp += line[i+k]*kernel[k]
//                runs in tap order on all four channels;
//   row pass     the column-pass row goes to LDS (double buffered: one barrier per row), thread = output
//                column reads its 2C+1 neighbours as b128 and applies the same taps in order.
// The intermediate plane never reaches HBM; a source pixel is read once per band (+ the segment halo).
// FIRST (level 0 -> 1): the band of level 0 (multiband.cc:75-110) is written here as well.  After the
// winner-takes-all map exactly one image has weight 1 at a target pixel and every other one 0, so the reference's
// sum over images has a single term, (cur - next) * 1 / 1: the winner's thread holds cur (its column window) and next
// (just computed) and stores the band into the untouched target -- the level-0 pass of k_mb_accumulate (a read of
// every image's ROI plane) disappears.
// Waits (round 5).  vmcnt counts loads AND stores, in issue order, and __syncthreads() carries a workgroup fence that waits
// for every store in flight: the first form of this kernel -- store row r, load row r + CT + 1, __syncthreads() -- waited
// for the acknowledgement of the row it had just written, once per row.  Now (i) the barrier is LDS-only (the rows written
// are never read back here), (ii) the source row of the NEXT iteration is requested at the top of an iteration, i.e. BEFORE
// this iteration's stores, so that its wait (one iteration later) has only those stores behind it, and (iii) every
// wavefront issues the same number of store instructions per row -- lanes that must not write get an offset beyond the
// buffer descriptor's range, which the hardware drops -- so that number is a compile-time constant and the compiler's own
// wait for the load is `vmcnt(<stores>)`: no store acknowledgement is ever waited for.  The window holds one row more
// (2 CT + 2 slots: the row in flight must not land on a row still in use), hence segments of k x (2 CT + 2) rows.
__device__ __forceinline__ void blur_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
template <int CT, bool FIRST>
__global__ void __launch_bounds__(256) k_mb_blur_fused(const BlendImg* __restrict__ imgs, BlurTaps taps,
		const float4* __restrict__ src, float4* __restrict__ dst, float* __restrict__ target, unsigned char* __restrict__ tmask, int H, int W) {
	constexpr int NT = 2 * CT + 1;          // taps
	constexpr int NS = NT + 1;              // window slots
	constexpr int TWO = 256 - 2 * CT;       // output columns of a band
	constexpr int SEG = (CT <= 6 ? 8 : 6) * NS;   // rows of a segment (a whole number of window rotations; its 2 CT halo rows are re-read)
	__shared__ float4 s_mid[2][256];
	const BlendImg& im = imgs[blockIdx.y];
	const int rw = im.rw, rh = im.rh;
	const int nbands = (rw + TWO - 1) / TWO, nsegs = (rh + SEG - 1) / SEG;
	if ((int)blockIdx.x >= nbands * nsegs) return;
	const int band = blockIdx.x % nbands, seg = blockIdx.x / nbands;
	const int t = threadIdx.x;
	const int x = band * TWO - CT + t;                        // this thread's column (may lie in the halo / outside)
	const int xc = x < 0 ? 0 : (x > rw - 1 ? rw - 1 : x);      // replicate border
	const int r0 = seg * SEG, r1 = r0 + SEG < rh ? r0 + SEG : rh;
	const float4* base = src + im.roi_off;
	float4* out = dst + im.roi_off;
	const bool writer = t >= CT && t < 256 - CT && x < rw;     // x >= 0 for these threads
	float kk[NT];
#pragma unroll
	for (int k = 0; k < NT; ++k) kk[k] = taps.k[k];
	auto row_at = [&](int r) { const int rc = r < 0 ? 0 : (r > rh - 1 ? rh - 1 : r); return base[(long long)rc * rw + xc]; };
	float4 v[NS];                           // slot of row q is (q - (r0 - CT)) % NS
#pragma unroll
	for (int k = 0; k < NT; ++k) v[k] = row_at(r0 - CT + k);   // rows r0 - CT .. r0 + CT: the first row's whole window
	const unsigned row_bytes = (unsigned)rw * 16u;
	const unsigned xoff = writer ? (unsigned)x * 16u : 0x80000000u;       // beyond any row: dropped
	for (int rb = r0; rb < r1; rb += NS) {
#pragma unroll
		for (int u = 0; u < NS; ++u) {
			const int r = rb + u;
			// the NEXT row's last source row, requested before this row's stores; its slot held row r - CT - 1, dead since the previous row
			v[(u + NT) % NS] = row_at(r + CT + 1);
			float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
			for (int k = 0; k < NT; ++k) {                   // tap k multiplies row r - CT + k = slot (u + k) % NS
				const float4 s = v[(u + k) % NS]; const float kv = kk[k];
				c.w += s.w * kv; c.x += s.x * kv; c.y += s.y * kv; c.z += s.z * kv;
			}
			s_mid[(r - r0) & 1][t] = c;                       // double buffered: the readers of row r - 1 may still be at work
			blur_lds_barrier();
			float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
			if (writer) {
				const float4* m = &s_mid[(r - r0) & 1][t - CT];
#pragma unroll
				for (int k = 0; k < NT; ++k) {
					const float4 s = m[k]; const float kv = kk[k];
					o.w += s.w * kv; o.x += s.x * kv; o.y += s.y * kv; o.z += s.z * kv;
				}
			}
			// one 16-byte store instruction per wavefront and row, whatever its lanes do (rows past the segment: dropped as well)
			const bool live = r < r1;
			{
				const int rc = live ? r : 0;
				const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (long long)rc * rw, 0, live ? row_bytes : 0u, 0x00020000);
				__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4b, o), rs, xoff, 0, 0);
			}
			if (FIRST) {
				const float4 cc = v[(u + CT) % NS];              // level-0 pixel (r, x): the centre tap's row of this thread's own column
				const int ti = im.y0 + r, tj = im.x0 + x;
				const bool win = writer && live && cc.w > 0 && ti < H && tj < W;     // the winner (weight 1); a masked pixel has weight 0
				float s0 = 0.f, s1 = 0.f, s2 = 0.f, wsum = 0.f;
				s0 += (cc.x - o.x) * cc.w; s1 += (cc.y - o.y) * cc.w; s2 += (cc.z - o.z) * cc.w; wsum += cc.w;
				s0 /= wsum; s1 /= wsum; s2 /= wsum;
				const bool rowok = live && ti < H;
				const int tic = rowok ? ti : 0;
				const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(target + (long long)tic * W * 3, 0, rowok ? (unsigned)W * 12u : 0u, 0x00020000);
				const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(tmask + (long long)tic * W, 0, rowok ? (unsigned)W : 0u, 0x00020000);
				const unsigned to = win ? (unsigned)tj * 12u : 0x80000000u, mo = win ? (unsigned)tj : 0x80000000u;
				__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s0), rt, to, 0, 0);
				__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s1), rt, to, 4, 0);
				__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s2), rt, to, 8, 0);
				__builtin_amdgcn_raw_buffer_store_b8((unsigned char)1, rm, mo, 0, 0);
			}
		}
	}
}

// ---- one band (multiband.cc:75-110): thread per canvas pixel; the last band also clamps (:112-121) ----
__global__ void __launch_bounds__(256) k_mb_accumulate(const BlendImg* __restrict__ imgs, int n,
		const float4* __restrict__ cur, const float4* __restrict__ nxt, const unsigned char* __restrict__ mask,
		float* __restrict__ out, unsigned char* __restrict__ tmask, int H, int W, int is_last) {
	const int j = blockIdx.x * 64 + (threadIdx.x & 63);
	const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (i >= H || j >= W) return;
	float s0 = 0.f, s1 = 0.f, s2 = 0.f, wsum = 0.f;
	for (int k = 0; k < n; ++k) {
		const BlendImg& im = imgs[k];
		if (!(i >= im.y0 && i <= im.y1 && j >= im.x0 && j <= im.x1)) continue;
		const long long e = im.roi_off + (long long)(i - im.y0) * im.rw + (j - im.x0);
		if (mask[e]) continue;
		const float4 cc = cur[e];
		if (cc.w <= 0) continue;
		if (!is_last) {
			const float4 cn = nxt[e];
			s0 += (cc.x - cn.x) * cc.w; s1 += (cc.y - cn.y) * cc.w; s2 += (cc.z - cn.z) * cc.w;
		} else {
			s0 += cc.x * cc.w; s1 += cc.y * cc.w; s2 += cc.z * cc.w;
		}
		wsum += cc.w;
	}
	const long long pe = (long long)i * W + j;
	float* p = out + pe * 3;
	bool seen = tmask[pe] != 0;
	float p0 = p[0], p1 = p[1], p2 = p[2];
	if (!((double)wsum < 1e-6)) {
		s0 /= wsum; s1 /= wsum; s2 /= wsum;
		if (!seen) { p0 = s0; p1 = s1; p2 = s2; seen = true; tmask[pe] = 1; }
		else { p0 += s0; p1 += s1; p2 += s2; }
	}
	if (is_last && seen) {
		p0 = fmaxf(fminf(p0, 1.0f), 0.f); p1 = fmaxf(fminf(p1, 1.0f), 0.f); p2 = fmaxf(fminf(p2, 1.0f), 0.f);
	}
	p[0] = p0; p[1] = p1; p[2] = p2;
}

// ---- all remaining bands in ONE pass over the canvas.  The per-level form above reads, for every canvas pixel and every
// image covering it, the level's plane AND the next one (which the next level's pass reads again as its own), and moves
// the canvas through HBM once per level.  With every level's plane kept (they are a few hundred MB), a thread walks the
// covering images once, reads each level's WeightedPixel once, keeps one (sum, weight) accumulator per level -- each
// level's sum still runs over the images in index order -- and applies the bands to its canvas pixel in level order:
// the same operations on the same operands in the same order as NL launches of k_mb_accumulate.
#define OP_MB_MAX_LEVELS 6
struct BandPlanes { const float4* lv[OP_MB_MAX_LEVELS]; };
template <int NL>          // levels P.lv[0 .. NL-1]; the last one is the final level of the pyramid (no next plane; clamps)
__global__ void __launch_bounds__(256) k_mb_bands(const BlendImg* __restrict__ imgs, int n, BandPlanes P,
		const unsigned char* __restrict__ mask, float* __restrict__ out, unsigned char* __restrict__ tmask, int H, int W) {
	const int j = blockIdx.x * 64 + (threadIdx.x & 63);
	const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
	__shared__ unsigned long long s_cover[COVER_WORDS];
	const bool live = i < H && j < W;
	float s0[NL], s1[NL], s2[NL], ws[NL];
#pragma unroll
	for (int l = 0; l < NL; ++l) { s0[l] = 0.f; s1[l] = 0.f; s2[l] = 0.f; ws[l] = 0.f; }
	for (int k0 = 0; k0 < n; k0 += COVER_WORDS * 64) {
		tile_cover(imgs, k0, n, blockIdx.y * 4, blockIdx.x * 64, 0, s_cover);
		if (!live) continue;
		walk_cover(s_cover, k0, n, [&](int k) {
			const BlendImg& im = imgs[k];
			if (!(i >= im.y0 && i <= im.y1 && j >= im.x0 && j <= im.x1)) return;
			const long long e = im.roi_off + (long long)(i - im.y0) * im.rw + (j - im.x0);
			if (mask[e]) return;
			float4 lvl[NL];
#pragma unroll
			for (int l = 0; l < NL; ++l) lvl[l] = P.lv[l][e];
#pragma unroll
			for (int l = 0; l < NL; ++l) {
				const float4 cc = lvl[l];
				if (cc.w <= 0) continue;
				if (l < NL - 1) {
					const float4 cn = lvl[l + 1];
					s0[l] += (cc.x - cn.x) * cc.w; s1[l] += (cc.y - cn.y) * cc.w; s2[l] += (cc.z - cn.z) * cc.w;
				} else {
					s0[l] += cc.x * cc.w; s1[l] += cc.y * cc.w; s2[l] += cc.z * cc.w;
				}
				ws[l] += cc.w;
			}
		});
	}
	if (!live) return;
	const long long pe = (long long)i * W + j;
	float* p = out + pe * 3;
	const bool seen0 = tmask[pe] != 0;
	bool seen = seen0;
	float p0 = p[0], p1 = p[1], p2 = p[2];
#pragma unroll
	for (int l = 0; l < NL; ++l) {
		if (!((double)ws[l] < 1e-6)) {
			const float a0 = s0[l] / ws[l], a1 = s1[l] / ws[l], a2 = s2[l] / ws[l];
			if (!seen) { p0 = a0; p1 = a1; p2 = a2; seen = true; }
			else { p0 += a0; p1 += a1; p2 += a2; }
		}
	}
	if (seen) { p0 = fmaxf(fminf(p0, 1.0f), 0.f); p1 = fmaxf(fminf(p1, 1.0f), 0.f); p2 = fmaxf(fminf(p2, 1.0f), 0.f); }
	p[0] = p0; p[1] = p1; p[2] = p2;
	if (seen && !seen0) tmask[pe] = 1;
}

// ---- CylinderProject::project (stitch/warp.cc:25-44): thread per output pixel ----
// coltc[j] = (tan, cos) of the column's angle (j - offset.x) * sizefactor_inv, from the host libm (cyl_tables below)
struct CylParams { double cx, cy, offx, offy, sizefactor_inv; int r; };
__global__ void __launch_bounds__(256) k_cyl_project(CylParams P, const double2* __restrict__ coltc, const float* __restrict__ img, int h, int w,
		float* __restrict__ out, int nh, int nw) {
	const int j = blockIdx.x * 64 + (threadIdx.x & 63);
	const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (i >= nh || j >= nw) return;
	const double py = ((double)i - P.offy) * P.sizefactor_inv;
	const double2 tc = coltc[j];
	const double ox = (double)P.r * tc.x + P.cx;                    // proj_r (warp.cc:19-23)
	const double oy = py * (double)P.r / tc.y + P.cy;
	float c[3] = {-1.f, -1.f, -1.f};
	// between(a, b, c) = a >= b && a <= c - 1 (lib/utils.hh:27)
	if (ox >= 0 && ox <= (double)(w - 1) && oy >= 0 && oy <= (double)(h - 1))
		interpolate(img, h, w, (float)oy, (float)ox, c);
	float* p = out + ((long long)i * nw + j) * 3;
	p[0] = c[0]; p[1] = c[1]; p[2] = c[2];
}

// ---- crop (lib/imgproc.cc:200-235): the largest rectangle of valid pixels ----
// Stage 1: column histograms.  height[line][k] = number of consecutive valid pixels ending at
// (line, k); thread per column walks the lines (coalesced across k).
__global__ void __launch_bounds__(256) k_crop_heights(const float* __restrict__ mat, int h, int w, int* __restrict__ height) {
	const int k = blockIdx.x * 256 + threadIdx.x;
	if (k >= w) return;
	int run = 0;
	for (int line = 0; line < h; ++line) {
		const float* p = mat + ((long long)line * w + k) * 3;
		const float m = fmaxf(fmaxf(p[0], p[1]), p[2]);
		run = m < 0 ? 0 : run + 1;                        // find Color::NO (:209)
		height[(long long)line * w + k] = run;
	}
}
// Stage 2: per line, the largest rectangle under the histogram.  left/right of the reference's
// pointer-jumping loops (:212-221) are the extents over which height >= height[k]; each thread
// finds them with a two-level search (own 64-chunk, chunk minima, target chunk).  The block's
// best (area, k) keeps the reference's first-maximum rule (:222-224: strict update in k order).
constexpr int CROP_CHUNK = 64;
__global__ void __launch_bounds__(256) k_crop_lines(const int* __restrict__ height, int h, int w, int4* __restrict__ line_best) {
	extern __shared__ int s_crop[];
	int* hs = s_crop;                    // w heights of this line
	int* cmin = hs + w;                  // minima of 64-chunks
	__shared__ int s_area[256], s_k[256], s_l[256], s_r[256];
	const int line = blockIdx.x, tid = threadIdx.x;
	const int nchunk = (w + CROP_CHUNK - 1) / CROP_CHUNK;
	for (int k = tid; k < w; k += 256) hs[k] = height[(long long)line * w + k];
	__syncthreads();
	for (int c = tid; c < nchunk; c += 256) {
		int m = 0x7fffffff;
		for (int k = c * CROP_CHUNK; k < w && k < (c + 1) * CROP_CHUNK; ++k) m = hs[k] < m ? hs[k] : m;
		cmin[c] = m;
	}
	__syncthreads();
	int barea = 0, bk = 0x7fffffff, bl = 0, br = 0;
	for (int k = tid; k < w; k += 256) {
		const int hk = hs[k];
		if (hk == 0) continue;                           // area 0 never beats maxarea (strict >, initial 0)
		// left: first j < k with hs[j] < hk, +1
		int j = k - 1;
		const int c0 = k / CROP_CHUNK;
		while (j >= c0 * CROP_CHUNK && hs[j] >= hk) --j;
		if (j < c0 * CROP_CHUNK && j >= 0) {
			int c = c0 - 1;
			while (c >= 0 && cmin[c] >= hk) --c;
			if (c < 0) j = -1;
			else { j = (c + 1) * CROP_CHUNK - 1; while (hs[j] >= hk) --j; }
		}
		const int left = j + 1;
		// right: first j > k with hs[j] < hk, -1
		j = k + 1;
		const int cend = (c0 + 1) * CROP_CHUNK < w ? (c0 + 1) * CROP_CHUNK : w;
		while (j < cend && hs[j] >= hk) ++j;
		if (j >= cend && j < w) {
			int c = c0 + 1;
			while (c < nchunk && cmin[c] >= hk) ++c;
			if (c >= nchunk) j = w;
			else { j = c * CROP_CHUNK; while (hs[j] >= hk) ++j; }
		}
		const int right = j - 1;
		const int area = (right - left + 1) * hk;
		if (area > barea) { barea = area; bk = k; bl = left; br = right; }   // ascending k per thread: first max
	}
	s_area[tid] = barea; s_k[tid] = bk; s_l[tid] = bl; s_r[tid] = br;
	__syncthreads();
	for (int st = 128; st > 0; st >>= 1) {
		if (tid < st) {
			const int oa = s_area[tid + st], ok = s_k[tid + st];
			if (oa > s_area[tid] || (oa == s_area[tid] && ok < s_k[tid])) { s_area[tid] = oa; s_k[tid] = ok; s_l[tid] = s_l[tid + st]; s_r[tid] = s_r[tid + st]; }
		}
		__syncthreads();
	}
	if (tid == 0) line_best[line] = make_int4(s_area[0], s_l[0], s_r[0], s_area[0] > 0 ? hs[s_k[0]] : 0);
}
// Stage 3: first line with the maximal area (update_max is strict, lines ascend)
__global__ void __launch_bounds__(256) k_crop_pick(const int4* __restrict__ line_best, int h, int* __restrict__ rect /* x0,y0,w,h */) {
	__shared__ int s_area[256], s_line[256];
	int ba = 0, bl = 0x7fffffff;
	for (int l = threadIdx.x; l < h; l += 256) { const int a = line_best[l].x; if (a > ba) { ba = a; bl = l; } }
	s_area[threadIdx.x] = ba; s_line[threadIdx.x] = bl;
	__syncthreads();
	for (int st = 128; st > 0; st >>= 1) {
		if (threadIdx.x < st) {
			const int oa = s_area[threadIdx.x + st], ol = s_line[threadIdx.x + st];
			if (oa > s_area[threadIdx.x] || (oa == s_area[threadIdx.x] && ol < s_line[threadIdx.x])) { s_area[threadIdx.x] = oa; s_line[threadIdx.x] = ol; }
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		if (s_area[0] <= 0) { rect[0] = 0; rect[1] = 1; rect[2] = 1; rect[3] = 0; }     // ll = rr = hh = nl = 0 (:205)
		else {
			const int4 b = line_best[s_line[0]];
			rect[0] = b.y; rect[1] = s_line[0] - b.w + 1; rect[2] = b.z - b.y + 1; rect[3] = b.w;
		}
	}
}
__global__ void __launch_bounds__(256) k_crop_copy(const float* __restrict__ src, int sw, int x0, int y0, float* __restrict__ dst, int dh, int dw) {
	const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
	if (e >= (long long)dh * dw * 3) return;
	const int row = (int)(e / (dw * 3)), c = (int)(e % (dw * 3));
	dst[e] = src[((long long)(row + y0) * sw + x0) * 3 + c];
}
// write_rgb / write_png quantisation (lib/imgio.cc:25-40,98-113): Color::NO -> white, float * 255 truncated
__global__ void __launch_bounds__(256) k_to_u8(const float* __restrict__ src, long long n, unsigned char* __restrict__ dst) {
	const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
	if (e >= n) return;
	const float v = src[e];
	dst[e] = (unsigned char)((v < 0 ? 1.f : v) * 255.f);
}

// ------------------------------- host geometry (fp64, host libm) -------------------------------
// Eigen::FullPivLU 3x3 inverse used by Homography::inverse (stitch/homography.cc:25-39)
bool inverse3_host(const double a[9], double inv[9]) {
	double lu[9]; memcpy(lu, a, sizeof(lu));
	int rowt[3], colt[3], nonzero = 3; double maxpivot = 0;
	for (int k = 0; k < 3; ++k) {
		int br = k, bc = k; double best = -1;
		for (int i = k; i < 3; ++i) for (int j = k; j < 3; ++j) { double v = std::fabs(lu[i * 3 + j]); if (v > best) { best = v; br = i; bc = j; } }
		if (best == 0.0) { nonzero = k; for (int i = k; i < 3; ++i) rowt[i] = colt[i] = i; break; }
		if (best > maxpivot) maxpivot = best;
		rowt[k] = br; colt[k] = bc;
		if (br != k) for (int j = 0; j < 3; ++j) std::swap(lu[k * 3 + j], lu[br * 3 + j]);
		if (bc != k) for (int i = 0; i < 3; ++i) std::swap(lu[i * 3 + k], lu[i * 3 + bc]);
		for (int i = k + 1; i < 3; ++i) lu[i * 3 + k] /= lu[k * 3 + k];
		for (int i = k + 1; i < 3; ++i) for (int j = k + 1; j < 3; ++j) lu[i * 3 + j] -= lu[i * 3 + k] * lu[k * 3 + j];
	}
	const double thr = std::fabs(maxpivot) * (DBL_EPSILON * 3);
	int rank = 0;
	for (int i = 0; i < nonzero; ++i) rank += (std::fabs(lu[i * 3 + i]) > thr);
	if (rank != 3) return false;
	for (int col = 0; col < 3; ++col) {
		double c[3];
		for (int i = 0; i < 3; ++i) c[i] = (i == col) ? 1.0 : 0.0;
		for (int i = 0; i < 3; ++i) std::swap(c[i], c[rowt[i]]);
		for (int i = 0; i < 3; ++i) for (int j = 0; j < i; ++j) c[i] -= lu[i * 3 + j] * c[j];
		for (int i = 2; i >= 0; --i) { for (int j = i + 1; j < 3; ++j) c[i] -= lu[i * 3 + j] * c[j]; c[i] /= lu[i * 3 + i]; }
		for (int i = 2; i >= 0; --i) std::swap(c[i], c[colt[i]]);
		for (int i = 0; i < 3; ++i) inv[i * 3 + col] = c[i];
	}
	return true;
}

void htrans_host(const double* d, double x, double y, double z, double out[3]) {
	out[0] = d[0] * x + d[1] * y + d[2] * z;
	out[1] = d[3] * x + d[4] * y + d[5] * z;
	out[2] = d[6] * x + d[7] * y + d[8] * z;
}
void homo2proj_host(int method, const double h[3], double out[2]) {   // projection.hh:16-18,33-36,48-51
	if (method == 0) { out[0] = h[0] / h[2]; out[1] = h[1] / h[2]; }
	else if (method == 1) { out[0] = std::atan2(h[0], h[2]); out[1] = h[1] / (std::hypot(h[0], h[2])); }
	else { out[0] = std::atan2(h[0], h[2]); out[1] = std::atan2(h[1], std::hypot(h[0], h[2])); }
}

void roi_of(const op_blend_geom* g, const double* range, int roi[4]) {   // Coor(double, double) truncation
	roi[0] = (int)((range[0] - g->proj_min[0]) / g->resolution[0]);
	roi[1] = (int)((range[1] - g->proj_min[1]) / g->resolution[1]);
	roi[2] = (int)((range[2] - g->proj_min[0]) / g->resolution[0]);
	roi[3] = (int)((range[3] - g->proj_min[1]) / g->resolution[1]);
}

// GaussCache (feature/gaussian.cc:17-40)
int gauss_taps(float sigma, int window_factor, BlurTaps& t) {
	int kw = (int)(std::ceil(0.3 * (sigma / 2 - 1) + 0.8) * window_factor);
	if (kw % 2 == 0) kw++;
	const int center = kw / 2;
	if (center > OP_MAX_KCENTER || kw < 1) return -1;
	float* kernel = &t.k[center];
	kernel[0] = 1;
	float exp_coeff = (float)(-1.0 / (sigma * sigma * 2)), wsum = 1;
	for (int i = 1; i <= center; i++)
		wsum += (kernel[i] = std::exp((float)(i * i) * exp_coeff)) * 2;
	float fac = (float)(1.0 / wsum);
	kernel[0] = fac;
	for (int i = 1; i <= center; i++) kernel[-i] = (kernel[i] *= fac);
	t.center = center;
	return 0;
}

struct CylProj { double cx, cy; int r, sizefactor; };
CylProj cyl_projector(int w, int h, double h_factor, float focal_length) {   // warp.cc:70-75
	CylProj p;
	p.r = (int)(std::hypot((double)w, (double)h) * (focal_length / 43.266));
	p.cx = w / 2; p.cy = h / 2 * h_factor;
	p.sizefactor = p.r;
	return p;
}
void cyl_proj(const CylProj& P, double px, double py, double out[2]) {       // warp.cc:13-17
	out[0] = std::atan((px - P.cx) / P.r);
	out[1] = (py - P.cy) / (std::hypot(px - P.cx, (double)P.r));
}

struct Freer { std::vector<void*> v; ~Freer() { for (void* p : v) pool_free(p); } };

// The transcendentals of the canvas -> space map, evaluated by the host libm exactly where the reference evaluates
// them (stitcher_image.cc:144-145 + projection.hh:38-40,66-68): per column j  sin / cos of  j * resolution.x + min.x,
// per row i  tan of  i * resolution.y + min.y  (spherical) or that value itself (cylindrical).  (w1, h1) = canvas + 1.
// Layout: [w1 x (sin, cos)][h1 x row value].  Kept on the context while the geometry stays the same.
hipError_t trig_tables(op_ctx* ctx, const BlendGeom& g, int w1, int h1, BlendTrig* out) {
	op_ctx::TrigTables& T = ctx->blend_trig;
	const bool hit = T.method == g.method && T.w1 == w1 && T.h1 == h1 && T.minx == g.minx && T.miny == g.miny && T.resx == g.resx && T.resy == g.resy && T.dev.p;
	if (!hit) {
		T.method = -1;
		T.host.resize((size_t)2 * w1 + h1);
		double* col = T.host.data(); double* row = col + (size_t)2 * w1;
		const int chunks = (w1 + h1 + 1023) / 1024;
		auto fill = [&](int c) {
			const int a = c * 1024, b = std::min(w1 + h1, a + 1024);
			for (int e = a; e < b; ++e) {
				if (e < w1) { const double x = (double)e * g.resx + g.minx; col[2 * e] = std::sin(x); col[2 * e + 1] = std::cos(x); }
				else { const int i = e - w1; const double y = (double)i * g.resy + g.miny; row[i] = g.method == 2 ? std::tan(y) : y; }
			}
		};
		if (chunks > 2) host_parallel_for(chunks, fill); else for (int c = 0; c < chunks; ++c) fill(c);
		hipError_t e = T.dev.ensure(sizeof(double) * T.host.size());
		if (e != hipSuccess) return e;
		e = hipMemcpyAsync(T.dev.p, T.host.data(), sizeof(double) * T.host.size(), hipMemcpyHostToDevice, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // the host vector may be rewritten by the next miss
		if (e != hipSuccess) return e;
		T.method = g.method; T.w1 = w1; T.h1 = h1; T.minx = g.minx; T.miny = g.miny; T.resx = g.resx; T.resy = g.resy;
	}
	out->colsc = (const double2*)T.dev.p; out->rowt = (const double*)T.dev.p + (size_t)2 * w1; out->w1 = w1; out->h1 = h1;
	return hipSuccess;
}
// CylinderProject::project's per-column tan / cos (warp.cc:19-23,31): [nw x (tan, cos)]
hipError_t cyl_tables(op_ctx* ctx, double offx, double sizefactor_inv, int nw, const double2** out) {
	op_ctx::TrigTables& T = ctx->cyl_trig;
	const bool hit = T.method == 3 && T.w1 == nw && T.minx == offx && T.resx == sizefactor_inv && T.dev.p;
	if (!hit) {
		T.method = -1;
		T.host.resize((size_t)2 * nw);
		double* col = T.host.data();
		for (int j = 0; j < nw; ++j) { const double px = ((double)j - offx) * sizefactor_inv; col[2 * j] = std::tan(px); col[2 * j + 1] = std::cos(px); }
		hipError_t e = T.dev.ensure(sizeof(double) * T.host.size());
		if (e != hipSuccess) return e;
		e = hipMemcpyAsync(T.dev.p, T.host.data(), sizeof(double) * T.host.size(), hipMemcpyHostToDevice, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess) return e;
		T.method = 3; T.w1 = nw; T.minx = offx; T.resx = sizefactor_inv;
	}
	*out = (const double2*)T.dev.p;
	return hipSuccess;
}

}	// namespace

extern "C" {

int op_blend_prepare(const op_config* cfg, int proj_method, int identity_idx, int n, const int* shapes_wh,
		const double* homo, op_blend_geom* g, double* homo_inv, double* ranges) {
	if (!cfg || !shapes_wh || !homo || !g || !homo_inv || !ranges || n <= 0 || identity_idx < 0 || identity_idx >= n ||
			proj_method < 0 || proj_method > 2)
		OP_FAIL(OP_ERR_INVALID, "op_blend_prepare: bad argument");
	for (int i = 0; i < n; ++i)
		if (!inverse3_host(homo + 9 * i, homo_inv + 9 * i))
			OP_FAIL(OP_ERR_INVALID, "op_blend_prepare: homography " + std::to_string(i) + " is not invertible (homography.cc:33)");
	const int CORNER_SAMPLE = 100;        // stitcher_image.cc:43
	std::vector<double> cx, cy;
	for (int i = 0; i < CORNER_SAMPLE; ++i) {
		const double xi = (double)i / CORNER_SAMPLE - 0.5;
		cx.push_back(xi); cy.push_back(-0.5);
		cx.push_back(xi); cy.push_back(0.5);
	}
	for (int j = 0; j < CORNER_SAMPLE; ++j) {
		const double yj = (double)j / CORNER_SAMPLE - 0.5;
		cx.push_back(-0.5); cy.push_back(yj);
		cx.push_back(0.5); cy.push_back(yj);
	}
	double pmin[2] = {DBL_MAX, DBL_MAX}, pmax[2] = {-DBL_MAX, -DBL_MAX};
	for (int m = 0; m < n; ++m) {
		const int w = shapes_wh[2 * m], h = shapes_wh[2 * m + 1];
		double nmin[2] = {DBL_MAX, DBL_MAX}, nmax[2] = {-DBL_MAX, -DBL_MAX};
		for (size_t k = 0; k < cx.size(); ++k) {
			double hv[3], t[2];
			htrans_host(homo + 9 * m, cx[k] * w, cy[k] * h, 1, hv);
			homo2proj_host(proj_method, hv, t);
			for (int c = 0; c < 2; ++c) { if (t[c] < nmin[c]) nmin[c] = t[c]; if (nmax[c] < t[c]) nmax[c] = t[c]; }
		}
		ranges[4 * m] = nmin[0]; ranges[4 * m + 1] = nmin[1]; ranges[4 * m + 2] = nmax[0]; ranges[4 * m + 3] = nmax[1];
		for (int c = 0; c < 2; ++c) { if (nmin[c] < pmin[c]) pmin[c] = nmin[c]; if (pmax[c] < nmax[c]) pmax[c] = nmax[c]; }
	}
	g->proj_method = proj_method;
	g->proj_min[0] = pmin[0]; g->proj_min[1] = pmin[1]; g->proj_max[0] = pmax[0]; g->proj_max[1] = pmax[1];
	// get_final_resolution (stitcher_image.cc:79-114)
	const int refw = shapes_wh[2 * identity_idx], refh = shapes_wh[2 * identity_idx + 1];
	double c2[3], c1[3], p2[2], p1[2];
	htrans_host(homo + 9 * identity_idx, refw / 2.0, refh / 2.0, 1, c2);
	htrans_host(homo + 9 * identity_idx, -refw / 2.0, -refh / 2.0, 1, c1);
	homo2proj_host(proj_method, c2, p2); homo2proj_host(proj_method, c1, p1);
	double rx = p2[0] - p1[0], ry = p2[1] - p1[1];
	if (proj_method != 0) {
		if (rx < 0) rx = 2 * M_PI + rx;
		if (ry < 0) ry = M_PI + ry;
	}
	double resx = std::fabs(rx) / (double)refw, resy = std::fabs(ry) / (double)refh;
	const double tsx = (pmax[0] - pmin[0]) / resx, tsy = (pmax[1] - pmin[1]) / resy;
	const double max_edge = std::max(tsx, tsy);
	if (max_edge > 80000 || tsx * tsy > 1e9)
		OP_FAIL(OP_ERR_INVALID, "Target size too large. Looks like a stitching failure!");   // stitcher_image.cc:105-106
	if (max_edge > cfg->MAX_OUTPUT_SIZE) {
		const float ratio = (float)(max_edge / cfg->MAX_OUTPUT_SIZE);
		resx *= ratio; resy *= ratio;
	}
	g->resolution[0] = resx; g->resolution[1] = resy;
	return OP_OK;
}

int op_blend_canvas_dims(const op_blend_geom* g, const op_blend_image* imgs, int n, int* h, int* w) {
	if (!g || !imgs || n <= 0 || !h || !w) OP_FAIL(OP_ERR_INVALID, "op_blend_canvas_dims: bad argument");
	int tx = 0, ty = 0;       // Coor target_size{0,0}; update_max(bottom_right) (blender.cc:21)
	for (int i = 0; i < n; ++i) {
		int roi[4]; roi_of(g, imgs[i].range, roi);
		tx = std::max(tx, roi[2]); ty = std::max(ty, roi[3]);
	}
	*h = ty; *w = tx;
	return OP_OK;
}

int op_blend(op_ctx* ctx, const op_config* cfg, const op_blend_geom* g, const op_blend_image* imgs, int n, op_canvas** out) {
	if (!ctx || !cfg || !g || !imgs || n <= 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_blend: bad argument");
	if (g->proj_method < 0 || g->proj_method > 2) OP_FAIL(OP_ERR_INVALID, "op_blend: bad projection method");
	if (!(g->resolution[0] > 0) || !(g->resolution[1] > 0)) OP_FAIL(OP_ERR_INVALID, "op_blend: resolution must be positive");
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t st = ctx->stream;
	int H, W;
	int rc = op_blend_canvas_dims(g, imgs, n, &H, &W);
	if (rc != OP_OK) return rc;
	if (H <= 0 || W <= 0) OP_FAIL(OP_ERR_INVALID, "op_blend: empty canvas");
	Freer fr;
	std::vector<BlendImg> h_imgs(n);
	long long roi_total = 0; long long max_roi = 0;
	for (int k = 0; k < n; ++k) {
		const op_blend_image& s = imgs[k];
		if (!s.data || s.h < 2 || s.w < 2) OP_FAIL(OP_ERR_INVALID, "op_blend: bad image " + std::to_string(k));
		BlendImg& b = h_imgs[k];
		b.h = s.h; b.w = s.w;
		b.mh = s.mat_h > 0 ? s.mat_h : s.h; b.mw = s.mat_w > 0 ? s.mat_w : s.w;
		if (b.mh < 2 || b.mw < 2) OP_FAIL(OP_ERR_INVALID, "op_blend: bad pixel buffer size of image " + std::to_string(k));
		if (s.on_device) b.data = s.data;
		else {
			float* d = nullptr;
			HIPCHK(pool_alloc((void**)&d, sizeof(float) * 3 * (size_t)b.mh * b.mw)); fr.v.push_back(d);
			HIPCHK(hipMemcpyAsync(d, s.data, sizeof(float) * 3 * (size_t)b.mh * b.mw, hipMemcpyHostToDevice, st));
			b.data = d;
		}
		int roi[4]; roi_of(g, s.range, roi);
		if (roi[0] < 0 || roi[1] < 0 || roi[2] < roi[0] || roi[3] < roi[1]) OP_FAIL(OP_ERR_INVALID, "op_blend: image range outside proj_range");
		b.x0 = roi[0]; b.y0 = roi[1]; b.x1 = roi[2]; b.y1 = roi[3];
		memcpy(b.hinv, s.homo_inv, sizeof(b.hinv));
		b.rw = roi[2] - roi[0] + 1; b.rh = roi[3] - roi[1] + 1;      // Range::width/height, inclusive
		b.roi_off = roi_total; roi_total += (long long)b.rw * b.rh;
		max_roi = std::max(max_roi, (long long)b.rw * b.rh);
	}
	BlendImg* d_imgs = nullptr;
	HIPCHK(pool_alloc((void**)&d_imgs, sizeof(BlendImg) * n)); fr.v.push_back(d_imgs);
	HIPCHK(hipMemcpyAsync(d_imgs, h_imgs.data(), sizeof(BlendImg) * n, hipMemcpyHostToDevice, st));
	op_canvas* cv = new op_canvas;
	cv->h = H; cv->w = W; cv->device = ctx->device;
	if (pool_alloc((void**)&cv->data, sizeof(float) * 3 * (size_t)H * W) != hipSuccess) { delete cv; OP_FAIL(OP_ERR_HIP, "op_blend: canvas allocation failed"); }
	const BlendGeom bg{g->proj_method, g->proj_min[0], g->proj_min[1], g->resolution[0], g->resolution[1]};
	const dim3 cgrid((W + 63) / 64, (H + 3) / 4);
	BlendTrig trig{nullptr, nullptr, 0, 0};
	if (bg.method != 0) {
		HostScope hs(ctx, "blend trig tables (host)");
		hipError_t e = trig_tables(ctx, bg, W + 1, H + 1, &trig);
		if (e != hipSuccess) { pool_free(cv->data); delete cv; OP_FAIL(OP_ERR_HIP, std::string("op_blend: trig tables: ") + hipGetErrorString(e)); }
	}
#define BCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { op_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
	pool_free(cv->data); delete cv; return OP_ERR_HIP; } } while (0)
	if (cfg->MULTIBAND <= 0) {
		ProfScope ps(ctx, "blend linear");
		hipLaunchKernelGGL(k_blend_linear, cgrid, dim3(256), 0, st, bg, trig, d_imgs, n, cv->data, H, W, cfg->ORDERED_INPUT, cfg->LAZY_READ);
		BCHK(hipGetLastError());
	} else {
		const int L = cfg->MULTIBAND;
		// every level's plane is kept when they fit comfortably (L x 16 bytes per ROI pixel: 0.2 GB per level for 38 views):
		// the bands are then applied in one pass over the canvas (k_mb_bands); otherwise two planes alternate and every
		// level has its own band pass
		const bool keep_all = L <= OP_MB_MAX_LEVELS && sizeof(float4) * (size_t)roi_total * (size_t)L <= ((size_t)64 << 30);
		const int nplanes = keep_all ? L : 2;
		std::vector<float4*> lv(nplanes, nullptr);
		float4* tmp = nullptr; unsigned char *mask = nullptr, *tmask = nullptr;
		for (int l = 0; l < nplanes; ++l) { BCHK(pool_alloc((void**)&lv[l], sizeof(float4) * roi_total)); fr.v.push_back(lv[l]); }
		BCHK(pool_alloc((void**)&mask, roi_total)); fr.v.push_back(mask);
		BCHK(pool_alloc((void**)&tmask, (size_t)H * W)); fr.v.push_back(tmask);
		const dim3 rgrid((unsigned)((max_roi + 255) / 256), n);
		{ ProfScope ps(ctx, "multiband first level");
		  hipLaunchKernelGGL(k_mb_first_fused, dim3((W + 1 + 63) / 64, (H + 1 + 3) / 4), dim3(256), 0, st, bg, trig, d_imgs, n, lv[0], mask, cv->data, tmask, H, W);
		  BCHK(hipGetLastError()); }
		bool band0_done = false;                 // level 0's band written by the fused blur
		for (int level = 0; level < L; ++level) {
			const int is_last = (level == L - 1);
			float4* cur = keep_all ? lv[level] : lv[level & 1];
			float4* nxt = is_last ? nullptr : (keep_all ? lv[level + 1] : lv[(level + 1) & 1]);
			bool band_done = false;
			if (!is_last) {
				ProfScope ps(ctx, "multiband blur");
				BlurTaps taps; memset(&taps, 0, sizeof(taps));
				if (gauss_taps((float)(std::sqrt(level * 2 + 1.0) * 4), cfg->GAUSS_WINDOW_FACTOR, taps) != 0) {
					pool_free(cv->data); delete cv; OP_FAIL(OP_ERR_UNSUPPORTED, "op_blend: Gaussian kernel wider than 31 taps");
				}
				if (taps.center == 6 || taps.center == 9) {       // shipped GAUSS_WINDOW_FACTOR: both passes in one kernel
					const int C = taps.center, two = 256 - 2 * C, segr = (C <= 6 ? 8 : 6) * (2 * C + 2);      // k_mb_blur_fused: SEG
					unsigned items = 1;
					for (int k = 0; k < n; ++k)
						items = std::max(items, (unsigned)(((h_imgs[k].rw + two - 1) / two) * ((h_imgs[k].rh + segr - 1) / segr)));
					band_done = level == 0;
					if (C == 6 && level == 0) hipLaunchKernelGGL((k_mb_blur_fused<6, true>), dim3(items, n), dim3(256), 0, st, d_imgs, taps, cur, nxt, cv->data, tmask, H, W);
					else if (C == 6) hipLaunchKernelGGL((k_mb_blur_fused<6, false>), dim3(items, n), dim3(256), 0, st, d_imgs, taps, cur, nxt, cv->data, tmask, H, W);
					else if (level == 0) hipLaunchKernelGGL((k_mb_blur_fused<9, true>), dim3(items, n), dim3(256), 0, st, d_imgs, taps, cur, nxt, cv->data, tmask, H, W);
					else hipLaunchKernelGGL((k_mb_blur_fused<9, false>), dim3(items, n), dim3(256), 0, st, d_imgs, taps, cur, nxt, cv->data, tmask, H, W);
				} else {
					if (!tmp) { BCHK(pool_alloc((void**)&tmp, sizeof(float4) * roi_total)); fr.v.push_back(tmp); }
					hipLaunchKernelGGL((k_mb_blur<true, 0>), rgrid, dim3(256), 0, st, d_imgs, taps, cur, tmp);
					hipLaunchKernelGGL((k_mb_blur<false, 0>), rgrid, dim3(256), 0, st, d_imgs, taps, tmp, nxt);
				}
				BCHK(hipGetLastError());
			}
			if (level == 0) band0_done = band_done;
			if (!keep_all && !band_done) { ProfScope ps(ctx, "multiband band");
			  hipLaunchKernelGGL(k_mb_accumulate, cgrid, dim3(256), 0, st, d_imgs, n, cur, nxt, mask, cv->data, tmask, H, W, is_last);
			  BCHK(hipGetLastError()); }
		}
		if (keep_all) {
			ProfScope ps(ctx, "multiband band");
			const int first = band0_done ? 1 : 0, NL = L - first;
			BandPlanes P; memset(&P, 0, sizeof(P));
			for (int l = 0; l < NL; ++l) P.lv[l] = lv[first + l];
			switch (NL) {
#define OP_MB_CASE(N) case N: hipLaunchKernelGGL((k_mb_bands<N>), cgrid, dim3(256), 0, st, d_imgs, n, P, mask, cv->data, tmask, H, W); break;
				OP_MB_CASE(1) OP_MB_CASE(2) OP_MB_CASE(3) OP_MB_CASE(4) OP_MB_CASE(5) OP_MB_CASE(6)
#undef OP_MB_CASE
				default: break;
			}
			BCHK(hipGetLastError());
		}
	}
	BCHK(hipStreamSynchronize(st));
#undef BCHK
	resolve_profile(ctx);
	*out = cv;
	return OP_OK;
}

int op_canvas_dims(const op_canvas* c, int* h, int* w) {
	if (!c || !h || !w) OP_FAIL(OP_ERR_INVALID, "op_canvas_dims: bad argument");
	*h = c->h; *w = c->w; return OP_OK;
}
const float* op_canvas_device(const op_canvas* c) { return c ? c->data : nullptr; }
int op_canvas_copy(op_ctx* ctx, const op_canvas* c, float* host) {
	if (!ctx || !c || !host) OP_FAIL(OP_ERR_INVALID, "op_canvas_copy: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipMemcpyAsync(host, c->data, sizeof(float) * 3 * (size_t)c->h * c->w, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return OP_OK;
}
void op_canvas_free(op_canvas* c) {
	if (!c) return;
	hipSetDevice(c->device);
	pool_free(c->data);
	delete c;
}

int op_canvas_crop(op_ctx* ctx, const op_canvas* c, op_canvas** out, int* x0, int* y0) {
	if (!ctx || !c || !out) OP_FAIL(OP_ERR_INVALID, "op_canvas_crop: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t st = ctx->stream;
	const int h = c->h, w = c->w;
	if ((size_t)(w + (w + CROP_CHUNK - 1) / CROP_CHUNK) * sizeof(int) > 150 * 1024) OP_FAIL(OP_ERR_UNSUPPORTED, "op_canvas_crop: canvas wider than 38000 px");
	Freer fr;
	int* d_height = nullptr; int4* d_best = nullptr; int* d_rect = nullptr;
	HIPCHK(pool_alloc((void**)&d_height, sizeof(int) * (size_t)h * w)); fr.v.push_back(d_height);
	HIPCHK(pool_alloc((void**)&d_best, sizeof(int4) * h)); fr.v.push_back(d_best);
	HIPCHK(pool_alloc((void**)&d_rect, sizeof(int) * 4)); fr.v.push_back(d_rect);
	HIPCHK(hipFuncSetAttribute((const void*)k_crop_lines, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));   // idempotent
	int rect[4];
	{ ProfScope ps(ctx, "crop");
	  hipLaunchKernelGGL(k_crop_heights, dim3((w + 255) / 256), dim3(256), 0, st, c->data, h, w, d_height);
	  HIPCHK(hipGetLastError());
	  const size_t lds = sizeof(int) * (size_t)(w + (w + CROP_CHUNK - 1) / CROP_CHUNK);
	  hipLaunchKernelGGL(k_crop_lines, dim3(h), dim3(256), lds, st, d_height, h, w, d_best);
	  HIPCHK(hipGetLastError());
	  hipLaunchKernelGGL(k_crop_pick, dim3(1), dim3(256), 0, st, d_best, h, d_rect);
	  HIPCHK(hipGetLastError()); }
	HIPCHK(hipMemcpyAsync(rect, d_rect, sizeof(rect), hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	op_canvas* cv = new op_canvas;
	cv->h = rect[3]; cv->w = rect[2]; cv->device = ctx->device;
	const size_t n = (size_t)cv->h * cv->w * 3;
	if (pool_alloc((void**)&cv->data, sizeof(float) * (n ? n : 1)) != hipSuccess) { delete cv; OP_FAIL(OP_ERR_HIP, "op_canvas_crop: allocation failed"); }
	if (n) {
		hipLaunchKernelGGL(k_crop_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c->data, w, rect[0], rect[1], cv->data, cv->h, cv->w);
		hipError_t e = hipGetLastError();
		if (e == hipSuccess) e = hipStreamSynchronize(st);
		if (e != hipSuccess) { pool_free(cv->data); delete cv; OP_FAIL(OP_ERR_HIP, std::string("op_canvas_crop: ") + hipGetErrorString(e)); }
	}
	resolve_profile(ctx);
	if (x0) *x0 = rect[0];
	if (y0) *y0 = rect[1];
	*out = cv;
	return OP_OK;
}

int op_canvas_copy_u8(op_ctx* ctx, const op_canvas* c, unsigned char* host) {
	if (!ctx || !c || !host) OP_FAIL(OP_ERR_INVALID, "op_canvas_copy_u8: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	const long long n = (long long)c->h * c->w * 3;
	if (n == 0) return OP_OK;
	Freer fr;
	unsigned char* d = nullptr;
	HIPCHK(pool_alloc((void**)&d, (size_t)n)); fr.v.push_back(d);
	hipLaunchKernelGGL(k_to_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, c->data, n, d);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(host, d, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return OP_OK;
}

int op_cyl_warp_shape(const op_config* cfg, int w, int h, double h_factor, double* pts, int npts,
		int* new_w, int* new_h, double* offset) {
	if (!cfg || w < 2 || h < 2 || npts < 0 || (npts && !pts) || !new_w || !new_h || !offset)
		OP_FAIL(OP_ERR_INVALID, "op_cyl_warp_shape: bad argument");
	const CylProj P = cyl_projector(w, h, h_factor, cfg->FOCAL_LENGTH);
	if (P.r <= 0) OP_FAIL(OP_ERR_INVALID, "op_cyl_warp_shape: degenerate projector radius");
	// warp.cc:47-52 scans all w*h pixels.  x = atan((j-cx)/r) does not depend on i, and for a fixed
	// j the quotient y = (i-cy)/hypot(j-cx, r) is monotone in i in floating point as well (one
	// subtraction and one division by a fixed positive number), so rows 0 and h-1 hold both
	// extremes of every column: 2w evaluations give the identical min/max.
	double mn[2] = {DBL_MAX, DBL_MAX}, mx[2] = {0, 0};
	for (int j = 0; j < w; ++j) for (int e = 0; e < 2; ++e) {
		double c[2]; cyl_proj(P, j, e ? h - 1 : 0, c);
		for (int q = 0; q < 2; ++q) { if (c[q] < mn[q]) mn[q] = c[q]; if (mx[q] < c[q]) mx[q] = c[q]; }
	}
	for (int q = 0; q < 2; ++q) { mx[q] = mx[q] * P.sizefactor; mn[q] = mn[q] * P.sizefactor; }
	const double rsx = mx[0] - mn[0], rsy = mx[1] - mn[1];
	offset[0] = mn[0] * (-1); offset[1] = mn[1] * (-1);
	const int sx = (int)rsx, sy = (int)rsy;
	for (int k = 0; k < npts; ++k) {              // warp.cc:59-65
		double c[2];
		cyl_proj(P, pts[2 * k] + w / 2, pts[2 * k + 1] + h / 2, c);
		pts[2 * k] = c[0] * P.sizefactor + offset[0];
		pts[2 * k + 1] = c[1] * P.sizefactor + offset[1];
		pts[2 * k] -= sx / 2;
		pts[2 * k + 1] -= sy / 2;
	}
	*new_w = sx; *new_h = sy;
	return OP_OK;
}

int op_cyl_warp(op_ctx* ctx, const op_config* cfg, const op_image* img, double h_factor, op_canvas** out) {
	if (!ctx || !cfg || !img || !img->data || !out) OP_FAIL(OP_ERR_INVALID, "op_cyl_warp: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t st = ctx->stream;
	int nw, nh; double off[2];
	int rc = op_cyl_warp_shape(cfg, img->w, img->h, h_factor, nullptr, 0, &nw, &nh, off);
	if (rc != OP_OK) return rc;
	if (nw <= 0 || nh <= 0) OP_FAIL(OP_ERR_INVALID, "op_cyl_warp: empty output");
	Freer fr;
	if (img->dtype != OP_F32) OP_FAIL(OP_ERR_UNSUPPORTED, "op_cyl_warp: fp32 images only");
	const float* src = (const float*)img->data;
	if (!img->on_device) {
		float* d = nullptr;
		HIPCHK(pool_alloc((void**)&d, sizeof(float) * 3 * (size_t)img->h * img->w)); fr.v.push_back(d);
		HIPCHK(hipMemcpyAsync(d, img->data, sizeof(float) * 3 * (size_t)img->h * img->w, hipMemcpyHostToDevice, st));
		src = d;
	}
	const CylProj P = cyl_projector(img->w, img->h, h_factor, cfg->FOCAL_LENGTH);
	op_canvas* cv = new op_canvas;
	cv->h = nh; cv->w = nw; cv->device = ctx->device;
	if (pool_alloc((void**)&cv->data, sizeof(float) * 3 * (size_t)nh * nw) != hipSuccess) { delete cv; OP_FAIL(OP_ERR_HIP, "op_cyl_warp: allocation failed"); }
	const CylParams cp{P.cx, P.cy, off[0], off[1], 1.0 / P.sizefactor, P.r};
	const double2* coltc = nullptr;
	{ hipError_t e = cyl_tables(ctx, cp.offx, cp.sizefactor_inv, nw, &coltc);
	  if (e != hipSuccess) { pool_free(cv->data); delete cv; OP_FAIL(OP_ERR_HIP, std::string("op_cyl_warp: trig table: ") + hipGetErrorString(e)); } }
	{ ProfScope ps(ctx, "cylinder warp");
	  hipLaunchKernelGGL(k_cyl_project, dim3((nw + 63) / 64, (nh + 3) / 4), dim3(256), 0, st, cp, coltc, src, img->h, img->w, cv->data, nh, nw); }
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	if (e != hipSuccess) { pool_free(cv->data); delete cv; OP_FAIL(OP_ERR_HIP, std::string("op_cyl_warp: ") + hipGetErrorString(e)); }
	resolve_profile(ctx);
	*out = cv;
	return OP_OK;
}

}	// extern "C"
