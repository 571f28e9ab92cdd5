// descriptor.hip -- 4x4x8 RootSIFT descriptor, one wavefront per oriented keypoint.
//
// Replaces SIFT::calc_descriptor / trilinear_interpolate / hist_to_descriptor
// (feature/sift.cc:15-152) and the coordinate shift of FeatureDetector::detect_feature
// (feature/feature.cc:20-28).
//
// Bit-exactness: the reference accumulates hist[bin] += w in window order (xx outer, yy inner)
// in fp32, so the summation ORDER of every one of the 128 bins is part of the result.
// Structure per keypoint (one wavefront):
//   1a  all 64 lanes run the cheap window tests (bounds, circle, rotated bin range) in the
//       reference's sample order; survivors are queued IN ORDER (wave ballot ranks);
//   1b  dense batches of 64 queued survivors: gradient magnitude / orientation on the Gaussian
//       plane, weight, and the <= 8 trilinear contributions (bin, value) of each sample
//       (sift.cc:48-67) -- in registers;
//   2   a stable counting sort of the batch's contributions by bin, without ballots: every lane
//       ORs its lane bit into the 64-bit LDS mask of each bin it touches (order-free atomics),
//       a contribution's rank inside its bin is the popcount of that mask below the lane, bin
//       offsets are a scan of the mask popcounts; the values land bin-major in LDS, in sample
//       order inside each bin;  (a workgroup is ONE wavefront: its LDS accesses execute in
//       program order, a compiler fence orders them, nothing has to wait)
//   3   the non-empty lists (a third of the 128 bins per batch, of very uneven lengths) are dealt
//       one per lane; a lane continues its bin's running fp32 sum (the histogram lives in LDS)
//       through the list in order -- every bin sees exactly the reference's sequence of additions.
// The window is not scanned whole: per window column only the rows that can pass the circle and
// rotated-square tests are enumerated (a conservative interval), the exact tests decide.
#include "internal.hpp"
#include "devmath.hpp"
#include <stddef.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef DESC_WAVES
#define DESC_WAVES 6          // resident wavefronts per SIMD the register allocation aims at (80 VGPRs; 5.1 KB of LDS allow 8)
#endif
#ifndef DESC_GRID_MULT
#define DESC_GRID_MULT 8
#endif
constexpr int QCAP = 128;            // survivor queue (power of two, >= 2 x 64)
constexpr int COLCAP = 1024;         // candidate samples of one keypoint in the column-interval enumeration (the shipped config needs < 800)
// List arena of one batch's counting sort, in floats.  64 samples x 8 contributions, every non-empty list padded to a
// multiple of 4: at most 512 + 3 x 128 = 896 -- when ALL 128 bins are hit by 1 (mod 4) contributions.  The arena holds
// 640: a batch whose padded lists need more (seen on no keypoint of the test and bench images) is sorted and accumulated in
// two passes, lanes 0..31 then lanes 32..63 (at most 256 + 384 slots each) -- sample order inside every bin is kept.
// What the arena saves is occupancy: this kernel waits on LDS and memory round trips, more resident wavefronts hide them.
constexpr int LIST_CAP = OP_DESC_LIST_CAP;
constexpr int ZERO4 = LIST_CAP;      // one float4 of zeros: what a lane reads once its own list has ended

struct DescLds {
	float sorted[LIST_CAP + 4] __attribute__((aligned(16)));   // the batch's contributions, bin-major, sample order inside a bin; every list starts on a 16-byte boundary and is zero-padded to a multiple of 4
	unsigned long long mask[128];    // per bin: bit l = lane l's sample of the current batch contributes; after the scatter its first 512 bytes hold the batch's list records
	float hist[128];                 // the keypoint's histogram (sift.cc:105): the running fp32 sum of every bin
	unsigned short off[128];         // byte offset of every bin's first slot in sorted[]
	unsigned q[QCAP];                // survivor queue (ring): window position (xx + 32768) | (yy + 32768) << 16 (16 + 16 bits: any window radius)
	uint64_t exptab[32];             // glibc's exp2f table (devmath.hpp), staged once per workgroup
	unsigned long long startbits[COLCAP / 64];   // bit e: candidate e is the first of its window column
	unsigned colpk[64];              // k-th non-empty window column: index of its first candidate << 16 | (first candidate row & 0xFF) << 8 | column
};
// LDS accesses of ONE wavefront execute in program order; what has to be kept from moving is the compiler
#define WAVE_FENCE() asm volatile("" ::: "memory")

// Wave64 inclusive add-scan and max-reduction on the VALU data-parallel primitives (row_shr within the
// four rows of 16 lanes, then row_bcast:15 / row_bcast:31 across rows): a dozen VALU instructions instead
// of six ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ int wave_scan_add(int v) {
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);      // row_shr:1
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);      // row_shr:2
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);      // row_shr:4
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);      // row_shr:8
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);      // row_bcast:15 -> rows 1, 3
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);      // row_bcast:31 -> rows 2, 3
	return v;
}
__device__ __forceinline__ int wave_max_i(int v) {                        // v >= 0
	int t;
	t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false); v = t > v ? t : v;
	t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false); v = t > v ? t : v;
	t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false); v = t > v ? t : v;
	t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false); v = t > v ? t : v;
	t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); v = t > v ? t : v;
	t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); v = t > v ? t : v;
	return __builtin_amdgcn_readlane(v, 63);
}

// popcount of the mask bits BELOW this lane: v_mbcnt_lo + v_mbcnt_hi, no explicit lane mask
__device__ __forceinline__ int rank_below(unsigned long long m) {
	return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// A value every lane holds alike, moved to scalar registers (v_readfirstlane): the per-keypoint constants are computed by
// the vector unit (there is no scalar float arithmetic) but need not occupy 64 lanes of a vector register for the
// whole keypoint -- the registers they free are resident wavefronts.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ double uni(double v) {
	const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
	const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
	return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Launch shape: up to DESC_GRID_MULT times the wavefronts the device holds at this kernel's occupancy, each
// taking every gridDim-th keypoint; no workgroups far past the device-side count.  Measured on 46 K keypoints: exactly
// the resident wavefronts 0.338 ms (no slack to balance the keypoints' very different windows), 8 x / 16 x / 32 x
// 0.297 ms; keypoints drawn from ONE atomic ticket counter 0.62 ms (46 K atomics on one address outlast the kernel).
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DESC_WAVES, DESC_WAVES))) k_descriptor(SiftPlan p, const KeyPoint* __restrict__ oriented,
		const long long* __restrict__ total_ptr, long long cap, float* __restrict__ desc, double* __restrict__ coor, double* __restrict__ real) {
	__shared__ DescLds S;
	const int lane = threadIdx.x;
	const float pi2 = (float)(2 * 3.14159265358979323846);
	const float nbin_per_rad = 8 / pi2;
	S.mask[lane] = 0ULL; S.mask[lane + 64] = 0ULL;
	if (lane < 4) S.sorted[ZERO4 + lane] = 0.f;
	if (lane < 32) S.exptab[lane] = opdev::kExp2fTab[lane];
	// the two bins this lane owns: histogram bins lane and lane + 64 = cells (lane >> 3) and (lane >> 3) + 8 (two
	// cell rows apart: when one is crowded the other is not, which evens the list lengths across the wave)
	const int ownp = (lane & ~7) | ((lane + 7) & 7);        // the same cell's previous orientation bin
	const int list_cap4 = p.desc_list_cap >> 2;             // LIST_CAP / 4 (a test may lower it to force the two-pass sort)
	__syncthreads();

	long long total = *total_ptr;                      // device-side count (k_expand_oriented)
	total = total < cap ? total : cap;                 // speculative capacity: the host re-runs on overflow
	for (long long kk = blockIdx.x; kk < total; kk += gridDim.x) {
		const KeyPoint kp = oriented[kk];
		const int img = uni(kp.pad);                       // written by k_expand_oriented
		const int kpx = uni(kp.x), kpy = uni(kp.y), koct = uni(kp.oct), kscale = uni(kp.scale);
		const OctDesc od = p.oct[koct];
		const int w = od.w, h = od.h;
		const float* base = p.ws + (long long)img * p.ws_stride;
		// mag / ort of GaussianPyramid::cal_mag_ort (feature/dog.cc:76-84) are evaluated on the
		// Gaussian plane for the surviving window samples only
		const float* g_img = base + plane_off_gauss(od, p.nscale, kscale);
		const float ort = uni(kp.dir);
		const float hist_w = uni(kp.sf * (float)p.desc_scale_factor);
		// x / hist_w for the two rotated coordinates of every window sample: (float)((double)x * rd) with
		// rd = 1 / (double)hist_w is the correctly rounded fp32 quotient (the double product is within 2^-52
		// of x / hist_w, and a quotient of two fp32 numbers is never closer than 2^-49 (relative) to a
		// rounding boundary of fp32), at a third of the instructions of an IEEE fp32 division
		const double rd = uni(1.0 / (double)hist_w);
		const float exp_denom = 2 * (4.f * 4.f);
		const int radius = uni((int)round(0.70710678118654752440 * (double)hist_w * (4 + 1)));
		const float cosort = uni(opdev::cosf_glibc(ort)), sinort = uni(opdev::sinf_glibc(ort));
		const int side = 2 * radius + 1, nsamp = side * side;
		const float fr2 = (float)radius * (float)radius;
		S.hist[lane] = 0.f; S.hist[lane + 64] = 0.f;
		int qhead = 0, qn = 0;            // survivor queue state (wave-uniform)

		// ---- candidate enumeration.  The reference visits the whole (2 radius + 1)^2 window in (xx outer,
		// yy inner) order and keeps a sample only inside the circle and the rotated 4 x 4 bin square
		// (sift.cc:110-126) -- about a third of the window.  Each window column's rows that CAN pass are an
		// interval, computed here with a safety margin (tests/test_descriptor_candidate_intervals.py: 1.02 candidates
		// per kept sample, none lost); the exact float tests of the reference then run on those candidates only, in
		// the reference's order.  Windows wider than 64
		// columns or with more than COLCAP candidates (other DESC_HIST_SCALE_FACTORs) walk the full window.
		int ncand = nsamp;
		bool cols = side <= 64;
		if (cols) {
			int lo = 0, len = 0;
			if (lane < side) {
				const int xx = lane - radius;
				const float fxx = (float)xx;
				const int nowx = kpx + xx;
				// inside the circle (sift.cc:115, a test on integers):  |yy| <= floor(sqrt(radius^2 - xx^2))
				const float yc = floorf(sqrtf(fr2 - fxx * fxx) + 1e-3f);
				float ylo = -yc, yhi = yc;
				// -2.5 <= rot / hist_w <= 1.5 for both rotated coordinates (bin in [-1, 3] after the +1.5 shift), with a
				// margin of a twentieth of a pixel and more (the fp32 roundings of either side are below 1e-5 pixels)
				const float m = 0.005f * hist_w + 0.05f;
				const float blo = -2.5f * hist_w - m, bhi = 1.5f * hist_w + m;
				if (fabsf(cosort) > 0.01f) {           // y_rot * hist_w = -xx sin + yy cos
					const float rc = 1.f / cosort;
					const float a = (blo + fxx * sinort) * rc, b = (bhi + fxx * sinort) * rc;
					ylo = fmaxf(ylo, fminf(a, b)); yhi = fminf(yhi, fmaxf(a, b));
				}
				if (fabsf(sinort) > 0.01f) {           // x_rot * hist_w = xx cos + yy sin
					const float rs = 1.f / sinort;
					const float a = (blo - fxx * cosort) * rs, b = (bhi - fxx * cosort) * rs;
					ylo = fmaxf(ylo, fminf(a, b)); yhi = fminf(yhi, fmaxf(a, b));
				}
				int ilo = (int)ceilf(ylo), ihi = (int)floorf(yhi);
				ilo = ilo < -radius ? -radius : ilo; ihi = ihi > radius ? radius : ihi;
				ilo = ilo < 1 - kpy ? 1 - kpy : ilo; ihi = ihi > h - 2 - kpy ? h - 2 - kpy : ihi;     // between(nowy, 1, h - 1)
				if (nowx >= 1 && nowx <= w - 2 && ihi >= ilo) { lo = ilo; len = ihi - ilo + 1; }
			}
			const int incl = wave_scan_add(len);
			ncand = __builtin_amdgcn_readlane(incl, 63);
			cols = ncand <= COLCAP;
			if (cols) {
				// candidate -> (column, row) without a per-candidate table: one bit per column marks its first candidate,
				// the columns that have candidates are listed compactly; candidate e of a 64-candidate step then belongs to
				// the (columns started before the step + marks at or below e)-th listed column
				const int start = incl - len;
				const unsigned long long nonempty = __ballot(len > 0);
				// (the zero is made here: hoisted out of the keypoint loop as a register pair it was the one value the allocator
				// spilled, and every keypoint then waited for a scratch load to clear sixteen words of LDS)
				unsigned long long zero64 = 0ULL;
				asm volatile("" : "+v"(zero64));
				if (lane < COLCAP / 64) S.startbits[lane] = zero64;
				WAVE_FENCE();
				if (len > 0) {
					atomicOr(&S.startbits[start >> 6], 1ULL << (start & 63));
					S.colpk[rank_below(nonempty)] = ((unsigned)start << 16) | (((unsigned)lo & 0xFFu) << 8) | (unsigned)lane;
				}
			} else ncand = nsamp;
			WAVE_FENCE();
		}

		// the reference's window tests on one sample (sift.cc:113-126); x_rot, y_rot as the reference rounds them
		auto rotate = [&](int xx, int yy, float& x_rot, float& y_rot) {
			const float fxx = (float)xx, fyy = (float)yy;
			y_rot = (float)((double)((float)(-xx) * sinort + fyy * cosort) * rd);
			x_rot = (float)((double)(fxx * cosort + fyy * sinort) * rd);
		};

		// phases 1b - 3 on one batch of up to 64 window samples in window order (lanes with cand == false carry none)
		auto process_batch = [&](bool cand, int xx, int yy) {
			// the reference's window tests (sift.cc:113-126): inside the circle, both rotated coordinates inside the
			// 4 x 4 bin square;  between(bin, -1, 4) on floats is  -1 <= bin <= 3  (lib/utils.hh:27)
			float x_rot, y_rot;
			rotate(xx, yy, x_rot, y_rot);
			const float ybin = (y_rot + 2.f) - 0.5f, xbin = (x_rot + 2.f) - 0.5f;
			const bool ok = cand && !((float)xx * (float)xx + (float)yy * (float)yy > fr2)
				&& ybin >= -1.f && ybin <= 3.f && xbin >= -1.f && xbin <= 3.f;
			int a0 = 0;                                        // byte offset of mask[(ybinf * 4 + xbinf) * 8 + hbinf % 8]
			int a1 = 0;                                        // the same for (hbinf + 1) % 8
			float vA[4], vB[4];                                // per touched cell: the values for orientation bins h0 and h0 + 1
			bool vy0 = false, vy1 = false, vx0 = false, vx1 = false;      // between(ybinf + dy, 0, DESC_HIST_WIDTH), between(xbinf + dx, ...)
#pragma unroll
			for (int u = 0; u < 4; ++u) { vA[u] = 0.f; vB[u] = 0.f; }
			if (ok) {
				const int gi = (kpy + yy) * w + (kpx + xx);
				const float gdy = g_img[gi + w] - g_img[gi - w];
				const float gdx = g_img[gi + 1] - g_img[gi - 1];
				const float now_mag = opdev::hypotf_glibc(gdx, gdy);
				float now_ort = opdev::fast_atan_plus_pi(gdy, gdx);
				float weight = opdev::expf_glibc(-(x_rot * x_rot + y_rot * y_rot) / exp_denom, S.exptab);
				weight = weight * now_mag;
				now_ort -= ort;
				if (now_ort < 0) now_ort += pi2;
				if (now_ort > pi2) now_ort -= pi2;
				const float hbin = now_ort * nbin_per_rad;
				// trilinear_interpolate (sift.cc:48-67); ybinf, xbinf are in [-1, 3] (phase 1a's window test)
				const float yf = floorf(ybin), xf = floorf(xbin), hf = floorf(hbin);
				const float ybind = ybin - yf, xbind = xbin - xf, hbind = hbin - hf;
				const float omh = 1 - hbind;
				const int yb = (int)yf, xb = (int)xf;
				vy0 = yb >= 0; vy1 = yb <= 2; vx0 = xb >= 0; vx1 = xb <= 2;
				const int hq8 = ((int)hf & 7) * 8;                    // hbinf % 8; the second bin is (hbinf + 1) % 8
				a0 = (yb * 4 + xb) * 64 + hq8;
				a1 = (yb * 4 + xb) * 64 + ((hq8 + 8) & 56);
#pragma unroll
				for (int dy = 0; dy < 2; ++dy) {
					const float w_y = weight * (dy ? ybind : 1 - ybind);
#pragma unroll
					for (int dx = 0; dx < 2; ++dx) {
						const float w_x = w_y * (dx ? xbind : 1 - xbind);
						vA[dy * 2 + dx] = w_x * omh; vB[dy * 2 + dx] = w_x * hbind;
					}
				}
			}
			const bool valid[4] = {vy0 && vx0, vy0 && vx1, vy1 && vx0, vy1 && vx1};
			// phase 2: stable counting sort of the contributions by bin.  A sample touches, per cell, the two
			// adjacent orientation bins h0 and h0 + 1: ONE mask bit per (cell, h0) records both (4 LDS atomics
			// per sample instead of 8), the contributors of bin (cell, k) are mask[cell][k] (their first value)
			// and mask[cell][k - 1] (their second) -- disjoint sets, merged in lane = sample order.
			// The sample's four cells are at +0, +1, +4, +5 cells from its first one: immediate offsets of the ds
			// instructions (cells outside the histogram, sift.cc:59,61, are masked out, their addresses never used)
			char* const lds = (char*)&S;
			constexpr int MASK0 = (int)offsetof(DescLds, mask), OFF0 = (int)offsetof(DescLds, off);
			constexpr int CELL_OFF[4] = {0, 1, 4, 5};
#pragma unroll
			for (int u = 0; u < 4; ++u)
				if (valid[u]) atomicOr((unsigned long long*)(lds + MASK0 + a0 + CELL_OFF[u] * 64), 1ULL << lane);
			WAVE_FENCE();
			f32x4* const sorted4 = (f32x4*)S.sorted;
			unsigned* const rec = (unsigned*)S.mask;
			// contributors of this lane's two bins (histogram bins lane and lane + 64), their list lengths in whole float4s
			const unsigned long long c0 = S.mask[lane] | S.mask[ownp], c1 = S.mask[lane + 64] | S.mask[ownp + 64];
			unsigned long long u0 = c0, u1 = c1;
			int n0 = (__popcll(u0) + 3) >> 2, n1 = (__popcll(u1) + 3) >> 2;
			int incl = wave_scan_add(n0 + n1);                                     // inclusive wave scan of the per-lane pair sizes
			// one pass over all 64 samples when their padded lists fit the arena, else lanes 0..31 and lanes 32..63 in turn
			const int npass = __builtin_amdgcn_readlane(incl, 63) <= list_cap4 ? 1 : 2;
			for (int pass = 0; pass < npass; ++pass) {
				unsigned long long lm = ~0ULL;
				if (npass == 2) {
					lm = pass == 0 ? 0xFFFFFFFFULL : 0xFFFFFFFF00000000ULL;
					u0 = c0 & lm; u1 = c1 & lm;
					n0 = (__popcll(u0) + 3) >> 2; n1 = (__popcll(u1) + 3) >> 2;
					incl = wave_scan_add(n0 + n1);
				}
				const int q0 = incl - (n0 + n1), q1 = q0 + n0;                                    // first float4 of this lane's two lists
				S.off[lane] = (unsigned short)(q0 * 16); S.off[lane + 64] = (unsigned short)(q1 * 16);
				// zero the padding slots (at most 3 per list, all inside the list's last float4): +0.0f leaves an fp32 sum of
				// non-negative terms unchanged.  The whole last float4 is cleared with one 16-byte write (an empty list clears
				// the float4 of zeros); the scatter below overwrites the slots that hold values.
				sorted4[n0 ? q0 + n0 - 1 : ZERO4 / 4] = f32x4{0.f, 0.f, 0.f, 0.f};
				sorted4[n1 ? q1 + n1 - 1 : ZERO4 / 4] = f32x4{0.f, 0.f, 0.f, 0.f};
				// from here on only the contributor sets are needed: they replace the raw masks in place (every lane's reads
				// above precede every lane's writes below: one wavefront, LDS accesses in program order)
				WAVE_FENCE();
				S.mask[lane] = u0; S.mask[lane + 64] = u1;
				WAVE_FENCE();
				if ((lm >> lane) & 1ULL) {
					const int o0 = a0 >> 2, o1 = a1 >> 2;                      // the same bins in S.off (2 bytes per bin)
#pragma unroll
					for (int u = 0; u < 4; ++u)
						if (valid[u]) {
							const unsigned long long ma = *(const unsigned long long*)(lds + MASK0 + a0 + CELL_OFF[u] * 64);
							const unsigned long long mb = *(const unsigned long long*)(lds + MASK0 + a1 + CELL_OFF[u] * 64);
							const int sa = *(const unsigned short*)(lds + OFF0 + o0 + CELL_OFF[u] * 16);
							const int sb = *(const unsigned short*)(lds + OFF0 + o1 + CELL_OFF[u] * 16);
							*(float*)(lds + sa + rank_below(ma) * 4) = vA[u];
							*(float*)(lds + sb + rank_below(mb) * 4) = vB[u];
						}
				}
				WAVE_FENCE();
				// phase 3: ordered accumulation.  A batch fills a third of the 128 bins, and very unevenly (the longest list
				// is 2 - 3 times the mean), so the non-empty lists are dealt one per lane: every owner leaves a record
				// (bin, first float4, float4s) per non-empty list in a compact table (it takes the place of the masks, which
				// are dead now), lane r takes record r, continues the bin's running sum from the histogram in LDS through
				// the list, IN ORDER, two float4s per step (a lane whose list has ended adds the float4 of zeros: x + 0 = x
				// for the non-negative sums), and puts the sum back.  The steps end when no lane has list left.
				{
					const unsigned long long b0 = __ballot(n0 > 0), b1 = __ballot(n1 > 0);
					const int nb0 = __popcll(b0), nrec = nb0 + __popcll(b1);
					if (n0) rec[rank_below(b0)] = (unsigned)lane | ((unsigned)q0 << 8) | ((unsigned)n0 << 16);
					if (n1) rec[nb0 + rank_below(b1)] = (unsigned)(lane + 64) | ((unsigned)q1 << 8) | ((unsigned)n1 << 16);
					WAVE_FENCE();
					for (int r0 = 0; r0 < nrec; r0 += 64) {
						const unsigned my = r0 + lane < nrec ? rec[r0 + lane] : 0u;
						const int bin = (int)(my & 0xFFu), q = (int)((my >> 8) & 0xFFu), nn = (int)(my >> 16);
						float hsum = S.hist[bin];
						for (int e = 0; __ballot(e < nn) != 0ULL; e += 2) {
							const f32x4 a = sorted4[e < nn ? q + e : ZERO4 / 4], c = sorted4[e + 1 < nn ? q + e + 1 : ZERO4 / 4];
							hsum += a.x; hsum += a.y; hsum += a.z; hsum += a.w; hsum += c.x; hsum += c.y; hsum += c.z; hsum += c.w;
						}
						if (nn) S.hist[bin] = hsum;
					}
				}
				WAVE_FENCE();
			}
			S.mask[lane] = 0ULL; S.mask[lane + 64] = 0ULL;
			WAVE_FENCE();
		};

		// phase 1a: the window in the reference's order (xx outer, yy inner: sift.cc:110-113).  With column intervals
		// nearly every candidate passes the window tests (1.02 candidates per kept sample), so 64 consecutive candidates
		// ARE a batch.  The full-window walk keeps a third of its samples: there the cheap tests run first and the
		// survivors are queued in order and handed on 64 at a time, so the expensive phases run with full wavefronts.
		int qx = lane / side, qy = lane % side;          // full-window walk: sample e = i0 + lane  ->  (e / side, e % side)
		int kbase = -1;                                  // listed columns started before this step, minus one
		for (int i0 = 0; i0 < ncand || qn > 0; i0 += 64) {
			bool cand = i0 + lane < ncand, run = true;
			int xx = 0, yy = 0;
			if (cols) {
				const unsigned long long sb = S.startbits[i0 >> 6];
				const int ord = kbase + rank_below(sb) + (int)((unsigned)(sb >> lane) & 1u);
				kbase += __popcll(sb);
				if (cand) {
					const unsigned pk = S.colpk[ord];
					xx = (int)(pk & 0xFFu) - radius; yy = (int)(signed char)(pk >> 8) + (i0 + lane - (int)(pk >> 16));
				}
			} else {
				bool ok = false;
				xx = qx - radius; yy = qy - radius;
				qy += 64; while (qy >= side) { qy -= side; ++qx; }
				const int nowx = kpx + xx, nowy = kpy + yy;
				if (cand && nowx >= 1 && nowx <= w - 2 && nowy >= 1 && nowy <= h - 2) {
					const float fxx = (float)xx, fyy = (float)yy;
					if (!(fxx * fxx + fyy * fyy > fr2)) {
						float x_rot, y_rot;
						rotate(xx, yy, x_rot, y_rot);
						const float ybin = (y_rot + 2.f) - 0.5f, xbin = (x_rot + 2.f) - 0.5f;
						ok = (ybin >= -1.f && ybin <= 3.f && xbin >= -1.f && xbin <= 3.f);
					}
				}
				const unsigned long long mask = __ballot(ok);
				if (ok) S.q[(qhead + qn + rank_below(mask)) & (QCAP - 1)] = (unsigned)(xx + 32768) | ((unsigned)(yy + 32768) << 16);   // 16 + 16 bits: any window radius
				qn += __popcll(mask);
				run = qn >= 64 || (i0 + 64 >= ncand && qn > 0);
				if (run) {
					WAVE_FENCE();
					const int n = qn < 64 ? qn : 64;
					const unsigned pk = S.q[(qhead + lane) & (QCAP - 1)];
					cand = lane < n;
					xx = (int)(pk & 0xFFFFu) - 32768; yy = (int)(pk >> 16) - 32768;
					qhead = (qhead + n) & (QCAP - 1); qn -= n;
					WAVE_FENCE();
				}
			}
			if (run) process_batch(cand, xx, yy);
		}

		// hist_to_descriptor (:15-46): L1-normalise (sequential fp32 sum), sqrt, * DESC_INT_FACTOR
		float* hist = S.hist;
		WAVE_FENCE();
		float sum = 0.f;            // the reference's sequential fp32 sum (sift.cc:39-40), 16 bytes per LDS read
#pragma unroll 4
		for (int i = 0; i < 32; ++i) { const f32x4 q = ((const f32x4*)hist)[i]; sum += q.x; sum += q.y; sum += q.z; sum += q.w; }
		float* out = desc + kk * 128;
#pragma unroll
		for (int t = 0; t < 2; ++t) {
			const int i = lane + 64 * t;
			const float v = hist[i] / sum;
			out[i] = sqrtf(v) * (float)p.desc_int_factor;
		}
		if (lane == 0) {   // feature/feature.cc:23-26
			// (the two conversions are made here, per keypoint: as loop invariants they were register pairs held -- and spilled -- across the loop)
			int sw = p.sw, sh = p.sh;
			asm volatile("" : "+s"(sw), "+s"(sh));
			coor[kk * 2] = (kp.rx - 0.5) * (double)sw;
			coor[kk * 2 + 1] = (kp.ry - 0.5) * (double)sh;
			real[kk * 2] = kp.rx; real[kk * 2 + 1] = kp.ry;      // do_detect_feature's own [0,1) output
		}
		WAVE_FENCE();
	}
}

}	// namespace

hipError_t launch_descriptor(const SiftPlan& p, const KeyPoint* oriented, const long long* total,
		long long cap, float* desc, double* coor, double* real, hipStream_t st) {
	if (cap <= 0) return hipSuccess;
	const long long most = (long long)(p.num_cu > 0 ? p.num_cu : 256) * 4 * DESC_WAVES * DESC_GRID_MULT;
	const int grid = (int)(cap < most ? cap : most);
	hipLaunchKernelGGL(k_descriptor, dim3(grid), dim3(64), 0, st, p, oriented, total, cap, desc, coor, real);
	return hipGetLastError();
}
