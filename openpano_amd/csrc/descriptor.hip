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
//       offsets are a 128-entry scan of the mask popcounts; the values land bin-major in LDS,
//       in sample order inside each bin;  (a workgroup is ONE wavefront: its LDS accesses execute
//       in program order, and a barrier is a compiler fence plus a wait, not a rendezvous)
//   3   lane L owns bins L and L+64 and adds their segments in order to its two fp32
//       accumulators -- every bin sees exactly the reference's sequence of additions.
// The window is not scanned whole: per window column only the rows that can pass the circle and
// rotated-square tests are enumerated (a conservative interval), the exact tests decide.
#include "internal.hpp"
#include "devmath.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef DESC_GRID_MULT
#define DESC_GRID_MULT 8
#endif
constexpr int QCAP = 128;            // survivor queue (power of two, >= 2 x 64)
constexpr int COLCAP = 1024;         // candidate samples of one keypoint in the column-interval enumeration (the shipped config needs < 800)

struct DescLds {
	unsigned long long mask[128];    // per bin: bit l = lane l's sample of the current batch contributes
	float sorted[512 + 3 * 128 + 64] __attribute__((aligned(16)));   // the batch's contributions, bin-major, sample order inside a bin; every list starts on a 16-byte boundary and is zero-padded to a multiple of 4
	unsigned short off[128];         // first slot of every bin in sorted[]
	int q_gi[QCAP];                  // survivor queue (ring): plane offset, rotated coordinates
	float q_xr[QCAP], q_yr[QCAP];
	uint64_t exptab[32];             // glibc's exp2f table (devmath.hpp), staged once per workgroup
	unsigned long long startbits[COLCAP / 64];   // bit e: candidate e is the first of its window column
	unsigned colpk[64];              // k-th non-empty window column: index of its first candidate << 16 | (first candidate row & 0xFF) << 8 | column
};

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

// Launch shape: up to DESC_GRID_MULT times the wavefronts the device holds at this kernel's occupancy (20 per CU), each
// taking every gridDim-th keypoint; no workgroups far past the device-side count.  Measured on 46 K keypoints: exactly
// the resident 5120 wavefronts 0.338 ms (no slack to balance the keypoints' very different windows), 8 x / 16 x / 32 x
// 0.297 ms; keypoints drawn from ONE atomic ticket counter 0.62 ms (46 K atomics on one address outlast the kernel).
__global__ void __launch_bounds__(64) k_descriptor(SiftPlan p, const KeyPoint* oriented,
		const long long* total_ptr, long long cap, float* desc, double* coor, double* real) {
	__shared__ DescLds S;
	const int lane = threadIdx.x;
	const float pi2 = (float)(2 * 3.14159265358979323846);
	const float nbin_per_rad = 8 / pi2;
	S.mask[2 * lane] = 0ULL; S.mask[2 * lane + 1] = 0ULL;
	if (lane < 32) S.exptab[lane] = opdev::kExp2fTab[lane];
	__syncthreads();

	long long total = *total_ptr;                      // device-side count (k_expand_oriented)
	total = total < cap ? total : cap;                 // speculative capacity: the host re-runs on overflow
	for (long long kk = blockIdx.x; kk < total; kk += gridDim.x) {
		const KeyPoint kp = oriented[kk];
		const int img = kp.pad;                            // written by k_expand_oriented
		const OctDesc od = p.oct[kp.oct];
		const int w = od.w, h = od.h;
		const float* base = p.ws + (long long)img * p.ws_stride;
		// mag / ort of GaussianPyramid::cal_mag_ort (feature/dog.cc:76-84) are evaluated on the
		// Gaussian plane for the surviving window samples only
		const float* g_img = base + plane_off_gauss(od, p.nscale, kp.scale);
		const float ort = kp.dir;
		const float hist_w = kp.sf * (float)p.desc_scale_factor;
		// x / hist_w for the two rotated coordinates of every window sample: (float)((double)x * rd) with
		// rd = 1 / (double)hist_w is the correctly rounded fp32 quotient (the double product is within 2^-52
		// of x / hist_w, and a quotient of two fp32 numbers is never closer than 2^-49 (relative) to a
		// rounding boundary of fp32), at a third of the instructions of an IEEE fp32 division
		const double rd = 1.0 / (double)hist_w;
		const float exp_denom = 2 * (4.f * 4.f);
		const int radius = (int)round(0.70710678118654752440 * (double)hist_w * (4 + 1));
		const float cosort = opdev::cosf_glibc(ort), sinort = opdev::sinf_glibc(ort);
		const int side = 2 * radius + 1, nsamp = side * side;
		const float fr2 = (float)radius * (float)radius;
		float acc0 = 0.f, acc1 = 0.f;     // bins lane and lane + 64
		int qhead = 0, qn = 0;            // survivor queue state (wave-uniform)

		// ---- candidate enumeration.  The reference visits the whole (2 radius + 1)^2 window in (xx outer,
		// yy inner) order and keeps a sample only inside the circle and the rotated 4 x 4 bin square
		// (sift.cc:110-126) -- about a third of the window.  Each window column's rows that CAN pass are an
		// interval, computed here with a safety margin of a row on either side; the exact float tests of
		// the reference then run on those candidates only, in the reference's order.  Windows wider than 64
		// columns or with more than COLCAP candidates (other DESC_HIST_SCALE_FACTORs) walk the full window.
		int ncand = nsamp;
		bool cols = side <= 64;
		if (cols) {
			int lo = 0, len = 0;
			if (lane < side) {
				const int xx = lane - radius;
				const float fxx = (float)xx;
				const int nowx = kp.x + xx;
				float ylo = -(float)radius, yhi = (float)radius;
				const float rem = fr2 - fxx * fxx;
				const float yc = rem > 0.f ? sqrtf(rem) + 1.f : 1.f;
				ylo = fmaxf(ylo, -yc); yhi = fminf(yhi, yc);
				// -2.5 <= rot / hist_w <= 1.5 for both rotated coordinates (bin in [-1, 3] after the +1.5 shift)
				const float m = 0.02f * hist_w + 0.25f;
				const float blo = -2.5f * hist_w - m, bhi = 1.5f * hist_w + m;
				if (fabsf(cosort) > 0.05f) {           // y_rot * hist_w = -xx sin + yy cos
					const float a = (blo + fxx * sinort) / cosort, b = (bhi + fxx * sinort) / cosort;
					ylo = fmaxf(ylo, fminf(a, b)); yhi = fminf(yhi, fmaxf(a, b));
				}
				if (fabsf(sinort) > 0.05f) {           // x_rot * hist_w = xx cos + yy sin
					const float a = (blo - fxx * cosort) / sinort, b = (bhi - fxx * cosort) / sinort;
					ylo = fmaxf(ylo, fminf(a, b)); yhi = fminf(yhi, fmaxf(a, b));
				}
				int ilo = (int)floorf(ylo) - 1, ihi = (int)ceilf(yhi) + 1;
				ilo = ilo < -radius ? -radius : ilo; ihi = ihi > radius ? radius : ihi;
				ilo = ilo < 1 - kp.y ? 1 - kp.y : ilo; ihi = ihi > h - 2 - kp.y ? h - 2 - kp.y : ihi;     // between(nowy, 1, h - 1)
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
				if (lane < COLCAP / 64) S.startbits[lane] = 0ULL;
				__syncthreads();
				if (len > 0) {
					atomicOr(&S.startbits[start >> 6], 1ULL << (start & 63));
					S.colpk[rank_below(nonempty)] = ((unsigned)start << 16) | (((unsigned)lo & 0xFFu) << 8) | (unsigned)lane;
				}
			} else ncand = nsamp;
			__syncthreads();
		}

		// phases 1b - 3 on one dense batch of queued survivors (window order preserved)
		auto process_batch = [&](int n) {
			const bool ok = lane < n;
			const int qi = (qhead + lane) & (QCAP - 1);
			int cb[4]; float vA[4], vB[4]; int hq = 0;       // per touched cell: first bin of the cell, values for bins h0 and h0 + 1
#pragma unroll
			for (int u = 0; u < 4; ++u) { cb[u] = -1; vA[u] = 0.f; vB[u] = 0.f; }
			if (ok) {
				const int gi = S.q_gi[qi];
				const float x_rot = S.q_xr[qi], y_rot = S.q_yr[qi];
				const float ybin = (y_rot + 2.f) - 0.5f, xbin = (x_rot + 2.f) - 0.5f;
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
				// trilinear_interpolate (sift.cc:48-67)
				const float yf = floorf(ybin), xf = floorf(xbin), hf = floorf(hbin);
				const int yb = (int)yf, xb = (int)xf, h0 = (int)hf;
				const float ybind = ybin - yf, xbind = xbin - xf, hbind = hbin - hf;
				const float omh = 1 - hbind;
				hq = h0 & 7;                                           // hbinf % 8; the second bin is (hbinf + 1) % 8
#pragma unroll
				for (int dy = 0; dy < 2; ++dy) {
					const float w_y = weight * (dy ? ybind : 1 - ybind);
#pragma unroll
					for (int dx = 0; dx < 2; ++dx) {
						const int cy = yb + dy, cx = xb + dx;
						if ((unsigned)cy < 4u && (unsigned)cx < 4u) {        // between(., 0, DESC_HIST_WIDTH)
							const float w_x = w_y * (dx ? xbind : 1 - xbind);
							const int u = dy * 2 + dx;
							cb[u] = (cy * 4 + cx) * 8; vA[u] = w_x * omh; vB[u] = w_x * hbind;
						}
					}
				}
			}
			// phase 2: stable counting sort of the contributions by bin.  A sample touches, per cell, the two
			// adjacent orientation bins h0 and h0 + 1: ONE mask bit per (cell, h0) records both (4 LDS atomics
			// per sample instead of 8), the contributors of bin (cell, k) are mask[cell][k] (their first value)
			// and mask[cell][k - 1] (their second) -- disjoint sets, merged in lane = sample order.
#pragma unroll
			for (int u = 0; u < 4; ++u)
				if (cb[u] >= 0) atomicOr(&S.mask[cb[u] + hq], 1ULL << lane);
			__syncthreads();
			{
				const int base = (lane >> 2) * 8, k0 = 2 * (lane & 3);          // this lane computes the offsets of bins 2 lane, 2 lane + 1
				const unsigned long long mp = S.mask[base + ((k0 + 7) & 7)], m0 = S.mask[base + k0], m1 = S.mask[base + k0 + 1];
				const unsigned long long u0 = m0 | mp, u1 = m1 | m0;       // contributors of bins 2 lane and 2 lane + 1
				const int c0 = __popcll(u0), c1 = __popcll(u1);
				const int p0 = (c0 + 3) & ~3, p1 = (c1 + 3) & ~3;         // list lengths rounded up to whole float4s
				const int incl = wave_scan_add(p0 + p1);                   // inclusive wave scan of the per-lane pair sizes
				const int ex = incl - (p0 + p1);
				S.off[2 * lane] = (unsigned short)ex; S.off[2 * lane + 1] = (unsigned short)(ex + p0);
				// zero the padding slots (at most 3 per list, all inside the list's last float4): +0.0f leaves an fp32 sum of
				// non-negative terms unchanged.  The whole last float4 is cleared with one 16-byte write; the scatter below
				// (after the barrier) overwrites the slots that hold values.
				if (p0) *(f32x4*)&S.sorted[ex + p0 - 4] = f32x4{0.f, 0.f, 0.f, 0.f};
				if (p1) *(f32x4*)&S.sorted[ex + p0 + p1 - 4] = f32x4{0.f, 0.f, 0.f, 0.f};
				// from here on only the contributor sets are needed: they replace the raw masks in place (every lane's three
				// reads above precede every lane's two writes below: one wavefront, LDS accesses in program order)
				__syncthreads();
				S.mask[base + k0] = u0; S.mask[base + k0 + 1] = u1;
			}
			__syncthreads();
#pragma unroll
			for (int u = 0; u < 4; ++u)
				if (cb[u] >= 0) {
					const int hn = (hq + 1) & 7;
					S.sorted[S.off[cb[u] + hq] + rank_below(S.mask[cb[u] + hq])] = vA[u];
					S.sorted[S.off[cb[u] + hn] + rank_below(S.mask[cb[u] + hn])] = vB[u];
				}
			__syncthreads();
			// phase 3: ordered accumulation.  Lane L owns bins L and L + 64 (cells 8 apart: when one is
			// crowded the other is not, which evens the list lengths across the wave).  Lists are read a
			// float4 at a time (16-byte aligned, zero-padded), the four additions of a group stay in order;
			// the trip count is wave-uniform, lanes whose list has ended skip the group.
			{
				const int na = (__popcll(S.mask[lane]) + 3) & ~3, nb = (__popcll(S.mask[lane + 64]) + 3) & ~3;
				const f32x4* la = (const f32x4*)&S.sorted[S.off[lane]];
				const f32x4* lb = (const f32x4*)&S.sorted[S.off[lane + 64]];
				const int T = wave_max_i(na > nb ? na : nb);
				for (int e = 0; e < T; e += 8) {
					f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
					if (e < na) a0 = la[e >> 2];
					if (e + 4 < na) a1 = la[(e >> 2) + 1];
					if (e < nb) b0 = lb[e >> 2];
					if (e + 4 < nb) b1 = lb[(e >> 2) + 1];
					acc0 += a0.x; acc0 += a0.y; acc0 += a0.z; acc0 += a0.w; acc0 += a1.x; acc0 += a1.y; acc0 += a1.z; acc0 += a1.w;
					acc1 += b0.x; acc1 += b0.y; acc1 += b0.z; acc1 += b0.w; acc1 += b1.x; acc1 += b1.y; acc1 += b1.z; acc1 += b1.w;
				}
			}
			__syncthreads();
			S.mask[2 * lane] = 0ULL; S.mask[2 * lane + 1] = 0ULL;
			__syncthreads();
			qhead = (qhead + n) & (QCAP - 1); qn -= n;
		};

		// phase 1a: candidates in the reference's order (xx outer, yy inner: sift.cc:110-113); the
		// cheap tests run on all candidates, survivors are queued in order and handed to the expensive
		// phases 64 at a time, so those always run with full wavefronts
		int qx = lane / side, qy = lane % side;          // full-window walk: sample e = i0 + lane  ->  (e / side, e % side)
		int kbase = -1;                                  // listed columns started before this step, minus one
		for (int i0 = 0; i0 < ncand; i0 += 64) {
			bool ok = false;
			float x_rot = 0.f, y_rot = 0.f;
			int gi = 0;
			int ord = 0;
			if (cols) {
				const unsigned long long sb = S.startbits[i0 >> 6];
				ord = kbase + rank_below(sb) + (int)((unsigned)(sb >> lane) & 1u);
				kbase += __popcll(sb);
			}
			if (i0 + lane < ncand) {
				int xx, yy;
				if (cols) {
					const unsigned pk = S.colpk[ord];
					xx = (int)(pk & 0xFFu) - radius; yy = (int)(signed char)(pk >> 8) + (i0 + lane - (int)(pk >> 16));
				} else { xx = qx - radius; yy = qy - radius; }
				const int nowx = kp.x + xx, nowy = kp.y + yy;
				if (nowx >= 1 && nowx <= w - 2 && nowy >= 1 && nowy <= h - 2) {
					const float fxx = (float)xx, fyy = (float)yy;
					if (!(fxx * fxx + fyy * fyy > fr2)) {
						y_rot = (float)((double)((float)(-xx) * sinort + fyy * cosort) * rd);
						x_rot = (float)((double)(fxx * cosort + fyy * sinort) * rd);
						const float ybin = (y_rot + 2.f) - 0.5f, xbin = (x_rot + 2.f) - 0.5f;
						// between(bin, -1, 4) on floats is  -1 <= bin <= 3  (lib/utils.hh:27)
						ok = (ybin >= -1.f && ybin <= 3.f && xbin >= -1.f && xbin <= 3.f);
						gi = nowy * w + nowx;
					}
				}
			}
			if (!cols) { qy += 64; while (qy >= side) { qy -= side; ++qx; } }
			const unsigned long long mask = __ballot(ok);
			if (ok) {
				const int qi = (qhead + qn + rank_below(mask)) & (QCAP - 1);
				S.q_gi[qi] = gi; S.q_xr[qi] = x_rot; S.q_yr[qi] = y_rot;
			}
			qn += __popcll(mask);
			if (qn >= 64) { __syncthreads(); process_batch(64); }
		}
		if (qn > 0) { __syncthreads(); process_batch(qn); }

		// hist_to_descriptor (:15-46): L1-normalise (sequential fp32 sum), sqrt, * DESC_INT_FACTOR
		// (the histogram reuses the list buffer: 8 KB of LDS per workgroup = 20 workgroups, 5 waves per SIMD)
		float* hist = S.sorted;
		hist[lane] = acc0;
		hist[lane + 64] = acc1;
		__syncthreads();
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
			coor[kk * 2] = (kp.rx - 0.5) * (double)p.sw;
			coor[kk * 2 + 1] = (kp.ry - 0.5) * (double)p.sh;
			real[kk * 2] = kp.rx; real[kk * 2 + 1] = kp.ry;      // do_detect_feature's own [0,1) output
		}
		__syncthreads();
	}
}

}	// namespace

hipError_t launch_descriptor(const SiftPlan& p, const KeyPoint* oriented, const long long* total,
		long long cap, float* desc, double* coor, double* real, hipStream_t st) {
	if (cap <= 0) return hipSuccess;
	const long long most = (long long)(p.num_cu > 0 ? p.num_cu : 256) * 20 * DESC_GRID_MULT;      // 20 = 5 wavefronts per SIMD (90 VGPRs, 7.3 KB of LDS)
	const int grid = (int)(cap < most ? cap : most);
	hipLaunchKernelGGL(k_descriptor, dim3(grid), dim3(64), 0, st, p, oriented, total, cap, desc, coor, real);
	return hipGetLastError();
}
