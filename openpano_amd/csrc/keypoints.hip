// keypoints.hip -- extrema scan, sub-pixel refinement, canonical ordering, orientation.
//
// Replaces ExtremaDetector::get_extrema and helpers (feature/extrema.cc:36-216) and
// OrientationAssign (feature/orientation.cc:22-100) for a whole batch per launch.
// Built with -ffp-contract=off; fp types and evaluation order follow the reference line by line
// so that every accept/reject decision and every emitted number is identical to the CPU path.
#include "internal.hpp"
#include "devmath.hpp"

namespace {

// DoG layer l at pixel c (feature/dog.cc:126): the planes are not materialised, the value is the same fp32
// |G[l] - G[l+1]| evaluated on the Gaussian stack (G[0] = grey), whose planes lie `plane` floats apart
struct DogView {
	const float* g0; long long plane;
	__device__ __forceinline__ float operator()(int l, long long c) const {
		const float* a = g0 + (long long)l * plane + c;
		return fabsf(a[0] - a[plane]);
	}
};

// Eigen::FullPivLU 3x3 inverse as used by Matrix::inverse (lib/matrix.cc:76-87): complete
// pivoting, rank threshold |pivot| > |maxpivot| * eps * 3, inverse = solve(Identity).
// Written without a single run-time array index: every loop is unrolled and a data-dependent
// transposition is a chain of conditional swaps against the constant candidates, so the nine
// entries, the right-hand side and the two permutation records stay in registers (indexed through
// run-time pivots they went to scratch memory, a memory round trip per swapped element on the
// critical path of a kernel that has two waves per SIMD to hide it with).
__device__ __forceinline__ void cswap(bool c, double& a, double& b) { const double t = a; a = c ? b : a; b = c ? t : b; }
__device__ __forceinline__ bool inverse3_fullpiv(const double a[9], double inv[9]) {
	double lu[9];
#pragma unroll
	for (int i = 0; i < 9; ++i) lu[i] = a[i];
	int rowt[3] = {0, 1, 2}, colt[3] = {0, 1, 2}, nonzero = 3;
	double maxpivot = 0;
	bool live = true;                     // false once a zero pivot block ended the elimination (the reference's break)
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		int br = k, bc = k; double best = -1;
#pragma unroll
		for (int i = k; i < 3; ++i)
#pragma unroll
			for (int j = k; j < 3; ++j) {
				const double v = fabs(lu[i * 3 + j]);
				if (v > best) { best = v; br = i; bc = j; }
			}
		if (live && best == 0.0) { nonzero = k; live = false; }       // rowt / colt of the remaining steps stay the identity
		if (live) {
			if (best > maxpivot) maxpivot = best;
			rowt[k] = br; colt[k] = bc;
#pragma unroll
			for (int r = k + 1; r < 3; ++r)
#pragma unroll
				for (int j = 0; j < 3; ++j) cswap(br == r, lu[k * 3 + j], lu[r * 3 + j]);
#pragma unroll
			for (int c = k + 1; c < 3; ++c)
#pragma unroll
				for (int i = 0; i < 3; ++i) cswap(bc == c, lu[i * 3 + k], lu[i * 3 + c]);
#pragma unroll
			for (int i = k + 1; i < 3; ++i) lu[i * 3 + k] /= lu[k * 3 + k];
#pragma unroll
			for (int i = k + 1; i < 3; ++i)
#pragma unroll
				for (int j = k + 1; j < 3; ++j)
					lu[i * 3 + j] -= lu[i * 3 + k] * lu[k * 3 + j];
		}
	}
	const double thr = fabs(maxpivot) * (2.220446049250313e-16 * 3);
	int rank = 0;
#pragma unroll
	for (int i = 0; i < 3; ++i) rank += (i < nonzero && fabs(lu[i * 3 + i]) > thr);
	if (rank != 3) return false;
#pragma unroll
	for (int col = 0; col < 3; ++col) {
		double c[3];
#pragma unroll
		for (int i = 0; i < 3; ++i) c[i] = (i == col) ? 1.0 : 0.0;
#pragma unroll
		for (int i = 0; i < 3; ++i)                                      // c[i] <-> c[rowt[i]], rowt[i] >= i
#pragma unroll
			for (int r = i + 1; r < 3; ++r) cswap(rowt[i] == r, c[i], c[r]);
#pragma unroll
		for (int i = 0; i < 3; ++i)
#pragma unroll
			for (int j = 0; j < i; ++j) c[i] -= lu[i * 3 + j] * c[j];
#pragma unroll
		for (int i = 2; i >= 0; --i) {
#pragma unroll
			for (int j = i + 1; j < 3; ++j) c[i] -= lu[i * 3 + j] * c[j];
			c[i] /= lu[i * 3 + i];
		}
#pragma unroll
		for (int i = 2; i >= 0; --i)                                     // c[i] <-> c[colt[i]], colt[i] >= i
#pragma unroll
			for (int r = i + 1; r < 3; ++r) cswap(colt[i] == r, c[i], c[r]);
#pragma unroll
		for (int i = 0; i < 3; ++i) inv[i * 3 + col] = c[i];
	}
	return true;
}

// Matrix::pseudo_inverse (lib/matrix.cc:89-106) of a 3x3 through a one-sided Jacobi SVD; only
// reached for a rank-deficient Hessian.
__device__ void pinv3_jacobi(const double a[9], double out[9]) {
	double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	for (int i = 0; i < 9; ++i) A[i] = a[i];
	for (int sweep = 0; sweep < 60; ++sweep) {
		double off = 0;
		for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
			double alpha = 0, beta = 0, gamma = 0;
			for (int i = 0; i < 3; ++i) { alpha += A[i*3+p]*A[i*3+p]; beta += A[i*3+q]*A[i*3+q]; gamma += A[i*3+p]*A[i*3+q]; }
			if (gamma == 0.0) continue;
			double lim = sqrt(alpha * beta);
			if (fabs(gamma) <= 1e-16 * lim) continue;
			double rel = fabs(gamma) / (lim > 0 ? lim : 1);
			if (rel > off) off = rel;
			double zeta = (beta - alpha) / (2.0 * gamma);
			double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
			double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
			for (int i = 0; i < 3; ++i) {
				double x = A[i*3+p], y = A[i*3+q];
				A[i*3+p] = cs * x - sn * y; A[i*3+q] = sn * x + cs * y;
				x = V[i*3+p]; y = V[i*3+q];
				V[i*3+p] = cs * x - sn * y; V[i*3+q] = sn * x + cs * y;
			}
		}
		if (off < 1e-15) break;
	}
	for (int i = 0; i < 9; ++i) out[i] = 0;
	for (int j = 0; j < 3; ++j) {
		double s = 0;
		for (int i = 0; i < 3; ++i) s += A[i*3+j] * A[i*3+j];
		s = sqrt(s);
		if (!(s > 1e-6)) continue;
		for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
			out[r*3+c] += V[r*3+j] * (1.0 / s) * (A[c*3+j] / s);
	}
}

// ---- calc_kp_offset(_iter) + is_edge_response (feature/extrema.cc:63-168): thread per candidate
__device__ __forceinline__ void refine_one(const SiftPlan& p, const int* raw, int cap, KeyPoint* refined, int* refined_count, int img, int i) {
	const int* q = raw + ((long long)img * cap + i) * 4;
	int nowx = q[0], nowy = q[1];
	const int o = q[2];
	int nows = q[3];
	const OctDesc od = p.oct[o];
	const int w = od.w, h = od.h, nscale = p.nscale;
	const float* base = p.ws + (long long)img * p.ws_stride;
	const DogView dog{base + plane_off_gauss(od, nscale, 0), od.plane};
	double offset[3] = {0, 0, 0}, delta[3] = {0, 0, 0};
	int niter = 0;
	for (; niter < p.calc_offset_depth; ++niter) {
		if (!(nowx >= 1 && nowx <= w - 2) || !(nowy >= 1 && nowy <= h - 2) || !(nows >= 1 && nows <= nscale - 3))
			return;
		const long long c = (long long)nowy * w + nowx;
		auto d0 = [&](long long i) { return dog(nows - 1, i); };
		auto d1 = [&](long long i) { return dog(nows, i); };
		auto d2 = [&](long long i) { return dog(nows + 1, i); };
		const float val = d1(c);
		const float xp = d1(c + 1), xm = d1(c - 1), yp = d1(c + w), ym = d1(c - w);
		const float sp = d2(c), sm = d0(c);
		delta[0] = (double)((xp - xm) / 2);
		delta[1] = (double)((yp - ym) / 2);
		delta[2] = (double)((sp - sm) / 2);
		const double dxx = (double)(xp + xm - val - val);
		const double dyy = (double)(yp + ym - val - val);
		const double dss = (double)(sp + sm - val - val);
		const double dxy = (double)((d1(c + w + 1) - d1(c - w + 1) - d1(c + w - 1) + d1(c - w - 1)) / 4);
		const double dys = (double)((d2(c + w) - d2(c - w) - d0(c + w) + d0(c - w)) / 4);
		const double dsx = (double)((d2(c + 1) - d2(c - 1) - d0(c + 1) + d0(c - 1)) / 4);
		const double m[9] = {dxx, dxy, dsx, dxy, dyy, dys, dsx, dys, dss};
		double inv[9];
		if (!inverse3_fullpiv(m, inv)) pinv3_jacobi(m, inv);
		for (int r = 0; r < 3; ++r) {
			double acc = 0;
			for (int k = 0; k < 3; ++k) acc += inv[r * 3 + k] * delta[k];
			offset[r] = acc;
		}
		const double ax = fabs(offset[0]), ay = fabs(offset[1]), az = fabs(offset[2]);
		double mx = ay > az ? ay : az; mx = ax > mx ? ax : mx;
		if (mx < (double)p.offset_thres) break;
		nowx = (int)((double)nowx + round(offset[0]));
		nowy = (int)((double)nowy + round(offset[1]));
		nows = (int)((double)nows + round(offset[2]));
	}
	if (niter == p.calc_offset_depth) return;
	const long long c = (long long)nowy * w + nowx;
	auto dn = [&](long long i) { return dog(nows, i); };
	double dextr = offset[0] * delta[0] + offset[1] * delta[1] + offset[2] * delta[2];
	dextr = (double)dn(c) + dextr / 2;
	if (dextr < (double)p.contrast_thres) return;
	// is_edge_response (:152-168) on the refined position
	{
		const float val = dn(c);
		const float dxx = dn(c + 1) + dn(c - 1) - val - val;
		const float dyy = dn(c + w) + dn(c - w) - val - val;
		const float dxy = (dn(c + w + 1) + dn(c - w - 1) - dn(c + w - 1) - dn(c - w + 1)) / 4;
		const float det = dxx * dyy - dxy * dxy;
		if (det <= 0) return;
		const float tr = dxx + dyy;
		const float tr2 = tr * tr;
		const float e1 = p.edge_ratio + 1;
		if (!(tr2 / det < (e1 * e1) / p.edge_ratio)) return;
	}
	KeyPoint kp;
	kp.x = nowx; kp.y = nowy; kp.oct = o; kp.scale = nows;
	kp.sf = (float)((double)p.gauss_sigma * pow((double)p.scale_factor, ((double)nows + offset[2]) / nscale));
	kp.rx = ((double)nowx + offset[0]) / w;
	kp.ry = ((double)nowy + offset[1]) / h;
	kp.dir = 0.f; kp.src = i; kp.pad = 0;
	const int slot = atomicAdd(&refined_count[img], 1);
	refined[(long long)img * cap + slot] = kp;
}
__global__ void __launch_bounds__(128) k_refine(SiftPlan p, const int* raw, const int* raw_count, int cap,
		KeyPoint* refined, int* refined_count) {
	const int img = blockIdx.y;
	int n = raw_count[img]; n = n < cap ? n : cap;
	for (int i = blockIdx.x * 128 + threadIdx.x; i < n; i += gridDim.x * 128) refine_one(p, raw, cap, refined, refined_count, img, i);
}

// canonical order of refined keypoints: (oct, scale, y, x, rx, ry), ties by candidate payload
__device__ __forceinline__ bool kp_less(const KeyPoint& a, const KeyPoint& b) {
	if (a.oct != b.oct) return a.oct < b.oct;
	if (a.scale != b.scale) return a.scale < b.scale;
	if (a.y != b.y) return a.y < b.y;
	if (a.x != b.x) return a.x < b.x;
	if (a.rx != b.rx) return a.rx < b.rx;
	if (a.ry != b.ry) return a.ry < b.ry;
	if (a.sf != b.sf) return a.sf < b.sf;
	return false;
}

// Rank sort into the canonical order: every keypoint is ranked against all keypoints of its image, whose packed
// 64-bit primary keys (oct, scale, y, x) are staged through LDS in chunks; the full comparison only runs on
// primary-key ties.  The orientation histograms do not depend on the order, so the ranking is not a kernel of its own:
// it runs on extra wavefronts of k_orientation (below), 8 keypoints x 8 lanes each, beside the histogram wavefronts,
// and leaves the sorted list plus, per sorted position, the keypoint's slot in the unsorted list (where its
// histogram / peaks are); k_expand_oriented reads both.
__device__ __forceinline__ unsigned long long kp_key(const KeyPoint& k) {
	return ((unsigned long long)(unsigned)k.oct << 48) | ((unsigned long long)(unsigned)k.scale << 40) |
		((unsigned long long)(unsigned)(k.y & 0xFFFFF) << 20) | (unsigned long long)(unsigned)(k.x & 0xFFFFF);
}
constexpr int SORT_CHUNK = 256;            // keys staged per round (2 KB of LDS)
constexpr int SORT_SPLIT = 8;              // lanes sharing one keypoint's rank scan
constexpr int SORT_KEYS = 64 / SORT_SPLIT; // keypoints per wavefront
__device__ __forceinline__ void sort_refined_part(const KeyPoint* a, int n, KeyPoint* b, int* slot_of, int kb0, int kstride,
		unsigned long long* s_key) {
	const int lane = threadIdx.x;
	for (int kb = kb0; kb * SORT_KEYS < n; kb += kstride) {          // (uniform over the wavefront)
		const int i = kb * SORT_KEYS + lane / SORT_SPLIT, sub = lane % SORT_SPLIT;
		const bool live = i < n;
		KeyPoint me;
		if (live) me = a[i];
		const unsigned long long mykey = live ? kp_key(me) : 0ULL;
		// branch-free scan: keys below mine, and keys equal to mine (one: myself, unless primary keys tie).  The chunk is
		// padded with the largest key, so every lane runs the same SORT_CHUNK / 8 steps at immediate LDS offsets.
		int rank = 0, same = 0;
		for (int cb = 0; cb < n; cb += SORT_CHUNK) {
			__syncthreads();
#pragma unroll
			for (int q = 0; q < SORT_CHUNK / 64; ++q) {
				const int j = cb + q * 64 + lane;
				s_key[q * 64 + lane] = j < n ? kp_key(a[j]) : ~0ULL;
			}
			__syncthreads();
			const unsigned long long* sk = s_key + sub;            // the 8 lanes of a keypoint read 8 consecutive keys
#pragma unroll 8
			for (int j = 0; j < SORT_CHUNK; j += SORT_SPLIT) {
				const unsigned long long kj = sk[j];
				// rank += kj < mykey; same += kj == mykey: a compare into VCC and an add-with-carry each
				asm("v_cmp_lt_u64 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_cmp_eq_u64 vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
						: "+v"(rank), "+v"(same) : "v"(kj), "v"(mykey) : "vcc");
			}
		}
#pragma unroll
		for (int off = 1; off < SORT_SPLIT; off <<= 1) { rank += __shfl_xor(rank, off); same += __shfl_xor(same, off); }
		if (live && same > 1) {                                   // primary-key ties (rare): the full comparison against the tied records
			int extra = 0;
			for (int j = sub; j < n; j += SORT_SPLIT) {
				if (j == i) continue;
				const KeyPoint o = a[j];
				if (kp_key(o) != mykey) continue;
				// identical records tie-break on the slot index: either assignment is the same output
				extra += (kp_less(o, me) || (!kp_less(me, o) && j < i)) ? 1 : 0;
			}
#pragma unroll
			for (int off = 1; off < SORT_SPLIT; off <<= 1) extra += __shfl_xor(extra, off);
			rank += extra;
		}
		if (live && sub == 0) { me.src = rank; b[rank] = me; slot_of[rank] = i; }
	}
}

// ---- OrientationAssign::calc_dir (feature/orientation.cc:34-100) in two kernels.
// k_orientation: one wavefront per keypoint builds the 36-bin histogram (:49-66).  Samples are evaluated 64 at a
// time, but each histogram bin is accumulated by ONE lane walking the samples in the reference's (xx outer, yy inner)
// order, so the fp32 sums round identically.  k_orient_peaks: one THREAD per keypoint smooths its histogram
// (:70-75, a recurrence of 2 x 36 dependent steps that no wavefront can share) and picks the peaks (:78-99).
constexpr int ORI_BINS = 36;
#ifndef ORI_GRID_X
#define ORI_GRID_X 512
#endif
constexpr int ORI_COLCAP = 1024;     // in-circle samples of one keypoint in the column-interval enumeration (the shipped config needs ~200)
// The per-image descriptor counters that k_orient_peaks' wavefronts add to lie one per 128-byte line: a thousand
// atomics per image are nothing, but 38 adjacent counters are ONE line on ONE L2 channel, and 38 000 atomics in a row
// on it took twice as long as the whole kernel.
constexpr int OCNT_STRIDE = OP_OCNT_STRIDE;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// wave64 inclusive add-scan / max on the VALU data-parallel primitives (see descriptor.hip)
__device__ __forceinline__ int ori_scan_add(int v) {
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
	return v;
}
__device__ __forceinline__ int ori_rank_below(unsigned long long m) {
	return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ int ori_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float ori_uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
#define ORI_FENCE() asm volatile("" ::: "memory")      // one wavefront per workgroup: LDS accesses execute in program order

__global__ void __launch_bounds__(64) k_orientation(SiftPlan p, const KeyPoint* __restrict__ refined, const int* __restrict__ refined_count,
		int cap, float* __restrict__ hist_out, int sort_blocks, KeyPoint* __restrict__ sorted, int* __restrict__ slot_of) {
	__shared__ unsigned long long s_key[SORT_CHUNK];
	if ((int)blockIdx.x < sort_blocks) {               // the first sort_blocks wavefronts of every image rank its keypoints
		const long long at = (long long)blockIdx.y * cap;
		sort_refined_part(refined + at, refined_count[blockIdx.y], sorted + at, slot_of + at, blockIdx.x, sort_blocks, s_key);
		return;
	}
	const int first = blockIdx.x - sort_blocks, nblocks = gridDim.x - sort_blocks;
	__shared__ unsigned long long s_mask[ORI_BINS];   // per bin: bit l = lane l's sample of this round falls into it
	__shared__ __attribute__((aligned(16))) float s_sorted[64 + 3 * ORI_BINS + 12 + 4];   // the round's values, bin-major, sample order inside a bin; lists 16-byte aligned, zero-padded to float4s; last float4: zeros
	__shared__ unsigned short s_off[64];
	__shared__ unsigned long long s_startbits[ORI_COLCAP / 64];   // bit e: in-circle sample e is the first of its window column
	__shared__ unsigned s_colpk[64];                  // k-th non-empty window column: index of its first sample << 16 | (first row & 0xFF) << 8 | column
	constexpr int ZERO4 = (64 + 3 * ORI_BINS + 12) / 4;
	const int img = blockIdx.y;
	const int count = refined_count[img];
	const int lane = threadIdx.x;
	const float* base = p.ws + (long long)img * p.ws_stride;
	const float halfipi = (float)(0.5f / 3.14159265358979323846);
	if (lane < ORI_BINS) s_mask[lane] = 0ULL;
	if (lane < 4) s_sorted[ZERO4 * 4 + lane] = 0.f;
	__syncthreads();
	for (int k = first; k < count; k += nblocks) {
		const KeyPoint kp = refined[(long long)img * cap + k];
		const int kpx = ori_uni(kp.x), kpy = ori_uni(kp.y);
		const OctDesc od = p.oct[ori_uni(kp.oct)];
		const int w = od.w, h = od.h;
		// gradient magnitude / orientation of GaussianPyramid::cal_mag_ort (feature/dog.cc:76-84),
		// evaluated on the Gaussian plane for the window samples only
		const float* g_img = base + plane_off_gauss(od, p.nscale, ori_uni(kp.scale));
		const float sf = ori_uni(kp.sf);
		const float gauss_weight_sigma = sf * 1.5f;                 // ORI_WINDOW_FACTOR
		const int rad = ori_uni((int)roundf(sf * p.ori_radius));
		const float exp_denom = 2 * (gauss_weight_sigma * gauss_weight_sigma);
		const int side = 2 * rad, nsamp = side * side;
		const float frad2 = (float)rad * (float)rad;
		float hsum = 0.f;
		// ---- the samples the reference keeps (:49-57): window columns xx in [-rad, rad), rows yy in [-rad, rad), inside
		// the image's interior, inside the circle xx^2 + yy^2 <= rad^2 (a test on integers).  Per column those rows are
		// an interval, known exactly: the wavefront enumerates only them -- three quarters of the window -- in the
		// reference's (xx outer, yy inner) order, 64 per round.  Windows wider than 64 columns walk the full window.
		int ncand = nsamp;
		bool cols = side <= 64;
		if (cols) {
			int lo = 0, len = 0;
			if (lane < side) {
				const int xx = lane - rad;
				const float fxx = (float)xx;
				const int yc = (int)floorf(sqrtf(frad2 - fxx * fxx) + 1e-3f);       // |yy| <= floor(sqrt(rad^2 - xx^2)); exact for these integers
				int ilo = -yc, ihi = yc < rad - 1 ? yc : rad - 1;                    // yy < rad
				ilo = ilo < 1 - kpy ? 1 - kpy : ilo; ihi = ihi > h - 2 - kpy ? h - 2 - kpy : ihi;     // between(newy, 1, h - 1)
				const int newx = kpx + xx;
				if (newx >= 1 && newx <= w - 2 && ihi >= ilo) { lo = ilo; len = ihi - ilo + 1; }
			}
			const int incl = ori_scan_add(len);
			ncand = __builtin_amdgcn_readlane(incl, 63);
			cols = ncand <= ORI_COLCAP;
			if (cols) {
				const int start = incl - len;
				const unsigned long long nonempty = __ballot(len > 0);
				if (lane < ORI_COLCAP / 64) s_startbits[lane] = 0ULL;
				ORI_FENCE();
				if (len > 0) {
					atomicOr(&s_startbits[start >> 6], 1ULL << (start & 63));
					s_colpk[ori_rank_below(nonempty)] = ((unsigned)start << 16) | (((unsigned)lo & 0xFFu) << 8) | (unsigned)lane;
				}
			} else ncand = nsamp;
			ORI_FENCE();
		}
		// 64 samples per round in the reference's (xx outer, yy inner) order.  The round's values
		// are sorted by bin with order-free LDS mask ORs and popcount ranks (stable: sample order is
		// kept inside a bin), then lane b adds bin b's segment in order -- the fp32 sums round like
		// the sequential  hist[bin] += ...  of orientation.cc:49-66.
		int qx = lane / (side > 0 ? side : 1), qy = lane % (side > 0 ? side : 1);
		int kbase = -1;                                  // listed columns started before this round, minus one
		for (int i0 = 0; i0 < ncand; i0 += 64) {
			int bin = -1; float val = 0.f;
			int ord = 0;
			if (cols) {
				const unsigned long long sb = s_startbits[i0 >> 6];
				ord = kbase + ori_rank_below(sb) + (int)((unsigned)(sb >> lane) & 1u);
				kbase += __popcll(sb);
			}
			if (i0 + lane < ncand) {
				int xx, yy; bool in;
				if (cols) {
					const unsigned pk = s_colpk[ord];
					xx = (int)(pk & 0xFFu) - rad; yy = (int)(signed char)(pk >> 8) + (i0 + lane - (int)(pk >> 16));
					in = true;                                                           // the intervals lie inside the image and the circle
				} else {
					xx = qx - rad; yy = qy - rad;
					const int newx = kpx + xx, newy = kpy + yy;
					in = newx >= 1 && newx <= w - 2 && newy >= 1 && newy <= h - 2;
				}
				const float fxx = (float)xx, fyy = (float)yy;
				const float r2 = fxx * fxx + fyy * fyy;
				if (in && !(r2 > frad2)) {
					const int gi = (kpy + yy) * w + (kpx + xx);
					const float gdy = g_img[gi + w] - g_img[gi - w];
					const float gdx = g_img[gi + 1] - g_img[gi - 1];
					const float orient = opdev::fast_atan_plus_pi(gdy, gdx);
					bin = (int)roundf(36 * halfipi * orient);
					if (bin == ORI_BINS) bin = 0;
					const float weight = opdev::expf_glibc(-r2 / exp_denom);
					val = weight * opdev::hypotf_glibc(gdx, gdy);
				}
			}
			if (!cols) { qy += 64; while (qy >= side) { qy -= side; ++qx; } }
			if (bin >= 0) atomicOr(&s_mask[bin], 1ULL << lane);
			ORI_FENCE();
			const unsigned long long m = lane < ORI_BINS ? s_mask[lane] : 0ULL;
			const int pc = (__popcll(m) + 3) >> 2;             // list length in whole float4s
			const int ex = ori_scan_add(pc) - pc;              // exclusive wave scan of the padded bin sizes
			s_off[lane] = (unsigned short)(ex * 16);
			// +0.0f padding leaves the non-negative fp32 sums unchanged: the list's last float4 is cleared with one 16-byte
			// write (an empty list clears the float4 of zeros), the scatter below overwrites the slots that hold values
			((f32x4*)s_sorted)[pc ? ex + pc - 1 : ZERO4] = f32x4{0.f, 0.f, 0.f, 0.f};
			ORI_FENCE();
			if (bin >= 0) *(float*)((char*)s_sorted + s_off[bin] + ori_rank_below(s_mask[bin]) * 4) = val;
			ORI_FENCE();
			for (int e = 0; __ballot(e < pc) != 0ULL; e += 2) {
				const f32x4 a = ((const f32x4*)s_sorted)[e < pc ? ex + e : ZERO4], c = ((const f32x4*)s_sorted)[e + 1 < pc ? ex + e + 1 : ZERO4];
				hsum += a.x; hsum += a.y; hsum += a.z; hsum += a.w; hsum += c.x; hsum += c.y; hsum += c.z; hsum += c.w;
			}
			if (lane < ORI_BINS) s_mask[lane] = 0ULL;
			ORI_FENCE();
		}
		if (lane < ORI_BINS) hist_out[((long long)img * cap + k) * ORI_BINS + lane] = hsum;
	}
}

// Smoothing and peaks, one thread per keypoint.  The in-place smoothing (orientation.cc:70-75) is a recurrence: every
// step reads the ALREADY smoothed left neighbour, 36 dependent steps per pass.  A wavefront per keypoint cannot share
// them (all 64 lanes wait on one chain: it was a fifth of k_orientation's instructions); with a keypoint per LANE the
// histogram lives in 36 registers and every lane walks its own chain -- the reference's expression in the reference's
// types, no reformulation: hist[i] = (float)((double)hist[i] * 0.5 + (double)(prev + next) * 0.25).
// The raw histogram arrives in, and the peak directions leave through, the same 36 floats per keypoint.
__global__ void __launch_bounds__(256) k_orient_peaks(const int* __restrict__ refined_count, int cap, int smooth,
		float* dirs, int* __restrict__ ndirs, int* per_image) {
	const int img = blockIdx.y;
	const int count = refined_count[img];
	int npeaks = 0;
	for (int k = blockIdx.x * 256 + threadIdx.x; k < count; k += gridDim.x * 256) {
		float* row = dirs + ((long long)img * cap + k) * ORI_BINS;
		float hist[ORI_BINS];
#pragma unroll
		for (int i = 0; i < ORI_BINS / 4; ++i) {
			const f32x4 q = ((const f32x4*)row)[i];
			hist[4 * i] = q.x; hist[4 * i + 1] = q.y; hist[4 * i + 2] = q.z; hist[4 * i + 3] = q.w;
		}
		for (int K = smooth; K--;) {
#pragma unroll
			for (int i = 0; i < ORI_BINS; ++i) {
				const float prev = hist[i == 0 ? ORI_BINS - 1 : i - 1];
				const float next = hist[i == ORI_BINS - 1 ? 0 : i + 1];
				hist[i] = (float)((double)hist[i] * 0.5 + (double)(prev + next) * 0.25);
			}
		}
		float maxbin = 0.f;                                   // :78-80, update_max from 0: a NaN never replaces the maximum
#pragma unroll
		for (int i = 0; i < ORI_BINS; ++i) maxbin = hist[i] > maxbin ? hist[i] : maxbin;
		const float thres = maxbin * 0.8f;                              // ORI_HIST_PEAK_RATIO
		int np = 0;
#pragma unroll
		for (int i = 0; i < ORI_BINS; ++i) {
			const float hv = hist[i];
			const float prev = hist[i == 0 ? ORI_BINS - 1 : i - 1];
			const float next = hist[i == ORI_BINS - 1 ? 0 : i + 1];
			const float mpn = prev < next ? next : prev;
			if (hv > thres && hv > mpn) {
				double newbin = (double)(float)i - 0.5 + (double)((hv - prev) / (prev + next - 2 * hv));
				if (newbin < 0) newbin += ORI_BINS;
				else if (newbin >= ORI_BINS) newbin -= ORI_BINS;
				row[np++] = (float)(newbin / ORI_BINS * 2 * 3.14159265358979323846);
			}
		}
		ndirs[(long long)img * cap + k] = np;
		npeaks += np;
	}
	// the image's descriptor count (order-free): one atomic per wavefront
	const int wave_total = __builtin_amdgcn_readlane(ori_scan_add(npeaks), 63);
	if ((threadIdx.x & 63) == 0 && wave_total) atomicAdd(&per_image[img * OCNT_STRIDE], wave_total);
}

// expansion refined -> oriented in (refined order, peak order): OrientationAssign::work (:22-32).  One workgroup per
// image; its first output slot is the sum of the earlier images' counts (a few dozen ints, summed by every workgroup for
// itself: no offsets kernel), workgroup 0 also leaves the batch total for the descriptor kernel and the host.
__device__ __forceinline__ int exp_scan_add(int v) {
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
	return v;
}
constexpr int EXP_T = 1024;           // one pass over a thousand keypoints per image: every dependent round trip (counts -> keypoint + peaks -> store) once
__global__ void __launch_bounds__(EXP_T) k_expand_oriented(const KeyPoint* refined, const int* slot_of, const int* refined_count, int cap,
		const float* dirs, const int* ndirs, const int* per_image, int nimg, long long* total, int* count_out, KeyPoint* oriented, long long oriented_cap) {
	__shared__ long long s_before[EXP_T], s_all[EXP_T];
	__shared__ int s_wave[EXP_T / 64];
	const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int n = refined_count[img];
	{
		long long before = 0, all = 0;
		for (int i = tid; i < nimg; i += EXP_T) {
			const int v = per_image[i * OCNT_STRIDE]; all += v; if (i < img) before += v;
			if (img == 0) count_out[i] = v;                 // the counts, packed, next to the other counters the host reads
		}
		s_before[tid] = before; s_all[tid] = all;
		__syncthreads();
		for (int st = EXP_T / 2; st > 0; st >>= 1) {
			if (tid < st) { s_before[tid] += s_before[tid + st]; s_all[tid] += s_all[tid + st]; }
			__syncthreads();
		}
	}
	long long base = s_before[0];
	if (img == 0 && tid == 0) *total = s_all[0];
	for (int start = 0; start < n; start += EXP_T) {
		const int i = start + tid;
		const long long at = (long long)img * cap + (i < n ? slot_of[(long long)img * cap + i] : 0);      // the keypoint's slot in the unsorted list
		const int cnt = i < n ? ndirs[at] : 0;
		const int incl = exp_scan_add(cnt);
		if (lane == 63) s_wave[wave] = incl;
		__syncthreads();
		int wbase = 0, chunk = 0;
#pragma unroll
		for (int w = 0; w < EXP_T / 64; ++w) { const int v = s_wave[w]; wbase += w < wave ? v : 0; chunk += v; }
		const long long first = base + wbase + (incl - cnt);
		if (i < n) {
			KeyPoint kp = refined[(long long)img * cap + i];
			const float* d = dirs + at * ORI_BINS;
			for (int j = 0; j < cnt; ++j) {
				kp.dir = d[j]; kp.src = i; kp.pad = img;      // pad carries the image index to the descriptor kernel
				const long long slot = first + j;
				if (slot < oriented_cap) oriented[slot] = kp;       // speculative capacity: the host re-runs on overflow
			}
		}
		base += chunk;
		__syncthreads();
	}
}

}	// namespace

hipError_t launch_refine(const SiftPlan& p, const int* raw, const int* raw_count, int cap, int expect,
		KeyPoint* refined, int* refined_count, hipStream_t st) {
	const int want = expect > 0 ? expect + expect / 4 + 128 : cap;
	dim3 grid(((want < cap ? want : cap) + 127) / 128, p.n);
	hipLaunchKernelGGL(k_refine, grid, dim3(128), 0, st, p, raw, raw_count, cap, refined, refined_count);
	return hipGetLastError();
}

hipError_t launch_orientation(const SiftPlan& p, const KeyPoint* refined, const int* refined_count, int cap, int expect,
		float* dirs, int* ndirs, int* per_image, KeyPoint* sorted, int* slot_of, hipStream_t st) {
	// ORI_GRID_X wavefronts per image striding over the (device-side) keypoint count, about two keypoints each for the
	// thousand keypoints of a 1300 x 867 view (measured: 256 / 512 / 1024 / 2048 per image = 0.100 / 0.094 / 0.104 / 0.103 ms)
	// in front of them, per image, the wavefronts that rank the keypoints (8 each, striding: any number is correct)
	const int want = expect > 0 ? expect + expect / 4 + 256 : cap;
	const int sort_blocks = ((want < cap ? want : cap) + SORT_KEYS - 1) / SORT_KEYS;
	dim3 grid(sort_blocks + (cap < ORI_GRID_X ? cap : ORI_GRID_X), p.n);
	hipLaunchKernelGGL(k_orientation, grid, dim3(64), 0, st, p, refined, refined_count, cap, dirs, sort_blocks, sorted, slot_of);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(k_orient_peaks, dim3(((want < cap ? want : cap) + 255) / 256, p.n), dim3(256), 0, st, refined_count, cap, p.ori_smooth, dirs, ndirs, per_image);
	return hipGetLastError();
}

hipError_t launch_expand_oriented(const SiftPlan& p, const KeyPoint* refined, const int* slot_of, const int* refined_count, int cap,
		const float* dirs, const int* ndirs, const int* per_image, long long* total, int* count_out, KeyPoint* oriented, long long oriented_cap, hipStream_t st) {
	hipLaunchKernelGGL(k_expand_oriented, dim3(p.n), dim3(EXP_T), 0, st, refined, slot_of, refined_count, cap, dirs, ndirs, per_image, p.n, total, count_out, oriented, oriented_cap);
	return hipGetLastError();
}
