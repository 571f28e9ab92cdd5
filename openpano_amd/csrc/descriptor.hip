// descriptor.hip -- 4x4x8 RootSIFT descriptor, one wavefront per oriented keypoint.
//
// Replaces SIFT::calc_descriptor / trilinear_interpolate / hist_to_descriptor
// (feature/sift.cc:15-152) and the coordinate shift of FeatureDetector::detect_feature
// (feature/feature.cc:20-28).
//
// Bit-exactness: the reference accumulates hist[bin] += w in window order (xx outer, yy inner)
// in fp32, so the summation ORDER is part of the result.  Structure per keypoint:
//   1a  all 64 lanes test window samples (bounds, circle, rotated bin range), 64 per round;
//   1b  surviving samples are compacted IN ORDER into LDS records (weights of the 2x2 spatial
//       cells, orientation fraction) -- the expensive part (expf, gathers) runs on survivors only;
//   1c  per spatial cell an ordered list of the records that touch it is built with wave
//       ballots (16 ballots per 64 records);
//   2   lane (cell, hj) walks its cell's list in order and adds the record's contribution to
//       the orientation bins hj and hj+4 it owns.  A non-matching orientation adds +0.0f, which
//       is exact, so every bin sees exactly the reference's sequence of fp32 additions.
// Records are flushed through 1c/2 whenever an LDS buffer would overflow, so any window size
// is handled with the same ordering guarantee.
#include "internal.hpp"
#include "devmath.hpp"

namespace {

constexpr int REC_CAP = 256;         // records (surviving samples) buffered per flush
constexpr int LIST_CAP = 128;        // entries per spatial-cell list per flush
constexpr int QCAP = 128;            // survivor queue (power of two, >= 2 x 64)

struct DescLds {
	float w[4][REC_CAP];             // w_x of (dy,dx) = (0,0),(0,1),(1,0),(1,1)   (sift.cc:59-61)
	float hb[REC_CAP];               // hbind
	float omh[REC_CAP];              // 1 - hbind
	// entry = record | u << 9 | h0 << 11 ; row pitch LIST_CAP + 8: the 16 cells' 16-byte reads hit disjoint banks
	__attribute__((aligned(16))) unsigned short list[16][LIST_CAP + 8];
	int len[16];
	float hist[128];
	int q_gi[QCAP];                  // survivor queue (ring): plane offset, rotated coordinates
	float q_xr[QCAP], q_yr[QCAP];
};

__global__ void __launch_bounds__(64) k_descriptor(SiftPlan p, const KeyPoint* oriented,
		const long long* img_offset, long long cap, float* desc, double* coor, double* real) {
	__shared__ DescLds S;
	const int lane = threadIdx.x;
	const unsigned long long lt_mask = (1ULL << lane) - 1ULL;
	const float pi2 = (float)(2 * 3.14159265358979323846);
	const float nbin_per_rad = 8 / pi2;
	const int cell = lane >> 2, hj = lane & 3;

	long long total = img_offset[p.n];                 // device-side count (k_image_offsets)
	total = total < cap ? total : cap;                 // speculative capacity: the host re-runs on overflow
	for (long long kk = blockIdx.x; kk < total; kk += gridDim.x) {
		int img = 0;
		while (img + 1 < p.n && kk >= img_offset[img + 1]) ++img;
		const KeyPoint kp = oriented[kk];
		const OctDesc od = p.oct[kp.oct];
		const int w = od.w, h = od.h;
		const float* base = p.ws + (long long)img * p.ws_stride;
		// mag / ort of GaussianPyramid::cal_mag_ort (feature/dog.cc:76-84) are evaluated on the
		// Gaussian plane for the surviving window samples only
		const float* g_img = base + plane_off_gauss(od, p.nscale, kp.scale);
		const float ort = kp.dir;
		const float hist_w = kp.sf * (float)p.desc_scale_factor;
		const float exp_denom = 2 * (4.f * 4.f);
		const int radius = (int)round(0.70710678118654752440 * (double)hist_w * (4 + 1));
		const float cosort = opdev::cosf_glibc(ort), sinort = opdev::sinf_glibc(ort);
		const int side = 2 * radius + 1, nsamp = side * side;
		const float fr2 = (float)radius * (float)radius;
		float acc0 = 0.f, acc1 = 0.f;     // bins (cell, hj) and (cell, hj + 4)
		int nrec = 0;                     // records buffered (wave-uniform)
		int len[16];                      // list lengths (wave-uniform)
#pragma unroll
		for (int c = 0; c < 16; ++c) len[c] = 0;
		int maxlen = 0;
		int qhead = 0, qn = 0;            // survivor queue state (wave-uniform)

		auto flush = [&]() {
			// phase 2: ordered accumulation from the cell lists
			if (lane < 16) {
				int v = 0;
#pragma unroll
				for (int c = 0; c < 16; ++c) v = (lane == c) ? len[c] : v;
				S.len[lane] = v;
			}
			__syncthreads();
			const int mylen = S.len[cell];
			const unsigned short* mylist = S.list[cell];
			// 8 list entries per step: one 16-byte LDS read for the entries, then their 16 operand
			// reads back to back, then the 8 ordered additions.  Entries past the end of this
			// cell's list are stale data: they are masked to a +0.0f contribution (exact).
			for (int t0 = 0; t0 < maxlen; t0 += 8) {
				const uint4 pk = *(const uint4*)(mylist + t0);
				const unsigned e2[4] = {pk.x, pk.y, pk.z, pk.w};
				float wx[8], fac[8]; int dd[8], hh[8];
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const unsigned e = (e2[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
					const int ridx = e & (REC_CAP - 1), u = (e >> 9) & 3, h0 = (e >> 11) & 7;
					const int d = (hj - h0) & 3;
					wx[k] = S.w[u][ridx];
					fac[k] = d ? S.hb[ridx] : S.omh[ridx];
					dd[k] = (t0 + k < mylen) ? d : 3;          // 3: contributes +0.0f
					hh[k] = ((h0 + d) >> 2) & 1;                // bin hbinf%8 / (hbinf+1)%8 in the upper half?
				}
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const float v = wx[k] * fac[k];             // sift.cc:63-64
					const float c = dd[k] < 2 ? v : 0.f;        // + 0.0f is exact
					acc0 += hh[k] ? 0.f : c;
					acc1 += hh[k] ? c : 0.f;
				}
			}
			__syncthreads();
			nrec = 0; maxlen = 0;
#pragma unroll
			for (int c = 0; c < 16; ++c) len[c] = 0;
		};

		// phases 1b + 1c on one dense batch of queued survivors (window order preserved)
		auto process_batch = [&](int n) {
			// flush first if this batch could overflow a buffer
			if (nrec + 64 > REC_CAP || maxlen + 64 > LIST_CAP) flush();
			const bool ok = lane < n;
			const int qi = (qhead + lane) & (QCAP - 1);
			const int ridx = nrec + lane;
			int yb = -9, xb = -9, h0 = 0;
			if (ok) {
				const int gi = S.q_gi[qi];
				const float x_rot = S.q_xr[qi], y_rot = S.q_yr[qi];
				const float ybin = (y_rot + 2.f) - 0.5f, xbin = (x_rot + 2.f) - 0.5f;
				const float gdy = g_img[gi + w] - g_img[gi - w];
				const float gdx = g_img[gi + 1] - g_img[gi - 1];
				const float now_mag = opdev::hypotf_glibc(gdx, gdy);
				float now_ort = opdev::fast_atan_plus_pi(gdy, gdx);
				float weight = opdev::expf_glibc(-(x_rot * x_rot + y_rot * y_rot) / exp_denom);
				weight = weight * now_mag;
				now_ort -= ort;
				if (now_ort < 0) now_ort += pi2;
				if (now_ort > pi2) now_ort -= pi2;
				const float hbin = now_ort * nbin_per_rad;
				const float yf = floorf(ybin), xf = floorf(xbin), hf = floorf(hbin);
				yb = (int)yf; xb = (int)xf; h0 = ((int)hf) & 7;     // hbinf % 8 (hbinf in 0..8)
				const float ybind = ybin - yf, xbind = xbin - xf, hbind = hbin - hf;
				const float wy0 = weight * (1 - ybind), wy1 = weight * ybind;
				S.w[0][ridx] = wy0 * (1 - xbind); S.w[1][ridx] = wy0 * xbind;
				S.w[2][ridx] = wy1 * (1 - xbind); S.w[3][ridx] = wy1 * xbind;
				S.hb[ridx] = hbind; S.omh[ridx] = 1 - hbind;
			}
			nrec += n;
			// phase 1c: ordered per-cell lists
#pragma unroll
			for (int c = 0; c < 16; ++c) {
				const int dy = (c >> 2) - yb, dx = (c & 3) - xb;
				const bool touch = ok && (unsigned)dy < 2u && (unsigned)dx < 2u;
				const unsigned long long m = __ballot(touch);
				if (touch) S.list[c][len[c] + __popcll(m & lt_mask)] = (unsigned short)(ridx | ((dy * 2 + dx) << 9) | (h0 << 11));
				len[c] += __popcll(m);
				maxlen = len[c] > maxlen ? len[c] : maxlen;
			}
			qhead = (qhead + n) & (QCAP - 1); qn -= n;
		};

		// phase 1a: window scan in the reference's order (xx outer, yy inner: sift.cc:110-113); the
		// cheap tests run on all samples, survivors are queued in order and handed to the expensive
		// phases 64 at a time, so those always run with full wavefronts
		int qx = lane / side, qy = lane % side;          // sample e = i0 + lane  ->  (e / side, e % side)
		for (int i0 = 0; i0 < nsamp; i0 += 64) {
			bool ok = false;
			float x_rot = 0.f, y_rot = 0.f;
			int gi = 0;
			if (i0 + lane < nsamp) {
				const int xx = qx - radius, yy = qy - radius;
				const int nowx = kp.x + xx, nowy = kp.y + yy;
				if (nowx >= 1 && nowx <= w - 2 && nowy >= 1 && nowy <= h - 2) {
					const float fxx = (float)xx, fyy = (float)yy;
					if (!(fxx * fxx + fyy * fyy > fr2)) {
						y_rot = ((float)(-xx) * sinort + fyy * cosort) / hist_w;
						x_rot = (fxx * cosort + fyy * sinort) / hist_w;
						const float ybin = (y_rot + 2.f) - 0.5f, xbin = (x_rot + 2.f) - 0.5f;
						// between(bin, -1, 4) on floats is  -1 <= bin <= 3  (lib/utils.hh:27)
						ok = (ybin >= -1.f && ybin <= 3.f && xbin >= -1.f && xbin <= 3.f);
						gi = nowy * w + nowx;
					}
				}
			}
			qy += 64;
			while (qy >= side) { qy -= side; ++qx; }
			const unsigned long long mask = __ballot(ok);
			if (ok) {
				const int qi = (qhead + qn + __popcll(mask & lt_mask)) & (QCAP - 1);
				S.q_gi[qi] = gi; S.q_xr[qi] = x_rot; S.q_yr[qi] = y_rot;
			}
			qn += __popcll(mask);
			if (qn >= 64) { __syncthreads(); process_batch(64); }
		}
		if (qn > 0) { __syncthreads(); process_batch(qn); }
		flush();

		// hist_to_descriptor (:15-46): L1-normalise (sequential fp32 sum), sqrt, * DESC_INT_FACTOR
		S.hist[cell * 8 + hj] = acc0;
		S.hist[cell * 8 + hj + 4] = acc1;
		__syncthreads();
		float sum = 0.f;
		for (int i = 0; i < 128; ++i) sum += S.hist[i];
		float* out = desc + kk * 128;
#pragma unroll
		for (int t = 0; t < 2; ++t) {
			const int i = lane + 64 * t;
			const float v = S.hist[i] / sum;
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

hipError_t launch_descriptor(const SiftPlan& p, const KeyPoint* oriented, const long long* img_offset,
		long long cap, float* desc, double* coor, double* real, hipStream_t st) {
	if (cap <= 0) return hipSuccess;
	const int grid = (int)(cap < 32768 ? cap : 32768);          // wavefronts beyond the device-side count exit at once
	hipLaunchKernelGGL(k_descriptor, dim3(grid), dim3(64), 0, st, p, oriented, img_offset, cap, desc, coor, real);
	return hipGetLastError();
}
