// descriptor.hip -- 4x4x8 RootSIFT descriptor, one wavefront per oriented keypoint.
//
// Replaces SIFT::calc_descriptor / trilinear_interpolate / hist_to_descriptor
// (feature/sift.cc:15-152) and the coordinate shift of FeatureDetector::detect_feature
// (feature/feature.cc:20-28).
//
// Bit-exactness: the reference accumulates hist[bin] += w in window order (xx outer, yy inner)
// in fp32, so the summation order is part of the result.  Samples are therefore evaluated 64 at
// a time (all lanes), compacted in order into LDS, and then every histogram bin is accumulated
// by exactly one lane walking the samples in that order (lane l owns bins l and l+64: cell
// l>>2 (+0), orientation bins (l&3) and (l&3)+4 ... see below).
#include "internal.hpp"
#include "devmath.hpp"

namespace {

constexpr int DESC_CHUNK = 512;      // window samples staged per pass

struct SampleRec {                   // one window sample that passed every test (sift.cc:110-128)
	float w00, w01, w10, w11;        // weight * {1-ybind, ybind} * {1-xbind, xbind}  (w_x of :61)
	float hbind;                     // fractional orientation bin
	int packed;                      // (ybinf+1) | (xbinf+1) << 4 | (hbinf & 15) << 8
};

// lane l owns spatial cell (l >> 2) and the two orientation bins (l & 3) and (l & 3) + 4
__global__ void __launch_bounds__(64) k_descriptor(SiftPlan p, const KeyPoint* oriented,
		const long long* img_offset, long long total, float* desc, double* coor) {
	__shared__ SampleRec s_rec[DESC_CHUNK];
	__shared__ int s_count;
	__shared__ float s_hist[128];
	const int lane = threadIdx.x;
	const float pi2 = (float)(2 * 3.14159265358979323846);
	const float nbin_per_rad = 8 / pi2;
	for (long long kk = blockIdx.x; kk < total; kk += gridDim.x) {
		// image of this keypoint: img_offset is a short ascending table
		int img = 0;
		while (img + 1 < p.n && kk >= img_offset[img + 1]) ++img;
		const KeyPoint kp = oriented[kk];
		const OctDesc od = p.oct[kp.oct];
		const int w = od.w, h = od.h;
		const float* base = p.ws + (long long)img * p.ws_stride;
		const float* mag_img = base + plane_off_mag(od, p.nscale, kp.scale);
		const float* ort_img = base + plane_off_ort(od, p.nscale, kp.scale);
		const float ort = kp.dir;
		const float hist_w = kp.sf * (float)p.desc_scale_factor;
		const float exp_denom = 2 * (4.f * 4.f);
		const int radius = (int)round(0.70710678118654752440 * (double)hist_w * (4 + 1));
		const float cosort = opdev::cosf_glibc(ort), sinort = opdev::sinf_glibc(ort);
		const int side = 2 * radius + 1, nsamp = side * side;
		const float fr2 = (float)radius * (float)radius;
		const int cell = lane >> 2, by = cell >> 2, bx = cell & 3, hj = lane & 3;
		float acc0 = 0.f, acc1 = 0.f;     // bins (cell, hj) and (cell, hj + 4)

		for (int cb = 0; cb < nsamp; cb += DESC_CHUNK) {
			if (lane == 0) s_count = 0;
			__syncthreads();
			// phase 1: evaluate up to DESC_CHUNK window samples, ordered compaction into s_rec
			for (int i0 = 0; i0 < DESC_CHUNK && cb + i0 < nsamp; i0 += 64) {
				const int e = cb + i0 + lane;
				bool ok = false;
				SampleRec rec;
				if (e < nsamp && i0 + lane < DESC_CHUNK) {
					const int xx = e / side - radius, yy = e % side - radius;
					const int nowx = kp.x + xx, nowy = kp.y + yy;
					if (nowx >= 1 && nowx <= w - 2 && nowy >= 1 && nowy <= h - 2) {
						const float fxx = (float)xx, fyy = (float)yy;
						if (!(fxx * fxx + fyy * fyy > fr2)) {
							const float y_rot = ((float)(-xx) * sinort + fyy * cosort) / hist_w;
							const float x_rot = (fxx * cosort + fyy * sinort) / hist_w;
							const float ybin = (y_rot + 2.f) - 0.5f, xbin = (x_rot + 2.f) - 0.5f;
							// between(bin, -1, 4) on floats is  -1 <= bin <= 3  (lib/utils.hh:27)
							if (ybin >= -1.f && ybin <= 3.f && xbin >= -1.f && xbin <= 3.f) {
								const long long gi = (long long)nowy * w + nowx;
								const float now_mag = mag_img[gi];
								float now_ort = ort_img[gi];
								float weight = opdev::expf_glibc(-(x_rot * x_rot + y_rot * y_rot) / exp_denom);
								weight = weight * now_mag;
								now_ort -= ort;
								if (now_ort < 0) now_ort += pi2;
								if (now_ort > pi2) now_ort -= pi2;
								const float hbin = now_ort * nbin_per_rad;
								// trilinear_interpolate (:48-67)
								const float yf = floorf(ybin), xf = floorf(xbin), hf = floorf(hbin);
								const int ybinf = (int)yf, xbinf = (int)xf, hbinf = (int)hf;
								const float ybind = ybin - (float)ybinf, xbind = xbin - (float)xbinf;
								const float wy0 = weight * (1 - ybind), wy1 = weight * ybind;
								rec.w00 = wy0 * (1 - xbind); rec.w01 = wy0 * xbind;
								rec.w10 = wy1 * (1 - xbind); rec.w11 = wy1 * xbind;
								rec.hbind = hbin - (float)hbinf;
								rec.packed = (ybinf + 1) | ((xbinf + 1) << 4) | ((hbinf & 15) << 8);
								ok = true;
							}
						}
					}
				}
				const unsigned long long mask = __ballot(ok);
				const int basec = s_count;
				if (ok) s_rec[basec + __popcll(mask & ((1ULL << lane) - 1ULL))] = rec;
				__syncthreads();
				if (lane == 0) s_count = basec + __popcll(mask);
				__syncthreads();
			}
			// phase 2: each lane folds the samples, in order, into the bins it owns
			const int cnt = s_count;
			for (int i = 0; i < cnt; ++i) {
				const SampleRec r = s_rec[i];
				const int dy = by - ((r.packed & 15) - 1), dx = bx - (((r.packed >> 4) & 15) - 1);
				if ((unsigned)dy < 2u && (unsigned)dx < 2u) {
					const float wx = dy ? (dx ? r.w11 : r.w10) : (dx ? r.w01 : r.w00);
					const int hb = (r.packed >> 8) & 15;        // hbinf in 0..8
					const int h0 = hb & 7, h1 = (hb + 1) & 7;   // hbinf % 8, (hbinf + 1) % 8
					if ((h0 & 3) == hj) {
						const float v = wx * (1 - r.hbind);
						if (h0 >> 2) acc1 += v; else acc0 += v;
					} else if ((h1 & 3) == hj) {
						const float v = wx * r.hbind;
						if (h1 >> 2) acc1 += v; else acc0 += v;
					}
				}
			}
			__syncthreads();
		}
		// hist_to_descriptor (:15-46): L1-normalise (sequential fp32 sum), sqrt, * DESC_INT_FACTOR
		s_hist[cell * 8 + hj] = acc0;
		s_hist[cell * 8 + hj + 4] = acc1;
		__syncthreads();
		float sum = 0.f;
		for (int i = 0; i < 128; ++i) sum += s_hist[i];
		float* out = desc + kk * 128;
#pragma unroll
		for (int t = 0; t < 2; ++t) {
			const int i = lane + 64 * t;
			const float v = s_hist[i] / sum;
			out[i] = sqrtf(v) * (float)p.desc_int_factor;
		}
		if (lane == 0) {   // feature/feature.cc:23-26
			coor[kk * 2] = (kp.rx - 0.5) * (double)p.sw;
			coor[kk * 2 + 1] = (kp.ry - 0.5) * (double)p.sh;
		}
		__syncthreads();
	}
}

}	// namespace

hipError_t launch_descriptor(const SiftPlan& p, const KeyPoint* oriented, const long long* img_offset,
		long long total, float* desc, double* coor, hipStream_t st) {
	if (total <= 0) return hipSuccess;
	const int grid = (int)(total < 16384 ? total : 16384);
	hipLaunchKernelGGL(k_descriptor, dim3(grid), dim3(64), 0, st, p, oriented, img_offset, total, desc, coor);
	return hipGetLastError();
}
