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
//       in sample order inside each bin;
//   3   lane L owns bins 2L and 2L+1 and adds their segments in order to its two fp32
//       accumulators -- every bin sees exactly the reference's sequence of additions.
#include "internal.hpp"
#include "devmath.hpp"

namespace {

constexpr int QCAP = 128;            // survivor queue (power of two, >= 2 x 64)

struct DescLds {
	unsigned long long mask[128];    // per bin: bit l = lane l's sample of the current batch contributes
	float sorted[512];               // the batch's contributions, bin-major, sample order inside a bin
	unsigned short off[128];         // first slot of every bin in sorted[]
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
	S.mask[2 * lane] = 0ULL; S.mask[2 * lane + 1] = 0ULL;
	__syncthreads();

	long long total = img_offset[p.n];                 // device-side count (k_image_offsets)
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
		const float exp_denom = 2 * (4.f * 4.f);
		const int radius = (int)round(0.70710678118654752440 * (double)hist_w * (4 + 1));
		const float cosort = opdev::cosf_glibc(ort), sinort = opdev::sinf_glibc(ort);
		const int side = 2 * radius + 1, nsamp = side * side;
		const float fr2 = (float)radius * (float)radius;
		float acc0 = 0.f, acc1 = 0.f;     // bins 2 * lane and 2 * lane + 1
		int qhead = 0, qn = 0;            // survivor queue state (wave-uniform)

		// phases 1b - 3 on one dense batch of queued survivors (window order preserved)
		auto process_batch = [&](int n) {
			const bool ok = lane < n;
			const int qi = (qhead + lane) & (QCAP - 1);
			int bin[8]; float val[8];
#pragma unroll
			for (int c = 0; c < 8; ++c) { bin[c] = -1; val[c] = 0.f; }
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
				// trilinear_interpolate (sift.cc:48-67)
				const float yf = floorf(ybin), xf = floorf(xbin), hf = floorf(hbin);
				const int yb = (int)yf, xb = (int)xf, h0 = (int)hf;
				const float ybind = ybin - yf, xbind = xbin - xf, hbind = hbin - hf;
				const float omh = 1 - hbind;
#pragma unroll
				for (int dy = 0; dy < 2; ++dy) {
					const float w_y = weight * (dy ? ybind : 1 - ybind);
#pragma unroll
					for (int dx = 0; dx < 2; ++dx) {
						const int cy = yb + dy, cx = xb + dx;
						if ((unsigned)cy < 4u && (unsigned)cx < 4u) {        // between(., 0, DESC_HIST_WIDTH)
							const float w_x = w_y * (dx ? xbind : 1 - xbind);
							const int cellbase = (cy * 4 + cx) * 8, u = dy * 2 + dx;
							bin[2 * u] = cellbase + (h0 & 7);          val[2 * u] = w_x * omh;       // hbinf % 8
							bin[2 * u + 1] = cellbase + ((h0 + 1) & 7);  val[2 * u + 1] = w_x * hbind;  // (hbinf + 1) % 8
						}
					}
				}
			}
			// phase 2: stable counting sort of the contributions by bin
#pragma unroll
			for (int c = 0; c < 8; ++c)
				if (bin[c] >= 0) atomicOr(&S.mask[bin[c]], 1ULL << lane);
			__syncthreads();
			const unsigned long long m0 = S.mask[2 * lane], m1 = S.mask[2 * lane + 1];
			const int c0 = __popcll(m0), c1 = __popcll(m1);
			int incl = c0 + c1;                               // inclusive wave scan of the per-lane pair counts
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
			const int ex = incl - (c0 + c1);
			S.off[2 * lane] = (unsigned short)ex; S.off[2 * lane + 1] = (unsigned short)(ex + c0);
			__syncthreads();
#pragma unroll
			for (int c = 0; c < 8; ++c)
				if (bin[c] >= 0) S.sorted[S.off[bin[c]] + __popcll(S.mask[bin[c]] & lt_mask)] = val[c];
			__syncthreads();
			// phase 3: ordered accumulation of this lane's two bins
			for (int e = 0; e < c0; ++e) acc0 += S.sorted[ex + e];
			for (int e = 0; e < c1; ++e) acc1 += S.sorted[ex + c0 + e];
			S.mask[2 * lane] = 0ULL; S.mask[2 * lane + 1] = 0ULL;
			__syncthreads();
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

		// hist_to_descriptor (:15-46): L1-normalise (sequential fp32 sum), sqrt, * DESC_INT_FACTOR
		S.hist[2 * lane] = acc0;
		S.hist[2 * lane + 1] = acc1;
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
	const int grid = (int)(cap < (1 << 20) ? cap : (1 << 20));  // one keypoint per wavefront; wavefronts beyond the device-side count exit at once
	hipLaunchKernelGGL(k_descriptor, dim3(grid), dim3(64), 0, st, p, oriented, img_offset, cap, desc, coor, real);
	return hipGetLastError();
}
