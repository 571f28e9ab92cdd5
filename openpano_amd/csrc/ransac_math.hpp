// ransac_math.hpp -- fp64 geometry shared by the RANSAC device kernel and its host epilogue.
//
// Restates TransformEstimation::calc_transform / get_inliers (stitch/transform_estimate.cc:89-148),
// getPerspectiveTransform / getAffineTransform (lib/imgproc.cc:251-317) and Homography::health
// (stitch/homography.hh:106-127).  The reference solves the DLT least-squares system with
// Eigen's JacobiSVD (system Eigen, absent here and unpinned by any reference test); this
// implementation uses a backward-stable Givens QR with row-by-row updating, which needs only the
// 8x8 triangle as state (one hypothesis per GPU lane) and handles any number of rows (the host
// refit on all inliers).  The same source is compiled for host and device with
// -ffp-contract=off, so both sides produce bit-identical homographies.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace opransac {

struct P2 { double x, y; };

// least squares min |A h - b| by Givens rotations, rows fed one at a time. NV = 8 (homography
// with h33 = 1) or 6 (affine).
template <int NV>
struct GivensLS {
	double R[NV][NV];
	double qtb[NV];
	__host__ __device__ void reset() {
		for (int i = 0; i < NV; ++i) { qtb[i] = 0; for (int j = 0; j < NV; ++j) R[i][j] = 0; }
	}
	__host__ __device__ void add_row(double (&a)[NV], double beta) {
#pragma unroll
		for (int k = 0; k < NV; ++k) {
			const double ak = a[k];
			if (ak == 0.0) continue;
			const double rkk = R[k][k];
			const double r = sqrt(rkk * rkk + ak * ak);
			const double c = rkk / r, s = ak / r;
			R[k][k] = r;
#pragma unroll
			for (int j = k + 1; j < NV; ++j) {
				const double t = c * R[k][j] + s * a[j];
				a[j] = c * a[j] - s * R[k][j];
				R[k][j] = t;
			}
			const double t = c * qtb[k] + s * beta;
			beta = c * beta - s * qtb[k];
			qtb[k] = t;
		}
	}
	__host__ __device__ void solve(double (&x)[NV]) const {
		double dmax = 0;
		for (int k = 0; k < NV; ++k) { const double d = fabs(R[k][k]); dmax = d > dmax ? d : dmax; }
		const double tiny = dmax * 1e-13;
#pragma unroll
		for (int k = NV - 1; k >= 0; --k) {
			double acc = qtb[k];
#pragma unroll
			for (int j = k + 1; j < NV; ++j) acc -= R[k][j] * x[j];
			x[k] = fabs(R[k][k]) > tiny ? acc / R[k][k] : 0.0;   // rank-deficient sample: drop the direction
		}
	}
};

// scale factor sqrt(2 / mean |p|^2) of transform_estimate.cc:99-114 (no centring: :104-107)
template <typename GetP>
__host__ __device__ inline double norm_scale(int n, GetP getp) {
	const double sizeinv = 1.0 / n;
	double sqrsum = 0;
	for (int i = 0; i < n; ++i) { const P2 p = getp(i); sqrsum += (p.x * p.x + p.y * p.y) * sizeinv; }
	return sqrt(2.0 / sqrsum);
}

// calc_transform: homography (affine = false) or affine map from image-2 to image-1 points.
// get1(i) / get2(i) return the i-th sample's point in image 1 / image 2. H is row-major 3x3.
template <typename Get1, typename Get2>
__host__ __device__ inline void calc_transform(int n, Get1 get1, Get2 get2, bool affine, double (&H)[9]) {
	const double s1 = norm_scale(n, get1), s2 = norm_scale(n, get2);
	double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
	if (!affine) {
		GivensLS<8> ls; ls.reset();
		// lib/imgproc.cc:267-274: rows 0..n-1 are the x equations, rows n..2n-1 the y equations
		for (int i = 0; i < n; ++i) {
			P2 m0 = get1(i), m1 = get2(i);
			m0.x *= s1; m0.y *= s1; m1.x *= s2; m1.y *= s2;
			double row[8] = {m1.x, m1.y, 1, 0, 0, 0, -m1.x * m0.x, -m1.y * m0.x};
			ls.add_row(row, m0.x);
		}
		for (int i = 0; i < n; ++i) {
			P2 m0 = get1(i), m1 = get2(i);
			m0.x *= s1; m0.y *= s1; m1.x *= s2; m1.y *= s2;
			double row[8] = {0, 0, 0, m1.x, m1.y, 1, -m1.x * m0.y, -m1.y * m0.y};
			ls.add_row(row, m0.y);
		}
		double x[8];
		ls.solve(x);
		for (int i = 0; i < 8; ++i) h[i] = x[i];
	} else {
		GivensLS<6> ls; ls.reset();
		// lib/imgproc.cc:304-310: rows interleaved x, y per point
		for (int i = 0; i < n; ++i) {
			P2 m0 = get1(i), m1 = get2(i);
			m0.x *= s1; m0.y *= s1; m1.x *= s2; m1.y *= s2;
			double r0[6] = {m1.x, m1.y, 1, 0, 0, 0};
			ls.add_row(r0, m0.x);
			double r1[6] = {0, 0, 0, m1.x, m1.y, 1};
			ls.add_row(r1, m0.y);
		}
		double x[6];
		ls.solve(x);
		for (int i = 0; i < 6; ++i) h[i] = x[i];
	}
	// t1.inverse() * H * t2 with t = diag(s, s, 1) (transform_estimate.cc:121-128)
	const double i1 = 1.0 / s1;
	const double l[3] = {i1, i1, 1.0}, r[3] = {s2, s2, 1.0};
	for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) H[a * 3 + b] = (l[a] * h[a * 3 + b]) * r[b];
}

// Homography::health (stitch/homography.hh:106-127)
__host__ __device__ inline bool health(const double (&m)[9]) {
	const double HOMO_MAX_PERSPECTIVE = (double)2e-3f;
	if (fabs(m[6]) > HOMO_MAX_PERSPECTIVE) return false;
	if (fabs(m[7]) > HOMO_MAX_PERSPECTIVE) return false;
	const double x0y = m[5], x1x = m[1] + m[2], x1y = m[4] + m[5];
	if (x1y <= x0y) return false;
	const double x2x = m[0] + m[1] + m[2];
	if (x2x <= x1x) return false;
	return true;
}

// one point of get_inliers (transform_estimate.cc:138-146): p2 -> image 1, squared distance to p1
__host__ __device__ inline bool is_inlier(const double (&H)[9], P2 p1, P2 p2, double inlier_dist) {
	const double tx = p2.x * H[0] + p2.y * H[1] + 1.0 * H[2];
	const double ty = p2.x * H[3] + p2.y * H[4] + 1.0 * H[5];
	const double tz = p2.x * H[6] + p2.y * H[7] + 1.0 * H[8];
	const double idenom = 1.0 / tz;
	const double dx = tx * idenom - p1.x, dy = ty * idenom - p1.y;
	return dx * dx + dy * dy < inlier_dist;
}

// std::mt19937
struct MT19937 {
	unsigned mt[624]; int idx;
	__host__ __device__ void seed(unsigned s) {
		mt[0] = s;
		for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned)i;
		idx = 624;
	}
	__host__ __device__ unsigned next() {
		if (idx >= 624) {
			for (int i = 0; i < 624; ++i) {
				const unsigned y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
				mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
			}
			idx = 0;
		}
		unsigned y = mt[idx++];
		y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
		return y;
	}
};

}	// namespace opransac
