// ransac_accept.hpp -- the per-pair acceptance epilogue of TransformEstimation: fill_inliers_to_matchinfo
// (stitch/transform_estimate.cc:150-218) with overlap_region (stitch/homography.cc:50-90), convex_hull / polygon_area /
// PointInPolygon (lib/polygon.cc:17-82, lib/polygon.hh:30-52) and Homography::inverse (stitch/homography.cc:25-39).
// Host code, fp64 with the host libm, shared by op_ransac_pairs (ransac.hip) and a CPU test harness
// (tests/test_ransac_accept_cpu.py compiles it with g++): no device symbols in here.
#pragma once
#include "ransac_math.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

namespace opaccept {
using opransac::P2;

struct Shape { int w, h; };
inline bool shifted_in(const Shape& s, P2 p) {        // match_info.hh:68-70
	return p.x >= -s.w * 0.5 && p.x < s.w * 0.5 && p.y >= -s.h * 0.5 && p.y < s.h * 0.5;
}
inline double side(P2 a, P2 b, P2 p) { return (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x); }   // polygon.cc:9-11

inline std::vector<P2> convex_hull(std::vector<P2>& pts) {  // lib/polygon.cc:17-46
	if (pts.size() <= 3) return pts;
	std::sort(pts.begin(), pts.end(), [](const P2& a, const P2& b) { if (a.y == b.y) return a.x < b.x; return a.y < b.y; });
	std::vector<P2> ret;
	ret.push_back(pts[0]); ret.push_back(pts[1]);
	const int n = (int)pts.size();
	for (int i = 2; i < n; ++i) {
		while (ret.size() >= 2 && side(ret[ret.size() - 2], ret.back(), pts[i]) <= 0) ret.pop_back();
		ret.push_back(pts[i]);
	}
	const size_t mid = ret.size();
	ret.push_back(pts[n - 2]);
	for (int i = n - 3; i >= 0; --i) {
		while (ret.size() > mid && side(ret[ret.size() - 2], ret.back(), pts[i]) <= 0) ret.pop_back();
		ret.push_back(pts[i]);
	}
	return ret;
}

inline double polygon_area(const std::vector<P2>& poly) {   // lib/polygon.cc:48-60
	const int n = (int)poly.size();
	double sum = 0;
	for (int i = 0; i < n; ++i) sum += poly[i].x * (poly[(i + 1) % n].y - poly[(i + n - 1) % n].y);
	return 0.5 * std::fabs(sum);
}

// atan2 for the wedge search below: |fast_atan2(y, x) - atan2(y, x)| < 1e-10 for finite arguments (octant reduction, one
// more reduction at tan(pi/8), a degree-6 minimax polynomial in the squared argument: 8e-12; tests/test_ransac_accept_cpu.py
// measures it against libm).  The sign conventions at the cut are atan2's: (+-0, x < 0) -> +-pi.
inline double fast_atan2(double y, double x) {
	const double ay = std::fabs(y), ax = std::fabs(x);
	const double hi = ax > ay ? ax : ay, lo = ax > ay ? ay : ax;
	if (!(hi > 0.0) || !(hi < 1.7976931348623157e308)) return std::atan2(y, x);      // zeros, infinities, NaNs: libm decides
	double a = lo / hi, base = 0.0;
	if (a > 0.41421356237309503) { a = (a - 1.0) / (a + 1.0); base = 0.78539816339744828; }
	const double s = a * a;
	double p = 0.047129968897061815;
	p = p * s + -0.08459108968229646; p = p * s + 0.11041054122251703; p = p * s + -0.14281639256027157;
	p = p * s + 0.19999885898543113; p = p * s + -0.3333333212787719; p = p * s + 0.9999999999791288;
	double r = base + a * p;                                   // atan(lo / hi) in [0, pi/4]
	if (ay > ax) r = 1.5707963267948966 - r;                   // [0, pi/2]
	if (std::signbit(x)) r = 3.141592653589793 - r;            // [0, pi]
	return std::signbit(y) ? -r : r;
}

struct PointInPolygon {                               // lib/polygon.hh:30-52, polygon.cc:62-82
	const std::vector<P2>& poly; P2 com; std::vector<std::pair<float, int>> slopes;
	// per wedge position w = 0..n (the lower_bound result; n = past the end, which wraps): the edge (p1, p2) the reference
	// tests against and side(p1, p2, com) -- point-independent, so they are made once per polygon
	struct Wedge { P2 p1, p2; double o1; };
	std::vector<Wedge> wedges;
	std::vector<double> ang;                          // the sorted vertex angles as doubles (for the margin test)
	explicit PointInPolygon(const std::vector<P2>& p): poly(p) {
		com = P2{0, 0};
		for (auto& c : poly) { com.x += c.x; com.y += c.y; }
		const double f = 1.0 / poly.size();
		com.x *= f; com.y *= f;
		for (size_t i = 0; i < p.size(); ++i) slopes.emplace_back((float)std::atan2(p[i].y - com.y, p[i].x - com.x), (int)i);
		std::sort(slopes.begin(), slopes.end());
		const int n = (int)slopes.size();
		wedges.resize(n + 1); ang.resize(n);
		for (int w = 0; w <= n; ++w) {
			int idx1, idx2;                            // polygon.cc:70-77
			if (w == n) { idx1 = slopes.back().second; idx2 = slopes.front().second; }
			else { idx2 = slopes[w].second; idx1 = w > 0 ? slopes[w - 1].second : slopes.back().second; }
			wedges[w] = Wedge{poly[idx1], poly[idx2], side(poly[idx1], poly[idx2], com)};
			if (w < n) ang[w] = (double)slopes[w].first;
		}
	}
	// the reference's test, word for word: k = (float)atan2(...), lower_bound over the sorted slopes, side of the wedge's edge
	bool in_polygon_exact(P2 p) const {
		const float k = (float)std::atan2(p.y - com.y, p.x - com.x);
		return decide((int)(std::lower_bound(slopes.begin(), slopes.end(), std::make_pair(k, 0)) - slopes.begin()), p);
	}
	// The same answer without libm's atan2 for (nearly) every point.  The wedge is the number of vertex angles below
	// k = fl32(atan2): with t = fast_atan2 (|t - atan2| < 1e-10) and |k - atan2| <= 2^-24 pi < 1.9e-7, every vertex angle
	// farther than 3e-7 from t compares with k as it compares with t -- so when NO vertex angle lies within 3e-7 of t the
	// count of angles below t is the reference's lower_bound position; otherwise (a point within 3e-7 rad of a vertex
	// direction: one in a million) the reference's own expression decides.  An overlap polygon has a handful of vertices:
	// the count is a branch-free pass over them.  A stitching job asks this for every keypoint of both images of every
	// candidate pair: libm's atan2 and the binary search on (float, int) pairs were most of the acceptance epilogue.
	bool in_polygon(P2 p) const {
		const double t = fast_atan2(p.y - com.y, p.x - com.x);
		const int n = (int)ang.size();
		int below = 0; double nearest = 1e30;
		for (int i = 0; i < n; ++i) {
			const double d = t - ang[i];
			below += d > 0.0 ? 1 : 0;
			const double ad = std::fabs(d);
			nearest = ad < nearest ? ad : nearest;
		}
		if (!(nearest > 3e-7) || t != t) return in_polygon_exact(p);
		return decide(below, p);
	}
	private:
	bool decide(int w, P2 p) const {
		const Wedge& e = wedges[w];
		return !(e.o1 * side(e.p1, e.p2, p) < -1e-6);
	}
};

// ---- the keypoint count of the acceptance gates, eight points at a time -------------------------------------------------
// fill_inliers_to_matchinfo asks in_polygon for every keypoint of both images of every pair that passes the first gates
// (transform_estimate.cc:191-199): 2 x ~1200 points against a hull of 10-40 vertices, 35 ns each as scalar code -- it WAS
// the acceptance epilogue (7.8 ms of the 8 ms of serial work of a 703-pair call).  count_in_polygon_v answers the same
// question for 8 points per step with GCC/clang vector types: every lane performs in_polygon()'s operations in
// in_polygon()'s order (IEEE add / multiply / divide are the same per lane as scalar; this TU is built without
// contraction), branches become selects; a lane that needs the reference's own expression (a point within 3e-7 rad of a
// vertex direction, a zero / non-finite offset) is handed to the scalar in_polygon().  The ISA-specific clones live in
// ransac_accept_simd.cc (function multiversioning); the harness compares the count with in_polygon_exact point by point.
template <int W> struct AccVec {
	typedef double vd __attribute__((vector_size(8 * W)));
	typedef long long vi __attribute__((vector_size(8 * W)));
};

// flat copy of what in_polygon() reads, one cache-friendly block per polygon
struct PolygonTables {
	const PointInPolygon* pip;
	std::vector<double> ang;                          // -1e30, the n sorted vertex angles, +1e30: wedge w lies between ang[w] and ang[w + 1]
	std::vector<double> ax, ay, ex, ey, o1;           // n + 1 wedges: p1, p2 - p1 (the two differences side() forms first), o1
	explicit PolygonTables(const PointInPolygon& P): pip(&P) {
		ang.push_back(-1e30); ang.insert(ang.end(), P.ang.begin(), P.ang.end()); ang.push_back(1e30);
		for (const auto& w : P.wedges) {
			ax.push_back(w.p1.x); ay.push_back(w.p1.y); ex.push_back(w.p2.x - w.p1.x); ey.push_back(w.p2.y - w.p1.y); o1.push_back(w.o1);
		}
	}
};

// points k = 0..n-1 at xy[k * stride], xy[k * stride + 1]; W lanes per step (8 for 512-bit registers, 4 for 256, 2 for 128)
template <int W>
inline __attribute__((always_inline)) int count_in_polygon_v(const PolygonTables& T, const double* xy, size_t stride, int n) {
	typedef typename AccVec<W>::vd vd;
	typedef typename AccVec<W>::vi vi;
#define ACC_SPLAT(v) ((vd){} + (v))
#define ACC_SEL(m, a, b) ((vd)(((vi)(a) & (m)) | ((vi)(b) & ~(m))))
#define ACC_ABS(a) ((vd)((vi)(a) & 0x7FFFFFFFFFFFFFFFLL))
	const PointInPolygon& pip = *T.pip;
	const int nv = (int)T.ang.size() - 2;
	const double* ang = T.ang.data();
	const double* wax = T.ax.data(); const double* way = T.ay.data(); const double* wex = T.ex.data(); const double* wey = T.ey.data(); const double* wo1 = T.o1.data();
	const vd comx = ACC_SPLAT(pip.com.x), comy = ACC_SPLAT(pip.com.y);
	constexpr int CH = 256;                                 // points per pass: coordinates transposed once, angles and wedge numbers handed over through memory
	alignas(64) double xs[CH], ys[CH], ts[CH];
	alignas(64) long long wedge[CH], scalar[CH];
	int count = 0;
	for (int k0 = 0; k0 < n; k0 += CH) {
		const int m = n - k0 < CH ? n - k0 : CH, mp = (m + W - 1) / W * W;
		for (int l = 0; l < m; ++l) { xs[l] = xy[(size_t)(k0 + l) * stride]; ys[l] = xy[(size_t)(k0 + l) * stride + 1]; }
		for (int l = m; l < mp; ++l) { xs[l] = 0; ys[l] = 0; }                       // padding lanes: computed, never read
		for (int b = 0; b < mp; b += W) {
			const vd px = *(const vd*)(xs + b), py = *(const vd*)(ys + b);
			const vd y = py - comy, x = px - comx;
			// fast_atan2(y, x), lane for lane
			const vd ay = ACC_ABS(y), ax = ACC_ABS(x);
			const vi gt = ax > ay;
			const vd hi = ACC_SEL(gt, ax, ay), lo = ACC_SEL(gt, ay, ax);
			vd a = lo / hi;
			const vi big = a > ACC_SPLAT(0.41421356237309503);
			a = ACC_SEL(big, (a - ACC_SPLAT(1.0)) / (a + ACC_SPLAT(1.0)), a);
			const vd base = ACC_SEL(big, ACC_SPLAT(0.78539816339744828), ACC_SPLAT(0.0));
			const vd s = a * a;
			vd p = ACC_SPLAT(0.047129968897061815);
			p = p * s + ACC_SPLAT(-0.08459108968229646); p = p * s + ACC_SPLAT(0.11041054122251703); p = p * s + ACC_SPLAT(-0.14281639256027157);
			p = p * s + ACC_SPLAT(0.19999885898543113); p = p * s + ACC_SPLAT(-0.3333333212787719); p = p * s + ACC_SPLAT(0.9999999999791288);
			vd r = base + a * p;
			r = ACC_SEL(ay > ax, ACC_SPLAT(1.5707963267948966) - r, r);
			r = ACC_SEL((vi)x < 0, ACC_SPLAT(3.141592653589793) - r, r);                // signbit(x)
			const vd t = ACC_SEL((vi)y < 0, -r, r);                                       // signbit(y)
			// wedge = number of vertex angles below t (t - ang > 0 exactly when t > ang: a difference of two doubles is never rounded to zero)
			vi below = {};
			for (int i = 1; i <= nv; ++i) below -= t > ACC_SPLAT(ang[i]);                 // a true lane is -1
			// libm decides zero / non-finite offsets (fast_atan2's first line) and whatever made t a NaN
			*(vi*)(scalar + b) = ~(hi > ACC_SPLAT(0.0)) | ~(hi < ACC_SPLAT(1.7976931348623157e308)) | (t != t);
			*(vi*)(wedge + b) = below; *(vd*)(ts + b) = t;
		}
		// per point: the nearest vertex angle is one of the wedge's two (the angles are sorted) -- within 3e-7 of t the float rounding of the
		// reference's k could matter and its own expression decides; otherwise decide(): !(o1 * side(p1, p2, p) < -1e-6) with the wedge's edge
		for (int l = 0; l < m; ++l) {
			const int w = (int)wedge[l];
			const double dl = ts[l] - ang[w], dr = ang[w + 1] - ts[l];
			if (scalar[l] || !((dl < dr ? dl : dr) > 3e-7)) { count += pip.in_polygon(P2{xs[l], ys[l]}) ? 1 : 0; continue; }
			const double sd = wex[w] * (ys[l] - way[w]) - wey[w] * (xs[l] - wax[w]);
			count += !(wo1[w] * sd < -1e-6) ? 1 : 0;
		}
	}
#undef ACC_SPLAT
#undef ACC_SEL
#undef ACC_ABS
	return count;
}
// the dispatching entry (ransac_accept_simd.cc): 512- / 256- / 128-bit clones of the loop above
int count_in_polygon(const PolygonTables& T, const double* xy, size_t stride, int n);

// ---- the refit on all inliers (transform_estimate.cc:179), rows in flight ----------------------------------------------------
// opransac::calc_transform feeds the least-squares rows to GivensLS one at a time: 2 n rows x up to NV rotations, every
// rotation a chain of square root -> divide -> multiply -> add that the next one waits for (40 cycles each: 23 us for the
// 122 inliers of an average accepted pair -- the largest piece of an accepted pair's epilogue once the keypoint count was
// vectorised).  Rotation (row r, column k) needs only (r - 1, k) and (r, k - 1): the rotations with r + k = t are independent
// of each other.  calc_transform_skewed walks the (row, column) grid by anti-diagonals, so up to NV chains are in flight in
// the core's out-of-order window instead of one.  Every rotation reads and writes exactly the values it reads and writes in
// the row-by-row order: the triangle, and the homography, are bit-identical (the harness compares them).  Host only.
template <int NV>
inline void givens_rows_skewed(opransac::GivensLS<NV>& ls, double* rows, int nrows) {     // rows: nrows x (NV + 1), [a | beta]
	constexpr int S = NV + 1;
	for (int t = 0; t < nrows + NV - 1; ++t) {
		const int k_lo = t - (nrows - 1) > 0 ? t - (nrows - 1) : 0, k_hi = t < NV - 1 ? t : NV - 1;
		for (int k = k_lo; k <= k_hi; ++k) {
			double* a = rows + (size_t)(t - k) * S;
			const double ak = a[k];
			if (ak == 0.0) continue;
			const double rkk = ls.R[k][k];
			const double r = std::sqrt(rkk * rkk + ak * ak);
			const double c = rkk / r, sn = ak / r;
			ls.R[k][k] = r;
			for (int j = k + 1; j < NV; ++j) {
				const double tt = c * ls.R[k][j] + sn * a[j];
				a[j] = c * a[j] - sn * ls.R[k][j];
				ls.R[k][j] = tt;
			}
			const double tt = c * ls.qtb[k] + sn * a[NV];
			a[NV] = c * a[NV] - sn * ls.qtb[k];
			ls.qtb[k] = tt;
		}
	}
}

template <typename Get1, typename Get2>
inline void calc_transform_skewed(int n, Get1 get1, Get2 get2, bool affine, double (&H)[9]) {
	if (n < 16) { opransac::calc_transform(n, get1, get2, affine, H); return; }
	const double s1 = opransac::norm_scale(n, get1), s2 = opransac::norm_scale(n, get2);
	double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
	if (!affine) {
		std::vector<double> rows((size_t)2 * n * 9);
		for (int i = 0; i < n; ++i) {                        // lib/imgproc.cc:267-274: rows 0..n-1 the x equations, n..2n-1 the y equations
			P2 m0 = get1(i), m1 = get2(i);
			m0.x *= s1; m0.y *= s1; m1.x *= s2; m1.y *= s2;
			const double rx[9] = {m1.x, m1.y, 1, 0, 0, 0, -m1.x * m0.x, -m1.y * m0.x, m0.x};
			const double ry[9] = {0, 0, 0, m1.x, m1.y, 1, -m1.x * m0.y, -m1.y * m0.y, m0.y};
			std::memcpy(&rows[(size_t)i * 9], rx, sizeof(rx)); std::memcpy(&rows[(size_t)(n + i) * 9], ry, sizeof(ry));
		}
		opransac::GivensLS<8> ls; ls.reset();
		givens_rows_skewed<8>(ls, rows.data(), 2 * n);
		double x[8];
		ls.solve(x);
		for (int i = 0; i < 8; ++i) h[i] = x[i];
	} else {
		std::vector<double> rows((size_t)2 * n * 7);
		for (int i = 0; i < n; ++i) {                        // lib/imgproc.cc:304-310: rows interleaved x, y per point
			P2 m0 = get1(i), m1 = get2(i);
			m0.x *= s1; m0.y *= s1; m1.x *= s2; m1.y *= s2;
			const double r0[7] = {m1.x, m1.y, 1, 0, 0, 0, m0.x}, r1[7] = {0, 0, 0, m1.x, m1.y, 1, m0.y};
			std::memcpy(&rows[(size_t)(2 * i) * 7], r0, sizeof(r0)); std::memcpy(&rows[(size_t)(2 * i + 1) * 7], r1, sizeof(r1));
		}
		opransac::GivensLS<6> ls; ls.reset();
		givens_rows_skewed<6>(ls, rows.data(), 2 * n);
		double x[6];
		ls.solve(x);
		for (int i = 0; i < 6; ++i) h[i] = x[i];
	}
	const double i1 = 1.0 / s1;                                // t1.inverse() * H * t2 with t = diag(s, s, 1) (transform_estimate.cc:121-128)
	const double l[3] = {i1, i1, 1.0}, r[3] = {s2, s2, 1.0};
	for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) H[a * 3 + b] = (l[a] * h[a * 3 + b]) * r[b];
}

inline P2 trans2d(const double (&H)[9], P2 m) {       // homography.hh:53-76
	const double x = H[0] * m.x + H[1] * m.y + H[2] * 1.0, y = H[3] * m.x + H[4] * m.y + H[5] * 1.0, z = H[6] * m.x + H[7] * m.y + H[8] * 1.0;
	const double d = 1.0 / z;
	return P2{x * d, y * d};
}

// 3x3 inverse with complete pivoting (Homography::inverse, stitch/homography.cc:25-39)
inline bool inverse3(const double (&a)[9], double (&inv)[9]) {
	double lu[9]; std::memcpy(lu, a, sizeof(lu));
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
	const double thr = std::fabs(maxpivot) * (2.220446049250313e-16 * 3);
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

// overlap_region (stitch/homography.cc:50-90): homo maps shape2 -> shape1, inv the reverse
inline std::vector<P2> overlap_region(const Shape& shape1, const Shape& shape2, const double (&homo)[9], const double (&inv)[9]) {
	const int NR = 100;
	const float stepw = (float)(shape2.w * 1.0 / NR), steph = (float)(shape2.h * 1.0 / NR);
	const double hw = shape2.w * 0.5, hh = shape2.h * 0.5;
	std::vector<P2> pts2in1;
	for (int i = 0; i < NR; ++i) {
		const P2 e[4] = { P2{-hw + i * stepw, -hh}, P2{-hw + i * stepw, hh}, P2{-hw, -hh + i * steph}, P2{hw, -hh + i * steph} };
		for (int k = 0; k < 4; ++k) {
			// Matrix product 3x3 * 3x(4 NR) then float denom = 1.0 / z (:72-76)
			const double x = homo[0] * e[k].x + homo[1] * e[k].y + homo[2] * 1.0;
			const double y = homo[3] * e[k].x + homo[4] * e[k].y + homo[5] * 1.0;
			const double z = homo[6] * e[k].x + homo[7] * e[k].y + homo[8] * 1.0;
			const float denom = (float)(1.0 / z);
			const P2 pin1{x * denom, y * denom};
			if (shifted_in(shape1, pin1)) pts2in1.push_back(pin1);
		}
	}
	const P2 corners[4] = { P2{-shape1.w * 0.5, -shape1.h * 0.5}, P2{shape1.w * 0.5, -shape1.h * 0.5}, P2{-shape1.w * 0.5, shape1.h * 0.5}, P2{shape1.w * 0.5, shape1.h * 0.5} };
	for (auto& c : corners) if (shifted_in(shape2, trans2d(inv, c))) pts2in1.push_back(c);
	return convex_hull(pts2in1);
}

}	// namespace opaccept
