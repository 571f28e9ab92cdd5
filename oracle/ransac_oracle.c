/* oracle/ransac_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of TransformEstimation (stitch/transform_estimate.cc:26-218) with its helpers:
 * getPerspectiveTransform / getAffineTransform (lib/imgproc.cc:251-317), Homography::health /
 * inverse / trans2d (stitch/homography.hh:53-131, homography.cc:25-39), overlap_region
 * (homography.cc:50-90), convex_hull / polygon_area / PointInPolygon (lib/polygon.cc:17-82,
 * lib/polygon.hh:30-52), std::mt19937 sampling (:64-77).
 *
 * Unpinned boundary: the reference solves the DLT system with Eigen::JacobiSVD(...).solve(b)
 * (system Eigen 3, unpinned version, absent from /root/reference and from this image).  The
 * least-squares solution is restated with a backward-stable Givens QR; against oracle/_ref
 * (which compiles the reference's own code over the mini-Eigen stand-in) homographies agree to
 * ~1e-9 relative and inlier sets are equal except for points within that distance of the
 * threshold (tests/test_ransac_vs_ref.py).  The RNG is injected: the reference draws its seed
 * from std::random_device (transform_estimate.cc:64), here it is an argument.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "oracle.h"

typedef struct { double x, y; } p2;

/* ---- std::mt19937 ---- */
typedef struct { unsigned mt[624]; int idx; } mt19937;
static void mt_seed(mt19937* g, unsigned s) {
	g->mt[0] = s;
	for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (unsigned)i;
	g->idx = 624;
}
static unsigned mt_next(mt19937* g) {
	if (g->idx >= 624) {
		for (int i = 0; i < 624; ++i) {
			unsigned y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
			g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
		}
		g->idx = 0;
	}
	unsigned y = g->mt[g->idx++];
	y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
	return y;
}

/* ---- least squares by Givens rotations, one row at a time ---- */
static void ls_add_row(int nv, double* R /* nv x nv */, double* qtb, double* a, double beta) {
	for (int k = 0; k < nv; ++k) {
		const double ak = a[k];
		if (ak == 0.0) continue;
		const double rkk = R[k * nv + k];
		const double r = sqrt(rkk * rkk + ak * ak);
		const double c = rkk / r, s = ak / r;
		R[k * nv + k] = r;
		for (int j = k + 1; j < nv; ++j) {
			const double t = c * R[k * nv + j] + s * a[j];
			a[j] = c * a[j] - s * R[k * nv + j];
			R[k * nv + j] = t;
		}
		const double t = c * qtb[k] + s * beta;
		beta = c * beta - s * qtb[k];
		qtb[k] = t;
	}
}
static void ls_solve(int nv, const double* R, const double* qtb, double* x) {
	double dmax = 0;
	for (int k = 0; k < nv; ++k) { const double d = fabs(R[k * nv + k]); dmax = d > dmax ? d : dmax; }
	const double tiny = dmax * 1e-13;
	for (int k = nv - 1; k >= 0; --k) {
		double acc = qtb[k];
		for (int j = k + 1; j < nv; ++j) acc -= R[k * nv + j] * x[j];
		x[k] = fabs(R[k * nv + k]) > tiny ? acc / R[k * nv + k] : 0.0;
	}
}

static double norm_scale(int n, const p2* pts) {	/* transform_estimate.cc:99-114 */
	const double sizeinv = 1.0 / n;
	double sqrsum = 0;
	for (int i = 0; i < n; ++i) sqrsum += (pts[i].x * pts[i].x + pts[i].y * pts[i].y) * sizeinv;
	return sqrt(2.0 / sqrsum);
}

/* calc_transform (:89-130): p1[i] <- H p2[i] */
static void calc_transform(int n, const p2* p1, const p2* p2_, int affine, double H[9]) {
	const double s1 = norm_scale(n, p1), s2 = norm_scale(n, p2_);
	double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
	double R[64], qtb[8], x[8];
	memset(R, 0, sizeof(R)); memset(qtb, 0, sizeof(qtb));
	if (!affine) {
		for (int i = 0; i < n; ++i) {	/* lib/imgproc.cc:267-274 */
			const double m0x = p1[i].x * s1, m1x = p2_[i].x * s2, m1y = p2_[i].y * s2;
			double row[8] = {m1x, m1y, 1, 0, 0, 0, -m1x * m0x, -m1y * m0x};
			ls_add_row(8, R, qtb, row, m0x);
		}
		for (int i = 0; i < n; ++i) {
			const double m0y = p1[i].y * s1, m1x = p2_[i].x * s2, m1y = p2_[i].y * s2;
			double row[8] = {0, 0, 0, m1x, m1y, 1, -m1x * m0y, -m1y * m0y};
			ls_add_row(8, R, qtb, row, m0y);
		}
		ls_solve(8, R, qtb, x);
		for (int i = 0; i < 8; ++i) h[i] = x[i];
	} else {
		for (int i = 0; i < n; ++i) {	/* lib/imgproc.cc:304-310 */
			const double m0x = p1[i].x * s1, m0y = p1[i].y * s1, m1x = p2_[i].x * s2, m1y = p2_[i].y * s2;
			double r0[6] = {m1x, m1y, 1, 0, 0, 0};
			ls_add_row(6, R, qtb, r0, m0x);
			double r1[6] = {0, 0, 0, m1x, m1y, 1};
			ls_add_row(6, R, qtb, r1, m0y);
		}
		ls_solve(6, R, qtb, x);
		for (int i = 0; i < 6; ++i) h[i] = x[i];
	}
	const double i1 = 1.0 / s1;
	const double l[3] = {i1, i1, 1.0}, r[3] = {s2, s2, 1.0};
	for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) H[a * 3 + b] = (l[a] * h[a * 3 + b]) * r[b];
}

static int health(const double* m) {	/* homography.hh:106-127 */
	const double lim = (double)2e-3f;
	if (fabs(m[6]) > lim) return 0;
	if (fabs(m[7]) > lim) return 0;
	const double x0y = m[5], x1x = m[1] + m[2], x1y = m[4] + m[5];
	if (x1y <= x0y) return 0;
	const double x2x = m[0] + m[1] + m[2];
	if (x2x <= x1x) return 0;
	return 1;
}

static int is_inlier(const double* H, p2 q1, p2 q2, double inlier_dist) {	/* :138-146 */
	const double tx = q2.x * H[0] + q2.y * H[1] + 1.0 * H[2];
	const double ty = q2.x * H[3] + q2.y * H[4] + 1.0 * H[5];
	const double tz = q2.x * H[6] + q2.y * H[7] + 1.0 * H[8];
	const double idenom = 1.0 / tz;
	const double dx = tx * idenom - q1.x, dy = ty * idenom - q1.y;
	return dx * dx + dy * dy < inlier_dist;
}

/* ---- polygons ---- */
static double side(p2 a, p2 b, p2 p) { return (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x); }
static int cmp_yx(const void* a_, const void* b_) {
	const p2* a = (const p2*)a_; const p2* b = (const p2*)b_;
	if (a->y == b->y) return a->x < b->x ? -1 : (a->x > b->x ? 1 : 0);
	return a->y < b->y ? -1 : 1;
}
static int convex_hull(p2* pts, int n, p2* ret) {	/* lib/polygon.cc:17-46 */
	if (n <= 3) { memcpy(ret, pts, sizeof(p2) * n); return n; }
	qsort(pts, n, sizeof(p2), cmp_yx);
	int sz = 0;
	ret[sz++] = pts[0]; ret[sz++] = pts[1];
	for (int i = 2; i < n; ++i) {
		while (sz >= 2 && side(ret[sz - 2], ret[sz - 1], pts[i]) <= 0) sz--;
		ret[sz++] = pts[i];
	}
	const int mid = sz;
	ret[sz++] = pts[n - 2];
	for (int i = n - 3; i >= 0; --i) {
		while (sz > mid && side(ret[sz - 2], ret[sz - 1], pts[i]) <= 0) sz--;
		ret[sz++] = pts[i];
	}
	return sz;
}
static double polygon_area(const p2* poly, int n) {
	double sum = 0;
	for (int i = 0; i < n; ++i) sum += poly[i].x * (poly[(i + 1) % n].y - poly[(i + n - 1) % n].y);
	return 0.5 * fabs(sum);
}
typedef struct { float k; int i; } slope_t;
static int cmp_slope(const void* a_, const void* b_) {
	const slope_t* a = (const slope_t*)a_; const slope_t* b = (const slope_t*)b_;
	if (a->k != b->k) return a->k < b->k ? -1 : 1;
	return a->i < b->i ? -1 : (a->i > b->i ? 1 : 0);
}
typedef struct { const p2* poly; int n; p2 com; slope_t* slopes; } pip_t;
static void pip_init(pip_t* P, const p2* poly, int n) {
	P->poly = poly; P->n = n; P->com.x = P->com.y = 0;
	for (int i = 0; i < n; ++i) { P->com.x += poly[i].x; P->com.y += poly[i].y; }
	const double f = 1.0 / n;
	P->com.x *= f; P->com.y *= f;
	P->slopes = (slope_t*)malloc(sizeof(slope_t) * n);
	for (int i = 0; i < n; ++i) { P->slopes[i].k = (float)atan2(poly[i].y - P->com.y, poly[i].x - P->com.x); P->slopes[i].i = i; }
	qsort(P->slopes, n, sizeof(slope_t), cmp_slope);
}
static int pip_in(const pip_t* P, p2 p) {	/* lib/polygon.cc:62-82 */
	const float k = (float)atan2(p.y - P->com.y, p.x - P->com.x);
	/* lower_bound of (k, 0) in lexicographic (float, int) order */
	int lo = 0, hi = P->n;
	while (lo < hi) {
		int mid = (lo + hi) / 2;
		const slope_t* s = &P->slopes[mid];
		int less = (s->k < k) || (s->k == k && s->i < 0);
		if (less) lo = mid + 1; else hi = mid;
	}
	int idx1, idx2;
	if (lo == P->n) { idx1 = P->slopes[P->n - 1].i; idx2 = P->slopes[0].i; }
	else { idx2 = P->slopes[lo].i; idx1 = lo != 0 ? P->slopes[lo - 1].i : P->slopes[P->n - 1].i; }
	const p2 a = P->poly[idx1], b = P->poly[idx2];
	const double o1 = side(a, b, P->com), o2 = side(a, b, p);
	return !(o1 * o2 < -1e-6);
}

static p2 trans2d(const double* H, p2 m) {
	const double x = H[0] * m.x + H[1] * m.y + H[2] * 1.0, y = H[3] * m.x + H[4] * m.y + H[5] * 1.0, z = H[6] * m.x + H[7] * m.y + H[8] * 1.0;
	const double d = 1.0 / z;
	p2 r = {x * d, y * d};
	return r;
}
static int shifted_in(int w, int h, p2 p) { return p.x >= -w * 0.5 && p.x < w * 0.5 && p.y >= -h * 0.5 && p.y < h * 0.5; }

/* complete-pivoting 3x3 inverse, as in sift_oracle.c (Homography::inverse, homography.cc:25-39) */
static int inverse3(const double a[9], double inv[9]) {
	double lu[9]; memcpy(lu, a, sizeof(lu));
	int rowt[3], colt[3], nonzero = 3; double maxpivot = 0;
	for (int k = 0; k < 3; ++k) {
		int br = k, bc = k; double best = -1;
		for (int i = k; i < 3; ++i) for (int j = k; j < 3; ++j) { double v = fabs(lu[i * 3 + j]); if (v > best) { best = v; br = i; bc = j; } }
		if (best == 0.0) { nonzero = k; for (int i = k; i < 3; ++i) rowt[i] = colt[i] = i; break; }
		if (best > maxpivot) maxpivot = best;
		rowt[k] = br; colt[k] = bc;
		if (br != k) for (int j = 0; j < 3; ++j) { double t = lu[k * 3 + j]; lu[k * 3 + j] = lu[br * 3 + j]; lu[br * 3 + j] = t; }
		if (bc != k) for (int i = 0; i < 3; ++i) { double t = lu[i * 3 + k]; lu[i * 3 + k] = lu[i * 3 + bc]; lu[i * 3 + bc] = t; }
		for (int i = k + 1; i < 3; ++i) lu[i * 3 + k] /= lu[k * 3 + k];
		for (int i = k + 1; i < 3; ++i) for (int j = k + 1; j < 3; ++j) lu[i * 3 + j] -= lu[i * 3 + k] * lu[k * 3 + j];
	}
	double thr = fabs(maxpivot) * (DBL_EPSILON * 3);
	int rank = 0;
	for (int i = 0; i < nonzero; ++i) rank += (fabs(lu[i * 3 + i]) > thr);
	if (rank != 3) return 0;
	for (int col = 0; col < 3; ++col) {
		double c[3];
		for (int i = 0; i < 3; ++i) c[i] = (i == col) ? 1.0 : 0.0;
		for (int i = 0; i < 3; ++i) { double t = c[i]; c[i] = c[rowt[i]]; c[rowt[i]] = t; }
		for (int i = 0; i < 3; ++i) for (int j = 0; j < i; ++j) c[i] -= lu[i * 3 + j] * c[j];
		for (int i = 2; i >= 0; --i) { for (int j = i + 1; j < 3; ++j) c[i] -= lu[i * 3 + j] * c[j]; c[i] /= lu[i * 3 + i]; }
		for (int i = 2; i >= 0; --i) { double t = c[i]; c[i] = c[colt[i]]; c[colt[i]] = t; }
		for (int i = 0; i < 3; ++i) inv[i * 3 + col] = c[i];
	}
	return 1;
}

/* overlap_region (stitch/homography.cc:50-90); returns hull size, hull in out (>= 404 entries) */
static int overlap_region(int w1, int h1, int w2, int h2, const double* homo, const double* inv, p2* out) {
	const int NR = 100;
	const float stepw = (float)(w2 * 1.0 / NR), steph = (float)(h2 * 1.0 / NR);
	const double hw = w2 * 0.5, hh = h2 * 0.5;
	p2 pts[4 * 100 + 4]; int np = 0;
	for (int i = 0; i < NR; ++i) {
		p2 e[4];
		e[0].x = -hw + i * stepw; e[0].y = -hh;
		e[1].x = -hw + i * stepw; e[1].y = hh;
		e[2].x = -hw; e[2].y = -hh + i * steph;
		e[3].x = hw; e[3].y = -hh + i * steph;
		for (int k = 0; k < 4; ++k) {
			const double x = homo[0] * e[k].x + homo[1] * e[k].y + homo[2] * 1.0;
			const double y = homo[3] * e[k].x + homo[4] * e[k].y + homo[5] * 1.0;
			const double z = homo[6] * e[k].x + homo[7] * e[k].y + homo[8] * 1.0;
			const float denom = (float)(1.0 / z);
			p2 pin1 = {x * denom, y * denom};
			if (shifted_in(w1, h1, pin1)) pts[np++] = pin1;
		}
	}
	const p2 corners[4] = {{-w1 * 0.5, -h1 * 0.5}, {w1 * 0.5, -h1 * 0.5}, {-w1 * 0.5, h1 * 0.5}, {w1 * 0.5, h1 * 0.5}};
	for (int k = 0; k < 4; ++k) if (shifted_in(w2, h2, trans2d(inv, corners[k]))) pts[np++] = corners[k];
	return convex_hull(pts, np, out);
}

static int count_in(const p2* poly, int n, const p2* pts, int npts, int* valid) {
	*valid = n >= 3;
	if (n < 3) return 0;
	pip_t P; pip_init(&P, poly, n);
	int c = 0;
	for (int i = 0; i < npts; ++i) c += pip_in(&P, pts[i]);
	free(P.slopes);
	return c;
}

/* TransformEstimation(match, kp1, kp2, shape1, shape2).get_transform(info) with an injected seed.
 * match: m x (first, second); kp1/kp2: centred keypoints (x, y); affine = CYLINDER || TRANS.
 * Outputs: *confidence, homo[9], inliers (<= m ints) + *n_inliers, *best_hyp/*best_count.
 * Returns get_transform()'s bool. */
int orc_ransac(const int* match, int m, const double* kp1, int nk1, const double* kp2, int nk2,
		int w1, int h1, int w2, int h2, int affine, int iterations, double ransac_inlier_thres_cfg,
		float inlier_in_match_ratio, float inlier_in_points_ratio, unsigned seed,
		float* confidence, double* homo_out, int* inliers, int* n_inliers, int* best_hyp, int* best_count) {
	*confidence = 0; *n_inliers = 0; *best_hyp = -1; *best_count = -1;
	const int nused = (affine ? 6 : 8) / 2 + 4;
	if (m < nused || m < 8) return 0;	/* :55 and ESTIMATE_MIN_NR_MATCH (:21,39) */
	p2* q1 = (p2*)malloc(sizeof(p2) * m); p2* q2 = (p2*)malloc(sizeof(p2) * m);
	for (int i = 0; i < m; ++i) {
		q1[i].x = kp1[2 * match[2 * i]]; q1[i].y = kp1[2 * match[2 * i] + 1];
		q2[i].x = kp2[2 * match[2 * i + 1]]; q2[i].y = kp2[2 * match[2 * i + 1] + 1];
	}
	const float thres = (float)((w1 + h1) * 0.5 / 800 * ransac_inlier_thres_cfg);	/* :46 */
	const double inlier_dist = (double)(thres * thres);				/* :133 */
	mt19937 rng; mt_seed(&rng, seed);
	double best[9]; int maxcnt = -1, have = 0;
	for (int K = 0; K < iterations; ++K) {
		int sel[8]; p2 s1[8], s2[8];
		for (int t = 0; t < nused; ++t) {
			int r, dup;
			do {
				r = (int)(mt_next(&rng) % (unsigned)m);
				dup = 0;
				for (int u = 0; u < t; ++u) dup |= (sel[u] == r);
			} while (dup);
			sel[t] = r; s1[t] = q1[r]; s2[t] = q2[r];
		}
		double H[9];
		calc_transform(nused, s1, s2, affine, H);
		if (!health(H)) continue;
		int cnt = 0;
		for (int i = 0; i < m; ++i) cnt += is_inlier(H, q1[i], q2[i], inlier_dist);
		if (maxcnt < cnt) { maxcnt = cnt; memcpy(best, H, sizeof(best)); *best_hyp = K; have = 1; }
	}
	*best_count = maxcnt;
	int ok = 0;
	if (have) {
		int ni = 0;
		for (int i = 0; i < m; ++i) if (is_inlier(best, q1[i], q2[i], inlier_dist)) inliers[ni++] = i;
		*n_inliers = ni;
		*confidence = -(float)ni;	/* :153 */
		if (ni >= 8) {
			p2* a1 = (p2*)malloc(sizeof(p2) * ni); p2* a2 = (p2*)malloc(sizeof(p2) * ni);
			for (int i = 0; i < ni; ++i) { a1[i] = q1[inliers[i]]; a2[i] = q2[inliers[i]]; }
			double homo[9], inv[9];
			calc_transform(ni, a1, a2, affine, homo);
			free(a1); free(a2);
			if (inverse3(homo, inv)) {
				p2 hull[420]; int valid;
				p2* k1 = (p2*)malloc(sizeof(p2) * (nk1 > 0 ? nk1 : 1)); p2* k2 = (p2*)malloc(sizeof(p2) * (nk2 > 0 ? nk2 : 1));
				for (int i = 0; i < nk1; ++i) { k1[i].x = kp1[2 * i]; k1[i].y = kp1[2 * i + 1]; }
				for (int i = 0; i < nk2; ++i) { k2[i].x = kp2[2 * i]; k2[i].y = kp2[2 * i + 1]; }
				do {
					int nh = overlap_region(w1, h1, w2, h2, homo, inv, hull);
					float r1m = ni * 1.0f / count_in(hull, nh, q1, m, &valid);
					if (r1m < inlier_in_match_ratio) break;
					float r1p = ni * 1.0f / count_in(hull, nh, k1, nk1, &valid);
					if (!valid || r1p < 0.01 || r1p > 1) break;
					nh = overlap_region(w2, h2, w1, h1, inv, homo, hull);
					float r2m = ni * 1.0f / count_in(hull, nh, q2, m, &valid);
					if (r2m < inlier_in_match_ratio) break;
					float r2p = ni * 1.0f / count_in(hull, nh, k2, nk2, &valid);
					if (!valid || r2p < 0.01 || r2p > 1) break;
					*confidence = (float)((r1p + r2p) * 0.5);
					if (*confidence < inlier_in_points_ratio) break;
					double area = polygon_area(hull, nh);
					double area1 = (double)(w1 * h1), area2 = (double)(w2 * h2);
					if (area / (area1 > area2 ? area1 : area2) < 0.15) break;
					memcpy(homo_out, homo, sizeof(homo));
					ok = 1;
				} while (0);
				free(k1); free(k2);
			}
		}
	}
	free(q1); free(q2);
	return ok;
}
