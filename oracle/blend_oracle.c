/* oracle/blend_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of the reference's rendering path, in the reference's floating-point
 * types and evaluation order (compiled -ffp-contract=off like oracle/_ref):
 *   ConnectedImages::{calc_inverse_homo, update_proj_range, get_final_resolution, blend}
 *                                 stitch/stitcher_image.cc:36-155, stitch/projection.hh:14-71
 *   LinearBlender::run            stitch/blender.cc:24-96 (both LAZY_READ branches, 1 thread)
 *   MultiBandBlender::run         stitch/multiband.cc:19-151 (1 thread: image index order)
 *   interpolate                   lib/imgproc.cc:135-156
 *   GaussianBlur::blur<T>         feature/gaussian.hh:30-91 on WeightedPixel (multiband.hh:13-24)
 *   CylinderProject::project      stitch/warp.cc:13-75
 * Citations are file:line under /root/reference/src.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "oracle.h"

#define ORC_EPS 1e-6		/* lib/utils.hh:22 */

/* ---- Eigen::FullPivLU 3x3 inverse (Homography::inverse, stitch/homography.cc:25-39);
 * same restatement as sift_oracle.c / ransac_oracle.c ---- */
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

/* Homography::trans (stitch/homography.hh:53-58) */
static void htrans(const double* d, double x, double y, double z, double out[3]) {
	out[0] = d[0] * x + d[1] * y + d[2] * z;
	out[1] = d[3] * x + d[4] * y + d[5] * z;
	out[2] = d[6] * x + d[7] * y + d[8] * z;
}

/* stitch/projection.hh:16-18,33-36,48-51 */
static void homo2proj(int method, const double h[3], double out[2]) {
	if (method == 0) { out[0] = h[0] / h[2]; out[1] = h[1] / h[2]; }
	else if (method == 1) { out[0] = atan2(h[0], h[2]); out[1] = h[1] / (hypot(h[0], h[2])); }
	else { out[0] = atan2(h[0], h[2]); out[1] = atan2(h[1], hypot(h[0], h[2])); }
}
/* stitch/projection.hh:29-31,38-40,66-68 */
static void proj2homo(int method, double x, double y, double out[3]) {
	if (method == 0) { out[0] = x; out[1] = y; out[2] = 1; }
	else if (method == 1) { out[0] = sin(x); out[1] = y; out[2] = cos(x); }
	else { out[0] = sin(x); out[1] = tan(y); out[2] = cos(x); }
}

/* calc_inverse_homo + update_proj_range + get_final_resolution (stitcher_image.cc:36-114) */
int orc_blend_prepare(int proj_method, int identity_idx, int n, const int* shapes_wh, const double* homo,
		int max_output_size, orc_blend_geom* g, double* homo_inv, double* ranges) {
	if (n <= 0 || identity_idx < 0 || identity_idx >= n) return -1;
	for (int i = 0; i < n; ++i)
		if (!inverse3(homo + 9 * i, homo_inv + 9 * i)) return -2;	/* m_assert(lu.isInvertible()) */
	enum { CORNER_SAMPLE = 100 };
	double cx[4 * CORNER_SAMPLE], cy[4 * CORNER_SAMPLE]; int nc = 0;
	for (int i = 0; i < CORNER_SAMPLE; ++i) {		/* :44-48 */
		double xi = (double)i / CORNER_SAMPLE - 0.5;
		cx[nc] = xi; cy[nc++] = -0.5;
		cx[nc] = xi; cy[nc++] = 0.5;
	}
	for (int j = 0; j < CORNER_SAMPLE; ++j) {		/* :49-53 */
		double yj = (double)j / CORNER_SAMPLE - 0.5;
		cx[nc] = -0.5; cy[nc++] = yj;
		cx[nc] = 0.5; cy[nc++] = yj;
	}
	double pmin[2] = {DBL_MAX, DBL_MAX}, pmax[2] = {DBL_MAX * (-1), DBL_MAX * (-1)};
	for (int m = 0; m < n; ++m) {
		const int w = shapes_wh[2 * m], h = shapes_wh[2 * m + 1];
		double nmin[2] = {DBL_MAX, DBL_MAX}, nmax[2] = {DBL_MAX * (-1), DBL_MAX * (-1)};
		for (int k = 0; k < nc; ++k) {
			double hv[3], t[2];
			htrans(homo + 9 * m, cx[k] * w, cy[k] * h, 1, hv);
			homo2proj(proj_method, hv, t);
			if (t[0] < nmin[0]) nmin[0] = t[0];
			if (t[1] < nmin[1]) nmin[1] = t[1];
			if (t[0] > nmax[0]) nmax[0] = t[0];
			if (t[1] > nmax[1]) nmax[1] = t[1];
		}
		ranges[4 * m] = nmin[0]; ranges[4 * m + 1] = nmin[1]; ranges[4 * m + 2] = nmax[0]; ranges[4 * m + 3] = nmax[1];
		for (int c = 0; c < 2; ++c) { if (nmin[c] < pmin[c]) pmin[c] = nmin[c]; if (nmax[c] > pmax[c]) pmax[c] = nmax[c]; }
	}
	g->proj_method = proj_method;
	g->proj_min[0] = pmin[0]; g->proj_min[1] = pmin[1]; g->proj_max[0] = pmax[0]; g->proj_max[1] = pmax[1];
	/* get_final_resolution (:79-114) */
	const int refw = shapes_wh[2 * identity_idx], refh = shapes_wh[2 * identity_idx + 1];
	double c2[3], c1[3], p2[2], p1[2];
	htrans(homo + 9 * identity_idx, refw / 2.0, refh / 2.0, 1, c2);
	htrans(homo + 9 * identity_idx, -refw / 2.0, -refh / 2.0, 1, c1);
	homo2proj(proj_method, c2, p2); homo2proj(proj_method, c1, p1);
	double rx = p2[0] - p1[0], ry = p2[1] - p1[1];
	if (proj_method != 0) {
		if (rx < 0) rx = 2 * M_PI + rx;
		if (ry < 0) ry = M_PI + ry;
	}
	double resx = fabs(rx) / (double)refw, resy = fabs(ry) / (double)refh;
	const double tsx = (pmax[0] - pmin[0]) / resx, tsy = (pmax[1] - pmin[1]) / resy;
	const double max_edge = tsx > tsy ? tsx : tsy;
	if (max_edge > 80000 || tsx * tsy > 1e9) return -3;		/* error_exit("Target size too large") */
	if (max_edge > max_output_size) {
		float ratio = (float)(max_edge / max_output_size);
		resx *= ratio; resy *= ratio;
	}
	g->resolution[0] = resx; g->resolution[1] = resy;
	return 0;
}

/* Coor(double, double): truncation toward zero (stitcher_image.cc:123,126-129) */
static void roi_of(const orc_blend_geom* g, const double* range, int roi[4]) {
	roi[0] = (int)((range[0] - g->proj_min[0]) / g->resolution[0]);
	roi[1] = (int)((range[1] - g->proj_min[1]) / g->resolution[1]);
	roi[2] = (int)((range[2] - g->proj_min[0]) / g->resolution[0]);
	roi[3] = (int)((range[3] - g->proj_min[1]) / g->resolution[1]);
}

int orc_blend_dims(const orc_blend_geom* g, const orc_blend_image* imgs, int n, int* h, int* w) {
	int tx = 0, ty = 0;		/* Coor target_size{0,0}; update_max(bottom_right) (blender.cc:21) */
	for (int i = 0; i < n; ++i) {
		int roi[4]; roi_of(g, imgs[i].range, roi);
		if (roi[2] > tx) tx = roi[2];
		if (roi[3] > ty) ty = roi[3];
	}
	*h = ty; *w = tx;
	return 0;
}

/* the lambda of ConnectedImages::blend (stitcher_image.cc:143-151) */
static void canvas_to_image(const orc_blend_geom* g, const orc_blend_image* im, int tx, int ty, double out[2]) {
	const double cx = (double)tx * g->resolution[0] + g->proj_min[0];
	const double cy = (double)ty * g->resolution[1] + g->proj_min[1];
	double hv[3], ret[3];
	proj2homo(g->proj_method, cx, cy, hv);
	htrans(im->homo_inv, hv[0], hv[1], hv[2], ret);
	if (ret[2] < 0) { out[0] = -10; out[1] = -10; return; }
	const double denom = 1.0 / ret[2];
	out[0] = ret[0] * denom + im->w * 0.5;
	out[1] = ret[1] * denom + im->h * 0.5;
}

/* interpolate (lib/imgproc.cc:135-156); returns 0 for Color::NO */
static int interpolate(const float* img, int rows, int cols, float r, float c, float out[3]) {
	int fr = (int)floor(r), fc = (int)floor(c);
	if (fr < 0 || fc < 0 || fc + 1 >= cols || fr + 1 >= rows) return 0;
	float ret[3] = {0, 0, 0};
	r -= fr; c -= fc;
	const float* p = img + ((size_t)fr * cols + fc) * 3;
	float wt;
	if (*p < 0) return 0;
	wt = (1 - r) * (1 - c); ret[0] += p[0] * wt; ret[1] += p[1] * wt; ret[2] += p[2] * wt;
	p = img + ((size_t)(fr + 1) * cols + fc) * 3;
	if (*p < 0) return 0;
	wt = r * (1 - c); ret[0] += p[0] * wt; ret[1] += p[1] * wt; ret[2] += p[2] * wt;
	p = img + ((size_t)(fr + 1) * cols + fc + 1) * 3;
	if (*p < 0) return 0;
	wt = r * c; ret[0] += p[0] * wt; ret[1] += p[1] * wt; ret[2] += p[2] * wt;
	p = img + ((size_t)fr * cols + fc + 1) * 3;
	if (*p < 0) return 0;
	wt = (1 - r) * c; ret[0] += p[0] * wt; ret[1] += p[1] * wt; ret[2] += p[2] * wt;
	out[0] = ret[0]; out[1] = ret[1]; out[2] = ret[2];
	return 1;
}

/* GET_COLOR_AND_W (blender.cc:26-36); returns 0 for "continue" */
static int color_and_w(const orc_blend_geom* g, const orc_blend_image* im, int i, int j, int ordered_input,
		float color[3], float* w_out) {
	double ic[2];
	canvas_to_image(g, im, j, i, ic);
	/* ImageToAdd::map_coor (blender.hh:39-44) */
	if (ic[0] < 0 || ic[0] >= im->w || ic[1] < 0 || ic[1] >= im->h) return 0;
	float r = (float)ic[1], c = (float)ic[0];
	if (!interpolate(im->data, im->h, im->w, r, c, color)) return 0;
	if (color[0] < 0) return 0;
	float w = (float)(0.5 - fabs(c / im->w - 0.5));
	if (!ordered_input) w = (float)(w * (0.5 - fabs(r / im->h - 0.5)));
	color[0] *= w; color[1] *= w; color[2] *= w;
	*w_out = w;
	return 1;
}

int orc_blend_linear(const orc_blend_geom* g, const orc_blend_image* imgs, int n, int ordered_input, int lazy_read, float* out) {
	int H, W; orc_blend_dims(g, imgs, n, &H, &W);
	int* roi = (int*)malloc(sizeof(int) * 4 * n);
	for (int k = 0; k < n; ++k) roi_of(g, imgs[k].range, roi + 4 * k);
	if (lazy_read) {		/* blender.cc:38-76, single thread */
		float* weight = (float*)calloc((size_t)H * W, sizeof(float));
		memset(out, 0, sizeof(float) * (size_t)H * W * 3);
		for (int k = 0; k < n; ++k) {
			const int* q = roi + 4 * k;
			for (int i = q[1]; i < q[3]; ++i) for (int j = q[0]; j < q[2]; ++j) {
				float color[3], w;
				if (!color_and_w(g, &imgs[k], i, j, ordered_input, color, &w)) continue;
				float* row = out + ((size_t)i * W + j) * 3;
				row[0] += color[0]; row[1] += color[1]; row[2] += color[2];
				weight[(size_t)i * W + j] += w;
			}
		}
		for (size_t e = 0; e < (size_t)H * W; ++e) {
			float* row = out + e * 3;
			if (weight[e]) { row[0] /= weight[e]; row[1] /= weight[e]; row[2] /= weight[e]; }
			else { row[0] = -1; row[1] = -1; row[2] = -1; }
		}
		free(weight);
	} else {				/* blender.cc:77-93 */
		for (int i = 0; i < H; ++i) for (int j = 0; j < W; ++j) {
			float isum[3] = {0, 0, 0}, wsum = 0;
			for (int k = 0; k < n; ++k) {
				const int* q = roi + 4 * k;
				if (!(i >= q[1] && i <= q[3] && j >= q[0] && j <= q[2])) continue;	/* Range::contain, inclusive */
				float color[3], w;
				if (!color_and_w(g, &imgs[k], i, j, ordered_input, color, &w)) continue;
				isum[0] += color[0]; isum[1] += color[1]; isum[2] += color[2];
				wsum += w;
			}
			float* row = out + ((size_t)i * W + j) * 3;
			if (wsum > 0) {		/* Vector::operator/(T p) = *this * (1.0 / p), lib/geometry.hh:123-124 */
				const float inv = (float)(1.0 / wsum);
				row[0] = isum[0] * inv; row[1] = isum[1] * inv; row[2] = isum[2] * inv;
			} else { row[0] = -1; row[1] = -1; row[2] = -1; }		/* Color::NO */
		}
	}
	free(roi);
	return 0;
}

/* ---- multi-band ---- */
typedef struct { float c[3], w; } wpix;			/* multiband.hh:13-24 */

/* GaussCache (feature/gaussian.cc:17-40) with GAUSS_WINDOW_FACTOR */
static int mb_gauss_kernel(float sigma, int window_factor, float* k /* >= 64 */) {
	int kw = (int)(ceil(0.3 * (sigma / 2 - 1) + 0.8) * window_factor);
	if (kw % 2 == 0) kw++;
	const int center = kw / 2;
	float* kernel = k + center;
	kernel[0] = 1;
	float exp_coeff = (float)(-1.0 / (sigma * sigma * 2)), wsum = 1;
	for (int i = 1; i <= center; i++)
		wsum += (kernel[i] = expf(i * i * exp_coeff)) * 2;
	float fac = (float)(1.0 / wsum);
	kernel[0] = fac;
	for (int i = 1; i <= center; i++)
		kernel[-i] = (kernel[i] *= fac);
	return kw;
}

/* GaussianBlur::blur<WeightedPixel> (feature/gaussian.hh:30-91): columns, then rows in place */
static void blur_wpix(const wpix* img, wpix* ret, int h, int w, const float* kbuf, int kw) {
	const int center = kw / 2;
	const float* kernel = kbuf + center;
	const int mx = w > h ? w : h;
	wpix* mem = (wpix*)calloc(center * 2 + mx, sizeof(wpix));
	wpix* cur = mem + center;
	for (int j = 0; j < w; ++j) {
		for (int i = 0; i < h; ++i) cur[i] = img[(size_t)i * w + j];
		for (int i = 1; i <= center; i++) cur[-i] = cur[0];
		for (int i = 0; i < center; i++) cur[h + i] = cur[h - 1];
		for (int i = 0; i < h; ++i) {
			wpix t = {{0, 0, 0}, 0};
			for (int k = -center; k <= center; k++) {
				const float v = kernel[k];
				t.w += cur[i + k].w * v;
				t.c[0] += cur[i + k].c[0] * v; t.c[1] += cur[i + k].c[1] * v; t.c[2] += cur[i + k].c[2] * v;
			}
			ret[(size_t)i * w + j] = t;
		}
	}
	for (int i = 0; i < h; ++i) {
		wpix* dest = ret + (size_t)i * w;
		memcpy(cur, dest, sizeof(wpix) * w);
		for (int j = 1; j <= center; j++) cur[-j] = cur[0];
		for (int j = 0; j < center; j++) cur[w + j] = cur[w - 1];
		for (int j = 0; j < w; ++j) {
			wpix t = {{0, 0, 0}, 0};
			for (int k = -center; k <= center; k++) {
				const float v = kernel[k];
				t.w += cur[j + k].w * v;
				t.c[0] += cur[j + k].c[0] * v; t.c[1] += cur[j + k].c[1] * v; t.c[2] += cur[j + k].c[2] * v;
			}
			dest[j] = t;
		}
	}
	free(mem);
}

int orc_blend_multiband(const orc_blend_geom* g, const orc_blend_image* imgs, int n, int band_level, int window_factor, float* out) {
	int H, W; orc_blend_dims(g, imgs, n, &H, &W);
	int* roi = (int*)malloc(sizeof(int) * 4 * n);
	wpix** cur = (wpix**)calloc(n, sizeof(wpix*));
	wpix** nxt = (wpix**)calloc(n, sizeof(wpix*));
	unsigned char** mask = (unsigned char**)calloc(n, sizeof(unsigned char*));
	/* create_first_level (multiband.cc:19-56) */
	for (int k = 0; k < n; ++k) {
		int* q = roi + 4 * k; roi_of(g, imgs[k].range, q);
		const int rh = q[3] - q[1] + 1, rw = q[2] - q[0] + 1;		/* Range::height/width, inclusive */
		cur[k] = (wpix*)malloc(sizeof(wpix) * (size_t)rh * rw);
		nxt[k] = (wpix*)malloc(sizeof(wpix) * (size_t)rh * rw);
		mask[k] = (unsigned char*)calloc((size_t)rh * rw, 1);
		for (int i = 0; i < rh; ++i) for (int j = 0; j < rw; ++j) {
			double oc[2];
			canvas_to_image(g, &imgs[k], j + q[0], i + q[1], oc);
			float c[3];
			wpix* px = &cur[k][(size_t)i * rw + j];
			int ok = interpolate(imgs[k].data, imgs[k].h, imgs[k].w, (float)oc[1], (float)oc[0], c);
			if (ok) { float mn = c[0] < c[1] ? c[0] : c[1]; mn = mn < c[2] ? mn : c[2]; if (mn < 0) ok = 0; }
			if (!ok) {
				px->w = 0; px->c[0] = px->c[1] = px->c[2] = 0;
				mask[k][(size_t)i * rw + j] = 1;
			} else {
				px->c[0] = c[0]; px->c[1] = c[1]; px->c[2] = c[2];
				oc[0] = oc[0] / imgs[k].w - 0.5;
				oc[1] = oc[1] / imgs[k].h - 0.5;
				double v = (0.5f - fabs(oc[0])) * (0.5f - fabs(oc[1]));
				px->w = (float)((v > 0.0 ? v : 0.0) + ORC_EPS);
			}
		}
	}
	/* update_weight_map (multiband.cc:125-143) */
	for (int i = 0; i < H; ++i) for (int j = 0; j < W; ++j) {
		float mx = 0.f; float* maxp = NULL;
		for (int k = 0; k < n; ++k) {
			const int* q = roi + 4 * k;
			if (!(i >= q[1] && i <= q[3] && j >= q[0] && j <= q[2])) continue;
			const int rw = q[2] - q[0] + 1;
			float* w = &cur[k][(size_t)(i - q[1]) * rw + (j - q[0])].w;
			if (*w > mx) { mx = *w; maxp = w; }
			*w = 0;
		}
		if (maxp) *maxp = 1;
	}
	for (size_t e = 0; e < (size_t)H * W * 3; ++e) out[e] = -1;		/* fill(target, Color::NO) */
	unsigned char* tmask = (unsigned char*)calloc((size_t)H * W, 1);
	float kbuf[128];
	for (int level = 0; level < band_level; ++level) {
		const int is_last = (level == band_level - 1);
		if (!is_last) {		/* create_next_level (:145-151) */
			const int kw = mb_gauss_kernel((float)(sqrt(level * 2 + 1.0) * 4), window_factor, kbuf);
			for (int k = 0; k < n; ++k) {
				const int* q = roi + 4 * k;
				blur_wpix(cur[k], nxt[k], q[3] - q[1] + 1, q[2] - q[0] + 1, kbuf, kw);
			}
		}
		for (int i = 0; i < H; ++i) for (int j = 0; j < W; ++j) {		/* :75-110 */
			float isum[3] = {0, 0, 0}, wsum = 0;
			for (int k = 0; k < n; ++k) {
				const int* q = roi + 4 * k;
				if (!(i >= q[1] && i <= q[3] && j >= q[0] && j <= q[2])) continue;
				const int rw = q[2] - q[0] + 1;
				const size_t e = (size_t)(i - q[1]) * rw + (j - q[0]);
				if (mask[k][e]) continue;
				const float w = cur[k][e].w;
				if (w <= 0) continue;
				const float* cc = cur[k][e].c;
				if (!is_last) {
					const float* cn = nxt[k][e].c;
					isum[0] += (cc[0] - cn[0]) * w; isum[1] += (cc[1] - cn[1]) * w; isum[2] += (cc[2] - cn[2]) * w;
				} else {
					isum[0] += cc[0] * w; isum[1] += cc[1] * w; isum[2] += cc[2] * w;
				}
				wsum += w;
			}
			if (wsum < ORC_EPS) continue;		/* float < double */
			isum[0] /= wsum; isum[1] /= wsum; isum[2] /= wsum;
			float* p = out + ((size_t)i * W + j) * 3;
			if (!tmask[(size_t)i * W + j]) { p[0] = isum[0]; p[1] = isum[1]; p[2] = isum[2]; tmask[(size_t)i * W + j] = 1; }
			else { p[0] += isum[0]; p[1] += isum[1]; p[2] += isum[2]; }
		}
		wpix** t = cur; cur = nxt; nxt = t;		/* swap(next_lvl_images, images) */
	}
	for (size_t e = 0; e < (size_t)H * W; ++e) if (tmask[e]) {		/* :112-121 */
		float* p = out + e * 3;
		for (int c = 0; c < 3; ++c) { float v = p[c] < 1.0f ? p[c] : 1.0f; p[c] = v > 0.f ? v : 0.f; }
	}
	for (int k = 0; k < n; ++k) { free(cur[k]); free(nxt[k]); free(mask[k]); }
	free(cur); free(nxt); free(mask); free(roi); free(tmask);
	return 0;
}

/* ---- CylinderWarper::get_projector + CylinderProject::project (stitch/warp.cc:13-75) ---- */
typedef struct { double cx, cy; int r, sizefactor; } cylproj;

static cylproj get_projector(int w, int h, double h_factor, float focal_length) {
	cylproj p;
	p.r = (int)(hypot(w, h) * (focal_length / 43.266));		/* float / double -> double */
	p.cx = w / 2; p.cy = h / 2 * h_factor;						/* integer halves (warp.cc:73) */
	p.sizefactor = p.r;
	return p;
}
static void cyl_proj(const cylproj* P, double px, double py, double out[2]) {	/* :13-17 */
	out[0] = atan((px - P->cx) / P->r);
	out[1] = (py - P->cy) / (hypot(px - P->cx, P->r));
}
static void cyl_proj_r(const cylproj* P, double x, double y, double out[2]) {	/* :19-23 */
	out[0] = P->r * tan(x) + P->cx;
	out[1] = y * P->r / cos(x) + P->cy;
}

/* project(Shape2D&, pts) (:46-67): returns offset, updates shape and points (centred coords) */
int orc_cyl_shape(int w, int h, double h_factor, float focal_length, double* pts, int npts,
		int* new_w, int* new_h, double* offset) {
	const cylproj P = get_projector(w, h, h_factor, focal_length);
	double mn[2] = {DBL_MAX, DBL_MAX}, mx[2] = {0, 0};
	for (int i = 0; i < h; ++i) for (int j = 0; j < w; ++j) {
		double c[2]; cyl_proj(&P, j, i, c);
		if (c[0] < mn[0]) mn[0] = c[0];
		if (c[1] < mn[1]) mn[1] = c[1];
		if (c[0] > mx[0]) mx[0] = c[0];
		if (c[1] > mx[1]) mx[1] = c[1];
	}
	for (int c = 0; c < 2; ++c) { mx[c] = mx[c] * P.sizefactor; mn[c] = mn[c] * P.sizefactor; }
	const double rsx = mx[0] - mn[0], rsy = mx[1] - mn[1];
	offset[0] = mn[0] * (-1); offset[1] = mn[1] * (-1);
	const int sx = (int)rsx, sy = (int)rsy;
	for (int k = 0; k < npts; ++k) {
		double c[2];
		cyl_proj(&P, pts[2 * k] + w / 2, pts[2 * k + 1] + h / 2, c);
		pts[2 * k] = c[0] * P.sizefactor + offset[0];
		pts[2 * k + 1] = c[1] * P.sizefactor + offset[1];
		pts[2 * k] -= sx / 2;
		pts[2 * k + 1] -= sy / 2;
	}
	*new_w = sx; *new_h = sy;
	return 0;
}

/* project(const Mat32f&, pts) (:25-44); out must hold new_h*new_w*3 floats (from orc_cyl_shape) */
int orc_cyl_project(const float* img, int h, int w, double h_factor, float focal_length, float* out) {
	const cylproj P = get_projector(w, h, h_factor, focal_length);
	int nw, nh; double offset[2];
	orc_cyl_shape(w, h, h_factor, focal_length, NULL, 0, &nw, &nh, offset);
	const double sizefactor_inv = 1.0 / P.sizefactor;
	for (size_t e = 0; e < (size_t)nh * nw * 3; ++e) out[e] = -1;
	for (int i = 0; i < nh; ++i) for (int j = 0; j < nw; ++j) {
		double o[2];
		cyl_proj_r(&P, ((double)j - offset[0]) * sizefactor_inv, ((double)i - offset[1]) * sizefactor_inv, o);
		if (o[0] >= 0 && o[0] <= w - 1 && o[1] >= 0 && o[1] <= h - 1) {	/* between(a,b,c) = a >= b && a <= c-1, lib/utils.hh:27 */
			float c[3] = {-1, -1, -1};
			interpolate(img, h, w, (float)o[1], (float)o[0], c);	/* Color::NO leaves -1 */
			float* p = out + ((size_t)i * nw + j) * 3;
			p[0] = c[0]; p[1] = c[1]; p[2] = c[2];
		}
	}
	return 0;
}

/* ---- crop (lib/imgproc.cc:200-235): largest all-valid rectangle, first maximum in (line, k) order ---- */
int orc_crop_rect(const float* mat, int h, int w, int* x0, int* y0, int* cw, int* ch) {
	int* height = (int*)calloc(w, sizeof(int));
	int* left = (int*)malloc(sizeof(int) * w);
	int* right = (int*)malloc(sizeof(int) * w);
	int maxarea = 0, ll = 0, rr = 0, hh = 0, nl = 0;
	for (int line = 0; line < h; ++line) {
		for (int k = 0; k < w; ++k) {
			const float* p = mat + ((size_t)line * w + k) * 3;
			float m = p[0] > p[1] ? p[0] : p[1]; m = m > p[2] ? m : p[2];	/* max(max(p0,p1),p2) */
			height[k] = m < 0 ? 0 : height[k] + 1;
		}
		for (int k = 0; k < w; ++k) {
			left[k] = k;
			while (left[k] > 0 && height[k] <= height[left[k] - 1]) left[k] = left[left[k] - 1];
		}
		for (int k = w - 1; k >= 0; --k) {
			right[k] = k;
			while (right[k] < w - 1 && height[k] <= height[right[k] + 1]) right[k] = right[right[k] + 1];
		}
		for (int k = 0; k < w; ++k) {
			const int area = (right[k] - left[k] + 1) * height[k];
			if (maxarea < area) { maxarea = area; ll = left[k]; rr = right[k]; hh = height[k]; nl = line; }
		}
	}
	*x0 = ll; *y0 = nl - hh + 1; *cw = rr - ll + 1; *ch = hh;
	free(height); free(left); free(right);
	return maxarea;
}

/* write_rgb / write_png quantisation (lib/imgio.cc:25-40,98-113): Color::NO -> white, float * 255 truncated */
void orc_to_u8(const float* mat, long n, unsigned char* out) {
	for (long i = 0; i < n; ++i) out[i] = (unsigned char)((mat[i] < 0 ? 1 : mat[i]) * 255);
}
