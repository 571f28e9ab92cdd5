/* oracle/sift_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of the reference SIFT path, operation for operation and in the same
 * floating-point types and evaluation order as the C++ source (compiled -ffp-contract=off,
 * as is oracle/_ref).  Citations are file:line under /root/reference/src.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <omp.h>
#include "oracle.h"

#define MAX_OCT 8
#define MAX_SCALE 12

typedef struct { int h, w; float* p; } plane_t;

typedef struct {
	int x, y, pyr, scale;
	double rx, ry;		/* real_coor, [0,1) */
	float dir, sf;		/* dir, scale_factor */
} sspoint_t;			/* feature/feature.hh:33-39 */

typedef struct { int n, cap; sspoint_t* v; } spvec_t;
typedef struct { int n, cap; int* xy; } coorvec_t;

struct orc_sift_run {
	orc_sift_cfg cfg;
	plane_t work;				/* working RGB, HWC */
	int noct, nscale;
	int oh[MAX_OCT], ow[MAX_OCT];
	plane_t gauss[MAX_OCT][MAX_SCALE];	/* data[s], s=0 grey */
	plane_t mag[MAX_OCT][MAX_SCALE], ort[MAX_OCT][MAX_SCALE];
	plane_t dog[MAX_OCT][MAX_SCALE];
	coorvec_t raw[MAX_OCT][MAX_SCALE];
	spvec_t refined, oriented;
	float* desc;				/* oriented.n * 128 */
};

void orc_sift_cfg_default(orc_sift_cfg* c) {
	/* src/config.cfg:19-49; each literal parsed as float like ConfigParser (lib/config.cc:19-26) */
	c->SIFT_WORKING_SIZE = 800; c->NUM_OCTAVE = 4; c->NUM_SCALE = 7;
	c->SCALE_FACTOR = 1.4142135623f; c->GAUSS_SIGMA = 1.4142135623f;
	c->GAUSS_WINDOW_FACTOR = 6;
	c->JUDGE_EXTREMA_DIFF_THRES = 2e-3f; c->CONTRAST_THRES = 4e-2f;
	c->PRE_COLOR_THRES = 5e-2f; c->EDGE_RATIO = 6.f;
	c->CALC_OFFSET_DEPTH = 4; c->OFFSET_THRES = 0.5f;
	c->ORI_RADIUS = 4.5f; c->ORI_HIST_SMOOTH_COUNT = 2;
	c->DESC_HIST_SCALE_FACTOR = 3; c->DESC_INT_FACTOR = 512;
	c->MATCH_REJECT_NEXT_RATIO = 0.8f;
}

static plane_t plane_new(int h, int w, int ch) {
	plane_t p; p.h = h; p.w = w;
	p.p = (float*)malloc(sizeof(float) * (size_t)h * w * ch);
	return p;
}

/* ---- lib/imgproc.cc:22-80 resize_bilinear (x = row index, y = column index there) ---- */
static void resize_bilinear(const float* src, int sh, int sw, float* dst, int dh, int dw, int ch) {
	int* tabsx = (int*)malloc(sizeof(int) * dh);
	int* tabsy = (int*)malloc(sizeof(int) * dw);
	float* tabrx = (float*)malloc(sizeof(float) * dh);
	float* tabry = (float*)malloc(sizeof(float) * dw);
	const float fx = (float)dh / sh;
	const float fy = (float)dw / sw;
	const float ifx = 1.f / fx;
	const float ify = 1.f / fy;
	for (int dx = 0; dx < dh; ++dx) {
		float rx = (dx + 0.5f) * ifx - 0.5f;
		int sx = (int)floorf(rx);
		rx -= sx;
		if (sx < 0) { sx = 0; rx = 0; }
		else if (sx + 1 >= sh) { sx = sh - 2; rx = 1; }
		tabsx[dx] = sx; tabrx[dx] = rx;
	}
	for (int dy = 0; dy < dw; ++dy) {
		float ry = (dy + 0.5f) * ify - 0.5f;
		int sy = (int)floorf(ry);
		ry -= sy;
		if (sy < 0) { sy = 0; ry = 0; }
		else if (sy + 1 >= sw) { sy = sw - 2; ry = 1; }
		tabsy[dy] = sy; tabry[dy] = ry;
	}
	for (int dx = 0; dx < dh; ++dx) {
		const float* p0 = src + (size_t)(tabsx[dx] + 0) * sw * ch;
		const float* p1 = src + (size_t)(tabsx[dx] + 1) * sw * ch;
		float* pdst = dst + (size_t)dx * dw * ch;
		float rx = tabrx[dx], irx = 1.0f - rx;
		for (int dy = 0; dy < dw; ++dy) {
			const float* pc00 = p0 + (tabsy[dy] + 0) * ch;
			const float* pc01 = p0 + (tabsy[dy] + 1) * ch;
			const float* pc10 = p1 + (tabsy[dy] + 0) * ch;
			const float* pc11 = p1 + (tabsy[dy] + 1) * ch;
			float ry = tabry[dy], iry = 1.0f - ry;
			for (int c = 0; c < ch; ++c)
				pdst[dy * ch + c] = rx * (pc11[c] * ry + pc10[c] * iry)
					+ irx * (pc01[c] * ry + pc00[c] * iry);	/* :74-75 */
		}
	}
	free(tabsx); free(tabsy); free(tabrx); free(tabry);
}

/* ---- feature/gaussian.cc:17-40 GaussCache ---- */
static int gauss_kernel(const orc_sift_cfg* cfg, float sigma, float* k /* >= 64 */) {
	int kw = (int)(ceil(0.3 * (sigma / 2 - 1) + 0.8) * cfg->GAUSS_WINDOW_FACTOR);
	if (kw % 2 == 0) kw++;
	const int center = kw / 2;
	float* kernel = k + center;
	kernel[0] = 1;
	float exp_coeff = (float)(-1.0 / (sigma * sigma * 2)), wsum = 1;
	for (int i = 1; i <= center; i++)
		wsum += (kernel[i] = expf(i * i * exp_coeff)) * 2;	/* std::exp(float) overload */
	float fac = (float)(1.0 / wsum);
	kernel[0] = fac;
	for (int i = 1; i <= center; i++)
		kernel[-i] = (kernel[i] *= fac);
	return kw;
}
int orc_gauss_kernel(const orc_sift_cfg* cfg, float sigma, float* out) { return gauss_kernel(cfg, sigma, out); }

/* ---- feature/gaussian.hh:30-91 GaussianBlur::blur<float> ---- */
static void blur(const plane_t* img, plane_t* ret, const float* kbuf, int kw) {
	const int w = img->w, h = img->h;
	const int center = kw / 2;
	const float* kernel = kbuf + center;
	int mx = w > h ? w : h;
	float* mem = (float*)calloc(center * 2 + mx, sizeof(float));
	float* cur_line = mem + center;
	for (int j = 0; j < w; ++j) {		/* columns first */
		const float* src = img->p + j;
		for (int i = 0; i < h; ++i) { cur_line[i] = *src; src += w; }
		float v0 = cur_line[0];
		for (int i = 1; i <= center; i++) cur_line[-i] = v0;
		v0 = cur_line[h - 1];
		for (int i = 0; i < center; i++) cur_line[h + i] = v0;
		float* dest = ret->p + j;
		for (int i = 0; i < h; ++i) {
			float tmp = 0;
			for (int k = -center; k <= center; k++)
				tmp += cur_line[i + k] * kernel[k];
			*dest = tmp; dest += w;
		}
	}
	for (int i = 0; i < h; ++i) {		/* then rows, in place */
		float* dest = ret->p + (size_t)i * w;
		memcpy(cur_line, dest, sizeof(float) * w);
		float v0 = cur_line[0];
		for (int j = 1; j <= center; j++) cur_line[-j] = v0;
		v0 = cur_line[w - 1];
		for (int j = 0; j < center; j++) cur_line[w + j] = v0;
		for (int j = 0; j < w; ++j) {
			float tmp = 0;
			for (int k = -center; k <= center; k++)
				tmp += cur_line[j + k] * kernel[k];
			*(dest++) = tmp;
		}
	}
	free(mem);
}

/* ---- feature/dog.cc:22-37 fast_atan ---- */
static float fast_atan(float y, float x) {
	float absx = fabsf(x), absy = fabsf(y);
	float m = absx > absy ? absx : absy;	/* std::max(absx, absy) */
	if (m < 1e-6) return (float)-M_PI;	/* EPS is real_t 1e-6 (lib/utils.hh:22) */
	float a = (absx < absy ? absx : absy) / m;	/* std::min(absx, absy): returns absx on tie;
	                                               value identical either way */
	float s = a * a;
	float r = (float)(((-0.0464964749 * s + 0.15931422) * s - 0.327622764) * s * a + a);
	if (absy > absx) r = (float)(M_PI_2 - r);
	if (x < 0) r = (float)(M_PI - r);
	if (y < 0) r = -r;
	return r;
}

/* ---- feature/dog.cc:60-94 cal_mag_ort ---- */
static void cal_mag_ort(const plane_t* orig, plane_t* mag, plane_t* ort) {
	int w = orig->w, h = orig->h;
	for (int y = 0; y < h; ++y) {
		float* mag_row = mag->p + (size_t)y * w;
		float* ort_row = ort->p + (size_t)y * w;
		const float* orig_row = orig->p + (size_t)y * w;
		const float* orig_plus = orig_row + w;
		const float* orig_minus = orig_row - w;
		mag_row[0] = 0; ort_row[0] = (float)M_PI;
		for (int x = 1; x < w - 1; ++x) {
			if (y >= 1 && y <= h - 2) {
				float dy = orig_plus[x] - orig_minus[x], dx = orig_row[x + 1] - orig_row[x - 1];
				mag_row[x] = hypotf(dx, dy);
				ort_row[x] = (float)(fast_atan(dy, dx) + M_PI);
			} else {
				mag_row[x] = 0; ort_row[x] = (float)M_PI;
			}
		}
		mag_row[w - 1] = 0; ort_row[w - 1] = (float)M_PI;
	}
}

/* ---- feature/dog.cc:42-58 GaussianPyramid ctor + :116-143 DOGSpace ---- */
static void build_octave(orc_sift_run* r, int o, const float* rgb) {
	const orc_sift_cfg* cfg = &r->cfg;
	int h = r->oh[o], w = r->ow[o], ns = r->nscale;
	r->gauss[o][0] = plane_new(h, w, 1);
	float* g = r->gauss[o][0].p;
	int n = h * w;
	for (int i = 0; i < n; ++i)	/* lib/imgproc.cc:237-249 rgb2grey */
		g[i] = (rgb[3 * i] + rgb[3 * i + 1] + rgb[3 * i + 2]) / 3.f;
	float sigma = cfg->GAUSS_SIGMA;		/* gaussian.hh:96-103 */
	for (int i = 1; i < ns; i++) {
		float kbuf[128];
		int kw = gauss_kernel(cfg, sigma, kbuf);
		sigma *= cfg->SCALE_FACTOR;
		r->gauss[o][i] = plane_new(h, w, 1);
		blur(&r->gauss[o][0], &r->gauss[o][i], kbuf, kw);
		r->mag[o][i] = plane_new(h, w, 1);
		r->ort[o][i] = plane_new(h, w, 1);
		cal_mag_ort(&r->gauss[o][i], &r->mag[o][i], &r->ort[o][i]);
	}
	for (int j = 0; j < ns - 1; ++j) {	/* dog.cc:116-129: fabs(p1 - p2) */
		r->dog[o][j] = plane_new(h, w, 1);
		const float* p1 = r->gauss[o][j].p; const float* p2 = r->gauss[o][j + 1].p;
		float* p = r->dog[o][j].p;
		for (int i = 0; i < n; ++i) p[i] = fabsf(p1[i] - p2[i]);
	}
}

/* ---- feature/extrema.cc:170-216 get_local_raw_extrema ---- */
static void coor_push(coorvec_t* v, int x, int y) {
	if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->xy = (int*)realloc(v->xy, sizeof(int) * 2 * v->cap); }
	v->xy[2 * v->n] = x; v->xy[2 * v->n + 1] = y; v->n++;
}
static int is_extrema(const orc_sift_run* R, int o, int s, int r, int c) {
	const orc_sift_cfg* cfg = &R->cfg;
	const plane_t* now = &R->dog[o][s];
	int w = now->w;
	float center = now->p[(size_t)r * w + c];
	if (center < cfg->PRE_COLOR_THRES) return 0;
	int max = 1, min = 1;
	float cmp1 = center - cfg->JUDGE_EXTREMA_DIFF_THRES, cmp2 = center + cfg->JUDGE_EXTREMA_DIFF_THRES;
	for (int di = -1; di < 2; ++di) for (int dj = -1; dj < 2; ++dj) {
		if (!di && !dj) continue;
		float newval = now->p[(size_t)(r + di) * w + c + dj];
		if (newval >= cmp1) max = 0;
		if (newval <= cmp2) min = 0;
		if (!max && !min) return 0;
	}
	for (int ds = -1; ds < 2; ds += 2) {
		const plane_t* mat = &R->dog[o][s + ds];
		for (int di = -1; di < 2; ++di) {
			const float* p = mat->p + (size_t)(r + di) * w + c - 1;
			for (int i = 0; i < 3; ++i) {
				float newval = p[i];
				if (newval >= cmp1) max = 0;
				if (newval <= cmp2) min = 0;
				if (!max && !min) return 0;
			}
		}
	}
	return 1;
}

/* Eigen::FullPivLU<3x3> inverse as used by Matrix::inverse (lib/matrix.cc:76-87): complete
 * pivoting, rank threshold |pivot| > |maxpivot| * eps * 3, inverse = solve(I).  Restated from
 * the published algorithm (Eigen is absent; same sequence as oracle/ref_shim/Eigen/Dense). */
static int inverse3_fullpiv(const double a[9], double inv[9]) {
	double lu[9]; memcpy(lu, a, sizeof(lu));
	int rowt[3], colt[3], nonzero = 3; double maxpivot = 0;
	for (int k = 0; k < 3; ++k) {
		int br = k, bc = k; double best = -1;
		for (int i = k; i < 3; ++i) for (int j = k; j < 3; ++j) {
			double v = fabs(lu[i * 3 + j]);
			if (v > best) { best = v; br = i; bc = j; }
		}
		if (best == 0.0) { nonzero = k; for (int i = k; i < 3; ++i) rowt[i] = colt[i] = i; break; }
		if (best > maxpivot) maxpivot = best;
		rowt[k] = br; colt[k] = bc;
		if (br != k) for (int j = 0; j < 3; ++j) { double t = lu[k * 3 + j]; lu[k * 3 + j] = lu[br * 3 + j]; lu[br * 3 + j] = t; }
		if (bc != k) for (int i = 0; i < 3; ++i) { double t = lu[i * 3 + k]; lu[i * 3 + k] = lu[i * 3 + bc]; lu[i * 3 + bc] = t; }
		for (int i = k + 1; i < 3; ++i) lu[i * 3 + k] /= lu[k * 3 + k];
		for (int i = k + 1; i < 3; ++i) for (int j = k + 1; j < 3; ++j)
			lu[i * 3 + j] -= lu[i * 3 + k] * lu[k * 3 + j];
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
		for (int i = 2; i >= 0; --i) {
			for (int j = i + 1; j < 3; ++j) c[i] -= lu[i * 3 + j] * c[j];
			c[i] /= lu[i * 3 + i];
		}
		for (int i = 2; i >= 0; --i) { double t = c[i]; c[i] = c[colt[i]]; c[colt[i]] = t; }
		for (int i = 0; i < 3; ++i) inv[i * 3 + col] = c[i];
	}
	return 1;
}

/* Matrix::pseudo_inverse (lib/matrix.cc:89-106) for a symmetric 3x3: V diag(1/s > EPS) U^T via
 * one-sided Jacobi (same sequence as the ref_shim JacobiSVD). Only reached for singular Hessians. */
static void pinv3_jacobi(const double a[9], double out[9]) {
	double A[9], V[9] = {1,0,0, 0,1,0, 0,0,1};
	memcpy(A, a, sizeof(A));
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
	memset(out, 0, sizeof(double) * 9);
	for (int j = 0; j < 3; ++j) {
		double s = 0;
		for (int i = 0; i < 3; ++i) s += A[i*3+j] * A[i*3+j];
		s = sqrt(s);
		if (!(s > 1e-6)) continue;		/* EPS, lib/matrix.cc:96 */
		/* column j: u = A[:,j]/s, v = V[:,j]; pinv += v * (1/s) * u^T */
		for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
			out[r*3+c] += V[r*3+j] * (1.0 / s) * (A[c*3+j] / s);
	}
}

/* ---- feature/extrema.cc:108-150 calc_kp_offset_iter ---- */
static void kp_offset_iter(const orc_sift_run* R, int o, int x, int y, int s, double offset[3], double delta[3]) {
	int w = R->ow[o];
#define D(x, y, s) (R->dog[o][s].p[(size_t)(y) * w + (x)])
#define DS(x, y) D(x, y, s)
	float val = DS(x, y);
	delta[0] = (DS(x + 1, y) - DS(x - 1, y)) / 2;
	delta[1] = (DS(x, y + 1) - DS(x, y - 1)) / 2;
	delta[2] = (D(x, y, s + 1) - D(x, y, s - 1)) / 2;
	double dxx = DS(x + 1, y) + DS(x - 1, y) - val - val;
	double dyy = DS(x, y + 1) + DS(x, y - 1) - val - val;
	double dss = D(x, y, s + 1) + D(x, y, s - 1) - val - val;
	double dxy = (DS(x + 1, y + 1) - DS(x + 1, y - 1) - DS(x - 1, y + 1) + DS(x - 1, y - 1)) / 4;
	double dys = (D(x, y + 1, s + 1) - D(x, y - 1, s + 1) - D(x, y + 1, s - 1) + D(x, y - 1, s - 1)) / 4;
	double dsx = (D(x + 1, y, s + 1) - D(x - 1, y, s + 1) - D(x + 1, y, s - 1) + D(x - 1, y, s - 1)) / 4;
#undef D
#undef DS
	double m[9] = { dxx, dxy, dsx,  dxy, dyy, dys,  dsx, dys, dss }, inv[9];
	if (!inverse3_fullpiv(m, inv)) pinv3_jacobi(m, inv);
	for (int i = 0; i < 3; ++i) {		/* inv.prod(pdpx), lib/matrix.cc:40-48 */
		double sacc = 0;
		for (int k = 0; k < 3; ++k) sacc += inv[i * 3 + k] * delta[k];
		offset[i] = sacc;
	}
}

/* ---- feature/extrema.cc:63-106 calc_kp_offset ---- */
static int calc_kp_offset(const orc_sift_run* R, sspoint_t* sp) {
	const orc_sift_cfg* cfg = &R->cfg;
	int o = sp->pyr, w = R->ow[o], h = R->oh[o], nscale = R->nscale;
	double offset[3] = {0, 0, 0}, delta[3] = {0, 0, 0};
	int nowx = sp->x, nowy = sp->y, nows = sp->scale;
	int niter = 0;
	for (; niter < cfg->CALC_OFFSET_DEPTH; ++niter) {
		if (!(nowx >= 1 && nowx <= w - 2) || !(nowy >= 1 && nowy <= h - 2) || !(nows >= 1 && nows <= nscale - 3))
			return 0;
		kp_offset_iter(R, o, nowx, nowy, nows, offset, delta);
		double ax = fabs(offset[0]), ay = fabs(offset[1]), az = fabs(offset[2]);
		double mx = ay > az ? ay : az; mx = ax > mx ? ax : mx;	/* std::max(|x|, std::max(|y|,|z|)) */
		if (mx < cfg->OFFSET_THRES) break;
		nowx += round(offset[0]);
		nowy += round(offset[1]);
		nows += round(offset[2]);
	}
	if (niter == cfg->CALC_OFFSET_DEPTH) return 0;
	double dextr = offset[0] * delta[0] + offset[1] * delta[1] + offset[2] * delta[2];
	dextr = R->dog[o][nows].p[(size_t)nowy * w + nowx] + dextr / 2;
	if (dextr < cfg->CONTRAST_THRES) return 0;
	sp->x = nowx; sp->y = nowy; sp->scale = nows;
	sp->sf = (float)(cfg->GAUSS_SIGMA * pow(cfg->SCALE_FACTOR, ((double)nows + offset[2]) / nscale));
	sp->rx = ((double)nowx + offset[0]) / w;
	sp->ry = ((double)nowy + offset[1]) / h;
	return 1;
}

/* ---- feature/extrema.cc:152-168 is_edge_response ---- */
static int is_edge_response(const orc_sift_cfg* cfg, int x, int y, const plane_t* img) {
	int w = img->w;
#define AT(r, c) (img->p[(size_t)(r) * w + (c)])
	float val = AT(y, x);
	float dxx = AT(y, x + 1) + AT(y, x - 1) - val - val;
	float dyy = AT(y + 1, x) + AT(y - 1, x) - val - val;
	float dxy = (AT(y + 1, x + 1) + AT(y - 1, x - 1) - AT(y + 1, x - 1) - AT(y - 1, x + 1)) / 4;
#undef AT
	float det = dxx * dyy - dxy * dxy;
	if (det <= 0) return 1;
	float t = dxx + dyy;
	float tr2 = t * t;
	float e1 = cfg->EDGE_RATIO + 1;
	if (tr2 / det < (e1 * e1) / cfg->EDGE_RATIO) return 0;
	return 1;
}

static void sp_push(spvec_t* v, const sspoint_t* p) {
	if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->v = (sspoint_t*)realloc(v->v, sizeof(sspoint_t) * v->cap); }
	v->v[v->n++] = *p;
}

/* canonical order (the reference's is thread-timing dependent, extrema.cc:56); same key as
 * ref_driver.cc: (pyr, scale, y, x, real_x, real_y) on the *refined* point */
static int sp_cmp(const void* a_, const void* b_) {
	const sspoint_t* a = (const sspoint_t*)a_; const sspoint_t* b = (const sspoint_t*)b_;
	if (a->pyr != b->pyr) return a->pyr < b->pyr ? -1 : 1;
	if (a->scale != b->scale) return a->scale < b->scale ? -1 : 1;
	if (a->y != b->y) return a->y < b->y ? -1 : 1;
	if (a->x != b->x) return a->x < b->x ? -1 : 1;
	if (a->rx != b->rx) return a->rx < b->rx ? -1 : 1;
	if (a->ry != b->ry) return a->ry < b->ry ? -1 : 1;
	return 0;
}

/* ---- feature/orientation.cc:34-100 calc_dir ---- */
static int calc_dir(const orc_sift_run* R, const sspoint_t* p, float* out /* <= 36 */) {
	const orc_sift_cfg* cfg = &R->cfg;
	const float halfipi = (float)(0.5f / M_PI);
	const plane_t* orient_img = &R->ort[p->pyr][p->scale];
	const plane_t* mag_img = &R->mag[p->pyr][p->scale];
	int pw = R->ow[p->pyr], ph = R->oh[p->pyr];
	float gauss_weight_sigma = p->sf * 1.5f;			/* ORI_WINDOW_FACTOR */
	int rad = (int)roundf(p->sf * cfg->ORI_RADIUS);
	float exp_denom = 2 * (gauss_weight_sigma * gauss_weight_sigma);
	float hist[36];
	memset(hist, 0, sizeof(hist));
	for (int xx = -rad; xx < rad; xx++) {
		int newx = p->x + xx;
		if (!(newx >= 1 && newx <= pw - 2)) continue;
		for (int yy = -rad; yy < rad; yy++) {
			int newy = p->y + yy;
			if (!(newy >= 1 && newy <= ph - 2)) continue;
			float fxx = (float)xx, fyy = (float)yy, frad = (float)rad;
			if (fxx * fxx + fyy * fyy > frad * frad) continue;
			float orient = orient_img->p[(size_t)newy * pw + newx];
			int bin = (int)roundf(36 * halfipi * orient);
			if (bin == 36) bin = 0;
			float weight = expf(-(fxx * fxx + fyy * fyy) / exp_denom);
			hist[bin] += weight * mag_img->p[(size_t)newy * pw + newx];
		}
	}
	for (int K = cfg->ORI_HIST_SMOOTH_COUNT; K--;)
		for (int i = 0; i < 36; ++i) {
			float prev = hist[i == 0 ? 35 : i - 1];
			float next = hist[i == 35 ? 0 : i + 1];
			hist[i] = (float)(hist[i] * 0.5 + (prev + next) * 0.25);
		}
	float maxbin = 0;
	for (int i = 0; i < 36; ++i) if (maxbin < hist[i]) maxbin = hist[i];
	float thres = maxbin * 0.8f;					/* ORI_HIST_PEAK_RATIO */
	int n = 0;
	for (int i = 0; i < 36; ++i) {
		float prev = hist[i == 0 ? 35 : i - 1];
		float next = hist[i == 35 ? 0 : i + 1];
		float mpn = prev < next ? next : prev;		/* std::max(prev, next) */
		if (hist[i] > thres && hist[i] > mpn) {
			double newbin = (float)i - 0.5 + (hist[i] - prev) / (prev + next - 2 * hist[i]);
			if (newbin < 0) newbin += 36;
			else if (newbin >= 36) newbin -= 36;
			out[n++] = (float)(newbin / 36 * 2 * M_PI);
		}
	}
	return n;
}

/* ---- feature/sift.cc:48-67 trilinear_interpolate ---- */
static void trilinear(float xbin, float ybin, float hbin, float weight, float hist[16][8]) {
	int ybinf = (int)floorf(ybin), xbinf = (int)floorf(xbin), hbinf = (int)floorf(hbin);
	float ybind = ybin - ybinf, xbind = xbin - xbinf, hbind = hbin - hbinf;
	for (int dy = 0; dy < 2; ++dy) if (ybinf + dy >= 0 && ybinf + dy <= 3) {
		float w_y = weight * (dy ? ybind : 1 - ybind);
		for (int dx = 0; dx < 2; ++dx) if (xbinf + dx >= 0 && xbinf + dx <= 3) {
			float w_x = w_y * (dx ? xbind : 1 - xbind);
			int idx = (ybinf + dy) * 4 + (xbinf + dx);
			hist[idx][hbinf % 8] += w_x * (1 - hbind);
			hist[idx][(hbinf + 1) % 8] += w_x * hbind;
		}
	}
}

/* ---- feature/sift.cc:87-152 calc_descriptor + :15-46 hist_to_descriptor ---- */
static void calc_descriptor(const orc_sift_run* R, const sspoint_t* p, float* out) {
	const orc_sift_cfg* cfg = &R->cfg;
	const float pi2 = (float)(2 * M_PI);
	const float nbin_per_rad = 8 / pi2;
	int w = R->ow[p->pyr], h = R->oh[p->pyr];
	const plane_t* mag_img = &R->mag[p->pyr][p->scale];
	const plane_t* ort_img = &R->ort[p->pyr][p->scale];
	float ort = p->dir, hist_w = p->sf * cfg->DESC_HIST_SCALE_FACTOR, exp_denom = 2 * (4.f * 4.f);
	int radius = (int)round(M_SQRT1_2 * hist_w * (4 + 1));
	float hist[16][8];
	memset(hist, 0, sizeof(hist));
	float cosort = cosf(ort), sinort = sinf(ort);
	for (int xx = -radius; xx <= radius; xx++) {
		int nowx = p->x + xx;
		if (!(nowx >= 1 && nowx <= w - 2)) continue;
		for (int yy = -radius; yy <= radius; yy++) {
			int nowy = p->y + yy;
			if (!(nowy >= 1 && nowy <= h - 2)) continue;
			float fxx = (float)xx, fyy = (float)yy, fr = (float)radius;
			if (fxx * fxx + fyy * fyy > fr * fr) continue;
			float y_rot = (-xx * sinort + yy * cosort) / hist_w,
				  x_rot = (xx * cosort + yy * sinort) / hist_w;
			float ybin = (float)(y_rot + 4 / 2 - 0.5), xbin = (float)(x_rot + 4 / 2 - 0.5);
			/* between(a, -1, 4) on floats = a >= -1 && a <= 3 (lib/utils.hh:27) */
			if (!(ybin >= -1 && ybin <= 3) || !(xbin >= -1 && xbin <= 3)) continue;
			float now_mag = mag_img->p[(size_t)nowy * w + nowx], now_ort = ort_img->p[(size_t)nowy * w + nowx];
			float weight = expf(-(x_rot * x_rot + y_rot * y_rot) / exp_denom);
			weight = weight * now_mag;
			now_ort -= ort;
			if (now_ort < 0) now_ort += pi2;
			if (now_ort > pi2) now_ort -= pi2;
			float hist_bin = now_ort * nbin_per_rad;
			trilinear(xbin, ybin, hist_bin, weight, hist);
		}
	}
	const float* hf = &hist[0][0];
	float sum = 0;
	for (int i = 0; i < 128; ++i) sum += hf[i];
	for (int i = 0; i < 128; ++i) {
		float v = hf[i] / sum;
		out[i] = sqrtf(v) * cfg->DESC_INT_FACTOR;
	}
}

orc_sift_run* orc_sift_new(const orc_sift_cfg* cfg, const float* rgb, int h, int w) {
	orc_sift_run* r = (orc_sift_run*)calloc(1, sizeof(orc_sift_run));
	r->cfg = *cfg;
	/* feature/feature.cc:33-35 */
	float ratio = cfg->SIFT_WORKING_SIZE * 2.0f / (w + h);
	int wh = (int)(h * ratio), ww = (int)(w * ratio);
	r->work = plane_new(wh, ww, 3);
	resize_bilinear(rgb, h, w, r->work.p, wh, ww, 3);
	r->noct = cfg->NUM_OCTAVE; r->nscale = cfg->NUM_SCALE;
	/* feature/dog.cc:96-114 ScaleSpace */
	for (int i = 0; i < r->noct; ++i) {
		if (!i) {
			r->oh[0] = wh; r->ow[0] = ww;
			build_octave(r, 0, r->work.p);
		} else {
			float factor = (float)pow(cfg->SCALE_FACTOR, -i);
			int neww = (int)ceilf(ww * factor), newh = (int)ceilf(wh * factor);
			r->oh[i] = newh; r->ow[i] = neww;
			float* tmp = (float*)malloc(sizeof(float) * (size_t)newh * neww * 3);
			resize_bilinear(r->work.p, wh, ww, tmp, newh, neww, 3);
			build_octave(r, i, tmp);
			free(tmp);
		}
	}
	/* feature/extrema.cc:36-61 */
	for (int i = 0; i < r->noct; ++i)
		for (int j = 1; j < r->nscale - 2; ++j) {
			int ww_ = r->ow[i], hh_ = r->oh[i];
			for (int y = 1; y < hh_ - 1; ++y) for (int x = 1; x < ww_ - 1; ++x)
				if (is_extrema(r, i, j, y, x)) coor_push(&r->raw[i][j], x, y);
			for (int k = 0; k < r->raw[i][j].n; ++k) {
				sspoint_t sp; memset(&sp, 0, sizeof(sp));
				sp.x = r->raw[i][j].xy[2 * k]; sp.y = r->raw[i][j].xy[2 * k + 1];
				sp.pyr = i; sp.scale = j;
				if (!calc_kp_offset(r, &sp)) continue;
				if (is_edge_response(cfg, sp.x, sp.y, &r->dog[i][sp.scale])) continue;
				sp_push(&r->refined, &sp);
			}
		}
	qsort(r->refined.v, r->refined.n, sizeof(sspoint_t), sp_cmp);
	/* feature/orientation.cc:22-32 */
	for (int k = 0; k < r->refined.n; ++k) {
		float dirs[36];
		int nd = calc_dir(r, &r->refined.v[k], dirs);
		for (int d = 0; d < nd; ++d) {
			sspoint_t sp = r->refined.v[k];
			sp.dir = dirs[d];
			sp_push(&r->oriented, &sp);
		}
	}
	/* feature/sift.cc:77-85 */
	r->desc = (float*)malloc(sizeof(float) * 128 * (size_t)(r->oriented.n ? r->oriented.n : 1));
	for (int k = 0; k < r->oriented.n; ++k)
		calc_descriptor(r, &r->oriented.v[k], r->desc + 128 * (size_t)k);
	return r;
}

void orc_sift_free(orc_sift_run* r) {
	if (!r) return;
	free(r->work.p);
	for (int o = 0; o < MAX_OCT; ++o) for (int s = 0; s < MAX_SCALE; ++s) {
		free(r->gauss[o][s].p); free(r->mag[o][s].p); free(r->ort[o][s].p); free(r->dog[o][s].p);
		free(r->raw[o][s].xy);
	}
	free(r->refined.v); free(r->oriented.v); free(r->desc);
	free(r);
}

void orc_sift_working_dims(const orc_sift_run* r, int* h, int* w) { *h = r->work.h; *w = r->work.w; }
void orc_sift_octave_dims(const orc_sift_run* r, int oct, int* h, int* w) { *h = r->oh[oct]; *w = r->ow[oct]; }
int orc_sift_plane(const orc_sift_run* r, int kind, int oct, int s, float* out) {
	const plane_t* m = NULL; int ch = 1;
	if (kind == 4) { m = &r->work; ch = 3; }
	else if (kind == 0) m = &r->gauss[oct][s];
	else if (kind == 1) m = &r->dog[oct][s];
	else if (kind == 2) m = &r->mag[oct][s];
	else if (kind == 3) m = &r->ort[oct][s];
	if (!m || !m->p) return -1;
	memcpy(out, m->p, sizeof(float) * (size_t)m->h * m->w * ch);
	return 0;
}
int orc_sift_raw_count(const orc_sift_run* r, int oct, int s) { return r->raw[oct][s].n; }
void orc_sift_raw(const orc_sift_run* r, int oct, int s, int* xy) {
	memcpy(xy, r->raw[oct][s].xy, sizeof(int) * 2 * r->raw[oct][s].n);
}
int orc_sift_kp_count(const orc_sift_run* r, int which) { return which ? r->oriented.n : r->refined.n; }
void orc_sift_kp(const orc_sift_run* r, int which, int* ints, double* real, float* fl) {
	const spvec_t* v = which ? &r->oriented : &r->refined;
	for (int i = 0; i < v->n; ++i) {
		ints[4 * i] = v->v[i].x; ints[4 * i + 1] = v->v[i].y; ints[4 * i + 2] = v->v[i].pyr; ints[4 * i + 3] = v->v[i].scale;
		real[2 * i] = v->v[i].rx; real[2 * i + 1] = v->v[i].ry;
		fl[2 * i] = which ? v->v[i].dir : 0.f; fl[2 * i + 1] = v->v[i].sf;
	}
}
int orc_sift_desc_count(const orc_sift_run* r) { return r->oriented.n; }
void orc_sift_desc(const orc_sift_run* r, float* desc, double* coor) {
	memcpy(desc, r->desc, sizeof(float) * 128 * (size_t)r->oriented.n);
	for (int i = 0; i < r->oriented.n; ++i) { coor[2 * i] = r->oriented.v[i].rx; coor[2 * i + 1] = r->oriented.v[i].ry; }
}

/* feature/feature.cc:20-28 */
int orc_detect_feature(const orc_sift_cfg* cfg, const float* rgb, int h, int w, float** desc, double** coor) {
	orc_sift_run* r = orc_sift_new(cfg, rgb, h, w);
	int k = r->oriented.n;
	*desc = (float*)malloc(sizeof(float) * 128 * (size_t)(k ? k : 1));
	*coor = (double*)malloc(sizeof(double) * 2 * (size_t)(k ? k : 1));
	orc_sift_desc(r, *desc, *coor);
	for (int i = 0; i < k; ++i) {
		(*coor)[2 * i] = ((*coor)[2 * i] - 0.5) * w;
		(*coor)[2 * i + 1] = ((*coor)[2 * i + 1] - 0.5) * h;
	}
	orc_sift_free(r);
	return k;
}
void orc_free(void* p) { free(p); }

long orc_calc_feature_batch(const orc_sift_cfg* cfg, const float* rgb, int n, int h, int w, int nthreads) {
	long total = 0;
	omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic) reduction(+:total)
	for (int i = 0; i < n; ++i) {
		float* d; double* c;
		total += orc_detect_feature(cfg, rgb + (size_t)i * h * w * 3, h, w, &d, &c);
		free(d); free(c);
	}
	return total;
}

/* libm (the real glibc calls the reference makes) and fast_atan over arrays: the GPU tests
 * compare the device twins in openpano_amd/csrc/devmath.hpp against these on the same box.
 * which: 0 expf, 1 cosf, 2 sinf, 3 hypotf(x,y), 4 fast_atan(y,x)+pi as stored by cal_mag_ort */
void orc_libm_batch(int which, const float* x, const float* y, long n, float* out) {
	for (long i = 0; i < n; ++i) {
		switch (which) {
			case 0: out[i] = expf(x[i]); break;
			case 1: out[i] = cosf(x[i]); break;
			case 2: out[i] = sinf(x[i]); break;
			case 3: out[i] = hypotf(x[i], y[i]); break;
			default: out[i] = (float)(fast_atan(y[i], x[i]) + M_PI); break;
		}
	}
}
