// oracle/ref_driver.cc -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// extern "C" accessors over the *unmodified* reference classes, compiled together with
// the reference's own translation units (in place, from /root/reference/src) into
// oracle/_ref/libopenpano_ref.so by oracle/Makefile.  Used to
//   (1) pin the plain-C restatement in oracle/sift_oracle.c etc. (bit-exact, per stage),
//   (2) generate the golden vectors under tests/golden/ (tests/golden/make_golden.py),
//   (3) serve as the "reference" CPU baseline in bench.py when the .so travelled.
// Nothing here is reference source: it only *calls* the reference API
// (feature/feature.hh:42-57, feature/dog.hh, feature/extrema.hh, feature/orientation.hh,
//  feature/sift.hh, feature/matcher.hh:31-51, lib/config.hh:24-85).
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <algorithm>
#include <omp.h>
#include <random>
#include <sstream>

#include "lib/config.hh"
#include "lib/mat.h"
#include "lib/imgproc.hh"
#include "feature/feature.hh"
#include "feature/dog.hh"
#include "feature/extrema.hh"
#include "feature/orientation.hh"
#include "feature/sift.hh"
#include "feature/matcher.hh"
#include "feature/gaussian.hh"
#include "stitch/transform_estimate.hh"
#include "stitch/match_info.hh"
#include "stitch/stitcher_image.hh"
#include "stitch/warp.hh"
#include "stitch/camera.hh"
#include "stitch/camera_estimator.hh"
#include "stitch/incremental_bundle_adjuster.hh"
#include <Eigen/Dense>

using namespace pano;
using namespace config;

// RNG injection seam: TransformEstimation::get_transform seeds std::mt19937 from
// std::random_device (transform_estimate.cc:64-65).  The reference TU calls the out-of-line
// libstdc++ member _M_getval(); this definition (bound inside this .so by -Wl,-Bsymbolic)
// returns the seed chosen by the test instead of entropy.
static unsigned g_ref_seed = 0;
namespace std { unsigned int random_device::_M_getval() { return g_ref_seed; } }

namespace {

struct ExtremaPub : public ExtremaDetector {
	explicit ExtremaPub(const DOGSpace& d): ExtremaDetector(d) {}
	using ExtremaDetector::get_local_raw_extrema;
};

struct SiftRun {
	Mat32f resized;
	std::unique_ptr<ScaleSpace> ss;
	std::unique_ptr<DOGSpace> dog;
	std::vector<std::vector<Coor>> raw;	// [octave*(nscale-3) + (scale-1)]
	std::vector<SSPoint> refined, oriented;
	std::vector<Descriptor> desc;
};

Mat32f wrap_rgb(const float* rgb, int h, int w) {
	Mat32f m(h, w, 3);
	memcpy(m.ptr(), rgb, sizeof(float) * (size_t)h * w * 3);
	return m;
}

bool sspoint_less(const SSPoint& a, const SSPoint& b) {
	if (a.pyr_id != b.pyr_id) return a.pyr_id < b.pyr_id;
	if (a.scale_id != b.scale_id) return a.scale_id < b.scale_id;
	if (a.coor.y != b.coor.y) return a.coor.y < b.coor.y;
	if (a.coor.x != b.coor.x) return a.coor.x < b.coor.x;
	if (a.real_coor.x != b.real_coor.x) return a.real_coor.x < b.real_coor.x;
	return a.real_coor.y < b.real_coor.y;
}

}	// namespace

extern "C" {

// mirrors init_config() (main.cc:237-292): every key assigned from a float
int ref_config_set(const char* key, float v) {
	std::string k(key);
#define CFG(x) if (k == #x) { x = v; return 0; }
	CFG(CYLINDER) CFG(TRANS) CFG(ESTIMATE_CAMERA) CFG(ORDERED_INPUT) CFG(CROP) CFG(STRAIGHTEN)
	CFG(FOCAL_LENGTH) CFG(MAX_OUTPUT_SIZE) CFG(LAZY_READ) CFG(SIFT_WORKING_SIZE) CFG(NUM_OCTAVE)
	CFG(NUM_SCALE) CFG(SCALE_FACTOR) CFG(GAUSS_SIGMA) CFG(GAUSS_WINDOW_FACTOR)
	CFG(JUDGE_EXTREMA_DIFF_THRES) CFG(CONTRAST_THRES) CFG(PRE_COLOR_THRES) CFG(EDGE_RATIO)
	CFG(CALC_OFFSET_DEPTH) CFG(OFFSET_THRES) CFG(ORI_RADIUS) CFG(ORI_HIST_SMOOTH_COUNT)
	CFG(DESC_HIST_SCALE_FACTOR) CFG(DESC_INT_FACTOR) CFG(MATCH_REJECT_NEXT_RATIO)
	CFG(RANSAC_ITERATIONS) CFG(RANSAC_INLIER_THRES) CFG(INLIER_IN_MATCH_RATIO)
	CFG(INLIER_IN_POINTS_RATIO) CFG(SLOPE_PLAIN) CFG(LM_LAMBDA) CFG(MULTIPASS_BA) CFG(MULTIBAND)
#undef CFG
	return -1;
}

void ref_set_threads(int n) { omp_set_num_threads(n); }
// seed every TransformEstimation::get_transform of this library draws from now on (the random_device seam above)
void ref_set_seed(unsigned seed) { g_ref_seed = seed; }

// ---- staged SIFT run (body of SIFTDetector::do_detect_feature, feature.cc:31-47,
//      with every intermediate kept) ----
void* ref_sift_new(const float* rgb, int h, int w) {
	SiftRun* r = new SiftRun;
	Mat32f mat = wrap_rgb(rgb, h, w);
	float ratio = SIFT_WORKING_SIZE * 2.0f / (mat.width() + mat.height());
	r->resized = Mat32f(mat.rows() * ratio, mat.cols() * ratio, 3);
	resize(mat, r->resized);
	r->ss.reset(new ScaleSpace(r->resized, NUM_OCTAVE, NUM_SCALE));
	r->dog.reset(new DOGSpace(*r->ss));
	ExtremaPub ex(*r->dog);
	for (int i = 0; i < NUM_OCTAVE; ++i)
		for (int j = 1; j < NUM_SCALE - 2; ++j)
			r->raw.emplace_back(ex.get_local_raw_extrema(i, j));
	r->refined = ex.get_extrema();
	// reference order is thread-timing dependent (extrema.cc:56) -> canonicalise
	std::sort(r->refined.begin(), r->refined.end(), sspoint_less);
	OrientationAssign ort(*r->dog, *r->ss, r->refined);
	r->oriented = ort.work();
	SIFT sift(*r->ss, r->oriented);
	r->desc = sift.get_descriptor();
	return r;
}

void ref_sift_free(void* hd) { delete (SiftRun*)hd; }

void ref_sift_working_dims(void* hd, int* h, int* w) {
	SiftRun* r = (SiftRun*)hd;
	*h = r->resized.height(); *w = r->resized.width();
}

void ref_sift_octave_dims(void* hd, int oct, int* h, int* w) {
	SiftRun* r = (SiftRun*)hd;
	*h = r->ss->pyramids[oct].h; *w = r->ss->pyramids[oct].w;
}

// kind: 0 gaussian stack data[s] (s=0 is the grey base), 1 DoG[s], 2 mag[s], 3 ort[s], 4 working RGB
int ref_sift_plane(void* hd, int kind, int oct, int s, float* out) {
	SiftRun* r = (SiftRun*)hd;
	const Mat32f* m = nullptr;
	if (kind == 4) m = &r->resized;
	else if (kind == 0) m = &r->ss->pyramids[oct].get(s);
	else if (kind == 1) m = &r->dog->dogs[oct][s];
	else if (kind == 2) m = &r->ss->pyramids[oct].get_mag(s);
	else if (kind == 3) m = &r->ss->pyramids[oct].get_ort(s);
	if (!m || m->rows() == 0) return -1;
	memcpy(out, m->ptr(), sizeof(float) * (size_t)m->rows() * m->cols() * m->channels());
	return 0;
}

int ref_sift_raw_count(void* hd, int oct, int s) {
	SiftRun* r = (SiftRun*)hd;
	return (int)r->raw[oct * (NUM_SCALE - 3) + (s - 1)].size();
}
void ref_sift_raw(void* hd, int oct, int s, int* xy) {
	SiftRun* r = (SiftRun*)hd;
	auto& v = r->raw[oct * (NUM_SCALE - 3) + (s - 1)];
	for (size_t i = 0; i < v.size(); ++i) { xy[2 * i] = v[i].x; xy[2 * i + 1] = v[i].y; }
}

// which: 0 refined (after calc_kp_offset + edge test), 1 oriented
int ref_sift_kp_count(void* hd, int which) {
	SiftRun* r = (SiftRun*)hd;
	return (int)(which ? r->oriented.size() : r->refined.size());
}
// ints: x, y, pyr_id, scale_id ; real: real_coor.x, real_coor.y ; fl: dir, scale_factor
void ref_sift_kp(void* hd, int which, int* ints, double* real, float* fl) {
	SiftRun* r = (SiftRun*)hd;
	auto& v = which ? r->oriented : r->refined;
	for (size_t i = 0; i < v.size(); ++i) {
		ints[4 * i] = v[i].coor.x; ints[4 * i + 1] = v[i].coor.y;
		ints[4 * i + 2] = v[i].pyr_id; ints[4 * i + 3] = v[i].scale_id;
		real[2 * i] = v[i].real_coor.x; real[2 * i + 1] = v[i].real_coor.y;
		fl[2 * i] = which ? v[i].dir : 0.f; fl[2 * i + 1] = v[i].scale_factor;
	}
}
int ref_sift_desc_count(void* hd) { return (int)((SiftRun*)hd)->desc.size(); }
void ref_sift_desc(void* hd, float* desc, double* coor) {
	SiftRun* r = (SiftRun*)hd;
	for (size_t i = 0; i < r->desc.size(); ++i) {
		memcpy(desc + 128 * i, r->desc[i].descriptor.data(), 128 * sizeof(float));
		coor[2 * i] = r->desc[i].coor.x; coor[2 * i + 1] = r->desc[i].coor.y;
	}
}

// ---- the public entry point: FeatureDetector::detect_feature (feature.cc:20-28) ----
// Two-phase: returns a handle holding the vector<Descriptor>.
void* ref_detect_feature(const float* rgb, int h, int w) {
	SIFTDetector det;
	auto* v = new std::vector<Descriptor>(det.detect_feature(wrap_rgb(rgb, h, w)));
	return v;
}
int ref_features_count(void* hd) { return (int)((std::vector<Descriptor>*)hd)->size(); }
void ref_features_get(void* hd, float* desc, double* coor) {
	auto& v = *(std::vector<Descriptor>*)hd;
	for (size_t i = 0; i < v.size(); ++i) {
		memcpy(desc + 128 * i, v[i].descriptor.data(), 128 * sizeof(float));
		coor[2 * i] = v[i].coor.x; coor[2 * i + 1] = v[i].coor.y;
	}
}
void ref_features_free(void* hd) { delete (std::vector<Descriptor>*)hd; }

// body of StitcherBase::calc_feature() (stitcherbase.cc:14-25) minus ImageRef::load:
// omp-parallel loop of detect_feature over n images of equal size. Returns total #descriptors.
long ref_calc_feature_batch(const float* rgb, int n, int h, int w, int nthreads) {
	long total = 0;
	SIFTDetector det;
	omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic) reduction(+:total)
	for (int i = 0; i < n; ++i) {
		auto v = det.detect_feature(wrap_rgb(rgb + (size_t)i * h * w * 3, h, w));
		total += (long)v.size();
	}
	return total;
}

// ---- exact matcher: FeatureMatcher::match (matcher.cc:15-71) ----
static std::vector<Descriptor> wrap_desc(const float* d, int n) {
	std::vector<Descriptor> v(n);
	for (int i = 0; i < n; ++i) v[i].descriptor.assign(d + 128 * (size_t)i, d + 128 * (size_t)(i + 1));
	return v;
}
// out must hold 2*min(n1,n2) ints; returns #matches, pairs sorted by (first, second)
int ref_match_exact(const float* d1, int n1, const float* d2, int n2, int* out) {
	auto f1 = wrap_desc(d1, n1), f2 = wrap_desc(d2, n2);
	FeatureMatcher m(f1, f2);
	auto md = m.match();
	std::sort(md.data.begin(), md.data.end());
	for (size_t i = 0; i < md.data.size(); ++i) { out[2 * i] = md.data[i].first; out[2 * i + 1] = md.data[i].second; }
	return md.size();
}
// PairWiseMatcher (FLANN kd-forest; approximate, non-deterministic: SURVEY F2/F3) over 2 images
int ref_match_flann(const float* d1, int n1, const float* d2, int n2, int* out) {
	std::vector<std::vector<Descriptor>> feats;
	feats.emplace_back(wrap_desc(d1, n1));
	feats.emplace_back(wrap_desc(d2, n2));
	PairWiseMatcher pw(feats);
	auto md = pw.match(0, 1);
	std::sort(md.data.begin(), md.data.end());
	for (size_t i = 0; i < md.data.size(); ++i) { out[2 * i] = md.data[i].first; out[2 * i + 1] = md.data[i].second; }
	return md.size();
}

// The match loop of Stitcher::pairwise_match (stitch/stitcher.cc:96-113) on the host cores:
// omp-parallel over the pair list with (use_flann=1) the shipped PairWiseMatcher incl. its
// kd-forest build, or (0) the exact FeatureMatcher.  Returns the total number of matches.
long ref_match_pairs_batch(const float* desc, const int* counts, int n, const int* pairs, int npairs, int nthreads, int use_flann) {
	std::vector<std::vector<Descriptor>> feats(n);
	size_t off = 0;
	for (int i = 0; i < n; ++i) { feats[i] = wrap_desc(desc + off * 128, counts[i]); off += counts[i]; }
	omp_set_num_threads(nthreads);
	long total = 0;
	if (use_flann) {
		PairWiseMatcher pw(feats);
#pragma omp parallel for schedule(dynamic) reduction(+:total)
		for (int p = 0; p < npairs; ++p) total += pw.match(pairs[2 * p], pairs[2 * p + 1]).size();
	} else {
#pragma omp parallel for schedule(dynamic) reduction(+:total)
		for (int p = 0; p < npairs; ++p) {
			FeatureMatcher m(feats[pairs[2 * p]], feats[pairs[2 * p + 1]]);
			total += m.match().size();
		}
	}
	return total;
}

// TransformEstimation(...).get_transform(&info) (stitch/transform_estimate.cc:26-87) with an injected
// seed. inlier_pts receives info.match as (to.x, to.y, from.x, from.y) rows. Returns the bool.
int ref_ransac(const int* match, int m, const double* kp1, int nk1, const double* kp2, int nk2,
		int w1, int h1, int w2, int h2, unsigned seed,
		float* confidence, double* homo, double* inlier_pts, int* n_inliers) {
	MatchData md;
	for (int i = 0; i < m; ++i) md.data.emplace_back(match[2 * i], match[2 * i + 1]);
	std::vector<Vec2D> k1, k2;
	for (int i = 0; i < nk1; ++i) k1.emplace_back(kp1[2 * i], kp1[2 * i + 1]);
	for (int i = 0; i < nk2; ++i) k2.emplace_back(kp2[2 * i], kp2[2 * i + 1]);
	g_ref_seed = seed;
	MatchInfo info;
	TransformEstimation te(md, k1, k2, Shape2D(w1, h1), Shape2D(w2, h2));
	bool ok = te.get_transform(&info);
	*confidence = info.confidence;
	*n_inliers = 0;
	if (ok) {
		for (int i = 0; i < 9; ++i) homo[i] = info.homo[i];
		*n_inliers = (int)info.match.size();
		for (size_t i = 0; i < info.match.size(); ++i) {
			inlier_pts[4 * i] = info.match[i].first.x; inlier_pts[4 * i + 1] = info.match[i].first.y;
			inlier_pts[4 * i + 2] = info.match[i].second.x; inlier_pts[4 * i + 3] = info.match[i].second.y;
		}
	}
	return ok ? 1 : 0;
}

// The RANSAC half of the pair loop of Stitcher::pairwise_match (stitch/stitcher.cc:100-113: one TransformEstimation per
// matched pair inside `omp parallel for schedule(dynamic)`) over GIVEN match lists on the host cores -- the CPU baseline of
// op_ransac_pairs (bench.py).  match: the pairs' <first, second> lists back to back (mcount[p] entries each); coor: the
// images' keypoint coordinates back to back (kcount[i] points each).  Every estimation draws from the same injected seed
// (the seam above is one global): this entry is for timing, not for parity.  Returns the number of accepted pairs.
long ref_ransac_pairs_batch(const int* match, const int* mcount, int npairs, const int* pairs, const double* coor, const int* kcount, int n,
		const int* shapes_wh, int nthreads, unsigned seed, long* inliers_total) {
	std::vector<std::vector<Vec2D>> kp(n);
	size_t off = 0;
	for (int i = 0; i < n; ++i) { for (int k = 0; k < kcount[i]; ++k, ++off) kp[i].emplace_back(coor[2 * off], coor[2 * off + 1]); }
	std::vector<MatchData> md(npairs);
	size_t at = 0;
	for (int p = 0; p < npairs; ++p) for (int k = 0; k < mcount[p]; ++k, ++at) md[p].data.emplace_back(match[2 * at], match[2 * at + 1]);
	g_ref_seed = seed;
	omp_set_num_threads(nthreads);
	long ok = 0, inl = 0;
#pragma omp parallel for schedule(dynamic) reduction(+:ok, inl)
	for (int p = 0; p < npairs; ++p) {
		const int i = pairs[2 * p], j = pairs[2 * p + 1];
		MatchInfo info;
		TransformEstimation te(md[p], kp[i], kp[j], Shape2D(shapes_wh[2 * i], shapes_wh[2 * i + 1]), Shape2D(shapes_wh[2 * j], shapes_wh[2 * j + 1]));
		if (te.get_transform(&info)) { ++ok; inl += (long)info.match.size(); }
	}
	if (inliers_total) *inliers_total = inl;
	return ok;
}

// CameraEstimator{pairwise_matches, shapes}.estimate() (stitch/camera_estimator.cc:31-103, the
// body of Stitcher::estimate_camera, stitcher.cc:146-158) on a given pairwise MatchInfo table.
// entries: np directed entries (i, j) -> pairwise_matches[i][j] = {conf, homo, pts}; pts rows are
// (to.x, to.y, from.x, from.y).  out: per image focal, aspect, ppx, ppy, R[9] (13 doubles).
// Also exposes the helpers for unit checks.
int ref_estimate_cameras(int n, const int* shapes_wh, int np, const int* ij, const float* conf, const double* homo,
		const int* cnt, const double* pts, double* out) {
	std::vector<std::vector<MatchInfo>> pm(n, std::vector<MatchInfo>(n));
	size_t at = 0;
	for (int e = 0; e < np; ++e) {
		MatchInfo& m = pm[ij[2 * e]][ij[2 * e + 1]];
		m.confidence = conf[e];
		for (int k = 0; k < 9; ++k) m.homo[k] = homo[9 * e + k];
		for (int k = 0; k < cnt[e]; ++k, ++at)
			m.match.emplace_back(Vec2D(pts[4 * at], pts[4 * at + 1]), Vec2D(pts[4 * at + 2], pts[4 * at + 3]));
	}
	std::vector<Shape2D> shapes;
	for (int i = 0; i < n; ++i) shapes.emplace_back(shapes_wh[2 * i], shapes_wh[2 * i + 1]);
	std::vector<Camera> cams = CameraEstimator{pm, shapes}.estimate();
	for (int i = 0; i < n; ++i) {
		double* o = out + 13 * i;
		o[0] = cams[i].focal; o[1] = cams[i].aspect; o[2] = cams[i].ppx; o[3] = cams[i].ppy;
		for (int k = 0; k < 9; ++k) o[4 + k] = cams[i].R[k];
	}
	return 0;
}
// One Levenberg-Marquardt step of the bundle adjuster on given cameras, with its internals
// exposed: residuals (calcError), damped JtJ and the parameter update (get_param_update) --
// IncrementalBundleAdjuster's protected members reached through a subclass.
struct IbaProbe : public IncrementalBundleAdjuster {
	using IncrementalBundleAdjuster::IncrementalBundleAdjuster;
	void probe(int identity, double* resid, double* jtj, double* upd) {
		set_identity_idx(identity);
		update_index_map();
		int nr_img = idx_added.size();
		J = Eigen::MatrixXd{2 * nr_pointwise_match, 6 * nr_img};
		JtJ = Eigen::MatrixXd{6 * nr_img, 6 * nr_img};
		ParamState state;
		for (auto& idx : idx_added) state.cameras.emplace_back(result_cameras[idx]);
		state.ensure_params();
		state.cameras.clear();
		auto err = calcError(state);
		for (size_t i = 0; i < err.residuals.size(); ++i) resid[i] = err.residuals[i];
		Eigen::VectorXd u = get_param_update(state, err.residuals, LM_LAMBDA);
		for (int i = 0; i < 36 * nr_img * nr_img; ++i) jtj[i] = JtJ.p[i];
		for (int i = 0; i < 6 * nr_img; ++i) upd[i] = u(i);
	}
};
int ref_iba_probe(int n, const double* cams, int np, const int* ij, const int* cnt, const double* pts, int identity,
		double* resid, double* jtj, double* upd) {
	std::vector<Camera> cameras(n);
	for (int i = 0; i < n; ++i) {
		const double* o = cams + 13 * i;
		cameras[i].focal = o[0]; cameras[i].aspect = o[1]; cameras[i].ppx = o[2]; cameras[i].ppy = o[3];
		for (int k = 0; k < 9; ++k) cameras[i].R[k] = o[4 + k];
	}
	std::vector<MatchInfo> infos(np);
	size_t at = 0;
	for (int e = 0; e < np; ++e)
		for (int k = 0; k < cnt[e]; ++k, ++at)
			infos[e].match.emplace_back(Vec2D(pts[4 * at], pts[4 * at + 1]), Vec2D(pts[4 * at + 2], pts[4 * at + 3]));
	IbaProbe iba(cameras);
	for (int e = 0; e < np; ++e) iba.add_match(ij[2 * e], ij[2 * e + 1], infos[e]);
	iba.probe(identity, resid, jtj, upd);
	return 0;
}
void ref_rotation_to_angle(const double* r, double* v) {
	Homography h; for (int k = 0; k < 9; ++k) h[k] = r[k];
	Camera::rotation_to_angle(h, v[0], v[1], v[2]);
}
void ref_angle_to_rotation(const double* v, double* r) {
	Homography h; Camera::angle_to_rotation(v[0], v[1], v[2], h);
	for (int k = 0; k < 9; ++k) r[k] = h[k];
}
int ref_homography_inverse(const double* a, double* inv) {
	Homography h; for (int k = 0; k < 9; ++k) h[k] = a[k];
	bool ok = false;
	Homography r = h.inverse(&ok);
	if (ok) for (int k = 0; k < 9; ++k) inv[k] = r[k];
	return ok ? 1 : 0;
}
// solve through the stand-in ColPivHouseholderQR (what IBA::get_param_update calls)
void ref_colpiv_solve(const double* A, int n, const double* b, double* x) {
	Eigen::MatrixXd M(n, n); Eigen::VectorXd B(n);
	for (int i = 0; i < n * n; ++i) M.p[i] = A[i];
	for (int i = 0; i < n; ++i) B(i) = b[i];
	Eigen::VectorXd X = M.colPivHouseholderQr().solve(B).eval();
	for (int i = 0; i < n; ++i) x[i] = X(i);
}

float ref_euclidean_sqr(const float* x, const float* y, int n, float thres) {
	return pano::euclidean_sqr(x, y, n, thres);
}

// GaussCache weights (gaussian.cc:17-40); out needs >= kw floats, returns kw
int ref_gauss_kernel(float sigma, float* out) {
	GaussCache g(sigma);
	for (int i = 0; i < g.kw; ++i) out[i] = g.kernel[i - g.kw / 2];
	return g.kw;
}

// ---- ConnectedImages::blend (stitch/stitcher_image.cc:116-155) on in-memory images ----
// homo: n x 9 (ImageComponent::homo).  Runs calc_inverse_homo, update_proj_range, blend with the
// blender the config selects (MULTIBAND / LAZY_READ / ORDERED_INPUT globals).
struct BlendRun {
	std::vector<std::unique_ptr<ImageRef>> refs;
	ConnectedImages bundle;
	Mat32f result;
	Vec2D resolution;
	double blend_seconds = 0;      // wall time of ConnectedImages::blend() alone (bench.py's CPU baseline of op_blend)
};
void* ref_blend_new(int proj_method, int identity_idx, int n, const float* const* rgb, const int* hw, const double* homo) {
	BlendRun* r = new BlendRun;
	r->bundle.proj_method = (ConnectedImages::ProjectionMethod)proj_method;
	r->bundle.identity_idx = identity_idx;
	for (int i = 0; i < n; ++i) {
		r->refs.emplace_back(new ImageRef("<memory>"));
		ImageRef* ir = r->refs.back().get();
		ir->img = new Mat32f(wrap_rgb(rgb[i], hw[2 * i], hw[2 * i + 1]));	// load() is then a no-op (imageref.hh:25-26)
		ir->_width = hw[2 * i + 1]; ir->_height = hw[2 * i];
		r->bundle.component.emplace_back(ir);
		for (int k = 0; k < 9; ++k) r->bundle.component.back().homo[k] = homo[9 * i + k];
	}
	r->bundle.calc_inverse_homo();
	r->bundle.update_proj_range();
	r->resolution = r->bundle.get_final_resolution();
	const double t0 = omp_get_wtime();
	r->result = r->bundle.blend();
	r->blend_seconds = omp_get_wtime() - t0;
	return r;
}
double ref_blend_seconds(void* hd) { return ((BlendRun*)hd)->blend_seconds; }
void ref_blend_dims(void* hd, int* h, int* w) { BlendRun* r = (BlendRun*)hd; *h = r->result.height(); *w = r->result.width(); }
void ref_blend_get(void* hd, float* out) {
	BlendRun* r = (BlendRun*)hd;
	memcpy(out, r->result.ptr(), sizeof(float) * (size_t)r->result.height() * r->result.width() * 3);
}
// geom: proj_min.xy, proj_max.xy, resolution.xy ; ranges: n x 4 ; homo_inv: n x 9
void ref_blend_meta(void* hd, double* geom, double* ranges, double* homo_inv) {
	BlendRun* r = (BlendRun*)hd;
	geom[0] = r->bundle.proj_range.min.x; geom[1] = r->bundle.proj_range.min.y;
	geom[2] = r->bundle.proj_range.max.x; geom[3] = r->bundle.proj_range.max.y;
	geom[4] = r->resolution.x; geom[5] = r->resolution.y;
	for (size_t i = 0; i < r->bundle.component.size(); ++i) {
		auto& c = r->bundle.component[i];
		ranges[4 * i] = c.range.min.x; ranges[4 * i + 1] = c.range.min.y;
		ranges[4 * i + 2] = c.range.max.x; ranges[4 * i + 3] = c.range.max.y;
		for (int k = 0; k < 9; ++k) homo_inv[9 * i + k] = c.homo_inv[k];
	}
}
void ref_blend_free(void* hd) { delete (BlendRun*)hd; }

// ---- CylinderWarper::warp (stitch/warp.hh:47-55) ----
struct CylRun { Mat32f mat; std::vector<Vec2D> pts; };
void* ref_cyl_warp_new(const float* rgb, int h, int w, double h_factor, const double* pts, int npts) {
	CylRun* r = new CylRun;
	r->mat = wrap_rgb(rgb, h, w);
	for (int i = 0; i < npts; ++i) r->pts.emplace_back(pts[2 * i], pts[2 * i + 1]);
	CylinderWarper warper(h_factor);
	warper.warp(r->mat, r->pts);
	return r;
}
void ref_cyl_warp_dims(void* hd, int* h, int* w) { CylRun* r = (CylRun*)hd; *h = r->mat.height(); *w = r->mat.width(); }
void ref_cyl_warp_get(void* hd, float* out, double* pts) {
	CylRun* r = (CylRun*)hd;
	memcpy(out, r->mat.ptr(), sizeof(float) * (size_t)r->mat.height() * r->mat.width() * 3);
	for (size_t i = 0; i < r->pts.size(); ++i) { pts[2 * i] = r->pts[i].x; pts[2 * i + 1] = r->pts[i].y; }
}
void ref_cyl_warp_free(void* hd) { delete (CylRun*)hd; }

// crop(mat) (lib/imgproc.cc:200-235): returns the cropped dims; out (if non-null) receives the pixels
void ref_crop(const float* rgb, int h, int w, int* ch, int* cw, float* out) {
	Mat32f r = crop(wrap_rgb(rgb, h, w));
	*ch = r.height(); *cw = r.width();
	if (out) memcpy(out, r.ptr(), sizeof(float) * (size_t)r.height() * r.width() * 3);
}

// MatchInfo::serialize (stitch/match_info.hh:26-36): text into buf (cap bytes), returns length
int ref_matchinfo_serialize(float confidence, const double* homo, const double* pts, int n, char* buf, int cap) {
	MatchInfo info;
	info.confidence = confidence;
	for (int i = 0; i < 9; ++i) info.homo[i] = homo[i];
	for (int i = 0; i < n; ++i) info.match.emplace_back(Vec2D(pts[4 * i], pts[4 * i + 1]), Vec2D(pts[4 * i + 2], pts[4 * i + 3]));
	std::ostringstream os;
	info.serialize(os);
	std::string t = os.str();
	if ((int)t.size() + 1 > cap) return -1;
	memcpy(buf, t.c_str(), t.size() + 1);
	return (int)t.size();
}

}	// extern "C"
