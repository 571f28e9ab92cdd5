// oracle/ref_dropin_test.cc -- TEST INFRASTRUCTURE ONLY.
//
// Drop-in proof: ONE process holds the reference's own classes (compiled in place into
// oracle/_ref/libopenpano_ref.so) and the HIP adapters of openpano_amd/host/pano_hip.hh built
// against the reference's OWN headers (-DOPENPANO_WITH_REFERENCE).  Each stage is run through
// the reference class and through the adapter that would replace it, on the same Mat32f /
// Descriptor / MatchData / ConnectedImages objects, and compared:
//   SIFTDetector::detect_feature      vs HipSIFTDetector::detect_feature   (FeatureDetector hierarchy)  exact
//   FeatureMatcher::match (exact)     vs HipPairWiseMatcher::match                                      exact
//   TransformEstimation::get_transform vs HipTransformEstimation::get_transform (same injected seed)     same inliers
//   ConnectedImages::blend            vs hip_blend(bundle)                                              bit-exact
//   CylinderWarper::warp              vs HipCylinderWarper::warp                                        bit-exact
// Built by oracle/Makefile (target dropin) when /root/reference exists; the binary travels to the
// GPU box and tests/test_gpu_dropin.py runs it.  Exit code 0 = all stages agree.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "pano_hip.hh"
#include "feature/gaussian.hh"
#include "stitch/transform_estimate.hh"
#include "stitch/warp.hh"
#include <omp.h>

using namespace pano;

extern "C" {
int ref_config_set(const char* key, float v);
int ref_ransac(const int* match, int m, const double* kp1, int nk1, const double* kp2, int nk2,
		int w1, int h1, int w2, int h2, unsigned seed, float* confidence, double* homo, double* inlier_pts, int* n_inliers);
}

static int g_fail = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++g_fail; printf("  FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static void set_local(const std::string& k, float v) {
	using namespace config;
#define CFG(x) if (k == #x) { x = v; return; }
	CFG(CYLINDER) CFG(TRANS) CFG(ESTIMATE_CAMERA) CFG(ORDERED_INPUT) CFG(CROP) CFG(STRAIGHTEN)
	CFG(FOCAL_LENGTH) CFG(MAX_OUTPUT_SIZE) CFG(LAZY_READ) CFG(SIFT_WORKING_SIZE) CFG(NUM_OCTAVE)
	CFG(NUM_SCALE) CFG(SCALE_FACTOR) CFG(GAUSS_SIGMA) CFG(GAUSS_WINDOW_FACTOR)
	CFG(JUDGE_EXTREMA_DIFF_THRES) CFG(CONTRAST_THRES) CFG(PRE_COLOR_THRES) CFG(EDGE_RATIO)
	CFG(CALC_OFFSET_DEPTH) CFG(OFFSET_THRES) CFG(ORI_RADIUS) CFG(ORI_HIST_SMOOTH_COUNT)
	CFG(DESC_HIST_SCALE_FACTOR) CFG(DESC_INT_FACTOR) CFG(MATCH_REJECT_NEXT_RATIO)
	CFG(RANSAC_ITERATIONS) CFG(RANSAC_INLIER_THRES) CFG(INLIER_IN_MATCH_RATIO)
	CFG(INLIER_IN_POINTS_RATIO) CFG(SLOPE_PLAIN) CFG(LM_LAMBDA) CFG(MULTIPASS_BA) CFG(MULTIBAND)
#undef CFG
}
static void set_both(const char* k, float v) { ref_config_set(k, v); set_local(k, v); }

// src/config.cfg defaults, through the same float narrowing as init_config (main.cc:237-292)
static void default_config() {
	const struct { const char* k; float v; } kv[] = {
		{"CYLINDER", 0}, {"ESTIMATE_CAMERA", 1}, {"TRANS", 0}, {"ORDERED_INPUT", 0}, {"CROP", 1}, {"MAX_OUTPUT_SIZE", 8000},
		{"LAZY_READ", 0}, {"FOCAL_LENGTH", 37}, {"SIFT_WORKING_SIZE", 800}, {"NUM_OCTAVE", 4}, {"NUM_SCALE", 7},
		{"SCALE_FACTOR", 1.4142135623f}, {"GAUSS_SIGMA", 1.4142135623f}, {"GAUSS_WINDOW_FACTOR", 6}, {"CONTRAST_THRES", 4e-2f},
		{"JUDGE_EXTREMA_DIFF_THRES", 2e-3f}, {"EDGE_RATIO", 6}, {"PRE_COLOR_THRES", 5e-2f}, {"CALC_OFFSET_DEPTH", 4},
		{"OFFSET_THRES", 0.5f}, {"ORI_RADIUS", 4.5f}, {"ORI_HIST_SMOOTH_COUNT", 2}, {"DESC_HIST_SCALE_FACTOR", 3},
		{"DESC_INT_FACTOR", 512}, {"MATCH_REJECT_NEXT_RATIO", 0.8f}, {"RANSAC_ITERATIONS", 1500}, {"RANSAC_INLIER_THRES", 3.5f},
		{"INLIER_IN_MATCH_RATIO", 0.1f}, {"INLIER_IN_POINTS_RATIO", 0.04f}, {"STRAIGHTEN", 1}, {"SLOPE_PLAIN", 8e-3f},
		{"LM_LAMBDA", 5}, {"MULTIPASS_BA", 1}, {"MULTIBAND", 0}};
	for (auto& e : kv) if (ref_config_set(e.k, e.v) != 0) { printf("unknown config key %s\n", e.k); exit(2); }
	// libopenpano_ref.so is linked -Bsymbolic (the RNG seam needs it), so it keeps its own copy of
	// the config:: globals; this executable's copy -- the one the adapters snapshot -- is filled the
	// same way.  In a real integration the adapters are part of the reference build: one copy.
	for (auto& e : kv) set_local(e.k, e.v);
}

// procedural world: blobs + bars over smooth noise (like openpano_amd/synth.py, in C++)
static Mat32f make_world(int h, int w, unsigned seed) {
	std::mt19937 g(seed);
	std::uniform_real_distribution<float> U(0.f, 1.f);
	Mat32f m(h, w, 3);
	for (int i = 0; i < h; ++i) for (int j = 0; j < w; ++j) for (int c = 0; c < 3; ++c)
		m.at(i, j, c) = 0.45f + 0.1f * sinf(0.013f * (i + 3 * c) + 0.021f * j) + 0.05f * cosf(0.05f * j - 0.03f * i);
	const int nblobs = h * w / 60;
	for (int b = 0; b < nblobs; ++b) {
		const float cy = U(g) * h, cx = U(g) * w, sy = 0.8f + 2.2f * U(g), sx = sy * (0.6f + U(g));
		float amp[3]; for (int c = 0; c < 3; ++c) amp[c] = 1.3f * U(g) - 0.65f;
		const int r = (int)(3 * std::max(sx, sy)) + 1;
		for (int i = std::max(0, (int)cy - r); i < std::min(h, (int)cy + r + 1); ++i)
			for (int j = std::max(0, (int)cx - r); j < std::min(w, (int)cx + r + 1); ++j) {
				const float e = expf(-(i - cy) * (i - cy) / (2 * sy * sy) - (j - cx) * (j - cx) / (2 * sx * sx));
				for (int c = 0; c < 3; ++c) m.at(i, j, c) += e * amp[c];
			}
	}
	for (int i = 0; i < h; ++i) for (int j = 0; j < w; ++j) for (int c = 0; c < 3; ++c)
		m.at(i, j, c) = std::min(1.f, std::max(0.f, m.at(i, j, c)));
	return m;
}
static Mat32f cut(const Mat32f& world, int top, int left, int h, int w) {
	Mat32f m(h, w, 3);
	for (int i = 0; i < h; ++i) memcpy(m.ptr(i), world.ptr(top + i) + (size_t)left * 3, sizeof(float) * w * 3);
	return m;
}

static bool desc_less(const Descriptor& a, const Descriptor& b) {
	if (a.coor.y != b.coor.y) return a.coor.y < b.coor.y;
	if (a.coor.x != b.coor.x) return a.coor.x < b.coor.x;
	return a.descriptor < b.descriptor;
}

static void compare_canvas(const Mat32f& a, const Mat32f& b, const char* what) {
	EXPECT(a.rows() == b.rows() && a.cols() == b.cols(), "%s: shape %dx%d vs %dx%d", what, a.rows(), a.cols(), b.rows(), b.cols());
	if (a.rows() != b.rows() || a.cols() != b.cols()) return;
	long n = (long)a.rows() * a.cols(), mask_diff = 0, exact = 0, valid = 0; double maxd = 0;
	for (long e = 0; e < n; ++e) {
		const float* p = a.ptr() + e * 3; const float* q = b.ptr() + e * 3;
		const bool na = p[0] < 0, nb = q[0] < 0;
		if (na != nb) { ++mask_diff; continue; }
		if (na) continue;
		++valid;
		bool eq = true;
		for (int c = 0; c < 3; ++c) { maxd = std::max(maxd, (double)fabsf(p[c] - q[c])); eq &= (p[c] == q[c]); }
		exact += eq;
	}
	printf("  %s: %dx%d, covered %.1f%%, max |diff| %.3g, bit-equal %.4f%%, mask flips %ld\n", what, a.rows(), a.cols(),
			100.0 * valid / n, maxd, 100.0 * exact / std::max(1L, valid), mask_diff);
	// same homographies on both sides and the map's sin / cos / tan from the host libm on both sides (csrc/blend.hip): bit-exact
	EXPECT(maxd == 0, "%s: max diff %g", what, maxd);
	EXPECT(mask_diff == 0, "%s: %ld no-pixel mask flips", what, mask_diff);
	EXPECT(valid > n / 3, "%s: canvas barely covered", what);
}

int main() {
	default_config();
	omp_set_num_threads(1);          // deterministic reference order (extrema.cc:56, multiband.cc:55)
	Mat32f world = make_world(300, 560, 7);
	Mat32f A = cut(world, 20, 20, 240, 320), B = cut(world, 34, 200, 240, 320);

	// ---------------- SIFT: the FeatureDetector hierarchy ----------------
	printf("[sift]\n");
	SIFTDetector ref_det; HipSIFTDetector hip_det;
	const FeatureDetector& base = hip_det;            // used through the reference's base class
	std::vector<std::vector<Descriptor>> rf, hf;
	for (const Mat32f* m : {&A, &B}) {
		rf.push_back(ref_det.detect_feature(*m));
		hf.push_back(base.detect_feature(*m));
		auto r = rf.back(), h = hf.back();
		std::sort(r.begin(), r.end(), desc_less); std::sort(h.begin(), h.end(), desc_less);
		EXPECT(r.size() == h.size() && r.size() > 100, "keypoint count %zu vs %zu", r.size(), h.size());
		size_t bad = 0;
		for (size_t i = 0; i < std::min(r.size(), h.size()); ++i)
			bad += !(r[i].coor.x == h[i].coor.x && r[i].coor.y == h[i].coor.y && r[i].descriptor == h[i].descriptor);
		EXPECT(bad == 0, "%zu descriptors differ", bad);
		printf("  %zu descriptors, identical\n", r.size());
	}
	// batched calc_feature == per-image calls
	{
		HipFeatureSet fs = hip_det.calc_feature({&A, &B});
		for (int k = 0; k < 2; ++k) {
			EXPECT(fs.feats[k].size() == hf[k].size(), "batched count differs");
			size_t bad = 0;
			for (size_t i = 0; i < std::min(fs.feats[k].size(), hf[k].size()); ++i)
				bad += !(fs.feats[k][i].coor.x == hf[k][i].coor.x && fs.feats[k][i].descriptor == hf[k][i].descriptor);
			EXPECT(bad == 0, "batched calc_feature differs from detect_feature (%zu)", bad);
		}

		// ---------------- match ----------------
		printf("[match]\n");
		FeatureMatcher exact(hf[0], hf[1]);
		MatchData want = exact.match();
		std::sort(want.data.begin(), want.data.end());
		HipPairWiseMatcher pw(hf);                     // same ctor as PairWiseMatcher(feats)
		MatchData got = pw.match(0, 1);
		EXPECT(want.data == got.data && want.size() > 20, "match sets differ: %d vs %d", want.size(), got.size());
		HipPairWiseMatcher pw2(fs);                    // device-resident descriptors
		MatchData got2 = pw2.match(0, 1);
		EXPECT(want.data == got2.data, "resident-feature matcher differs");
		MatchData rev = pw.match(1, 0);
		MatchData wrev = want; wrev.reverse(); std::sort(wrev.data.begin(), wrev.data.end());
		EXPECT(rev.data == wrev.data, "match(j, i) is not the reverse of match(i, j)");
		printf("  %d matches, identical to FeatureMatcher::match\n", got.size());

		// ---------------- RANSAC ----------------
		printf("[ransac]\n");
		std::vector<Vec2D> k1, k2;
		for (auto& d : hf[0]) k1.push_back(d.coor);
		for (auto& d : hf[1]) k2.push_back(d.coor);
		std::vector<int> mi; for (auto& p : got.data) { mi.push_back(p.first); mi.push_back(p.second); }
		std::vector<double> c1, c2;
		for (auto& v : k1) { c1.push_back(v.x); c1.push_back(v.y); }
		for (auto& v : k2) { c2.push_back(v.x); c2.push_back(v.y); }
		for (unsigned seed : {1u, 12345u, 777u}) {
			float rconf; double rh[9]; std::vector<double> rpts(got.size() * 4 + 4); int rn = 0;
			const int rok = ref_ransac(mi.data(), got.size(), c1.data(), (int)k1.size(), c2.data(), (int)k2.size(),
					A.width(), A.height(), B.width(), B.height(), seed, &rconf, rh, rpts.data(), &rn);
			HipTransformEstimation::seed_injected() = true; HipTransformEstimation::injected_seed() = seed;
			HipTransformEstimation te(got, k1, k2, Shape2D(A.width(), A.height()), Shape2D(B.width(), B.height()));
			MatchInfo info;
			const bool ok = te.get_transform(&info);
			EXPECT(ok == (bool)rok, "seed %u: ok %d vs %d", seed, (int)ok, rok);
			EXPECT(info.confidence == rconf, "seed %u: confidence %g vs %g", seed, info.confidence, rconf);
			if (ok && rok) {
				EXPECT((int)info.match.size() == rn, "seed %u: %zu vs %d inliers", seed, info.match.size(), rn);
				size_t bad = 0;
				for (int i = 0; i < std::min(rn, (int)info.match.size()); ++i)
					bad += !(info.match[i].first.x == rpts[4 * i] && info.match[i].first.y == rpts[4 * i + 1] &&
							info.match[i].second.x == rpts[4 * i + 2] && info.match[i].second.y == rpts[4 * i + 3]);
				EXPECT(bad == 0, "seed %u: %zu inlier pairs differ", seed, bad);
				double md = 0; for (int i = 0; i < 9; ++i) md = std::max(md, fabs(info.homo[i] - rh[i]) / (fabs(rh[i]) + 1e-3));
				EXPECT(md < 1e-8, "seed %u: homography differs by %g", seed, md);
				printf("  seed %u: ok, %d inliers, confidence %g, homography rel diff %.2g\n", seed, rn, rconf, md);
			}
		}
	}

	// ---------------- warp + blend ----------------
	printf("[blend]\n");
	{
		std::vector<Mat32f> views = {cut(world, 10, 10, 200, 280), cut(world, 22, 150, 200, 280), cut(world, 4, 270, 200, 280)};
		const double f = 0.9 * 280;
		for (int mb : {0, 3}) for (int pm : {0, 2}) {
			set_both("MULTIBAND", (float)mb);
			std::vector<std::unique_ptr<ImageRef>> refs;
			ConnectedImages bundle;
			bundle.proj_method = (ConnectedImages::ProjectionMethod)pm;
			bundle.identity_idx = 1;
			for (int i = 0; i < 3; ++i) {
				refs.emplace_back(new ImageRef("<memory>"));
				refs.back()->img = new Mat32f(views[i].clone());
				refs.back()->_width = views[i].width(); refs.back()->_height = views[i].height();
				bundle.component.emplace_back(refs.back().get());
				Homography& H = bundle.component.back().homo;
				const double dx = (i - 1) * 140.0, dy = (i == 0 ? 0 : (i == 1 ? 12 : -6));
				if (pm == 0) { const double d[9] = {1, 0.01 * (i - 1), dx, -0.01 * (i - 1), 1, dy, 2e-5 * (i - 1), 0, 1}; H = Homography(d); }
				else {
					const double yaw = dx / f, pitch = dy / f, cy = cos(yaw), sy = sin(yaw), cp = cos(pitch), sp = sin(pitch);
					const double d[9] = {cy / f, sy * sp / f, sy * cp, 0, cp / f, -sp, -sy / f, cy * sp / f, cy * cp};
					H = Homography(d);
				}
			}
			bundle.calc_inverse_homo();
			bundle.update_proj_range();
			Mat32f got = hip_blend(bundle);             // what ConnectedImages::blend() would call
			Mat32f want = bundle.blend();               // (multiband releases the images: run it last)
			char what[64]; snprintf(what, sizeof(what), "proj %d multiband %d", pm, mb);
			compare_canvas(want, got, what);
		}
		set_both("MULTIBAND", 0);
		// cylinder pre-warp
		Mat32f m1 = views[0].clone(), m2 = views[0].clone();
		std::vector<Vec2D> p1 = {Vec2D(-100, -50), Vec2D(0, 0), Vec2D(120.5, 77.25)}, p2 = p1;
		CylinderWarper(1).warp(m1, p1);
		HipCylinderWarper(1).warp(m2, p2);
		compare_canvas(m1, m2, "cylinder warp");
		for (size_t i = 0; i < p1.size(); ++i) EXPECT(p1[i].x == p2[i].x && p1[i].y == p2[i].y, "warped keypoint %zu differs", i);
	}
	printf(g_fail ? "DROPIN FAILED (%d)\n" : "DROPIN OK\n", g_fail);
	return g_fail ? 1 : 0;
}
