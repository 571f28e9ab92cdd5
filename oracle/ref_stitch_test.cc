// oracle/ref_stitch_test.cc -- TEST INFRASTRUCTURE ONLY.
//
// The drop-in proven on the reference's OWN orchestration: Stitcher::build() (stitch/stitcher.cc:32-64)
// and CylinderStitcher::build() (stitch/cylstitcher.cc:20-29) are compiled twice from the reference
// sources by oracle/apply_hooks.py --
//   Exact*   the reference path on the CPU (its SIFTDetector, TransformEstimation, CylinderWarper,
//            ConnectedImages::blend, CameraEstimator), made deterministic: exact FeatureMatcher
//            instead of the FLANN forest, injected mt19937 seed;
//   Hooked*  the same files with INTEGRATION.md's five construction-site edits, i.e. the HIP library
//            behind the reference's loops (one image per SIFT call, one pair per RANSAC call);
//   Batched* the same files with INTEGRATION.md's batched hooks: calc_feature() is ONE op_sift_batch, the
//            matcher object runs the whole task list through ONE op_match_pairs + ONE op_ransac_pairs, and
//            (hook 6, host-only) estimate_camera() goes through libpano_host.so instead of the reference's
//            CameraEstimator --
// and run on the same image files in one process.  Compared: the "Final Image Size" of
// stitcher_image.cc:124 (canvas dimensions) and every pixel of the panorama (<= 1e-4, north_star).
//   ref_stitch_test <cylinder|camera|camera_ordered|trans> <seed> <multiband> img0.png img1.png ...
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <omp.h>

#include "exact/stitcher.hh"
#include "exact/cylstitcher.hh"
#include "hip/stitcher.hh"
#include "hip/cylstitcher.hh"
#include "hipfast/stitcher.hh"
#include "hipfast/cylstitcher.hh"
#include <chrono>
#include "lib/imgproc.hh"

using namespace pano;

extern "C" {
int ref_config_set(const char* key, float v);
void ref_set_seed(unsigned seed);
void ref_set_threads(int n);
}

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
// libopenpano_ref.so is linked -Bsymbolic (the RNG seam needs it) and keeps its own copy of the
// config:: globals; the orchestration variants compiled into this executable read the executable's
static void set_both(const char* k, float v) {
	if (ref_config_set(k, v) != 0) { printf("unknown config key %s\n", k); exit(2); }
	set_local(k, v);
}

int main(int argc, char** argv) {
	if (argc < 6) { printf("usage: ref_stitch_test <cylinder|camera|camera_ordered|trans> <seed> <multiband> img0 img1 ...\n"); return 2; }
	const std::string mode = argv[1];
	const unsigned seed = (unsigned)atol(argv[2]);
	const int multiband = atoi(argv[3]);
	std::vector<std::string> files(argv + 4, argv + argc);
	// src/config.cfg defaults, through the same float narrowing as init_config (main.cc:237-292)
	const struct { const char* k; float v; } kv[] = {
		{"CYLINDER", 0}, {"ESTIMATE_CAMERA", 1}, {"TRANS", 0}, {"ORDERED_INPUT", 0}, {"CROP", 1}, {"MAX_OUTPUT_SIZE", 8000},
		{"LAZY_READ", 0}, {"FOCAL_LENGTH", 37}, {"SIFT_WORKING_SIZE", 800}, {"NUM_OCTAVE", 4}, {"NUM_SCALE", 7},
		{"SCALE_FACTOR", 1.4142135623f}, {"GAUSS_SIGMA", 1.4142135623f}, {"GAUSS_WINDOW_FACTOR", 6}, {"CONTRAST_THRES", 4e-2f},
		{"JUDGE_EXTREMA_DIFF_THRES", 2e-3f}, {"EDGE_RATIO", 6}, {"PRE_COLOR_THRES", 5e-2f}, {"CALC_OFFSET_DEPTH", 4},
		{"OFFSET_THRES", 0.5f}, {"ORI_RADIUS", 4.5f}, {"ORI_HIST_SMOOTH_COUNT", 2}, {"DESC_HIST_SCALE_FACTOR", 3},
		{"DESC_INT_FACTOR", 512}, {"MATCH_REJECT_NEXT_RATIO", 0.8f}, {"RANSAC_ITERATIONS", 1500}, {"RANSAC_INLIER_THRES", 3.5f},
		{"INLIER_IN_MATCH_RATIO", 0.1f}, {"INLIER_IN_POINTS_RATIO", 0.04f}, {"STRAIGHTEN", 1}, {"SLOPE_PLAIN", 8e-3f},
		{"LM_LAMBDA", 5}, {"MULTIPASS_BA", 1}, {"MULTIBAND", 0}};
	for (auto& e : kv) set_both(e.k, e.v);
	bool cyl = false;
	if (mode == "cylinder") { cyl = true; set_both("CYLINDER", 1); set_both("ESTIMATE_CAMERA", 0); set_both("ORDERED_INPUT", 1); }   // main.cc: CYLINDER implies ordered input
	else if (mode == "camera") { }
	else if (mode == "camera_ordered") { set_both("ORDERED_INPUT", 1); }
	else if (mode == "trans") { set_both("TRANS", 1); set_both("ESTIMATE_CAMERA", 0); set_both("ORDERED_INPUT", 1); }
	else { printf("unknown mode %s\n", mode.c_str()); return 2; }
	set_both("MULTIBAND", (float)multiband);
	// one thread: the reference's keypoint / match orders are thread-timing dependent (extrema.cc:56)
	omp_set_num_threads(1); ref_set_threads(1);
	ref_set_seed(seed);
	HipTransformEstimation::seed_injected() = true; HipTransformEstimation::injected_seed() = seed;

	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	// "time" as the seed argument's suffix (e.g. 38t): the CPU leg runs on all host threads (its outputs then depend
	// on thread timing, extrema.cc:56 -- sizes only are compared) and every variant's build() wall time is printed
	const bool timing = strchr(argv[2], 't') != nullptr;
	if (timing) { omp_set_num_threads(omp_get_num_procs()); ref_set_threads(omp_get_num_procs()); }
	printf("[reference orchestration, CPU, %zu images, mode %s]\n", files.size(), mode.c_str()); fflush(stdout);
	double t0 = now();
	Mat32f want = cyl ? ExactCylinderStitcher(files).build() : ExactStitcher(files).build();
	const double t_cpu = now() - t0;
	int fail = 0;
	auto compare = [&](const char* what, const Mat32f& got) {
		printf("FINAL_SIZE reference %dx%d %s %dx%d\n", want.width(), want.height(), what, got.width(), got.height());
		if (timing) return;      // a multi-threaded CPU leg orders its keypoints by thread timing (extrema.cc:56): its RANSAC draws differ
		if (want.rows() != got.rows() || want.cols() != got.cols()) { printf("FAIL: canvas size differs\n"); ++fail; return; }
		const long n = (long)want.rows() * want.cols();
		long mask_diff = 0, exact = 0, valid = 0; double maxd = 0;
		for (long e = 0; e < n; ++e) {
			const float* p = want.ptr() + e * 3; const float* q = got.ptr() + e * 3;
			const bool na = p[0] < 0, nb = q[0] < 0;
			if (na != nb) { ++mask_diff; continue; }
			if (na) continue;
			++valid;
			bool eq = true;
			for (int c = 0; c < 3; ++c) { maxd = std::max(maxd, (double)fabsf(p[c] - q[c])); eq &= (p[c] == q[c]); }
			exact += eq;
		}
		printf("PANORAMA %s covered %.1f%%, max |diff| %.3g, bit-equal %.4f%%, no-pixel mask flips %ld\n", what, 100.0 * valid / n, maxd,
				100.0 * exact / std::max(1L, valid), mask_diff);
		if (!(maxd <= 1e-4)) { printf("FAIL: max diff %g > 1e-4\n", maxd); ++fail; }
		if (mask_diff > n / 20000 + 2) { printf("FAIL: %ld mask flips\n", mask_diff); ++fail; }
		if (valid < n / 4) { printf("FAIL: canvas barely covered\n"); ++fail; }
		// main.cc:226-229: crop under config CROP -- same rectangle from both
		Mat32f cw = crop(want), cg = crop(got);
		printf("CROPPED reference %dx%d %s %dx%d\n", cw.width(), cw.height(), what, cg.width(), cg.height());
		if (cw.rows() != cg.rows() || cw.cols() != cg.cols()) { printf("FAIL: crop rectangle differs\n"); ++fail; }
	};
	// the device library is warm for the timed runs (context, kernels, pools): one throw-away build in timing mode
	if (timing) { printf("[warm-up of the device library, not timed]\n"); fflush(stdout); if (cyl) BatchedCylinderStitcher(files).build(); else BatchedStitcher(files).build(); }
	printf("[reference orchestration + the five hooks -> libopenpano_hip.so]\n"); fflush(stdout);
	t0 = now();
	Mat32f got = cyl ? HookedCylinderStitcher(files).build() : HookedStitcher(files).build();
	const double t_hook = now() - t0;
	compare("hooked", got);
	printf("[reference orchestration + the batched hooks -> libopenpano_hip.so]\n"); fflush(stdout);
	t0 = now();
	Mat32f got2 = cyl ? BatchedCylinderStitcher(files).build() : BatchedStitcher(files).build();
	const double t_batch = now() - t0;
	compare("batched", got2);
	// Hook 6: the batched variant estimates the cameras with libpano_host.so (HostCameraEstimator, host/pano_camera.hh), the hooked
	// one with the reference's own CameraEstimator / IncrementalBundleAdjuster (over the Eigen stand-in: parity unpinned at Eigen).
	// Same match table in, so the same cameras must come out, digit for digit -- i.e. the two panoramas are bit-identical.
	if (!timing && !cyl && mode != "trans") {
		const bool same = got.rows() == got2.rows() && got.cols() == got2.cols() &&
			memcmp(got.ptr(), got2.ptr(), sizeof(float) * 3 * (size_t)got.rows() * got.cols()) == 0;
		printf("HOST_ESTIMATOR cameras of libpano_host.so == cameras of the reference's CameraEstimator (panoramas bit-identical): %s\n", same ? "yes" : "NO");
		if (!same) { printf("FAIL: the host camera estimation hook changed the panorama\n"); ++fail; }
	}
	printf("DROPIN_MS {\"images\": %zu, \"mode\": \"%s\", \"cpu_threads\": %d, \"reference_cpu_build_ms\": %.1f, \"five_hooks_build_ms\": %.1f, \"batched_hooks_build_ms\": %.1f}\n",
			files.size(), mode.c_str(), timing ? omp_get_num_procs() : 1, t_cpu, t_hook, t_batch);
	printf(fail ? "STITCH DROPIN FAILED\n" : "STITCH DROPIN OK\n");
	return fail ? 1 : 0;
}
