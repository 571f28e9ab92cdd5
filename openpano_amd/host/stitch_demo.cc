// stitch_demo.cc -- standalone C++ host program over the adapters of pano_hip.hh (no reference
// tree, no Python): the hot path in the order Stitcher::build() runs it (stitch/stitcher.cc:32-64),
//   calc_feature -> pairwise_match (+ RANSAC per pair) -> [host camera estimation: out of scope,
//   replaced here by chaining the pairwise homographies to the middle image] -> blend,
// written with the reference's class and method names.
//
//   stitch_demo <in.bin> <out.bin> [base_seed] [camera|camera_ordered]
// ("camera_ordered": ORDERED_INPUT 1 as in BASELINE configs 2-3 -- only (i, i+1 mod n) are matched,
// Stitcher::linear_pairwise_match, stitcher.cc:115-136; head and tail need not connect.)
// With "camera" the program runs the ESTIMATE_CAMERA branch of Stitcher::build() instead (BASELINE
// configs 2-4): all pairs matched, homography RANSAC, host camera estimation + bundle adjustment
// (pano_camera.hh), spherical blend; out.bin then carries n*13 f64 cameras (focal, aspect, ppx,
// ppy, R) in place of the chain homographies, before the canvas.
// in.bin : int32 n, h, w ; n*h*w*3 float32 (Mat32f layout)
// out.bin: per image   int32 K ; K*128 f32 ; K*2 f64
//          int32 npairs ; per pair int32 i, j, M ; M*2 int32 ; int32 ok ; f32 confidence ; 9 f64 ; int32 ninl ; ninl*4 f64
//          int32 H, W ; H*W*3 f32 (flat-projection linear blend of the chain that connected; H=W=0 if none)
// tests/test_gpu_host_cpp.py runs it on the GPU box and checks every section against the oracle.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "pano_hip.hh"

using namespace pano;

template <typename T> static void put(FILE* f, const T* p, size_t n) { if (n && fwrite(p, sizeof(T), n, f) != n) { perror("write"); exit(1); } }
template <typename T> static void put1(FILE* f, T v) { put(f, &v, 1); }

// 3x3 product (Homography::operator*, stitch/homography.cc:41-48)
static Homography mul(const Homography& a, const Homography& b) {
	Homography r;
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
		double s = 0;
		for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
		r[i * 3 + j] = s;
	}
	return r;
}

static void put_features(FILE* fo, const HipFeatureSet& fs) {
	for (size_t k = 0; k < fs.feats.size(); ++k) {
		put1<int32_t>(fo, (int32_t)fs.feats[k].size());
		for (auto& d : fs.feats[k]) put(fo, d.descriptor.data(), 128);
		for (auto& d : fs.feats[k]) { double c[2] = {d.coor.x, d.coor.y}; put(fo, c, 2); }
		fprintf(stderr, "Image %zu has %zu features\n", k, fs.feats[k].size());
	}
}
static void put_pairs(FILE* fo, const HipFeatureSet& fs, const std::vector<std::pair<int, int>>& tasks,
		const std::vector<std::pair<bool, MatchInfo>>& infos) {
	PairWiseMatcher pwmatcher(fs);
	pwmatcher.precompute(tasks);
	put1<int32_t>(fo, (int32_t)tasks.size());
	for (size_t p = 0; p < tasks.size(); ++p) {
		MatchData md = pwmatcher.match(tasks[p].first, tasks[p].second);
		put1<int32_t>(fo, tasks[p].first); put1<int32_t>(fo, tasks[p].second); put1<int32_t>(fo, md.size());
		for (auto& q : md.data) { int32_t v[2] = {q.first, q.second}; put(fo, v, 2); }
		const MatchInfo& info = infos[p].second;
		put1<int32_t>(fo, infos[p].first ? 1 : 0); put1<float>(fo, info.confidence);
		double hh[9]; for (int i = 0; i < 9; ++i) hh[i] = infos[p].first ? info.homo[i] : 0.0;
		put(fo, hh, 9);
		put1<int32_t>(fo, (int32_t)(infos[p].first ? info.match.size() : 0));
		if (infos[p].first) for (auto& m : info.match) { double v[4] = {m.first.x, m.first.y, m.second.x, m.second.y}; put(fo, v, 4); }
		fprintf(stderr, "pair (%d,%d): %d matches, %s, confidence %g\n", tasks[p].first, tasks[p].second, md.size(),
				infos[p].first ? "connected" : "rejected", info.confidence);
	}
}

// Stitcher::build() under ESTIMATE_CAMERA (stitch/stitcher.cc:32-64), stage by stage so that every
// intermediate can be written out
static int run_camera_mode(const std::vector<Mat32f>& mats, const char* out_path, uint32_t base_seed, bool ordered) {
	Stitcher st(mats, base_seed);
	FILE* fo = fopen(out_path, "wb");
	if (!fo) { perror(out_path); return 2; }
	st.calc_feature();
	put_features(fo, st.feats);
	const int n = (int)mats.size();
	st.pairwise_matches.assign(n, std::vector<MatchInfo>(n));
	std::vector<std::pair<int, int>> tasks;
	if (ordered) for (int i = 0; i < n; ++i) tasks.emplace_back(i, (i + 1) % n);
	else for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) tasks.emplace_back(i, j);
	std::vector<Shape2D> shapes;
	for (auto& r : st.imgs) shapes.push_back(r.shape());
	auto infos = hip_match_images(st.feats, shapes, tasks, base_seed);
	put_pairs(fo, st.feats, tasks, infos);
	st.record_matches(tasks, infos, ordered);
	st.assign_center();
	st.estimate_camera();
	for (auto& c : st.cameras) {
		double v[13] = {c.focal, c.aspect, c.ppx, c.ppy};
		for (int k = 0; k < 9; ++k) v[4 + k] = c.R[k];
		put(fo, v, 13);
		fprintf(stderr, "camera focal=%g ppx=%g ppy=%g\n", c.focal, c.ppx, c.ppy);
	}
	st.bundle.proj_method = ConnectedImages::spherical;
	st.bundle.update_proj_range();
	Mat32f pano = st.bundle.blend();
	put1<int32_t>(fo, pano.rows()); put1<int32_t>(fo, pano.cols());
	put(fo, pano.ptr(), (size_t)pano.rows() * pano.cols() * 3);
	fprintf(stderr, "Final Image Size: (%d, %d)\n", pano.cols(), pano.rows());
	fclose(fo);
	return 0;
}

int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin [base_seed]\n", argv[0]); return 2; }
	const uint32_t base_seed = argc > 3 ? (uint32_t)strtoul(argv[3], nullptr, 10) : 42u;
	FILE* fi = fopen(argv[1], "rb");
	if (!fi) { perror(argv[1]); return 2; }
	int32_t hdr[3];
	if (fread(hdr, 4, 3, fi) != 3) return 2;
	const int n = hdr[0], h = hdr[1], w = hdr[2];
	std::vector<Mat32f> mats;
	for (int k = 0; k < n; ++k) {
		mats.emplace_back(h, w, 3);
		if (fread(mats.back().ptr(), sizeof(float), (size_t)h * w * 3, fi) != (size_t)h * w * 3) return 2;
	}
	fclose(fi);
	const std::string mode = argc > 4 ? argv[4] : "";
	const bool camera_mode = mode == "camera" || mode == "camera_ordered";
	config::ORDERED_INPUT = !camera_mode || mode == "camera_ordered"; config::ESTIMATE_CAMERA = camera_mode; config::TRANS = !camera_mode;   // TRANS mode: affine RANSAC, flat blend
	config::LAZY_READ = false;
	if (camera_mode) return run_camera_mode(mats, argv[2], base_seed, mode == "camera_ordered");

	// ---- StitcherBase::calc_feature (stitch/stitcherbase.cc:9-27)
	std::vector<ImageRef> imgs;
	for (auto& m : mats) imgs.emplace_back(m);
	SIFTDetector feature_det;
	std::vector<const Mat32f*> ptrs;
	for (auto& r : imgs) ptrs.push_back(r.img);
	HipFeatureSet fs = feature_det.calc_feature(ptrs);
	FILE* fo = fopen(argv[2], "wb");
	if (!fo) { perror(argv[2]); return 2; }
	for (int k = 0; k < n; ++k) {
		put1<int32_t>(fo, (int32_t)fs.feats[k].size());
		for (auto& d : fs.feats[k]) put(fo, d.descriptor.data(), 128);
		for (auto& d : fs.feats[k]) { double c[2] = {d.coor.x, d.coor.y}; put(fo, c, 2); }
		fprintf(stderr, "Image %d has %zu features\n", k, fs.feats[k].size());
	}

	// ---- Stitcher::linear_pairwise_match (stitch/stitcher.cc:115-136): (i, i+1) for an ordered input
	std::vector<std::pair<int, int>> tasks;
	for (int i = 0; i + 1 < n; ++i) tasks.emplace_back(i, i + 1);
	std::vector<Shape2D> shapes;
	for (auto& r : imgs) shapes.push_back(r.shape());
	PairWiseMatcher pwmatcher(fs);
	pwmatcher.precompute(tasks);
	auto infos = hip_match_images(fs, shapes, tasks, base_seed);
	put1<int32_t>(fo, (int32_t)tasks.size());
	for (size_t p = 0; p < tasks.size(); ++p) {
		MatchData md = pwmatcher.match(tasks[p].first, tasks[p].second);
		put1<int32_t>(fo, tasks[p].first); put1<int32_t>(fo, tasks[p].second); put1<int32_t>(fo, md.size());
		for (auto& q : md.data) { int32_t v[2] = {q.first, q.second}; put(fo, v, 2); }
		const MatchInfo& info = infos[p].second;
		put1<int32_t>(fo, infos[p].first ? 1 : 0); put1<float>(fo, info.confidence);
		double hh[9]; for (int i = 0; i < 9; ++i) hh[i] = infos[p].first ? info.homo[i] : 0.0;
		put(fo, hh, 9);
		put1<int32_t>(fo, (int32_t)(infos[p].first ? info.match.size() : 0));
		if (infos[p].first) for (auto& m : info.match) { double v[4] = {m.first.x, m.first.y, m.second.x, m.second.y}; put(fo, v, 4); }
		fprintf(stderr, "pair (%d,%d): %d matches, %s, confidence %g\n", tasks[p].first, tasks[p].second, md.size(),
				infos[p].first ? "connected" : "rejected", info.confidence);
	}

	// ---- chain to the middle image (stand-in for the host-only camera estimation), then
	// ConnectedImages::{calc_inverse_homo, update_proj_range, blend} (stitch/stitcher_image.cc)
	bool all = !tasks.empty();
	for (auto& r : infos) all = all && r.first;
	if (all) {
		ConnectedImages bundle;
		bundle.proj_method = ConnectedImages::flat;
		bundle.identity_idx = n >> 1;                               // stitcher.cc:139
		std::vector<Homography> to_mid(n, Homography::I());
		for (int i = bundle.identity_idx + 1; i < n; ++i) to_mid[i] = mul(to_mid[i - 1], infos[i - 1].second.homo);     // H(i -> i-1)
		// images left of the middle need the inverse direction: invert through the C-ABI's host helper
		for (int i = bundle.identity_idx - 1; i >= 0; --i) {
			const op_config cfg = hip_config_snapshot();
			double hinv[9], rng[4]; op_blend_geom g; int sh[2] = {w, h};
			PANO_HIP_CHECK(op_blend_prepare(&cfg, 0, 0, 1, sh, infos[i].second.homo.data, &g, hinv, rng));
			Homography inv; for (int k = 0; k < 9; ++k) inv[k] = hinv[k];
			to_mid[i] = mul(to_mid[i + 1], inv);
		}
		for (int i = 0; i < n; ++i) {
			bundle.component.emplace_back(&imgs[i]);
			bundle.component.back().homo = to_mid[i];
		}
		bundle.calc_inverse_homo();
		bundle.update_proj_range();
		Mat32f pano = bundle.blend();
		put1<int32_t>(fo, pano.rows()); put1<int32_t>(fo, pano.cols());
		put(fo, pano.ptr(), (size_t)pano.rows() * pano.cols() * 3);
		for (auto& t : to_mid) put(fo, t.data, 9);
		fprintf(stderr, "Final Image Size: (%d, %d)\n", pano.cols(), pano.rows());
	} else {
		put1<int32_t>(fo, 0); put1<int32_t>(fo, 0);
	}
	fclose(fo);
	return 0;
}
