// pano_hip.hh -- C++ host mirror of the reference's class surface over the C-ABI
// (include/openpano_hip.h).  Header-only; link with -lopenpano_hip.
//
// Two build modes, same adapter code:
//   -DOPENPANO_WITH_REFERENCE (+ -I<reference>/src): the adapters are written against the
//        reference's OWN types (Mat32f, Descriptor, MatchData, MatchInfo, ConnectedImages, the
//        config:: globals) and plug into its class hierarchy -- HipSIFTDetector IS-A
//        pano::FeatureDetector, so `feature_det.reset(new HipSIFTDetector)` in
//        StitcherBase's constructor (stitch/stitcherbase.hh:53) is the whole integration for
//        SIFT.  INTEGRATION.md lists every such hook; oracle/ref_dropin_test.cc exercises them
//        against the reference's CPU classes in one process.
//   standalone: pano_types.hh supplies value types with the same names and members.
//
// Interfaces mirrored (file:line under /root/reference/src):
//   FeatureDetector::detect_feature / do_detect_feature   feature/feature.hh:42-57
//   StitcherBase::calc_feature (batched form)             stitch/stitcherbase.cc:9-27
//   PairWiseMatcher(feats).match(i, j)                    feature/matcher.hh:40-51
//   TransformEstimation(...).get_transform(MatchInfo*)    stitch/transform_estimate.hh:22-31
//   Stitcher::pairwise_match body (batched form)          stitch/stitcher.cc:66-136
//   ConnectedImages::blend                                stitch/stitcher_image.hh:92
//   CylinderWarper::warp                                  stitch/warp.hh:47-55
// Error behaviour follows the reference: unrecoverable conditions end in error_exit()
// (lib/debugutils.cc:57-60: message on stderr, exit(1)); no exception crosses the C-ABI.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <cstdlib>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <utility>
#include <vector>

#include "openpano_hip.h"

#ifdef OPENPANO_WITH_REFERENCE
#include "lib/config.hh"
#include "lib/mat.h"
#include "lib/geometry.hh"
#include "feature/feature.hh"
#include "feature/matcher.hh"
#include "stitch/match_info.hh"
#include "stitch/homography.hh"
#include "stitch/imageref.hh"
#include "stitch/stitcher_image.hh"
#include "stitch/camera.hh"
#include "lib/debugutils.hh"
#include "lib/timer.hh"
#include "pano_host.h"
#else
#include "pano_types.hh"
#include "pano_camera.hh"
#endif

namespace pano {

// ---- errors: the reference's error_exit (lib/debugutils.cc:57-60) ----
[[noreturn]] inline void hip_error_exit(const std::string& where) {
	fprintf(stderr, "%s: %s\n", where.c_str(), op_last_error());
	exit(1);
}
#define PANO_HIP_CHECK(expr) do { if ((expr) != OP_OK) ::pano::hip_error_exit(#expr); } while (0)

// ---- config: POD snapshot of namespace config at every call (the CLI fills the globals after
// static init, main.cc:237-292, so nothing is cached) ----
inline op_config hip_config_snapshot() {
	op_config c;
	op_config_default(&c);
	using namespace config;
	c.SIFT_WORKING_SIZE = SIFT_WORKING_SIZE; c.NUM_OCTAVE = NUM_OCTAVE; c.NUM_SCALE = NUM_SCALE;
	c.SCALE_FACTOR = SCALE_FACTOR; c.GAUSS_SIGMA = GAUSS_SIGMA; c.GAUSS_WINDOW_FACTOR = GAUSS_WINDOW_FACTOR;
	c.JUDGE_EXTREMA_DIFF_THRES = JUDGE_EXTREMA_DIFF_THRES; c.CONTRAST_THRES = CONTRAST_THRES;
	c.PRE_COLOR_THRES = PRE_COLOR_THRES; c.EDGE_RATIO = EDGE_RATIO;
	c.CALC_OFFSET_DEPTH = CALC_OFFSET_DEPTH; c.OFFSET_THRES = OFFSET_THRES;
	c.ORI_RADIUS = ORI_RADIUS; c.ORI_HIST_SMOOTH_COUNT = ORI_HIST_SMOOTH_COUNT;
	c.DESC_HIST_SCALE_FACTOR = DESC_HIST_SCALE_FACTOR; c.DESC_INT_FACTOR = DESC_INT_FACTOR;
	c.MATCH_REJECT_NEXT_RATIO = MATCH_REJECT_NEXT_RATIO;
	c.RANSAC_ITERATIONS = RANSAC_ITERATIONS; c.RANSAC_INLIER_THRES = RANSAC_INLIER_THRES;
	c.INLIER_IN_MATCH_RATIO = INLIER_IN_MATCH_RATIO; c.INLIER_IN_POINTS_RATIO = INLIER_IN_POINTS_RATIO;
	c.CYLINDER = CYLINDER; c.TRANS = TRANS; c.ESTIMATE_CAMERA = ESTIMATE_CAMERA;
	c.ORDERED_INPUT = ORDERED_INPUT; c.LAZY_READ = LAZY_READ; c.MULTIBAND = MULTIBAND;
	c.MAX_OUTPUT_SIZE = MAX_OUTPUT_SIZE; c.FOCAL_LENGTH = FOCAL_LENGTH;
	return c;
}

// ---- one op_ctx per host thread: the reference calls detect_feature / match concurrently from
// OpenMP threads (stitcherbase.cc:14, stitcher.cc:106); contexts are thread-compatible ----
class HipContext {
	public:
		static op_ctx* get() {
			thread_local HipContext c;
			return c.ctx;
		}
		static int& device() { static int d = 0; return d; }     // set before first use to pick a GPU
		// Several GPUs in this process: set_devices({0, 1, ...}) -- or OPENPANO_DEVICES=0,1,... in
		// the environment -- makes the batched adapters (HipSIFTDetector::calc_feature, HipPairWiseMatcher)
		// shard their image / pair loops over the listed devices; results land on the first one.
		static void set_devices(const std::vector<int>& devs) {
			if (group_slot()) { op_group_destroy(group_slot()); group_slot() = nullptr; }
			if (devs.size() > 1) PANO_HIP_CHECK(op_group_create(devs.data(), (int)devs.size(), &group_slot()));
			if (!devs.empty()) device() = devs[0];
			group_env_done() = true;
		}
		static op_group* group() {
			if (!group_env_done()) {
				group_env_done() = true;
				if (const char* e = std::getenv("OPENPANO_DEVICES")) {
					std::vector<int> devs;
					for (const char* p = e; *p;) { devs.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
					set_devices(devs);
				}
			}
			return group_slot();
		}
		// the context that owns gathered results: context 0 of the group, else this thread's own
		static op_ctx* home() { return group() ? op_group_ctx(group(), 0) : get(); }
	private:
		static op_group*& group_slot() { static op_group* g = nullptr; return g; }
		static bool& group_env_done() { static bool b = false; return b; }
		op_ctx* ctx = nullptr;
		HipContext() { PANO_HIP_CHECK(op_ctx_create(device(), nullptr, &ctx)); }
		~HipContext() { op_ctx_destroy(ctx); }
};

// ===================================== SIFT =====================================
#ifdef OPENPANO_WITH_REFERENCE
#define PANO_DETECTOR_BASE : public FeatureDetector
#define PANO_OVERRIDE override
#else
// the reference's base class (feature/feature.hh:42-52): detect_feature re-centres the [0,1)
// coordinates do_detect_feature returns (feature/feature.cc:20-28)
class FeatureDetector {
	public:
		FeatureDetector() = default;
		virtual ~FeatureDetector() = default;
		FeatureDetector(const FeatureDetector&) = delete;
		FeatureDetector& operator=(const FeatureDetector&) = delete;
		std::vector<Descriptor> detect_feature(const Mat32f& img) const {
			auto ret = do_detect_feature(img);
			for (auto& d : ret) {
				d.coor.x = (d.coor.x - 0.5) * img.width();
				d.coor.y = (d.coor.y - 0.5) * img.height();
			}
			return ret;
		}
		virtual std::vector<Descriptor> do_detect_feature(const Mat32f& img) const = 0;
};
#define PANO_DETECTOR_BASE : public FeatureDetector
#define PANO_OVERRIDE override
#endif

// device-resident features of a whole image set: what StitcherBase keeps as `feats`, plus the
// op_features handle so that the matcher does not re-upload descriptors
struct HipFeatureSet {
	op_features* handle = nullptr;
	std::vector<std::vector<Descriptor>> feats;     // centred coordinates, like StitcherBase::feats
	HipFeatureSet() = default;
	HipFeatureSet(const HipFeatureSet&) = delete;
	HipFeatureSet& operator=(const HipFeatureSet&) = delete;
	HipFeatureSet(HipFeatureSet&& o): handle(o.handle), feats(std::move(o.feats)) { o.handle = nullptr; }
	HipFeatureSet& operator=(HipFeatureSet&& o) { if (this != &o) { op_features_free(handle); handle = o.handle; feats = std::move(o.feats); o.handle = nullptr; } return *this; }
	~HipFeatureSet() { op_features_free(handle); }
};

class HipSIFTDetector PANO_DETECTOR_BASE {
	public:
		// SIFTDetector::do_detect_feature (feature/feature.cc:31-47): [0,1) coordinates
		std::vector<Descriptor> do_detect_feature(const Mat32f& mat) const PANO_OVERRIDE {
			op_ctx* ctx = HipContext::get();
			const op_config cfg = hip_config_snapshot();
			op_image im{mat.ptr(), mat.rows(), mat.cols(), 0, OP_F32};
			if (mat.channels() != 3) { fprintf(stderr, "HipSIFTDetector: image must have 3 channels\n"); exit(1); }
			op_features* f = nullptr;
			PANO_HIP_CHECK(op_sift_batch(ctx, &cfg, &im, 1, &f));
			const int k = op_features_count(f, 0);
			std::vector<float> desc((size_t)k * 128);
			std::vector<double> real((size_t)k * 2);
			if (k) {
				PANO_HIP_CHECK(op_features_copy(ctx, f, 0, desc.data(), nullptr));
				PANO_HIP_CHECK(op_features_copy_real(ctx, f, 0, real.data()));
			}
			op_features_free(f);
			std::vector<Descriptor> ret(k);
			for (int i = 0; i < k; ++i) {
				ret[i].coor = Vec2D(real[2 * i], real[2 * i + 1]);
				ret[i].descriptor.assign(desc.begin() + (size_t)i * 128, desc.begin() + (size_t)(i + 1) * 128);
			}
			return ret;
		}

		// StitcherBase::calc_feature (stitch/stitcherbase.cc:9-27) as ONE batched device call:
		// all images in one launch series, descriptors left resident for the matcher.
		HipFeatureSet calc_feature(const std::vector<const Mat32f*>& imgs) const {
			op_ctx* ctx = HipContext::get();
			const op_config cfg = hip_config_snapshot();
			std::vector<op_image> ims;
			for (auto* m : imgs) ims.push_back(op_image{m->ptr(), m->rows(), m->cols(), 0, OP_F32});
			HipFeatureSet fs;
			std::vector<float> desc; std::vector<double> coor;
			// Images that came out of a decoder are bytes / 255 (read_img, lib/imgio.cc:54-56,75-77: (float)((double)byte / 255.0)).
			// When EVERY value of every image is exactly such a float, the bytes travel instead -- a quarter of the PCIe traffic,
			// pageable memory at that -- and the device converts them with the same expression (OP_U8): the same features, bit for bit.
			// Non-decoder inputs (warped, normalised images) must not pay for this: a probe of every image's first values decides
			// whether the byte buffers are allocated at all, the full scan stops at the first chunk that holds a mismatch, and
			// the comparison is on BIT PATTERNS (-0.0f is not the float of byte 0).
			std::vector<std::unique_ptr<unsigned char[]>> bytes(imgs.size());
			if (!HipContext::group()) {
				auto as_byte = [](float v, unsigned char& b) {
					const float q = v * 255.f + 0.5f;
					const int c = q >= 0.f && q < 256.f ? (int)q : 0;
					b = (unsigned char)c;
					const float back = (float)((double)c / 255.0);
					uint32_t x, y; memcpy(&x, &back, 4); memcpy(&y, &v, 4);
					return x == y;
				};
				bool plausible = true;
				for (size_t k = 0; k < imgs.size() && plausible; ++k) {
					const size_t n = std::min<size_t>((size_t)imgs[k]->rows() * imgs[k]->cols() * 3, 4096);
					const float* v = imgs[k]->ptr();
					unsigned char b;
					for (size_t e = 0; e < n && plausible; ++e) plausible = as_byte(v[e], b);
				}
				std::atomic<int> all_bytes{plausible ? 1 : 0};
				if (plausible) {
					for (size_t k = 0; k < imgs.size(); ++k) bytes[k].reset(new unsigned char[(size_t)imgs[k]->rows() * imgs[k]->cols() * 3]);
#pragma omp parallel for schedule(dynamic)
					for (long job = 0; job < (long)imgs.size() * 8; ++job) {
						const size_t k = (size_t)(job >> 3), part = (size_t)(job & 7), n = (size_t)imgs[k]->rows() * imgs[k]->cols() * 3;
						const float* v = imgs[k]->ptr();
						unsigned char* b = bytes[k].get();
						const size_t e1 = n * (part + 1) / 8;
						for (size_t e0 = n * part / 8; e0 < e1 && all_bytes.load(std::memory_order_relaxed); e0 += 65536) {
							bool ok = true;
							for (size_t e = e0; e < std::min(e1, e0 + 65536); ++e) ok &= as_byte(v[e], b[e]);
							if (!ok) all_bytes.store(0, std::memory_order_relaxed);
						}
					}
				}
				if (all_bytes.load()) for (size_t k = 0; k < imgs.size(); ++k) { ims[k].data = bytes[k].get(); ims[k].dtype = OP_U8; }
			}
			if (op_group* g = HipContext::group()) {        // images dealt over the group's GPUs, features all-gathered
				PANO_HIP_CHECK(op_sift_batch_multi(g, &cfg, ims.data(), (int)ims.size(), &fs.handle));
				ctx = op_group_ctx(g, 0);
				const size_t total = (size_t)op_features_total(fs.handle);
				desc.resize(total * 128 + 1); coor.resize(total * 2 + 1);
				for (size_t k = 0; k < imgs.size(); ++k)
					if (op_features_count(fs.handle, (int)k))
						PANO_HIP_CHECK(op_features_copy(ctx, fs.handle, (int)k, desc.data() + (size_t)op_features_offset(fs.handle, (int)k) * 128,
								coor.data() + (size_t)op_features_offset(fs.handle, (int)k) * 2));
			} else {
				// uploads, kernels and the copy back to the host pipelined over chunks of the batch; the host buffers are
				// sized from a first guess and the call repeated once if the images hold more features than that
				size_t cap = (size_t)imgs.size() * 4096;
				for (int attempt = 0; attempt < 2; ++attempt) {
					desc.resize(cap * 128 + 1); coor.resize(cap * 2 + 1);
					const int rc = op_sift_batch_host(ctx, &cfg, ims.data(), (int)ims.size(), desc.data(), coor.data(), (int64_t)cap, &fs.handle);
					if (rc == OP_OK) break;
					if (rc != OP_ERR_CAPACITY || attempt == 1) hip_error_exit("op_sift_batch_host");
					cap = (size_t)op_features_total(fs.handle) + 16;
					op_features_free(fs.handle); fs.handle = nullptr;
				}
			}
			fs.feats.resize(imgs.size());
			for (size_t k = 0; k < imgs.size(); ++k) {
				const int n = op_features_count(fs.handle, (int)k);
				if (n == 0) {    // stitcherbase.cc:20-21
					fprintf(stderr, "Cannot find feature in image %d!\n", (int)k);
					exit(1);
				}
				const float* d = desc.data() + (size_t)op_features_offset(fs.handle, (int)k) * 128;
				const double* c = coor.data() + (size_t)op_features_offset(fs.handle, (int)k) * 2;
				fs.feats[k].resize(n);
				for (int i = 0; i < n; ++i) {
					fs.feats[k][i].coor = Vec2D(c[2 * i], c[2 * i + 1]);
					fs.feats[k][i].descriptor.assign(d + (size_t)i * 128, d + (size_t)(i + 1) * 128);
				}
			}
			return fs;
		}
};

// ===================================== MATCH =====================================
// Same public signature as the reference's PairWiseMatcher (feature/matcher.hh:40-51).  match()
// may be called concurrently (stitcher.cc:106-109): the first call matches the whole task list
// in one device launch series, later calls are lookups.
class HipPairWiseMatcher {
	public:
		explicit HipPairWiseMatcher(const std::vector<std::vector<Descriptor>>& feats): feats(feats) {
			if (feats.empty() || feats.at(0).empty()) { fprintf(stderr, "PairWiseMatcher: no features\n"); exit(1); }
			upload();
		}
		// descriptors already resident (HipSIFTDetector::calc_feature): no second H2D pass
		explicit HipPairWiseMatcher(const HipFeatureSet& fs): feats(fs.feats), handle(fs.handle), owns(false) {}
		HipPairWiseMatcher(const HipPairWiseMatcher&) = delete;
		HipPairWiseMatcher& operator=(const HipPairWiseMatcher&) = delete;
		~HipPairWiseMatcher() { if (owns) op_features_free(handle); }

		// return pair of <idx in i, idx in j>
		MatchData match(int i, int j) const {
			std::lock_guard<std::mutex> lk(mu);
			if (cache.empty()) precompute_default();
			auto it = cache.find(std::make_pair(i, j));
			if (it == cache.end()) {
				run({{i, j}});
				it = cache.find(std::make_pair(i, j));
			}
			return it->second;
		}

		// match an explicit task list in one call (Stitcher::pairwise_match's `tasks`)
		void precompute(const std::vector<std::pair<int, int>>& tasks) const {
			std::lock_guard<std::mutex> lk(mu);
			run(tasks);
		}
		op_features* device_features() const { return handle; }

	protected:
		const std::vector<std::vector<Descriptor>>& feats;    // must outlive the matcher (matcher.hh:59)
		op_features* handle = nullptr;
		bool owns = true;
		mutable std::mutex mu;
		mutable std::map<std::pair<int, int>, MatchData> cache;

		void upload() {
			const int n = (int)feats.size();
			std::vector<std::vector<float>> flat(n);
			std::vector<std::vector<double>> coor(n);
			std::vector<const float*> dp(n); std::vector<const double*> cp(n); std::vector<int> counts(n);
			for (int k = 0; k < n; ++k) {       // PairWiseMatcher::build flattening (matcher.cc:75-81)
				counts[k] = (int)feats[k].size();
				flat[k].resize((size_t)counts[k] * 128 + 1); coor[k].resize((size_t)counts[k] * 2 + 1);
				for (int i = 0; i < counts[k]; ++i) {
					if (feats[k][i].descriptor.size() != 128) { fprintf(stderr, "PairWiseMatcher: descriptors must be 128-D\n"); exit(1); }
					memcpy(&flat[k][(size_t)i * 128], feats[k][i].descriptor.data(), 128 * sizeof(float));
					coor[k][2 * i] = feats[k][i].coor.x; coor[k][2 * i + 1] = feats[k][i].coor.y;
				}
				dp[k] = flat[k].data(); cp[k] = coor[k].data();
			}
			PANO_HIP_CHECK(op_features_from_host(HipContext::home(), dp.data(), cp.data(), counts.data(), n, &handle));
		}
		void precompute_default() const {
			const int n = (int)feats.size();
			std::vector<std::pair<int, int>> tasks;
			if (config::ORDERED_INPUT) for (int i = 0; i < n; ++i) tasks.emplace_back(i, (i + 1) % n);   // stitcher.cc:121-124
			else for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) tasks.emplace_back(i, j);    // stitcher.cc:99-100
			run(tasks);
		}
		void run(const std::vector<std::pair<int, int>>& tasks) const {
			if (tasks.empty()) return;
			std::vector<int> pr;
			for (auto& t : tasks) { pr.push_back(t.first); pr.push_back(t.second); }
			const op_config cfg = hip_config_snapshot();
			op_matches* m = nullptr;
			if (op_group* g = HipContext::group())          // pair list dealt over the group's GPUs (K_i * K_j balanced)
				PANO_HIP_CHECK(op_match_pairs_multi(g, &cfg, handle, pr.data(), (int)tasks.size(), &m));
			else
				PANO_HIP_CHECK(op_match_pairs(HipContext::get(), &cfg, handle, pr.data(), (int)tasks.size(), &m));
			for (size_t p = 0; p < tasks.size(); ++p) {
				const int c = op_matches_count(m, (int)p);
				std::vector<int> idx((size_t)c * 2 + 2);
				if (c) PANO_HIP_CHECK(op_matches_copy(m, (int)p, idx.data()));
				MatchData md;
				for (int q = 0; q < c; ++q) md.data.emplace_back(idx[2 * q], idx[2 * q + 1]);
				cache[tasks[p]] = std::move(md);
			}
			op_matches_free(m);
		}
};

// ===================================== RANSAC =====================================
// Same constructor and get_transform as the reference's TransformEstimation
// (stitch/transform_estimate.hh:22-31).  The reference seeds std::mt19937 from
// std::random_device per call (transform_estimate.cc:64-65); so does this adapter unless a seed
// is injected (tests, reproducible runs).
class HipTransformEstimation {
	public:
		HipTransformEstimation(const MatchData& m_match, const std::vector<Vec2D>& kp1, const std::vector<Vec2D>& kp2,
				const Shape2D& shape1, const Shape2D& shape2):
			match(m_match), kp1(kp1), kp2(kp2), shape1(shape1), shape2(shape2) {}
		HipTransformEstimation(const HipTransformEstimation&) = delete;
		HipTransformEstimation& operator=(const HipTransformEstimation&) = delete;

		static bool& seed_injected() { static bool b = false; return b; }
		static uint32_t& injected_seed() { static uint32_t s = 0; return s; }

		// get a transform matrix from second(f2) -> first(f1)
		bool get_transform(MatchInfo* info) {
			op_ctx* ctx = HipContext::get();
			const op_config cfg = hip_config_snapshot();
			std::vector<double> c1(kp1.size() * 2 + 2), c2(kp2.size() * 2 + 2);
			for (size_t i = 0; i < kp1.size(); ++i) { c1[2 * i] = kp1[i].x; c1[2 * i + 1] = kp1[i].y; }
			for (size_t i = 0; i < kp2.size(); ++i) { c2[2 * i] = kp2[i].x; c2[2 * i + 1] = kp2[i].y; }
			const double* cp[2] = {c1.data(), c2.data()};
			const int counts[2] = {(int)kp1.size(), (int)kp2.size()};
			op_features* f = nullptr;
			PANO_HIP_CHECK(op_features_from_host(ctx, nullptr, cp, counts, 2, &f));
			std::vector<int> idx(match.data.size() * 2 + 2);
			for (size_t i = 0; i < match.data.size(); ++i) { idx[2 * i] = match.data[i].first; idx[2 * i + 1] = match.data[i].second; }
			const int* ip[1] = {idx.data()};
			const int mc[1] = {(int)match.data.size()};
			op_matches* m = nullptr;
			PANO_HIP_CHECK(op_matches_from_host(ip, mc, 1, &m));
			const int pairs[2] = {0, 1};
			const int shapes[4] = {shape1.w, shape1.h, shape2.w, shape2.h};
			uint32_t seed = seed_injected() ? injected_seed() : std::random_device{}();
			op_ransac_result* r = nullptr;
			PANO_HIP_CHECK(op_ransac_pairs(ctx, &cfg, f, m, pairs, 1, shapes, &seed, 0, &r));
			const bool ok = fill(r, 0, match, kp1, kp2, info);
			op_ransac_free(r); op_matches_free(m); op_features_free(f);
			return ok;
		}

		// MatchInfo of pair p of a batched result (fill_inliers_to_matchinfo's outputs,
		// transform_estimate.cc:150-218)
		static bool fill(const op_ransac_result* r, int p, const MatchData& match, const std::vector<Vec2D>& kp1,
				const std::vector<Vec2D>& kp2, MatchInfo* info) {
			info->confidence = op_ransac_confidence(r, p);
			if (!op_ransac_ok(r, p)) return false;
			double h[9];
			PANO_HIP_CHECK(op_ransac_homo(r, p, h));
			for (int i = 0; i < 9; ++i) info->homo[i] = h[i];
			const int n = op_ransac_inlier_count(r, p);
			std::vector<int> inl(n + 1);
			if (n) PANO_HIP_CHECK(op_ransac_inliers(r, p, inl.data()));
			info->match.clear();
			for (int i = 0; i < n; ++i)
				info->match.emplace_back(kp1[match.data[inl[i]].first], kp2[match.data[inl[i]].second]);
			return true;
		}

	private:
		const MatchData& match;
		const std::vector<Vec2D>&kp1, &kp2;
		const Shape2D shape1, shape2;
};

// One op_match_pairs + one op_ransac_pairs for a whole task list; pair p's RANSAC seed is seeds[p] (or derived from
// base_seed and p when seeds is empty).  The match lists never leave the device between the two calls; they come
// to the host once, for MatchInfo::match and the caller's MatchData.
inline void hip_match_and_estimate(const HipFeatureSet& fs, const std::vector<Shape2D>& shapes, const std::vector<std::pair<int, int>>& tasks,
		const std::vector<uint32_t>& seeds, uint32_t base_seed, std::vector<MatchData>& matches, std::vector<std::pair<bool, MatchInfo>>& out) {
	op_ctx* ctx = HipContext::get();
	const op_config cfg = hip_config_snapshot();
	std::vector<int> pr, sh;
	for (auto& t : tasks) { pr.push_back(t.first); pr.push_back(t.second); }
	for (auto& s : shapes) { sh.push_back(s.w); sh.push_back(s.h); }
	op_matches* m = nullptr;
	op_ransac_result* r = nullptr;
	if (op_group* g = HipContext::group()) {           // pair list dealt over the group's GPUs; RANSAC follows the deal
		PANO_HIP_CHECK(op_match_pairs_multi(g, &cfg, fs.handle, pr.data(), (int)tasks.size(), &m));
		PANO_HIP_CHECK(op_ransac_pairs_multi(g, &cfg, fs.handle, m, pr.data(), (int)tasks.size(), sh.data(), seeds.empty() ? nullptr : seeds.data(), base_seed, &r));
	} else {
		PANO_HIP_CHECK(op_match_pairs(ctx, &cfg, fs.handle, pr.data(), (int)tasks.size(), &m));
		PANO_HIP_CHECK(op_ransac_pairs(ctx, &cfg, fs.handle, m, pr.data(), (int)tasks.size(), sh.data(), seeds.empty() ? nullptr : seeds.data(), base_seed, &r));
	}
	std::vector<int64_t> off(tasks.size() + 1);
	std::vector<int> idx((size_t)op_matches_total(m) * 2 + 2);
	PANO_HIP_CHECK(op_matches_copy_all(m, idx.data(), off.data()));
	matches.assign(tasks.size(), MatchData());
	out.assign(tasks.size(), std::pair<bool, MatchInfo>());
	std::vector<std::vector<Vec2D>> kps(fs.feats.size());
	for (size_t k = 0; k < fs.feats.size(); ++k) { kps[k].reserve(fs.feats[k].size()); for (auto& d : fs.feats[k]) kps[k].push_back(d.coor); }
	for (size_t p = 0; p < tasks.size(); ++p) {
		MatchData& md = matches[p];
		md.data.reserve((size_t)(off[p + 1] - off[p]));
		for (int64_t q = off[p]; q < off[p + 1]; ++q) md.data.emplace_back(idx[2 * q], idx[2 * q + 1]);
		out[p].first = HipTransformEstimation::fill(r, (int)p, md, kps[tasks[p].first], kps[tasks[p].second], &out[p].second);
	}
	op_ransac_free(r); op_matches_free(m);
}

// ---- the batched hooks inside the reference's OWN loops ("batched form") ----
// Stitcher::pairwise_match / linear_pairwise_match (stitch/stitcher.cc:96-136) construct one matcher and then call
// match_image(pwmatcher, i, j) per task from an OpenMP loop; match_image (:66-94) asks the matcher for the pair's
// MatchData, runs a TransformEstimation on it and keeps the bookkeeping.  HipBatchedMatcher has PairWiseMatcher's
// place and match(i, j) signature, but its constructor runs the WHOLE default task list (all i < j, or (i, i + 1)
// under ORDERED_INPUT: the lists of :99-100 / :121-124) through one op_match_pairs + one op_ransac_pairs; match()
// returns a MatchData that remembers where its pair's RANSAC result lies, and HipBatchedTransformEstimation -- in
// TransformEstimation's place, same constructor and get_transform -- hands that result over.  The reference's loop
// bodies stay as they are.
struct HipMatchData : public MatchData {
	const std::pair<bool, MatchInfo>* estimated = nullptr;
};
class HipBatchedMatcher {
	public:
		// imgs: anything with shape() per image (std::vector<ImageRef>, stitcherbase.hh:28)
		template <typename ImageList>
		HipBatchedMatcher(const HipFeatureSet& fs, const ImageList& imgs) {
			const int n = (int)fs.feats.size();
			if (config::ORDERED_INPUT) for (int i = 0; i < n; ++i) tasks.emplace_back(i, (i + 1) % n);         // stitcher.cc:121-124
			else for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) tasks.emplace_back(i, j);          // stitcher.cc:99-100
			std::vector<Shape2D> shapes;
			for (auto& r : imgs) shapes.push_back(r.shape());
			// the reference seeds every get_transform from std::random_device (transform_estimate.cc:64-65)
			std::vector<uint32_t> seeds(tasks.size());
			for (auto& s : seeds) s = HipTransformEstimation::seed_injected() ? HipTransformEstimation::injected_seed() : std::random_device{}();
			hip_match_and_estimate(fs, shapes, tasks, seeds, 0, matches, results);
			for (size_t p = 0; p < tasks.size(); ++p) index[tasks[p]] = p;
		}
		HipBatchedMatcher(const HipBatchedMatcher&) = delete;
		HipBatchedMatcher& operator=(const HipBatchedMatcher&) = delete;
		// return pair of <idx in i, idx in j>, plus the pair's transform estimate
		HipMatchData match(int i, int j) const {
			auto it = index.find(std::make_pair(i, j));
			if (it == index.end()) { fprintf(stderr, "HipBatchedMatcher: pair (%d, %d) is not in the task list\n", i, j); exit(1); }
			HipMatchData md;
			md.data = matches[it->second].data;
			md.estimated = &results[it->second];
			return md;
		}
	private:
		std::vector<std::pair<int, int>> tasks;
		std::vector<MatchData> matches;
		std::vector<std::pair<bool, MatchInfo>> results;
		std::map<std::pair<int, int>, size_t> index;
};
class HipBatchedTransformEstimation {
	public:
		HipBatchedTransformEstimation(const HipMatchData& m_match, const std::vector<Vec2D>&, const std::vector<Vec2D>&, const Shape2D&, const Shape2D&):
			match(m_match) {}
		bool get_transform(MatchInfo* info) {
			*info = match.estimated->second;
			return match.estimated->first;
		}
	private:
		const HipMatchData& match;
};

// Body of Stitcher::pairwise_match / linear_pairwise_match + match_image (stitch/stitcher.cc:66-136)
// for a whole task list: ONE matcher call and ONE batched RANSAC call.  out[k] = (succ, info of
// tasks[k], homography from j to i) -- the caller keeps the reference's bookkeeping
// (pairwise_matches[i][j] / inverse for [j][i], stitcher.cc:79-93).
inline std::vector<std::pair<bool, MatchInfo>> hip_match_images(const HipFeatureSet& fs,
		const std::vector<Shape2D>& shapes, const std::vector<std::pair<int, int>>& tasks, uint32_t base_seed) {
	std::vector<MatchData> matches;
	std::vector<std::pair<bool, MatchInfo>> out;
	hip_match_and_estimate(fs, shapes, tasks, std::vector<uint32_t>(), base_seed, matches, out);
	return out;
}

// ===================================== WARP + BLEND =====================================
#ifndef OPENPANO_WITH_REFERENCE
// stitch/stitcher_image.hh:15-98 (fields and method names as in the reference)
struct ConnectedImages {
	ConnectedImages() = default;
	ConnectedImages(const ConnectedImages&) = delete;
	ConnectedImages& operator=(const ConnectedImages&) = delete;
	struct Range {
		Vec2D min, max;
		Range() {}
		Range(const Vec2D& a, const Vec2D& b): min(a), max(b) {}
		Vec2D size() const { return max - min; }
	};
	enum ProjectionMethod { flat, cylindrical, spherical };
	ProjectionMethod proj_method = flat;
	Range proj_range;
	int identity_idx = 0;
	struct ImageComponent {
		Homography homo, homo_inv;
		ImageRef* imgptr = nullptr;
		Range range;
		ImageComponent() {}
		ImageComponent(ImageRef* img): imgptr(img) {}
	};
	std::vector<ImageComponent> component;
	// stitcher_image.cc:36-77: homo_inv from homo; ranges / proj_range / resolution from homo.  Kept
	// separate like the reference's: Stitcher::estimate_camera sets homo_inv = K R itself and only
	// calls update_proj_range() (stitcher.cc:154-158, :59)
	void calc_inverse_homo() { prepare(true, false); }
	void update_proj_range() { prepare(false, true); }
	Vec2D get_final_resolution() const { return resolution; }
	Mat32f blend() const;
	Vec2D resolution;
	private:
	void prepare(bool set_inverse, bool set_range);       // one host call of the C-ABI's geometry helper
};
#endif

// Geometry of a bundle through the C-ABI's host helper; fills homo_inv / range / proj_range the
// way ConnectedImages::calc_inverse_homo / update_proj_range do (stitcher_image.cc:36-77)
template <typename Bundle>
inline op_blend_geom hip_blend_prepare(const Bundle& b, std::vector<double>& hinv, std::vector<double>& ranges) {
	const int n = (int)b.component.size();
	std::vector<double> homo((size_t)n * 9); std::vector<int> shapes((size_t)n * 2);
	for (int i = 0; i < n; ++i) {
		for (int k = 0; k < 9; ++k) homo[(size_t)i * 9 + k] = b.component[i].homo[k];
		shapes[2 * i] = b.component[i].imgptr->width(); shapes[2 * i + 1] = b.component[i].imgptr->height();
	}
	hinv.assign((size_t)n * 9, 0); ranges.assign((size_t)n * 4, 0);
	op_blend_geom g;
	const op_config cfg = hip_config_snapshot();
	PANO_HIP_CHECK(op_blend_prepare(&cfg, (int)b.proj_method, b.identity_idx, n, shapes.data(), homo.data(), &g, hinv.data(), ranges.data()));
	return g;
}

// ConnectedImages::blend() (stitch/stitcher_image.cc:116-155) on the device.  Uses the bundle's
// own homo_inv / range / proj_range (already filled by calc_inverse_homo / update_proj_range)
// and the resolution of get_final_resolution(); the blender is chosen like the reference does
// (config::MULTIBAND > 0 ? MultiBandBlender : LinearBlender).
// crop = true additionally applies crop() (lib/imgproc.cc:200-235; main.cc:226-229 under config
// CROP) on the device, so only the cropped pixels cross PCIe.
template <typename Bundle>
inline Mat32f hip_blend(const Bundle& b, bool crop = false) {
	op_ctx* ctx = HipContext::get();
	const op_config cfg = hip_config_snapshot();
	const int n = (int)b.component.size();
	const Vec2D res = b.get_final_resolution();
	op_blend_geom g;
	g.proj_method = (int)b.proj_method;
	g.proj_min[0] = b.proj_range.min.x; g.proj_min[1] = b.proj_range.min.y;
	g.proj_max[0] = b.proj_range.max.x; g.proj_max[1] = b.proj_range.max.y;
	g.resolution[0] = res.x; g.resolution[1] = res.y;
#ifdef OPENPANO_WITH_REFERENCE
	{	// the line ConnectedImages::blend prints (stitcher_image.cc:121-124); the reference's run_test.py scrapes it
		const Vec2D size_d = b.proj_range.size() / res;
		const Coor size(size_d.x, size_d.y);
		print_debug("Final Image Size: (%d, %d)\n", size.x, size.y);
	}
#endif
	std::vector<op_blend_image> ims(n);
	for (int i = 0; i < n; ++i) {
		auto& c = b.component[i];
		c.imgptr->load();
		ims[i].data = c.imgptr->img->ptr(); ims[i].h = c.imgptr->height(); ims[i].w = c.imgptr->width(); ims[i].on_device = 0;
		ims[i].mat_h = c.imgptr->img->height(); ims[i].mat_w = c.imgptr->img->width();      // != h / w after a cylinder pre-warp (op_blend_image)
		for (int k = 0; k < 9; ++k) ims[i].homo_inv[k] = c.homo_inv[k];
		ims[i].range[0] = c.range.min.x; ims[i].range[1] = c.range.min.y; ims[i].range[2] = c.range.max.x; ims[i].range[3] = c.range.max.y;
	}
	op_canvas* cv = nullptr;
	PANO_HIP_CHECK(op_blend(ctx, &cfg, &g, ims.data(), n, &cv));
	if (crop) {
		op_canvas* cc = nullptr;
		PANO_HIP_CHECK(op_canvas_crop(ctx, cv, &cc, nullptr, nullptr));
		op_canvas_free(cv);
		cv = cc;
	}
	int h, w;
	PANO_HIP_CHECK(op_canvas_dims(cv, &h, &w));
	Mat32f out(h, w, 3);
	PANO_HIP_CHECK(op_canvas_copy(ctx, cv, out.ptr()));
	op_canvas_free(cv);
	return out;
}

#ifndef OPENPANO_WITH_REFERENCE
inline void ConnectedImages::prepare(bool set_inverse, bool set_range) {
	std::vector<double> hinv, ranges;
	const op_blend_geom g = hip_blend_prepare(*this, hinv, ranges);
	for (size_t i = 0; i < component.size(); ++i) {
		if (set_inverse) for (int k = 0; k < 9; ++k) component[i].homo_inv[k] = hinv[i * 9 + k];
		if (set_range) component[i].range = Range(Vec2D(ranges[4 * i], ranges[4 * i + 1]), Vec2D(ranges[4 * i + 2], ranges[4 * i + 3]));
	}
	if (set_range) {
		proj_range = Range(Vec2D(g.proj_min[0], g.proj_min[1]), Vec2D(g.proj_max[0], g.proj_max[1]));
		resolution = Vec2D(g.resolution[0], g.resolution[1]);
	}
}
inline Mat32f ConnectedImages::blend() const { return hip_blend(*this); }
#endif

#ifdef OPENPANO_WITH_REFERENCE
// Hook 6 (HOST-ONLY, optional): CameraEstimator{pairwise_matches, shapes}.estimate() of Stitcher::estimate_camera
// (stitch/stitcher.cc:143-146 -> camera_estimator.cc:46-103 -> incremental_bundle_adjuster.cc:117-385) through
// libpano_host.so, the Eigen-free estimator of host/pano_camera.hh: same constructor arguments, same estimate().  The
// reference's bundle adjuster materialises and zeroes a (2 x matches) x (6 x images) Jacobian per iteration
// (incremental_bundle_adjuster.cc:280); the mirror accumulates J^T J block by block in the reference's summation order --
// the cameras are the reference's, digit for digit (tests/test_camera_vs_ref.py; both sides then solve through the same
// QR / SVD arithmetic, i.e. parity is unpinned at Eigen's own rounding only).  Unlike the reference class this one does
// not write back into `matches` (camera_estimator.hh:33); Stitcher::build() clears them right after (stitcher.cc:54).
class HostCameraEstimator {
	public:
		HostCameraEstimator(std::vector<std::vector<MatchInfo>>& matches, const std::vector<Shape2D>& image_shapes):
			matches(matches), shapes(image_shapes) {}
		std::vector<Camera> estimate() {
			GuardedTimer tm("Estimate Camera");                  // the reference's own label (camera_estimator.cc:47)
			pano_config_set("STRAIGHTEN", (float)config::STRAIGHTEN); pano_config_set("MULTIPASS_BA", (float)config::MULTIPASS_BA);
			pano_config_set("LM_LAMBDA", (float)config::LM_LAMBDA); pano_config_set("ESTIMATE_CAMERA", (float)config::ESTIMATE_CAMERA);
			pano_config_set("ORDERED_INPUT", (float)config::ORDERED_INPUT); pano_config_set("TRANS", (float)config::TRANS);
			pano_config_set("CYLINDER", (float)config::CYLINDER);
			const int n = (int)matches.size();
			std::vector<int> wh, ij, cnt; std::vector<float> conf; std::vector<double> homo, pts;
			for (auto& s : shapes) { wh.push_back(s.w); wh.push_back(s.h); }
			for (int i = 0; i < n; ++i)
				for (int j = 0; j < n; ++j) {
					const MatchInfo& m = matches[i][j];
					if (m.match.empty() && m.confidence == 0) continue;           // never assigned by match_image (stitcher.cc:79-93)
					ij.push_back(i); ij.push_back(j); conf.push_back(m.confidence); cnt.push_back((int)m.match.size());
					for (int k = 0; k < 9; ++k) homo.push_back(m.homo[k]);
					for (auto& p : m.match) { pts.push_back(p.first.x); pts.push_back(p.first.y); pts.push_back(p.second.x); pts.push_back(p.second.y); }
				}
			std::vector<double> out((size_t)n * 13);
			if (pts.empty()) pts.push_back(0);
			pano_estimate_cameras(n, wh.data(), (int)cnt.size(), ij.data(), conf.data(), homo.data(), cnt.data(), pts.data(), out.data());
			std::vector<Camera> cams(n);
			for (int i = 0; i < n; ++i) {
				const double* o = out.data() + 13 * (size_t)i;
				cams[i].focal = o[0]; cams[i].aspect = o[1]; cams[i].ppx = o[2]; cams[i].ppy = o[3];
				for (int k = 0; k < 9; ++k) cams[i].R[k] = o[4 + k];
			}
			return cams;
		}
	private:
		std::vector<std::vector<MatchInfo>>& matches;
		const std::vector<Shape2D>& shapes;
};
#endif

// CylinderWarper (stitch/warp.hh:42-61): same method set
class HipCylinderWarper {
	public:
		explicit HipCylinderWarper(double m_hfactor): h_factor(m_hfactor) {}
		// warp image together with key points
		void warp(Mat32f& mat, std::vector<Vec2D>& kpts) const {
			op_ctx* ctx = HipContext::get();
			const op_config cfg = hip_config_snapshot();
			Shape2D shape(mat.width(), mat.height());
			op_image im{mat.ptr(), mat.rows(), mat.cols(), 0, OP_F32};
			op_canvas* cv = nullptr;
			PANO_HIP_CHECK(op_cyl_warp(ctx, &cfg, &im, h_factor, &cv));
			warp(shape, kpts);
			int h, w;
			PANO_HIP_CHECK(op_canvas_dims(cv, &h, &w));
			Mat32f out(h, w, 3);
			PANO_HIP_CHECK(op_canvas_copy(ctx, cv, out.ptr()));
			op_canvas_free(cv);
			mat = out;
		}
		// warp keypoints given image shape
		void warp(Shape2D& shape, std::vector<Vec2D>& kpts) const {
			const op_config cfg = hip_config_snapshot();
			std::vector<double> p(kpts.size() * 2 + 2);
			for (size_t i = 0; i < kpts.size(); ++i) { p[2 * i] = kpts[i].x; p[2 * i + 1] = kpts[i].y; }
			int nw, nh; double off[2];
			PANO_HIP_CHECK(op_cyl_warp_shape(&cfg, shape.w, shape.h, h_factor, kpts.empty() ? nullptr : p.data(), (int)kpts.size(), &nw, &nh, off));
			for (size_t i = 0; i < kpts.size(); ++i) kpts[i] = Vec2D(p[2 * i], p[2 * i + 1]);
			shape.w = nw; shape.h = nh;
		}
		// warp image only
		void warp(Mat32f& mat) const { std::vector<Vec2D> a; warp(mat, a); }
	protected:
		const double h_factor;
};

#ifndef OPENPANO_WITH_REFERENCE
// ===================================== STITCHER =====================================
// Stitcher (stitch/stitcher.hh:17-62, stitcher.cc:32-198) + StitcherBase (stitcherbase.hh:17-64),
// standalone: the whole of Stitcher::build() with the device stages batched -- one SIFT call for
// all images, one match call and one RANSAC call for the whole pair list -- and the host stages
// (camera estimation / bundle adjustment, pano_camera.hh) in between.  Members keep the
// reference's names; `cameras` and `base_seed` (RANSAC seed injection) are additions.
class HipStitcher {
	public:
		explicit HipStitcher(const std::vector<Mat32f>& mats, uint32_t base_seed = 42u): base_seed(base_seed) {
			if (mats.size() <= 1) { fprintf(stderr, "Cannot stitch with only %zu images.\n", mats.size()); exit(1); }   // stitcherbase.hh:46-48
			for (auto& m : mats) imgs.emplace_back(m);
			for (auto& r : imgs) bundle.component.emplace_back(&r);
		}
		HipStitcher(const HipStitcher&) = delete;
		HipStitcher& operator=(const HipStitcher&) = delete;

		Mat32f build() {                                            // stitcher.cc:32-64
			calc_feature();
			pairwise_matches.assign(imgs.size(), std::vector<MatchInfo>(imgs.size()));
			if (config::ORDERED_INPUT) linear_pairwise_match();
			else pairwise_match();
			assign_center();
			if (config::ESTIMATE_CAMERA) estimate_camera();
			else build_linear_simple();
			bundle.proj_method = config::ESTIMATE_CAMERA ? ConnectedImages::spherical : ConnectedImages::flat;
			bundle.update_proj_range();
			return bundle.blend();
		}

		std::vector<ImageRef> imgs;
		HipFeatureSet feats;                                        // StitcherBase::feats + the resident device copy
		std::vector<std::vector<Vec2D>> keypoints;
		std::vector<std::vector<MatchInfo>> pairwise_matches;
		ConnectedImages bundle;
		std::vector<Camera> cameras;
		uint32_t base_seed;

		void calc_feature() {                                       // stitcherbase.cc:9-27
			std::vector<const Mat32f*> ptrs;
			for (auto& r : imgs) ptrs.push_back(r.img);
			feats = HipSIFTDetector().calc_feature(ptrs);
			keypoints.resize(imgs.size());
			for (size_t k = 0; k < imgs.size(); ++k) {
				keypoints[k].clear();
				for (auto& d : feats.feats[k]) keypoints[k].emplace_back(d.coor);
			}
		}
		void pairwise_match() {                                     // stitcher.cc:96-113
			std::vector<std::pair<int, int>> tasks;
			for (int i = 0; i < (int)imgs.size(); ++i) for (int j = i + 1; j < (int)imgs.size(); ++j) tasks.emplace_back(i, j);
			match_tasks(tasks, false);
		}
		void linear_pairwise_match() {                              // stitcher.cc:115-136
			const int n = (int)imgs.size();
			std::vector<std::pair<int, int>> tasks;
			for (int i = 0; i < n; ++i) tasks.emplace_back(i, (i + 1) % n);
			match_tasks(tasks, true);
		}
		void assign_center() { bundle.identity_idx = (int)imgs.size() >> 1; }   // stitcher.cc:138-141
		void estimate_camera() {                                    // stitcher.cc:143-158
			std::vector<Shape2D> shapes;
			for (auto& m : imgs) shapes.emplace_back(m.shape());
			cameras = CameraEstimator{pairwise_matches, shapes}.estimate();
			for (size_t i = 0; i < imgs.size(); ++i) {
				bundle.component[i].homo_inv = cameras[i].K() * cameras[i].R;
				bundle.component[i].homo = cameras[i].Rinv() * cameras[i].K().inverse();
			}
		}
		void build_linear_simple() {                                // stitcher.cc:160-198
			const int n = (int)imgs.size(), mid = bundle.identity_idx;
			auto& comp = bundle.component;
			comp[mid].homo = Homography::I();
			if (mid + 1 < n) {
				comp[mid + 1].homo = pairwise_matches[mid][mid + 1].homo;
				for (int k = mid + 2; k < n; ++k) comp[k].homo = comp[k - 1].homo * pairwise_matches[k - 1][k].homo;
			}
			if (mid - 1 >= 0) {
				comp[mid - 1].homo = pairwise_matches[mid][mid - 1].homo;
				for (int k = mid - 2; k >= 0; --k) comp[k].homo = comp[k + 1].homo * pairwise_matches[k + 1][k].homo;
			}
			double f = -1;
			if (!config::TRANS) f = Camera::estimate_focal(pairwise_matches);
			if (f <= 0) f = 0.5 * (imgs[mid].width() + imgs[mid].height());
			for (int i = 0; i < n; ++i) {
				const double t[9] = {1.0 / f, 0, 0, 0, 1.0 / f, 0, 0, 0, 1};
				comp[i].homo = Homography(t) * comp[i].homo;
			}
			bundle.calc_inverse_homo();
		}

		// match_image (stitcher.cc:66-94) for a whole task list: one match call + one RANSAC call,
		// then the reference's bookkeeping per connected pair
		void match_tasks(const std::vector<std::pair<int, int>>& tasks, bool linear) {
			std::vector<Shape2D> shapes;
			for (auto& r : imgs) shapes.push_back(r.shape());
			record_matches(tasks, hip_match_images(feats, shapes, tasks, base_seed), linear);
		}
		void record_matches(const std::vector<std::pair<int, int>>& tasks, const std::vector<std::pair<bool, MatchInfo>>& infos, bool linear) {
			const int n = (int)imgs.size();
			for (size_t k = 0; k < tasks.size(); ++k) {
				const int i = tasks[k].first, j = tasks[k].second;
				if (!infos[k].first) {
					if (linear && i != n - 1) {       // head and tail don't have to match (stitcher.cc:123-127)
						fprintf(stderr, "error: Image %d and %d don't match\n", i, j);
						exit(1);
					}
					continue;
				}
				MatchInfo info = infos[k].second;
				Homography inv = info.homo.inverse();       // TransformEstimation ensures invertible
				inv.mult(1.0 / inv[8]);
				pairwise_matches[i][j] = info;
				info.homo = inv;
				info.reverse();
				pairwise_matches[j][i] = std::move(info);
			}
		}
};

// standalone builds read like the reference
using Stitcher = HipStitcher;
using SIFTDetector = HipSIFTDetector;
using PairWiseMatcher = HipPairWiseMatcher;
using TransformEstimation = HipTransformEstimation;
using CylinderWarper = HipCylinderWarper;
#endif

}	// namespace pano
