// pano_host_capi.cc -- C entry points over the Eigen-free host classes of pano_camera.hh, for
// callers that are not C++ (tests/test_camera_*.py through ctypes; any FFI).  Host-only: built
// with g++ into openpano_amd/libpano_host.so, no HIP.  Signatures mirror oracle/ref_driver.cc's
// ref_estimate_cameras so that the parity tests feed both sides the same arrays.
#include <algorithm>
#include <vector>

#include "pano_camera.hh"

using namespace pano;

// Only the C entry points leave this library (built with -fvisibility=hidden): its classes carry the REFERENCE's names
// (pano::Camera, pano::CameraEstimator, ...) and it is loaded into processes that contain the reference's own classes of
// those names (the hooked CLI, oracle/ref_stitch_test) -- exported, the two sets would interpose each other.
#pragma GCC visibility push(default)
extern "C" {

// The pair deal of a sharded job (openpano_amd/distributed.py: partition_pairs): pairs whose two images one rank owns go to
// that rank first (they are matched while the features travel), the rest is dealt longest first (cost K_i * K_j, ties by
// position in the list) to the rank with the least load so far (ties: the lower rank).  Deterministic, identical on every
// rank; mine[k] = 1 for the pairs of `rank`.  Plain host code: at 8128 pairs the Python loop it replaces took 6-19 ms of
// every exchange, more than a rank's share of the SIFT phase.  owner[g] = rank that owns image g, or -1; may be null.
int pano_deal_pairs(int npairs, const int* pairs, const long long* cost, int world, int rank, const int* owner, int nimg, unsigned char* mine) {
	if (npairs < 0 || world <= 0 || rank < 0 || rank >= world || (npairs && (!pairs || !cost || !mine))) return -1;
	std::vector<long long> load(world, 0);
	std::vector<int> rest; rest.reserve(npairs);
	for (int k = 0; k < npairs; ++k) {
		mine[k] = 0;
		const int i = pairs[2 * k], j = pairs[2 * k + 1];
		const int oi = (owner && i >= 0 && i < nimg) ? owner[i] : -1, oj = (owner && j >= 0 && j < nimg) ? owner[j] : -1;
		if (oi >= 0 && oi == oj && oi < world) { load[oi] += cost[k]; if (oi == rank) mine[k] = 1; }
		else rest.push_back(k);
	}
	std::stable_sort(rest.begin(), rest.end(), [&](int a, int b) { return cost[a] > cost[b]; });
	for (int k : rest) {
		int r = 0;
		for (int q = 1; q < world; ++q) if (load[q] < load[r]) r = q;
		load[r] += cost[k];
		if (r == rank) mine[k] = 1;
	}
	return 0;
}

int pano_config_set(const char* key, float v) {
	const std::string k(key);
#define CFG(x) if (k == #x) { config::x = v; return 0; }
	CFG(STRAIGHTEN) CFG(MULTIPASS_BA) CFG(LM_LAMBDA) CFG(ESTIMATE_CAMERA) CFG(ORDERED_INPUT) CFG(TRANS) CFG(CYLINDER)
#undef CFG
	return -1;
}

// CameraEstimator{pairwise_matches, shapes}.estimate(): np directed entries (i, j) ->
// pairwise_matches[i][j] = {conf, homo, pts rows (to.x, to.y, from.x, from.y)};
// out: per image focal, aspect, ppx, ppy, R[9] (13 doubles)
int pano_estimate_cameras(int n, const int* shapes_wh, int np, const int* ij, const float* conf, const double* homo,
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
// one Levenberg-Marquardt step with its internals (see oracle/ref_driver.cc: ref_iba_probe)
struct IbaProbe : public IncrementalBundleAdjuster {
	using IncrementalBundleAdjuster::IncrementalBundleAdjuster;
	void probe(int identity, double* resid, double* jtj, double* upd) {
		set_identity_idx(identity);
		update_index_map();
		const int nr_img = (int)idx_added.size();
		JtJ.assign((size_t)36 * nr_img * nr_img, 0.0); Jtr.assign((size_t)6 * nr_img, 0.0);
		ParamState state;
		for (auto& idx : idx_added) state.cameras.emplace_back(result_cameras[idx]);
		state.ensure_params();
		state.cameras.clear();
		auto err = calcError(state);
		for (size_t i = 0; i < err.residuals.size(); ++i) resid[i] = err.residuals[i];
		const std::vector<double> u = get_param_update(state, err.residuals, config::LM_LAMBDA);
		for (size_t i = 0; i < JtJ.size(); ++i) jtj[i] = JtJ[i];
		for (size_t i = 0; i < u.size(); ++i) upd[i] = u[i];
	}
};
int pano_iba_probe(int n, const double* cams, int np, const int* ij, const int* cnt, const double* pts, int identity,
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
void pano_rotation_to_angle(const double* r, double* v) {
	Homography h; for (int k = 0; k < 9; ++k) h[k] = r[k];
	Camera::rotation_to_angle(h, v[0], v[1], v[2]);
}
void pano_angle_to_rotation(const double* v, double* r) {
	Homography h; Camera::angle_to_rotation(v[0], v[1], v[2], h);
	for (int k = 0; k < 9; ++k) r[k] = h[k];
}
int pano_homography_inverse(const double* a, double* inv) {
	Homography h; for (int k = 0; k < 9; ++k) h[k] = a[k];
	bool ok = false;
	Homography r = h.inverse(&ok);
	if (ok) for (int k = 0; k < 9; ++k) inv[k] = r[k];
	return ok ? 1 : 0;
}
void pano_colpiv_solve(const double* A, int n, const double* b, double* x) { pano_la::colpiv_qr_solve(A, n, b, x); }

}	// extern "C"
#pragma GCC visibility pop
