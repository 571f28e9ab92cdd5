// multi.hip -- the hot path sharded over several GPUs of ONE process ("single
// process, one host thread + stream per device").
//
// The reference's two parallel loops are the axes: StitcherBase::calc_feature is an OpenMP loop over
// images (stitch/stitcherbase.cc:14-25), Stitcher::pairwise_match one over image pairs
// (stitch/stitcher.cc:96-113).  op_sift_batch_multi deals the images round-robin to the group's
// contexts in contiguous blocks (one host thread each) and all-gathers the features: EVERY device ends
// with the whole image-indexed table, each pulling the other devices' slices over its own xGMI links --
// the descriptor all-gather of SURVEY 8(e).2 as direct peer copies, because all devices belong to this
// process (the process-per-GPU form with RCCL lives in openpano_amd/distributed.py).  op_match_pairs_multi
// deals the pair list balanced by K_i * K_j and keeps every device's match lists where they were made;
// op_ransac_pairs_multi follows the same deal.  Results are identical to the single-device calls, item for item.
#include "internal.hpp"
#include <algorithm>
#include <numeric>
#include <thread>
#include <mutex>

struct op_features;
struct op_matches;
// sift_host.hip / match.hip internals
int op_features_allgather_blocks(op_ctx* const* ctxs, int nctx, op_features* const* parts, const int* start, int n, op_features** tables);
int op_features_replicate(op_ctx* dst, const op_features* f, op_features** out);
std::vector<op_features*>& op_features_replicas(op_features* f);
int op_features_device(const op_features* f);
op_matches* op_matches_merge(op_matches* const* parts, const std::vector<std::vector<int>>& index, int npairs);
const std::vector<op_matches*>& op_matches_parts(const op_matches* m);
const std::vector<std::vector<int>>& op_matches_part_index(const op_matches* m);
int op_matches_num_pairs(const op_matches* m);
struct op_ransac_result;
op_ransac_result* op_ransac_merge(op_ransac_result* const* parts, const std::vector<std::vector<int>>& index, int npairs);

struct op_group {
	std::vector<op_ctx*> ctxs;
};

extern "C" {

int op_group_create(const int* devices, int ndev, op_group** out) {
	if (!devices || ndev < 1 || !out) OP_FAIL(OP_ERR_INVALID, "op_group_create: bad argument");
	op_group* g = new op_group;
	for (int k = 0; k < ndev; ++k) {
		op_ctx* c = nullptr;
		const int rc = op_ctx_create(devices[k], nullptr, &c);
		if (rc != OP_OK) { for (op_ctx* x : g->ctxs) op_ctx_destroy(x); delete g; return rc; }
		g->ctxs.push_back(c);
	}
	// direct peer copies over xGMI where the pair of devices supports them (a failure only means staged copies)
	for (int a = 0; a < ndev; ++a) for (int b = 0; b < ndev; ++b) {
		if (devices[a] == devices[b]) continue;
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
			if (hipSetDevice(devices[a]) == hipSuccess) { hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0); (void)e; (void)hipGetLastError(); }
		}
	}
	*out = g;
	return OP_OK;
}
void op_group_destroy(op_group* g) { if (!g) return; for (op_ctx* c : g->ctxs) op_ctx_destroy(c); delete g; }
int op_group_size(const op_group* g) { return g ? (int)g->ctxs.size() : 0; }
op_ctx* op_group_ctx(op_group* g, int k) { return (g && k >= 0 && k < (int)g->ctxs.size()) ? g->ctxs[k] : nullptr; }

// image i of an n-image job -> context: contiguous blocks (n / nd images, +1 on the first n % nd contexts), the same
// deal as openpano_amd/distributed.py -- a context's features are ONE slice of the image-indexed table
static std::vector<int> block_starts(int n, int nd) {
	std::vector<int> st(nd + 1, 0);
	for (int k = 0; k < nd; ++k) st[k + 1] = st[k] + n / nd + (k < n % nd ? 1 : 0);
	return st;
}

int op_sift_batch_multi(op_group* g, const op_config* cfg, const op_image* imgs, int n, op_features** out) {
	if (!g || g->ctxs.empty() || !cfg || !imgs || n <= 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_sift_batch_multi: bad argument");
	const int nd = (int)std::min<size_t>(g->ctxs.size(), (size_t)n);
	if (nd == 1) return op_sift_batch(g->ctxs[0], cfg, imgs, n, out);
	const std::vector<int> st = block_starts(n, nd);
	std::vector<op_features*> parts(nd, nullptr);
	std::vector<int> rcs(nd, OP_OK);
	std::vector<std::string> errs(nd);
	std::vector<std::thread> th;
	for (int k = 0; k < nd; ++k)
		th.emplace_back([&, k] {
			rcs[k] = op_sift_batch(g->ctxs[k], cfg, imgs + st[k], st[k + 1] - st[k], &parts[k]);
			if (rcs[k] != OP_OK) errs[k] = op_last_error();                   // the error text is thread-local
		});
	for (auto& t : th) t.join();
	int rc = OP_OK;
	for (int k = 0; k < nd; ++k) if (rcs[k] != OP_OK && rc == OP_OK) { rc = rcs[k]; op_set_error("device shard " + std::to_string(k) + ": " + errs[k]); }
	if (rc == OP_OK) {
		// the all-gather: every device ends with the whole table (the matcher and RANSAC need it everywhere)
		std::vector<op_features*> tables(nd, nullptr);
		rc = op_features_allgather_blocks(g->ctxs.data(), nd, parts.data(), st.data(), n, tables.data());
		if (rc == OP_OK) {
			std::vector<op_features*>& rep = op_features_replicas(tables[0]);
			rep.assign(g->ctxs.size(), nullptr);
			for (int k = 1; k < nd; ++k) rep[k] = tables[k];
			*out = tables[0];
		}
	}
	for (op_features* p : parts) op_features_free(p);
	return rc;
}

// the table of f on every device of the group: replicas made by op_sift_batch_multi, or pulled from f's device now
// (in parallel, one host thread per destination) and kept with f
static int ensure_replicas(op_group* g, const op_features* f, int nd) {
	// the replicas are cached on f by slot; a slot is only reused for the device it was made for (the same features
	// may meet a group of other devices, or the same devices in another order), and f itself must be slot 0's table
	static std::mutex rep_mu;
	std::lock_guard<std::mutex> lk(rep_mu);
	if (op_features_device(f) != g->ctxs[0]->device) OP_FAIL(OP_ERR_INVALID, "op_group: the features do not live on the group's first device");
	std::vector<op_features*>& rep = op_features_replicas(const_cast<op_features*>(f));
	if (rep.size() < g->ctxs.size()) rep.resize(g->ctxs.size(), nullptr);
	for (int k = 1; k < nd; ++k)
		if (rep[k] && op_features_device(rep[k]) != g->ctxs[k]->device) { op_features_free(rep[k]); rep[k] = nullptr; }
	std::vector<int> rcs(nd, OP_OK);
	std::vector<std::string> errs(nd);
	std::vector<std::thread> th;
	for (int k = 1; k < nd; ++k)
		if (!rep[k])
			th.emplace_back([&, k] {
				rcs[k] = op_features_replicate(g->ctxs[k], f, &rep[k]);
				if (rcs[k] != OP_OK) errs[k] = op_last_error();
			});
	for (auto& t : th) t.join();
	for (int k = 1; k < nd; ++k) if (rcs[k] != OP_OK) { op_set_error("device " + std::to_string(k) + ": " + errs[k]); return rcs[k]; }
	return OP_OK;
}

int op_match_pairs_multi(op_group* g, const op_config* cfg, const op_features* f, const int* pairs, int npairs, op_matches** out) {
	if (!g || g->ctxs.empty() || !cfg || !f || !pairs || npairs < 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_match_pairs_multi: bad argument");
	const int nd = (int)std::min<size_t>(g->ctxs.size(), (size_t)std::max(npairs, 1));
	if (nd == 1) return op_match_pairs(g->ctxs[0], cfg, f, pairs, npairs, out);
	const int nimg = op_features_num_images(f);
	// the deal: longest pair first onto the least loaded device (cost K_i * K_j), ties by index -- the same
	// partition openpano_amd/distributed.py makes
	std::vector<long long> cost(npairs);
	for (int p = 0; p < npairs; ++p) {
		const int i = pairs[2 * p], j = pairs[2 * p + 1];
		if (i < 0 || j < 0 || i >= nimg || j >= nimg) OP_FAIL(OP_ERR_INVALID, "op_match_pairs_multi: image index out of range");
		cost[p] = (long long)op_features_count(f, i) * op_features_count(f, j);
	}
	std::vector<int> order(npairs);
	std::iota(order.begin(), order.end(), 0);
	std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
	std::vector<long long> load(nd, 0);
	std::vector<std::vector<int>> mine(nd);
	for (int p : order) {
		const int k = (int)(std::min_element(load.begin(), load.end()) - load.begin());
		load[k] += cost[p]; mine[k].push_back(p);
	}
	for (auto& v : mine) std::sort(v.begin(), v.end());
	int rc = ensure_replicas(g, f, nd);
	if (rc != OP_OK) return rc;
	const std::vector<op_features*>& rep = op_features_replicas(const_cast<op_features*>(f));
	std::vector<op_matches*> parts(nd, nullptr);
	std::vector<int> rcs(nd, OP_OK);
	std::vector<std::string> errs(nd);
	std::vector<std::thread> th;
	for (int k = 0; k < nd; ++k)
		th.emplace_back([&, k] {
			std::vector<int> pr;
			for (int p : mine[k]) { pr.push_back(pairs[2 * p]); pr.push_back(pairs[2 * p + 1]); }
			rcs[k] = op_match_pairs(g->ctxs[k], cfg, k == 0 ? f : rep[k], pr.data(), (int)mine[k].size(), &parts[k]);
			if (rcs[k] != OP_OK) errs[k] = op_last_error();
		});
	for (auto& t : th) t.join();
	for (int k = 0; k < nd; ++k) if (rcs[k] != OP_OK && rc == OP_OK) { rc = rcs[k]; op_set_error("device shard " + std::to_string(k) + ": " + errs[k]); }
	if (rc == OP_OK) *out = op_matches_merge(parts.data(), mine, npairs);     // owns the parts from here on
	else for (op_matches* m : parts) op_matches_free(m);
	return rc;
}

// TransformEstimation for a job that op_match_pairs_multi matched: every device runs RANSAC on ITS pairs -- the match
// lists are resident there, the keypoint coordinates are in its replica of the table -- with the seeds the single-device
// call would give those pairs; results come back in the order of `pairs`.
int op_ransac_pairs_multi(op_group* g, const op_config* cfg, const op_features* f, const op_matches* m,
		const int* pairs, int npairs, const int* shapes_wh, const uint32_t* seeds, uint32_t base_seed, op_ransac_result** out) {
	if (!g || g->ctxs.empty() || !cfg || !f || !m || !pairs || npairs < 0 || !shapes_wh || !out) OP_FAIL(OP_ERR_INVALID, "op_ransac_pairs_multi: bad argument");
	const std::vector<op_matches*>& parts = op_matches_parts(m);
	const std::vector<std::vector<int>>& index = op_matches_part_index(m);
	const int nd = (int)parts.size();
	if (nd <= 1 || nd > (int)g->ctxs.size())                                  // matched on one device (or wrapped from host lists)
		return op_ransac_pairs(g->ctxs[0], cfg, f, m, pairs, npairs, shapes_wh, seeds, base_seed, out);
	if (op_matches_num_pairs(m) != npairs) OP_FAIL(OP_ERR_INVALID, "op_ransac_pairs_multi: op_matches holds a different number of pairs than the pair list");
	int rc = ensure_replicas(g, f, nd);
	if (rc != OP_OK) return rc;
	const std::vector<op_features*>& rep = op_features_replicas(const_cast<op_features*>(f));
	std::vector<op_ransac_result*> res(nd, nullptr);
	std::vector<int> rcs(nd, OP_OK);
	std::vector<std::string> errs(nd);
	std::vector<std::thread> th;
	for (int k = 0; k < nd; ++k)
		th.emplace_back([&, k] {
			std::vector<int> pr; std::vector<uint32_t> sd;
			for (int p : index[k]) {
				pr.push_back(pairs[2 * p]); pr.push_back(pairs[2 * p + 1]);
				sd.push_back(seeds ? seeds[p] : (base_seed * 2654435761u) ^ (uint32_t)(p * 40503u + 12345u));   // op_ransac_pairs' own derivation, by JOB index
			}
			rcs[k] = op_ransac_pairs(g->ctxs[k], cfg, k == 0 ? f : rep[k], parts[k], pr.data(), (int)index[k].size(), shapes_wh, sd.data(), 0, &res[k]);
			if (rcs[k] != OP_OK) errs[k] = op_last_error();
		});
	for (auto& t : th) t.join();
	for (int k = 0; k < nd; ++k) if (rcs[k] != OP_OK && rc == OP_OK) { rc = rcs[k]; op_set_error("device shard " + std::to_string(k) + ": " + errs[k]); }
	if (rc == OP_OK) *out = op_ransac_merge(res.data(), index, npairs);
	for (op_ransac_result* r : res) op_ransac_free(r);
	return rc;
}

}	// extern "C"
