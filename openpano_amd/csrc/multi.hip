// multi.hip -- the hot path sharded over several GPUs of ONE process ("single
// process, one host thread + stream per device").
//
// The reference's two parallel loops are the axes: StitcherBase::calc_feature is an OpenMP loop over
// images (stitch/stitcherbase.cc:14-25), Stitcher::pairwise_match one over image pairs
// (stitch/stitcher.cc:96-113).  op_sift_batch_multi deals the images round-robin to the group's
// contexts (one host thread each) and gathers the features into one table on the first device;
// op_match_pairs_multi replicates that table to every device -- the descriptor all-gather of
// SURVEY 8(e).2, as device-to-device copies over xGMI because all devices belong to this process (the
// process-per-GPU form with RCCL lives in openpano_amd/distributed.py) -- and deals the pair list
// balanced by K_i * K_j.  Results are identical to the single-device calls, item for item.
#include "internal.hpp"
#include <algorithm>
#include <numeric>
#include <thread>

struct op_features;
struct op_matches;
// sift_host.hip / match.hip internals
int op_features_gather_sharded(op_ctx* dst, op_features* const* parts, int nparts, int n, op_features** out);
int op_features_replicate(op_ctx* dst, const op_features* f, op_features** out);
op_matches* op_matches_merge(op_matches* const* parts, const std::vector<std::vector<int>>& index, int npairs);

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

int op_sift_batch_multi(op_group* g, const op_config* cfg, const op_image* imgs, int n, op_features** out) {
	if (!g || g->ctxs.empty() || !cfg || !imgs || n <= 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_sift_batch_multi: bad argument");
	const int nd = (int)std::min<size_t>(g->ctxs.size(), (size_t)n);
	if (nd == 1) return op_sift_batch(g->ctxs[0], cfg, imgs, n, out);
	std::vector<std::vector<op_image>> shard(nd);
	for (int i = 0; i < n; ++i) shard[i % nd].push_back(imgs[i]);           // image i -> context i % nd, local index i / nd
	std::vector<op_features*> parts(nd, nullptr);
	std::vector<int> rcs(nd, OP_OK);
	std::vector<std::string> errs(nd);
	std::vector<std::thread> th;
	for (int k = 0; k < nd; ++k)
		th.emplace_back([&, k] {
			rcs[k] = op_sift_batch(g->ctxs[k], cfg, shard[k].data(), (int)shard[k].size(), &parts[k]);
			if (rcs[k] != OP_OK) errs[k] = op_last_error();                   // the error text is thread-local
		});
	for (auto& t : th) t.join();
	int rc = OP_OK;
	for (int k = 0; k < nd; ++k) if (rcs[k] != OP_OK && rc == OP_OK) { rc = rcs[k]; op_set_error("device shard " + std::to_string(k) + ": " + errs[k]); }
	if (rc == OP_OK) rc = op_features_gather_sharded(g->ctxs[0], parts.data(), nd, n, out);
	for (op_features* p : parts) op_features_free(p);
	return rc;
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
	std::vector<op_matches*> parts(nd, nullptr);
	std::vector<op_features*> replica(nd, nullptr);
	std::vector<int> rcs(nd, OP_OK);
	std::vector<std::string> errs(nd);
	std::vector<std::thread> th;
	for (int k = 0; k < nd; ++k)
		th.emplace_back([&, k] {
			const op_features* fk = f;
			if (k > 0) {                                                    // context 0 holds the table already
				rcs[k] = op_features_replicate(g->ctxs[k], f, &replica[k]);
				if (rcs[k] != OP_OK) { errs[k] = op_last_error(); return; }
				fk = replica[k];
			}
			std::vector<int> pr;
			for (int p : mine[k]) { pr.push_back(pairs[2 * p]); pr.push_back(pairs[2 * p + 1]); }
			rcs[k] = op_match_pairs(g->ctxs[k], cfg, fk, pr.data(), (int)mine[k].size(), &parts[k]);
			if (rcs[k] != OP_OK) errs[k] = op_last_error();
		});
	for (auto& t : th) t.join();
	int rc = OP_OK;
	for (int k = 0; k < nd; ++k) if (rcs[k] != OP_OK && rc == OP_OK) { rc = rcs[k]; op_set_error("device shard " + std::to_string(k) + ": " + errs[k]); }
	if (rc == OP_OK) { *out = op_matches_merge(parts.data(), mine, npairs); if (!*out) rc = OP_ERR_HIP; }
	for (op_matches* m : parts) op_matches_free(m);
	for (op_features* r : replica) op_features_free(r);
	return rc;
}

}	// extern "C"
