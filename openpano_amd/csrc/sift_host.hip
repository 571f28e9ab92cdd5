// sift_host.hip -- host orchestration of the batched SIFT op (op_sift_batch / op_sift_staged).
//
// Mirrors SIFTDetector::do_detect_feature (feature/feature.cc:31-47) for n images at once:
// one launch per stage for the whole batch (grid z/y = image), everything resident in HBM,
// two small D2H reads of per-image counts (the caller needs K_i anyway).
#include "internal.hpp"
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

size_t pyramid_lds_bytes(int halo);

namespace {

struct DevBuf {
	void* p = nullptr; size_t cap = 0;
	hipError_t ensure(size_t bytes) {
		if (bytes <= cap) return hipSuccess;
		if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
		size_t want = bytes + bytes / 8;
		hipError_t e = hipMalloc(&p, want);
		if (e != hipSuccess) return e;
		cap = want;
		return hipSuccess;
	}
	void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
};

}	// namespace

struct Workspace {
	DevBuf ws, work, srcs, staging, raw, counts, refinedA, refinedB, slot_of, dirs, ndirs, oriented;
	void* pinned = nullptr; size_t pinned_cap = 0;   // host-pinned scratch: source pointer table, counter block
	std::vector<const void*> srcs_last; void* srcs_dev = nullptr;   // the pointer table the device holds (and where)
	long long last_total = 0;                        // descriptors of the previous batch: predicts output capacity
	int last_raw_max = 0, last_refined_max = 0;      // longest per-image lists of the previous batch: size the refine / sort launches
	int raw_cap = 16384;                             // per-image capacity of the raw / refined lists; grows on overflow (run_group)
	int desc_list_cap = OP_DESC_LIST_CAP;            // list arena of the descriptor kernel's sorting pass (op_debug_set_desc_list_cap)
	void release() {
		ws.release(); work.release(); srcs.release(); staging.release(); raw.release(); counts.release();
		refinedA.release(); refinedB.release(); slot_of.release(); dirs.release(); ndirs.release(); oriented.release();
		if (pinned) hipHostFree(pinned); pinned = nullptr; pinned_cap = 0;
		srcs_last.clear(); srcs_dev = nullptr;
	}
};

// contexts are few and long-lived; the table is shared by the host threads that own them (the
// reference calls detect_feature concurrently from OpenMP threads, stitcherbase.cc:14)
static std::map<op_ctx*, Workspace*> g_ws;
static std::mutex g_ws_mu;

static Workspace* ctx_workspace(op_ctx* c) {
	std::lock_guard<std::mutex> lk(g_ws_mu);
	auto it = g_ws.find(c);
	if (it != g_ws.end()) return it->second;
	Workspace* w = new Workspace;
	g_ws[c] = w;
	return w;
}
void op_ctx_release_workspace(op_ctx* c) {
	Workspace* w = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_ws_mu);
		auto it = g_ws.find(c);
		if (it == g_ws.end()) return;
		w = it->second; g_ws.erase(it);
	}
	w->release(); delete w;
}

struct op_features {
	int n = 0;
	std::vector<int> counts;
	std::vector<int64_t> offsets;      // n + 1
	float* desc = nullptr;             // device, total x 128
	double* coor = nullptr;            // device, total x 2 (centred original-image pixels)
	double* real = nullptr;            // device, total x 2 (real_coor in [0,1)); null when built from host/device arrays
	bool has_desc = true;              // false: coordinates only (RANSAC-only use)
	bool owns = true;                  // false: desc / coor belong to the caller (op_features_adopt_device)
	int device = 0;
	// multi.hip: the same table on the other devices of an op_group (index = context of the group; [0] unused),
	// filled by the all-gather of op_sift_batch_multi / the first op_match_pairs_multi and owned by this object
	std::vector<op_features*> replicas;
	// host mirror of coor, fetched the first time a host stage asks for it (the acceptance epilogue of
	// op_ransac_pairs walks every keypoint of both images; features are immutable, so one copy serves every call)
	mutable std::vector<double> h_coor; mutable bool h_coor_valid = false; mutable std::mutex h_mu;
};

struct FeatView { int n; const int* counts; const int64_t* offsets; const float* desc; int device; };
FeatView op_features_view(const op_features* f) {
	return FeatView{f->n, f->counts.data(), f->offsets.data(), f->has_desc ? f->desc : nullptr, f->device};
}

// total x 2 centred coordinates on the host (nullptr + error set on failure); ordered on ctx's stream
const double* op_features_coor_host(const op_features* f, op_ctx* ctx) {
	std::lock_guard<std::mutex> lk(f->h_mu);
	if (f->h_coor_valid) return f->h_coor.data();
	const int64_t total = f->offsets[f->n];
	f->h_coor.resize((size_t)std::max<int64_t>(total, 1) * 2);
	if (total) {
		hipError_t e = hipMemcpyAsync(f->h_coor.data(), f->coor, sizeof(double) * 2 * (size_t)total, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess) { op_set_error(std::string("op_features: coordinate copy failed: ") + hipGetErrorString(e)); return nullptr; }
	}
	f->h_coor_valid = true;
	return f->h_coor.data();
}

struct op_sift_dump {
	Workspace w;                       // private workspace kept alive for plane reads
	SiftPlan plan;
	int cap = 0;
	std::vector<int> raw;              // sorted (o, s, y, x) quads: x, y, o, s
	std::vector<KeyPoint> refined, oriented;
	std::vector<float> desc; std::vector<double> coor01;
};

namespace {

// feature/gaussian.cc:17-40 (GaussCache) + gaussian.hh:96-103 (sigma bank), host side
int build_gauss_bank(const op_config& cfg, SiftPlan& p) {
	float sigma = cfg.GAUSS_SIGMA;
	p.halo = 0;
	memset(p.kern, 0, sizeof(p.kern));
	memset(p.kcenter, 0, sizeof(p.kcenter));
	for (int s = 1; s < cfg.NUM_SCALE; ++s) {
		int kw = (int)(std::ceil(0.3 * (sigma / 2 - 1) + 0.8) * cfg.GAUSS_WINDOW_FACTOR);
		if (kw % 2 == 0) kw++;
		const int center = kw / 2;
		if (center > OP_MAX_KCENTER || kw < 1) return -1;
		float* kernel = &p.kern[s][OP_MAX_KCENTER];
		kernel[0] = 1;
		float exp_coeff = (float)(-1.0 / (sigma * sigma * 2)), wsum = 1;
		for (int i = 1; i <= center; i++)
			wsum += (kernel[i] = std::exp((float)(i * i) * exp_coeff)) * 2;   // std::exp(float) = expf
		float fac = (float)(1.0 / wsum);
		kernel[0] = fac;
		for (int i = 1; i <= center; i++)
			kernel[-i] = (kernel[i] *= fac);
		p.kcenter[s] = center;
		p.halo = std::max(p.halo, center);
		sigma *= cfg.SCALE_FACTOR;
	}
	return 0;
}

int build_plan(const op_config& cfg, int n, int sh, int sw, SiftPlan& p, int num_cu = 256) {
	memset(&p, 0, sizeof(p));
	if (cfg.NUM_OCTAVE < 1 || cfg.NUM_OCTAVE > OP_MAX_OCT || cfg.NUM_SCALE < 4 || cfg.NUM_SCALE > OP_MAX_SCALE)
		OP_FAIL(OP_ERR_UNSUPPORTED, "NUM_OCTAVE must be in [1,8] and NUM_SCALE in [4,12]");
	if (sh < 2 || sw < 2) OP_FAIL(OP_ERR_INVALID, "image must be at least 2x2 (lib/imgproc.cc:321)");
	p.n = n; p.sh = sh; p.sw = sw;
	// feature/feature.cc:33-34
	float ratio = cfg.SIFT_WORKING_SIZE * 2.0f / (sw + sh);
	p.wh = (int)(sh * ratio); p.ww = (int)(sw * ratio);
	if (p.wh < 2 || p.ww < 2) OP_FAIL(OP_ERR_INVALID, "working image degenerate");
	p.noct = cfg.NUM_OCTAVE; p.nscale = cfg.NUM_SCALE;
	long long off = 0; int tile_begin = 0;
	for (int i = 0; i < p.noct; ++i) {
		OctDesc& o = p.oct[i];
		if (i == 0) { o.h = p.wh; o.w = p.ww; }
		else {      // feature/dog.cc:105-107
			float factor = (float)std::pow((double)cfg.SCALE_FACTOR, (double)-i);
			o.w = (int)std::ceil(p.ww * factor); o.h = (int)std::ceil(p.wh * factor);
			if (!(o.w > 5 && o.h > 5)) OP_FAIL(OP_ERR_INVALID, "octave smaller than 6 px (feature/dog.cc:108 assertion)");
		}
		o.plane = (long long)o.h * o.w;
		o.off = off; off += o.plane * planes_per_octave(p.nscale);
		o.tiles_x = (o.w + OP_PYR_TW - 1) / OP_PYR_TW; o.tiles_y = (o.h + OP_PYR_TH - 1) / OP_PYR_TH;
		o.tile_begin = tile_begin; tile_begin += o.tiles_x * o.tiles_y;
	}
	p.total_tiles = tile_begin;
	p.ws_stride = (off + 63) & ~63LL;
	if (build_gauss_bank(cfg, p) != 0) OP_FAIL(OP_ERR_UNSUPPORTED, "Gaussian kernel wider than 31 taps");
	{
		static const int shipped[6] = {3, 3, 3, 6, 6, 6};
		p.rows_ok = p.nscale == 7;
		for (int s = 1; s < 7 && p.rows_ok; ++s) if (p.kcenter[s] != shipped[s - 1]) p.rows_ok = 0;
		if (p.rows_ok) {
			// work items of k_pyramid_rows: bands of OP_RW_OWN columns x segments of rw_seg rows.
			// Every segment re-reads 14 halo rows and adds three rows of redundant passes; long segments pay in ramp-up
			// and tail (a workgroup's lifetime grows with the segment).  On a large batch the height hardly matters
			// (config 4, 38 images = 8 fills of the device: 20 / 24 / 28 / 40 / 48 rows = 0.376 / 0.374 / 0.373 / 0.371 / 0.376 ms,
			// profiles/r06_sift_seg.txt).  On a small one it decides how many times the device is filled: four workgroups
			// fit a CU (LDS), and 5 images at 24 rows are 1065 workgroups on 1024 places -- 41 of them run alone behind the
			// rest and the kernel takes two workgroup lifetimes instead of one.  So the height is chosen per batch: the one
			// that minimises (fills of the device, rounded up) x (a workgroup's steps at that height).
			auto items_at = [&](int seg) {
				long long it = 0;
				for (int i = 0; i < p.noct; ++i) it += (long long)((p.oct[i].w + OP_RW_OWN - 1) / OP_RW_OWN) * ((p.oct[i].h + seg - 1) / seg);
				return it;
			};
			int seg = OP_RW_SEG_DEFAULT;
#ifdef OP_RW_SEG
			seg = OP_RW_SEG;
#else
			{
				const long long places = (long long)(num_cu > 0 ? num_cu : 256) * 4;
				auto cost_at = [&](int c) {
					const long long items = items_at(c) * n;
					const long long fills = (items + places - 1) / places;
					return (double)fills * ((c + 3) / 2 + 1.2);       // steps of a workgroup: row pairs + the 14-row window fill (~1.2 steps)
				};
				const double dflt = cost_at(OP_RW_SEG_DEFAULT);
				double best = dflt;
				for (int c = OP_RW_SEG_MIN; c <= OP_RW_SEG_MAX; c += 2) {
					const double cost = cost_at(c);
					if (cost < best) { best = cost; seg = c; }
				}
				if (best > 0.93 * dflt) seg = OP_RW_SEG_DEFAULT;       // the model is coarse: only a clear win moves the height (measured: equal within 1 % on 8 fills)
			}
#endif
			p.rw_seg = seg;
			p.rw_items = 0;
			for (int i = 0; i < p.noct; ++i) {
				OctDesc& o = p.oct[i];
				o.rw_nb = (o.w + OP_RW_OWN - 1) / OP_RW_OWN;
				o.rw_nseg = (o.h + seg - 1) / seg;
				p.rw_items += o.rw_nb * o.rw_nseg;
			}
		}
		if (p.rows_ok)
			for (int pl = 0; pl < 3; ++pl) for (int d = 0; d < 7; ++d) for (int e = 0; e < 2; ++e) {
				const int s = 2 * pl + 1 + e;
				p.kpair[pl][d][e] = d <= p.kcenter[s] ? p.kern[s][OP_MAX_KCENTER + d] : 0.f;
			}
	}
	if (pyramid_lds_bytes(p.halo) > 160 * 1024 - 256) OP_FAIL(OP_ERR_UNSUPPORTED, "Gaussian halo does not fit LDS");
	p.pre_color_thres = cfg.PRE_COLOR_THRES; p.judge_thres = cfg.JUDGE_EXTREMA_DIFF_THRES;
	p.contrast_thres = cfg.CONTRAST_THRES; p.edge_ratio = cfg.EDGE_RATIO; p.offset_thres = cfg.OFFSET_THRES;
	p.calc_offset_depth = cfg.CALC_OFFSET_DEPTH;
	p.gauss_sigma = cfg.GAUSS_SIGMA; p.scale_factor = cfg.SCALE_FACTOR; p.ori_radius = cfg.ORI_RADIUS;
	p.ori_smooth = cfg.ORI_HIST_SMOOTH_COUNT; p.desc_scale_factor = cfg.DESC_HIST_SCALE_FACTOR;
	p.desc_int_factor = cfg.DESC_INT_FACTOR;
	return OP_OK;
}

struct GroupResult {
	std::vector<int> counts;          // per image of the group
	float* desc = nullptr; double* coor = nullptr; double* real = nullptr;   // device (hipMalloc), group-local flat
	long long total = 0;
};

// Run the whole pipeline for images of one size. `keep` (staged dump) additionally copies the
// intermediate lists to the host.
int run_group(op_ctx* ctx, const op_config& cfg, const std::vector<const op_image*>& imgs, Workspace& W,
		SiftPlan& plan, GroupResult& res, op_sift_dump* keep) {
	const int n = (int)imgs.size();
	const int sh = imgs[0]->h, sw = imgs[0]->w;
	int rc = build_plan(cfg, n, sh, sw, plan, ctx->num_cu);
	if (rc != OP_OK) return rc;
	hipStream_t st = ctx->stream;
	// per-image capacity of the raw / refined lists: speculative like capK below.  The kernels clamp
	// their writes and keep counting; a batch whose densest image outgrew the capacity grows the
	// buffers to the observed maximum and re-runs once (the reference has no such limit:
	// extrema.cc:36-61 appends to std::vectors).  The capacity sticks to the context.
	const int cap = W.raw_cap;
	plan.desc_list_cap = W.desc_list_cap;

	// --- buffers
	HIPCHK(W.ws.ensure(sizeof(float) * (size_t)plan.ws_stride * n));
	if (keep) HIPCHK(W.work.ensure(sizeof(float) * (size_t)plan.wh * plan.ww * 3 * n));
	HIPCHK(W.srcs.ensure(sizeof(float*) * n));
	HIPCHK(W.raw.ensure(sizeof(int) * 4 * (size_t)cap * n));
	// batch total (64 bits) | raw | refined | oriented (the block the host reads) | padding to a line | k_orientation's counters, one per line
	const size_t cnt_hdr = ((3 * (size_t)n + 2) + 31) & ~(size_t)31;
	HIPCHK(W.counts.ensure(sizeof(int) * (cnt_hdr + (size_t)n * OP_OCNT_STRIDE)));
	HIPCHK(W.refinedA.ensure(sizeof(KeyPoint) * (size_t)cap * n));
	HIPCHK(W.refinedB.ensure(sizeof(KeyPoint) * (size_t)cap * n));
	HIPCHK(W.dirs.ensure(sizeof(float) * 36 * (size_t)cap * n));
	HIPCHK(W.ndirs.ensure(sizeof(int) * (size_t)cap * n));
	HIPCHK(W.slot_of.ensure(sizeof(int) * (size_t)cap * n));
	const size_t pin_need = sizeof(long long) * (size_t)(8 * n + 16);
	if (W.pinned_cap < pin_need) {
		if (W.pinned) hipHostFree(W.pinned);
		HIPCHK(hipHostMalloc(&W.pinned, pin_need));
		W.pinned_cap = pin_need;
	}
	plan.ws = (float*)W.ws.p; plan.work = (float*)W.work.p;

	// --- sources: device pointers as they are, host images staged through one H2D copy each
	plan.src_u8 = imgs[0]->dtype == OP_U8 ? 1 : 0;
	const size_t img_bytes = (plan.src_u8 ? 1 : sizeof(float)) * (size_t)sh * sw * 3;
	const size_t img_stride = (img_bytes + 255) & ~(size_t)255;
	size_t host_bytes = 0;
	for (auto* im : imgs) if (!im->on_device) host_bytes += img_stride;
	if (host_bytes) HIPCHK(W.staging.ensure(host_bytes));
	{
		const void** hs = (const void**)W.pinned;
		// host images that lie at ONE constant stride >= their size (a decoder's output pool, one block sliced into
		// frames; a contiguous array of images: stride == size) travel in ONE strided copy: 38 separate 3 MB copies
		// reach about half the link rate.  Any other arrangement is copied image by image.
		bool packed = host_bytes == img_stride * (size_t)n && n > 1;
		ptrdiff_t hstride = 0;
		if (packed) hstride = (const char*)imgs[1]->data - (const char*)imgs[0]->data;
		packed = packed && hstride >= (ptrdiff_t)img_bytes;
		for (int i = 2; i < n && packed; ++i) packed = (const char*)imgs[i]->data - (const char*)imgs[i - 1]->data == hstride;
		if (packed) {
			if ((size_t)hstride == img_stride) HIPCHK(hipMemcpyAsync(W.staging.p, imgs[0]->data, img_stride * (size_t)(n - 1) + img_bytes, hipMemcpyHostToDevice, st));
			else HIPCHK(hipMemcpy2DAsync(W.staging.p, img_stride, imgs[0]->data, (size_t)hstride, img_bytes, (size_t)n, hipMemcpyHostToDevice, st));
			for (int i = 0; i < n; ++i) hs[i] = (char*)W.staging.p + (size_t)i * img_stride;
		} else {
			size_t so = 0;
			for (int i = 0; i < n; ++i) {
				if (imgs[i]->on_device) hs[i] = imgs[i]->data;
				else {
					void* d = (char*)W.staging.p + so;
					HIPCHK(hipMemcpyAsync(d, imgs[i]->data, img_bytes, hipMemcpyHostToDevice, st));
					hs[i] = d; so += img_stride;
				}
			}
		}
		// the pointer table only travels when it differs from the one the device already holds (a caller that keeps
		// its images resident passes the same pointers batch after batch)
		if (W.srcs_dev != W.srcs.p || W.srcs_last.size() != (size_t)n || !std::equal(W.srcs_last.begin(), W.srcs_last.end(), hs)) {
			HIPCHK(hipMemcpyAsync(W.srcs.p, hs, sizeof(void*) * n, hipMemcpyHostToDevice, st));
			W.srcs_last.assign(hs, hs + n); W.srcs_dev = W.srcs.p;
		}
	}
	plan.srcs = (const void* const*)W.srcs.p;

	long long* d_total = (long long*)W.counts.p;
	int* d_raw_count = (int*)W.counts.p + 2;
	int* d_refined_count = d_raw_count + n;
	int* d_oriented_count = d_raw_count + 2 * n;
	int* d_ocnt = (int*)W.counts.p + cnt_hdr;
	plan.num_cu = ctx->num_cu;
	plan.zero = (int*)W.counts.p; plan.zero_n = (int)(cnt_hdr + (size_t)n * OP_OCNT_STRIDE);      // cleared by k_grey_octaves

	{ ProfScope ps(ctx, "resize + octave grey"); HIPCHK(launch_grey_octaves(plan, keep != nullptr, st)); }
	{ ProfScope ps(ctx, "build pyramid"); HIPCHK(launch_pyramid(plan, (int*)W.raw.p, d_raw_count, cap, st)); }
	{ ProfScope ps(ctx, "extrema refine");
	  HIPCHK(launch_refine(plan, (const int*)W.raw.p, d_raw_count, cap, W.last_raw_max, (KeyPoint*)W.refinedA.p, d_refined_count, st)); }
	{ ProfScope ps(ctx, "orientation");       // histograms + peaks on the unsorted list; the canonical order (refinedB, slot_of) beside them
	  HIPCHK(launch_orientation(plan, (const KeyPoint*)W.refinedA.p, d_refined_count, cap, W.last_refined_max, (float*)W.dirs.p, (int*)W.ndirs.p, d_ocnt,
				(KeyPoint*)W.refinedB.p, (int*)W.slot_of.p, st)); }

	// The descriptor count is only known on the device at this point.  Instead of a round trip,
	// the output buffers get a capacity predicted from the previous call of this context (x1.25,
	// at least 2048 per image), output offsets and the total are computed on the device and the remaining stages are
	// enqueued right away; ONE synchronisation at the end returns the counts, and a batch that
	// outgrew the prediction re-runs its last two stages with exact sizes.
	long long capK = std::max<long long>((long long)n * 2048, W.last_total + W.last_total / 4);
	long long* h_total = (long long*)((char*)W.pinned + 16 * (size_t)n);     // pinned: [16n, ..) the counter block as it lies on the device
	int* h_counts = (int*)h_total + 2;
	long long total = 0;
	for (int attempt = 0; attempt < 2; ++attempt) {
		HIPCHK(W.oriented.ensure(sizeof(KeyPoint) * (size_t)capK));
		HIPCHK(pool_alloc((void**)&res.desc, sizeof(float) * 128 * (size_t)capK));
		HIPCHK(pool_alloc((void**)&res.coor, sizeof(double) * 2 * (size_t)capK));
		HIPCHK(pool_alloc((void**)&res.real, sizeof(double) * 2 * (size_t)capK));
		{ ProfScope ps(ctx, "orientation");
		  HIPCHK(launch_expand_oriented(plan, (const KeyPoint*)W.refinedB.p, (const int*)W.slot_of.p, d_refined_count, cap, (const float*)W.dirs.p,
					(const int*)W.ndirs.p, d_ocnt, d_total, d_oriented_count, (KeyPoint*)W.oriented.p, capK, st)); }
		{ ProfScope ps(ctx, "sift descriptor");
		  HIPCHK(launch_descriptor(plan, (const KeyPoint*)W.oriented.p, d_total, capK, res.desc, res.coor, res.real, st)); }
		HIPCHK(hipMemcpyAsync(h_total, W.counts.p, sizeof(int) * (3 * (size_t)n + 2), hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		total = *h_total;
		if (total <= capK) break;
		pool_free(res.desc); pool_free(res.coor); pool_free(res.real); res.desc = nullptr; res.coor = nullptr; res.real = nullptr;
		capK = total;                                   // exact size, second and last attempt
	}
	resolve_profile(ctx);
	W.last_total = total;
	std::vector<int> raw_count(h_counts, h_counts + n), refined_count(h_counts + n, h_counts + 2 * n);
	W.last_raw_max = *std::max_element(raw_count.begin(), raw_count.end());
	W.last_refined_max = *std::max_element(refined_count.begin(), refined_count.end());
	res.counts.assign(h_counts + 2 * n, h_counts + 3 * n);
	res.total = total;
	{
		int mx = 0;
		for (int i = 0; i < n; ++i) mx = std::max(mx, raw_count[i]);
		if (mx > cap) {
			if (mx > (1 << 26)) OP_FAIL(OP_ERR_CAPACITY, "raw extrema list overflow: " + std::to_string(mx));
			pool_free(res.desc); pool_free(res.coor); pool_free(res.real); res.desc = nullptr; res.coor = nullptr; res.real = nullptr;
			W.raw_cap = mx + mx / 4 + 64;          // counts are deterministic: the re-run cannot overflow again
			return run_group(ctx, cfg, imgs, W, plan, res, keep);
		}
	}

	if (keep) {
		keep->cap = cap;
		keep->raw.resize((size_t)raw_count[0] * 4);
		if (raw_count[0]) HIPCHK(hipMemcpy(keep->raw.data(), W.raw.p, sizeof(int) * 4 * raw_count[0], hipMemcpyDeviceToHost));
		keep->refined.resize(refined_count[0]);
		if (refined_count[0]) HIPCHK(hipMemcpy(keep->refined.data(), W.refinedB.p, sizeof(KeyPoint) * refined_count[0], hipMemcpyDeviceToHost));
		keep->oriented.resize(total);
		if (total) HIPCHK(hipMemcpy(keep->oriented.data(), W.oriented.p, sizeof(KeyPoint) * total, hipMemcpyDeviceToHost));
		keep->desc.resize((size_t)total * 128);
		if (total) HIPCHK(hipMemcpy(keep->desc.data(), res.desc, sizeof(float) * 128 * total, hipMemcpyDeviceToHost));
		// sort the raw list into scan order (o, s, y, x)
		std::vector<std::array<int, 4>> q(raw_count[0]);
		for (int i = 0; i < raw_count[0]; ++i) q[i] = {keep->raw[4 * i + 2], keep->raw[4 * i + 3], keep->raw[4 * i + 1], keep->raw[4 * i]};
		std::sort(q.begin(), q.end());
		for (int i = 0; i < raw_count[0]; ++i) { keep->raw[4 * i] = q[i][3]; keep->raw[4 * i + 1] = q[i][2]; keep->raw[4 * i + 2] = q[i][0]; keep->raw[4 * i + 3] = q[i][1]; }
	}
	return OP_OK;
}

}	// namespace

extern "C" {

int op_sift_batch(op_ctx* ctx, const op_config* cfg, const op_image* imgs, int n, op_features** out) {
	if (!ctx || !cfg || !imgs || n <= 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_sift_batch: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	for (int i = 0; i < n; ++i)
		if (!imgs[i].data || imgs[i].h < 2 || imgs[i].w < 2 || (imgs[i].dtype != OP_F32 && imgs[i].dtype != OP_U8))
			OP_FAIL(OP_ERR_INVALID, "op_sift_batch: bad image " + std::to_string(i));
	// group by (size, element type), keep first-appearance order
	std::vector<std::pair<std::pair<int, int>, std::vector<int>>> groups;
	for (int i = 0; i < n; ++i) {
		std::pair<int, int> key(imgs[i].h * 2 + imgs[i].dtype, imgs[i].w);
		bool found = false;
		for (auto& g : groups) if (g.first == key) { g.second.push_back(i); found = true; break; }
		if (!found) groups.push_back({key, {i}});
	}
	Workspace* W = ctx_workspace(ctx);
	op_features* f = new op_features;
	f->n = n; f->counts.assign(n, 0); f->offsets.assign(n + 1, 0); f->device = ctx->device;
	std::vector<GroupResult> results(groups.size());
	for (size_t g = 0; g < groups.size(); ++g) {
		std::vector<const op_image*> gi;
		for (int idx : groups[g].second) gi.push_back(&imgs[idx]);
		SiftPlan plan;
		int rc = run_group(ctx, *cfg, gi, *W, plan, results[g], nullptr);
		if (rc != OP_OK) {
			for (auto& r : results) { pool_free(r.desc); pool_free(r.coor); pool_free(r.real); }
			delete f; return rc;
		}
		for (size_t k = 0; k < gi.size(); ++k) f->counts[groups[g].second[k]] = results[g].counts[k];
	}
	int64_t total = 0;
	for (int i = 0; i < n; ++i) { f->offsets[i] = total; total += f->counts[i]; }
	f->offsets[n] = total;
	if (groups.size() == 1) {
		f->desc = results[0].desc; f->coor = results[0].coor; f->real = results[0].real;
	} else {
		HIPCHK(pool_alloc((void**)&f->desc, sizeof(float) * 128 * (size_t)std::max<int64_t>(total, 1)));
		HIPCHK(pool_alloc((void**)&f->coor, sizeof(double) * 2 * (size_t)std::max<int64_t>(total, 1)));
		HIPCHK(pool_alloc((void**)&f->real, sizeof(double) * 2 * (size_t)std::max<int64_t>(total, 1)));
		for (size_t g = 0; g < groups.size(); ++g) {
			long long go = 0;
			for (size_t k = 0; k < groups[g].second.size(); ++k) {
				int idx = groups[g].second[k]; int c = results[g].counts[k];
				if (c) {
					HIPCHK(hipMemcpyAsync(f->desc + f->offsets[idx] * 128, results[g].desc + go * 128, sizeof(float) * 128 * c, hipMemcpyDeviceToDevice, ctx->stream));
					HIPCHK(hipMemcpyAsync(f->coor + f->offsets[idx] * 2, results[g].coor + go * 2, sizeof(double) * 2 * c, hipMemcpyDeviceToDevice, ctx->stream));
					HIPCHK(hipMemcpyAsync(f->real + f->offsets[idx] * 2, results[g].real + go * 2, sizeof(double) * 2 * c, hipMemcpyDeviceToDevice, ctx->stream));
				}
				go += c;
			}
		}
		HIPCHK(hipStreamSynchronize(ctx->stream));
		for (auto& r : results) { pool_free(r.desc); pool_free(r.coor); pool_free(r.real); }
	}
	*out = f;
	return OP_OK;
}

// StitcherBase::calc_feature end to end for HOST images (stitch/stitcherbase.cc:9-27: Mat32f in host memory in,
// descriptors and keypoints in host memory out) as a three-stage pipeline over chunks of the batch: the uploads of
// all chunks are queued on a copy stream at once, the kernels of chunk k start when its upload has landed, and its
// results leave on a second copy stream while chunk k+1 computes -- PCIe in, kernels and PCIe out overlap instead of
// running back to back.  Images of one size and element type (the usual job); anything else takes op_sift_batch.
int op_sift_batch_host(op_ctx* ctx, const op_config* cfg, const op_image* imgs, int n,
		float* desc_out, double* coor_out, int64_t capacity_rows, op_features** out) {
	if (!ctx || !cfg || !imgs || n <= 0 || !out || capacity_rows < 0) OP_FAIL(OP_ERR_INVALID, "op_sift_batch_host: bad argument");
	bool uniform = true;
	for (int i = 0; i < n; ++i) {
		if (!imgs[i].data || imgs[i].h < 2 || imgs[i].w < 2 || (imgs[i].dtype != OP_F32 && imgs[i].dtype != OP_U8)) OP_FAIL(OP_ERR_INVALID, "op_sift_batch_host: bad image " + std::to_string(i));
		uniform = uniform && !imgs[i].on_device && imgs[i].h == imgs[0].h && imgs[i].w == imgs[0].w && imgs[i].dtype == imgs[0].dtype;
	}
	HIPCHK(hipSetDevice(ctx->device));
	const int nchunks = uniform ? std::max(1, std::min(6, n / 8)) : 1;
	if (nchunks == 1) {                      // nothing to overlap: the plain call, then one copy out
		int rc = op_sift_batch(ctx, cfg, imgs, n, out);
		if (rc != OP_OK) return rc;
		const int64_t total = (*out)->offsets[n];
		if ((desc_out || coor_out) && total > capacity_rows) OP_FAIL(OP_ERR_CAPACITY, "op_sift_batch_host: " + std::to_string(total) + " descriptors, room for " + std::to_string(capacity_rows));
		if (desc_out && total) HIPCHK(hipMemcpyAsync(desc_out, (*out)->desc, sizeof(float) * 128 * (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
		if (coor_out && total) HIPCHK(hipMemcpyAsync(coor_out, (*out)->coor, sizeof(double) * 2 * (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
		HIPCHK(hipStreamSynchronize(ctx->stream));
		return OP_OK;
	}
	HIPCHK(ctx->copy_streams());
	Workspace* W = ctx_workspace(ctx);
	const size_t img_bytes = (imgs[0].dtype == OP_U8 ? 1 : sizeof(float)) * (size_t)imgs[0].h * imgs[0].w * 3;
	const size_t img_stride = (img_bytes + 255) & ~(size_t)255;
	HIPCHK(W->staging.ensure(img_stride * (size_t)n));
	std::vector<int> cb(nchunks + 1, 0);
	for (int k = 0; k < nchunks; ++k) cb[k + 1] = cb[k] + n / nchunks + (k < n % nchunks ? 1 : 0);
	// constant host stride -> one (strided) copy per chunk
	ptrdiff_t hstride = n > 1 ? (const char*)imgs[1].data - (const char*)imgs[0].data : (ptrdiff_t)img_bytes;
	bool strided = hstride >= (ptrdiff_t)img_bytes;
	for (int i = 2; i < n && strided; ++i) strided = (const char*)imgs[i].data - (const char*)imgs[i - 1].data == hstride;
	std::vector<hipEvent_t> ev(nchunks, nullptr);
	std::vector<GroupResult> results(nchunks);
	std::unique_ptr<op_features, void (*)(op_features*)> f(new op_features, op_features_free);
	f->n = n; f->counts.assign(n, 0); f->offsets.assign(n + 1, 0); f->device = ctx->device;
	int rc = OP_OK;
	auto cleanup = [&] {
		hipStreamSynchronize(ctx->h2d_stream); hipStreamSynchronize(ctx->d2h_stream); hipStreamSynchronize(ctx->stream);
		for (auto& r : results) { pool_free(r.desc); pool_free(r.coor); pool_free(r.real); r.desc = nullptr; r.coor = nullptr; r.real = nullptr; }
		for (hipEvent_t e : ev) if (e) hipEventDestroy(e);
	};
#define PCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { op_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); cleanup(); return OP_ERR_HIP; } } while (0)
	for (int k = 0; k < nchunks; ++k) {
		char* dst = (char*)W->staging.p + (size_t)cb[k] * img_stride;
		const int m = cb[k + 1] - cb[k];
		if (strided) PCHK(hipMemcpy2DAsync(dst, img_stride, imgs[cb[k]].data, (size_t)hstride, img_bytes, (size_t)m, hipMemcpyHostToDevice, ctx->h2d_stream));
		else for (int i = 0; i < m; ++i) PCHK(hipMemcpyAsync(dst + (size_t)i * img_stride, imgs[cb[k] + i].data, img_bytes, hipMemcpyHostToDevice, ctx->h2d_stream));
		PCHK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
		PCHK(hipEventRecord(ev[k], ctx->h2d_stream));
	}
	int64_t rows = 0;
	bool overflow = false;
	for (int k = 0; k < nchunks; ++k) {
		const int m = cb[k + 1] - cb[k];
		std::vector<op_image> dev(m);
		std::vector<const op_image*> gi(m);
		for (int i = 0; i < m; ++i) {
			dev[i] = imgs[cb[k] + i]; dev[i].data = (char*)W->staging.p + (size_t)(cb[k] + i) * img_stride; dev[i].on_device = 1;
			gi[i] = &dev[i];
		}
		PCHK(hipStreamWaitEvent(ctx->stream, ev[k], 0));
		SiftPlan plan;
		rc = run_group(ctx, *cfg, gi, *W, plan, results[k], nullptr);
		if (rc != OP_OK) { cleanup(); return rc; }
		for (int i = 0; i < m; ++i) f->counts[cb[k] + i] = results[k].counts[i];
		const int64_t t = results[k].total;
		if ((desc_out || coor_out) && rows + t > capacity_rows) overflow = true;
		if (!overflow && t) {
			if (desc_out) PCHK(hipMemcpyAsync(desc_out + rows * 128, results[k].desc, sizeof(float) * 128 * (size_t)t, hipMemcpyDeviceToHost, ctx->d2h_stream));
			if (coor_out) PCHK(hipMemcpyAsync(coor_out + rows * 2, results[k].coor, sizeof(double) * 2 * (size_t)t, hipMemcpyDeviceToHost, ctx->d2h_stream));
		}
		rows += t;
	}
	for (int i = 0; i < n; ++i) f->offsets[i + 1] = f->offsets[i] + f->counts[i];
	{	// the resident table of the whole batch (the matcher's input): chunk results back to back
		const size_t cnt = (size_t)std::max<int64_t>(rows, 1);
		PCHK(pool_alloc((void**)&f->desc, sizeof(float) * 128 * cnt));
		PCHK(pool_alloc((void**)&f->coor, sizeof(double) * 2 * cnt));
		PCHK(pool_alloc((void**)&f->real, sizeof(double) * 2 * cnt));
		int64_t o = 0;
		for (int k = 0; k < nchunks; ++k) {
			const size_t t = (size_t)results[k].total;
			if (t) {
				PCHK(hipMemcpyAsync(f->desc + o * 128, results[k].desc, sizeof(float) * 128 * t, hipMemcpyDeviceToDevice, ctx->stream));
				PCHK(hipMemcpyAsync(f->coor + o * 2, results[k].coor, sizeof(double) * 2 * t, hipMemcpyDeviceToDevice, ctx->stream));
				PCHK(hipMemcpyAsync(f->real + o * 2, results[k].real, sizeof(double) * 2 * t, hipMemcpyDeviceToDevice, ctx->stream));
			}
			o += (int64_t)t;
		}
	}
#undef PCHK
	cleanup();
	*out = f.release();
	if (overflow) OP_FAIL(OP_ERR_CAPACITY, "op_sift_batch_host: " + std::to_string(rows) + " descriptors, room for " + std::to_string(capacity_rows) + " (the resident features are returned)");
	return OP_OK;
}

int op_debug_set_raw_capacity(op_ctx* ctx, int cap) {
	if (!ctx || cap < 64) OP_FAIL(OP_ERR_INVALID, "op_debug_set_raw_capacity: bad argument");
	ctx_workspace(ctx)->raw_cap = cap;
	return OP_OK;
}

int op_debug_set_desc_list_cap(op_ctx* ctx, int floats) {
	if (!ctx || floats < 0 || floats > OP_DESC_LIST_CAP) OP_FAIL(OP_ERR_INVALID, "op_debug_set_desc_list_cap: bad argument");
	ctx_workspace(ctx)->desc_list_cap = floats;
	return OP_OK;
}

int op_features_num_images(const op_features* f) { return f ? f->n : 0; }
int op_features_count(const op_features* f, int i) { return (f && i >= 0 && i < f->n) ? f->counts[i] : 0; }
int64_t op_features_offset(const op_features* f, int i) { return (f && i >= 0 && i <= f->n) ? f->offsets[i] : 0; }
int64_t op_features_total(const op_features* f) { return f ? f->offsets[f->n] : 0; }
const float* op_features_desc_device(const op_features* f) { return f ? f->desc : nullptr; }
const double* op_features_coor_device(const op_features* f) { return f ? f->coor : nullptr; }

int op_features_copy(op_ctx* ctx, const op_features* f, int i, float* desc, double* coor) {
	if (!ctx || !f || i < 0 || i >= f->n) OP_FAIL(OP_ERR_INVALID, "op_features_copy: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	const int c = f->counts[i];
	if (c == 0) return OP_OK;
	if (desc) HIPCHK(hipMemcpyAsync(desc, f->desc + f->offsets[i] * 128, sizeof(float) * 128 * c, hipMemcpyDeviceToHost, ctx->stream));
	if (coor) HIPCHK(hipMemcpyAsync(coor, f->coor + f->offsets[i] * 2, sizeof(double) * 2 * c, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return OP_OK;
}

int op_features_copy_real(op_ctx* ctx, const op_features* f, int i, double* real) {
	if (!ctx || !f || i < 0 || i >= f->n || !real) OP_FAIL(OP_ERR_INVALID, "op_features_copy_real: bad argument");
	if (!f->real) OP_FAIL(OP_ERR_INVALID, "op_features_copy_real: features were not produced by op_sift_batch");
	HIPCHK(hipSetDevice(ctx->device));
	const int c = f->counts[i];
	if (c == 0) return OP_OK;
	HIPCHK(hipMemcpyAsync(real, f->real + f->offsets[i] * 2, sizeof(double) * 2 * c, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return OP_OK;
}

int op_features_from_host(op_ctx* ctx, const float* const* desc, const double* const* coor, const int* counts, int n, op_features** out) {
	if (!ctx || (!desc && !coor) || !counts || n <= 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_features_from_host: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	op_features* f = new op_features;
	f->n = n; f->counts.assign(counts, counts + n); f->offsets.assign(n + 1, 0); f->device = ctx->device;
	int64_t total = 0;
	for (int i = 0; i < n; ++i) { if (counts[i] < 0) { delete f; OP_FAIL(OP_ERR_INVALID, "negative count"); } f->offsets[i] = total; total += counts[i]; }
	f->offsets[n] = total;
	f->has_desc = desc != nullptr;       // coordinates only: enough for op_ransac_pairs, rejected by op_match_pairs
	HIPCHK(pool_alloc((void**)&f->desc, f->has_desc ? sizeof(float) * 128 * (size_t)std::max<int64_t>(total, 1) : sizeof(float)));
	HIPCHK(pool_alloc((void**)&f->coor, sizeof(double) * 2 * (size_t)std::max<int64_t>(total, 1)));
	HIPCHK(hipMemsetAsync(f->coor, 0, sizeof(double) * 2 * (size_t)std::max<int64_t>(total, 1), ctx->stream));
	for (int i = 0; i < n; ++i) {
		if (!counts[i]) continue;
		if (f->has_desc) HIPCHK(hipMemcpyAsync(f->desc + f->offsets[i] * 128, desc[i], sizeof(float) * 128 * counts[i], hipMemcpyHostToDevice, ctx->stream));
		if (coor && coor[i]) HIPCHK(hipMemcpyAsync(f->coor + f->offsets[i] * 2, coor[i], sizeof(double) * 2 * counts[i], hipMemcpyHostToDevice, ctx->stream));
	}
	HIPCHK(hipStreamSynchronize(ctx->stream));
	*out = f;
	return OP_OK;
}

int op_features_adopt_device(op_ctx* ctx, const float* desc_dev, const double* coor_dev, const int* counts, int n, op_features** out) {
	if (!ctx || !desc_dev || !coor_dev || !counts || n <= 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_features_adopt_device: bad argument");
	op_features* f = new op_features;
	f->n = n; f->counts.assign(counts, counts + n); f->offsets.assign(n + 1, 0); f->device = ctx->device; f->owns = false;
	int64_t total = 0;
	for (int i = 0; i < n; ++i) { if (counts[i] < 0) { delete f; OP_FAIL(OP_ERR_INVALID, "negative count"); } f->offsets[i] = total; total += counts[i]; }
	f->offsets[n] = total;
	f->desc = const_cast<float*>(desc_dev); f->coor = const_cast<double*>(coor_dev);
	*out = f;
	return OP_OK;
}

int op_features_from_device(op_ctx* ctx, const float* desc_dev, const double* coor_dev, const int* counts, int n, op_features** out) {
	if (!ctx || !desc_dev || !counts || n <= 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_features_from_device: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	op_features* f = new op_features;
	f->n = n; f->counts.assign(counts, counts + n); f->offsets.assign(n + 1, 0); f->device = ctx->device;
	int64_t total = 0;
	for (int i = 0; i < n; ++i) { if (counts[i] < 0) { delete f; OP_FAIL(OP_ERR_INVALID, "negative count"); } f->offsets[i] = total; total += counts[i]; }
	f->offsets[n] = total;
	const size_t cnt = (size_t)std::max<int64_t>(total, 1);
	HIPCHK(pool_alloc((void**)&f->desc, sizeof(float) * 128 * cnt));
	HIPCHK(pool_alloc((void**)&f->coor, sizeof(double) * 2 * cnt));
	if (total) HIPCHK(hipMemcpyAsync(f->desc, desc_dev, sizeof(float) * 128 * total, hipMemcpyDeviceToDevice, ctx->stream));
	if (coor_dev && total) HIPCHK(hipMemcpyAsync(f->coor, coor_dev, sizeof(double) * 2 * total, hipMemcpyDeviceToDevice, ctx->stream));
	else HIPCHK(hipMemsetAsync(f->coor, 0, sizeof(double) * 2 * cnt, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	*out = f;
	return OP_OK;
}

}	// extern "C"

extern "C" void op_features_free(op_features* f);
// device-to-device copy between two contexts' devices, ordered on dst's stream (xGMI peer copy when the
// devices differ; multi.hip enables direct peer access where the hardware allows it)
static hipError_t copy_between(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t st) {
	if (!bytes) return hipSuccess;
	if (dst_dev == src_dev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
	return hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st);
}

// multi.hip, the all-gather of SURVEY 8(e).2 inside one process: part k holds the images [start[k], start[k+1]) of the
// job (contiguous blocks); EVERY device of the group gets the whole image-indexed table, each pulling the other
// devices' slices over its own xGMI links (one host thread per destination, all pairs of devices busy at once; no
// device relays another one's data).  tables[k] lands on ctxs[k]'s device.
int op_features_allgather_blocks(op_ctx* const* ctxs, int nctx, op_features* const* parts, const int* start, int n, op_features** tables) {
	std::vector<int> counts(n);
	std::vector<int64_t> offsets(n + 1, 0);
	for (int k = 0; k < nctx; ++k)
		for (int i = start[k]; i < start[k + 1]; ++i) counts[i] = parts[k]->counts[i - start[k]];
	for (int i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + counts[i];
	const int64_t total = offsets[n];
	const size_t cnt = (size_t)std::max<int64_t>(total, 1);
	std::vector<int> rcs(nctx, OP_OK);
	std::vector<std::string> errs(nctx);
	std::vector<std::thread> th;
	for (int k = 0; k < nctx; ++k) tables[k] = nullptr;
	for (int k = 0; k < nctx; ++k)
		th.emplace_back([&, k] {
			auto fail = [&](hipError_t e, const char* what) { errs[k] = std::string(what) + ": " + hipGetErrorString(e); rcs[k] = OP_ERR_HIP; };
			op_ctx* dst = ctxs[k];
			hipError_t e = hipSetDevice(dst->device);
			if (e != hipSuccess) return fail(e, "hipSetDevice");
			std::unique_ptr<op_features, void (*)(op_features*)> f(new op_features, op_features_free);
			f->n = n; f->counts = counts; f->offsets = offsets; f->device = dst->device;
			if ((e = pool_alloc((void**)&f->desc, sizeof(float) * 128 * cnt)) != hipSuccess) return fail(e, "pool_alloc");
			if ((e = pool_alloc((void**)&f->coor, sizeof(double) * 2 * cnt)) != hipSuccess) return fail(e, "pool_alloc");
			if ((e = pool_alloc((void**)&f->real, sizeof(double) * 2 * cnt)) != hipSuccess) return fail(e, "pool_alloc");
			for (int s = 0; s < nctx; ++s) {
				const int src = (k + s) % nctx;                     // start with the own slice, then rotate: no hot source
				const op_features* p = parts[src];
				const size_t c = (size_t)(offsets[start[src + 1]] - offsets[start[src]]);
				const int64_t o = offsets[start[src]];
				if ((e = copy_between(f->desc + o * 128, dst->device, p->desc, p->device, sizeof(float) * 128 * c, dst->stream)) != hipSuccess) return fail(e, "peer copy");
				if ((e = copy_between(f->coor + o * 2, dst->device, p->coor, p->device, sizeof(double) * 2 * c, dst->stream)) != hipSuccess) return fail(e, "peer copy");
				if ((e = copy_between(f->real + o * 2, dst->device, p->real, p->device, sizeof(double) * 2 * c, dst->stream)) != hipSuccess) return fail(e, "peer copy");
			}
			if ((e = hipStreamSynchronize(dst->stream)) != hipSuccess) return fail(e, "hipStreamSynchronize");
			tables[k] = f.release();
		});
	for (auto& t : th) t.join();
	int rc = OP_OK;
	for (int k = 0; k < nctx; ++k) if (rcs[k] != OP_OK && rc == OP_OK) { rc = rcs[k]; op_set_error("device " + std::to_string(k) + ": " + errs[k]); }
	if (rc != OP_OK) for (int k = 0; k < nctx; ++k) { op_features_free(tables[k]); tables[k] = nullptr; }
	return rc;
}

// multi.hip: the whole table of f on dst's device (one peer copy per array)
int op_features_replicate(op_ctx* dst, const op_features* src, op_features** out) {
	HIPCHK(hipSetDevice(dst->device));
	std::unique_ptr<op_features, void (*)(op_features*)> f(new op_features, op_features_free);
	f->n = src->n; f->counts = src->counts; f->offsets = src->offsets; f->device = dst->device; f->has_desc = src->has_desc;
	const int64_t total = src->offsets[src->n];
	const size_t cnt = (size_t)std::max<int64_t>(total, 1);
	HIPCHK(pool_alloc((void**)&f->desc, src->has_desc ? sizeof(float) * 128 * cnt : sizeof(float)));
	HIPCHK(pool_alloc((void**)&f->coor, sizeof(double) * 2 * cnt));
	if (src->has_desc) HIPCHK(copy_between(f->desc, dst->device, src->desc, src->device, sizeof(float) * 128 * (size_t)total, dst->stream));
	HIPCHK(copy_between(f->coor, dst->device, src->coor, src->device, sizeof(double) * 2 * (size_t)total, dst->stream));
	HIPCHK(hipStreamSynchronize(dst->stream));
	*out = f.release();
	return OP_OK;
}
std::vector<op_features*>& op_features_replicas(op_features* f) { return f->replicas; }
int op_features_device(const op_features* f) { return f->device; }

extern "C" {

void op_features_free(op_features* f) {
	if (!f) return;
	hipSetDevice(f->device);
	if (f->owns) { pool_free(f->desc); pool_free(f->coor); pool_free(f->real); }
	for (op_features* r : f->replicas) op_features_free(r);
	delete f;
}

// ---- staged dump ----
int op_sift_staged(op_ctx* ctx, const op_config* cfg, const op_image* img, op_sift_dump** out) {
	if (!ctx || !cfg || !img || !img->data || !out || (img->dtype != OP_F32 && img->dtype != OP_U8)) OP_FAIL(OP_ERR_INVALID, "op_sift_staged: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	op_sift_dump* d = new op_sift_dump;
	GroupResult res;
	std::vector<const op_image*> gi{img};
	int rc = run_group(ctx, *cfg, gi, d->w, d->plan, res, d);
	if (rc == OP_OK && res.total) {
		d->coor01.resize((size_t)res.total * 2);
		for (long long i = 0; i < res.total; ++i) { d->coor01[2 * i] = d->oriented[i].rx; d->coor01[2 * i + 1] = d->oriented[i].ry; }
	}
	pool_free(res.desc); pool_free(res.coor); pool_free(res.real);
	if (rc != OP_OK) { d->w.release(); delete d; return rc; }
	*out = d;
	return OP_OK;
}

void op_sift_dump_free(op_sift_dump* d) { if (!d) return; d->w.release(); delete d; }

int op_sift_dump_working_dims(const op_sift_dump* d, int* h, int* w) { *h = d->plan.wh; *w = d->plan.ww; return OP_OK; }
int op_sift_dump_octave_dims(const op_sift_dump* d, int oct, int* h, int* w) {
	if (oct < 0 || oct >= d->plan.noct) OP_FAIL(OP_ERR_INVALID, "bad octave");
	*h = d->plan.oct[oct].h; *w = d->plan.oct[oct].w; return OP_OK;
}

int op_sift_dump_plane(op_ctx* ctx, const op_sift_dump* d, int kind, int oct, int s, float* out) {
	if (!ctx || !d || !out) OP_FAIL(OP_ERR_INVALID, "op_sift_dump_plane: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	const SiftPlan& p = d->plan;
	if (kind == 4) {
		HIPCHK(hipMemcpy(out, p.work, sizeof(float) * (size_t)p.wh * p.ww * 3, hipMemcpyDeviceToHost));
		return OP_OK;
	}
	if (oct < 0 || oct >= p.noct) OP_FAIL(OP_ERR_INVALID, "bad octave");
	const OctDesc& o = p.oct[oct];
	long long off;
	if (kind == 5) off = plane_off_grey(o);
	else if (kind == 1 && s >= 0 && s <= p.nscale - 2) {
		// DoG planes are never materialised by the product path (internal.hpp); the dump evaluates
		// dog[s] = |G[s] - G[s+1]| (feature/dog.cc:126) on the two Gaussian planes, one fp32 subtraction per pixel
		std::vector<float> nxt((size_t)o.plane);
		HIPCHK(hipMemcpy(out, p.ws + plane_off_gauss(o, p.nscale, s), sizeof(float) * (size_t)o.plane, hipMemcpyDeviceToHost));
		HIPCHK(hipMemcpy(nxt.data(), p.ws + plane_off_gauss(o, p.nscale, s + 1), sizeof(float) * (size_t)o.plane, hipMemcpyDeviceToHost));
		for (long long i = 0; i < o.plane; ++i) out[i] = std::fabs(out[i] - nxt[i]);
		return OP_OK;
	}
	else if ((kind == 2 || kind == 3) && s >= 1 && s <= p.nscale - 3) {
		// mag / ort planes are never materialised by the product path; the dump computes them
		float* tmp = nullptr;
		HIPCHK(hipMalloc(&tmp, sizeof(float) * 2 * (size_t)o.plane));
		hipError_t e = launch_magort_plane(p, 0, oct, s, tmp, tmp + o.plane, ctx->stream);
		if (e == hipSuccess) e = hipMemcpyAsync(out, tmp + (kind == 3 ? o.plane : 0), sizeof(float) * (size_t)o.plane, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		hipFree(tmp);
		if (e != hipSuccess) OP_FAIL(OP_ERR_HIP, std::string("op_sift_dump_plane: ") + hipGetErrorString(e));
		return OP_OK;
	}
	else if (kind == 6 && s >= 1 && s <= p.nscale - 1) off = plane_off_gauss(o, p.nscale, s);
	else OP_FAIL(OP_ERR_INVALID, "bad plane kind/scale");
	HIPCHK(hipMemcpy(out, p.ws + off, sizeof(float) * (size_t)o.plane, hipMemcpyDeviceToHost));
	return OP_OK;
}

int op_sift_dump_raw_count(const op_sift_dump* d, int oct, int s) {
	int c = 0;
	for (size_t i = 0; i < d->raw.size() / 4; ++i) c += (d->raw[4 * i + 2] == oct && d->raw[4 * i + 3] == s);
	return c;
}
int op_sift_dump_raw(const op_sift_dump* d, int oct, int s, int* xy) {
	int c = 0;
	for (size_t i = 0; i < d->raw.size() / 4; ++i)
		if (d->raw[4 * i + 2] == oct && d->raw[4 * i + 3] == s) { xy[2 * c] = d->raw[4 * i]; xy[2 * c + 1] = d->raw[4 * i + 1]; ++c; }
	return c;
}
int op_sift_dump_kp_count(const op_sift_dump* d, int which) { return (int)(which ? d->oriented.size() : d->refined.size()); }
int op_sift_dump_kp(const op_sift_dump* d, int which, int* ints, double* real, float* fl) {
	const std::vector<KeyPoint>& v = which ? d->oriented : d->refined;
	for (size_t i = 0; i < v.size(); ++i) {
		ints[4 * i] = v[i].x; ints[4 * i + 1] = v[i].y; ints[4 * i + 2] = v[i].oct; ints[4 * i + 3] = v[i].scale;
		real[2 * i] = v[i].rx; real[2 * i + 1] = v[i].ry;
		fl[2 * i] = which ? v[i].dir : 0.f; fl[2 * i + 1] = v[i].sf;
	}
	return OP_OK;
}
int op_sift_dump_desc(const op_sift_dump* d, float* desc, double* coor) {
	if (desc && !d->desc.empty()) memcpy(desc, d->desc.data(), sizeof(float) * d->desc.size());
	if (coor && !d->coor01.empty()) memcpy(coor, d->coor01.data(), sizeof(double) * d->coor01.size());
	return OP_OK;
}

}	// extern "C"
