// match.hip -- all-pairs descriptor matching (exact 2-NN ratio test, both directions).
//
// Replaces PairWiseMatcher::match as called by Stitcher::pairwise_match (feature/matcher.cc:90-135,
// stitch/stitcher.cc:96-136) with the semantics of the reference's exact matcher
// FeatureMatcher::match (feature/matcher.cc:15-71) -- see SURVEY.md F2/F3 for why the
// kd-forest's approximate, non-deterministic answers cannot be the parity target.
//
// Two kernels for ALL requested image pairs at once:
//  k_match_top4   fp32 MFMA (v_mfma_f32_32x32x2_f32) dot-product tiles; the epilogue keeps, for
//                 every descriptor x of set X, the 4 best columns of set Y by
//                 score = x.y - |y|^2/2  (= const - d(x,y)/2).  Run for both directions of a pair.
//                 MFMA results only RANK candidates; they never decide a match.
//  k_match_decide per row of the smaller set: re-scores the candidates with the reference's
//                 exact squared-L2 (feature/dist.cc:22-57: four stride-4 fp32 partial sums,
//                 (v0+v1)+(v2+v3)), applies both ratio tests with the reference's float
//                 arithmetic.  Candidate sets are provably complete: every column whose score is
//                 within E of the 2nd best is re-scored, where E bounds twice the worst-case
//                 fp32 error of score vs. exact distance; if all 4 kept entries fall inside the
//                 margin the row falls back to an exact full scan.
#include "internal.hpp"
#include <cfloat>
#include <algorithm>

struct op_features;   // sift_host.hip
struct FeatView { int n; const int* counts; const int64_t* offsets; const float* desc; int device; };
FeatView op_features_view(const op_features* f);

struct op_matches {
	int npairs = 0;
	std::vector<std::vector<int>> pairs;   // per image pair: flat (first, second) sorted
	int64_t total = 0;
};

const std::vector<int>& op_matches_pair_vector(const op_matches* m, int p) { return m->pairs[p]; }

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WorkItem { int x_off, kx, y_off, ky, rowblock, out_off; };   // offsets in descriptors

constexpr int YP = 132;   // LDS pitch of a Y tile row (floats): 16-B slot rotation -> conflict-free b128

__global__ void __launch_bounds__(256) k_norms(const float* desc, long long total, float* norms, unsigned* gmax_bits) {
	const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
	if (i >= total) return;
	const f32x4* p = (const f32x4*)(desc + i * 128);
	float s = 0.f;
	for (int t = 0; t < 32; ++t) { f32x4 v = p[t]; s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
	norms[i] = s;
	atomicMax(gmax_bits, __float_as_uint(s));
}

__device__ __forceinline__ void top4_insert(float (&ts)[4], int (&ti)[4], float s, int idx) {
	if (!(s > ts[3])) return;
	if (s > ts[0]) { ts[3] = ts[2]; ti[3] = ti[2]; ts[2] = ts[1]; ti[2] = ti[1]; ts[1] = ts[0]; ti[1] = ti[0]; ts[0] = s; ti[0] = idx; }
	else if (s > ts[1]) { ts[3] = ts[2]; ti[3] = ti[2]; ts[2] = ts[1]; ti[2] = ti[1]; ts[1] = s; ti[1] = idx; }
	else if (s > ts[2]) { ts[3] = ts[2]; ti[3] = ti[2]; ts[2] = s; ti[2] = idx; }
	else { ts[3] = s; ti[3] = idx; }
}

// One workgroup = 128 rows of X (4 waves x 32 rows, X fragments resident in VGPRs) against all
// of Y, streamed through LDS 32 columns at a time.  D = Ytile * X^T so that every lane ends up
// holding 16 scores of ONE X row (C/D layout: col = lane&31), which makes the running top-4 a
// purely per-lane update; the two lane halves are merged once at the end.
__global__ void __launch_bounds__(256) k_match_top4(const float* __restrict__ desc, const float* __restrict__ norms,
		const WorkItem* __restrict__ work, float* __restrict__ top_s, int* __restrict__ top_i) {
	__shared__ __attribute__((aligned(16))) float s_y[2][32 * YP];
	__shared__ float s_nyh[2][32];
	__shared__ float s_ms[4][32][2][4];
	__shared__ int s_mi[4][32][2][4];
	const WorkItem wk = work[blockIdx.x];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const int j = lane & 31, h = lane >> 5;
	const float* X = desc + (long long)wk.x_off * 128;
	const float* Y = desc + (long long)wk.y_off * 128;
	const float* ny = norms + wk.y_off;
	const int row = wk.rowblock * 128 + wave * 32 + j;
	const int rowc = row < wk.kx ? row : wk.kx - 1;
	// X fragment: row j, k in [64h, 64h+64)
	f32x4 xf[16];
	{
		const f32x4* px = (const f32x4*)(X + (long long)rowc * 128 + 64 * h);
#pragma unroll
		for (int q = 0; q < 16; ++q) xf[q] = px[q];
	}
	float ts[4] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
	int ti[4] = {-1, -1, -1, -1};
	const int ntiles = (wk.ky + 31) / 32;

	auto load_tile = [&](int t, int buf) {
		// 32 rows x 128 floats = 1024 float4, 4 per thread
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int e = tid + 256 * r;          // float4 index
			const int yr = e >> 5, c4 = e & 31;
			const int gy = t * 32 + yr;
			f32x4 v = {0.f, 0.f, 0.f, 0.f};
			if (gy < wk.ky) v = *(const f32x4*)(Y + (long long)gy * 128 + c4 * 4);
			*(f32x4*)(&s_y[buf][yr * YP + c4 * 4]) = v;
		}
		if (tid < 32) {
			const int gy = t * 32 + tid;
			s_nyh[buf][tid] = gy < wk.ky ? 0.5f * ny[gy] : FLT_MAX;   // padded columns can never rank
		}
	};

	load_tile(0, 0);
	__syncthreads();
	for (int t = 0; t < ntiles; ++t) {
		const int buf = t & 1;
		if (t + 1 < ntiles) load_tile(t + 1, buf ^ 1);
		f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		const float* yrow = &s_y[buf][j * YP + 64 * h];
#pragma unroll
		for (int q = 0; q < 16; ++q) {
			const f32x4 a = *(const f32x4*)(yrow + 4 * q);
			acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xf[q].x, acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xf[q].y, acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xf[q].z, acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xf[q].w, acc, 0, 0, 0);
		}
		// lane holds D[i][j] for i = (reg&3) + 8*(reg>>2) + 4*h : 16 Y columns of X row j
#pragma unroll
		for (int reg = 0; reg < 16; ++reg) {
			const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
			const float s = acc[reg] - s_nyh[buf][i];
			top4_insert(ts, ti, s, t * 32 + i);
		}
		__syncthreads();
	}
	// merge the two lane halves of each X row
#pragma unroll
	for (int r = 0; r < 4; ++r) { s_ms[wave][j][h][r] = ts[r]; s_mi[wave][j][h][r] = ti[r]; }
	__syncthreads();
	if (h == 0 && row < wk.kx) {
#pragma unroll
		for (int r = 0; r < 4; ++r) top4_insert(ts, ti, s_ms[wave][j][1][r], s_mi[wave][j][1][r]);
		float* os = top_s + ((long long)wk.out_off + row) * 4;
		int* oi = top_i + ((long long)wk.out_off + row) * 4;
#pragma unroll
		for (int r = 0; r < 4; ++r) { os[r] = ts[r]; oi[r] = ti[r]; }
	}
}

// feature/dist.cc:22-57 without the early-out (which never changes a result: partial sums are
// monotone, so a distance it cuts off could not have lowered the running minima)
__device__ __forceinline__ float euclidean_sqr_exact(const float* __restrict__ x, const float* __restrict__ y) {
	float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
	const f32x4* px = (const f32x4*)x; const f32x4* py = (const f32x4*)y;
#pragma unroll 8
	for (int t = 0; t < 32; ++t) {
		const f32x4 a = px[t], b = py[t];
		float d;
		d = a.x - b.x; v0 += d * d;
		d = a.y - b.y; v1 += d * d;
		d = a.z - b.z; v2 += d * d;
		d = a.w - b.w; v3 += d * d;
	}
	return (v0 + v1) + (v2 + v3);
}

struct PairDesc { int a_off, ka, b_off, kb; int topA_off, topB_off; int res_off; int rev; };

// thread per row of the smaller set (FeatureMatcher::match body, feature/matcher.cc:33-67)
__global__ void __launch_bounds__(128) k_match_decide(const float* __restrict__ desc, const float* __restrict__ norms,
		const unsigned* __restrict__ gmax_bits, const PairDesc* __restrict__ pairs, const int2* __restrict__ blocks,
		const float* __restrict__ top_s, const int* __restrict__ top_i, float reject_ratio_sqr, int* __restrict__ res) {
	const int2 blk = blocks[blockIdx.x];
	const PairDesc pd = pairs[blk.x];
	const int a = blk.y * 128 + threadIdx.x;
	if (a >= pd.ka) return;
	const float* A = desc + (long long)pd.a_off * 128;
	const float* B = desc + (long long)pd.b_off * 128;
	const float* xa = A + (long long)a * 128;
	const float gmax = __uint_as_float(*gmax_bits);
	// ---- forward: exact top-2 of row a over B
	float mn = FLT_MAX, next_min = FLT_MAX; int min_idx = -1;
	{
		const float* s = top_s + ((long long)pd.topA_off + a) * 4;
		const int* ix = top_i + ((long long)pd.topA_off + a) * 4;
		const float E = 3.2e-5f * (norms[pd.a_off + a] + gmax);
		const float thr = s[1] - E;
		const bool overflow = ix[3] >= 0 && s[3] >= thr;
		if (overflow || pd.kb <= 4) {
			for (int kk = 0; kk < pd.kb; ++kk) {
				const float d = euclidean_sqr_exact(xa, B + (long long)kk * 128);
				if (d < mn) { next_min = mn; mn = d; min_idx = kk; }
				else if (d < next_min) next_min = d;
			}
		} else {
			// candidates in ascending column order so that ties resolve to the first index (:42-48)
			int c[4]; int nc = 0;
#pragma unroll
			for (int r = 0; r < 4; ++r) if (ix[r] >= 0 && s[r] >= thr) c[nc++] = ix[r];
			for (int u = 1; u < nc; ++u) { int v = c[u], w = u; while (w > 0 && c[w - 1] > v) { c[w] = c[w - 1]; --w; } c[w] = v; }
			for (int u = 0; u < nc; ++u) {
				const float d = euclidean_sqr_exact(xa, B + (long long)c[u] * 128);
				if (d < mn) { next_min = mn; mn = d; min_idx = c[u]; }
				else if (d < next_min) next_min = d;
			}
		}
	}
	int out = -1;
	if (min_idx >= 0 && !(mn > reject_ratio_sqr * next_min)) {             // :52
		// ---- reverse: min over a' != a of d(b*, a'), folded into next_min (:57-61)
		const float* xb = B + (long long)min_idx * 128;
		const float* s = top_s + ((long long)pd.topB_off + min_idx) * 4;
		const int* ix = top_i + ((long long)pd.topB_off + min_idx) * 4;
		const float E = 3.2e-5f * (norms[pd.b_off + min_idx] + gmax);
		const float thr = s[1] - E;
		const bool overflow = ix[3] >= 0 && s[3] >= thr;
		if (overflow || pd.ka <= 4) {
			for (int kk = 0; kk < pd.ka; ++kk) if (kk != a) {
				const float d = euclidean_sqr_exact(xb, A + (long long)kk * 128);
				if (d < next_min) next_min = d;
			}
		} else {
#pragma unroll
			for (int r = 0; r < 4; ++r) if (ix[r] >= 0 && ix[r] != a && s[r] >= thr) {
				const float d = euclidean_sqr_exact(xb, A + (long long)ix[r] * 128);
				if (d < next_min) next_min = d;
			}
		}
		if (!(mn > reject_ratio_sqr * next_min)) out = min_idx;             // :62
	}
	res[pd.res_off + a] = out;
}

}	// namespace

extern "C" {

int op_match_pairs(op_ctx* ctx, const op_config* cfg, const op_features* f, const int* pairs, int npairs, op_matches** out) {
	if (!ctx || !cfg || !f || !pairs || npairs < 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_match_pairs: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	const FeatView fv = op_features_view(f);
	hipStream_t st = ctx->stream;
	const long long total = fv.offsets[fv.n];
	if (!fv.desc) OP_FAIL(OP_ERR_INVALID, "op_match_pairs: features hold coordinates only (built without descriptors)");
	op_matches* m = new op_matches;
	m->npairs = npairs; m->pairs.resize(npairs);
	if (npairs == 0 || total == 0) { *out = m; return OP_OK; }

	std::vector<WorkItem> work;
	std::vector<PairDesc> pds(npairs);
	std::vector<int2> blocks;
	long long top_rows = 0, res_rows = 0;
	for (int p = 0; p < npairs; ++p) {
		const int i = pairs[2 * p], j = pairs[2 * p + 1];
		if (i < 0 || j < 0 || i >= fv.n || j >= fv.n) { delete m; OP_FAIL(OP_ERR_INVALID, "op_match_pairs: image index out of range"); }
		const int rev = fv.counts[i] > fv.counts[j];                       // matcher.cc:21
		const int ia = rev ? j : i, ib = rev ? i : j;
		PairDesc& pd = pds[p];
		pd.a_off = (int)fv.offsets[ia]; pd.ka = fv.counts[ia];
		pd.b_off = (int)fv.offsets[ib]; pd.kb = fv.counts[ib];
		pd.rev = rev;
		pd.topA_off = (int)top_rows; top_rows += pd.ka;
		pd.topB_off = (int)top_rows; top_rows += pd.kb;
		pd.res_off = (int)res_rows; res_rows += pd.ka;
		if (pd.ka > 0 && pd.kb > 0) {
			for (int rb = 0; rb * 128 < pd.ka; ++rb) work.push_back({pd.a_off, pd.ka, pd.b_off, pd.kb, rb, pd.topA_off});
			for (int rb = 0; rb * 128 < pd.kb; ++rb) work.push_back({pd.b_off, pd.kb, pd.a_off, pd.ka, rb, pd.topB_off});
			for (int rb = 0; rb * 128 < pd.ka; ++rb) blocks.push_back(make_int2(p, rb));
		}
	}
	if (top_rows >= (1LL << 29)) { delete m; OP_FAIL(OP_ERR_CAPACITY, "op_match_pairs: too many rows in one call; split the pair list"); }

	float *d_norms = nullptr, *d_top_s = nullptr; int *d_top_i = nullptr, *d_res = nullptr; unsigned* d_gmax = nullptr;
	WorkItem* d_work = nullptr; PairDesc* d_pds = nullptr; int2* d_blocks = nullptr;
	std::vector<int> h_res(std::max<long long>(res_rows, 1), -1);
	int rc = OP_OK;
#define MCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { op_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); rc = OP_ERR_HIP; goto done; } } while (0)
	MCHK(pool_alloc((void**)&d_norms, sizeof(float) * total));
	MCHK(pool_alloc((void**)&d_gmax, sizeof(unsigned)));
	MCHK(pool_alloc((void**)&d_top_s, sizeof(float) * 4 * std::max<long long>(top_rows, 1)));
	MCHK(pool_alloc((void**)&d_top_i, sizeof(int) * 4 * std::max<long long>(top_rows, 1)));
	MCHK(pool_alloc((void**)&d_res, sizeof(int) * std::max<long long>(res_rows, 1)));
	MCHK(pool_alloc((void**)&d_work, sizeof(WorkItem) * std::max<size_t>(work.size(), 1)));
	MCHK(pool_alloc((void**)&d_pds, sizeof(PairDesc) * npairs));
	MCHK(pool_alloc((void**)&d_blocks, sizeof(int2) * std::max<size_t>(blocks.size(), 1)));
	MCHK(hipMemsetAsync(d_gmax, 0, sizeof(unsigned), st));
	MCHK(hipMemsetAsync(d_res, 0xff, sizeof(int) * std::max<long long>(res_rows, 1), st));
	if (!work.empty()) MCHK(hipMemcpyAsync(d_work, work.data(), sizeof(WorkItem) * work.size(), hipMemcpyHostToDevice, st));
	MCHK(hipMemcpyAsync(d_pds, pds.data(), sizeof(PairDesc) * npairs, hipMemcpyHostToDevice, st));
	if (!blocks.empty()) MCHK(hipMemcpyAsync(d_blocks, blocks.data(), sizeof(int2) * blocks.size(), hipMemcpyHostToDevice, st));
	{
		ProfScope ps(ctx, "matcher norms");
		hipLaunchKernelGGL(k_norms, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, fv.desc, total, d_norms, d_gmax);
		MCHK(hipGetLastError());
	}
	if (!work.empty()) {
		ProfScope ps(ctx, "matcher mfma top4");
		hipLaunchKernelGGL(k_match_top4, dim3((unsigned)work.size()), dim3(256), 0, st, fv.desc, d_norms, d_work, d_top_s, d_top_i);
		MCHK(hipGetLastError());
	}
	if (!blocks.empty()) {
		ProfScope ps(ctx, "matcher decide");
		const float rr = cfg->MATCH_REJECT_NEXT_RATIO * cfg->MATCH_REJECT_NEXT_RATIO;   // matcher.cc:16
		hipLaunchKernelGGL(k_match_decide, dim3((unsigned)blocks.size()), dim3(128), 0, st, fv.desc, d_norms, d_gmax, d_pds, d_blocks,
				d_top_s, d_top_i, rr, d_res);
		MCHK(hipGetLastError());
	}
	MCHK(hipMemcpyAsync(h_res.data(), d_res, sizeof(int) * std::max<long long>(res_rows, 1), hipMemcpyDeviceToHost, st));
	MCHK(hipStreamSynchronize(st));
	resolve_profile(ctx);
	for (int p = 0; p < npairs; ++p) {
		const PairDesc& pd = pds[p];
		std::vector<std::pair<int, int>> v;
		for (int a = 0; a < pd.ka; ++a) {
			const int b = h_res[pd.res_off + a];
			if (b >= 0) v.push_back(pd.rev ? std::make_pair(b, a) : std::make_pair(a, b));   // matcher.cc:68-69
		}
		std::sort(v.begin(), v.end());
		m->pairs[p].reserve(v.size() * 2);
		for (auto& q : v) { m->pairs[p].push_back(q.first); m->pairs[p].push_back(q.second); }
		m->total += (int64_t)v.size();
	}
done:
	pool_free(d_norms); pool_free(d_gmax); pool_free(d_top_s); pool_free(d_top_i);
	pool_free(d_res); pool_free(d_work); pool_free(d_pds); pool_free(d_blocks);
#undef MCHK
	if (rc != OP_OK) { delete m; return rc; }
	*out = m;
	return OP_OK;
}

int op_matches_from_host(const int* const* idx_pairs, const int* counts, int npairs, op_matches** out) {
	if (!idx_pairs || !counts || npairs < 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_matches_from_host: bad argument");
	op_matches* m = new op_matches;
	m->npairs = npairs; m->pairs.resize(npairs);
	for (int p = 0; p < npairs; ++p) {
		if (counts[p] < 0) { delete m; OP_FAIL(OP_ERR_INVALID, "negative count"); }
		m->pairs[p].assign(idx_pairs[p], idx_pairs[p] + 2 * (size_t)counts[p]);
		m->total += counts[p];
	}
	*out = m;
	return OP_OK;
}

int op_matches_count(const op_matches* m, int p) { return (m && p >= 0 && p < m->npairs) ? (int)(m->pairs[p].size() / 2) : 0; }
int op_matches_copy(const op_matches* m, int p, int* idx_pairs) {
	if (!m || p < 0 || p >= m->npairs || !idx_pairs) OP_FAIL(OP_ERR_INVALID, "op_matches_copy: bad argument");
	std::copy(m->pairs[p].begin(), m->pairs[p].end(), idx_pairs);
	return OP_OK;
}
int64_t op_matches_total(const op_matches* m) { return m ? m->total : 0; }
void op_matches_free(op_matches* m) { delete m; }

}	// extern "C"
