// match.hip -- all-pairs descriptor matching (exact 2-NN ratio test, both directions).
//
// Replaces PairWiseMatcher::match as called by Stitcher::pairwise_match (feature/matcher.cc:90-135,
// stitch/stitcher.cc:96-136) with the semantics of the reference's exact matcher
// FeatureMatcher::match (feature/matcher.cc:15-71) -- see SURVEY.md F2/F3 for why the
// kd-forest's approximate, non-deterministic answers cannot be the parity target.
//
// ALL requested image pairs go through two launches of one MFMA sweep kernel (dot-product tiles
// from three v_mfma_f32_32x32x16_bf16 per 16 elements on a two-term bf16 split of the descriptors
// -- x = hi + lo, x.y ~ hi.hi + hi.lo + lo.hi, fp32 accumulation: 2^-16-accurate at 5x the rate of
// the fp32 MFMA -- and a running per-row top-4 by score = x.y - |y|^2/2):
// a forward sweep of every row of the smaller set, whose epilogue re-scores the ranked
// candidates with the reference's exact squared L2 and applies the first ratio test, and a
// reverse sweep over the survivors only, whose epilogue applies the second test.  MFMA results
// only RANK candidates; the reference's float arithmetic decides every match.
#include "internal.hpp"
#include <cfloat>
#include <cstring>
#include <memory>
#include <algorithm>
#include <mutex>

struct op_features;   // sift_host.hip
struct FeatView { int n; const int* counts; const int64_t* offsets; const float* desc; int device; };
FeatView op_features_view(const op_features* f);

// The result of a match call stays in HBM: per image pair a (first, second) list sorted by (first, second),
// pairs back to back in job order.  op_ransac_pairs reads it there; the host only learns the per-pair counts
// with the call and fetches the index lists the first time somebody asks for them (op_matches_copy).
struct op_matches {
	int npairs = 0;
	std::vector<int> count;            // matches of pair p
	std::vector<int64_t> offset;       // npairs + 1: first entry of pair p in the flat list
	std::vector<int> lim;              // npairs x 2: keypoint counts of the two images the indices refer to (empty: unknown, lists came from the host)
	int64_t total = 0;
	int* d_idx = nullptr;              // device, total x (first, second); null when the lists only exist on the host
	int device = -1; hipStream_t stream = nullptr;       // where d_idx was produced (its producer kernels are ordered on this stream)
	hipEvent_t produced = nullptr;     // recorded behind the kernel that fills d_idx: the block may only go back to the pool (which is not stream-aware) once it has fired
	mutable std::vector<int> h_idx;    // host mirror of d_idx, fetched on first use
	mutable bool host_valid = false;
	mutable std::mutex mu;
	// multi.hip: a job matched on several devices keeps its per-device results (parts[k] holds the pairs
	// part_index[k][0..] of the job's pair list, resident on device k) so that RANSAC follows the same deal without
	// a round trip through the host; the flat host list is assembled from the parts on first use
	std::vector<op_matches*> parts;
	std::vector<std::vector<int>> part_index;
	~op_matches() {
		if (d_idx || produced) hipSetDevice(device);
		if (produced) { hipEventSynchronize(produced); hipEventDestroy(produced); }      // the per-pair sort may still be writing d_idx when the caller frees the result
		if (d_idx) pool_free(d_idx);
		for (op_matches* p : parts) delete p;
	}
};

int op_matches_num_pairs(const op_matches* m) { return m->npairs; }
const std::vector<int>& op_matches_counts(const op_matches* m) { return m->count; }
const std::vector<int64_t>& op_matches_offsets(const op_matches* m) { return m->offset; }
const std::vector<int>& op_matches_limits(const op_matches* m) { return m->lim; }
const std::vector<op_matches*>& op_matches_parts(const op_matches* m) { return m->parts; }
const std::vector<std::vector<int>>& op_matches_part_index(const op_matches* m) { return m->part_index; }
// flat host list (total x 2), fetched from the device once; nullptr + error set on failure
const int* op_matches_host(const op_matches* m) {
	std::lock_guard<std::mutex> lk(m->mu);
	if (m->host_valid) return m->h_idx.data();
	m->h_idx.resize((size_t)std::max<int64_t>(m->total, 1) * 2);
	if (!m->parts.empty()) {
		for (size_t k = 0; k < m->parts.size(); ++k) {
			const int* src = op_matches_host(m->parts[k]);
			if (!src) return nullptr;
			for (size_t q = 0; q < m->part_index[k].size(); ++q) {
				const int c = m->parts[k]->count[q];
				if (c) std::memcpy(m->h_idx.data() + 2 * m->offset[m->part_index[k][q]], src + 2 * m->parts[k]->offset[q], sizeof(int) * 2 * (size_t)c);
			}
		}
	} else if (m->total && m->d_idx) {
		hipError_t e = hipSetDevice(m->device);
		if (e == hipSuccess) e = hipMemcpyAsync(m->h_idx.data(), m->d_idx, sizeof(int) * 2 * (size_t)m->total, hipMemcpyDeviceToHost, m->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
		if (e != hipSuccess) { op_set_error(std::string("op_matches: result copy failed: ") + hipGetErrorString(e)); return nullptr; }
	}
	m->host_valid = true;
	return m->h_idx.data();
}
// device list on `device` if the matches live there (else nullptr: the caller uploads the host list).  The list's
// producer (the per-pair sort) is only ORDERED on the stream it was made on: a consumer on another stream waits for it.
const int* op_matches_device(const op_matches* m, int device, hipStream_t consumer) {
	if (!m->d_idx || m->device != device) return nullptr;
	if (m->stream != consumer && hipStreamSynchronize(m->stream) != hipSuccess) return nullptr;
	return m->d_idx;
}
// multi.hip: parts[k] holds the pairs index[k][0..] of the job's pair list -> one op_matches in job order that OWNS
// the parts (they stay on their devices); only counts are merged here
op_matches* op_matches_merge(op_matches* const* parts, const std::vector<std::vector<int>>& index, int npairs) {
	op_matches* m = new op_matches;
	m->npairs = npairs; m->count.assign(npairs, 0); m->offset.assign(npairs + 1, 0); m->lim.assign((size_t)npairs * 2, 0);
	for (size_t k = 0; k < index.size(); ++k)
		for (size_t q = 0; q < index[k].size(); ++q) {
			m->count[index[k][q]] = parts[k]->count[q];
			if (!parts[k]->lim.empty()) { m->lim[2 * (size_t)index[k][q]] = parts[k]->lim[2 * q]; m->lim[2 * (size_t)index[k][q] + 1] = parts[k]->lim[2 * q + 1]; }
		}
	for (int p = 0; p < npairs; ++p) m->offset[p + 1] = m->offset[p] + m->count[p];
	m->total = m->offset[npairs];
	m->parts.assign(parts, parts + index.size());
	m->part_index = index;
	return m;
}

#define STAMP(k) do { } while (0)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// One MFMA sweep kernel serves both directions of FeatureMatcher::match:
//   FWD  rows = every descriptor of the smaller set A, columns = B: exact 2-NN + the first ratio
//        test (feature/matcher.cc:33-55); rows that pass are appended to the pair's survivor list;
//   REV  rows = the best matches b* of the survivors only (typically a few percent of A),
//        columns = A: min over a' != a of d(b*, a') folded into next_min and the second ratio
//        test (matcher.cc:57-63).
// so the dense contraction runs once per unordered pair plus a small reverse strip, i.e. the
// algorithmic 2*128*K_i*K_j flop of SURVEY 8(d) instead of twice that.
struct WorkItem { int pair, rowblock; };
struct PairDesc { int a_off, ka, b_off, kb; int res_off; int rev; int ia, ib; };   // ia / ib: image indices of the A / B set (the exact-scan queue is grouped by the image it scans)

// LDS image of a Y tile: 32 split rows of 512 bytes, LINEAR (the tile arrives by LDS-DMA, whose destination is
// wave base + lane x 16 bytes), with the 16-byte blocks of row r stored at position (block ^ (r & 15)): the lanes of one
// ds_read_b128 group read the same logical block of 16 different rows, i.e. 16 different positions -> conflict-free.
constexpr int YROW = 128;  // floats per tile row in LDS

// Two-term bf16 split of every descriptor, row r -> [128 x hi][128 x lo] (512 B): hi = bf16(v)
// (round to nearest even), lo = bf16(v - hi); v - hi is exact in fp32, so |v - hi - lo| <= 2^-18 |v|.
// NaN stays NaN in both terms (such a row / column must simply never rank).
__device__ __forceinline__ unsigned bf16_rn(float v) {
	const unsigned u = __float_as_uint(v);
	return v != v ? 0x7fc0u : (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// The same pass writes |row|^2 (16 threads per row, shuffle tree) and the maximum norm of the call
// (grid-stride loop, one atomic per workgroup: tens of thousands of atomics on one address cost more
// than the whole split).
__global__ void __launch_bounds__(256) k_split_bf16(const float* __restrict__ desc, long long total, uint4* __restrict__ split,
		float* __restrict__ norms, unsigned* __restrict__ gmax_bits) {
	__shared__ unsigned s_max[4];
	unsigned mymax = 0u;                                       // norms are >= 0: their bit patterns order like the values
	const long long nthreads = total * 16, stride = (long long)gridDim.x * 256;
	for (long long i0 = (long long)blockIdx.x * 256; i0 < nthreads; i0 += stride) {      // one thread = 8 consecutive elements
		long long i = i0 + threadIdx.x;
		const bool live = i < nthreads;
		if (!live) i = nthreads - 1;
		const long long row = i >> 4; const int c = (int)(i & 15);
		const f32x4* p = (const f32x4*)(desc + row * 128 + c * 8);
		const f32x4 a = p[0], b = p[1];
		const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
		float sq = 0.f;
#pragma unroll
		for (int e = 0; e < 8; ++e) sq += v[e] * v[e];
#pragma unroll
		for (int d = 1; d < 16; d <<= 1) sq += __shfl_xor(sq, d);
		if (live && c == 0) norms[row] = sq;
		if (sq == sq) mymax = max(mymax, __float_as_uint(sq));   // a NaN descriptor must not poison the margin of every row
		if (!live) continue;
		unsigned hi[8], lo[8];
#pragma unroll
		for (int e = 0; e < 8; ++e) { hi[e] = bf16_rn(v[e]); lo[e] = bf16_rn(v[e] - __uint_as_float(hi[e] << 16)); }
		uint4 H, L;
		H.x = hi[0] | (hi[1] << 16); H.y = hi[2] | (hi[3] << 16); H.z = hi[4] | (hi[5] << 16); H.w = hi[6] | (hi[7] << 16);
		L.x = lo[0] | (lo[1] << 16); L.y = lo[2] | (lo[3] << 16); L.z = lo[4] | (lo[5] << 16); L.w = lo[6] | (lo[7] << 16);
		split[row * 32 + c] = H; split[row * 32 + 16 + c] = L;
	}
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) mymax = max(mymax, (unsigned)__shfl_xor((int)mymax, d));
	if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mymax;
	__syncthreads();
	if (threadIdx.x == 0) atomicMax(gmax_bits, max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
}

#ifndef MATCH_WAVES
#define MATCH_WAVES 4      // resident wavefronts per SIMD = workgroups per CU (33 KB of LDS each)
#endif
constexpr int NK = 4;      // kept entries per lane half (descending); a half whose NK entries all tie within the error margin sends its row to the exact full scan

// The sweep's running top-NK is kept on KEYS: the score with its low 4 mantissa bits replaced by the MFMA
// register slot (0..15) it came from.  A key is still an ordinary float within 16 ulp of the score
// (|key - score| < 2^-19 |score| <= 2^-20 (|x|^2 + max|y|^2), paid for in the margin E below), so the four
// kept keys are updated by one v_max and three v_med3 per score -- no compares, no selects, no branches --
// and the slot is recovered from the key's bits.  Which TILE a kept key came from is settled once per tile
// (topk_attribute): the new list is a merge of the old list (relative order kept) and this tile's keys, so
// walking the new list against the next unused old entry tells old (keeps its tile) from new (this tile);
// when a new key equals the old entry it is compared with, the old one is taken first -- equal keys are
// interchangeable, the other one is met at the next slot.
__device__ __forceinline__ float score_key(float s, int slot) { return __uint_as_float((__float_as_uint(s) & ~15u) | (unsigned)slot); }
__device__ __forceinline__ void topk_keys(float (&k)[NK], float key) {
	const float n3 = __builtin_amdgcn_fmed3f(k[2], k[3], key), n2 = __builtin_amdgcn_fmed3f(k[1], k[2], key), n1 = __builtin_amdgcn_fmed3f(k[0], k[1], key);
	float m; asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(k[0]), "v"(key));        // fmaxf() would canonicalise both operands first (two more VALU operations)
	k[0] = m;
	k[1] = n1; k[2] = n2; k[3] = n3;
}
__device__ __forceinline__ void topk_attribute(const float (&o)[NK], const float (&k)[NK], int (&tt)[NK], int t) {
	const int T0 = tt[0], T1 = tt[1], T2 = tt[2], T3 = tt[3];
	const bool e0 = k[0] == o[0];
	tt[0] = e0 ? T0 : t;
	const float v1 = e0 ? o[1] : o[0]; const int w1 = e0 ? T1 : T0;                 // next unused old entry
	const bool e1 = k[1] == v1;
	tt[1] = e1 ? w1 : t;
	const float v2 = e1 ? (e0 ? o[2] : o[1]) : v1; const int w2 = e1 ? (e0 ? T2 : T1) : w1;
	const bool e2 = k[2] == v2;
	tt[2] = e2 ? w2 : t;
	const bool b2 = e0 && e1, b1 = e0 || e1;                                       // old entries used by slots 0, 1: 2 / >= 1
	const float nx = b2 ? o[3] : (b1 ? o[2] : o[1]); const int nw = b2 ? T3 : (b1 ? T2 : T1);   // the old entry after v2
	const float v3 = e2 ? nx : v2; const int w3 = e2 ? nw : w2;
	tt[3] = (k[3] == v3) ? w3 : t;
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

// device-side state of one op_match_pairs call
struct MatchState {
	const float* desc; const uint4* split; const float* norms; const unsigned* gmax_bits;
	const PairDesc* pairs;
	int* fb;            // per A row: forward best column b*, -1 rejected, -2 needs the exact full scan
	float* fmn;         // per A row: exact min distance
	float* fnext;       // per A row: exact second-min distance
	int* surv;          // per pair region (res_off .. res_off+ka): A rows that passed the first ratio test
	int* nsurv;         // per pair
	int* mcount;        // accepted matches of the call ...
	int* mtrip;         // ... as (pair, a, b) triples in arrival order
	int* pcnt;          // ... and counted per pair
	int* slow_fwd; int* slow_rev;   // rows needing a full scan: (pair, row) pairs ...
	int* slow_sorted;   // ... and the queue being scanned, grouped by scanned image (k_slow_order)
	int* slow_fwd_n; int* slow_rev_n;   // ... and their counts
	int slow_cap;
	float rr;           // MATCH_REJECT_NEXT_RATIO^2 (matcher.cc:16)
};

// FeatureMatcher::match, first ratio test (matcher.cc:52); survivors are queued for the reverse pass
__device__ __forceinline__ void finish_forward(const MatchState& S, const PairDesc& pd, int pair, int a, float mn, float next_min, int min_idx) {
	const long long o = (long long)pd.res_off + a;
	S.fmn[o] = mn; S.fnext[o] = next_min;
	if (min_idx >= 0 && !(mn > S.rr * next_min)) {
		S.fb[o] = min_idx;
		const int slot = atomicAdd(&S.nsurv[pair], 1);
		S.surv[pd.res_off + slot] = a;
	} else S.fb[o] = -1;
}
// second ratio test (matcher.cc:62)
__device__ __forceinline__ void finish_reverse(const MatchState& S, const PairDesc& pd, int pair, int a, float next_min) {
	const long long o = (long long)pd.res_off + a;
	if (!(S.fmn[o] > S.rr * next_min)) {
		const int slot = atomicAdd(S.mcount, 1);           // a row is accepted at most once: slot < total rows
		int* q = S.mtrip + 3 * (long long)slot;
		q[0] = pair; q[1] = a; q[2] = S.fb[o];
		atomicAdd(&S.pcnt[pair], 1);
	}
}

// One workgroup = 128 rows of X (4 waves x 32 rows, the split X fragments resident in VGPRs) against
// all of Y, whose split rows stream through LDS 32 columns at a time.  D = Ytile * X^T so that every lane ends up
// holding 16 scores of ONE X row (C/D layout: col = lane&31), which makes the running top-4 a
// purely per-lane update (on keys, above); the two lane halves of a row (columns i & 4 == 0 / != 0 of
// every tile) exchange their lists once at the end.  The accumulators start from -|y|^2/2, so the MFMA
// chain ends on the scores x.y - |y|^2/2 = const - d/2, which only RANK columns.  The epilogue then
// re-scores the <= 6 ranked candidates of every row with the reference's exact squared L2
// (feature/dist.cc:22-57: four stride-4 fp32 partial sums walked in order, (v0+v1)+(v2+v3)):
// the row is still in registers, split across the two lane halves exactly at t = 16, so the
// lower half walks t = 0..15, hands its four partial sums to the upper half, which walks
// t = 16..31 -- candidates are software-pipelined through the two halves.  The candidate set is
// provably complete: every column whose key is within E of the row's 2nd best is re-scored, where E
// bounds twice the worst-case error of a key (dropped split terms 3 * 2^-18, 385 fp32
// accumulations, the fp32 norm, the 4 slot bits) plus the rounding of the reference's own fp32
// distance; a row one of whose halves has all 4 kept entries inside the margin is queued for an
// exact full scan instead.
template <bool REV>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MATCH_WAVES, MATCH_WAVES))) k_match_sweep(MatchState S, const WorkItem* __restrict__ work) {
	// two tile images as two OBJECTS (not one array indexed by the parity of the tile): the wait-count pass holds an LDS
	// read back behind a pending LDS-DMA unless it can tell their destinations apart, and it can for distinct variables
	__shared__ __attribute__((aligned(1024))) float s_y0[32 * YROW];
	__shared__ __attribute__((aligned(1024))) float s_y1[32 * YROW];
	__shared__ __attribute__((aligned(16))) float s_nyh[2][32];
	// the lane halves' top-4 lists of the epilogue take the place of the first tile image (dead once the sweep has ended)
	float (*s_ms)[32][2][NK] = (float (*)[32][2][NK])s_y0;
	int (*s_mi)[32][2][NK] = (int (*)[32][2][NK])(s_y0 + 4 * 32 * 2 * NK);
	const WorkItem wk = work[blockIdx.x];
	const PairDesc pd = S.pairs[wk.pair];
	const int nrows = REV ? S.nsurv[wk.pair] : pd.ka;
	if (wk.rowblock * 128 >= nrows) return;
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const int j = lane & 31, h = lane >> 5;
	const float* A = S.desc + (long long)pd.a_off * 128;
	const float* B = S.desc + (long long)pd.b_off * 128;
	const float* Y = REV ? A : B;
	const int ky = REV ? pd.ka : pd.kb;
	const float* ny = S.norms + (REV ? pd.a_off : pd.b_off);
	// the X row: FWD a = row ; REV a = survivor, X = B[b*(a)].  Evaluated twice, from the thread index: here for the split
	// fragments, and again behind the sweep for the epilogue (from an opaque copy of the index, so that the compiler
	// recomputes the five values instead of parking them in scratch memory across the tile loop, whose registers are all taken)
	auto x_row = [&](int tid_, int& row_, int& a_row_, int& x_idx_) {
		row_ = wk.rowblock * 128 + (tid_ >> 6) * 32 + (tid_ & 31);
		const int rowc = row_ < nrows ? row_ : nrows - 1;
		a_row_ = REV ? S.surv[pd.res_off + rowc] : rowc;
		x_idx_ = REV ? S.fb[pd.res_off + a_row_] : a_row_;
	};
	const uint4* YS = S.split + (long long)(REV ? pd.a_off : pd.b_off) * 32;
	uint4 xh[8], xl[8];   // MFMA B operands: block kb covers k = 16 kb + 8 h .. + 7 of row j
	{
		int row0, a0, x0;
		x_row(tid, row0, a0, x0);
		const uint4* XS = S.split + ((long long)(REV ? pd.b_off : pd.a_off) + x0) * 32;
#pragma unroll
		for (int kb = 0; kb < 8; ++kb) { xh[kb] = XS[2 * kb + h]; xl[kb] = XS[16 + 2 * kb + h]; }
	}
	// The X fragments must have LANDED before the tile loop: a load still pending at the loop header makes
	// the compiler's wait-count pass put `s_waitcnt vmcnt(0/1)` in front of the MFMAs that read it in EVERY
	// iteration.  Using the registers here forces the wait once, outside the loop.
#pragma unroll
	for (int kb = 0; kb < 8; ++kb) asm volatile("" :: "v"(xh[kb].x), "v"(xl[kb].x));
	float ts[NK]; int ti[NK];
#pragma unroll
	for (int r = 0; r < NK; ++r) { ts[r] = -FLT_MAX; ti[r] = -1; }
	const int ntiles = (ky + 31) / 32;

	// The Y tiles stream from HBM straight into LDS (buffer_load_dwordx4 ... lds: no staging registers, no address
	// arithmetic and no ds_write per tile): a tile is 16 wave-instructions of 1 KB, four per wavefront.  Instruction
	// q of wave w fills LDS rows 8 w + 2 q and + 1; lane L lands on 16-byte position (L & 31) of row 8 w + 2 q + (L >> 5),
	// so it FETCHES that row's block (L & 31) ^ (row & 15) -- the permutation the MFMA operand reads undo.  The tile
	// number goes into the instruction's scalar offset; rows past the end of Y lie outside the buffer descriptor and
	// read as zeros (their columns cannot rank: -|y|^2/2 is -FLT_MAX there).
	const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(YS), 0, (unsigned)ky * 512u, 0x00020000);
	// r & 15 = (8 w & 8) | 2 q | (L >> 5) without carries, so the fetched block is ((L & 31) ^ (8 w & 8) ^ (L >> 5)) ^ 2 q: ONE offset
	// register, piece q differs by an XOR of 32 q bytes and by 1024 q bytes that go into the instruction's immediate offset
	// (which moves the LDS destination by the same 1024 q: exactly where piece q belongs)
	const unsigned dma_off = (8u * (unsigned)wave + (unsigned)(lane >> 5)) * 512u + ((((unsigned)lane & 31u) ^ ((8u * (unsigned)wave) & 8u) ^ (unsigned)(lane >> 5)) << 4);
	float stage_ny = 0.f; bool stage_pad = false;
	auto fetch_tile = [&](int t, float* image) {
		__attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(image + 4 * wave * 256);
		__builtin_amdgcn_raw_ptr_buffer_load_lds(ysrc, dst, 16, dma_off, t * 16384, 0, 0);
		__builtin_amdgcn_raw_ptr_buffer_load_lds(ysrc, dst, 16, dma_off ^ 32u, t * 16384, 1024, 0);
		__builtin_amdgcn_raw_ptr_buffer_load_lds(ysrc, dst, 16, dma_off ^ 64u, t * 16384, 2048, 0);
		__builtin_amdgcn_raw_ptr_buffer_load_lds(ysrc, dst, 16, dma_off ^ 96u, t * 16384, 3072, 0);
		// the raw |y|^2 only: any arithmetic on it here would wait for the load at the top of the iteration
		const int gy = t * 32 + (tid & 31);
		stage_ny = ny[gy < ky ? gy : ky - 1];
		stage_pad = gy >= ky;
	};
	auto commit_tile = [&](int buf) {
		if (tid < 32) s_nyh[buf][tid] = stage_pad ? -FLT_MAX : -0.5f * stage_ny;   // -|y|^2/2: the accumulators START from it; padded columns can never rank
	};
	// this lane's operand blocks in a tile image: row j, logical block 2 kb + h (hi) and 16 + 2 kb + h (lo: + 256 bytes), i.e.
	// byte (j * 512 + ((h ^ (j & 15)) << 4)) ^ (32 kb) -- one register and one XOR per block (the images are 1 KB-aligned)
	const unsigned yoff0 = (unsigned)j * 512u + (((unsigned)h ^ ((unsigned)j & 15u)) << 4);

	// one tile: start the DMA of the next one into the OTHER image, MFMA chain + top-4 on this one
	auto tile_step = [&](int t, const float* ytile, float* other, int buf) {
		if (t + 1 < ntiles) fetch_tile(t + 1, other);
		STAMP(0);
		// the accumulators start from -|y|^2/2 of this lane's 16 columns i = (reg&3) + 8*(reg>>2) + 4*h (four
		// 16-byte LDS reads straight into the accumulator registers): the MFMA chain ends on the scores
		f32x16 acc;
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			const f32x4 v = *(const f32x4*)&s_nyh[buf][8 * g + 4 * h];
			acc[4 * g] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
		}
		// (raising the wave's priority over the chain, or splitting it over two accumulators, changes nothing:
		// DESIGN.md section 6)
#pragma unroll
		for (int kb = 0; kb < 8; ++kb) {
			const char* blk = (const char*)ytile + (yoff0 ^ (32u * (unsigned)kb));
			const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)blk), al = __builtin_bit_cast(bf16x8, *(const uint4*)(blk + 256));
			const bf16x8 bh = __builtin_bit_cast(bf16x8, xh[kb]), bl = __builtin_bit_cast(bf16x8, xl[kb]);
			acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
		}
		STAMP(1);
		// lane holds D[i][j] for i = (reg&3) + 8*(reg>>2) + 4*h : 16 Y columns of X row j
		const float old_keys[NK] = {ts[0], ts[1], ts[2], ts[3]};
#pragma unroll
		for (int reg = 0; reg < 16; ++reg) {
			topk_keys(ts, score_key(acc[reg], reg));
		}
		topk_attribute(old_keys, ts, ti, t);          // ti[] holds TILE numbers until the sweep ends
		STAMP(2);
		if (t + 1 < ntiles) commit_tile(buf ^ 1);
		STAMP(3);
		// this wavefront's DMA pieces of tile t + 1 must have LANDED before the barrier releases the other wavefronts onto
		// them.  A workgroup-scope fence only implies lgkmcnt on gfx9; the compiler happened to place vmcnt(1) here (the
		// DMAs are older than the |y|^2 load) -- an artefact of instruction order, so the wait is spelled out.  Every
		// load it covers was issued a whole MFMA chain ago.
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		STAMP(4);
	};
	fetch_tile(0, s_y0);
	commit_tile(0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (int t = 0; t < ntiles; t += 2) {
		tile_step(t, s_y0, s_y1, 0);
		if (t + 1 < ntiles) tile_step(t + 1, s_y1, s_y0, 1);
	}
	int row, a_row, x_idx;
	{
		int tid2 = tid;
		asm volatile("" : "+v"(tid2));
		x_row(tid2, row, a_row, x_idx);
	}
	const float* X = (REV ? B : A) + (long long)x_idx * 128;
	// (tile, slot) -> column: slot reg of lane half h is column (reg & 3) + 8 (reg >> 2) + 4 h of its tile; never-filled
	// entries (tile -1) and the padded columns of the last tile (score -FLT_MAX, they rank above nothing real) are no candidates
#pragma unroll
	for (int r = 0; r < NK; ++r) {
		const int reg = (int)(__float_as_uint(ts[r]) & 15u);
		const int col = ti[r] * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
		ti[r] = (ti[r] < 0 || col >= ky) ? -1 : col;
	}
	// the two lane halves of an X row exchange their lists; both then derive the same candidate set
#pragma unroll
	for (int r = 0; r < NK; ++r) { s_ms[wave][j][h][r] = ts[r]; s_mi[wave][j][h][r] = ti[r]; }
	__syncthreads();

	// ---- candidate set (identical in both halves): every kept entry within E of the row's 2nd best.  The lists
	// are NOT merged down to four: a half's list holds every column of that half above its 4th key, so as long as
	// each half's 4th key is below the threshold the union of the two lists is complete -- up to 3 + 3 candidates.
	// Only a half whose four entries ALL sit inside the margin sends the row to the exact full scan (four
	// near-ties in one half: an order of magnitude rarer than four in the row).
	const float gmax = __uint_as_float(*S.gmax_bits);
#ifndef OP_MATCH_MARGIN
#define OP_MATCH_MARGIN 8.2e-5f      // 8e-5 for the MFMA scores + 2 * 2^-20 for the keys' slot bits
#endif
	const float E = OP_MATCH_MARGIN * (S.norms[(REV ? pd.b_off : pd.a_off) + x_idx] + gmax);
	const float* ms = s_ms[wave][j][0]; const int* mi = s_mi[wave][j][0];            // [half][NK] contiguous
	const float second = fmaxf(fminf(ms[0], ms[NK]), fmaxf(ms[1], ms[NK + 1]));      // 2nd largest key of the row (invalid entries rank below every real one)
	const float thr = second - E;
	const bool overflow = (mi[NK - 1] >= 0 && ms[NK - 1] >= thr) || (mi[2 * NK - 1] >= 0 && ms[2 * NK - 1] >= thr);
	constexpr int NC = 2 * (NK - 1);
	int c[NC]; int nc = 0;
#pragma unroll
	for (int r = 0; r < NC; ++r) c[r] = 0x7fffffff;
#pragma unroll
	for (int r = 0; r < 2 * NK; ++r) {
		const int ci = mi[r];
		if (!overflow && ci >= 0 && ms[r] >= thr && !(REV && ci == a_row)) {      // REV: kk != k (matcher.cc:58)
			// insertion into ascending column order: distance ties resolve to the first index (matcher.cc:42-48)
			int v = ci;
#pragma unroll
			for (int q = 0; q < NC; ++q) { const int o = c[q]; const bool sw = v < o; c[q] = sw ? v : o; v = sw ? o : v; }
			++nc;
		}
	}
	const bool live = row < nrows;
	if (overflow) nc = 0;
	int ncmax = nc;                    // wave-uniform trip count
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(ncmax, off); ncmax = o > ncmax ? o : ncmax; }

	// ---- exact re-score, pipelined through the lane halves ----
	f32x4 xf[16];     // fp32 row elements [64h, 64h+64)
	{
		const f32x4* px = (const f32x4*)(X + 64 * h);
#pragma unroll
		for (int q = 0; q < 16; ++q) xf[q] = px[q];
	}
	float mn = REV ? 0.f : FLT_MAX, next_min = FLT_MAX; int min_idx = -1;
	if (REV) next_min = S.fnext[pd.res_off + a_row];
	float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
	for (int u = 0; u <= ncmax; ++u) {
		// lower half: first 64 elements of candidate u ; upper half: last 64 of candidate u-1
		const int cu = h == 0 ? u : u - 1;
		const bool act = cu >= 0 && cu < nc;
		float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
		if (h == 1) { w0 = v0; w1 = v1; w2 = v2; w3 = v3; }      // partial sums handed over below
		int ccol = 0;
#pragma unroll
		for (int r = 0; r < NC; ++r) ccol = (cu == r) ? c[r] : ccol;
		if (act) {
			// (scalar base + 32-bit lane offset: a per-lane 64-bit pointer Y + 64 h would be one more register pair held across the loop)
			const f32x4* py = (const f32x4*)((const char*)Y + (unsigned)(ccol * 512 + 256 * h));
#pragma unroll
			for (int q = 0; q < 16; ++q) {      // static indices keep the row in VGPRs; 4 loads in flight at a time
				const f32x4 b = py[q];
				float d;
				d = xf[q].x - b.x; w0 += d * d;
				d = xf[q].y - b.y; w1 += d * d;
				d = xf[q].z - b.z; w2 += d * d;
				d = xf[q].w - b.w; w3 += d * d;
#ifndef OP_MATCH_RESCORE_INFLIGHT
#define OP_MATCH_RESCORE_INFLIGHT 4
#endif
				if ((q & (OP_MATCH_RESCORE_INFLIGHT - 1)) == OP_MATCH_RESCORE_INFLIGHT - 1) __builtin_amdgcn_sched_barrier(0);
			}
		}
		if (h == 1 && act) {
			const float d = (w0 + w1) + (w2 + w3);
			if (REV) { if (d < next_min) next_min = d; }
			else if (d < mn) { next_min = mn; mn = d; min_idx = ccol; }
			else if (d < next_min) next_min = d;
		}
		// hand the lower half's partial sums to the upper half (lane j -> lane j + 32)
		v0 = __shfl(w0, j); v1 = __shfl(w1, j); v2 = __shfl(w2, j); v3 = __shfl(w3, j);
	}
	asm volatile("" : "+v"(a_row));        // the result addresses are formed HERE, not above the re-score loop (where they would be spilled)
	if (h == 1 && live) {
		if (overflow) {
			int* q = REV ? S.slow_rev : S.slow_fwd;
			const int slot = atomicAdd(REV ? S.slow_rev_n : S.slow_fwd_n, 1);
			if (slot < S.slow_cap) { q[2 * slot] = wk.pair; q[2 * slot + 1] = a_row; }
			if (!REV) S.fb[pd.res_off + a_row] = -2;
		} else if (REV) finish_reverse(S, pd, wk.pair, a_row, next_min);
		else finish_forward(S, pd, wk.pair, a_row, mn, next_min, min_idx);
	}
}

// rows whose NK ranked candidates all fell inside the error margin (near-duplicate descriptors):
// exact full scan, one workgroup of four wavefronts per row -- thread t scans columns t, t+256, ... with
// the sequential update of matcher.cc:42-48; the thread states are merged (lanes by shuffles, waves
// through LDS) with an order-free operator, so that the result is the one a single ascending scan
// produces (first index on equal minima, second-smallest as a multiset).
struct ScanState { float mn, next_min; int min_idx; };
__device__ __forceinline__ void scan_merge(ScanState& s, float omn, float onx, int oid) {
	const bool mine = s.mn < omn || (s.mn == omn && s.min_idx < oid);
	const float lo_next = mine ? s.next_min : onx, hi_min = mine ? omn : s.mn;
	s.next_min = lo_next < hi_min ? lo_next : hi_min;
	if (!mine) { s.mn = omn; s.min_idx = oid; }
}
// Groups the exact-scan queue by the image whose descriptors a row scans (counting sort, one workgroup; the
// order inside a group does not matter).  k_match_slow then hands each XCD a contiguous eighth of the grouped
// queue, so the rows that scan one image run on one XCD at about the same time and its descriptors come from
// HBM once and from that XCD's L2 afterwards -- the scan reads a whole descriptor set per row and is
// bandwidth-bound otherwise (5 TB/s measured with the queue in arrival order).
constexpr int kOrderBins = 4096;
template <bool REV>
__global__ void __launch_bounds__(1024) k_slow_order(MatchState S, int nimg) {
	__shared__ int s_cnt[kOrderBins];
	__shared__ int s_part[16];
	const int* q = REV ? S.slow_rev : S.slow_fwd;
	int n = *(REV ? S.slow_rev_n : S.slow_fwd_n); n = n < S.slow_cap ? n : S.slow_cap;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	for (int i = tid; i < kOrderBins; i += 1024) s_cnt[i] = 0;
	__syncthreads();
	for (int i = tid; i < n; i += 1024) { const PairDesc pd = S.pairs[q[2 * i]]; atomicAdd(&s_cnt[REV ? pd.ia : pd.ib], 1); }
	__syncthreads();
	int c[4], sum = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) { c[k] = s_cnt[tid * 4 + k]; sum += c[k]; }
	int incl = sum;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
	if (lane == 63) s_part[wave] = incl;
	__syncthreads();
	int base = incl - sum;
	for (int w = 0; w < wave; ++w) base += s_part[w];
#pragma unroll
	for (int k = 0; k < 4; ++k) { s_cnt[tid * 4 + k] = base; base += c[k]; }
	__syncthreads();
	for (int i = tid; i < n; i += 1024) {
		const int pair = q[2 * i]; const PairDesc pd = S.pairs[pair];
		const int pos = atomicAdd(&s_cnt[REV ? pd.ia : pd.ib], 1);
		S.slow_sorted[2 * pos] = pair; S.slow_sorted[2 * pos + 1] = q[2 * i + 1];
	}
}

template <bool REV>
__global__ void __launch_bounds__(256) k_match_slow(MatchState S, int grouped) {
	__shared__ f32x4 s_x[32];
	__shared__ float s_mn[4], s_nx[4];
	__shared__ int s_id[4];
	const int* q = grouped ? S.slow_sorted : (REV ? S.slow_rev : S.slow_fwd);
	int n = *(REV ? S.slow_rev_n : S.slow_fwd_n); n = n < S.slow_cap ? n : S.slow_cap;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int per = n >> 3;
	for (int lin = blockIdx.x; lin < n; lin += gridDim.x) {       // gridDim.x is a multiple of 8: lin & 7 is this workgroup's XCD
		const int i = lin < per * 8 ? (lin & 7) * per + (lin >> 3) : lin;
		const int pair = q[2 * i], a = q[2 * i + 1];
		const PairDesc pd = S.pairs[pair];
		const float* A = S.desc + (long long)pd.a_off * 128;
		const float* B = S.desc + (long long)pd.b_off * 128;
		const float* x = REV ? B + (long long)S.fb[pd.res_off + a] * 128 : A + (long long)a * 128;
		const float* Y = REV ? A : B;
		const int ky = REV ? pd.ka : pd.kb;
		if (tid < 32) s_x[tid] = ((const f32x4*)x)[tid];
		__syncthreads();
		ScanState st = {FLT_MAX, FLT_MAX, 0x7fffffff};
		for (int kk = tid; kk < ky; kk += 256) {
			if (REV && kk == a) continue;
			// euclidean_sqr_exact with the row x in LDS and the column fetched sixteen 16-byte blocks at a time
			const f32x4* py = (const f32x4*)(Y + (long long)kk * 128);
			float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll
			for (int half = 0; half < 2; ++half) {
				f32x4 b[16];
#pragma unroll
				for (int t = 0; t < 16; ++t) b[t] = py[16 * half + t];
#pragma unroll
				for (int t = 0; t < 16; ++t) {
					const f32x4 av = s_x[16 * half + t];
					float d;
					d = av.x - b[t].x; v0 += d * d;
					d = av.y - b[t].y; v1 += d * d;
					d = av.z - b[t].z; v2 += d * d;
					d = av.w - b[t].w; v3 += d * d;
				}
			}
			const float d = (v0 + v1) + (v2 + v3);
			if (d < st.mn) { st.next_min = st.mn; st.mn = d; st.min_idx = kk; }
			else if (d < st.next_min) st.next_min = d;
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) {
			const float omn = __shfl_xor(st.mn, off), onx = __shfl_xor(st.next_min, off); const int oid = __shfl_xor(st.min_idx, off);
			scan_merge(st, omn, onx, oid);
		}
		if (lane == 0) { s_mn[wave] = st.mn; s_nx[wave] = st.next_min; s_id[wave] = st.min_idx; }
		__syncthreads();
		if (tid == 0) {
#pragma unroll
			for (int w = 1; w < 4; ++w) scan_merge(st, s_mn[w], s_nx[w], s_id[w]);
			if (!REV) finish_forward(S, pd, pair, a, st.mn, st.next_min, st.min_idx == 0x7fffffff ? -1 : st.min_idx);
			else { const float f = S.fnext[pd.res_off + a]; finish_reverse(S, pd, pair, a, st.mn < f ? st.mn : f); }   // matcher.cc:57-61: the MINIMUM over kk != k joins next_min
		}
		__syncthreads();
	}
}

// ---- result lists on the device: (pair, a, b) triples in arrival order -> per pair <first, second> sorted ----
// exclusive prefix of the per-pair counts (one workgroup; a job has thousands of pairs)
__global__ void __launch_bounds__(1024) k_match_offsets(const int* __restrict__ pcnt, int npairs, int* __restrict__ poff) {
	__shared__ int s_part[16];
	__shared__ int s_carry;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (tid == 0) s_carry = 0;
	__syncthreads();
	for (int base = 0; base < npairs; base += 1024) {
		const int i = base + tid;
		const int v = i < npairs ? pcnt[i] : 0;
		int incl = v;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
		if (lane == 63) s_part[wave] = incl;
		__syncthreads();
		int pre = s_carry;
		for (int w = 0; w < wave; ++w) pre += s_part[w];
		if (i < npairs) poff[i] = pre + incl - v;
		__syncthreads();
		if (tid == 1023) s_carry = pre + incl;
		__syncthreads();
	}
	if (tid == 0) poff[npairs] = s_carry;
}
// every triple to an (unordered) slot of its pair's segment, in the reference's <first, second> form:
// <b, a> when the pair was matched with its sets swapped (matcher.cc:68-69)
__global__ void __launch_bounds__(256) k_match_place(MatchState S, const int* __restrict__ poff, int* __restrict__ fill, int2* __restrict__ seg) {
	const int n = *S.mcount;
	for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
		const int* q = S.mtrip + 3 * (long long)e;
		const int pair = q[0];
		const int slot = atomicAdd(&fill[pair], 1);
		seg[poff[pair] + slot] = S.pairs[pair].rev ? make_int2(q[2], q[1]) : make_int2(q[1], q[2]);
	}
}
// rank sort of every pair's segment by (first, second) -- arrival order on the device is arbitrary, the (a, b) of a
// pair are distinct, so ranks are a permutation.  Workgroup per pair; segments up to kSortLds entries are ranked
// from LDS, longer ones (thousands of matches in one image pair) straight from global memory.
constexpr int kSortLds = 4096;
__global__ void __launch_bounds__(256) k_match_sort(const int* __restrict__ pcnt, const int* __restrict__ poff, const int2* __restrict__ seg, int2* __restrict__ out) {
	__shared__ unsigned long long s_key[kSortLds];
	const int pair = blockIdx.x, m = pcnt[pair];
	if (m == 0) return;
	const int2* in = seg + poff[pair]; int2* o = out + poff[pair];
	const int tid = threadIdx.x;
	if (m == 1) { if (tid == 0) o[0] = in[0]; return; }
	auto key_of = [](int2 v) { return ((unsigned long long)(unsigned)v.x << 32) | (unsigned)v.y; };
	if (m <= kSortLds) {
		for (int i = tid; i < m; i += 256) s_key[i] = key_of(in[i]);
		__syncthreads();
		for (int i = tid; i < m; i += 256) {
			const unsigned long long k = s_key[i];
			int r = 0;
			for (int j = 0; j < m; ++j) r += s_key[j] < k ? 1 : 0;
			o[r] = make_int2((int)(k >> 32), (int)(unsigned)k);
		}
	} else {
		for (int i = tid; i < m; i += 256) {
			const unsigned long long k = key_of(in[i]);
			int r = 0;
			for (int j = 0; j < m; ++j) r += key_of(in[j]) < k ? 1 : 0;
			o[r] = make_int2((int)(k >> 32), (int)(unsigned)k);
		}
	}
}

}	// namespace

extern "C" {

// Workgroups of the exact-scan kernel: one per queued row up to 2048 (it strides over longer queues).  The queue length
// is only known on the device; the previous call's is the estimate (a few dozen rows for a 38-view job: 2048 workgroups
// that find nothing cost more dispatcher time than the rows take to scan).  A multiple of 8: lin & 7 is the XCD.
static unsigned slow_grid(int seen) {
	if (seen < 0) return 2048;
	const long long want = ((long long)seen + seen / 4 + 64 + 7) & ~7LL;
	return (unsigned)(want < 2048 ? want : 2048);
}

int op_match_pairs(op_ctx* ctx, const op_config* cfg, const op_features* f, const int* pairs, int npairs, op_matches** out) {
	if (!ctx || !cfg || !f || (!pairs && npairs != 0) || npairs < 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_match_pairs: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	const FeatView fv = op_features_view(f);
	if (fv.device != ctx->device) OP_FAIL(OP_ERR_INVALID, "op_match_pairs: the features live on another device than the context");
	hipStream_t st = ctx->stream;
	const long long total = fv.offsets[fv.n];
	if (!fv.desc) OP_FAIL(OP_ERR_INVALID, "op_match_pairs: features hold coordinates only (built without descriptors)");
	op_matches* m = new op_matches;
	m->npairs = npairs; m->count.assign(npairs, 0); m->offset.assign(npairs + 1, 0); m->lim.assign((size_t)npairs * 2, 0);
	m->device = ctx->device; m->stream = ctx->stream;
	if (npairs == 0 || total == 0) { m->host_valid = true; m->h_idx.resize(2); *out = m; return OP_OK; }

	std::unique_ptr<HostScope> hs(new HostScope(ctx, "matcher work list + launches (host)"));
	std::vector<WorkItem> work;
	std::vector<PairDesc> pds(npairs);
	long long res_rows = 0;
	size_t n_items = 0;                                                  // row blocks of all pairs = workgroups of a sweep
	for (int p = 0; p < npairs; ++p) {
		const int i = pairs[2 * p], j = pairs[2 * p + 1];
		if (i < 0 || j < 0 || i >= fv.n || j >= fv.n) { delete m; OP_FAIL(OP_ERR_INVALID, "op_match_pairs: image index out of range"); }
		const int rev = fv.counts[i] > fv.counts[j];                       // matcher.cc:21: the smaller set queries
		const int ia = rev ? j : i, ib = rev ? i : j;
		PairDesc& pd = pds[p];
		pd.a_off = (int)fv.offsets[ia]; pd.ka = fv.counts[ia];
		pd.b_off = (int)fv.offsets[ib]; pd.kb = fv.counts[ib];
		pd.rev = rev; pd.ia = ia; pd.ib = ib;
		pd.res_off = (int)res_rows; res_rows += pd.ka;
		m->lim[2 * (size_t)p] = fv.counts[i]; m->lim[2 * (size_t)p + 1] = fv.counts[j];
		if (pd.ka > 0 && pd.kb > 0) n_items += (size_t)((pd.ka + 127) / 128);
	}
	auto build_work_list = [&]() {
		{	// Workgroups are handed to the 8 XCDs round-robin by index.  The row blocks of one pair (which stream the
			// same Y set) go to ONE XCD, back to back: Y then comes from HBM once and from that XCD's L2 for the other
			// row blocks.  Pairs are dealt longest Y first to the XCD with the least work so far, so every XCD's
			// queue runs from its longest workgroups to its shortest and the last round of the launch is short.
			std::vector<int> order; order.reserve(npairs);
			for (int p = 0; p < npairs; ++p) if (pds[p].ka > 0 && pds[p].kb > 0) order.push_back(p);
			std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pds[a].kb > pds[b].kb; });
			std::vector<WorkItem> chunk[8]; long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for (int p : order) {
				int c = 0;
				for (int k = 1; k < 8; ++k) if (load[k] < load[c]) c = k;
				const int nrb = (pds[p].ka + 127) / 128;
				for (int rb = 0; rb < nrb; ++rb) chunk[c].push_back({p, rb});
				load[c] += (long long)nrb * pds[p].kb;
			}
			size_t longest = 0, total_items = 0;
			for (int c = 0; c < 8; ++c) { longest = std::max(longest, chunk[c].size()); total_items += chunk[c].size(); }
			work.reserve(total_items);
			for (size_t k = 0; k < longest; ++k)
				for (int c = 0; c < 8; ++c) if (k < chunk[c].size()) work.push_back(chunk[c][k]);
		}
	};
	if (res_rows >= (1LL << 30)) { delete m; OP_FAIL(OP_ERR_CAPACITY, "op_match_pairs: too many rows in one call; split the pair list"); }
	for (int i = 0; i < fv.n; ++i)
		if (fv.counts[i] >= (1 << 22)) { delete m; OP_FAIL(OP_ERR_CAPACITY, "op_match_pairs: an image with 4 M descriptors or more (the sweep addresses a descriptor set with 32-bit byte offsets)"); }
	const size_t nres = (size_t)std::max<long long>(res_rows, 1);
	const int slow_cap = (int)std::min<long long>(std::max<long long>(res_rows, 1), 1 << 22);

	// One device arena (the context's grow-only matcher scratch), one memset, one upload per call (every runtime
	// call costs microseconds).  Layout: a control block of counters -- zero-filled as one range, its head
	// [slow_fwd_n, slow_rev_n, match count, matches per pair] is what comes back to the host -- followed by
	// the per-call arrays.
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const size_t n_ctrl = 4 + 4 * (size_t)npairs + 1;                  // gmax, slow_fwd_n, slow_rev_n, mcount, pcnt[np], nsurv[np], fill[np], poff[np + 1]
	const size_t o_ctrl = 0;
	const size_t o_trip = al(sizeof(int) * n_ctrl);
	const size_t o_seg = al(o_trip + sizeof(int) * 3 * nres);          // unsorted <first, second> segments
	const size_t o_norms = al(o_seg + sizeof(int) * 2 * nres);
	const size_t o_split = al(o_norms + sizeof(float) * total);
	const size_t o_fmn = al(o_split + 512 * (size_t)total);
	const size_t o_fnext = al(o_fmn + sizeof(float) * nres);
	const size_t o_fb = al(o_fnext + sizeof(float) * nres);
	const size_t o_surv = al(o_fb + sizeof(int) * nres);
	const size_t o_slow = al(o_surv + sizeof(int) * nres);
	const size_t o_up = al(o_slow + sizeof(int) * 6 * (size_t)slow_cap);       // forward queue, reverse queue, the grouped copy of the one being scanned
	const size_t up_bytes = sizeof(PairDesc) * npairs + sizeof(WorkItem) * n_items;
	const size_t arena_bytes = o_up + al(up_bytes);
	char* arena = nullptr;
	const size_t hres_bytes = al(sizeof(int) * (3 + (size_t)npairs));
	char* pin = (char*)ctx->pinned_scratch(hres_bytes + al(up_bytes));          // pinned: copies run at link rate, asynchronously
	int rc = OP_OK;
	if (!pin) { delete m; OP_FAIL(OP_ERR_HIP, "op_match_pairs: pinned host allocation failed"); }
	int* h_res = (int*)pin;                                                // slow_fwd_n, slow_rev_n, match count, matches per pair
	h_res[0] = h_res[1] = h_res[2] = 0;
	std::memcpy(pin + hres_bytes, pds.data(), sizeof(PairDesc) * npairs);
#define MCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { op_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); rc = OP_ERR_HIP; goto done; } } while (0)
	MCHK(ctx->match_arena.ensure(arena_bytes));
	arena = (char*)ctx->match_arena.p;
	{
		int* ctrl = (int*)(arena + o_ctrl);
		int* d_pcnt = ctrl + 4; int* d_fill = ctrl + 4 + 2 * (size_t)npairs; int* d_poff = ctrl + 4 + 3 * (size_t)npairs;
		MCHK(hipMemsetAsync(ctrl, 0, sizeof(int) * n_ctrl, st));
		MatchState S;
		S.desc = fv.desc; S.split = (const uint4*)(arena + o_split); S.norms = (const float*)(arena + o_norms);
		S.gmax_bits = (const unsigned*)ctrl; S.pairs = (const PairDesc*)(arena + o_up);
		S.fb = (int*)(arena + o_fb); S.fmn = (float*)(arena + o_fmn); S.fnext = (float*)(arena + o_fnext);
		S.surv = (int*)(arena + o_surv); S.nsurv = ctrl + 4 + npairs; S.mcount = ctrl + 3; S.mtrip = (int*)(arena + o_trip); S.pcnt = d_pcnt;
		S.slow_fwd = (int*)(arena + o_slow); S.slow_rev = S.slow_fwd + 2 * (size_t)slow_cap; S.slow_sorted = S.slow_fwd + 4 * (size_t)slow_cap;
		S.slow_fwd_n = ctrl + 1; S.slow_rev_n = ctrl + 2; S.slow_cap = slow_cap;
		const int grouped = fv.n <= kOrderBins;                         // the grouping kernel keeps one LDS counter per image
		S.rr = cfg->MATCH_REJECT_NEXT_RATIO * cfg->MATCH_REJECT_NEXT_RATIO;   // matcher.cc:16
		const WorkItem* d_work = (const WorkItem*)(arena + o_up + sizeof(PairDesc) * npairs);
		{
			ProfScope ps(ctx, "matcher norms");
			hipLaunchKernelGGL(k_split_bf16, dim3((unsigned)std::min<long long>((total * 16 + 255) / 256, 1024)), dim3(256), 0, st, fv.desc, total,
					(uint4*)(arena + o_split), (float*)(arena + o_norms), (unsigned*)ctrl);
			MCHK(hipGetLastError());
		}
		// the split kernel is on its way: the work list (a sort and a deal of the pairs) is built while it runs
		build_work_list();
		if (work.size() != n_items) { op_set_error("op_match_pairs: work list size mismatch"); rc = OP_ERR_HIP; goto done; }
		if (!work.empty()) std::memcpy(pin + hres_bytes + sizeof(PairDesc) * npairs, work.data(), sizeof(WorkItem) * work.size());
		MCHK(hipMemcpyAsync(arena + o_up, pin + hres_bytes, up_bytes, hipMemcpyHostToDevice, st));
		if (!work.empty()) {
			{
				ProfScope ps(ctx, "matcher mfma forward");
				hipLaunchKernelGGL(k_match_sweep<false>, dim3((unsigned)work.size()), dim3(256), 0, st, S, d_work);
				MCHK(hipGetLastError());
				if (grouped) hipLaunchKernelGGL(k_slow_order<false>, dim3(1), dim3(1024), 0, st, S, fv.n);
				hipLaunchKernelGGL(k_match_slow<false>, dim3(slow_grid(ctx->match_slow_seen[0])), dim3(256), 0, st, S, grouped);
				MCHK(hipGetLastError());
			}
			{
				// the reverse strip: same work list; row blocks beyond a pair's survivor count exit at once
				ProfScope ps(ctx, "matcher mfma reverse");
				hipLaunchKernelGGL(k_match_sweep<true>, dim3((unsigned)work.size()), dim3(256), 0, st, S, d_work);
				MCHK(hipGetLastError());
				if (grouped) hipLaunchKernelGGL(k_slow_order<true>, dim3(1), dim3(1024), 0, st, S, fv.n);
				hipLaunchKernelGGL(k_match_slow<true>, dim3(slow_grid(ctx->match_slow_seen[1])), dim3(256), 0, st, S, grouped);
				MCHK(hipGetLastError());
			}
			{
				ProfScope ps(ctx, "matcher result lists");
				hipLaunchKernelGGL(k_match_offsets, dim3(1), dim3(1024), 0, st, (const int*)d_pcnt, npairs, d_poff);
				hipLaunchKernelGGL(k_match_place, dim3((unsigned)std::min<size_t>((nres + 255) / 256, 512)), dim3(256), 0, st, S, (const int*)d_poff, d_fill, (int2*)(arena + o_seg));
				MCHK(hipGetLastError());
			}
			MCHK(hipMemcpyAsync(h_res, ctrl + 1, sizeof(int) * (3 + (size_t)npairs), hipMemcpyDeviceToHost, st));
		}
		hs.reset(); hs.reset(new HostScope(ctx, "matcher wait + counts (host)"));
		MCHK(hipStreamSynchronize(st));
		resolve_profile(ctx);                            // (the sort below stays pending until the next resolve: no second wait)
		if (!work.empty()) {
			ctx->match_slow_seen[0] = h_res[0]; ctx->match_slow_seen[1] = h_res[1];
			if (h_res[0] > slow_cap || h_res[1] > slow_cap) {
				// more rows needed the exact full scan than the queue holds (> 4 M rows of near-duplicate
				// descriptors in one call): the rows beyond the queue were not matched -- never return that as OP_OK
				op_set_error("op_match_pairs: exact-scan queue overflow (" + std::to_string(std::max(h_res[0], h_res[1])) + " rows > " +
						std::to_string(slow_cap) + "); split the pair list");
				rc = OP_ERR_CAPACITY; goto done;
			}
			for (int p = 0; p < npairs; ++p) { m->count[p] = h_res[3 + p]; m->offset[p + 1] = m->offset[p] + h_res[3 + p]; }
			m->total = m->offset[npairs];
			if (m->total != h_res[2]) { op_set_error("op_match_pairs: per-pair counts do not add up to the match count"); rc = OP_ERR_HIP; goto done; }
			if (m->total) {
				// the sorted lists go straight into the result buffer (exact size, known now); the kernel reads the
				// context's arena, which stays valid: whatever uses this context next is ordered behind it
				MCHK(pool_alloc((void**)&m->d_idx, sizeof(int) * 2 * (size_t)m->total));
				ProfScope ps(ctx, "matcher result lists");
				hipLaunchKernelGGL(k_match_sort, dim3((unsigned)npairs), dim3(256), 0, st, (const int*)d_pcnt, (const int*)d_poff, (const int2*)(arena + o_seg), (int2*)m->d_idx);
				MCHK(hipGetLastError());
				MCHK(hipEventCreateWithFlags(&m->produced, hipEventDisableTiming));
				MCHK(hipEventRecord(m->produced, st));
			}
		}
	}
done:
	hs.reset();
#undef MCHK
	if (rc != OP_OK) { delete m; return rc; }
	*out = m;
	return OP_OK;
}


int op_matches_from_host(const int* const* idx_pairs, const int* counts, int npairs, op_matches** out) {
	if (!idx_pairs || !counts || npairs < 0 || !out) OP_FAIL(OP_ERR_INVALID, "op_matches_from_host: bad argument");
	op_matches* m = new op_matches;
	m->npairs = npairs; m->count.assign(npairs, 0); m->offset.assign(npairs + 1, 0);
	for (int p = 0; p < npairs; ++p) {
		if (counts[p] < 0) { delete m; OP_FAIL(OP_ERR_INVALID, "negative count"); }
		m->count[p] = counts[p]; m->offset[p + 1] = m->offset[p] + counts[p];
	}
	m->total = m->offset[npairs];
	m->h_idx.resize((size_t)std::max<int64_t>(m->total, 1) * 2);
	for (int p = 0; p < npairs; ++p)
		if (counts[p]) std::memcpy(m->h_idx.data() + 2 * m->offset[p], idx_pairs[p], sizeof(int) * 2 * (size_t)counts[p]);
	m->host_valid = true;
	*out = m;
	return OP_OK;
}

// Two results of op_match_pairs on ONE device as one: the pairs of a, then the pairs of b, lists back to back in a new device
// buffer (two device-to-device copies on the context's stream).  A rank of a sharded job matches the pairs of two own images
// while the other ranks' features travel and the rest afterwards -- two op_matches; joined, its RANSAC stage is ONE
// op_ransac_pairs call instead of two (the call's latency does not depend on the pair count: DESIGN section 6).
// The indices keep their meaning: they count keypoints inside an image, whichever feature table holds it.
int op_matches_concat(op_ctx* ctx, const op_matches* a, const op_matches* b, op_matches** out) {
	if (!ctx || !a || !b || !out) OP_FAIL(OP_ERR_INVALID, "op_matches_concat: bad argument");
	if (!a->parts.empty() || !b->parts.empty()) OP_FAIL(OP_ERR_UNSUPPORTED, "op_matches_concat: results of a device group cannot be joined");
	HIPCHK(hipSetDevice(ctx->device));
	std::unique_ptr<op_matches> m(new op_matches);
	m->npairs = a->npairs + b->npairs;
	m->count = a->count; m->count.insert(m->count.end(), b->count.begin(), b->count.end());
	m->offset.assign(m->npairs + 1, 0);
	for (int p = 0; p < m->npairs; ++p) m->offset[p + 1] = m->offset[p] + m->count[p];
	m->total = m->offset[m->npairs];
	if (!a->lim.empty() && !b->lim.empty()) { m->lim = a->lim; m->lim.insert(m->lim.end(), b->lim.begin(), b->lim.end()); }
	m->device = ctx->device; m->stream = ctx->stream;
	const bool resident = (a->total == 0 || (a->d_idx && a->device == ctx->device)) && (b->total == 0 || (b->d_idx && b->device == ctx->device));
	if (resident && m->total) {
		if (a->total && a->stream != ctx->stream) HIPCHK(hipStreamSynchronize(a->stream));     // the per-pair sort that fills a list is only ordered on its own stream
		if (b->total && b->stream != ctx->stream) HIPCHK(hipStreamSynchronize(b->stream));
		HIPCHK(pool_alloc((void**)&m->d_idx, sizeof(int) * 2 * (size_t)m->total));
		if (a->total) HIPCHK(hipMemcpyAsync(m->d_idx, a->d_idx, sizeof(int) * 2 * (size_t)a->total, hipMemcpyDeviceToDevice, ctx->stream));
		if (b->total) HIPCHK(hipMemcpyAsync(m->d_idx + 2 * a->total, b->d_idx, sizeof(int) * 2 * (size_t)b->total, hipMemcpyDeviceToDevice, ctx->stream));
		HIPCHK(hipEventCreateWithFlags(&m->produced, hipEventDisableTiming));
		HIPCHK(hipEventRecord(m->produced, ctx->stream));
	}
	// the host mirror comes along when both sides already have theirs (a job that gathers its lists has fetched them)
	bool both_host;
	{ std::lock_guard<std::mutex> la(a->mu); both_host = a->host_valid; }
	{ std::lock_guard<std::mutex> lb(b->mu); both_host = both_host && b->host_valid; }
	if (both_host || !resident) {
		const int* ha = op_matches_host(a); const int* hb = op_matches_host(b);
		if (!ha || !hb) return OP_ERR_HIP;
		m->h_idx.resize((size_t)std::max<int64_t>(m->total, 1) * 2);
		if (a->total) std::memcpy(m->h_idx.data(), ha, sizeof(int) * 2 * (size_t)a->total);
		if (b->total) std::memcpy(m->h_idx.data() + 2 * a->total, hb, sizeof(int) * 2 * (size_t)b->total);
		m->host_valid = true;
	}
	*out = m.release();
	return OP_OK;
}

int op_matches_count(const op_matches* m, int p) { return (m && p >= 0 && p < m->npairs) ? m->count[p] : 0; }
int op_matches_copy(const op_matches* m, int p, int* idx_pairs) {
	if (!m || p < 0 || p >= m->npairs || !idx_pairs) OP_FAIL(OP_ERR_INVALID, "op_matches_copy: bad argument");
	const int* h = op_matches_host(m);
	if (!h) return OP_ERR_HIP;
	if (m->count[p]) std::memcpy(idx_pairs, h + 2 * m->offset[p], sizeof(int) * 2 * (size_t)m->count[p]);
	return OP_OK;
}
int op_matches_copy_all(const op_matches* m, int* idx_pairs, int64_t* offsets) {
	if (!m) OP_FAIL(OP_ERR_INVALID, "op_matches_copy_all: bad argument");
	if (offsets) std::copy(m->offset.begin(), m->offset.end(), offsets);
	if (idx_pairs && m->total) {
		const int* h = op_matches_host(m);
		if (!h) return OP_ERR_HIP;
		std::memcpy(idx_pairs, h, sizeof(int) * 2 * (size_t)m->total);
	}
	return OP_OK;
}
const int* op_matches_device_list(const op_matches* m) { return m ? m->d_idx : nullptr; }
int64_t op_matches_total(const op_matches* m) { return m ? m->total : 0; }
void op_matches_free(op_matches* m) { delete m; }

}	// extern "C"
