// ransac.hip -- batched RANSAC homography / affine estimation for all matched image pairs.
//
// Replaces TransformEstimation::get_transform (stitch/transform_estimate.cc:49-87) as called per
// pair by Stitcher::match_image (stitch/stitcher.cc:66-94) and CylinderStitcher
// (stitch/cylstitcher.cc:75-77,115-117):
//   device  one lane per hypothesis: 8-point (7 for affine) sample -> scale-normalised DLT ->
//           health() -> inlier count over the pair's matches (staged in LDS); a second tiny
//           kernel picks the first hypothesis with the maximal count (update_max semantics);
//   host    the per-pair epilogue the reference runs once: inliers of the winner, refit on all
//           inliers and the geometric acceptance gates of fill_inliers_to_matchinfo (:150-218),
//           OpenMP-parallel over pairs.  It shares ransac_math.hpp with the kernel, so the
//           winner's homography is recomputed bit-identically.
// Sampling: the reference seeds std::mt19937 from std::random_device per call (unseeded,
// SURVEY F4).  Here every pair gets an explicit 32-bit seed (caller-supplied or derived from a
// base seed and the pair index); the draw sequence is std::mt19937's, with the reference's
// rejection of repeated indices (:70-77), so a run is reproducible and can be replayed against
// the CPU path with the same seed.
#include "internal.hpp"
#include "ransac_math.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <omp.h>

struct op_features;
struct FeatView { int n; const int* counts; const int64_t* offsets; const float* desc; int device; };
FeatView op_features_view(const op_features* f);
const double* op_features_coor_device(const op_features* f);
const double* op_features_coor_host(const op_features* f, op_ctx* ctx);
struct op_matches;
int op_matches_num_pairs(const op_matches* m);
const std::vector<int>& op_matches_counts(const op_matches* m);
const std::vector<int64_t>& op_matches_offsets(const op_matches* m);
const std::vector<int>& op_matches_limits(const op_matches* m);
const int* op_matches_host(const op_matches* m);
const int* op_matches_device(const op_matches* m, int device, hipStream_t consumer);

using opransac::P2;

struct op_ransac_result {
	struct Item {
		int ok = 0; float confidence = 0; double homo[9] = {0};
		std::vector<int> inliers;         // indices into the pair's match list
		int best_count = -1, best_hyp = -1;
	};
	std::vector<Item> items;
	uint64_t pairs_hash = 0;          // FNV-1a of the (i, j) list the result was computed for (op_pairwise_table checks its argument against it)
};
static uint64_t pair_list_hash(const int* pairs, int npairs) {
	uint64_t h = 1469598103934665603ULL;
	for (int k = 0; k < 2 * npairs; ++k) { h ^= (uint64_t)(uint32_t)pairs[k]; h *= 1099511628211ULL; }
	return h;
}

// multi.hip: parts[k] holds the results of the pairs index[k][0..] -> one result in job order (parts are consumed)
op_ransac_result* op_ransac_merge(op_ransac_result* const* parts, const std::vector<std::vector<int>>& index, int npairs) {
	op_ransac_result* r = new op_ransac_result;
	r->items.resize(npairs);
	for (size_t k = 0; k < index.size(); ++k)
		for (size_t q = 0; q < index[k].size(); ++q) r->items[index[k][q]] = std::move(parts[k]->items[q]);
	r->pairs_hash = 0;                // a merged result is checked by its size only (the parts hashed their own sub-lists)
	return r;
}

namespace {

struct PairArgs {
	int pts_off, m, affine, nsample; double inlier_dist; long long samp_off;
	long long moff;          // first entry of the pair's <first, second> list in the job's match list
	int off_i, off_j;        // first keypoint of image i / image j in the feature table
};

constexpr int RANSAC_PTS_CHUNK = 512;

// The matched point pairs of every live image pair, gathered where both inputs already are: the match lists
// op_match_pairs left in HBM and the keypoint coordinates of op_features (TransformEstimation's constructor
// arguments, transform_estimate.cc:26-33: match.data[k] -> kp1[first], kp2[second]).  grid (ceil(max m / 256), live pairs)
__global__ void __launch_bounds__(256) k_ransac_gather(const PairArgs* __restrict__ pairs, const int* __restrict__ active,
		const int2* __restrict__ midx, const double2* __restrict__ coor, double* __restrict__ pts) {
	const PairArgs pa = pairs[active[blockIdx.y]];
	const int k = blockIdx.x * 256 + threadIdx.x;
	if (k >= pa.m) return;
	const int2 ab = midx[pa.moff + k];
	const double2 p1 = coor[pa.off_i + ab.x], p2 = coor[pa.off_j + ab.y];
	double2* o = (double2*)(pts + ((long long)pa.pts_off + k) * 4);
	o[0] = p1; o[1] = p2;
}

// grid (ceil(iters/256), live pairs)
__global__ void __launch_bounds__(256) k_ransac_hyp(const PairArgs* __restrict__ pairs, const double* __restrict__ pts /* m x {p1x,p1y,p2x,p2y} */,
		const unsigned short* __restrict__ samples, int iters, int* __restrict__ counts /* npairs x iters */, const int* __restrict__ active) {
	__shared__ double s_pts[RANSAC_PTS_CHUNK * 4];
	const int pair = active[blockIdx.y];
	const PairArgs pa = pairs[pair];
	const int hyp = blockIdx.x * 256 + threadIdx.x;
	const double* P = pts + (long long)pa.pts_off * 4;
	double H[9];
	bool ok = false;
	if (hyp < iters && pa.m >= pa.nsample && pa.m >= 8) {      // same gate as k_ransac_samples: otherwise no sample exists
		const unsigned short* s = samples + pa.samp_off + (long long)hyp * 8;
		int idx[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) idx[i] = s[i];
		auto get1 = [&](int i) { const double* q = P + (long long)idx[i] * 4; return P2{q[0], q[1]}; };
		auto get2 = [&](int i) { const double* q = P + (long long)idx[i] * 4; return P2{q[2], q[3]}; };
		opransac::calc_transform(pa.nsample, get1, get2, pa.affine != 0, H);
		ok = opransac::health(H);
	}
	int cnt = 0;
	for (int cb = 0; cb < pa.m; cb += RANSAC_PTS_CHUNK) {
		const int cn = pa.m - cb < RANSAC_PTS_CHUNK ? pa.m - cb : RANSAC_PTS_CHUNK;
		__syncthreads();
		for (int e = threadIdx.x; e < cn * 4; e += 256) s_pts[e] = P[(long long)cb * 4 + e];
		__syncthreads();
		if (ok)
			for (int i = 0; i < cn; ++i)
				cnt += opransac::is_inlier(H, P2{s_pts[4 * i], s_pts[4 * i + 1]}, P2{s_pts[4 * i + 2], s_pts[4 * i + 3]}, pa.inlier_dist) ? 1 : 0;
	}
	if (hyp < iters) counts[(long long)blockIdx.y * iters + hyp] = ok ? cnt : -1;      // per LIVE pair (slot blockIdx.y)
}

// Sample tables: the std::mt19937 draw sequence of TransformEstimation::get_transform with its
// rejection of repeated indices (transform_estimate.cc:64-77), one workgroup per pair.
//
// The reference consumes the stream sequentially: hypothesis K takes draws until it holds ns distinct
// indices, so where a hypothesis starts depends on every rejection before it.  Instead of walking the
// stream draw by draw (round 1: 1.75 ms, the longest kernel of the whole pipeline), the walk is
// turned into pointer jumping over a chunk of the stream held in LDS:
//   1. the generator state is twisted block by block (three dependency phases, all 256 threads) and
//      every block of 624 draws is tempered and reduced mod m into rd[];
//   2. next(i) = the stream position right after the sample that STARTS at position i (the first
//      position by which ns distinct values were seen) is computed for every i independently;
//   3. the start S[k] of the k-th hypothesis is next^k(0): S[k + 2^r] = next^(2^r)(S[k]) while the
//      table is squared in place (next^(2^r) -> next^(2^(r+1))), log2(#hypotheses) rounds;
//   4. every hypothesis is then re-walked from its start in parallel and written out.
// A chunk that does not hold all hypotheses (tiny m: many rejections) carries the draws of its first
// incomplete sample to the front of the buffer and goes round again.  The result is, draw for draw,
// the table the sequential automaton produces (tests: op_ransac_pairs == oracle for injected seeds).
constexpr int RS_T = 512;                  // 8 wavefronts per workgroup, 4 workgroups per CU (LDS): every live pair of a config-4 job is resident at once
constexpr int RS_BLK = 9;                  // generator blocks per chunk (38 KB of LDS: four workgroups per CU)
constexpr int RS_N = RS_BLK * 624;         // 5616 draws; 1500 hypotheses of 8 need ~12.4 k at m ~ 100: a few chunks, each sized to what is left
constexpr int RS_SMAX = 1024;              // > RS_N / 7 + 2 hypothesis starts per chunk
constexpr int RS_W = 16;                   // draws of a sample walk that are pre-loaded into registers
constexpr unsigned short RS_END = 0xFFFF;  // "no complete sample starts here"

// Walk the sample that starts at stream position i: ns distinct values in draw order (v[]), returns
// the position right after its last draw, or -1 when the chunk [0, N) ends first.  The first RS_W
// draws are loaded up front (independent LDS reads) and consumed from registers; only a sample with
// more than RS_W - ns rejections continues with dependent reads.
__device__ __forceinline__ int rs_walk(const unsigned short* rd, int i, int N, int ns, int (&v)[8]) {
	int w[RS_W];
#pragma unroll
	for (int j = 0; j < RS_W; ++j) w[j] = i + j < N ? (int)rd[i + j] : -2 - j;
#pragma unroll
	for (int q = 0; q < 8; ++q) v[q] = -1;
	int cnt = 0, end = -1;
#pragma unroll
	for (int j = 0; j < RS_W; ++j) {
		const int r = w[j];
		bool take = cnt < ns && r >= 0;
#pragma unroll
		for (int q = 0; q < 8; ++q) take = take && v[q] != r;
#pragma unroll
		for (int q = 0; q < 8; ++q) v[q] = (take && q == cnt) ? r : v[q];
		cnt += take ? 1 : 0;
		end = (take && cnt == ns) ? i + j + 1 : end;
	}
	if (cnt < ns) {
		int j = i + RS_W;
		while (cnt < ns && j < N) {
			const int r = rd[j++];
			bool take = true;
#pragma unroll
			for (int q = 0; q < 8; ++q) take = take && v[q] != r;
#pragma unroll
			for (int q = 0; q < 8; ++q) v[q] = (take && q == cnt) ? r : v[q];
			cnt += take ? 1 : 0;
		}
		end = cnt == ns ? j : -1;
	}
	return end;
}

// m <= 64: the selected set of a sample is a 64-bit mask (transform_estimate.cc:73-75 as a bit test),
// exact for any sample length.  Returns the end position (or -1 when the chunk ends first); with
// OUT the accepted draws are also returned in order.
template <bool OUT>
__device__ __forceinline__ int rs_walk_mask(const unsigned short* rd, int i, int N, int ns, int (&v)[8]) {
	int w[RS_W];
#pragma unroll
	for (int j = 0; j < RS_W; ++j) w[j] = i + j < N ? (int)rd[i + j] : -1;
	unsigned long long mask = 0ULL;
	int cnt = 0, end = -1;
#pragma unroll
	for (int j = 0; j < RS_W; ++j) {
		const int r = w[j];
		const unsigned long long bit = 1ULL << (r & 63);
		const bool take = cnt < ns && r >= 0 && !(mask & bit);
		if (OUT) {
#pragma unroll
			for (int q = 0; q < 8; ++q) v[q] = (take && q == cnt) ? r : v[q];
		}
		mask |= take ? bit : 0ULL;
		cnt += take ? 1 : 0;
		end = (take && cnt == ns) ? i + j + 1 : end;
	}
	if (cnt < ns) {
		int j = i + RS_W;
		while (cnt < ns && j < N) {
			const int r = rd[j++];
			const unsigned long long bit = 1ULL << r;
			const bool take = !(mask & bit);
			if (OUT) {
#pragma unroll
				for (int q = 0; q < 8; ++q) v[q] = (take && q == cnt) ? r : v[q];
			}
			mask |= bit;
			cnt += take ? 1 : 0;
		}
		end = cnt == ns ? j : -1;
	}
	return end;
}

// next(i) for ALL start positions of a chunk.  The sample that starts at i accepts the first NS distinct values of
// rd[i..): next(i) - 1 is the position where the NS-th distinct value first occurs.  Keep, for the current i, the
// distinct values to the right ordered by their first occurrence (val[0] at pos[0] the nearest): stepping from i + 1
// to i moves x = rd[i] to the front -- it is taken out of the list where it stood (or the last entry drops off) and
// everything in front of that place moves one back.  So one thread walks a segment of starts right to left at one
// list update (NS compares, NS conditional moves of a value and a position) per start, instead of one whole sample
// walk per start; the list at the segment's right end is the sample walked forward from there (first occurrences in
// order), or, when the chunk ends before NS distinct values were seen, the same right-to-left walk from the chunk's end.
// Any m: values are compared directly.  The result is next(i) of the sequential rejection loop
// (transform_estimate.cc:70-77) for every i, END where the chunk ends first.
constexpr int RS_SEG = 24;                 // start positions per thread (a thread pays one forward sample walk for its segment)
static_assert(RS_SEG * RS_T >= RS_N, "rs_next_table: one segment per thread must cover a chunk");
template <int NS>
__device__ __forceinline__ void rs_next_table(const unsigned short* rd, unsigned short* J, int N, int ns_rt, int tid) {
	const int ns = NS ? NS : ns_rt;        // NS = 0: run-time sample size (<= 8)
	const int a0 = tid * RS_SEG;
	if (tid == 0) J[N] = RS_END;
	if (a0 >= N) return;
	const int b = a0 + RS_SEG < N ? a0 + RS_SEG : N;
	int val[8], pos[8];
#pragma unroll
	for (int q = 0; q < 8; ++q) { val[q] = -1; pos[q] = 0; }
	int cnt = 0;
	// x moves to the front of the list: entry q takes entry q - 1 while x was not met in front of q
	auto to_front = [&](int x, int at) {
		int pv = val[0], pp = pos[0];      // the entry that stood one place in front
		bool c = true, absent = true;      // x not met in front of q; x is none of the first ns entries
		val[0] = x; pos[0] = at;
#pragma unroll
		for (int q = 1; q < 8; ++q) {
			c = c && pv != x;
			absent = (q == ns) ? c : absent;
			const int tv = val[q], tp = pos[q];
			val[q] = c ? pv : tv; pos[q] = c ? pp : tp;
			pv = tv; pp = tp;
		}
		if (ns == 8) absent = c && pv != x;
		cnt += (absent && cnt < ns) ? 1 : 0;
	};
	// the list at b: the sample that starts at b, walked forward; new values go to the FRONT here (newest first) ...
	int t = b;
	while (t < N && cnt < ns) {
		const int x = rd[t];
		bool isnew = true;
#pragma unroll
		for (int q = 0; q < 8; ++q) isnew = isnew && val[q] != x;
		if (isnew) {
#pragma unroll
			for (int q = 7; q >= 1; --q) { val[q] = val[q - 1]; pos[q] = pos[q - 1]; }
			val[0] = x; pos[0] = t; ++cnt;
		}
		++t;
	}
	if (cnt == ns) {
		// ... and the first ns entries are turned round into first-occurrence order
#pragma unroll
		for (int q = 0; q < 4; ++q)
#pragma unroll
			for (int r = q + 1; r < 8; ++r)
				if (q + r == ns - 1) { const int tv = val[q], tp = pos[q]; val[q] = val[r]; pos[q] = pos[r]; val[r] = tv; pos[r] = tp; }
	} else {
		// the chunk ended first: fewer than ns distinct values in [b, N); their list, by the right-to-left walk
#pragma unroll
		for (int q = 0; q < 8; ++q) val[q] = -1;
		cnt = 0;
		for (int i = N - 1; i >= b; --i) to_front((int)rd[i], i);
	}
	for (int i = b - 1; i >= a0; --i) {
		to_front((int)rd[i], i);
		int last = pos[7];
#pragma unroll
		for (int q = 0; q < 7; ++q) last = (q == ns - 1) ? pos[q] : last;
		J[i] = cnt >= ns ? (unsigned short)(last + 1) : RS_END;
	}
}

// std::mt19937::seed for every pair at once (thread per pair; the recurrence is serial in i)
__global__ void __launch_bounds__(64) k_ransac_seed(const unsigned* __restrict__ seeds, const int* __restrict__ active, int nactive, unsigned* __restrict__ state /* 624 x nactive */) {
	const int q = blockIdx.x * 64 + threadIdx.x;
	if (q >= nactive) return;
	const int p = active[q];
	unsigned v = seeds[p];
	unsigned* st = state + q;                          // word-major: the lanes of a wavefront store to one cache line
	st[0] = v;
	for (int i = 1; i < 624; ++i) { v = 1812433253u * (v ^ (v >> 30)) + (unsigned)i; st[(long long)i * nactive] = v; }
}

__global__ void __launch_bounds__(RS_T) k_ransac_samples(const PairArgs* __restrict__ pairs, const unsigned* __restrict__ state,
		int iters, unsigned short* __restrict__ samples, const int* __restrict__ active) {
	__shared__ unsigned mt[624];
	__shared__ unsigned short rd[RS_N + 8];
	__shared__ unsigned short Ja[RS_N + 2], Jb[RS_N + RS_W + 2];         // Jb doubles as the previous-equal table before the squaring starts
	__shared__ unsigned short S[RS_SMAX];
	__shared__ int s_cnt;
	const int pair = active[blockIdx.x];               // only pairs with enough matches get a workgroup (the host compacts the list)
	const PairArgs pa = pairs[pair];
	const int m = pa.m, ns = pa.nsample, tid = threadIdx.x;
	if (m < 8 || m < ns) return;                       // ESTIMATE_MIN_NR_MATCH (:21,39) / :55
	for (int i = tid; i < 624; i += RS_T) mt[i] = state[(long long)i * gridDim.x + blockIdx.x];
	unsigned short* sp = samples + pa.samp_off;
	auto twist_word = [](unsigned hi, unsigned lo, unsigned far) {
		const unsigned y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
		return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
	};
	int kdone = 0, carry = 0;                          // workgroup-uniform
	// expected draws per sample, x16: sum over the ns picks of m / (m - picks so far); re-measured after every chunk
	long long draws_per_sample_x16 = 0;
	for (int q = 0; q < ns; ++q) draws_per_sample_x16 += (16LL * m + (m - q) - 1) / (m - q);
	while (kdone < iters) {
		int nb = (RS_N - carry) / 624;
		if (nb < 1) break;                             // one sample longer than 11 k draws: cannot happen for m >= ns
		{                                              // only what the remaining hypotheses are expected to draw (+12 %)
			const long long need = (long long)(iters - kdone) * draws_per_sample_x16 / 16;
			const long long want = (need + need / 8 + 32 - carry + 623) / 624;
			nb = want < 1 ? 1 : (want < nb ? (int)want : nb);
		}
		for (int b = 0; b < nb; ++b) {
			// ---- twist: thread t owns words t, 227 + t, 454 + t; the old values every phase needs are
			// read before anything is overwritten, each phase then reads only finished words ----
			__syncthreads();
			unsigned o[6] = {0, 0, 0, 0, 0, 0};
			unsigned lastold = 0;
			if (tid < 227) { o[0] = mt[tid]; o[1] = mt[tid + 1]; o[2] = mt[227 + tid]; o[3] = mt[228 + tid]; }
			if (tid < 169) { o[4] = mt[454 + tid]; o[5] = mt[455 + tid]; }
			if (tid == 255) lastold = mt[623];
			__syncthreads();
			if (tid < 227) mt[tid] = twist_word(o[0], o[1], mt[tid + 397]);
			__syncthreads();
			if (tid < 227) mt[227 + tid] = twist_word(o[2], o[3], mt[tid]);
			__syncthreads();
			if (tid < 169) mt[454 + tid] = twist_word(o[4], o[5], mt[227 + tid]);
			if (tid == 255) mt[623] = twist_word(lastold, mt[0], mt[396]);
			__syncthreads();
			// ---- temper + reduce the 624 draws of this block ----
			for (int i = tid; i < 624; i += RS_T) {
				unsigned y = mt[i];
				y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
				rd[carry + b * 624 + i] = (unsigned short)(y % (unsigned)m);
			}
		}
		const int N = carry + nb * 624;
		__syncthreads();
		// ---- next(i) for every start position; next(N) = END ----
		if (ns == 8) rs_next_table<8>(rd, Ja, N, ns, tid);
		else if (ns == 7) rs_next_table<7>(rd, Ja, N, ns, tid);
		else rs_next_table<0>(rd, Ja, N, ns, tid);
		if (tid == 0) { S[0] = 0; s_cnt = 0; }
		int maxS = N / ns + 2; maxS = maxS < RS_SMAX ? maxS : RS_SMAX;
		unsigned short* J = Ja; unsigned short* Jn = Jb;
		for (int known = 1; known < maxS; known <<= 1) {
			__syncthreads();
			for (int k = tid; k < known && k + known < maxS; k += RS_T) { const unsigned short sk = S[k]; S[k + known] = sk == RS_END ? RS_END : J[sk]; }
			if ((known << 1) < maxS) {
#pragma unroll 4
				for (int i = tid; i <= N; i += RS_T) { const unsigned short x = J[i]; Jn[i] = x == RS_END ? RS_END : J[x]; }
			}
			unsigned short* tsw = J; J = Jn; Jn = tsw;
		}
		__syncthreads();
		// ---- complete samples of this chunk: k with a valid S[k + 1] (S is increasing, then END) ----
		{
			int c = 0;
			for (int k = tid; k + 1 < maxS; k += RS_T) c += S[k + 1] != RS_END ? 1 : 0;
			if (c) atomicAdd(&s_cnt, c);
		}
		__syncthreads();
		const int ncomp = s_cnt;
		const int nemit = ncomp < iters - kdone ? ncomp : iters - kdone;
		for (int k = tid; k < nemit; k += RS_T) {      // re-walk hypothesis k from its start: its ns distinct draws in order
			int v[8];
			if (m <= 64) {
#pragma unroll
				for (int q = 0; q < 8; ++q) v[q] = -1;
				rs_walk_mask<true>(rd, S[k], N, ns, v);
			} else rs_walk(rd, S[k], N, ns, v);
			unsigned short* o = sp + (long long)(kdone + k) * 8;
#pragma unroll
			for (int q = 0; q < 8; ++q) if (q < ns) o[q] = (unsigned short)v[q];
		}
		kdone += ncomp;
		if (kdone >= iters) break;
		if (ncomp > 0) draws_per_sample_x16 = 16LL * S[ncomp] / ncomp + 1;
		// ---- carry the draws of the first incomplete sample to the front ----
		const int p0 = S[ncomp];                       // valid: next of the last complete sample (0 if none)
		const int L = N - p0;
		if (ncomp == 0 && L >= RS_N - 623) break;      // no progress possible (a sample longer than the whole buffer)
		for (int base = 0; base < L; base += RS_T) {
			__syncthreads();
			const unsigned short val = base + tid < L ? rd[p0 + base + tid] : (unsigned short)0;
			__syncthreads();
			if (base + tid < L) rd[base + tid] = val;
		}
		carry = L;
	}
}

// first hypothesis with the maximal inlier count (update_max, transform_estimate.cc:82)
__global__ void __launch_bounds__(256) k_ransac_best(const int* __restrict__ counts, int iters, int2* __restrict__ best,
		const PairArgs* __restrict__ pairs, const unsigned short* __restrict__ samples, unsigned short* __restrict__ best_samp, const int* __restrict__ active) {
	__shared__ int s_cnt[256], s_idx[256];
	const int pair = active[blockIdx.x];
	const int* c = counts + (long long)blockIdx.x * iters;
	int bc = -1, bi = -1;
	for (int i = threadIdx.x; i < iters; i += 256) { const int v = c[i]; if (v > bc) { bc = v; bi = i; } }
	s_cnt[threadIdx.x] = bc; s_idx[threadIdx.x] = bi;
	__syncthreads();
	for (int st = 128; st > 0; st >>= 1) {
		if (threadIdx.x < st) {
			const int oc = s_cnt[threadIdx.x + st], oi = s_idx[threadIdx.x + st];
			if (oc > s_cnt[threadIdx.x] || (oc == s_cnt[threadIdx.x] && oc >= 0 && oi < s_idx[threadIdx.x])) { s_cnt[threadIdx.x] = oc; s_idx[threadIdx.x] = oi; }
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) best[blockIdx.x] = make_int2(s_idx[0], s_cnt[0]);          // results are stored per live slot
	if (threadIdx.x < 8) {     // the winner's sample, for the host epilogue
		const int bi0 = s_idx[0];
		best_samp[blockIdx.x * 8 + threadIdx.x] = bi0 >= 0 ? samples[pairs[pair].samp_off + (long long)bi0 * 8 + threadIdx.x] : (unsigned short)0;
	}
}

// ---------------- host epilogue: fill_inliers_to_matchinfo and its helpers ----------------
struct Shape { int w, h; };
inline bool shifted_in(const Shape& s, P2 p) {        // match_info.hh:68-70
	return p.x >= -s.w * 0.5 && p.x < s.w * 0.5 && p.y >= -s.h * 0.5 && p.y < s.h * 0.5;
}
inline double side(P2 a, P2 b, P2 p) { return (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x); }   // polygon.cc:9-11

std::vector<P2> convex_hull(std::vector<P2>& pts) {  // lib/polygon.cc:17-46
	if (pts.size() <= 3) return pts;
	std::sort(pts.begin(), pts.end(), [](const P2& a, const P2& b) { if (a.y == b.y) return a.x < b.x; return a.y < b.y; });
	std::vector<P2> ret;
	ret.push_back(pts[0]); ret.push_back(pts[1]);
	const int n = (int)pts.size();
	for (int i = 2; i < n; ++i) {
		while (ret.size() >= 2 && side(ret[ret.size() - 2], ret.back(), pts[i]) <= 0) ret.pop_back();
		ret.push_back(pts[i]);
	}
	const size_t mid = ret.size();
	ret.push_back(pts[n - 2]);
	for (int i = n - 3; i >= 0; --i) {
		while (ret.size() > mid && side(ret[ret.size() - 2], ret.back(), pts[i]) <= 0) ret.pop_back();
		ret.push_back(pts[i]);
	}
	return ret;
}

double polygon_area(const std::vector<P2>& poly) {   // lib/polygon.cc:48-60
	const int n = (int)poly.size();
	double sum = 0;
	for (int i = 0; i < n; ++i) sum += poly[i].x * (poly[(i + 1) % n].y - poly[(i + n - 1) % n].y);
	return 0.5 * std::fabs(sum);
}

struct PointInPolygon {                               // lib/polygon.hh:30-52, polygon.cc:62-82
	const std::vector<P2>& poly; P2 com; std::vector<std::pair<float, int>> slopes;
	explicit PointInPolygon(const std::vector<P2>& p): poly(p) {
		com = P2{0, 0};
		for (auto& c : poly) { com.x += c.x; com.y += c.y; }
		const double f = 1.0 / poly.size();
		com.x *= f; com.y *= f;
		for (size_t i = 0; i < p.size(); ++i) slopes.emplace_back((float)std::atan2(p[i].y - com.y, p[i].x - com.x), (int)i);
		std::sort(slopes.begin(), slopes.end());
	}
	bool in_polygon(P2 p) const {
		const float k = (float)std::atan2(p.y - com.y, p.x - com.x);
		auto itr = std::lower_bound(slopes.begin(), slopes.end(), std::make_pair(k, 0));
		int idx1, idx2;
		if (itr == slopes.end()) { idx1 = slopes.back().second; idx2 = slopes.front().second; }
		else { idx2 = itr->second; idx1 = (itr != slopes.begin()) ? (--itr)->second : slopes.back().second; }
		const P2 p1 = poly[idx1], p2 = poly[idx2];
		const double o1 = side(p1, p2, com), o2 = side(p1, p2, p);
		return !(o1 * o2 < -1e-6);
	}
};

inline P2 trans2d(const double (&H)[9], P2 m) {       // homography.hh:53-76
	const double x = H[0] * m.x + H[1] * m.y + H[2] * 1.0, y = H[3] * m.x + H[4] * m.y + H[5] * 1.0, z = H[6] * m.x + H[7] * m.y + H[8] * 1.0;
	const double d = 1.0 / z;
	return P2{x * d, y * d};
}

// 3x3 inverse with complete pivoting (Homography::inverse, stitch/homography.cc:25-39)
bool inverse3(const double (&a)[9], double (&inv)[9]) {
	double lu[9]; std::memcpy(lu, a, sizeof(lu));
	int rowt[3], colt[3], nonzero = 3; double maxpivot = 0;
	for (int k = 0; k < 3; ++k) {
		int br = k, bc = k; double best = -1;
		for (int i = k; i < 3; ++i) for (int j = k; j < 3; ++j) { double v = std::fabs(lu[i * 3 + j]); if (v > best) { best = v; br = i; bc = j; } }
		if (best == 0.0) { nonzero = k; for (int i = k; i < 3; ++i) rowt[i] = colt[i] = i; break; }
		if (best > maxpivot) maxpivot = best;
		rowt[k] = br; colt[k] = bc;
		if (br != k) for (int j = 0; j < 3; ++j) std::swap(lu[k * 3 + j], lu[br * 3 + j]);
		if (bc != k) for (int i = 0; i < 3; ++i) std::swap(lu[i * 3 + k], lu[i * 3 + bc]);
		for (int i = k + 1; i < 3; ++i) lu[i * 3 + k] /= lu[k * 3 + k];
		for (int i = k + 1; i < 3; ++i) for (int j = k + 1; j < 3; ++j) lu[i * 3 + j] -= lu[i * 3 + k] * lu[k * 3 + j];
	}
	const double thr = std::fabs(maxpivot) * (2.220446049250313e-16 * 3);
	int rank = 0;
	for (int i = 0; i < nonzero; ++i) rank += (std::fabs(lu[i * 3 + i]) > thr);
	if (rank != 3) return false;
	for (int col = 0; col < 3; ++col) {
		double c[3];
		for (int i = 0; i < 3; ++i) c[i] = (i == col) ? 1.0 : 0.0;
		for (int i = 0; i < 3; ++i) std::swap(c[i], c[rowt[i]]);
		for (int i = 0; i < 3; ++i) for (int j = 0; j < i; ++j) c[i] -= lu[i * 3 + j] * c[j];
		for (int i = 2; i >= 0; --i) { for (int j = i + 1; j < 3; ++j) c[i] -= lu[i * 3 + j] * c[j]; c[i] /= lu[i * 3 + i]; }
		for (int i = 2; i >= 0; --i) std::swap(c[i], c[colt[i]]);
		for (int i = 0; i < 3; ++i) inv[i * 3 + col] = c[i];
	}
	return true;
}

// overlap_region (stitch/homography.cc:50-90): homo maps shape2 -> shape1, inv the reverse
std::vector<P2> overlap_region(const Shape& shape1, const Shape& shape2, const double (&homo)[9], const double (&inv)[9]) {
	const int NR = 100;
	const float stepw = (float)(shape2.w * 1.0 / NR), steph = (float)(shape2.h * 1.0 / NR);
	const double hw = shape2.w * 0.5, hh = shape2.h * 0.5;
	std::vector<P2> pts2in1;
	for (int i = 0; i < NR; ++i) {
		const P2 e[4] = { P2{-hw + i * stepw, -hh}, P2{-hw + i * stepw, hh}, P2{-hw, -hh + i * steph}, P2{hw, -hh + i * steph} };
		for (int k = 0; k < 4; ++k) {
			// Matrix product 3x3 * 3x(4 NR) then float denom = 1.0 / z (:72-76)
			const double x = homo[0] * e[k].x + homo[1] * e[k].y + homo[2] * 1.0;
			const double y = homo[3] * e[k].x + homo[4] * e[k].y + homo[5] * 1.0;
			const double z = homo[6] * e[k].x + homo[7] * e[k].y + homo[8] * 1.0;
			const float denom = (float)(1.0 / z);
			const P2 pin1{x * denom, y * denom};
			if (shifted_in(shape1, pin1)) pts2in1.push_back(pin1);
		}
	}
	const P2 corners[4] = { P2{-shape1.w * 0.5, -shape1.h * 0.5}, P2{shape1.w * 0.5, -shape1.h * 0.5}, P2{-shape1.w * 0.5, shape1.h * 0.5}, P2{shape1.w * 0.5, shape1.h * 0.5} };
	for (auto& c : corners) if (shifted_in(shape2, trans2d(inv, c))) pts2in1.push_back(c);
	return convex_hull(pts2in1);
}

struct PairHost {
	int i, j, m, slot;             // slot: index among the live pairs (-1: below the match-count gate)
	const double* kp1; int nk1;    // image i keypoints (x, y) centred
	const double* kp2; int nk2;
	Shape s1, s2;
	const double* pts;             // m x 4 (p1x, p1y, p2x, p2y), gathered on the device
	double inlier_dist;
};

}	// namespace

extern "C" {

int op_ransac_pairs(op_ctx* ctx, const op_config* cfg, const op_features* f, const op_matches* mt,
		const int* pairs, int npairs, const int* shapes_wh, const uint32_t* seeds, uint32_t base_seed,
		op_ransac_result** out) {
	if (!ctx || !cfg || !f || !mt || (!pairs && npairs != 0) || npairs < 0 || !shapes_wh || !out) OP_FAIL(OP_ERR_INVALID, "op_ransac_pairs: bad argument");
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t st = ctx->stream;
	const FeatView fv = op_features_view(f);
	if (fv.device != ctx->device) OP_FAIL(OP_ERR_INVALID, "op_ransac_pairs: the features live on another device than the context");
	std::unique_ptr<op_ransac_result> R(new op_ransac_result);
	R->items.resize(npairs);
	R->pairs_hash = pair_list_hash(pairs, npairs);
	if (npairs == 0) { *out = R.release(); return OP_OK; }
	if (op_matches_num_pairs(mt) != npairs) OP_FAIL(OP_ERR_INVALID, "op_ransac_pairs: op_matches holds a different number of pairs than the pair list");
	const bool affine = cfg->CYLINDER || cfg->TRANS;                    // transform_estimate.cc:34-37
	const int nsample = (affine ? 6 : 8) / 2 + 4;                       // :53
	const int iters = cfg->RANSAC_ITERATIONS;
	if (iters <= 0 || iters > 65536) OP_FAIL(OP_ERR_UNSUPPORTED, "RANSAC_ITERATIONS must be in [1, 65536]");

	std::unique_ptr<HostScope> hs(new HostScope(ctx, "ransac upload + launch (host)"));
	const std::vector<int>& mcount = op_matches_counts(mt);
	const std::vector<int64_t>& moffset = op_matches_offsets(mt);
	const std::vector<int>& mlim = op_matches_limits(mt);
	const long long mtotal = moffset[npairs];
	// The match lists index keypoints of f.  Lists made by op_match_pairs carry the keypoint counts they were made
	// for; lists wrapped from host arrays (op_matches_from_host) are checked entry by entry: an index outside its
	// image would be a read out of bounds on the device and in the epilogue below.
	const int* h_lists = mlim.empty() ? op_matches_host(mt) : nullptr;
	if (mlim.empty() && !h_lists) return OP_ERR_HIP;

	// pass 1: what the kernels need per pair -- counts and offsets only, no coordinates, no lists
	std::vector<PairHost> ph(npairs);
	std::vector<int> h_active;
	long long pts_total = 0; int max_m = 0;
	for (int p = 0; p < npairs; ++p) {
		const int i = pairs[2 * p], j = pairs[2 * p + 1];
		if (i < 0 || j < 0 || i >= fv.n || j >= fv.n) OP_FAIL(OP_ERR_INVALID, "op_ransac_pairs: image index out of range");
		PairHost& h = ph[p];
		h.i = i; h.j = j; h.m = mcount[p]; h.slot = -1; h.pts = nullptr;
		h.nk1 = fv.counts[i]; h.nk2 = fv.counts[j];
		h.s1 = Shape{shapes_wh[2 * i], shapes_wh[2 * i + 1]}; h.s2 = Shape{shapes_wh[2 * j], shapes_wh[2 * j + 1]};
		if (h.m > 65535) OP_FAIL(OP_ERR_CAPACITY, "more than 65535 matches in one pair");
		if (!mlim.empty()) {
			if (mlim[2 * (size_t)p] > h.nk1 || mlim[2 * (size_t)p + 1] > h.nk2)
				OP_FAIL(OP_ERR_INVALID, "op_ransac_pairs: pair " + std::to_string(p) + " was matched on images with more keypoints than f holds for it");
		} else {
			const int* q = h_lists + 2 * moffset[p];
			for (int k = 0; k < h.m; ++k)
				if ((unsigned)q[2 * k] >= (unsigned)h.nk1 || (unsigned)q[2 * k + 1] >= (unsigned)h.nk2)
					OP_FAIL(OP_ERR_INVALID, "op_ransac_pairs: pair " + std::to_string(p) + " match " + std::to_string(k) + " indexes a keypoint outside its image");
		}
		// ransac_inlier_thres (float) and INLIER_DIST = sqr(float) (transform_estimate.cc:46,133)
		const float thres = (float)((h.s1.w + h.s1.h) * 0.5 / 800 * cfg->RANSAC_INLIER_THRES);
		h.inlier_dist = (double)(thres * thres);
		if (h.m >= 8 && h.m >= nsample) {                                  // ESTIMATE_MIN_NR_MATCH (:21,39) / :55: otherwise get_transform -> false
			h.slot = (int)h_active.size(); h_active.push_back(p); pts_total += h.m; max_m = std::max(max_m, h.m);
		}
	}
	const int nactive = (int)h_active.size();
	if (nactive == 0) { *out = R.release(); return OP_OK; }

	// One device arena (the context's grow-only RANSAC scratch) and one pinned block: [PairArgs | seeds | live list |
	// the match lists if they only exist on the host] go up in one copy, [winner | its sample | gathered points] come
	// back in one copy.  Samples, generator states and hypothesis counts exist per LIVE pair only.
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const int* d_midx_resident = op_matches_device(mt, ctx->device, st);
	const bool upload_lists = !d_midx_resident && mtotal > 0;
	const size_t u_pa = 0, u_seeds = al(u_pa + sizeof(PairArgs) * npairs), u_active = al(u_seeds + sizeof(unsigned) * npairs),
			u_lists = al(u_active + sizeof(int) * nactive), up_bytes = al(u_lists + (upload_lists ? sizeof(int) * 2 * (size_t)mtotal : 0));
	const size_t r_best = 0, r_bsamp = al(r_best + sizeof(int2) * nactive), r_pts = al(r_bsamp + sizeof(unsigned short) * 8 * nactive),
			down_bytes = al(r_pts + sizeof(double) * 4 * (size_t)pts_total);
	const size_t o_up = 0, o_down = al(o_up + up_bytes), o_samp = al(o_down + down_bytes),
			o_state = al(o_samp + sizeof(unsigned short) * 8 * (size_t)nactive * iters),
			o_counts = al(o_state + sizeof(unsigned) * 624 * (size_t)nactive),
			arena_bytes = al(o_counts + sizeof(int) * (size_t)nactive * iters);
	char* pin = (char*)ctx->pinned_scratch(up_bytes + down_bytes);
	if (!pin) OP_FAIL(OP_ERR_HIP, "op_ransac_pairs: pinned host allocation failed");
	{
		PairArgs* pa = (PairArgs*)(pin + u_pa);
		unsigned* h_seeds = (unsigned*)(pin + u_seeds);
		long long pts_off = 0;
		for (int p = 0; p < npairs; ++p) {
			const PairHost& h = ph[p];
			pa[p] = PairArgs{(int)pts_off, h.m, affine ? 1 : 0, nsample, h.inlier_dist, (long long)std::max(h.slot, 0) * iters * 8,
					(long long)moffset[p], (int)fv.offsets[h.i], (int)fv.offsets[h.j]};
			if (h.slot >= 0) pts_off += h.m;
			// per-pair seeds; the draw sequence itself is generated on the device (k_ransac_samples)
			h_seeds[p] = seeds ? seeds[p] : (base_seed * 2654435761u) ^ (uint32_t)(p * 40503u + 12345u);
		}
		std::memcpy(pin + u_active, h_active.data(), sizeof(int) * nactive);
		if (upload_lists) {
			const int* hl = op_matches_host(mt);
			if (!hl) return OP_ERR_HIP;
			std::memcpy(pin + u_lists, hl, sizeof(int) * 2 * (size_t)mtotal);
		}
	}
	const char* down = pin + up_bytes;
	const int2* best = (const int2*)(down + r_best);
	const unsigned short* best_samp = (const unsigned short*)(down + r_bsamp);
	const double* coor_host = nullptr;
	int rc = OP_OK;
#define RCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { op_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); rc = OP_ERR_HIP; goto done; } } while (0)
	{
		RCHK(ctx->ransac_arena.ensure(arena_bytes));
		char* arena = (char*)ctx->ransac_arena.p;
		const PairArgs* d_pa = (const PairArgs*)(arena + o_up + u_pa);
		const unsigned* d_seeds = (const unsigned*)(arena + o_up + u_seeds);
		const int* d_active = (const int*)(arena + o_up + u_active);
		const int2* d_midx = upload_lists ? (const int2*)(arena + o_up + u_lists) : (const int2*)d_midx_resident;
		int2* d_best = (int2*)(arena + o_down + r_best);
		unsigned short* d_bsamp = (unsigned short*)(arena + o_down + r_bsamp);
		double* d_pts = (double*)(arena + o_down + r_pts);
		unsigned short* d_samp = (unsigned short*)(arena + o_samp);
		unsigned* d_state = (unsigned*)(arena + o_state);
		int* d_counts = (int*)(arena + o_counts);
		RCHK(hipMemcpyAsync(arena + o_up, pin, up_bytes, hipMemcpyHostToDevice, st));
		{
			ProfScope ps2(ctx, "ransac mt19937 samples");
			// pairs below the match-count gate never get a workgroup: launching groups that exit at once costs more
			// dispatcher time than the live ones compute (config 4: 703 pairs, ~640 live)
			hipLaunchKernelGGL(k_ransac_seed, dim3((nactive + 63) / 64), dim3(64), 0, st, d_seeds, d_active, nactive, d_state);
			RCHK(hipGetLastError());
			hipLaunchKernelGGL(k_ransac_samples, dim3(nactive), dim3(RS_T), 0, st, d_pa, (const unsigned*)d_state, iters, d_samp, d_active);
			RCHK(hipGetLastError());
		}
		{
			ProfScope ps(ctx, "ransac hypotheses");
			hipLaunchKernelGGL(k_ransac_gather, dim3((max_m + 255) / 256, nactive), dim3(256), 0, st, d_pa, d_active, d_midx,
					(const double2*)op_features_coor_device(f), d_pts);
			RCHK(hipGetLastError());
			hipLaunchKernelGGL(k_ransac_hyp, dim3((iters + 255) / 256, nactive), dim3(256), 0, st, d_pa, (const double*)d_pts, (const unsigned short*)d_samp, iters, d_counts, d_active);
			RCHK(hipGetLastError());
			hipLaunchKernelGGL(k_ransac_best, dim3(nactive), dim3(256), 0, st, (const int*)d_counts, iters, d_best, d_pa, (const unsigned short*)d_samp, d_bsamp, d_active);
			RCHK(hipGetLastError());
		}
		RCHK(hipMemcpyAsync(pin + up_bytes, arena + o_down, down_bytes, hipMemcpyDeviceToHost, st));
		// the acceptance gates count the keypoints of both images inside the overlap polygon (:191-199): every
		// coordinate of the job, fetched once per op_features (the first call waits for it; later ones find it)
		coor_host = op_features_coor_host(f, ctx);
		if (!coor_host) { rc = OP_ERR_HIP; goto done; }
		RCHK(hipStreamSynchronize(st));
	}
	resolve_profile(ctx);
	hs.reset(); hs.reset(new HostScope(ctx, "ransac acceptance epilogue (host)"));
	{
		const PairArgs* pa = (const PairArgs*)(pin + u_pa);
		for (int p = 0; p < npairs; ++p) {
			PairHost& h = ph[p];
			h.kp1 = coor_host + fv.offsets[h.i] * 2; h.kp2 = coor_host + fv.offsets[h.j] * 2;
			if (h.slot >= 0) h.pts = (const double*)(down + r_pts) + (size_t)pa[p].pts_off * 4;
		}
	}

	// ---- host epilogue per pair (transform_estimate.cc:85-86, 150-218) ----
	host_parallel_for(npairs, [&](int p) {
		op_ransac_result::Item& it = R->items[p];
		const PairHost& h = ph[p];
		if (h.slot < 0) return;                                                       // get_transform -> false (:55)
		it.best_hyp = best[h.slot].x; it.best_count = best[h.slot].y;
		if (it.best_hyp < 0 || it.best_count < 0) return;
		const unsigned short* sp = best_samp + (size_t)h.slot * 8;
		const double* P = h.pts;
		double Hb[9];
		opransac::calc_transform(nsample, [&](int q) { return P2{P[4 * sp[q]], P[4 * sp[q] + 1]}; },
				[&](int q) { return P2{P[4 * sp[q] + 2], P[4 * sp[q] + 3]}; }, affine, Hb);
		const double inlier_dist = h.inlier_dist;
		std::vector<int> inl;
		for (int k = 0; k < h.m; ++k)
			if (opransac::is_inlier(Hb, P2{P[4 * k], P[4 * k + 1]}, P2{P[4 * k + 2], P[4 * k + 3]}, inlier_dist)) inl.push_back(k);
		it.confidence = -(float)inl.size();                                           // :153
		it.inliers = inl;
		if (inl.size() < 8) return;                                                 // :154
		double homo[9], inv[9];
		opransac::calc_transform((int)inl.size(), [&](int q) { return P2{P[4 * inl[q]], P[4 * inl[q] + 1]}; },
				[&](int q) { return P2{P[4 * inl[q] + 2], P[4 * inl[q] + 3]}; }, affine, homo);   // :179
		if (!inverse3(homo, inv)) return;                                           // :182-184
		auto match_cnt = [&](const std::vector<P2>& poly, bool first) {
			if (poly.size() < 3) return 0;
			PointInPolygon pip(poly);
			int c = 0;
			for (int k = 0; k < h.m; ++k) c += pip.in_polygon(first ? P2{P[4 * k], P[4 * k + 1]} : P2{P[4 * k + 2], P[4 * k + 3]}) ? 1 : 0;
			return c;
		};
		auto keypoint_cnt = [&](const std::vector<P2>& poly, bool first, bool& valid) {
			valid = poly.size() >= 3;          // the reference asserts here (polygon.hh:32)
			if (!valid) return 0;
			PointInPolygon pip(poly);
			const double* kp = first ? h.kp1 : h.kp2; const int nk = first ? h.nk1 : h.nk2;
			int c = 0;
			for (int k = 0; k < nk; ++k) c += pip.in_polygon(P2{kp[2 * k], kp[2 * k + 1]}) ? 1 : 0;
			return c;
		};
		bool valid = true;
		std::vector<P2> overlap = overlap_region(h.s1, h.s2, homo, inv);
		const float r1m = inl.size() * 1.0f / match_cnt(overlap, true);
		if (r1m < cfg->INLIER_IN_MATCH_RATIO) return;
		const float r1p = inl.size() * 1.0f / keypoint_cnt(overlap, true, valid);
		if (!valid || r1p < 0.01 || r1p > 1) return;
		overlap = overlap_region(h.s2, h.s1, inv, homo);
		const float r2m = inl.size() * 1.0f / match_cnt(overlap, false);
		if (r2m < cfg->INLIER_IN_MATCH_RATIO) return;
		const float r2p = inl.size() * 1.0f / keypoint_cnt(overlap, false, valid);
		if (!valid || r2p < 0.01 || r2p > 1) return;
		it.confidence = (float)((r1p + r2p) * 0.5);                                   // :200
		if (it.confidence < cfg->INLIER_IN_POINTS_RATIO) return;
		const double area = polygon_area(overlap);
		const double area1 = (double)(h.s1.w * h.s1.h), area2 = (double)(h.s2.w * h.s2.h);
		if (area / std::max(area1, area2) < 0.15) return;
		std::memcpy(it.homo, homo, sizeof(homo));
		it.ok = 1;
	});
done:
	hs.reset();
#undef RCHK
	if (rc != OP_OK) return rc;
	*out = R.release();
	return OP_OK;
}

int op_ransac_ok(const op_ransac_result* r, int p) { return (r && p >= 0 && p < (int)r->items.size()) ? r->items[p].ok : 0; }
float op_ransac_confidence(const op_ransac_result* r, int p) { return (r && p >= 0 && p < (int)r->items.size()) ? r->items[p].confidence : 0.f; }
int op_ransac_homo(const op_ransac_result* r, int p, double* h9) {
	if (!r || p < 0 || p >= (int)r->items.size() || !h9) OP_FAIL(OP_ERR_INVALID, "op_ransac_homo: bad argument");
	std::memcpy(h9, r->items[p].homo, sizeof(double) * 9); return OP_OK;
}
int op_ransac_inlier_count(const op_ransac_result* r, int p) { return (r && p >= 0 && p < (int)r->items.size()) ? (int)r->items[p].inliers.size() : 0; }
int op_ransac_inliers(const op_ransac_result* r, int p, int* match_indices) {
	if (!r || p < 0 || p >= (int)r->items.size() || !match_indices) OP_FAIL(OP_ERR_INVALID, "op_ransac_inliers: bad argument");
	std::copy(r->items[p].inliers.begin(), r->items[p].inliers.end(), match_indices); return OP_OK;
}
int op_ransac_best(const op_ransac_result* r, int p, int* hyp, int* count) {
	if (!r || p < 0 || p >= (int)r->items.size()) OP_FAIL(OP_ERR_INVALID, "op_ransac_best: bad argument");
	if (hyp) *hyp = r->items[p].best_hyp; if (count) *count = r->items[p].best_count; return OP_OK;
}
int op_ransac_summary(const op_ransac_result* r, int* accepted_pairs, int64_t* inliers) {
	if (!r) OP_FAIL(OP_ERR_INVALID, "op_ransac_summary: bad argument");
	int ok = 0; int64_t inl = 0;
	for (auto& it : r->items) if (it.ok) { ++ok; inl += (int64_t)it.inliers.size(); }
	if (accepted_pairs) *accepted_pairs = ok; if (inliers) *inliers = inl;
	return OP_OK;
}
void op_ransac_free(op_ransac_result* r) { delete r; }

// ---- Stitcher::match_image's bookkeeping for the whole job (stitch/stitcher.cc:79-93) ----
// Every accepted pair (i, j) fills pairwise_matches[i][j] = info and pairwise_matches[j][i] = the same with the inverse
// homography scaled by 1 / inv[8] and every match reversed (match_info.hh:21-25).  The table comes out in the flat form
// pano_estimate_cameras (include/pano_host.h) takes: per directed entry (i, j): confidence, homo (j -> i), the number of
// inlier matches and the matched points (point in i, point in j) back to back.
int op_pairwise_table_size(const op_ransac_result* r, int* entries, int64_t* points) {
	if (!r || !entries || !points) OP_FAIL(OP_ERR_INVALID, "op_pairwise_table_size: bad argument");
	int e = 0; int64_t pt = 0;
	for (auto& it : r->items) if (it.ok) { e += 2; pt += 2 * (int64_t)it.inliers.size(); }
	*entries = e; *points = pt;
	return OP_OK;
}
int op_pairwise_table(op_ctx* ctx, const op_features* f, const op_matches* m, const op_ransac_result* r, const int* pairs, int npairs,
		int* ij, float* conf, double* homo, int* cnt, double* pts) {
	if (!ctx || !f || !m || !r || !pairs || npairs != (int)r->items.size() || npairs != op_matches_num_pairs(m) || !ij || !conf || !homo || !cnt || !pts)
		OP_FAIL(OP_ERR_INVALID, "op_pairwise_table: bad argument");
	if (r->pairs_hash && r->pairs_hash != pair_list_hash(pairs, npairs))
		OP_FAIL(OP_ERR_INVALID, "op_pairwise_table: `pairs` is not the pair list the RANSAC result was computed for");
	HIPCHK(hipSetDevice(ctx->device));
	const FeatView fv = op_features_view(f);
	const double* coor = op_features_coor_host(f, ctx);
	const int* lists = op_matches_host(m);
	if (!coor || !lists) return OP_ERR_HIP;
	const std::vector<int64_t>& moff = op_matches_offsets(m);
	const std::vector<int>& mcnt = op_matches_counts(m);
	int e = 0; int64_t at = 0;
	for (int p = 0; p < npairs; ++p) {
		const op_ransac_result::Item& it = r->items[p];
		if (!it.ok) continue;
		const int i = pairs[2 * p], j = pairs[2 * p + 1];
		if (i < 0 || j < 0 || i >= fv.n || j >= fv.n) OP_FAIL(OP_ERR_INVALID, "op_pairwise_table: image index out of range");
		double h[9], inv[9];
		std::memcpy(h, it.homo, sizeof(h));
		if (!inverse3(h, inv)) OP_FAIL(OP_ERR_INVALID, "op_pairwise_table: accepted homography is singular");   // cannot happen: the acceptance epilogue inverted it
		const double s = 1.0 / inv[8];                                                   // inv.mult(1.0 / inv[8]), stitcher.cc:80
		for (int k = 0; k < 9; ++k) inv[k] *= s;
		const int ni = (int)it.inliers.size();
		const double* ki = coor + fv.offsets[i] * 2; const double* kj = coor + fv.offsets[j] * 2;
		const int* lp = lists + 2 * moff[p];
		ij[2 * e] = i; ij[2 * e + 1] = j; ij[2 * e + 2] = j; ij[2 * e + 3] = i;
		conf[e] = it.confidence; conf[e + 1] = it.confidence;
		std::memcpy(homo + 9 * (size_t)e, h, sizeof(h)); std::memcpy(homo + 9 * (size_t)(e + 1), inv, sizeof(inv));
		cnt[e] = ni; cnt[e + 1] = ni;
		double* a = pts + 4 * at; double* b = a + 4 * (size_t)ni;
		for (int q = 0; q < ni; ++q) {
			const int k = it.inliers[q];
			if (k < 0 || k >= mcnt[p]) OP_FAIL(OP_ERR_INVALID, "op_pairwise_table: inlier index outside the pair's match list");
			const int fi = lp[2 * k], se = lp[2 * k + 1];
			if (fi < 0 || se < 0 || fi >= fv.counts[i] || se >= fv.counts[j]) OP_FAIL(OP_ERR_INVALID, "op_pairwise_table: match index outside the image's keypoints");
			a[4 * q] = ki[2 * fi]; a[4 * q + 1] = ki[2 * fi + 1]; a[4 * q + 2] = kj[2 * se]; a[4 * q + 3] = kj[2 * se + 1];
			b[4 * q] = kj[2 * se]; b[4 * q + 1] = kj[2 * se + 1]; b[4 * q + 2] = ki[2 * fi]; b[4 * q + 3] = ki[2 * fi + 1];
		}
		at += 2 * (int64_t)ni; e += 2;
	}
	return OP_OK;
}

}	// extern "C"
