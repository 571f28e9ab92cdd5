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
#include "ransac_accept.hpp"
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
using opaccept::Shape; using opaccept::PointInPolygon; using opaccept::overlap_region; using opaccept::polygon_area; using opaccept::inverse3;

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

// grid (ceil(iters/256), pairs of the launch's group); slot = slot_base + blockIdx.y is the pair's position in the live list.
// One lane per hypothesis: 8-point (7 for affine) sample -> DLT -> health() -> inlier count over the pair's matches (staged in
// LDS).  The winner -- the FIRST hypothesis with the maximal count (update_max, transform_estimate.cc:82) -- is chosen here as
// well: every workgroup folds its 256 hypotheses into one 64-bit key (count + 1) << 32 | ~hypothesis and takes an atomic
// maximum on the pair's word (zeroed by the pair's sample workgroup); the last of the pair's workgroups to finish (a counter)
// decodes it and copies the winner's sample for the host epilogue.  No per-hypothesis counts in HBM, no selection kernel.
__global__ void __launch_bounds__(256) k_ransac_hyp(const PairArgs* __restrict__ pairs, const double* __restrict__ pts /* m x {p1x,p1y,p2x,p2y} */,
		const unsigned short* __restrict__ samples, int iters, const int* __restrict__ active, int slot_base,
		unsigned long long* __restrict__ best64, int* __restrict__ done, int2* __restrict__ best, unsigned short* __restrict__ best_samp) {
	__shared__ double s_pts[RANSAC_PTS_CHUNK * 4];
	__shared__ unsigned long long s_key[4];
	__shared__ int s_win;
	const int slot = slot_base + blockIdx.y;
	const int pair = active[slot];
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
	// larger count wins, then the smaller hypothesis index; a hypothesis that failed health() (or lies past iters) is key 0
	unsigned long long key = (ok && hyp < iters) ? (((unsigned long long)(unsigned)(cnt + 1)) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)hyp) : 0ULL;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)key, off), hi = (unsigned)__shfl_xor((int)(unsigned)(key >> 32), off);
		const unsigned long long o = ((unsigned long long)hi << 32) | lo;
		key = o > key ? o : key;
	}
	if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = key;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long k = s_key[0];
		for (int w = 1; w < 4; ++w) k = s_key[w] > k ? s_key[w] : k;
		if (k) atomicMax(&best64[slot], k);
		__threadfence();                                       // the maximum is in place before this workgroup counts as done
		int win = -2;                                          // -2: not the pair's last workgroup
		if (atomicAdd(&done[slot], 1) == (int)gridDim.x - 1) {
			__threadfence();
			const unsigned long long w = atomicMax(&best64[slot], 0ULL);      // (an atomic read: the other workgroups' maxima were made in L2)
			win = w ? (int)(0xFFFFFFFFu - (unsigned)w) : -1;
			best[slot] = make_int2(win, w ? (int)(unsigned)(w >> 32) - 1 : -1);
		}
		s_win = win;
	}
	__syncthreads();
	const int win = s_win;
	if (win != -2 && threadIdx.x < 8)      // the winner's sample, for the host epilogue
		best_samp[slot * 8 + threadIdx.x] = win >= 0 ? samples[pa.samp_off + (long long)win * 8 + threadIdx.x] : (unsigned short)0;
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

// One workgroup per live pair.  Before the table it (a) gathers the pair's matched points where both inputs already are --
// the match lists op_match_pairs left in HBM and the keypoint coordinates of op_features (TransformEstimation's constructor
// arguments, transform_estimate.cc:26-33: match.data[k] -> kp1[first], kp2[second]) -- for the hypothesis kernel and the host
// epilogue, (b) clears the pair's winner word and finished-workgroups counter of the hypothesis kernel, and (c) seeds the
// generator: std::mt19937::seed is a serial recurrence over 624 words, wave-uniform, i.e. scalar-unit work of wavefront 0
// (~3 us) while the other wavefronts gather.  Rounds 1-5 ran (a) and (c) as two kernels of their own in front of this one.
__global__ void __launch_bounds__(RS_T) k_ransac_samples(const PairArgs* __restrict__ pairs, const unsigned* __restrict__ seeds,
		int iters, unsigned short* __restrict__ samples, const int* __restrict__ active, int slot_base,
		const int2* __restrict__ midx, const double2* __restrict__ coor, double* __restrict__ pts,
		unsigned long long* __restrict__ best64, int* __restrict__ done) {
	__shared__ unsigned mt[624];
	__shared__ unsigned short rd[RS_N + 8];
	__shared__ unsigned short Ja[RS_N + 2], Jb[RS_N + RS_W + 2];         // Jb doubles as the previous-equal table before the squaring starts
	__shared__ unsigned short S[RS_SMAX];
	__shared__ int s_cnt;
	const int slot = slot_base + blockIdx.x;
	const int pair = active[slot];                     // only pairs with enough matches get a workgroup (the host compacts the list)
	const PairArgs pa = pairs[pair];
	const int m = pa.m, ns = pa.nsample, tid = threadIdx.x;
	if (tid == 0) { best64[slot] = 0ULL; done[slot] = 0; }
	if (m < 8 || m < ns) return;                       // ESTIMATE_MIN_NR_MATCH (:21,39) / :55
	if (tid < 64) {
		unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)seeds[pair]);
		if (tid == 0) mt[0] = v;
		for (int i = 1; i < 624; ++i) {                 // wave-uniform chain: scalar registers; lane 0 stores
			v = 1812433253u * (v ^ (v >> 30)) + (unsigned)i;
			if (tid == 0) mt[i] = v;
		}
	} else {
		for (int k = tid - 64; k < m; k += RS_T - 64) {
			const int2 ab = midx[pa.moff + k];
			const double2 p1 = coor[pa.off_i + ab.x], p2 = coor[pa.off_j + ab.y];
			double2* o = (double2*)(pts + ((long long)pa.pts_off + k) * 4);
			o[0] = p1; o[1] = p2;
		}
	}
	__syncthreads();
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
	std::vector<int> h_active, order;
	long long pts_total = 0;
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
			h.slot = 0; h_active.push_back(p); pts_total += h.m;
		}
	}
	const int nactive = (int)h_active.size();
	if (nactive == 0) { *out = R.release(); return OP_OK; }
	// The live list in two groups, each with launches of its own on a stream of its own.  A sample is the first `nsample`
	// DISTINCT draws: at m = 8 a hypothesis consumes 21.7 draws on average, at m = 13 11.7, from m = 16 on about 10 -- the
	// sample table of a pair with few matches takes twice as long as the others' (and most candidate pairs of an unordered
	// job have few: median 13 on config 4).  With one launch per stage every pair's hypotheses waited for the slowest table;
	// now the hypotheses of the many-matches group run under the tail of the few-matches group's tables.
	const int small_m = 14;                                               // m < small_m: the few-matches group (second in the list)
	std::stable_partition(h_active.begin(), h_active.end(), [&](int p) { return ph[p].m >= small_m; });
	int n_large = 0;
	for (int q = 0; q < nactive; ++q) { ph[h_active[q]].slot = q; n_large += ph[h_active[q]].m >= small_m ? 1 : 0; }
	const int n_small = nactive - n_large;

	// One device arena (the context's grow-only RANSAC scratch) and one pinned block: [PairArgs | seeds | live list |
	// the match lists if they only exist on the host] go up in one copy, [winner | its sample | gathered points] come
	// back in one copy.  Sample tables and the winner words exist per LIVE pair only.
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const int* d_midx_resident = op_matches_device(mt, ctx->device, st);
	const bool upload_lists = !d_midx_resident && mtotal > 0;
	const size_t u_pa = 0, u_seeds = al(u_pa + sizeof(PairArgs) * npairs), u_active = al(u_seeds + sizeof(unsigned) * npairs),
			u_lists = al(u_active + sizeof(int) * nactive), up_bytes = al(u_lists + (upload_lists ? sizeof(int) * 2 * (size_t)mtotal : 0));
	const size_t r_best = 0, r_bsamp = al(r_best + sizeof(int2) * nactive), r_pts = al(r_bsamp + sizeof(unsigned short) * 8 * nactive),
			down_bytes = al(r_pts + sizeof(double) * 4 * (size_t)pts_total);
	const size_t o_up = 0, o_down = al(o_up + up_bytes), o_samp = al(o_down + down_bytes),
			o_best64 = al(o_samp + sizeof(unsigned short) * 8 * (size_t)nactive * iters),
			o_done = al(o_best64 + sizeof(unsigned long long) * (size_t)nactive),
			arena_bytes = al(o_done + sizeof(int) * (size_t)nactive);
	char* pin = (char*)ctx->pinned_scratch(up_bytes + down_bytes);
	if (!pin) OP_FAIL(OP_ERR_HIP, "op_ransac_pairs: pinned host allocation failed");
	{
		PairArgs* pa = (PairArgs*)(pin + u_pa);
		unsigned* h_seeds = (unsigned*)(pin + u_seeds);
		std::vector<long long> slot_pts(nactive + 1, 0);                   // gathered points lie in live-list order
		for (int q = 0; q < nactive; ++q) slot_pts[q + 1] = slot_pts[q] + ph[h_active[q]].m;
		for (int p = 0; p < npairs; ++p) {
			const PairHost& h = ph[p];
			pa[p] = PairArgs{h.slot >= 0 ? (int)slot_pts[h.slot] : 0, h.m, affine ? 1 : 0, nsample, h.inlier_dist, (long long)std::max(h.slot, 0) * iters * 8,
					(long long)moffset[p], (int)fv.offsets[h.i], (int)fv.offsets[h.j]};
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
		unsigned long long* d_best64 = (unsigned long long*)(arena + o_best64);
		int* d_done = (int*)(arena + o_done);
		const double2* d_coor = (const double2*)op_features_coor_device(f);
		const dim3 hyp_x((iters + 255) / 256);
		RCHK(hipMemcpyAsync(arena + o_up, pin, up_bytes, hipMemcpyHostToDevice, st));
		// pairs below the match-count gate never get a workgroup: launching groups that exit at once costs more
		// dispatcher time than the live ones compute (config 4: 703 pairs, ~640 live)
		auto launch_group = [&](hipStream_t s, int slot_base, int n) {
			hipLaunchKernelGGL(k_ransac_samples, dim3(n), dim3(RS_T), 0, s, d_pa, d_seeds, iters, d_samp, d_active, slot_base, d_midx, d_coor, d_pts, d_best64, d_done);
			hipLaunchKernelGGL(k_ransac_hyp, dim3(hyp_x.x, n), dim3(256), 0, s, d_pa, (const double*)d_pts, (const unsigned short*)d_samp, iters, d_active, slot_base,
					d_best64, d_done, d_best, d_bsamp);
			return hipGetLastError();
		};
		{
			ProfScope ps(ctx, "ransac kernels (both streams)");
			const bool fork = n_small > 0 && n_large > 0;
			if (fork) {
				RCHK(ctx->aux());
				RCHK(hipEventRecord(ctx->aux_fork, st));                     // the upload (and whatever made the inputs) is in front of both groups
				RCHK(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_fork, 0));
				RCHK(launch_group(ctx->aux_stream, n_large, n_small));       // the long tables first
				RCHK(hipEventRecord(ctx->aux_join, ctx->aux_stream));
				RCHK(launch_group(st, 0, n_large));
				RCHK(hipStreamWaitEvent(st, ctx->aux_join, 0));
			} else RCHK(launch_group(st, 0, nactive));
		}
		RCHK(hipMemcpyAsync(pin + up_bytes, arena + o_down, down_bytes, hipMemcpyDeviceToHost, st));
		// the acceptance gates count the keypoints of both images inside the overlap polygon (:191-199): every
		// coordinate of the job, fetched once per op_features (the first call waits for it; later ones find it)
		coor_host = op_features_coor_host(f, ctx);
		if (!coor_host) { rc = OP_ERR_HIP; goto done; }
		// what the epilogue needs besides the kernels' results is made while they run: where every pair's points and keypoints
		// will be, and the order the pairs are judged in (below)
		{
			const PairArgs* pa = (const PairArgs*)(pin + u_pa);
			for (int p = 0; p < npairs; ++p) {
				PairHost& h = ph[p];
				h.kp1 = coor_host + fv.offsets[h.i] * 2; h.kp2 = coor_host + fv.offsets[h.j] * 2;
				if (h.slot >= 0) h.pts = (const double*)(down + r_pts) + (size_t)pa[p].pts_off * 4;
			}
		}
		order = h_active;
		std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ph[a].m > ph[b].m; });
		RCHK(hipStreamSynchronize(st));
	}
	resolve_profile(ctx);
	hs.reset(); hs.reset(new HostScope(ctx, "ransac acceptance epilogue (host)"));

	// ---- host epilogue per pair (transform_estimate.cc:85-86, 150-218) ----
	// live pairs only (the others stay at get_transform -> false, :55), longest match lists first: a pair that passes the
	// first gates refits the homography on its inliers and counts the keypoints of both images against the overlap polygons
	// (~30 us), most pairs end after a few microseconds -- dealt in list order the long ones landed at the end of somebody's
	// queue (`order`, made above while the kernels ran).
	host_parallel_for(nactive, [&](int q) {
		const int p = order[q];
		op_ransac_result::Item& it = R->items[p];
		const PairHost& h = ph[p];
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
		opaccept::calc_transform_skewed((int)inl.size(), [&](int q) { return P2{P[4 * inl[q]], P[4 * inl[q] + 1]}; },
				[&](int q) { return P2{P[4 * inl[q] + 2], P[4 * inl[q] + 3]}; }, affine, homo);   // :179 (calc_transform's rotations, several rows in flight)
		if (!inverse3(homo, inv)) return;                                           // :182-184
		// the two counts of one overlap polygon (:186-199): the pair's matched points and every keypoint of that image inside it,
		// eight points per step (count_in_polygon, ransac_accept.hpp).  A polygon of fewer than three vertices: the match count
		// is 0 (the ratio infinite, the first gate passes) and the reference asserts at the keypoint count (polygon.hh:32)
		float r1p = 0, r2p = 0;
		auto gates = [&](const std::vector<P2>& poly, bool first) {
			int mc = 0, kc = 0;
			const bool valid = poly.size() >= 3;
			if (valid) {
				const PointInPolygon pip(poly);
				const opaccept::PolygonTables T(pip);
				mc = opaccept::count_in_polygon(T, P + (first ? 0 : 2), 4, h.m);
				const float rm = inl.size() * 1.0f / mc;
				if (rm < cfg->INLIER_IN_MATCH_RATIO) return false;
				kc = opaccept::count_in_polygon(T, first ? h.kp1 : h.kp2, 2, first ? h.nk1 : h.nk2);
			}
			const float rp = inl.size() * 1.0f / kc;
			if (!valid || rp < 0.01 || rp > 1) return false;
			(first ? r1p : r2p) = rp;
			return true;
		};
		std::vector<P2> overlap = overlap_region(h.s1, h.s2, homo, inv);
		if (!gates(overlap, true)) return;
		overlap = overlap_region(h.s2, h.s1, inv, homo);
		if (!gates(overlap, false)) return;
		it.confidence = (float)((r1p + r2p) * 0.5);                                   // :200
		if (it.confidence < cfg->INLIER_IN_POINTS_RATIO) return;
		const double area = polygon_area(overlap);
		const double area1 = (double)(h.s1.w * h.s1.h), area2 = (double)(h.s2.w * h.s2.h);
		if (area / std::max(area1, area2) < 0.15) return;
		std::memcpy(it.homo, homo, sizeof(homo));
		it.ok = 1;
	}, 4);
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
