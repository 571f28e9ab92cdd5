/* oracle/match_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the reference's exact brute-force matcher, FeatureMatcher::match
 * (feature/matcher.cc:15-71) and of the SSE squared-L2 it is built on
 * (feature/dist.cc:22-57, the variant every -march=native x86 build selects).
 */
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include "oracle.h"

/* feature/dist.cc:22-57: four stride-4 partial sums (one per SSE lane), horizontal add
 * (v0+v1)+(v2+v3); early-out check after elements 4, 36, 68, 100 (n % 32 == 0 with n counting
 * down from 128) returning FLT_MAX when the partial already exceeds now_thres. */
float orc_euclidean_sqr(const float* x, const float* y, int n, float now_thres) {
	float v0 = 0, v1 = 0, v2 = 0, v3 = 0, d;
	for (; n > 0; n -= 4) {
		d = x[0] - y[0]; v0 += d * d;
		d = x[1] - y[1]; v1 += d * d;
		d = x[2] - y[2]; v2 += d * d;
		d = x[3] - y[3]; v3 += d * d;
		if (n % 32 == 0) {
			float ans = (v0 + v1) + (v2 + v3);
			if (ans > now_thres) return FLT_MAX;
		}
		x += 4; y += 4;
	}
	return (v0 + v1) + (v2 + v3);
}

static int pair_cmp(const void* a, const void* b) {
	const int* p = (const int*)a; const int* q = (const int*)b;
	if (p[0] != q[0]) return p[0] < q[0] ? -1 : 1;
	if (p[1] != q[1]) return p[1] < q[1] ? -1 : 1;
	return 0;
}

/* feature/matcher.cc:15-71.  out_pairs holds 2*min(n1,n2) ints; result sorted by (first,second)
 * (the reference's order is thread-timing dependent, matcher.cc:65). Returns #matches. */
int orc_match_exact(const orc_sift_cfg* cfg, const float* d1, int n1, const float* d2, int n2, int* out) {
	const float REJECT_RATIO_SQR = cfg->MATCH_REJECT_NEXT_RATIO * cfg->MATCH_REJECT_NEXT_RATIO;
	int l1 = n1, l2 = n2;
	int rev = l1 > l2;
	const float *pf1 = d1, *pf2 = d2;
	if (rev) { l1 = n2; l2 = n1; pf1 = d2; pf2 = d1; }
	int cnt = 0;
	for (int k = 0; k < l1; ++k) {
		const float* dsc1 = pf1 + 128 * (size_t)k;
		int min_idx = -1;
		float min = FLT_MAX, next_min = min;
		for (int kk = 0; kk < l2; ++kk) {
			float dist = orc_euclidean_sqr(dsc1, pf2 + 128 * (size_t)kk, 128, next_min);
			if (dist < min) { next_min = min; min = dist; min_idx = kk; }
			else if (dist < next_min) next_min = dist;
		}
		if (min > REJECT_RATIO_SQR * next_min) continue;
		const float* dsc2 = pf2 + 128 * (size_t)min_idx;
		for (int kk = 0; kk < l1; ++kk) if (kk != k) {
			float dist = orc_euclidean_sqr(dsc2, pf1 + 128 * (size_t)kk, 128, next_min);
			if (dist < next_min) next_min = dist;
		}
		if (min > REJECT_RATIO_SQR * next_min) continue;
		if (rev) { out[2 * cnt] = min_idx; out[2 * cnt + 1] = k; }
		else { out[2 * cnt] = k; out[2 * cnt + 1] = min_idx; }
		cnt++;
	}
	qsort(out, cnt, 2 * sizeof(int), pair_cmp);
	return cnt;
}

/* the match loop of Stitcher::pairwise_match (stitch/stitcher.cc:96-113) with the exact matcher,
 * omp-parallel over the pair list; returns the total number of matches */
long orc_match_pairs_batch(const orc_sift_cfg* cfg, const float* desc, const int* counts, int n, const int* pairs, int npairs, int nthreads) {
	long* offs = (long*)malloc(sizeof(long) * (n + 1));
	offs[0] = 0;
	for (int i = 0; i < n; ++i) offs[i + 1] = offs[i] + counts[i];
	long total = 0;
	omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic) reduction(+:total)
	for (int p = 0; p < npairs; ++p) {
		const int i = pairs[2 * p], j = pairs[2 * p + 1];
		const int mn = counts[i] < counts[j] ? counts[i] : counts[j];
		int* out = (int*)malloc(sizeof(int) * 2 * (mn > 0 ? mn : 1));
		total += orc_match_exact(cfg, desc + offs[i] * 128, counts[i], desc + offs[j] * 128, counts[j], out);
		free(out);
	}
	free(offs);
	return total;
}

/* the same loop, keeping a per-pair digest: count[p] = #matches, digest[p] = sum over the pair's matches of a 64-bit
 * mix of (first, second) -- order-free, so a device result list can be digested the same way and compared pair by
 * pair at sizes where holding every list twice is pointless (8128 pairs of the config-5 job) */
static unsigned long long mix64(unsigned long long x) {
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
	return x;
}
unsigned long long orc_match_digest(const int* pairs2, int n) {
	unsigned long long d = 0;
	for (int k = 0; k < n; ++k) d += mix64(((unsigned long long)(unsigned)pairs2[2 * k] << 32) | (unsigned)pairs2[2 * k + 1]);
	return d;
}
long orc_match_pairs_digest(const orc_sift_cfg* cfg, const float* desc, const int* counts, int n, const int* pairs, int npairs,
		int nthreads, int* count, unsigned long long* digest) {
	long* offs = (long*)malloc(sizeof(long) * (n + 1));
	offs[0] = 0;
	for (int i = 0; i < n; ++i) offs[i + 1] = offs[i] + counts[i];
	long total = 0;
	omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic) reduction(+:total)
	for (int p = 0; p < npairs; ++p) {
		const int i = pairs[2 * p], j = pairs[2 * p + 1];
		const int mn = counts[i] < counts[j] ? counts[i] : counts[j];
		int* out = (int*)malloc(sizeof(int) * 2 * (mn > 0 ? mn : 1));
		count[p] = orc_match_exact(cfg, desc + offs[i] * 128, counts[i], desc + offs[j] * 128, counts[j], out);
		digest[p] = orc_match_digest(out, count[p]);
		total += count[p];
		free(out);
	}
	free(offs);
	return total;
}
