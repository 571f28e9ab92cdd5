// ransac_accept_simd.cc -- ISA clones of the acceptance epilogue's keypoint count (ransac_accept.hpp, count_in_polygon_v).
// Plain host C++ (no HIP): built with g++ without contraction.  Every clone performs the same IEEE operations per lane --
// the width of the registers is the only difference -- so the count does not depend on which one runs
// (tests/harness/ransac_accept_harness.cc runs all of them against the reference's expression).  Which one runs is settled
// at the first call by a trial of the clones this CPU executes (a few thousand points each, ~0.1 ms once per process): the
// widest is not the fastest everywhere (512-bit divides of some server parts run at a quarter of the 256-bit rate).
#include "ransac_accept.hpp"
#include <time.h>

namespace opaccept {
__attribute__((target("avx512f"))) int count_in_polygon_avx512(const PolygonTables& T, const double* xy, size_t stride, int n) { return count_in_polygon_v<8>(T, xy, stride, n); }
__attribute__((target("avx2"))) int count_in_polygon_avx2(const PolygonTables& T, const double* xy, size_t stride, int n) { return count_in_polygon_v<4>(T, xy, stride, n); }
int count_in_polygon_baseline(const PolygonTables& T, const double* xy, size_t stride, int n) { return count_in_polygon_v<2>(T, xy, stride, n); }

typedef int (*count_fn)(const PolygonTables&, const double*, size_t, int);
static count_fn pick_clone() {
	count_fn cand[3]; int nc = 0;
	if (__builtin_cpu_supports("avx512f")) cand[nc++] = count_in_polygon_avx512;
	if (__builtin_cpu_supports("avx2")) cand[nc++] = count_in_polygon_avx2;
	cand[nc++] = count_in_polygon_baseline;
	if (nc == 1) return cand[0];
	// a 12-gon and 1024 points around it
	std::vector<P2> poly;
	for (int i = 0; i < 12; ++i) { const double a = 0.5235987755982988 * i + 0.1; poly.push_back(P2{500 * std::cos(a), 330 * std::sin(a)}); }
	const PointInPolygon pip(poly);
	const PolygonTables T(pip);
	std::vector<double> pts(2048);
	unsigned long long z = 88172645463325252ULL;
	for (auto& v : pts) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = (double)(long long)(z % 1300) - 650.0 + 0.37; }
	auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
	count_fn best = cand[nc - 1]; double best_t = 1e30;
	for (int c = 0; c < nc; ++c) {
		volatile int sink = cand[c](T, pts.data(), 2, 1024);       // warm: code, tables, the wide units' power-up
		(void)sink;
		double t = 1e30;
		for (int rep = 0; rep < 3; ++rep) { const double t0 = now(); sink = cand[c](T, pts.data(), 2, 1024); const double dt = now() - t0; t = dt < t ? dt : t; }
		if (t < best_t) { best_t = t; best = cand[c]; }
	}
	return best;
}

int count_in_polygon(const PolygonTables& T, const double* xy, size_t stride, int n) {
	static const count_fn f = pick_clone();
	return f(T, xy, stride, n);
}
}	// namespace opaccept
