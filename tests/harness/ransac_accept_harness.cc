// CPU harness of openpano_amd/csrc/ransac_accept.hpp (compiled by tests/test_ransac_accept_cpu.py with g++):
//   1. fast_atan2 against libm over random and special arguments: prints the largest absolute difference;
//   2. PointInPolygon::in_polygon (fast wedge search, exact fallback) against in_polygon_exact (the reference's own
//      expression, lib/polygon.cc:62-82) on overlap polygons of random homographies (overlap_region, homography.cc:50-90):
//      random points, points ON vertex directions (the fallback's case) and points a few float ulps off them.
#include "ransac_accept.hpp"
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <time.h>

using namespace opaccept;
namespace opaccept {
int count_in_polygon_avx512(const PolygonTables& T, const double* xy, size_t stride, int n);
int count_in_polygon_avx2(const PolygonTables& T, const double* xy, size_t stride, int n);
int count_in_polygon_baseline(const PolygonTables& T, const double* xy, size_t stride, int n);
}

int main() {
	std::mt19937_64 rng(12345);
	std::uniform_real_distribution<double> U(-1.0, 1.0);
	double maxerr = 0; long nat = 0;
	auto chk = [&](double y, double x) { const double a = std::atan2(y, x), b = fast_atan2(y, x); const double e = std::fabs(a - b); if (e > maxerr) maxerr = e; ++nat; if ((a < 0) != (b < 0) && std::fabs(a) > 1e-9) maxerr = 10; };
	for (int i = 0; i < 4000000; ++i) { const double s = std::pow(10.0, 6 * U(rng)); chk(U(rng) * s, U(rng) * s); }
	const double sp[] = {0.0, -0.0, 1.0, -1.0, 1e-300, -1e-300, 1e300, -1e300, 0.41421356237309503, 2.414213562373095, 5e-324, -5e-324};
	for (double y : sp) for (double x : sp) chk(y, x);
	for (int i = 0; i < 200000; ++i) { const double th = U(rng) * 3.141592653589793; chk(std::sin(th), std::cos(th)); chk(std::sin(th) * 1e-7, -1.0); chk(-1.0, std::cos(th) * 1e-9); }
	long nvert = 0, maxvert = 0;
	long npoly = 0, npts = 0, mismatch = 0, on_dir = 0, inside = 0, count_mismatch = 0, count_calls = 0;
	const bool has512 = __builtin_cpu_supports("avx512f"), has2 = __builtin_cpu_supports("avx2");
	for (int trial = 0; trial < 3000; ++trial) {
		const Shape s1{600 + (int)(rng() % 900), 400 + (int)(rng() % 700)}, s2{600 + (int)(rng() % 900), 400 + (int)(rng() % 700)};
		const double ang = 0.2 * U(rng), sc = 1.0 + 0.2 * U(rng);
		double h[9] = {sc * std::cos(ang), -sc * std::sin(ang), 0.7 * s1.w * U(rng), sc * std::sin(ang), sc * std::cos(ang), 0.7 * s1.h * U(rng), 2e-4 * U(rng), 2e-4 * U(rng), 1.0};
		double inv[9];
		if (!inverse3(h, inv)) continue;
		std::vector<P2> poly = overlap_region(s1, s2, h, inv);
		if (poly.size() < 3) continue;
		++npoly; nvert += (long)poly.size(); maxvert = std::max<long>(maxvert, (long)poly.size());
		PointInPolygon pip(poly);
		std::vector<double> batch;                               // every point of this polygon, for the 8-at-a-time count (stride 4, like the match lists)
		long batch_inside = 0;
		auto test = [&](P2 p) { ++npts; const bool a = pip.in_polygon(p), b = pip.in_polygon_exact(p); mismatch += a != b; inside += b;
			batch.push_back(p.x); batch.push_back(p.y); batch.push_back(0); batch.push_back(0); batch_inside += b; };
		for (int k = 0; k < 400; ++k) test(P2{0.6 * s1.w * U(rng), 0.6 * s1.h * U(rng)});
		for (size_t v = 0; v < poly.size(); ++v) {            // on and next to every vertex direction, inside and outside the polygon
			for (double r : {0.3, 0.999999, 1.0, 1.000001, 1.7}) {
				const P2 d{poly[v].x - pip.com.x, poly[v].y - pip.com.y};
				test(P2{pip.com.x + r * d.x, pip.com.y + r * d.y}); ++on_dir;
				for (double eps : {1e-7, -1e-7, 4e-7, -4e-7, 1e-6, -1e-6}) {
					const double c = std::cos(eps), s = std::sin(eps);
					test(P2{pip.com.x + r * (c * d.x - s * d.y), pip.com.y + r * (s * d.x + c * d.y)});
				}
			}
		}
		{	// zero offsets, infinities and NaNs go to libm in every path
			const double inf = 1.0 / 0.0;
			for (P2 q : {pip.com, P2{inf, 0}, P2{0, -inf}, P2{inf, inf}, P2{0.0 / 0.0, 1}, P2{pip.com.x, 1e300}, P2{-1e300, pip.com.y}}) test(q);
		}
		const PolygonTables T(pip);
		const int nb = (int)(batch.size() / 4);
		for (int off : {0, 1, 5}) {                               // different alignments of the 8-lane blocks and of the scalar tail
			long want = 0;
			for (int k = off; k < nb; ++k) want += pip.in_polygon_exact(P2{batch[4 * k], batch[4 * k + 1]});
			if (off == 0 && want != batch_inside) ++count_mismatch;
			++count_calls;
			count_mismatch += count_in_polygon(T, batch.data() + 4 * off, 4, nb - off) != want;
			count_mismatch += count_in_polygon_baseline(T, batch.data() + 4 * off, 4, nb - off) != want;
			if (has2) count_mismatch += count_in_polygon_avx2(T, batch.data() + 4 * off, 4, nb - off) != want;
			if (has512) count_mismatch += count_in_polygon_avx512(T, batch.data() + 4 * off, 4, nb - off) != want;
		}
	}
	{	// what the change buys: 1200 keypoints against one overlap polygon, the epilogue's inner loop
		const Shape s1{1300, 867};
		double h[9] = {1, 0, 400, 0, 1, 30, 1e-5, 0, 1}, inv[9];
		inverse3(h, inv);
		std::vector<P2> poly = overlap_region(s1, s1, h, inv);
		PointInPolygon pip(poly);
		std::vector<P2> pts(1200);
		for (auto& q : pts) q = P2{650 * U(rng), 433 * U(rng)};
		auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
		long acc = 0;
		double t0 = now();
		for (int rep = 0; rep < 2000; ++rep) for (auto& q : pts) acc += pip.in_polygon_exact(q);
		double t1 = now();
		for (int rep = 0; rep < 2000; ++rep) for (auto& q : pts) acc -= pip.in_polygon(q);
		double t2 = now();
		std::printf("ns_per_point exact %.1f fast %.1f (polygon of %d vertices, check %ld)\n", (t1 - t0) / 2.4e6 * 1e9, (t2 - t1) / 2.4e6 * 1e9, (int)poly.size(), acc);
		const PolygonTables T(pip);
		typedef int (*fn)(const PolygonTables&, const double*, size_t, int);
		const struct { const char* name; fn f; bool ok; } cl[] = {{"dispatch", count_in_polygon, true}, {"baseline", count_in_polygon_baseline, true},
				{"avx2", count_in_polygon_avx2, has2}, {"avx512", count_in_polygon_avx512, has512}};
		for (auto& c : cl) {
			if (!c.ok) continue;
			long a2 = 0;
			const double u0 = now();
			for (int rep = 0; rep < 2000; ++rep) a2 += c.f(T, &pts[0].x, 2, (int)pts.size());
			std::printf("ns_per_point count_in_polygon %s %.2f (check %ld)\n", c.name, (now() - u0) / 2.4e6 * 1e9, a2);
		}
	}
	long refits = 0, refit_mismatch = 0;
	{	// the refit with several rows in flight == opransac::calc_transform, bit for bit: homography and affine, 4..3000 points, exact zeros among the coordinates
		for (int trial = 0; trial < 600; ++trial) {
			const int n = trial < 40 ? 4 + trial : 16 + (int)(rng() % (trial % 50 == 0 ? 3000 : 300));
			double hh[9] = {1 + 0.1 * U(rng), 0.05 * U(rng), 300 * U(rng), 0.05 * U(rng), 1 + 0.1 * U(rng), 200 * U(rng), 1e-4 * U(rng), 1e-4 * U(rng), 1.0};
			std::vector<double> P(4 * (size_t)n);
			for (int i = 0; i < n; ++i) {
				P2 b{650 * U(rng), 430 * U(rng)};
				if (trial % 7 == 0 && i % 5 == 0) b.x = 0.0;
				if (trial % 11 == 0 && i % 3 == 0) b.y = 0.0;
				if (trial % 13 == 0 && i == 2) { b.x = 0.0; b.y = 0.0; }
				const P2 a = trans2d(hh, b);
				P[4 * i] = a.x + U(rng); P[4 * i + 1] = a.y + U(rng); P[4 * i + 2] = b.x; P[4 * i + 3] = b.y;
				if (trial % 17 == 0 && i % 4 == 1) P[4 * i] = 0.0;
			}
			for (int affine = 0; affine < 2; ++affine) {
				double H1[9], H2[9];
				auto g1 = [&](int q) { return P2{P[4 * q], P[4 * q + 1]}; };
				auto g2 = [&](int q) { return P2{P[4 * q + 2], P[4 * q + 3]}; };
				opransac::calc_transform(n, g1, g2, affine != 0, H1);
				calc_transform_skewed(n, g1, g2, affine != 0, H2);
				++refits; refit_mismatch += std::memcmp(H1, H2, sizeof(H1)) != 0;
			}
		}
	}
	std::printf("refits %ld refit_mismatches %ld\n", refits, refit_mismatch);
	std::printf("vertices mean %.1f max %ld\n", (double)nvert / npoly, maxvert);
	std::printf("atan2_samples %ld max_abs_err %.3e polygons %ld points %ld inside %ld on_vertex_direction %ld mismatches %ld\n", nat, maxerr, npoly, npts, inside, on_dir, mismatch);
	std::printf("count_calls %ld count_mismatches %ld avx2 %d avx512 %d\n", count_calls, count_mismatch, (int)has2, (int)has512);
	return mismatch == 0 && count_mismatch == 0 && refit_mismatch == 0 && maxerr < 1e-10 ? 0 : 1;
}
