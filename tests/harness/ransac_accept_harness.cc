// CPU harness of openpano_amd/csrc/ransac_accept.hpp (compiled by tests/test_ransac_accept_cpu.py with g++):
//   1. fast_atan2 against libm over random and special arguments: prints the largest absolute difference;
//   2. PointInPolygon::in_polygon (fast wedge search, exact fallback) against in_polygon_exact (the reference's own
//      expression, lib/polygon.cc:62-82) on overlap polygons of random homographies (overlap_region, homography.cc:50-90):
//      random points, points ON vertex directions (the fallback's case) and points a few float ulps off them.
#include "ransac_accept.hpp"
#include <cstdio>
#include <cstdint>
#include <random>
#include <time.h>

using namespace opaccept;

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
	long npoly = 0, npts = 0, mismatch = 0, on_dir = 0, inside = 0;
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
		auto test = [&](P2 p) { ++npts; const bool a = pip.in_polygon(p), b = pip.in_polygon_exact(p); mismatch += a != b; inside += b; };
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
	}
	std::printf("vertices mean %.1f max %ld\n", (double)nvert / npoly, maxvert);
	std::printf("atan2_samples %ld max_abs_err %.3e polygons %ld points %ld inside %ld on_vertex_direction %ld mismatches %ld\n", nat, maxerr, npoly, npts, inside, on_dir, mismatch);
	return mismatch == 0 && maxerr < 1e-10 ? 0 : 1;
}
