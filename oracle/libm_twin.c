/* oracle/libm_twin.c -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference calls glibc's expf (feature/orientation.cc:63, feature/sift.cc:132),
 * cosf/sinf (feature/sift.cc:107-108, via the std::cos(float) overload) and hypotf
 * (feature/dog.cc:80).  glibc (pinned: 2.35, the image's libm.so.6; third-party, source not
 * under /root/reference) implements all four in double precision with one final rounding to
 * float (expf/sinf/cosf: the "optimized routines" algorithms published with glibc >= 2.27,
 * sysdeps/ieee754/flt-32/{e_expf.c,s_sinf.c,s_cosf.c,sincosf.h}; hypotf:
 * (float)sqrt((double)x*x + (double)y*y)).  The HIP kernels re-implement exactly these fp64
 * sequences (openpano_amd/csrc/devmath.hpp) so that weights, bins and magnitudes come out
 * bit-identical to the reference CPU path.  This file restates them on the host so that
 *   tests/test_libm_twin.py  can check  twin == libm  exhaustively over the argument ranges
 *   the hot path uses, and the GPU tests can check  device == twin.
 * On x86-64 hosts with FMA glibc dispatches to its FMA-compiled variants (__expf_fma,
 * __sinf_fma, __cosf_fma) in which GCC contracted the polynomial a*b+c steps; the twins use
 * explicit fma() in the same places.  (Either variant differs from the other only below
 * double precision, i.e. changes the float result with probability ~2^-29 per call.)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "oracle.h"

static inline uint64_t asu64(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double asf64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint32_t asu32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* 2^(i/32) table, T[i] = bits(2^(i/32)) - (i << 47) */
static const uint64_t EXP2F_T[32] = {
	0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
	0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
	0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
	0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
	0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
	0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
	0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
	0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};

/* valid for |x| < 88 (the hot path only evaluates expf on [-~12, 0]) */
float orc_expf_twin(float x) {
	const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
	const double SHIFT = 0x1.8p+52;
	const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32;
	const double C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32;
	const double C2 = 0x1.62e42ff0c52d6p-1 / 32;
	double xd = (double)x;
	double z = InvLn2N * xd;
	double kd = z + SHIFT;
	uint64_t ki = asu64(kd);
	kd -= SHIFT;
	double r = z - kd;
	uint64_t t = EXP2F_T[ki % 32];
	t += ki << (52 - 5);
	double s = asf64(t);
	z = fma(C0, r, C1);
	double r2 = r * r;
	double y = fma(C2, r, 1.0);
	y = fma(z, r2, y);
	y = y * s;
	return (float)y;
}

/* sincosf polynomial data: [0] for quadrants with positive cosine sign, [1] negated cosine */
static const double SC_HPI_INV = 0x1.45F306DC9C883p+23;	/* 2/pi * 2^24 */
static const double SC_HPI = 0x1.921FB54442D18p0;
static const double SC_C[2][5] = {
	{ 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16 },
	{ -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16 },
};
static const double SC_S[3] = { -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 };
static const double SC_SIGN[4] = { 1.0, -1.0, -1.0, 1.0 };

static inline float sinf_poly_twin(double x, double x2, int tab, int n) {
	if ((n & 1) == 0) {
		double x3 = x * x2;
		double s1 = fma(x2, SC_S[2], SC_S[1]);
		double x7 = x3 * x2;
		double s = fma(x3, SC_S[0], x);
		return (float)fma(x7, s1, s);
	} else {
		double x4 = x2 * x2;
		double c2 = fma(x2, SC_C[tab][4], SC_C[tab][3]);
		double c1 = fma(x2, SC_C[tab][1], SC_C[tab][0]);
		double x6 = x4 * x2;
		double c = fma(x4, SC_C[tab][2], c1);
		return (float)fma(x6, c2, c);
	}
}

static inline uint32_t abstop12(float x) { return (asu32(x) >> 20) & 0x7ff; }

/* valid for |y| < 120 */
float orc_sinf_twin(float y) {
	double x = y;
	if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
		double s = x * x;
		if (abstop12(y) < abstop12(0x1p-12f)) return y;
		return sinf_poly_twin(x, s, 0, 0);
	}
	double r = x * SC_HPI_INV;
	int n = ((int32_t)r + 0x800000) >> 24;
	x = fma(-(double)n, SC_HPI, x);
	double s = SC_SIGN[n & 3];
	int tab = (n & 2) ? 1 : 0;
	return sinf_poly_twin(x * s, x * x, tab, n);
}

float orc_cosf_twin(float y) {
	double x = y;
	if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
		double x2 = x * x;
		if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
		return sinf_poly_twin(x, x2, 0, 1);
	}
	double r = x * SC_HPI_INV;
	int n = ((int32_t)r + 0x800000) >> 24;
	x = fma(-(double)n, SC_HPI, x);
	double s = SC_SIGN[n & 3];
	int tab = (n & 2) ? 1 : 0;
	return sinf_poly_twin(x * s, x * x, tab, n ^ 1);
}

/* finite inputs only */
float orc_hypotf_twin(float x, float y) {
	double dx = x, dy = y;
	return (float)sqrt(fma(dx, dx, dy * dy));
}
