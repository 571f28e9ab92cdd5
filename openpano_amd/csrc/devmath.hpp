// devmath.hpp -- device (gfx950) twins of the libm calls on the reference's SIFT path.
//
// The reference evaluates glibc expf (feature/orientation.cc:63, feature/sift.cc:132),
// cosf/sinf (feature/sift.cc:107-108), hypotf (feature/dog.cc:80) and its own fast_atan
// (feature/dog.cc:22-37, an fp64 polynomial because its literals are double).  All of these are
// "evaluate in double, round once to float" algorithms, so they are reproduced here in fp64
// VALU arithmetic, operation for operation, and the results are bit-identical to the CPU path
// (checked exhaustively on the host twins in oracle/libm_twin.c and on the device through
// op_debug_math).  This file is compiled with -ffp-contract=off: every fma() below is explicit
// and matches a contraction in glibc's FMA build; nothing else is fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace opdev {

// bits(2^(i/32)) - (i << 47): the exp2f table of glibc's expf (N = 32)
__device__ const uint64_t kExp2fTab[32] = {
	0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
	0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
	0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
	0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
	0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
	0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
	0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
	0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};

// glibc expf for |x| < 88 (the hot path only ever passes x in [-~12, 0])
__device__ __forceinline__ float expf_glibc(float x, const uint64_t* tab) {
	const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
	const double SHIFT = 0x1.8p+52;
	const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32;
	const double C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32;
	const double C2 = 0x1.62e42ff0c52d6p-1 / 32;
	double z = InvLn2N * (double)x;
	double kd = z + SHIFT;
	uint64_t ki = (uint64_t)__double_as_longlong(kd);
	kd -= SHIFT;
	double r = z - kd;
	uint64_t t = tab[ki & 31];
	t += ki << 47;
	double s = __longlong_as_double((long long)t);
	z = fma(C0, r, C1);
	double r2 = r * r;
	double y = fma(C2, r, 1.0);
	y = fma(z, r2, y);
	y = y * s;
	return (float)y;
}
__device__ __forceinline__ float expf_glibc(float x) { return expf_glibc(x, kExp2fTab); }

__device__ __forceinline__ uint32_t abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ff; }

// polynomial of glibc sinf/cosf (sincosf.h): n even -> sine, n odd -> cosine; neg = use the
// table with negated cosine coefficients
__device__ __forceinline__ float sincosf_poly(double x, double x2, bool neg, int n) {
	if ((n & 1) == 0) {
		const double S0 = -0x1.555545995a603p-3, S1 = 0x1.1107605230bc4p-7, S2 = -0x1.994eb3774cf24p-13;
		double x3 = x * x2;
		double s1 = fma(x2, S2, S1);
		double x7 = x3 * x2;
		double s = fma(x3, S0, x);
		return (float)fma(x7, s1, s);
	} else {
		double sg = neg ? -1.0 : 1.0;
		const double C0 = 0x1p0 * sg, C1 = -0x1.ffffffd0c621cp-2 * sg, C2 = 0x1.55553e1068f19p-5 * sg,
			  C3 = -0x1.6c087e89a359dp-10 * sg, C4 = 0x1.99343027bf8c3p-16 * sg;
		double x4 = x2 * x2;
		double c2 = fma(x2, C4, C3);
		double c1 = fma(x2, C1, C0);
		double x6 = x4 * x2;
		double c = fma(x4, C2, c1);
		return (float)fma(x6, c2, c);
	}
}

// glibc sinf / cosf for |y| < 120; which = 0 sine, 1 cosine
__device__ __forceinline__ float sincosf_glibc(float y, int which) {
	double x = (double)y;
	if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
		if (abstop12(y) < abstop12(0x1p-12f)) return which ? 1.0f : y;
		return sincosf_poly(x, x * x, false, which);
	}
	double r = x * 0x1.45F306DC9C883p+23;
	int n = ((int32_t)r + 0x800000) >> 24;
	x = fma(-(double)n, 0x1.921FB54442D18p0, x);
	int q = n & 3;
	double s = (q == 1 || q == 2) ? -1.0 : 1.0;
	return sincosf_poly(x * s, x * x, (n & 2) != 0, n ^ which);
}
__device__ __forceinline__ float sinf_glibc(float y) { return sincosf_glibc(y, 0); }
__device__ __forceinline__ float cosf_glibc(float y) { return sincosf_glibc(y, 1); }

// glibc hypotf for finite inputs: (float) sqrt((double)x*x + (double)y*y)
__device__ __forceinline__ float hypotf_glibc(float x, float y) {
	double dx = (double)x, dy = (double)y;
	return (float)sqrt(fma(dx, dx, dy * dy));
}

// fast_atan(y, x) + M_PI, exactly as evaluated at feature/dog.cc:83 (ort in [0, 2pi])
__device__ __forceinline__ float fast_atan_plus_pi(float y, float x) {
	const double PI = 3.14159265358979323846, PI_2 = 1.57079632679489661923;
	float absx = fabsf(x), absy = fabsf(y);
	float m = fmaxf(absx, absy);
	float r;
	if ((double)m < 1e-6) {
		r = (float)-PI;
	} else {
		float a = fminf(absx, absy) / m;
		float s = a * a;
		double ds = (double)s, da = (double)a;
		r = (float)(((-0.0464964749 * ds + 0.15931422) * ds - 0.327622764) * ds * da + da);
		if (absy > absx) r = (float)(PI_2 - (double)r);
		if (x < 0) r = (float)(PI - (double)r);
		if (y < 0) r = -r;
	}
	return (float)((double)r + PI);
}

}	// namespace opdev
