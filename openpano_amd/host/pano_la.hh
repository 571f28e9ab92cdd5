// pano_la.hh -- the three dense linear-algebra routines the host-side camera estimation needs,
// without Eigen.  The reference calls Eigen for them:
//   Homography::inverse            FullPivLU<3x3>.inverse()              stitch/homography.cc:25-39
//   Camera::rotation_to_angle      JacobiSVD (U V^T of a 3x3)             stitch/camera.cc:91-98
//   Camera::straighten             jacobiSvd().matrixV() of a 3x3 cov     stitch/camera.cc:146-158
//   IBA::get_param_update          JtJ.colPivHouseholderQr().solve(b)     stitch/incremental_bundle_adjuster.cc:250
// These are the published algorithms (Gaussian elimination with complete pivoting; one-sided
// Hestenes-Jacobi SVD; Householder QR with column pivoting, Golub & Van Loan 5.4.1 with LAPACK's
// norm-downdating safeguard), written from scratch.  Row-major storage throughout.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

namespace pano_la {

// inverse of a 3x3 by LU with complete pivoting; returns false (and leaves inv untouched) when a
// pivot falls below eps * 3 * |max pivot| -- Eigen's FullPivLU::isInvertible()
inline bool inverse3(const double* a, double* inv) {
	const int n = 3;
	double lu[9];
	for (int i = 0; i < 9; ++i) lu[i] = a[i];
	int rowt[3], colt[3], nonzero = n;
	double maxpivot = 0;
	for (int k = 0; k < n; ++k) {
		int br = k, bc = k; double best = -1;
		for (int i = k; i < n; ++i) for (int j = k; j < n; ++j) {
			const double v = std::fabs(lu[i * n + j]);
			if (v > best) { best = v; br = i; bc = j; }
		}
		if (best == 0.0) {
			nonzero = k;
			for (int i = k; i < n; ++i) rowt[i] = colt[i] = i;
			break;
		}
		if (best > maxpivot) maxpivot = best;
		rowt[k] = br; colt[k] = bc;
		if (br != k) for (int j = 0; j < n; ++j) std::swap(lu[k * n + j], lu[br * n + j]);
		if (bc != k) for (int i = 0; i < n; ++i) std::swap(lu[i * n + k], lu[i * n + bc]);
		for (int i = k + 1; i < n; ++i) lu[i * n + k] /= lu[k * n + k];
		for (int i = k + 1; i < n; ++i)
			for (int j = k + 1; j < n; ++j)
				lu[i * n + j] -= lu[i * n + k] * lu[k * n + j];
	}
	const double thr = std::fabs(maxpivot) * (DBL_EPSILON * n);
	int rank = 0;
	for (int i = 0; i < nonzero; ++i) rank += (std::fabs(lu[i * n + i]) > thr);
	if (rank != n) return false;
	double c[3];
	for (int col = 0; col < n; ++col) {
		for (int i = 0; i < n; ++i) c[i] = (i == col) ? 1.0 : 0.0;
		for (int i = 0; i < n; ++i) std::swap(c[i], c[rowt[i]]);
		for (int i = 0; i < n; ++i)
			for (int j = 0; j < i; ++j) c[i] -= lu[i * n + j] * c[j];
		for (int i = n - 1; i >= 0; --i) {
			for (int j = i + 1; j < n; ++j) c[i] -= lu[i * n + j] * c[j];
			c[i] /= lu[i * n + i];
		}
		for (int i = n - 1; i >= 0; --i) std::swap(c[i], c[colt[i]]);
		for (int i = 0; i < n; ++i) inv[i * n + col] = c[i];
	}
	return true;
}

inline double det3(const double* m) {
	return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// One-sided Jacobi SVD of an m x n matrix (m >= n): a = U diag(S) V^T, S descending, U m x n, V n x n
inline void jacobi_svd(const double* a, int m, int n, double* U, double* S, double* V) {
	std::vector<double> A(a, a + (size_t)m * n), Vw((size_t)n * n, 0.0);
	for (int i = 0; i < n; ++i) Vw[(size_t)i * n + i] = 1;
	for (int sweep = 0; sweep < 60; ++sweep) {
		double off = 0;
		for (int p = 0; p < n - 1; ++p)
			for (int q = p + 1; q < n; ++q) {
				double alpha = 0, beta = 0, gamma = 0;
				for (int i = 0; i < m; ++i) {
					alpha += A[(size_t)i * n + p] * A[(size_t)i * n + p];
					beta += A[(size_t)i * n + q] * A[(size_t)i * n + q];
					gamma += A[(size_t)i * n + p] * A[(size_t)i * n + q];
				}
				if (gamma == 0.0) continue;
				const double lim = std::sqrt(alpha * beta);
				if (std::fabs(gamma) <= 1e-16 * lim) continue;
				off = std::max(off, std::fabs(gamma) / (lim > 0 ? lim : 1));
				const double zeta = (beta - alpha) / (2.0 * gamma);
				const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
				const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
				for (int i = 0; i < m; ++i) {
					const double x = A[(size_t)i * n + p], y = A[(size_t)i * n + q];
					A[(size_t)i * n + p] = cs * x - sn * y;
					A[(size_t)i * n + q] = sn * x + cs * y;
				}
				for (int i = 0; i < n; ++i) {
					const double x = Vw[(size_t)i * n + p], y = Vw[(size_t)i * n + q];
					Vw[(size_t)i * n + p] = cs * x - sn * y;
					Vw[(size_t)i * n + q] = sn * x + cs * y;
				}
			}
		if (off < 1e-15) break;
	}
	std::vector<double> sv(n);
	std::vector<int> order(n);
	for (int j = 0; j < n; ++j) {
		double s = 0;
		for (int i = 0; i < m; ++i) s += A[(size_t)i * n + j] * A[(size_t)i * n + j];
		sv[j] = std::sqrt(s);
		order[j] = j;
	}
	std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sv[x] > sv[y]; });
	for (int jj = 0; jj < n; ++jj) {
		const int j = order[jj];
		S[jj] = sv[j];
		for (int i = 0; i < m; ++i) U[(size_t)i * n + jj] = sv[j] > 0 ? A[(size_t)i * n + j] / sv[j] : 0.0;
		for (int i = 0; i < n; ++i) V[(size_t)i * n + jj] = Vw[(size_t)i * n + j];
	}
}

// Solve the square system A x = b through a Householder QR with column pivoting (A P = Q R);
// pivots below eps * n * |largest pivot| are treated as zero (minimum-norm-like basic solution).
// The trailing update runs as ROW sweeps (w_j += v_i R_ij for all j, i ascending): every w_j and
// every R_ij sees exactly the operations, in exactly the order, of the textbook column-by-column
// form -- so the result is bit-identical to it -- but the inner loops are contiguous, carry no
// dependence and vectorise (the column form spends its time in n^3/3 latency-bound scalar adds).
// The factorization and its application are separate steps: the Levenberg-Marquardt loop of the bundle adjuster solves
// with the SAME matrix again after every rejected step (only the right-hand side changed), and a factorization kept
// across those solves gives the same x as factoring again -- the same operations on the same values.
struct ColPivQR {
	int n = 0, rank = 0;
	std::vector<double> R, tau;       // R: upper triangle = R factor, below the diagonal the Householder vectors (v_k = 1 implied)
	std::vector<int> perm;
};
// AVX-512 where the host has it (the loops below are contiguous fp64 sweeps; -ffp-contract=off, so a wider vector changes
// no bit of any element's operation sequence)
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define PANO_LA_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define PANO_LA_CLONES
#endif
PANO_LA_CLONES inline void colpiv_qr_factor(const double* Ain, int n, ColPivQR& F) {
	F.n = n; F.R.assign(Ain, Ain + (size_t)n * n); F.tau.assign(n, 0.0); F.perm.resize(n);
	std::vector<double>& R = F.R; std::vector<double>& tau = F.tau; std::vector<int>& perm = F.perm;
	std::vector<double> nrm(n), nrm0(n), w(n), v(n);
	for (int j = 0; j < n; ++j) { nrm[j] = 0; perm[j] = j; }
	for (int i = 0; i < n; ++i) {
		const double* Ri = &R[(size_t)i * n];
		for (int j = 0; j < n; ++j) nrm[j] += Ri[j] * Ri[j];
	}
	for (int j = 0; j < n; ++j) nrm0[j] = nrm[j];
	double maxpivot = 0;
	for (int k = 0; k < n; ++k) {
		int best = k;
		for (int j = k + 1; j < n; ++j) if (nrm[j] > nrm[best]) best = j;
		if (best != k) {
			for (int i = 0; i < n; ++i) std::swap(R[(size_t)i * n + k], R[(size_t)i * n + best]);
			std::swap(perm[k], perm[best]); std::swap(nrm[k], nrm[best]); std::swap(nrm0[k], nrm0[best]);
		}
		// Householder reflector of column k, rows k..n-1: H = I - tau v v^T, v_k = 1
		const double alpha = R[(size_t)k * n + k];
		double sigma = 0;
		for (int i = k + 1; i < n; ++i) sigma += R[(size_t)i * n + k] * R[(size_t)i * n + k];
		double beta = alpha;
		if (sigma == 0.0) tau[k] = 0.0;
		else {
			beta = -std::copysign(std::sqrt(alpha * alpha + sigma), alpha);
			tau[k] = (beta - alpha) / beta;
			const double scale = 1.0 / (alpha - beta);
			for (int i = k + 1; i < n; ++i) R[(size_t)i * n + k] *= scale;
		}
		R[(size_t)k * n + k] = beta;
		if (std::fabs(beta) > maxpivot) maxpivot = std::fabs(beta);
		if (tau[k] != 0.0 && k + 1 < n) {
			const int m = n - (k + 1);
			double* __restrict__ wk = &w[k + 1];
			const double* Rk = &R[(size_t)k * n + k + 1];
			for (int j = 0; j < m; ++j) wk[j] = Rk[j];
			for (int i = k + 1; i < n; ++i) {
				const double vi = R[(size_t)i * n + k];
				const double* __restrict__ Ri = &R[(size_t)i * n + k + 1];
				v[i] = vi;
				for (int j = 0; j < m; ++j) wk[j] += vi * Ri[j];
			}
			const double t = tau[k];
			double* Rkw = &R[(size_t)k * n + k + 1];
			for (int j = 0; j < m; ++j) { wk[j] *= t; Rkw[j] -= wk[j]; }
			for (int i = k + 1; i < n; ++i) {
				const double vi = v[i];
				double* __restrict__ Ri = &R[(size_t)i * n + k + 1];
				for (int j = 0; j < m; ++j) Ri[j] -= wk[j] * vi;
			}
		}
		// downdate the remaining column norms; recompute when cancellation has eaten the value
		for (int j = k + 1; j < n; ++j) {
			nrm[j] -= R[(size_t)k * n + j] * R[(size_t)k * n + j];
			if (!(nrm[j] > 1e-8 * nrm0[j])) {
				double s = 0;
				for (int i = k + 1; i < n; ++i) s += R[(size_t)i * n + j] * R[(size_t)i * n + j];
				nrm[j] = nrm0[j] = s;
			}
		}
	}
	const double thr = maxpivot * (DBL_EPSILON * n);
	int rank = 0;
	while (rank < n && std::fabs(R[(size_t)rank * n + rank]) > thr) ++rank;
	F.rank = rank;
}
inline void colpiv_qr_apply(const ColPivQR& F, const double* b, double* x) {
	const int n = F.n, rank = F.rank;
	const std::vector<double>& R = F.R; const std::vector<double>& tau = F.tau;
	std::vector<double> c(b, b + n);
	// c = Q^T b
	for (int k = 0; k < n; ++k) {
		if (tau[k] == 0.0) continue;
		double ww = c[k];
		for (int i = k + 1; i < n; ++i) ww += R[(size_t)i * n + k] * c[i];
		ww *= tau[k];
		c[k] -= ww;
		for (int i = k + 1; i < n; ++i) c[i] -= ww * R[(size_t)i * n + k];
	}
	std::vector<double> y(n, 0.0);
	for (int i = rank - 1; i >= 0; --i) {
		double s = c[i];
		for (int j = i + 1; j < rank; ++j) s -= R[(size_t)i * n + j] * y[j];
		y[i] = s / R[(size_t)i * n + i];
	}
	for (int i = 0; i < n; ++i) x[F.perm[i]] = y[i];
}
inline void colpiv_qr_solve(const double* Ain, int n, const double* b, double* x) {
	ColPivQR F;
	colpiv_qr_factor(Ain, n, F);
	colpiv_qr_apply(F, b, x);
}

}	// namespace pano_la
