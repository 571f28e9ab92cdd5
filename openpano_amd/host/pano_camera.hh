// pano_camera.hh -- host-side camera estimation and bundle adjustment without Eigen
// (SURVEY 8(f).2): the step of Stitcher::build() between the device-side match/RANSAC stage and
// the device-side blend, under ESTIMATE_CAMERA (BASELINE configs 2-4).  Same classes, members and
// call order as the reference so that code written against it compiles unchanged:
//   Camera                        stitch/camera.hh:12-49, stitch/camera.cc:19-183
//   IncrementalBundleAdjuster     stitch/incremental_bundle_adjuster.hh:20-120, .cc:19-417
//   CameraEstimator               stitch/camera_estimator.hh, .cc:21-160
// Linear algebra comes from pano_la.hh (3x3 SVD / inverse, column-pivoted QR solve) where the
// reference calls Eigen.  Differences by construction, results unchanged:
//   * the Jacobian J (2M x 6n, up to 700 000 rows, whose setZero() the reference's author measured
//     at a third of the time, incremental_bundle_adjuster.cc:280) is never materialised: J^T r is
//     accumulated term by term in the same order a dense row-by-row product visits the non-zeros
//     (the zeros of J contribute exact zeros), JtJ exactly as the reference accumulates it;
//   * only the analytic Jacobian (SYMBOLIC_DIFF = true, :22) is provided.
// Used standalone (pano_types.hh); with -DOPENPANO_WITH_REFERENCE the reference's own classes
// are in scope instead and this header is not included.
#pragma once
#ifdef _OPENMP
#include <omp.h>
#if defined(__linux__)
#include <sched.h>
#endif
#endif
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <limits>
#include <queue>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "pano_types.hh"

namespace pano {

[[noreturn]] inline void pano_error_exit(const std::string& msg) {      // lib/debugutils.cc:57-60
	fprintf(stderr, "error: %s\n", msg.c_str());
	exit(1);
}

constexpr double PANO_EPS = 1e-6, PANO_GEO_EPS = 1e-7, PANO_GEO_EPS_SQR = 1e-14;   // lib/utils.hh:22-24
// lib/utils.hh:25 -- the reference's only sqr() takes and returns FLOAT: the two places of the bundle
// adjuster that call it on doubles (the error statistic, incremental_bundle_adjuster.cc:202, and
// 1/z^2 of the projection derivative, :309) round through fp32, and so do these
inline float pano_sqr(float x) { return x * x; }

class Camera {
	public:
		double focal = 1, aspect = 1, ppx = 0, ppy = 0;
		Homography R = Homography::I();

		Homography K() const {                                  // camera.cc:59-66
			Homography ret = Homography::I();
			ret[0] = focal; ret[2] = ppx; ret[4] = focal * aspect; ret[5] = ppy;
			return ret;
		}
		Homography Kinv() const { return K().inverse(); }
		Homography Rinv() const { return R.transpose(); }

		// Szeliski, "Creating Full View Panoramic Image Mosaics" (camera.cc:19-53)
		static double get_focal_from_matrix(const Homography& h) {
			double d1, d2, v1, v2, f1, f0;
			d1 = h[6] * h[7];
			d2 = (h[7] - h[6]) * (h[7] + h[6]);
			v1 = -(h[0] * h[1] + h[3] * h[4]) / d1;
			v2 = (h[0] * h[0] + h[3] * h[3] - h[1] * h[1] - h[4] * h[4]) / d2;
			if (v1 < v2) std::swap(v1, v2);
			if (v1 > 0 && v2 > 0) f1 = sqrt(std::abs(d1) > std::abs(d2) ? v1 : v2);
			else if (v1 > 0) f1 = sqrt(v1);
			else return 0;
			d1 = h[0] * h[3] + h[1] * h[4];
			d2 = h[0] * h[0] + h[1] * h[1] - h[3] * h[3] - h[4] * h[4];
			v1 = -h[2] * h[5] / d1;
			v2 = (h[5] * h[5] - h[2] * h[2]) / d2;
			if (v1 < v2) std::swap(v1, v2);
			if (v1 > 0 && v2 > 0) f0 = sqrt(std::abs(d1) > std::abs(d2) ? v1 : v2);
			else if (v1 > 0) f0 = sqrt(v1);
			else return 0;
			if (std::isinf(f1) || std::isinf(f0)) return 0;
			return sqrt(f1 * f0);
		}

		// median of the per-pair estimates over confident pairs i < j (camera.cc:68-87)
		static double estimate_focal(const std::vector<std::vector<MatchInfo>>& matches) {
			const int n = (int)matches.size();
			std::vector<double> estimates;
			for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) {
				const MatchInfo& match = matches[i][j];
				if (match.confidence < PANO_EPS) continue;
				estimates.emplace_back(get_focal_from_matrix(match.homo));
			}
			const int ne = (int)estimates.size();
			if (ne < std::min(n - 1, 3)) return -1;
			std::sort(estimates.begin(), estimates.end());
			if (ne % 2 == 1) return estimates[ne >> 1];
			return (estimates[ne >> 1] + estimates[(ne >> 1) - 1]) * 0.5;
		}

		// nearest rotation (U V^T of the SVD), then axis * angle (camera.cc:91-120)
		static void rotation_to_angle(const Homography& r, double& rx, double& ry, double& rz) {
			double U[9], S[3], V[9], Rn[9];
			pano_la::jacobi_svd(r.data, 3, 3, U, S, V);
			for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
				double s = 0;
				for (int k = 0; k < 3; ++k) s += U[i * 3 + k] * V[j * 3 + k];      // U * V^T
				Rn[i * 3 + j] = s;
			}
			if (pano_la::det3(Rn) < 0) for (int i = 0; i < 9; ++i) Rn[i] *= -1;
			rx = Rn[7] - Rn[5];
			ry = Rn[2] - Rn[6];
			rz = Rn[3] - Rn[1];
			const double s = sqrt(rx * rx + ry * ry + rz * rz);
			if (s < PANO_GEO_EPS) { rx = ry = rz = 0; }
			else {
				double c = (Rn[0] + Rn[4] + Rn[8] - 1) * 0.5;
				c = c > 1. ? 1. : c < -1. ? -1. : c;
				const double theta = acos(c);
				const double mul = 1.0 / s * theta;
				rx *= mul; ry *= mul; rz *= mul;
			}
		}

		// Rodrigues (camera.cc:123-147)
		static void angle_to_rotation(double rx, double ry, double rz, Homography& r) {
			double theta = rx * rx + ry * ry + rz * rz;
			if (theta < PANO_GEO_EPS_SQR) {
				const double t[9] = {1, -rz, ry, rz, 1, -rx, -ry, rx, 1};
				r = Homography(t);
				return;
			}
			theta = sqrt(theta);
			const double itheta = theta ? 1. / theta : 0.;
			rx *= itheta; ry *= itheta; rz *= itheta;
			const double u_outp[] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
			const double u_crossp[] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
			r = Homography::I();
			const double c = cos(theta), s = sin(theta), c1 = 1 - c;
			r.mult(c);
			for (int k = 0; k < 9; ++k) r[k] += c1 * u_outp[k] + s * u_crossp[k];
		}

		// wave correction: make the cameras' X axes orthogonal to a common up vector (camera.cc:149-183)
		static void straighten(std::vector<Camera>& cameras) {
			double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
			for (auto& c : cameras) {
				const double v[3] = {c.R[0], c.R[1], c.R[2]};
				for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; s += v[i] * v[j]; cov[i * 3 + j] += s; }
			}
			double U[9], S[3], V[9];
			pano_la::jacobi_svd(cov, 3, 3, U, S, V);
			Vec normY(V[2], V[5], V[8]);                        // V.col(2)
			Vec vz(0, 0, 0);
			for (auto& c : cameras) { vz.x += c.R[6]; vz.y += c.R[7]; vz.z += c.R[8]; }
			Vec normX = normY.cross(vz);
			{
				const double nn = sqrt(normX.x * normX.x + normX.y * normX.y + normX.z * normX.z);
				normX.x /= nn; normX.y /= nn; normX.z /= nn;
			}
			Vec normZ = normX.cross(normY);
			double s = 0;
			for (auto& c : cameras) s += normX.dot(Vec(c.R[0], c.R[1], c.R[2]));
			if (s < 0) { normX = normX * -1; normY = normY * -1; }
			Homography r;
			const double nx[3] = {normX.x, normX.y, normX.z}, ny[3] = {normY.x, normY.y, normY.z}, nz[3] = {normZ.x, normZ.y, normZ.z};
			for (int i = 0; i < 3; ++i) { r[i * 3] = nx[i]; r[i * 3 + 1] = ny[i]; r[i * 3 + 2] = nz[i]; }
			for (auto& c : cameras) c.R = c.R * r;
		}
};

// wall-clock split of the bundle adjuster (printed by CameraEstimator::estimate when PANO_BA_PROFILE is set)
struct BaProfile { double t_err = 0, t_jac = 0, t_solve = 0; long n_iter = 0, n_opt = 0; };
inline BaProfile& ba_prof() { static BaProfile p; return p; }
// OpenMP team of the bundle adjuster: a few hundred parallel regions of well under a millisecond
// each, so a team spanning every logical CPU of a big host spends more in fork/join than it gains
inline int ba_threads() {
	static const int n = [] {
		int t = 1;
#ifdef _OPENMP
		t = omp_get_max_threads();
		// measured on a 256-CPU host (2 x EPYC 9575F), natural-sized table, per estimate: 1 thread 101 ms, 8 threads 51 ms, 16 threads
		// 52-56 ms, 24 threads 55-100 ms: eight cores share one L3, and the sections pass derivative rows and residuals between the team's caches
		if (t > 8) t = 8;
#endif
#if defined(__linux__)
		{	// never more spinning team members than CPUs this process may run on (taskset, cpuset cgroups)
			cpu_set_t set;
			if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c >= 1 && t > c) t = c; }
		}
#endif
		if (const char* e = std::getenv("PANO_BA_THREADS")) { const int v = std::atoi(e); if (v > 0) t = v; }
		return t;
	}();
	return n;
}
inline double ba_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The bundle adjuster's loops are a few hundred parallel sections of 20-200 microseconds each; an OpenMP fork / join per
// section costs as much as the section on a big host (threads that went to sleep between two sections take tens of
// microseconds to come back: 747 LM iterations x 3 sections were 120 of the 127 ms of an estimate).  One parallel region
// therefore spans a whole optimize() call: thread 0 runs the serial Levenberg-Marquardt logic, the others spin on an epoch
// counter and join every section it publishes (items handed out by an atomic counter).  Sections only ever write disjoint
// outputs and every output's own arithmetic is sequential, so which thread runs an item changes no bit.
class BaTeam {
	public:
		explicit BaTeam(int threads): nthreads(threads) {}
		void set_threads(int t) { nthreads = t; }
		int threads() const { return nthreads; }
		// thread 0: run items 0..n-1 of f with the team; returns when all ITEMS are done -- not when every worker has
		// checked in: a worker the OS has descheduled (fewer runnable CPUs than team members: a cgroup quota, taskset,
		// several estimators at once) is not waited for, it finds the section over when it comes back
		void run(int n_items, const std::function<void(int)>& f) {
			if (nthreads <= 1 || n_items < 2) { for (int i = 0; i < n_items; ++i) f(i); return; }
			for (int base = 0; base < n_items; base += MAX_ITEMS) {       // (a section holds < 2^20 items: the count rides in the ticket)
				const int cnt = std::min(n_items - base, (int)MAX_ITEMS);
				if (base == 0 && cnt == n_items) { run_section(cnt, f); break; }
				const std::function<void(int)> g = [&f, base](int i) { f(base + i); };
				run_section(cnt, g);
			}
		}
		// threads 1..: until thread 0 calls finish()
		void worker_loop() {
			unsigned long long seen = 0;
			for (;;) {
				unsigned long long e;
				for (unsigned spins = 0; (e = section_of(ticket.load(std::memory_order_acquire))) == seen; ++spins) {
					if (quit.load(std::memory_order_acquire)) return;
					relax(spins);
				}
				seen = e;
				work(e);
			}
		}
		void finish() { quit.store(true, std::memory_order_release); }
	private:
		int nthreads;
		static constexpr unsigned long long MAX_ITEMS = (1ull << 20) - 1, FIELD = (1ull << 20) - 1;
		unsigned long long section = 0;                      // thread 0 only
		// (section << 40) | (item count << 20) | next item.  An item is claimed by ONE compare-exchange on the whole word, so the
		// claim itself proves that the section is the live one and that the item lies inside IT.  (The count must not live in a
		// variable of its own: a worker that read a spent ticket of section e and then the count thread 0 had already written for
		// section e + 1 -- but before the new ticket was out -- claimed a phantom item n(e) on the OLD ticket, ran it, and its
		// `done` landed in section e + 1: done == n + 1, thread 0 waits for ever.  Seen once in ~200 estimates on 8 cores.)
		// A worker would have to sleep through 2^24 sections of one optimize() call to meet its section number again.
		std::atomic<unsigned long long> ticket{0};
		std::atomic<int> done{0};
		std::atomic<bool> quit{false};
		std::atomic<const std::function<void(int)>*> fn{nullptr};
		static unsigned long long section_of(unsigned long long t) { return t >> 40; }
		void run_section(int n_items, const std::function<void(int)>& f) {
			fn.store(&f, std::memory_order_relaxed);
			done.store(0, std::memory_order_relaxed);
			section = (section + 1) & ((1ull << 24) - 1);
			if (section == 0) section = 1;
			const unsigned long long e = section;
			ticket.store((e << 40) | ((unsigned long long)n_items << 20), std::memory_order_release);
			work(e);
			for (unsigned spins = 0; done.load(std::memory_order_acquire) != n_items; ++spins) relax(spins);
		}
		void work(unsigned long long e) {
			for (;;) {
				unsigned long long t = ticket.load(std::memory_order_acquire);
				if (section_of(t) != e) return;
				const unsigned long long i = t & FIELD, cnt = (t >> 20) & FIELD;
				if (i >= cnt) return;
				if (!ticket.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
				// claimed under the live ticket: the section cannot end (done != cnt) before this item has, so fn is this section's
				(*fn.load(std::memory_order_relaxed))((int)i);
				done.fetch_add(1, std::memory_order_release);
			}
		}
		// a pause per poll; on a host with fewer free cores than team members the waiters must not starve the thread they wait for
		static void relax(unsigned spins) {
#if defined(__x86_64__)
			__builtin_ia32_pause();
#endif
			if ((spins & 0x3FFu) == 0x3FFu) std::this_thread::yield();
		}
};

class IncrementalBundleAdjuster {
	public:
		struct ErrorStats {
			std::vector<double> residuals;
			double max = 0, avg = 0;
			explicit ErrorStats(int size): residuals(size) {}
			int num_terms() const { return (int)residuals.size(); }
			void update_stats(int) {                           // :208-229 (squared error)
				avg = max = 0;
				for (auto& e : residuals) { avg += pano_sqr((float)e); if (fabs(e) > max) max = fabs(e); }
				avg /= residuals.size();
				avg = sqrt(avg);
			}
		};

		explicit IncrementalBundleAdjuster(std::vector<Camera>& cameras): result_cameras(cameras), index_map(cameras.size()) {}
		IncrementalBundleAdjuster(const IncrementalBundleAdjuster&) = delete;
		IncrementalBundleAdjuster& operator=(const IncrementalBundleAdjuster&) = delete;

		// m is matches[j][i] in the stitcher, i.e. from i to j (:116-123)
		void add_match(int i, int j, const MatchInfo& m) {
			match_pairs.emplace_back(i, j, m);
			match_cnt_prefix_sum.emplace_back(nr_pointwise_match);
			nr_pointwise_match += (int)m.match.size();
			idx_added.insert(i); idx_added.insert(j);
			++topo_gen;                              // the pair / index tables below are stale from here on
		}
		void set_identity_idx(int idx) { identity_idx = idx; }

		ErrorStats get_error_stat() {
			ParamState state;
			for (auto& c : result_cameras) state.cameras.emplace_back(c);
			state.ensure_params();
			return calcError(state);
		}

		// Levenberg-Marquardt with the acceptance rule of :125-177
		void optimize() {
			if (idx_added.empty()) pano_error_exit("Calling optimize() without adding any matches!");
			ba_prof().n_opt++;
#ifdef _OPENMP
			// one parallel region for the whole call (BaTeam above): thread 0 optimizes, the others serve its sections
			BaTeam t(ba_threads());
#pragma omp parallel num_threads(ba_threads()) proc_bind(close)
			{
				if (omp_get_thread_num() == 0) {
					t.set_threads(omp_get_num_threads());
					team = &t;
					optimize_serial();
					team = nullptr;
					t.finish();
				} else t.worker_loop();
			}
#else
			optimize_serial();
#endif
		}
		double last_error = 0; int last_iterations = 0;

	protected:
		BaTeam* team = nullptr;                 // set while optimize() runs: the sections below are shared out to it
		void team_for(int n, const std::function<void(int)>& f) {
			if (team) team->run(n, f);
			else for (int i = 0; i < n; ++i) f(i);
		}
		void optimize_serial() {
			update_index_map();
			const int nr_img = (int)idx_added.size();
			JtJ.assign((size_t)(NR_PARAM_PER_CAMERA * nr_img) * (NR_PARAM_PER_CAMERA * nr_img), 0.0);
			Jtr.assign((size_t)NR_PARAM_PER_CAMERA * nr_img, 0.0);
			ParamState state;
			for (auto& idx : idx_added) state.cameras.emplace_back(result_cameras[idx]);
			state.ensure_params();
			state.cameras.clear();          // the cameras are re-derived from the parameter vector, like the reference
			ErrorStats err_stat = calcError(state);
			double best_err = err_stat.avg;
			int itr = 0, nr_non_decrease = 0;
			inlier_threshold = std::numeric_limits<int>::max();
			const size_t idt = index_map[identity_idx];
			// A rejected step leaves `state` as it was: the next iteration's Jacobian -- hence JtJ, its damping and its
			// factorization -- is the one just computed; only the residuals (those of the REJECTED trial, as in the
			// reference, :146-160) and with them J^T r changed.  `fresh` says when the state moved.  An optimize() call
			// ends with six rejected steps in a row, so five of its ~6.6 iterations re-use the factorization.
			bool fresh = true;
			while (itr++ < LM_MAX_ITER) {
				const std::vector<double> update = get_param_update(state, err_stat.residuals, config::LM_LAMBDA, fresh);
				ParamState new_state;
				new_state.params = state.get_params();
				for (size_t i = 0; i < new_state.params.size(); ++i)
					if (i < idt * 6 + 3 || i >= idt * 6 + 6) new_state.params[i] -= update[i];     // R of the identity image stays
				// (a FRESH residual vector per trial, on purpose: the team's threads have just read the previous one, and writing the new
				// residuals over it meant invalidating every line in their caches -- 12 -> 23 ms of error statistic per estimate on the
				// 256-CPU host -- while a new allocation has no sharers)
				err_stat = calcError(new_state);
				if (err_stat.avg >= best_err - 1e-3) { nr_non_decrease++; fresh = false; }
				else { nr_non_decrease = 0; best_err = err_stat.avg; state = std::move(new_state); fresh = true; }
				if (nr_non_decrease > 5) break;
			}
			last_error = best_err; last_iterations = itr;
			auto results = state.get_cameras();
			int now = 0;
			for (auto& i : idx_added) result_cameras[i] = results[now++];
		}
		static constexpr int NR_PARAM_PER_CAMERA = 6, NR_TERM_PER_MATCH = 2, LM_MAX_ITER = 100;
		std::vector<Camera>& result_cameras;
		struct MatchPair {
			int from, to;
			const MatchInfo& m;
			MatchPair(int i, int j, const MatchInfo& m): from(i), to(j), m(m) {}
		};
		int nr_pointwise_match = 0;
		int inlier_threshold = std::numeric_limits<int>::max();
		std::vector<MatchPair> match_pairs;
		int identity_idx = -1;
		std::set<int> idx_added;
		std::vector<int> index_map, match_cnt_prefix_sum;
		void update_index_map() { int cnt = 0; for (auto& i : idx_added) index_map[i] = cnt++; ++topo_gen; }

		struct ParamState {
			std::vector<Camera> cameras;
			std::vector<double> params;
			std::vector<Camera>& get_cameras() {               // :387-395
				if (cameras.size()) return cameras;
				cameras.resize(params.size() / NR_PARAM_PER_CAMERA);
				for (size_t i = 0; i < cameras.size(); ++i) params_to_camera(params.data() + i * NR_PARAM_PER_CAMERA, cameras[i]);
				return cameras;
			}
			const std::vector<Camera>& get_cameras() const { return const_cast<ParamState*>(this)->get_cameras(); }
			void ensure_params() const {                        // :397-406
				if (params.size()) return;
				std::vector<double>& p = const_cast<std::vector<double>&>(params);
				p.resize(cameras.size() * NR_PARAM_PER_CAMERA);
				for (size_t i = 0; i < cameras.size(); ++i) camera_to_params(cameras[i], p.data() + i * NR_PARAM_PER_CAMERA);
			}
			const std::vector<double>& get_params() const { ensure_params(); return params; }
		};
		static void camera_to_params(const Camera& c, double* ptr) {   // :27-32
			ptr[0] = c.focal; ptr[1] = c.ppx; ptr[2] = c.ppy;
			Camera::rotation_to_angle(c.R, ptr[3], ptr[4], ptr[5]);
		}
		static void params_to_camera(const double* ptr, Camera& c) {   // :34-40
			c.focal = ptr[0]; c.ppx = ptr[1]; c.ppy = ptr[2]; c.aspect = 1;
			Camera::angle_to_rotation(ptr[3], ptr[4], ptr[5], c.R);
		}
		static Homography cross_product_matrix(double x, double y, double z) {
			const double t[9] = {0, -z, y, z, 0, -x, -y, x, 0};
			return Homography(t);
		}
		// dR/dv_i, Gallego & Yezzi, "A compact formula for the derivative of a 3-D rotation in
		// exponential coordinates" (:48-83)
		static std::array<Homography, 3> dRdvi(const Homography& R) {
			double v[3];
			Camera::rotation_to_angle(R, v[0], v[1], v[2]);
			const Vec vvec(v[0], v[1], v[2]);
			const double vsqr = vvec.sqr();
			if (vsqr < PANO_GEO_EPS_SQR)
				return std::array<Homography, 3>{cross_product_matrix(1, 0, 0), cross_product_matrix(0, 1, 0), cross_product_matrix(0, 0, 1)};
			const Homography r = cross_product_matrix(v[0], v[1], v[2]);
			std::array<Homography, 3> ret{r, r, r};
			for (int i = 0; i < 3; ++i) ret[i].mult(v[i]);
			Vec I_R_e(1 - R.data[0], -R.data[3], -R.data[6]);
			I_R_e = vvec.cross(I_R_e);
			ret[0] += cross_product_matrix(I_R_e.x, I_R_e.y, I_R_e.z);
			I_R_e = Vec(-R.data[1], 1 - R.data[4], -R.data[7]);
			I_R_e = vvec.cross(I_R_e);
			ret[1] += cross_product_matrix(I_R_e.x, I_R_e.y, I_R_e.z);
			I_R_e = Vec(-R.data[2], -R.data[5], 1 - R.data[8]);
			I_R_e = vvec.cross(I_R_e);
			ret[2] += cross_product_matrix(I_R_e.x, I_R_e.y, I_R_e.z);
			for (int i = 0; i < 3; ++i) { ret[i].mult(1.0 / vsqr); ret[i] = ret[i] * R; }
			return ret;
		}

		std::vector<double> JtJ, Jtr;       // (6n)^2 and 6n, kept across iterations

		// (a fresh result object per call, as in the reference: the LM loop keeps the accepted and the trial statistics side by side)
		ErrorStats calcError(const ParamState& state) {            // :179-206
			ErrorStats ret(nr_pointwise_match * NR_TERM_PER_MATCH);
			const double t0 = ba_now();
			auto cameras = state.get_cameras();
			const int npairs = (int)match_pairs.size();
			// (serial: 16 us of arithmetic per call at 49 k matches, and thread 0 reads every residual right after for the
			// sequential error statistic -- sharing the loop out moved the residuals through sixteen caches: 12 -> 18-27 ms per estimate)
			auto serial_for = [](int n, const std::function<void(int)>& f) { for (int i = 0; i < n; ++i) f(i); };
			serial_for(npairs, [&](int q) {                      // independent residuals: each pair writes its own slice
				const MatchPair& pair = match_pairs[q];
				int idx = match_cnt_prefix_sum[q] * 2;
				const int from = index_map[pair.from], to = index_map[pair.to];
				auto& c_from = cameras[from]; auto& c_to = cameras[to];
				const Homography Hto_to_from = (c_from.K() * c_from.R) * (c_to.Rinv() * c_to.K().inverse());
				for (const auto& p : pair.m.match) {
					const Vec2D to2 = p.first, from2 = p.second;
					const Vec2D transformed = Hto_to_from.trans2d(to2);
					ret.residuals[idx] = from2.x - transformed.x;
					ret.residuals[idx + 1] = from2.y - transformed.y;
					idx += 2;
				}
			});
			ret.update_stats(inlier_threshold);
			ba_prof().t_err += ba_now() - t0;
			return ret;
		}

		// (JtJ + damping) x = J^T r  (:231-251)
		// fresh = false: `state` is the one of the previous call (a rejected step in between): JtJ, its damping and its
		// factorization are kept, J^T r is re-accumulated from the kept derivative rows with the new residuals
		std::vector<double> get_param_update(const ParamState& state, const std::vector<double>& residual, float lambda, bool fresh = true) {
			const int nr_img = (int)idx_added.size(), np = nr_img * NR_PARAM_PER_CAMERA;
			double t0 = ba_now();
			calcJacobianSymbolic(state, residual, fresh);
			ba_prof().t_jac += ba_now() - t0; t0 = ba_now();
			if (fresh) {
				for (int i = 0; i < np; ++i) {
					if (i % NR_PARAM_PER_CAMERA >= 3) JtJ[(size_t)i * np + i] += lambda;
					else JtJ[(size_t)i * np + i] += lambda / 10.f;
				}
				pano_la::colpiv_qr_factor(JtJ.data(), np, qr);
			}
			std::vector<double> x(np, 0.0);
			pano_la::colpiv_qr_apply(qr, Jtr.data(), x.data());
			ba_prof().t_solve += ba_now() - t0; ba_prof().n_iter++;
			return x;
		}
		pano_la::ColPivQR qr;               // factorization of the damped JtJ of the current state

		// Analytic derivatives of the residuals (Brown & Lowe, IJCV'07, section 4) -> JtJ and J^T r (:276-385)
		//
		// The reference walks the matches once and adds every match's 12 x 12 outer product into JtJ
		// as it goes.  Here the same additions are made in the same order for every entry, but in two
		// parallel phases (OpenMP): (1) the derivative rows dx[12], dy[12] of every match, independent
		// of each other, are computed into a table; (2) every 6 x 6 block of JtJ (one per camera on the
		// diagonal, one per connected camera pair off it) and every camera's slice of J^T r is owned by
		// ONE task that walks the match pairs touching it in the reference's order and accumulates its
		// entries sequentially.  An entry's chain of fp64 additions is therefore the reference's chain,
		// term for term; only which thread executes it changed.
		std::vector<double> deriv;          // 24 doubles per pointwise match: dx[0..11], dy[0..11]
		std::vector<std::vector<int>> cam_pairs;                       // per camera: match-pair indices touching it, ascending
		std::vector<std::pair<std::pair<int, int>, std::vector<int>>> block_pairs;    // per connected camera pair (i < j): match-pair indices, ascending
		// the tables are rebuilt whenever a mutator (add_match, update_index_map) ran since they were built:
		// a generation counter, not the container sizes (a replaced pair keeps the sizes and changes the tables)
		unsigned long topo_gen = 0, topo_built = (unsigned long)-1; int topo_imgs = -1;

		void update_topology(int nr_img) {
			if (topo_built == topo_gen && topo_imgs == nr_img) return;
			cam_pairs.assign(nr_img, {});
			std::vector<std::vector<int>> blk((size_t)nr_img * nr_img);
			for (size_t q = 0; q < match_pairs.size(); ++q) {
				const int f = index_map[match_pairs[q].from], t = index_map[match_pairs[q].to];
				cam_pairs[f].push_back((int)q);
				if (t != f) cam_pairs[t].push_back((int)q);
				blk[(size_t)std::min(f, t) * nr_img + std::max(f, t)].push_back((int)q);
			}
			block_pairs.clear();
			for (int i = 0; i < nr_img; ++i) for (int j = i + 1; j < nr_img; ++j)
				if (!blk[(size_t)i * nr_img + j].empty()) block_pairs.push_back({{i, j}, std::move(blk[(size_t)i * nr_img + j])});
			topo_built = topo_gen; topo_imgs = nr_img;
		}

		void calcJacobianSymbolic(const ParamState& state, const std::vector<double>& residual, bool fresh = true) {
			const int nr_img = (int)idx_added.size();
			const int np = nr_img * NR_PARAM_PER_CAMERA;
			if (!fresh) {              // same state as the previous call: the derivative rows and JtJ stand; J^T r with the new residuals
				// one task per camera (split = 1).  A three-way split of a camera's six parameters (split = 3) shortens the busiest
				// camera's pass but reads every derivative row three times: measured 57 -> 67 ms per estimate on the 256-CPU host,
				// so it is not used; the constant stays so that the measurement can be repeated
				const int split = 1, rows = 6 / split;
				team_for(split * nr_img, [&](int task) {
					const int c = task / split, a0 = rows * (task % split);
					double g[6] = {0, 0, 0, 0, 0, 0};
					for (int q : cam_pairs[c]) {
						const MatchPair& pair = match_pairs[q];
						const int nm = (int)pair.m.match.size();
						const double* row = deriv.data() + (size_t)match_cnt_prefix_sum[q] * 24;
						const double* res = residual.data() + (size_t)match_cnt_prefix_sum[q] * 2;
						const int o = (index_map[pair.from] == c ? 0 : 6) + a0;
						for (int k = 0; k < nm; ++k, row += 24, res += 2) {
							const double* dx = row + o; const double* dy = row + 12 + o;
							const double rx = res[0], ry = res[1];
							for (int a = 0; a < rows; ++a) { g[a] += dx[a] * rx; g[a] += dy[a] * ry; }
						}
					}
					for (int a = 0; a < rows; ++a) Jtr[c * NR_PARAM_PER_CAMERA + a0 + a] = g[a];
				});
				return;
			}
			// (JtJ was zeroed when optimize() sized it: every block of a connected camera pair, every diagonal block and every entry of
			// J^T r is ASSIGNED below, the blocks of unconnected pairs are never touched -- 0.4 MB of fill per call were for nothing)
			const auto& cameras = state.get_cameras();
			std::vector<std::array<Homography, 3>> all_dRdvi(cameras.size());
			// (one 3 x 3 Jacobi SVD per camera inside rotation_to_angle: 0.2-0.3 ms per fresh Jacobian when one thread does them all,
			// more than the sixteenth of either phase below that a team member gets)
			team_for((int)cameras.size(), [&](int i) { all_dRdvi[i] = dRdvi(cameras[i].R); });
			const double kf[9] = {1, 0, 0, 0, 1, 0, 0, 0, 0}, kx[9] = {0, 0, 1, 0, 0, 0, 0, 0, 0}, ky[9] = {0, 0, 0, 0, 0, 1, 0, 0, 0};
			const Homography dKdfocal(kf), dKdppx(kx), dKdppy(ky);
			update_topology(nr_img);
			deriv.resize((size_t)nr_pointwise_match * 24);
			const int npairs = (int)match_pairs.size();

			// ---- phase 1: derivative rows of every match
			team_for(npairs, [&](int pair_idx) {
				const MatchPair& pair = match_pairs[pair_idx];
				const int from = index_map[pair.from], to = index_map[pair.to];
				const auto& c_from = cameras[from]; const auto& c_to = cameras[to];
				const auto fromK = c_from.K();
				const auto toKinv = c_to.Kinv();
				const auto toRinv = c_to.Rinv();
				const auto& dRfromdvi = all_dRdvi[from];
				auto dRtodviT = all_dRdvi[to];
				for (auto& m : dRtodviT) m = m.transpose();
				const Homography Hto_to_from = (fromK * c_from.R) * (toRinv * toKinv);
				// The matrices below do not depend on the match: the reference forms them again for
				// every point (:313-343); once per pair gives the same values.
				const Homography Mfrom = c_from.R * toRinv * toKinv;                 // d/d(K_from): dK * (Mfrom p)
				const Homography Prot = toRinv * toKinv;                             // d/d(R_from): (K_from dR_i) * (Prot p)
				const Homography Bfrom[3] = {fromK * dRfromdvi[0], fromK * dRfromdvi[1], fromK * dRfromdvi[2]};
				const Homography Mto = fromK * c_from.R * toRinv * toKinv;           // d/d(K_to): (Mto dK) * (-Kinv_to p)
				const Homography Cto[3] = {Mto * dKdfocal, Mto * dKdppx, Mto * dKdppy};
				const Homography Mrot = fromK * c_from.R;                            // d/d(R_to): (Mrot dR_i^T) * (Kinv_to p)
				const Homography Dto[3] = {Mrot * dRtodviT[0], Mrot * dRtodviT[1], Mrot * dRtodviT[2]};
				double* row = deriv.data() + (size_t)match_cnt_prefix_sum[pair_idx] * 24;
				for (const auto& p : pair.m.match) {
					const Vec2D to2 = p.first;
					const Vec homo = Hto_to_from.trans(to2);
					const double hz_sqr_inv = 1.0 / pano_sqr((float)homo.z);
					const double hz_inv = 1.0 / homo.z;
					// d(residual)/d(variable) = -d(point 2d)/d(homo 3d) * d(homo 3d)/d(variable);
					// dx / dy: 0..5 = d/d(from params), 6..11 = d/d(to params)
					double* dx = row; double* dy = row + 12;
					auto drdv = [&](int k, const Vec& dhdv) {
						dx[k] = -dhdv.x * hz_inv + dhdv.z * homo.x * hz_sqr_inv;
						dy[k] = -dhdv.y * hz_inv + dhdv.z * homo.y * hz_sqr_inv;
					};
					Vec dot_u2 = Mfrom.trans(to2);
					drdv(0, dKdfocal.trans(dot_u2));
					drdv(1, dKdppx.trans(dot_u2));
					drdv(2, dKdppy.trans(dot_u2));
					dot_u2 = Prot.trans(to2);
					drdv(3, Bfrom[0].trans(dot_u2));
					drdv(4, Bfrom[1].trans(dot_u2));
					drdv(5, Bfrom[2].trans(dot_u2));
					// d(Kinv)/dv = -Kinv dK/dv Kinv
					dot_u2 = toKinv.trans(to2) * (-1);
					drdv(6, Cto[0].trans(dot_u2));
					drdv(7, Cto[1].trans(dot_u2));
					drdv(8, Cto[2].trans(dot_u2));
					dot_u2 = toKinv.trans(to2);
					drdv(9, Dto[0].trans(dot_u2));
					drdv(10, Dto[1].trans(dot_u2));
					drdv(11, Dto[2].trans(dot_u2));
					row += 24;
				}
			});

			// ---- phase 2: one task per 6 x 6 block (diagonal blocks also own their camera's J^T r slice)
			// (a diagonal block is ONE task -- rows 0..5 of the block's upper triangle and the camera's six entries of J^T r; see
			// above: splitting it over three tasks did not pay)
			const int split = 1, rows = 6 / split;
			const int ndiag = split * nr_img, noff = (int)block_pairs.size();
			team_for(ndiag + noff, [&](int task) {
				if (task < ndiag) {
					const int c = task / split, a0 = rows * (task % split), a1 = a0 + rows;
					double L[6][6] = {{0}}, g[6] = {0, 0, 0, 0, 0, 0};
					for (int q : cam_pairs[c]) {
						const MatchPair& pair = match_pairs[q];
						const int from = index_map[pair.from], to = index_map[pair.to];
						const int nm = (int)pair.m.match.size();
						const double* row = deriv.data() + (size_t)match_cnt_prefix_sum[q] * 24;
						const double* res = residual.data() + (size_t)match_cnt_prefix_sum[q] * 2;
						// a pair from a camera to itself does not occur (add_match is called for i != j); the
						// offsets below are the local indices the reference's 12 x 12 block gives this camera
						const int o = from == c ? 0 : 6;
						(void)to;
						for (int k = 0; k < nm; ++k, row += 24, res += 2) {
							const double* dx = row + o; const double* dy = row + 12 + o;
							const double rx = res[0], ry = res[1];
							for (int a = a0; a < a1; ++a) { g[a] += dx[a] * rx; g[a] += dy[a] * ry; }
							for (int a = a0; a < a1; ++a)
								for (int b = a; b < 6; ++b) L[a][b] += dx[a] * dx[b] + dy[a] * dy[b];
						}
					}
					const int base = c * NR_PARAM_PER_CAMERA;
					for (int a = a0; a < a1; ++a) {
						Jtr[base + a] = g[a];
						for (int b = a; b < 6; ++b) { JtJ[(size_t)(base + a) * np + base + b] = L[a][b]; JtJ[(size_t)(base + b) * np + base + a] = L[a][b]; }
					}
				} else {
					const auto& blk = block_pairs[task - ndiag];
					const int ci = blk.first.first, cj = blk.first.second;          // ci < cj
					double L[6][6] = {{0}};                                          // L[p][q]: entry (param p of ci, param q of cj)
					for (int q : blk.second) {
						const MatchPair& pair = match_pairs[q];
						const int from = index_map[pair.from];
						const int nm = (int)pair.m.match.size();
						const double* row = deriv.data() + (size_t)match_cnt_prefix_sum[q] * 24;
						if (from == ci) {            // local (a, b) = (p, 6 + q'): dx[p] * dx[6 + q'] + dy[p] * dy[6 + q']
							for (int k = 0; k < nm; ++k, row += 24) {
								const double* dx = row; const double* dy = row + 12;
								for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) L[a][b] += dx[a] * dx[6 + b] + dy[a] * dy[6 + b];
							}
						} else {                     // from == cj: local (a, b) = (q', 6 + p): dx[q'] * dx[6 + p] + dy[q'] * dy[6 + p]
							for (int k = 0; k < nm; ++k, row += 24) {
								const double* dx = row; const double* dy = row + 12;
								for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) L[a][b] += dx[b] * dx[6 + a] + dy[b] * dy[6 + a];
							}
						}
					}
					const int bi = ci * NR_PARAM_PER_CAMERA, bj = cj * NR_PARAM_PER_CAMERA;
					for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) {
						JtJ[(size_t)(bi + a) * np + bj + b] = L[a][b];
						JtJ[(size_t)(bj + b) * np + bi + a] = L[a][b];
					}
				}
			});
		}
};

class CameraEstimator {
	public:
		CameraEstimator(std::vector<std::vector<MatchInfo>>& matches, const std::vector<Shape2D>& image_shapes):
			n((int)matches.size()), matches(matches), shapes(image_shapes), cameras(matches.size()) {}
		CameraEstimator(const CameraEstimator&) = delete;
		CameraEstimator& operator=(const CameraEstimator&) = delete;

		std::vector<Camera> estimate() {                        // camera_estimator.cc:47-103
			estimate_focal();
			IncrementalBundleAdjuster iba(cameras);
			std::vector<bool> vst(n, false);
			traverse(
				[&](int node) {
					cameras[node].R = Homography::I();
					cameras[node].ppx = cameras[node].ppy = 0;
					iba.set_identity_idx(node);
				},
				[&](int now, int next) {
					const auto Kfrom = cameras[now].K();
					const auto Kto = cameras[next].K();
					const auto Hinv = matches[now][next].homo;          // from next to now
					const auto Mat = Kfrom.inverse() * Hinv * Kto;
					cameras[next].R = (cameras[now].Rinv() * Mat).transpose();
					cameras[next].ppx = cameras[next].ppy = 0;
					if (config::MULTIPASS_BA > 0) {
						vst[now] = vst[next] = true;
						for (int i = 0; i < n; ++i) if (vst[i] && i != next) {
							const auto& m = matches[next][i];
							if (m.match.size() && m.confidence > 0) {
								iba.add_match(i, next, m);
								if (config::MULTIPASS_BA == 2) iba.optimize();
							}
						}
						if (config::MULTIPASS_BA == 1) iba.optimize();
					}
				});
			if (config::MULTIPASS_BA == 0) {
				for (int i = 1; i < n; ++i) for (int j = 0; j < i; ++j) {
					auto& m = matches[j][i];
					if (m.match.size() && m.confidence > 0) iba.add_match(i, j, m);
				}
				iba.optimize();
			}
			if (config::STRAIGHTEN) Camera::straighten(cameras);
			if (std::getenv("PANO_BA_PROFILE")) {
				const BaProfile& bp = ba_prof();
				std::fprintf(stderr, "[pano BA] optimize calls %ld, LM iterations %ld: error %.1f ms, jacobian %.1f ms, solve %.1f ms\n",
						bp.n_opt, bp.n_iter, bp.t_err * 1e3, bp.t_jac * 1e3, bp.t_solve * 1e3);
			}
			return cameras;
		}

	protected:
		int n;
		std::vector<std::vector<MatchInfo>>& matches;
		const std::vector<Shape2D>& shapes;
		std::vector<Camera> cameras;

		void estimate_focal() {                                  // :33-45
			const double focal = Camera::estimate_focal(matches);
			if (focal > 0) { for (auto& c : cameras) c.focal = focal; }
			else for (int i = 0; i < n; ++i) cameras[i].focal = (shapes[i].w + shapes[i].h) * 0.5;
		}

		// maximum spanning tree by confidence, grown from the best edge (:105-158)
		void traverse(std::function<void(int)> callback_init_node, std::function<void(int, int)> callback_edge) {
			struct Edge {
				int v1, v2; float weight;
				Edge(int a, int b, float v): v1(a), v2(b), weight(v) {}
				bool operator<(const Edge& r) const { return weight < r.weight; }
			};
			Edge best_edge{-1, -1, 0};
			for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) {
				auto& m = matches[i][j];
				if (m.confidence > best_edge.weight) best_edge = Edge{i, j, m.confidence};
			}
			if (best_edge.v1 == -1) pano_error_exit("No connected images are found!");
			callback_init_node(best_edge.v1);
			std::priority_queue<Edge> q;
			std::vector<bool> vst(n, false);
			auto enqueue_edges_from = [&](int from) {
				for (int i = 0; i < n; ++i) if (i != from && !vst[i]) {
					auto& m = matches[from][i];
					if (m.confidence > 0) q.emplace(from, i, m.confidence);
				}
			};
			vst[best_edge.v1] = true;
			enqueue_edges_from(best_edge.v1);
			int cnt = 1;
			while (q.size()) {
				do { best_edge = q.top(); q.pop(); } while (q.size() && vst[best_edge.v2]);
				if (vst[best_edge.v2]) break;
				vst[best_edge.v2] = true;
				cnt++;
				callback_edge(best_edge.v1, best_edge.v2);
				enqueue_edges_from(best_edge.v2);
			}
			if (cnt != n) {
				std::string unconnected;
				for (int i = 0; i < n; ++i) if (!vst[i]) unconnected += std::to_string(i) + " ";
				pano_error_exit("Found a tree of size " + std::to_string(cnt) + "!=" + std::to_string(n) + ", image " + unconnected + "are not connected well!");
			}
		}
};

}	// namespace pano
