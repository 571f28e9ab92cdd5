// internal.hpp -- shared host/device declarations of libopenpano_hip.so (not part of the ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "openpano_hip.h"

#define OP_MAX_OCT 8
#define OP_MAX_SCALE 12      // NUM_SCALE <= 12
#define OP_MAX_KCENTER 15    // Gaussian kernels up to 31 taps

void op_set_error(const std::string& msg);
struct op_ctx;
void op_ctx_release_workspace(op_ctx* c);
#define OP_FAIL(code, msg) do { op_set_error(msg); return (code); } while (0)
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	op_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return OP_ERR_HIP; } } while (0)

// per-stage device timing (HIP events on the context's stream), the counterpart of the
// reference's TotalTimer table (lib/timer.hh:63-83); labels reuse the reference's where one exists
struct ProfStage { std::string label; double total_ms = 0; long calls = 0; };
struct op_ctx {
	int device = 0;
	int num_cu = 256;                                // compute units of the device (sizes the persistent launches)
	hipStream_t stream = nullptr;
	bool owns_stream = false;
	bool profiling = false;
	std::string prof_only;                           // when not empty: only this stage is bracketed by events
	int match_slow_seen[2] = {-1, -1};               // exact-scan rows (forward, reverse) of the previous op_match_pairs call: sizes the next launch
	std::vector<ProfStage> prof;
	std::vector<hipEvent_t> ev_pool;                 // recycled events
	struct Pending { int stage; hipEvent_t a, b; };
	std::vector<Pending> pending;                    // recorded, not yet resolved
	void* pinned = nullptr; size_t pinned_cap = 0;   // grow-only pinned host scratch for D2H results
	// grow-only device scratch of the matcher and of RANSAC: a call's temporaries live here, so kernels that are
	// still queued when the call returns (the per-pair sort into the result buffer) keep valid inputs -- the next
	// call on this context is ordered behind them on the same stream (contexts are thread-compatible)
	struct DevScratch {
		void* p = nullptr; size_t cap = 0;
		hipError_t ensure(size_t bytes) {
			if (bytes <= cap) return hipSuccess;
			if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }   // hipFree waits for the device
			const size_t want = bytes + bytes / 8;
			hipError_t e = hipMalloc(&p, want);
			if (e != hipSuccess) return e;
			cap = want;
			return hipSuccess;
		}
		void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
	};
	DevScratch match_arena, ransac_arena;
	// blend: the canvas -> space map's transcendentals, tabulated per canvas column / row by the HOST libm (blend.hip,
	// trig_tables): kept across calls with the same canvas geometry
	struct TrigTables {
		int method = -1, w1 = 0, h1 = 0; double minx = 0, miny = 0, resx = 0, resy = 0;
		DevScratch dev; std::vector<double> host;
	} blend_trig, cyl_trig;
	// copy streams of the host-image pipeline (op_sift_batch_host): uploads run ahead of the kernels, results leave behind them
	hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
	// second compute stream of op_ransac_pairs (the pairs with few matches, whose sample tables take longest, run beside
	// the others) and the two events that fork it from / join it to the context's stream
	hipStream_t aux_stream = nullptr; hipEvent_t aux_fork = nullptr, aux_join = nullptr;
	hipError_t aux() {
		if (!aux_stream) { hipError_t e = hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking); if (e != hipSuccess) return e; }
		if (!aux_fork) { hipError_t e = hipEventCreateWithFlags(&aux_fork, hipEventDisableTiming); if (e != hipSuccess) return e; }
		if (!aux_join) { hipError_t e = hipEventCreateWithFlags(&aux_join, hipEventDisableTiming); if (e != hipSuccess) return e; }
		return hipSuccess;
	}
	hipError_t copy_streams() {
		if (!h2d_stream) { hipError_t e = hipStreamCreateWithFlags(&h2d_stream, hipStreamNonBlocking); if (e != hipSuccess) return e; }
		if (!d2h_stream) { hipError_t e = hipStreamCreateWithFlags(&d2h_stream, hipStreamNonBlocking); if (e != hipSuccess) return e; }
		return hipSuccess;
	}
	void* pinned_scratch(size_t bytes) {
		if (bytes <= pinned_cap) return pinned;
		if (pinned) hipHostFree(pinned);
		pinned = nullptr; pinned_cap = 0;
		if (hipHostMalloc(&pinned, bytes + bytes / 4) != hipSuccess) return nullptr;
		pinned_cap = bytes + bytes / 4;
		return pinned;
	}
	int prof_stage(const std::string& label) {
		for (size_t i = 0; i < prof.size(); ++i) if (prof[i].label == label) return (int)i;
		prof.push_back(ProfStage{label, 0, 0});
		return (int)prof.size() - 1;
	}
};
// RAII bracket around one stage's launches; resolve_profile() after a stream sync
struct ProfScope {
	op_ctx* c; int stage = -1; hipEvent_t a = nullptr, b = nullptr;
	ProfScope(op_ctx* ctx, const char* label);
	~ProfScope();
};
void resolve_profile(op_ctx* c);
// host-side wall time of a stage, reported in the same table (labels end in " (host)")
struct HostScope {
	op_ctx* c; const char* label; double t0;
	HostScope(op_ctx* ctx, const char* l);
	~HostScope();
};

// Host-side parallel loop over [0, n): a persistent pool of std::threads (the process may host
// several OpenMP runtimes -- PyTorch's, the reference's -- whose interplay cannot be relied on).
#include <functional>
// (`grain`: items a woken thread should find -- short items are not worth a thread each)
void host_parallel_for(int n, const std::function<void(int)>& body, int grain = 1);

// Size-class cache of device allocations (per device, process-wide, thread-safe): result buffers
// (op_features, op_canvas) and per-call temporaries come from here, so a steady-state call does
// no hipMalloc/hipFree (each costs tens of microseconds and synchronises the device).
hipError_t pool_alloc(void** p, size_t bytes);
void pool_free(void* p);
void pool_trim();     // release every cached block back to the runtime

// ---------------------------------------------------------------------------------------
// HBM layout of one image's scale space ("image workspace", ws_stride floats per image):
//   for each octave o:  [G 0 = grey][G 1 .. G ns-1]   each h_o*w_o fp32, row-major, octave blocks back to back
// -- the reference's Gaussian stack (feature/dog.cc:53-57: data[0] is the unblurred grey image) and NOTHING else.
// The DoG planes (dog.cc:116-129) are never materialised: the extrema scan runs on them while they are in LDS
// (pyramid.hip), and the sub-pixel refinement -- the only other reader, a few hundred sparse 3x3x3 neighbourhoods per
// image -- evaluates |G[l] - G[l+1]| (dog.cc:126) on the two Gaussian planes, the same fp32 operation on the same
// operands: 28 bytes per octave pixel cross the HBM boundary in the scale-space kernel instead of 44.  The mag/ort
// planes of GaussianPyramid::cal_mag_ort (dog.cc:60-94) are not materialised either -- the orientation and
// descriptor kernels evaluate the same expressions on the Gaussian plane for exactly the window samples they use.
// ---------------------------------------------------------------------------------------
struct OctDesc {
	int h, w;
	int tiles_x, tiles_y, tile_begin;   // pyramid-kernel tiling (k_pyramid)
	int rw_nb, rw_nseg;                 // k_pyramid_rows: bands x row segments of this octave
	long long off;                      // float offset of the octave block in the image workspace
	long long plane;                    // h*w
};

struct SiftPlan {
	int n;                      // images in the batch (all sh x sw)
	int sh, sw;                 // source size
	int wh, ww;                 // working size (feature.cc:33-35)
	int noct, nscale;
	OctDesc oct[OP_MAX_OCT];
	int total_tiles;
	long long ws_stride;        // floats per image workspace
	float* ws;
	float* work;                // n x wh x ww x 3 (only materialised for the staged dump)
	const void* const* srcs;    // device array of n source pointers (device memory)
	int src_u8;                 // 0: fp32 sources, 1: uint8 sources (converted like read_img, lib/imgio.cc:54-56)
	int* zero;                  // batch counters cleared by the first kernel of the step (k_grey_octaves), zero_n ints
	int zero_n;
	int num_cu;                 // compute units of the device
	// Gaussian bank (feature/gaussian.cc:17-40): kern[s][center + k], s = 1..nscale-1
	float kern[OP_MAX_SCALE][2 * OP_MAX_KCENTER + 1];
	int kcenter[OP_MAX_SCALE];
	int halo;                   // max kcenter
	// Row-streaming scale-space kernel (k_pyramid_rows): available when the bank is the shipped one
	// (7 scales, half-widths 3,3,3,6,6,6).  kpair[pl][d] = taps at distance d from the centre of the
	// two sigmas (2 pl + 1, 2 pl + 2) that share one packed accumulator; a shorter kernel is
	// zero-extended (adding +-0 leaves an fp32 sum unchanged, so the padding is exact).
	int rows_ok;
	int rw_seg;                 // rows of a segment (chosen per batch: the launch should be a whole number of device fills, sift_host.hip)
	int rw_items;               // work items per image: sum over octaves of bands x segments
	float kpair[3][7][2];
	// thresholds
	float pre_color_thres, judge_thres, contrast_thres, edge_ratio, offset_thres;
	int calc_offset_depth;
	float gauss_sigma, scale_factor, ori_radius;
	int ori_smooth, desc_scale_factor, desc_int_factor;
	int desc_list_cap;          // k_descriptor: floats of list arena one sorting pass may use (the arena's size; a test lowers it to force the two-pass path)
};

__host__ __device__ inline long long plane_off_grey(const OctDesc& o) { return o.off; }
__host__ __device__ inline long long plane_off_gauss(const OctDesc& o, int ns, int s) { (void)ns; return o.off + (long long)s * o.plane; }   // s = 0: grey
__host__ __device__ inline int planes_per_octave(int ns) { return ns; }

// a scale-space point (feature/feature.hh:33-39), 48 bytes
struct KeyPoint {
	int x, y, oct, scale;
	double rx, ry;          // real_coor in [0,1)
	float dir, sf;          // dir, scale_factor
	int src;                // index of the raw candidate / refined parent
	int pad;
};

#define OP_DESC_LIST_CAP 640   // k_descriptor: floats in the list arena of one batch's counting sort
#define OP_RW_OWN 240     // k_pyramid_rows: columns owned by a band
// rows of a segment: chosen per batch between OP_RW_SEG_MIN and OP_RW_SEG_MAX (a build with -DOP_RW_SEG=<rows> pins it: A/B runs)
#define OP_RW_SEG_DEFAULT 24
#define OP_RW_SEG_MIN 16
#define OP_RW_SEG_MAX 40
#define OP_PYR_TW 64
#ifndef OP_PYR_TH
#define OP_PYR_TH 16
#endif

// ---- kernel launchers (each returns hipGetLastError of its launch) ----
// source -> grey base of every octave (working image only written when write_work: staged dump)
hipError_t launch_grey_octaves(const SiftPlan& p, bool write_work, hipStream_t st);
// fused scale space + extrema scan: fills the DoG / Gaussian planes and appends raw extrema
hipError_t launch_pyramid(const SiftPlan& p, int* raw /* n x cap x 4 */, int* raw_count /* n */, int cap, hipStream_t st);
// debug/staged dump only: mag and ort of one Gaussian plane (GaussianPyramid::cal_mag_ort)
hipError_t launch_magort_plane(const SiftPlan& p, int img, int oct, int s, float* mag, float* ort, hipStream_t st);
// expect = the largest per-image list length seen in the previous batch (0: unknown).  Workgroups cost dispatcher time
// whether they find work or not (k_refine over the whole 16 K capacity: 4864 workgroups for 30 busy ones per image took
// 35 us more than a grid sized to the lists), so k_refine, the ranking wavefronts of k_orientation and k_orient_peaks get as
// many workgroups as the expectation needs; all of them stride over their lists, any grid is correct.
hipError_t launch_refine(const SiftPlan& p, const int* raw, const int* raw_count, int cap, int expect,
		KeyPoint* refined /* n x cap */, int* refined_count, hipStream_t st);
// per_image[img * OP_OCNT_STRIDE] += orientation peaks of every keypoint (atomic; one counter per 128-byte line; cleared at the start of the step)
#define OP_OCNT_STRIDE 32
hipError_t launch_orientation(const SiftPlan& p, const KeyPoint* refined, const int* refined_count, int cap, int expect,
		float* dirs /* n x cap x 36: histogram in, peak directions out */, int* ndirs /* n x cap */, int* per_image /* n x OP_OCNT_STRIDE */,
		KeyPoint* sorted /* n x cap: refined in the canonical order */, int* slot_of /* n x cap: sorted position -> slot in refined */, hipStream_t st);
// image img's keypoints land at [sum of the earlier images' counts, ...); *total = sum of all, count_out[0..n) = the counts packed
// (device-side, no host round trip)
hipError_t launch_expand_oriented(const SiftPlan& p, const KeyPoint* sorted, const int* slot_of, const int* refined_count, int cap,
		const float* dirs, const int* ndirs, const int* per_image /* n x OP_OCNT_STRIDE */, long long* total, int* count_out /* n */,
		KeyPoint* oriented, long long oriented_cap, hipStream_t st);
// the descriptor count is read on the device (*total); cap = capacity of the output buffers
hipError_t launch_descriptor(const SiftPlan& p, const KeyPoint* oriented, const long long* total /* device */,
		long long cap, float* desc, double* coor, double* real, hipStream_t st);
hipError_t launch_debug_math(int which, const float* x, const float* y, int n, float* out, hipStream_t st);
