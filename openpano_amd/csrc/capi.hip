// capi.hip -- context, errors, configuration defaults of the C-ABI (include/openpano_hip.h)
#include "internal.hpp"
#include <cstring>

static thread_local std::string g_last_error;
void op_set_error(const std::string& msg) { g_last_error = msg; }

static hipEvent_t take_event(op_ctx* c) {
	if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
	hipEvent_t e = nullptr;
	if (hipEventCreate(&e) != hipSuccess) return nullptr;
	return e;
}
ProfScope::ProfScope(op_ctx* ctx, const char* label): c(ctx) {
	if (!c->profiling || (!c->prof_only.empty() && c->prof_only != label)) return;
	stage = c->prof_stage(label);
	a = take_event(c); b = take_event(c);
	if (a) hipEventRecord(a, c->stream);
}
ProfScope::~ProfScope() {
	if (stage < 0 || !a || !b) return;
	hipEventRecord(b, c->stream);
	c->pending.push_back({stage, a, b});
}
#include <chrono>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
HostScope::HostScope(op_ctx* ctx, const char* l): c(ctx), label(l), t0(0) { if (c->profiling && c->prof_only.empty()) t0 = now_ms(); }
HostScope::~HostScope() {
	if (!c->profiling || !c->prof_only.empty()) return;
	const int st = c->prof_stage(label);
	c->prof[st].total_ms += now_ms() - t0; c->prof[st].calls += 1;
}
void resolve_profile(op_ctx* c) {
	for (auto& p : c->pending) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { c->prof[p.stage].total_ms += ms; c->prof[p.stage].calls += 1; }
		c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b);
	}
	c->pending.clear();
}

// ---- host thread pool ----
#include "host_pool.hpp"
using ophost::host_pool;
void host_parallel_for(int n, const std::function<void(int)>& body, int grain) {
	if (n <= 0) return;
	if (n == 1) { body(0); return; }
	host_pool().run(n, body, grain < 1 ? 1 : grain);
}

// ---- device allocation cache ----
#include <map>
#include <mutex>
namespace {
struct PoolState {
	std::mutex mu;
	std::multimap<std::pair<int, size_t>, void*> free_blocks;     // (device, size) -> block
	std::map<void*, std::pair<int, size_t>> live;                  // block -> (device, size)
	size_t cached_bytes = 0;
};
PoolState& pool() { static PoolState* p = new PoolState; return *p; }   // leaked on purpose: outlives static destructors
size_t size_class(size_t b) {
	if (b < 4096) return 4096;
	size_t c = 4096;
	while (c < b) c += (c >= (1u << 20) ? c / 4 : c);   // x2 up to 1 MiB, then +25 % steps
	return c;
}
}	// namespace

hipError_t pool_alloc(void** out, size_t bytes) {
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	const size_t cls = size_class(bytes ? bytes : 1);
	PoolState& P = pool();
	{
		std::lock_guard<std::mutex> lk(P.mu);
		auto it = P.free_blocks.find(std::make_pair(dev, cls));
		if (it != P.free_blocks.end()) {
			*out = it->second; P.free_blocks.erase(it); P.cached_bytes -= cls;
			P.live[*out] = std::make_pair(dev, cls);
			return hipSuccess;
		}
	}
	e = hipMalloc(out, cls);
	if (e != hipSuccess) {      // out of memory: give the cache back and retry once
		pool_trim();
		e = hipMalloc(out, cls);
		if (e != hipSuccess) return e;
	}
	std::lock_guard<std::mutex> lk(P.mu);
	P.live[*out] = std::make_pair(dev, cls);
	return hipSuccess;
}

void pool_free(void* p) {
	if (!p) return;
	PoolState& P = pool();
	std::lock_guard<std::mutex> lk(P.mu);
	auto it = P.live.find(p);
	if (it == P.live.end()) { hipFree(p); return; }
	P.free_blocks.insert(std::make_pair(it->second, p));
	P.cached_bytes += it->second.second;
	P.live.erase(it);
}

void pool_trim() {
	PoolState& P = pool();
	std::lock_guard<std::mutex> lk(P.mu);
	for (auto& kv : P.free_blocks) hipFree(kv.second);
	P.free_blocks.clear(); P.cached_bytes = 0;
}

extern "C" {

int op_ctx_set_profiling(op_ctx* c, int enable) {
	if (!c) OP_FAIL(OP_ERR_INVALID, "op_ctx_set_profiling: NULL context");
	c->profiling = enable != 0;
	return OP_OK;
}
int op_ctx_profile_only(op_ctx* c, const char* label) {
	if (!c) OP_FAIL(OP_ERR_INVALID, "op_ctx_profile_only: NULL context");
	c->prof_only = label ? label : "";
	return OP_OK;
}
int op_ctx_profile_reset(op_ctx* c) {
	if (!c) OP_FAIL(OP_ERR_INVALID, "op_ctx_profile_reset: NULL context");
	HIPCHK(hipStreamSynchronize(c->stream));
	resolve_profile(c);
	c->prof.clear();
	return OP_OK;
}
int op_ctx_profile_count(op_ctx* c) {
	if (!c) return 0;
	if (hipStreamSynchronize(c->stream) == hipSuccess) resolve_profile(c);
	return (int)c->prof.size();
}
int op_ctx_profile_get(op_ctx* c, int i, const char** label, double* total_ms, long* calls) {
	if (!c || i < 0 || i >= (int)c->prof.size()) OP_FAIL(OP_ERR_INVALID, "op_ctx_profile_get: bad index");
	if (label) *label = c->prof[i].label.c_str();
	if (total_ms) *total_ms = c->prof[i].total_ms;
	if (calls) *calls = c->prof[i].calls;
	return OP_OK;
}

const char* op_last_error(void) { return g_last_error.c_str(); }
int op_abi_version(void) { return 8; }      // 8: op_matches_concat; 3: op_blend_image.mat_h / mat_w; 4: resident match lists, op_sift_batch_host, op_ransac_pairs_multi; 5: op_ctx_profile_only; 6: op_debug_set_desc_list_cap; 7: op_pairwise_table

void op_config_default(op_config* c) {
	// src/config.cfg (every literal goes through a float, lib/config.cc:19-26)
	memset(c, 0, sizeof(*c));
	c->SIFT_WORKING_SIZE = 800; c->NUM_OCTAVE = 4; c->NUM_SCALE = 7;
	c->SCALE_FACTOR = 1.4142135623f; c->GAUSS_SIGMA = 1.4142135623f; c->GAUSS_WINDOW_FACTOR = 6;
	c->JUDGE_EXTREMA_DIFF_THRES = 2e-3f; c->CONTRAST_THRES = 4e-2f; c->PRE_COLOR_THRES = 5e-2f;
	c->EDGE_RATIO = 6.f; c->CALC_OFFSET_DEPTH = 4; c->OFFSET_THRES = 0.5f;
	c->ORI_RADIUS = 4.5f; c->ORI_HIST_SMOOTH_COUNT = 2;
	c->DESC_HIST_SCALE_FACTOR = 3; c->DESC_INT_FACTOR = 512;
	c->MATCH_REJECT_NEXT_RATIO = 0.8f;
	c->RANSAC_ITERATIONS = 1500; c->RANSAC_INLIER_THRES = (double)3.5f;
	c->INLIER_IN_MATCH_RATIO = 0.1f; c->INLIER_IN_POINTS_RATIO = 0.04f;
	c->CYLINDER = 0; c->TRANS = 0; c->ESTIMATE_CAMERA = 1; c->ORDERED_INPUT = 0; c->LAZY_READ = 1;
	c->MULTIBAND = 0; c->MAX_OUTPUT_SIZE = 8000; c->FOCAL_LENGTH = 37.f;
}

int op_ctx_create(int device, void* hip_stream, op_ctx** out) {
	if (!out) OP_FAIL(OP_ERR_INVALID, "op_ctx_create: out is NULL");
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		OP_FAIL(OP_ERR_HIP, std::string("op_ctx_create: no HIP device available (") + hipGetErrorString(e) +
				"); libopenpano_hip has no CPU fallback");
	if (device < 0 || device >= ndev) OP_FAIL(OP_ERR_INVALID, "op_ctx_create: bad device index");
	HIPCHK(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, device));
	if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
		OP_FAIL(OP_ERR_UNSUPPORTED, std::string("op_ctx_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
	op_ctx* c = new op_ctx;
	c->device = device;
	c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->owns_stream = false; }
	else { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->owns_stream = true; }
	*out = c;
	return OP_OK;
}

void op_ctx_destroy(op_ctx* c) {
	if (!c) return;
	hipSetDevice(c->device);
	hipStreamSynchronize(c->stream);
	op_ctx_release_workspace(c);
	c->match_arena.release(); c->ransac_arena.release();
	c->blend_trig.dev.release(); c->cyl_trig.dev.release();
	pool_trim();
	resolve_profile(c);
	for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
	if (c->pinned) hipHostFree(c->pinned);
	if (c->aux_stream) hipStreamDestroy(c->aux_stream);
	if (c->aux_fork) hipEventDestroy(c->aux_fork);
	if (c->aux_join) hipEventDestroy(c->aux_join);
	if (c->h2d_stream) hipStreamDestroy(c->h2d_stream);
	if (c->d2h_stream) hipStreamDestroy(c->d2h_stream);
	if (c->owns_stream) hipStreamDestroy(c->stream);
	delete c;
}

int op_ctx_sync(op_ctx* c) {
	if (!c) OP_FAIL(OP_ERR_INVALID, "op_ctx_sync: NULL context");
	HIPCHK(hipStreamSynchronize(c->stream));
	return OP_OK;
}

int op_debug_math(op_ctx* c, int which, const float* x, const float* y, int n, float* out) {
	if (!c || !x || !out || n < 0 || which < 0 || which > 4) OP_FAIL(OP_ERR_INVALID, "op_debug_math: bad argument");
	if (n == 0) return OP_OK;
	HIPCHK(hipSetDevice(c->device));
	float *dx = nullptr, *dy = nullptr, *dout = nullptr;
	HIPCHK(hipMalloc(&dx, sizeof(float) * n));
	HIPCHK(hipMalloc(&dy, sizeof(float) * n));
	HIPCHK(hipMalloc(&dout, sizeof(float) * n));
	HIPCHK(hipMemcpyAsync(dx, x, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(dy, y ? y : x, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
	HIPCHK(launch_debug_math(which, dx, dy, n, dout, c->stream));
	HIPCHK(hipMemcpyAsync(out, dout, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	hipFree(dx); hipFree(dy); hipFree(dout);
	return OP_OK;
}

}	// extern "C"
