/* include/openpano_hip.h -- C-ABI of libopenpano_hip.so, the MI355X (gfx950) implementation of
 * OpenPano's data-parallel hot path.
 *
 * The reference has no FFI: its seam is the C++ class surface.  Each entry point below names
 * the reference interface it stands in for (file:line under /root/reference/src); the C++
 * adapters in openpano_amd/host/pano_hip.hh re-expose the reference's own class names
 * (FeatureDetector::detect_feature, PairWiseMatcher::match, ...) on top of these calls, and
 * INTEGRATION.md shows the binding a maintainer would add to the reference tree.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types cross this boundary;
 *   - every function returns OP_OK (0) or a negative op_status; op_last_error() gives the
 *     message of the calling thread's last failure (the adapters turn it into the reference's
 *     error_exit(), lib/debugutils.cc:57-60); no exception crosses the ABI;
 *   - images are the reference's Mat32f layout (lib/mat.h:8-57): row-major H x W x 3 fp32 in
 *     [0,1]; a pointer may be host or device memory (flag on_device);
 *   - there is NO CPU fallback: without a gfx950 device every compute entry point fails.
 */
#ifndef OPENPANO_HIP_H
#define OPENPANO_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	OP_OK = 0,
	OP_ERR_INVALID = -1,      /* bad argument */
	OP_ERR_HIP = -2,          /* a HIP runtime call failed (message has the hipError string) */
	OP_ERR_CAPACITY = -3,     /* a device-side work list overflowed its capacity */
	OP_ERR_UNSUPPORTED = -4,  /* configuration outside what the kernels implement */
	OP_ERR_NO_FEATURE = -5    /* an image produced zero features (stitcherbase.cc:20-21) */
} op_status;

const char* op_last_error(void);
/* library/ABI version, bumped on incompatible change */
int op_abi_version(void);

/* ---- configuration: POD snapshot of namespace config (lib/config.hh:24-85), taken by the
 * adapters at every call because the CLI fills the globals after static init (main.cc:237-292) */
typedef struct op_config {
	/* SIFT (config.cfg:19-41) */
	int SIFT_WORKING_SIZE, NUM_OCTAVE, NUM_SCALE;
	float SCALE_FACTOR, GAUSS_SIGMA;
	int GAUSS_WINDOW_FACTOR;
	float JUDGE_EXTREMA_DIFF_THRES, CONTRAST_THRES, PRE_COLOR_THRES, EDGE_RATIO;
	int CALC_OFFSET_DEPTH;
	float OFFSET_THRES;
	float ORI_RADIUS;
	int ORI_HIST_SMOOTH_COUNT;
	int DESC_HIST_SCALE_FACTOR, DESC_INT_FACTOR;
	/* matching / RANSAC (config.cfg:45-54) */
	float MATCH_REJECT_NEXT_RATIO;
	int RANSAC_ITERATIONS;
	double RANSAC_INLIER_THRES;
	float INLIER_IN_MATCH_RATIO, INLIER_IN_POINTS_RATIO;
	/* modes that change kernel behaviour (config.cfg:2-10,69) */
	int CYLINDER, TRANS, ESTIMATE_CAMERA, ORDERED_INPUT, LAZY_READ, MULTIBAND;
	int MAX_OUTPUT_SIZE;
	float FOCAL_LENGTH;
} op_config;
/* shipped defaults, src/config.cfg */
void op_config_default(op_config* cfg);

/* ---- context: one per (process, device); owns the stream and the HBM workspaces.
 * stream = NULL creates a private hipStream; otherwise the caller's hipStream_t is used (e.g.
 * torch.cuda.current_stream().cuda_stream) and all work is ordered on it. Thread-compatible:
 * concurrent calls need distinct contexts (StitcherBase::calc_feature's omp loop,
 * stitcherbase.cc:14, maps to ONE batched call instead). */
typedef struct op_ctx op_ctx;
int op_ctx_create(int device, void* hip_stream, op_ctx** out);
void op_ctx_destroy(op_ctx* ctx);
int op_ctx_sync(op_ctx* ctx);
/* Per-stage device timing with HIP events on the context's stream -- the counterpart of the
 * reference's TotalTimer table (lib/timer.hh:63-83, dumped at exit by main.cc:336); stage labels
 * reuse the reference's ("build pyramid", "extrema", "sift descriptor", "matcher", ...). */
int op_ctx_set_profiling(op_ctx* ctx, int enable);
/* Restrict the timing to ONE stage label (NULL or "": all stages).  Every bracketed stage costs two event records and
 * the gap they put between kernels (measured: 0.05 ms on the 1.05 ms SIFT step with five stages bracketed); a caller
 * that times its own loop and wants one kernel's duration from inside it brackets that kernel only. */
int op_ctx_profile_only(op_ctx* ctx, const char* label);
int op_ctx_profile_reset(op_ctx* ctx);
int op_ctx_profile_count(op_ctx* ctx);
int op_ctx_profile_get(op_ctx* ctx, int i, const char** label, double* total_ms, long* calls);

typedef struct op_image {
	const void* data;    /* H x W x 3, row-major interleaved RGB; element type per dtype */
	int h, w;
	int on_device;       /* 0: host pointer (copied H2D inside the call), 1: device pointer.  Host images of a batch that lie at
	                      * one constant stride >= their size (a contiguous array of frames, a decoder pool) go up in ONE copy;
	                      * pinned memory copies at the link rate */
	int dtype;           /* OP_F32 (0): fp32 in [0,1] = Mat32f ; OP_U8 (1): decoder bytes, converted on the
	                      * device exactly like read_img does, (float)byte / 255.0 (lib/imgio.cc:54-56,75-77):
	                      * a quarter of the PCIe / HBM traffic of the fp32 image */
} op_image;
enum { OP_F32 = 0, OP_U8 = 1 };

/* =====================================================================================
 * SIFT  -- replaces FeatureDetector::detect_feature / SIFTDetector::do_detect_feature
 * (feature/feature.hh:50-57, feature/feature.cc:20-47) looped over images by
 * StitcherBase::calc_feature (stitch/stitcherbase.cc:9-27).
 * One call extracts the features of n images; results stay resident in HBM inside an
 * op_features object so that the matcher reads them without a second H2D pass.
 * Output order per image is canonical: sorted by (octave, scale, y, x, real_x, real_y) of the
 * refined keypoint, then orientation-peak order (the reference's order is thread-timing
 * dependent, feature/extrema.cc:56).
 * ===================================================================================== */
typedef struct op_features op_features;
int op_sift_batch(op_ctx* ctx, const op_config* cfg, const op_image* imgs, int n, op_features** out);
/* The same for HOST images with the transfers overlapped (StitcherBase::calc_feature end to end, stitch/stitcherbase.cc:9-27:
 * Mat32f / decoder bytes in host memory in, descriptors and keypoint coordinates in host memory out).  The batch runs as
 * a pipeline over chunks: uploads on one copy stream, kernels on the context's stream, results on a second copy stream
 * into desc_out (rows x 128 fp32) / coor_out (rows x 2 fp64), images back to back, capacity_rows rows of room (either
 * pointer may be NULL; pinned memory copies at the link rate).  *out receives the resident features as op_sift_batch
 * does.  OP_ERR_CAPACITY (with *out still set) when the host buffers are too small. */
int op_sift_batch_host(op_ctx* ctx, const op_config* cfg, const op_image* imgs, int n,
		float* desc_out, double* coor_out, int64_t capacity_rows, op_features** out);
int op_features_num_images(const op_features* f);
/* number of descriptors of image i (K_i) and the exclusive prefix offset of its first row */
int op_features_count(const op_features* f, int i);
int64_t op_features_offset(const op_features* f, int i);
int64_t op_features_total(const op_features* f);
/* device-resident flat buffers: desc = total x 128 fp32, coor = total x 2 fp64
 * (coordinates relative to the image centre in original-image pixels, feature.cc:23-26) */
const float* op_features_desc_device(const op_features* f);
const double* op_features_coor_device(const op_features* f);
/* D2H copy of image i's K_i x 128 descriptors and K_i x 2 coordinates (either may be NULL) */
int op_features_copy(op_ctx* ctx, const op_features* f, int i, float* desc, double* coor);
/* D2H copy of image i's K_i x 2 real_coor in [0,1) -- what SIFTDetector::do_detect_feature itself
 * returns (feature/feature.cc:31-47) before FeatureDetector::detect_feature re-centres it (:20-28);
 * the C++ adapter's do_detect_feature override hands these to the reference's base class */
int op_features_copy_real(op_ctx* ctx, const op_features* f, int i, double* real);
/* build an op_features from host arrays (debug commands / tests: match without SIFT).
 * desc may be NULL (coordinates only: enough for op_ransac_pairs; op_match_pairs then fails) */
int op_features_from_host(op_ctx* ctx, const float* const* desc, const double* const* coor,
		const int* counts, int n, op_features** out);
/* the same from a flat DEVICE buffer (images back to back; D2D copy) -- the multi-GPU path hands
 * the all-gathered descriptors of every rank's images to the matcher this way */
int op_features_from_device(op_ctx* ctx, const float* desc_dev, const double* coor_dev,
		const int* counts, int n, op_features** out);
/* the same without the copy: the table ADOPTS the caller's device buffers (total x 128 fp32, total x 2 fp64, images
 * back to back), which must stay valid and unchanged until op_features_free -- the exchange step of the multi-GPU
 * path receives every rank's features straight into such a table and hands it to the matcher as it is */
int op_features_adopt_device(op_ctx* ctx, const float* desc_dev, const double* coor_dev,
		const int* counts, int n, op_features** out);
void op_features_free(op_features* f);

/* Staged single-image run keeping every intermediate (the debug commands raw_extrema /
 * keypoint / orientation of main.cc:41-81 and the per-stage parity tests).
 * plane kinds: 1 DoG[s] (s in 0..nscale-2), 2 mag[s], 3 ort[s] (s in 1..nscale-3),
 *              4 working RGB, 5 grey octave base, 6 Gaussian[s] (s in 1..nscale-1).
 * Only the Gaussian stack reaches HBM in op_sift_batch: DoG planes live in LDS for the extrema scan and are
 * re-evaluated as |G[s] - G[s+1]| (feature/dog.cc:126) where the refinement needs them; mag/ort are evaluated
 * by the orientation / descriptor kernels (GaussianPyramid::cal_mag_ort, feature/dog.cc:60-94) on the Gaussian
 * plane for the samples they use.  The dump computes DoG, mag and ort planes on request. */
typedef struct op_sift_dump op_sift_dump;
int op_sift_staged(op_ctx* ctx, const op_config* cfg, const op_image* img, op_sift_dump** out);
void op_sift_dump_free(op_sift_dump* d);
int op_sift_dump_working_dims(const op_sift_dump* d, int* h, int* w);
int op_sift_dump_octave_dims(const op_sift_dump* d, int oct, int* h, int* w);
int op_sift_dump_plane(op_ctx* ctx, const op_sift_dump* d, int kind, int oct, int s, float* out);
int op_sift_dump_raw_count(const op_sift_dump* d, int oct, int s);
int op_sift_dump_raw(const op_sift_dump* d, int oct, int s, int* xy);       /* sorted (y, x) */
/* which: 0 refined, 1 oriented; ints = x,y,octave,scale; real = real_coor; fl = dir, scale_factor */
int op_sift_dump_kp_count(const op_sift_dump* d, int which);
int op_sift_dump_kp(const op_sift_dump* d, int which, int* ints, double* real, float* fl);
int op_sift_dump_desc(const op_sift_dump* d, float* desc, double* coor);    /* coor in [0,1) */

/* device math twins of glibc expf/cosf/sinf/hypotf and of fast_atan (feature/dog.cc:22-37),
 * evaluated on the GPU over n host values (tests: device == reference libm, bit for bit).
 * which: 0 expf(x), 1 cosf(x), 2 sinf(x), 3 hypotf(x,y), 4 fast_atan(y,x)+pi */
int op_debug_math(op_ctx* ctx, int which, const float* x, const float* y, int n, float* out);
/* Per-image capacity of the context's raw / refined candidate lists (default 16384).  The lists
 * are speculative: a batch that outgrows them re-runs once with the observed size and the context
 * keeps the larger capacity (the reference appends to std::vectors, extrema.cc:36-61).  Exposed so
 * that tests can force the overflow path with a tiny capacity; >= 64. */
int op_debug_set_raw_capacity(op_ctx* ctx, int cap);
/* Floats of LDS list arena one sorting pass of the descriptor kernel may use (default and maximum 640).  A batch of 64
 * window samples (sift.cc:110-146) whose per-bin lists, padded to float4s, need more is sorted and accumulated in two
 * passes of 32 samples; exposed so that tests can force that path (0: always). */
int op_debug_set_desc_list_cap(op_ctx* ctx, int floats);

/* =====================================================================================
 * MATCH -- replaces PairWiseMatcher (feature/matcher.hh:40-67, matcher.cc:73-135) as used by
 * Stitcher::pairwise_match / linear_pairwise_match (stitch/stitcher.cc:96-136), with the
 * semantics of the reference's exact matcher FeatureMatcher::match (matcher.cc:15-71): 2-NN
 * ratio test from the smaller set, then the reverse test.  All requested pairs are
 * matched in one launch series; MFMA tiles (two-term bf16 split, fp32 accumulate) rank candidates, an exact re-score in the
 * reference's summation order (feature/dist.cc:22-57) decides.
 * ===================================================================================== */
typedef struct op_matches op_matches;
/* pairs: npairs x 2 image indices (i, j) into f.  The result stays in HBM -- per image pair a list of
 * <idx in image i, idx in image j> sorted by (first, second) (MatchData, matcher.hh:14-25), pairs back to back in
 * the order of `pairs` -- where op_ransac_pairs reads it; the call itself brings back the per-pair counts only.
 * An op_matches must not outlive the context it was made with. */
int op_match_pairs(op_ctx* ctx, const op_config* cfg, const op_features* f,
		const int* pairs, int npairs, op_matches** out);
int op_matches_count(const op_matches* m, int p);
/* the list of pair p on the host; the first copy fetches the whole job's lists from the device (one D2H),
 * later ones are lookups.  Thread-safe (PairWiseMatcher::match is called concurrently, stitcher.cc:106-109). */
int op_matches_copy(const op_matches* m, int p, int* idx_pairs);
/* every list at once: idx_pairs = total x 2 ints (may be NULL), offsets = npairs + 1 entries (may be NULL) */
int op_matches_copy_all(const op_matches* m, int* idx_pairs, int64_t* offsets);
/* the device-resident flat list (total x 2 int32), NULL when the lists were wrapped from host arrays */
const int* op_matches_device_list(const op_matches* m);
int64_t op_matches_total(const op_matches* m);
/* wrap host match lists (npairs lists of counts[p] <first, second> pairs): debug / test entry */
int op_matches_from_host(const int* const* idx_pairs, const int* counts, int npairs, op_matches** out);
/* a's pairs followed by b's as ONE result (both made with contexts of ctx's device): what a rank of a sharded job passes to
 * op_ransac_pairs when it matched its pair list in two calls (stitcher.cc:100-113 is one loop over all pairs).  The new
 * object owns a copy of the lists; a and b stay valid and are freed by the caller. */
int op_matches_concat(op_ctx* ctx, const op_matches* a, const op_matches* b, op_matches** out);
void op_matches_free(op_matches* m);

/* =====================================================================================
 * SEVERAL GPUs IN ONE PROCESS -- SURVEY 8(e): the reference's two parallel loops are the axes
 * (images: stitch/stitcherbase.cc:14-25; image pairs: stitch/stitcher.cc:96-113).  A group owns
 * one context (stream, workspaces, one host thread per call) per listed device.  The *_multi calls
 * return exactly what the single-device calls return; with one device they ARE those calls.
 * (One process per GPU with an RCCL all-gather instead: openpano_amd/distributed.py.)
 * ===================================================================================== */
typedef struct op_group op_group;
int op_group_create(const int* devices, int ndev, op_group** out);   /* a device may be listed twice */
void op_group_destroy(op_group* g);
int op_group_size(const op_group* g);
op_ctx* op_group_ctx(op_group* g, int k);                             /* context k; context 0 holds gathered results */
/* StitcherBase::calc_feature sharded by image: contiguous blocks of n / ndev images per context; the features are
 * all-gathered device-to-device over xGMI (every device pulls the other devices' slices over its own links) -- the
 * returned op_features is the table on context 0 and carries its replicas on the other devices.  Host images only,
 * or device images resident on the device of the context that owns them. */
int op_sift_batch_multi(op_group* g, const op_config* cfg, const op_image* imgs, int n, op_features** out);
/* Stitcher::pairwise_match sharded by pair: every device uses its replica of the table (made by op_sift_batch_multi, or
 * pulled from f's device on first use), the pair list is dealt balanced by K_i * K_j, each device keeps the lists of
 * its pairs resident; counts / lists read back in the order of `pairs`. */
int op_match_pairs_multi(op_group* g, const op_config* cfg, const op_features* f, const int* pairs, int npairs, op_matches** out);
/* TransformEstimation for the same job (stitch/stitcher.cc:66-94 inside the pair loop): every device estimates the pairs it
 * matched -- their match lists are resident there -- with the seeds op_ransac_pairs would give them; the result is in the
 * order of `pairs`.  m from op_match_pairs (one device) or wrapped host lists runs on context 0. */
struct op_ransac_result;
int op_ransac_pairs_multi(op_group* g, const op_config* cfg, const op_features* f, const op_matches* m,
		const int* pairs, int npairs, const int* shapes_wh, const uint32_t* seeds, uint32_t base_seed,
		struct op_ransac_result** out);

/* =====================================================================================
 * RANSAC -- replaces TransformEstimation(...).get_transform(MatchInfo*)
 * (stitch/transform_estimate.hh:22-31, transform_estimate.cc:26-218) for every pair at once.
 * pairs / m must be the pair list and result of op_match_pairs (match p belongs to pairs[p]);
 * keypoint coordinates come from f (centred original-image pixels); shapes_wh holds (w, h) of
 * every image of f (Shape2D, stitch/match_info.hh:53-78).  The matched point pairs are gathered on the
 * device from m's resident lists and f's coordinates (nothing is re-uploaded); one D2H brings back the
 * winners and the gathered points for the host-side acceptance gates of fill_inliers_to_matchinfo.
 * Homography (8-point samples) unless cfg->CYLINDER || cfg->TRANS (affine, 7-point samples).
 * Sampling is std::mt19937 with the reference's duplicate rejection; pair p is seeded with
 * seeds[p], or from base_seed and p when seeds == NULL (the reference seeds from
 * std::random_device, i.e. is not reproducible: transform_estimate.cc:64-65).
 * The result mirrors MatchInfo (stitch/match_info.hh:14-51): ok = get_transform()'s return,
 * confidence (negative = -#inliers, as the reference leaves it on rejection), homo (image j ->
 * image i), inlier indices into the pair's match list.
 * ===================================================================================== */
typedef struct op_ransac_result op_ransac_result;
int op_ransac_pairs(op_ctx* ctx, const op_config* cfg, const op_features* f, const op_matches* m,
		const int* pairs, int npairs, const int* shapes_wh, const uint32_t* seeds, uint32_t base_seed,
		op_ransac_result** out);
int op_ransac_ok(const op_ransac_result* r, int p);
float op_ransac_confidence(const op_ransac_result* r, int p);
int op_ransac_homo(const op_ransac_result* r, int p, double* h9);
int op_ransac_inlier_count(const op_ransac_result* r, int p);
int op_ransac_inliers(const op_ransac_result* r, int p, int* match_indices);
/* index of the winning hypothesis and its inlier count (-1: no healthy hypothesis) */
int op_ransac_best(const op_ransac_result* r, int p, int* hyp, int* count);
/* number of accepted pairs and the inliers they hold, over the whole result */
int op_ransac_summary(const op_ransac_result* r, int* accepted_pairs, int64_t* inliers);
void op_ransac_free(op_ransac_result* r);

/* Stitcher::match_image's bookkeeping for a whole job (stitch/stitcher.cc:79-93): every ACCEPTED pair (i, j) of the result
 * becomes two directed entries -- pairwise_matches[i][j] = MatchInfo{confidence, homo (j -> i), matches (point in i, point
 * in j)} and pairwise_matches[j][i] = the same with homo.inverse() scaled by 1 / inv[8] and every match reversed
 * (match_info.hh:21-25) -- in the flat form pano_estimate_cameras (include/pano_host.h) takes: CameraEstimator's input
 * without a pass through the caller's own containers.  `pairs` / npairs MUST be those of the op_ransac_pairs call that made r,
 * m the match lists it read (r keeps a hash of that pair list: another list is refused with OP_ERR_INVALID).  On any failure
 * the output arrays may be partly written.  Sizes first: entries = 2 x accepted pairs, points = total rows of pts.
 *   ij      entries x 2        conf  entries        homo  entries x 9        cnt  entries
 *   pts     points x 4 (first.x, first.y, second.x, second.y), entries back to back */
int op_pairwise_table_size(const op_ransac_result* r, int* entries, int64_t* points);
int op_pairwise_table(op_ctx* ctx, const op_features* f, const op_matches* m, const op_ransac_result* r, const int* pairs, int npairs,
		int* ij, float* conf, double* homo, int* cnt, double* pts);

/* =====================================================================================
 * WARP + BLEND -- replaces ConnectedImages::blend() (stitch/stitcher_image.hh:92,
 * stitch/stitcher_image.cc:116-155) together with the blender it constructs:
 * LinearBlender (stitch/blender.cc:24-96; cfg->MULTIBAND == 0, both cfg->LAZY_READ branches,
 * weights per cfg->ORDERED_INPUT) or MultiBandBlender(cfg->MULTIBAND) (stitch/multiband.cc:19-151).
 * BlenderBase::add_image takes the coordinate map as a std::function (stitch/blender.hh:52-56),
 * which a device cannot call, so the seam sits one level up and the map travels as PODs:
 * projection method, proj_range.min, resolution and homo_inv per image.
 * Pixels are the reference's, bit for bit, for every projection (the map's sin / cos / tan are tabulated per canvas
 * column / row by the host libm, colour arithmetic is the reference's fp32 sequence); Color::NO = -1 marks
 * "no pixel" on input and output (lib/color.cc:11-15).  "The host libm" is the one this library is linked against: the
 * claim is bit-for-bit against a reference built on the same glibc (sin / cos / tan of other libms may differ in the last place).
 * Threading: the tables are cached per context -- geometry key + host and device copy, (2 W + H) doubles each, for the
 * context's lifetime).  Like every other per-context workspace they make an op_ctx THREAD-COMPATIBLE, not thread-safe:
 * one call at a time per context; concurrent callers use one context each (as the adapters in pano_hip.hh do).
 * ===================================================================================== */
typedef struct op_blend_image {
	const float* data;    /* H x W x 3 fp32 (ImageRef::img, stitch/imageref.hh:15-17) */
	int h, w;
	int on_device;
	double homo_inv[9];   /* ImageComponent::homo_inv (stitch/stitcher_image.hh:40-42) */
	double range[4];      /* ImageComponent::range: min.x, min.y, max.x, max.y (:48) */
	/* Dimensions of the pixel buffer `data` when they differ from h / w, else 0.  The reference keeps
	 * ImageRef::_width/_height from load() (stitch/imageref.hh:22-31) after CylinderStitcher replaced the
	 * Mat by its cylinder warp (cylstitcher.cc:66-67): bounds, blend weights and the image centre keep
	 * using the ORIGINAL size (blender.hh:39-44, blender.cc:31-35, stitcher_image.cc:150) while
	 * interpolate() reads the warped Mat with its own rows/cols (lib/imgproc.cc:135-156).  h / w are the
	 * ImageRef's, mat_h / mat_w the Mat's. */
	int mat_h, mat_w;
} op_blend_image;
typedef struct op_blend_geom {
	int proj_method;      /* ConnectedImages::ProjectionMethod (:30): 0 flat, 1 cylindrical, 2 spherical */
	double proj_min[2], proj_max[2];   /* ConnectedImages::proj_range (:33) */
	double resolution[2];              /* ConnectedImages::get_final_resolution() (stitcher_image.cc:79-114) */
} op_blend_geom;
/* HOST-side O(n) geometry the reference also runs on the host, in fp64 with the host libm:
 * calc_inverse_homo + update_proj_range + get_final_resolution (stitcher_image.cc:36-114).
 * homo: n x 9 ImageComponent::homo; shapes_wh: n x (w, h).  Fills g, homo_inv (n x 9) and
 * ranges (n x 4).  OP_ERR_INVALID where the reference would m_assert / error_exit. */
int op_blend_prepare(const op_config* cfg, int proj_method, int identity_idx, int n, const int* shapes_wh,
		const double* homo, op_blend_geom* g, double* homo_inv, double* ranges);
/* canvas size LinearBlender/MultiBandBlender::add_image arrive at (blender.cc:15-22) */
int op_blend_canvas_dims(const op_blend_geom* g, const op_blend_image* imgs, int n, int* h, int* w);
typedef struct op_canvas op_canvas;   /* device-resident H x W x 3 fp32 result (Mat32f) */
int op_blend(op_ctx* ctx, const op_config* cfg, const op_blend_geom* g, const op_blend_image* imgs, int n, op_canvas** out);
int op_canvas_dims(const op_canvas* c, int* h, int* w);
const float* op_canvas_device(const op_canvas* c);
int op_canvas_copy(op_ctx* ctx, const op_canvas* c, float* host);
void op_canvas_free(op_canvas* c);
/* crop(mat) (lib/imgproc.cc:200-235, called by main.cc:226-229 under config CROP): the largest
 * rectangle of valid pixels, first maximum in (line, column) scan order like the reference.
 * x0 / y0 (optional) receive the rectangle's origin in c.  The result may be empty (h = 0). */
int op_canvas_crop(op_ctx* ctx, const op_canvas* c, op_canvas** out, int* x0, int* y0);
/* write_rgb / write_png quantisation (lib/imgio.cc:25-40,98-113): Color::NO -> white, v * 255
 * truncated to a byte; D2H of H x W x 3 bytes instead of the fp32 canvas */
int op_canvas_copy_u8(op_ctx* ctx, const op_canvas* c, unsigned char* host);

/* CYLINDER mode pre-warp -- replaces CylinderWarper::warp (stitch/warp.hh:47-55, warp.cc:13-75).
 * op_cyl_warp_shape is the host part (projector, output shape, offset and the keypoints, which
 * are centred coordinates updated in place: warp.cc:46-67); op_cyl_warp renders the pixels. */
int op_cyl_warp_shape(const op_config* cfg, int w, int h, double h_factor, double* pts, int npts,
		int* new_w, int* new_h, double* offset);
int op_cyl_warp(op_ctx* ctx, const op_config* cfg, const op_image* img, double h_factor, op_canvas** out);

#ifdef __cplusplus
}
#endif
#endif /* OPENPANO_HIP_H */
