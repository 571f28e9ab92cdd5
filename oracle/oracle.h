/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the OpenPano hot path (SIFT -> match -> RANSAC -> warp/blend).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.so; the product (openpano_amd/, libopenpano_hip.so) never does.
 *
 * Parity status: PINNED.  The reference holds no golden vectors of its own,
 * so every function here is checked bit-for-bit (integer outputs, fp32 planes, descriptors)
 * against the reference's own sources compiled in place (oracle/_ref, tests/test_oracle_vs_ref.py)
 * and against the fixtures that build generated (tests/golden/, made by
 * tests/golden/make_golden.py).  The one unpinned boundary is the reference's use of the
 * system Eigen (absent here): FullPivLU 3x3 inverse and JacobiSVD least squares are restated
 * from their published algorithms (see oracle/ref_shim/Eigen/Dense and DESIGN.md).
 *
 * All citations are file:line in /root/reference/src.
 */
#ifndef OPENPANO_ORACLE_H
#define OPENPANO_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* mirror of namespace config (lib/config.hh:24-85), SIFT/match part; filled from floats
 * exactly like init_config() (main.cc:237-292) */
typedef struct {
	int SIFT_WORKING_SIZE, NUM_OCTAVE, NUM_SCALE;
	float SCALE_FACTOR, GAUSS_SIGMA;
	int GAUSS_WINDOW_FACTOR;
	float JUDGE_EXTREMA_DIFF_THRES, CONTRAST_THRES, PRE_COLOR_THRES, EDGE_RATIO;
	int CALC_OFFSET_DEPTH;
	float OFFSET_THRES;
	float ORI_RADIUS;
	int ORI_HIST_SMOOTH_COUNT;
	int DESC_HIST_SCALE_FACTOR, DESC_INT_FACTOR;
	float MATCH_REJECT_NEXT_RATIO;
} orc_sift_cfg;

void orc_sift_cfg_default(orc_sift_cfg* c);	/* src/config.cfg:19-49 */

/* ---- staged SIFT run: SIFTDetector::do_detect_feature (feature/feature.cc:31-47) ---- */
typedef struct orc_sift_run orc_sift_run;
orc_sift_run* orc_sift_new(const orc_sift_cfg* cfg, const float* rgb, int h, int w);
void orc_sift_free(orc_sift_run*);
void orc_sift_working_dims(const orc_sift_run*, int* h, int* w);
void orc_sift_octave_dims(const orc_sift_run*, int oct, int* h, int* w);
/* kind: 0 gaussian stack (s=0 grey base), 1 DoG, 2 mag, 3 ort, 4 working RGB */
int orc_sift_plane(const orc_sift_run*, int kind, int oct, int s, float* out);
int orc_sift_raw_count(const orc_sift_run*, int oct, int s);
void orc_sift_raw(const orc_sift_run*, int oct, int s, int* xy);
/* which: 0 refined, 1 oriented; ints: x,y,pyr,scale; real: real_coor; fl: dir, scale_factor */
int orc_sift_kp_count(const orc_sift_run*, int which);
void orc_sift_kp(const orc_sift_run*, int which, int* ints, double* real, float* fl);
int orc_sift_desc_count(const orc_sift_run*);
void orc_sift_desc(const orc_sift_run*, float* desc, double* coor);

/* FeatureDetector::detect_feature (feature/feature.cc:20-28): descriptors + centred coords.
 * Returns K; *desc (K*128 floats) and *coor (K*2 doubles) are malloc'd, free with orc_free. */
int orc_detect_feature(const orc_sift_cfg* cfg, const float* rgb, int h, int w,
		float** desc, double** coor);
void orc_free(void* p);
/* body of StitcherBase::calc_feature (stitch/stitcherbase.cc:14-25) over n equal-size images */
long orc_calc_feature_batch(const orc_sift_cfg* cfg, const float* rgb, int n, int h, int w, int nthreads);

/* GaussCache (feature/gaussian.cc:17-40) */
int orc_gauss_kernel(const orc_sift_cfg* cfg, float sigma, float* out);

/* glibc-compatible scalar kernels that the HIP side re-implements in fp64 (tests compare the
 * twins below with libm over exhaustive float ranges, and the device against the twins) */
float orc_expf_twin(float x);
float orc_cosf_twin(float x);
float orc_sinf_twin(float x);
float orc_hypotf_twin(float x, float y);
/* real libm / fast_atan over arrays: 0 expf, 1 cosf, 2 sinf, 3 hypotf(x,y), 4 fast_atan(y,x)+pi */
void orc_libm_batch(int which, const float* x, const float* y, long n, float* out);

/* ---- exact matcher: FeatureMatcher::match (feature/matcher.cc:15-71) ---- */
float orc_euclidean_sqr(const float* x, const float* y, int n, float now_thres);	/* feature/dist.cc:22-57 */
int orc_match_exact(const orc_sift_cfg* cfg, const float* d1, int n1, const float* d2, int n2, int* out_pairs);

long orc_match_pairs_batch(const orc_sift_cfg* cfg, const float* desc, const int* counts, int n, const int* pairs, int npairs, int nthreads);
unsigned long long orc_match_digest(const int* pairs2, int n);
long orc_match_pairs_digest(const orc_sift_cfg* cfg, const float* desc, const int* counts, int n, const int* pairs, int npairs,
		int nthreads, int* count, unsigned long long* digest);

/* ---- RANSAC: TransformEstimation::get_transform (stitch/transform_estimate.cc:26-218), seed injected ---- */
int orc_ransac(const int* match, int m, const double* kp1, int nk1, const double* kp2, int nk2,
		int w1, int h1, int w2, int h2, int affine, int iterations, double ransac_inlier_thres_cfg,
		float inlier_in_match_ratio, float inlier_in_points_ratio, unsigned seed,
		float* confidence, double* homo_out, int* inliers, int* n_inliers, int* best_hyp, int* best_count);

/* ---- warp + blend: ConnectedImages::blend (stitch/stitcher_image.cc:116-155) with
 * LinearBlender (stitch/blender.cc:24-96) or MultiBandBlender (stitch/multiband.cc:19-151) ---- */
typedef struct {
	const float* data;	/* H x W x 3 fp32, Color::NO = -1 allowed */
	int h, w;
	double homo_inv[9];	/* ImageComponent::homo_inv (stitcher_image.hh:40-42) */
	double range[4];	/* ImageComponent::range: min.x, min.y, max.x, max.y (projection coords) */
} orc_blend_image;
typedef struct {
	int proj_method;	/* ConnectedImages::ProjectionMethod: 0 flat, 1 cylindrical, 2 spherical */
	double proj_min[2], proj_max[2];	/* proj_range */
	double resolution[2];			/* get_final_resolution() */
} orc_blend_geom;
/* calc_inverse_homo + update_proj_range + get_final_resolution (stitcher_image.cc:36-114);
 * homo: n x 9 (centred image plane -> space); fills g, homo_inv (n x 9), ranges (n x 4) */
int orc_blend_prepare(int proj_method, int identity_idx, int n, const int* shapes_wh, const double* homo,
		int max_output_size, orc_blend_geom* g, double* homo_inv, double* ranges);
int orc_blend_dims(const orc_blend_geom* g, const orc_blend_image* imgs, int n, int* h, int* w);
int orc_blend_linear(const orc_blend_geom* g, const orc_blend_image* imgs, int n, int ordered_input, int lazy_read, float* out);
int orc_blend_multiband(const orc_blend_geom* g, const orc_blend_image* imgs, int n, int band_level, int window_factor, float* out);
/* CylinderWarper::warp (stitch/warp.hh:47-55, warp.cc:25-75): shape/keypoints, then pixels */
int orc_cyl_shape(int w, int h, double h_factor, float focal_length, double* pts, int npts,
		int* new_w, int* new_h, double* offset);
int orc_cyl_project(const float* img, int h, int w, double h_factor, float focal_length, float* out);

/* crop (lib/imgproc.cc:200-235): rectangle of the result inside mat; returns its area (0: nothing valid) */
int orc_crop_rect(const float* mat, int h, int w, int* x0, int* y0, int* cw, int* ch);
/* write_rgb quantisation (lib/imgio.cc:98-113) */
void orc_to_u8(const float* mat, long n, unsigned char* out);

#ifdef __cplusplus
}
#endif
#endif
