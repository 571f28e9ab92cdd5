/* pano_host.h -- C entry points of the HOST-ONLY stages of the stitching path
 * (openpano_amd/libpano_host.so, built by openpano_amd/csrc/Makefile with g++: no HIP, no Eigen).
 *
 * These are the stages the reference runs on the host between the device-side match/RANSAC stage
 * and the device-side blend when ESTIMATE_CAMERA is set:
 *   CameraEstimator::estimate            stitch/camera_estimator.cc:47-103
 *     Camera::estimate_focal             stitch/camera.cc:68-87
 *     IncrementalBundleAdjuster          stitch/incremental_bundle_adjuster.cc:116-385
 *     Camera::straighten                 stitch/camera.cc:149-183
 * C++ hosts use the classes of openpano_amd/host/pano_camera.hh directly (same names and members as
 * the reference); this header is for every other language.  The reference has no FFI: the
 * binding a maintainer would add is one call in Stitcher::estimate_camera (stitch/stitcher.cc:143-158),
 * see INTEGRATION.md. */
#ifndef PANO_HOST_H
#define PANO_HOST_H
#ifdef __cplusplus
extern "C" {
#endif

/* namespace config values read by the host stages (lib/config.hh): STRAIGHTEN, MULTIPASS_BA,
 * LM_LAMBDA (+ ESTIMATE_CAMERA, ORDERED_INPUT, TRANS, CYLINDER).  Returns 0, or -1 for an unknown key. */
int pano_config_set(const char* key, float value);
/* the K_i * K_j-balanced pair deal of a sharded job (stitcher.cc:100's pair list over the ranks): own pairs
 * first, the rest longest first to the least-loaded rank; mine[k] = 1 for the pairs of `rank`; owner may be NULL */
int pano_deal_pairs(int npairs, const int* pairs, const long long* cost, int world, int rank, const int* owner, int nimg, unsigned char* mine);

/* Stitcher::estimate_camera's CameraEstimator{pairwise_matches, shapes}.estimate().
 *   n            images; shapes_wh: n x (w, h)
 *   np           directed entries; ij: np x (i, j) -> pairwise_matches[i][j]
 *   conf, homo   per entry: MatchInfo::confidence, MatchInfo::homo (9 doubles, maps j -> i)
 *   cnt, pts     per entry: number of matches, then all matches back to back as
 *                (first.x, first.y, second.x, second.y) = (point in i, point in j), centred coordinates
 *   out          n x 13 doubles: focal, aspect, ppx, ppy, R[9] (row-major)
 * Ends the process like the reference's error_exit (message on stderr, exit status 1) when the
 * images are not connected (camera_estimator.cc:121,150-157). */
int pano_estimate_cameras(int n, const int* shapes_wh, int np, const int* ij, const float* conf, const double* homo,
		const int* cnt, const double* pts, double* out);

/* Camera::rotation_to_angle / angle_to_rotation (stitch/camera.cc:91-147): r = 9 doubles row-major, v = 3 */
void pano_rotation_to_angle(const double* r, double* v);
void pano_angle_to_rotation(const double* v, double* r);

/* Homography::inverse (stitch/homography.cc:25-39): returns 1 and fills inv, or 0 if singular */
int pano_homography_inverse(const double* a, double* inv);

/* A x = b through the column-pivoted Householder QR the bundle adjuster uses (n x n, row-major) */
void pano_colpiv_solve(const double* A, int n, const double* b, double* x);

/* One Levenberg-Marquardt step of IncrementalBundleAdjuster with its internals exposed (tests):
 * cams n x 13 as above; np add_match(i, j, matches) calls with cnt/pts as above; identity = index of
 * the image whose rotation is held fixed.  resid: 2 * sum(cnt); jtj: (6 m)^2 damped normal matrix,
 * upd: 6 m, m = number of distinct images among the entries. */
int pano_iba_probe(int n, const double* cams, int np, const int* ij, const int* cnt, const double* pts, int identity,
		double* resid, double* jtj, double* upd);

#ifdef __cplusplus
}
#endif
#endif
