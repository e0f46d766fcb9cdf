/*
 * svo_oracle.h -- CPU restatement of the SVO hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for android_svo_amd: a plain-C, single-threaded
 * restatement of the reference algorithms on flat SoA buffers.  It is NOT part
 * of the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Every function cites the reference file:line
 * it follows (paths relative to /root/reference/app/src/main/cpp/svo/, headers
 * under include/svo/).
 *
 * Pinning status (see oracle/README.md and DESIGN.md):
 *   pinned against the reference's own code compiled here (oracle/_ref):
 *     SE3/SO3 algebra, Frame::jacobian_xyz2uv, Eigen LDLT 6x6 solve,
 *     the whole SparseImgAlign::run (precomputeReferencePatches, computeResiduals,
 *     solve, update and the NLLSSolver Gauss-Newton control flow: H_ bit-identical,
 *     caches bit-exact, iteration counts equal), feature_alignment::align2D/align1D,
 *     ZMSSD, warp::getWarpMatrixAffine/getBestSearchLevel/warpAffine,
 *     depthFromTriangulation, Matcher::findEpipolarMatchDirect and
 *     Matcher::findMatchDirect end to end, vk::interpolateMat_8u, vk::halfSample,
 *     the whole Reprojector::reprojectMap on a real svo::Map (close keyframes, cell lists,
 *     Point::getCloseViewObs, point bookkeeping; tests/golden/reproject_map_ref.npz),
 *     vk::PinholeCamera::world2cam / cam2world / isInFrame, Point::optimize, shiTomasiScore.
 *   pinned by the known-answer vector recorded in SURVEY.md 8(a-9):
 *     Seed ctor + DepthFilter::updateSeed.
 *   PARITY UNPINNED (restated from source text only; depth_filter.cpp needs the
 *   Android NDK log header, absent here, so it is unbuildable under the
 *   no-stand-ins rule):
 *     the per-seed glue of DepthFilter::updateSeeds around the matcher call,
 *     DepthFilter::computeTau.
 *
 * Conventions: SE3 = double[7] {tx,ty,tz,qx,qy,qz,qw} (reference SE3.h/SO3.h
 * member order); images are u8, row-major, stride == cols; pyramids are arrays
 * of level pointers, level l has (width>>l) x (height>>l) pixels.
 */
#ifndef SVO_ORACLE_H_
#define SVO_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVO_ORACLE_MAX_LEVELS 8

/* Pinhole camera with the reference's radtan forward model
 * (pinhole_camera.cpp:73-106).  distortion != 0 enables d[0..4]. */
typedef struct {
  int width, height;
  double fx, fy, cx, cy;
  double d[5];
  int distortion;
} svo_orc_camera;

/* ---- SE3 / SO3 (SE3.h:35-61,153-182; SO3.h:468-488,523-526) ------------- */
void svo_orc_se3_identity(double T[7]);
void svo_orc_se3_mul(const double A[7], const double B[7], double out[7]);
void svo_orc_se3_inverse(const double T[7], double out[7]);
void svo_orc_se3_act(const double T[7], const double p[3], double out[3]);
void svo_orc_se3_exp(const double twist[6], double out[7]);
void svo_orc_so3_log(const double q[4], double out[3]);
void svo_orc_se3_rotation_matrix(const double T[7], double R[9]);

/* ---- camera (abstract_camera.h:41-72, pinhole_camera.cpp:44-106) -------- */
void svo_orc_world2cam_uv(const svo_orc_camera* cam, const double uv[2], double px[2]);
void svo_orc_world2cam(const svo_orc_camera* cam, const double xyz[3], double px[2]);
/* distortion-free branch only (pinhole_camera.cpp:47-52,64) */
void svo_orc_cam2world(const svo_orc_camera* cam, double u, double v, double f[3]);
/* vk::AbstractCamera::isInFrame (I/abstract_camera.h:58-72): level < 0 selects the overload without a level */
int svo_orc_is_in_frame(const svo_orc_camera* cam, int ox, int oy, int boundary, int level);

/* ---- small algebra ------------------------------------------------------ */
/* Frame::jacobian_xyz2uv, frame.h:110-132. J is 2x6 row-major. */
void svo_orc_jacobian_xyz2uv(const double xyz[3], double J[12]);
/* x = H.ldlt().solve(b) with Eigen 3.4.0's pivoted LDLT (Cholesky/LDLT.h:297-403,
 * 574-613).  H is 6x6 (symmetric; lower part is read).  Returns 1. */
int svo_orc_ldlt6_solve(const double H[36], const double b[6], double x[6]);

/* ---- image helpers ------------------------------------------------------ */
/* vk::halfSample scalar form (vision.cpp:89-110): (a+b+c+d)/4 truncating. */
void svo_orc_half_sample(const uint8_t* in, int w, int h, uint8_t* out);
/* vk::halfSample SSE2 form (vision.cpp:20-45): avg-of-avg, rounding up. */
void svo_orc_half_sample_sse2form(const uint8_t* in, int w, int h, uint8_t* out);
/* vk::interpolateMat_8u, vision.h:19-36 */
float svo_orc_interpolate_8u(const uint8_t* img, int stride, float u, float v);

/* ---- SparseImgAlign (sparse_img_align.cpp:51-308, nlls_solver_impl.hpp:25-100) */
typedef struct {
  int max_level, min_level;   /* coarse-to-fine: level = max..min           */
  int n_iter;                 /* max GN iterations per level                */
  double eps;                 /* 1e-6 in the reference ctor (:40)           */
  int early_stop;             /* 1 = reference semantics; 0 = fixed work    */
} svo_orc_sia_params;

typedef struct {
  double T_cur_w[7];          /* result pose                                */
  size_t n_tracked;           /* n_meas_/16 of the last residual evaluation */
  double H[36];               /* H_ of the last evaluation                  */
  double Jres[6];
  double chi2;                /* chi2_ member after the run                 */
  int stop;                   /* stop_ member after the run                 */
  int iters[SVO_ORACLE_MAX_LEVELS];   /* residual evaluations per level     */
  /* algorithmic work counters */
  long n_precompute_patches;  /* sum over levels of patches precomputed      */
  long n_residual_patches;    /* sum over evaluations of patches accumulated */
} svo_orc_sia_result;

int svo_orc_sparse_img_align(
    const svo_orc_camera* cam,
    const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
    int n_feat, const double* px /*[n][2]*/, const double* f /*[n][3]*/,
    const double* pos /*[n][3]*/, const uint8_t* has_point /*[n]*/,
    const double T_ref_w[7], const double T_cur_w_init[7],
    const svo_orc_sia_params* prm, svo_orc_sia_result* out);
/* The same with the other branches of vk::NLLSSolver (I/nlls_solver.h:46-48): method 0 GaussNewton / 1
 * LevenbergMarquardt (nlls_solver_impl.hpp:102-227; early_stop is ignored there), scale_estimator 0 Unit (no weights) /
 * 1 TDist / 2 MAD / 3 Normal, weight_function 0 Unit / 1 TDist / 2 Tukey / 3 Huber (S/robust_cost.cpp).  With LM,
 * `iters` counts computeResiduals calls (two per trial) and `H` is the damped matrix of the last trial.
 * scale_out (optional): scale_ as the run leaves it. */
int svo_orc_sparse_img_align_ex(
    const svo_orc_camera* cam,
    const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
    int n_feat, const double* px, const double* f, const double* pos, const uint8_t* has_point,
    const double T_ref_w[7], const double T_cur_w_init[7],
    const svo_orc_sia_params* prm, int method, int scale_estimator, int weight_function,
    svo_orc_sia_result* out, float* scale_out);

/* Single residual/linearisation evaluation, for kernel-level parity tests:
 * runs precompute at `level` (fresh visibility) then one computeResiduals at
 * T_cur_from_ref.  out28 = {H upper-tri (21, row-major i<=j), Jres (6), chi2sum}
 * and n_meas. */
int svo_orc_sia_single_eval(
    const svo_orc_camera* cam, const uint8_t* ref_img, const uint8_t* cur_img,
    int level, int n_feat, const double* px, const double* f, const double* pos,
    const uint8_t* has_point, const double T_ref_w[7], const double T_cur_from_ref[7],
    double out28[28], long* n_meas,
    float* ref_patch_cache /*[n][16] or NULL*/, double* jac_cache /*[n][16][6] or NULL*/,
    uint8_t* visible /*[n] or NULL*/);

/* Step-wise access to the restated residual body (used by oracle/ref to run the
 * reference's own NLLSSolver driver on top of it). */
void* svo_orc_sia_open(const svo_orc_camera* cam, const uint8_t* const* ref_pyr,
                       const uint8_t* const* cur_pyr, int n_feat, const double* px,
                       const double* f, const double* pos, const uint8_t* has_point,
                       const double T_ref_w[7]);
void svo_orc_sia_set_level(void* h, int level);
double svo_orc_sia_eval(void* h, const double T_cur_from_ref[7], int linearize, double H[36],
                        double Jres[6], size_t* n_meas);
void svo_orc_sia_close(void* h);

/* ---- feature_alignment (feature_alignment.cpp:35-152, 154-282) ---------- */
int svo_orc_align2d(const uint8_t* cur_img, int cols, int rows, int stride,
                    const uint8_t* ref_patch_with_border /*10x10*/,
                    const uint8_t* ref_patch /*8x8*/, int n_iter,
                    double px_inout[2], int* iters_done);
int svo_orc_align1d(const uint8_t* cur_img, int cols, int rows, int stride,
                    const float dir[2], const uint8_t* ref_patch_with_border,
                    const uint8_t* ref_patch, int n_iter, double px_inout[2],
                    double* h_inv, int* iters_done);

/* ---- matcher pieces (matcher.cpp:36-147, patch_score.h:40-220) ---------- */
void svo_orc_get_warp_matrix_affine(
    const svo_orc_camera* cam_ref, const svo_orc_camera* cam_cur,
    const double px_ref[2], const double f_ref[3], double depth_ref,
    const double T_cur_ref[7], int level_ref, double A_cur_ref[4] /*row-major*/);
int svo_orc_get_best_search_level(const double A_cur_ref[4], int max_level);
/* returns 0 when the inverse warp is NaN (patch left untouched, matcher.cpp:94-98) */
int svo_orc_warp_affine(const double A_cur_ref[4], const uint8_t* img_ref, int cols,
                        int rows, const double px_ref[2], int level_ref,
                        int search_level, int halfpatch_size, uint8_t* patch);
void svo_orc_patch_from_border(const uint8_t* patch_with_border, uint8_t* patch);
int svo_orc_zmssd(const uint8_t* ref_patch /*64*/, const uint8_t* cur, int stride);
int svo_orc_depth_from_triangulation(const double T_search_ref[7], const double f_ref[3],
                                     const double f_cur[3], double* depth);

typedef struct {
  int ok;                /* return value of findEpipolarMatchDirect          */
  double depth;
  double px_cur[2];
  int search_level;
  double epi_length;
  int n_zmssd;           /* ZMSSD evaluations actually executed (after dedupe) */
  int n_align_iters;     /* align2D iterations executed                        */
  int path;              /* 0 = short epipolar (direct align), 1 = search, 2 = skipped (too long) */
  uint8_t patch_with_border[100];
} svo_orc_epi_result;

/* Matcher::findEpipolarMatchDirect, matcher.cpp:207-355 (CORNER features,
 * align_1d=false, subpix_refinement=true: Matcher::Options defaults matcher.h:83-91);
 * n_pyr_levels = Config::nPyrLevels() (config.cpp:59 -> 3). */
int svo_orc_find_epipolar_match_direct(
    const svo_orc_camera* cam, const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
    const double T_cur_ref[7], const double px_ref[2], const double f_ref[3], int level_ref,
    double d_estimate, double d_min, double d_max, int n_pyr_levels, int align_max_iter,
    int max_epi_search_steps, svo_orc_epi_result* out);

/* Matcher::findMatchDirect, matcher.cpp:156-202, for a reference feature already chosen by the caller
 * (Point::getCloseViewObs is host bookkeeping).  px_cur in/out in level-0 pixels. */
int svo_orc_find_match_direct(const svo_orc_camera* cam, const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
                              const double T_ref_w[7], const double T_cur_w[7], const double px_ref[2],
                              const double f_ref[3], int level_ref, const double pt_pos[3], int edgelet,
                              const double grad[2], int n_pyr_levels, int align_max_iter, double px_cur[2],
                              int* search_level_out);

/* ---- depth filter (depth_filter.cpp:36-45, 237-341, 359-416) ------------ */
typedef struct { float a, b, mu, z_range, sigma2; } svo_orc_seed;
void svo_orc_seed_init(svo_orc_seed* s, float depth_mean, float depth_min);
void svo_orc_update_seed(float x, float tau2, svo_orc_seed* s);
double svo_orc_compute_tau(const double T_ref_cur[7], const double f[3], double z,
                           double px_error_angle);

/* per-seed outcome codes of one updateSeeds pass */
enum {
  SVO_SEED_BEHIND = 0,        /* xyz_f.z < 0                     (:268-271) */
  SVO_SEED_NOT_IN_FRAME = 1,  /* !isInFrame                      (:272-275) */
  SVO_SEED_NO_MATCH = 2,      /* match failed -> b++             (:283-290) */
  SVO_SEED_UPDATED = 3,       /* Bayes update done               (:299)     */
  SVO_SEED_CONVERGED = 4,     /* updated and converged -> erase  (:310-331) */
  SVO_SEED_NAN = 5            /* updated, z_inv_min NaN -> erase (:333-337) */
};

/* DepthFilter::updateSeeds body for seeds of ONE reference keyframe against one
 * current frame (no ageing/halt: host bookkeeping).  Arrays are SoA per seed. */
int svo_orc_update_seeds(
    const svo_orc_camera* cam, const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
    const double T_ref_w[7], const double T_cur_w[7],
    int n_seeds, const double* px /*[n][2]*/, const double* f /*[n][3]*/, const int* level,
    float* a, float* b, float* mu, const float* z_range, float* sigma2,
    int n_pyr_levels, int align_max_iter, int max_epi_search_steps,
    double convergence_sigma2_thresh,
    int* status, double* z_out, double* xyz_world /*[n][3], valid when converged*/,
    int* n_zmssd, int* n_align_iters);
/* the same, also reporting the matcher's public members after each seed's findEpipolarMatchDirect call:
 * px_cur[n][2] (Matcher::px_cur_, what updateSeeds hands to setGridOccpuancy on keyframes,
 * depth_filter.cpp:302-306; NaN unless the seed was updated) and search_level[n] (-1: matcher not reached) */
int svo_orc_update_seeds_ex(
    const svo_orc_camera* cam, const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
    const double T_ref_w[7], const double T_cur_w[7],
    int n_seeds, const double* px, const double* f, const int* level,
    float* a, float* b, float* mu, const float* z_range, float* sigma2,
    int n_pyr_levels, int align_max_iter, int max_epi_search_steps,
    double convergence_sigma2_thresh,
    int* status, double* z_out, double* xyz_world,
    int* n_zmssd, int* n_align_iters, double* px_cur_out, int* search_level_out);

/* ---- next rows f-4: pose_optimizer::optimizeGaussNewton (pose_optimizer.cpp:31-181) and
 * ---- Point::optimize (point.cpp:130-192) ------------------------------------------------ */
int svo_orc_ldlt3_solve(const double A[9], const double b[3], double x[3]);
float svo_orc_median_f(const float* v, int n);          /* vk::getMedian: element floor(n/2) of the sorted data */
float svo_orc_tukey_weight(float x);                    /* TukeyWeightFunction::value, b = 8.6851f */
void svo_orc_inverse6(const double A[36], double out[36]);   /* Matrix<double,6,6>::inverse() (partial-pivot LU) */

typedef struct {
  int ran;                /* 0: no observation with a point -> the reference returns before touching anything */
  double T_f_w[7];        /* refined pose (rolled back when the last step was rejected)                     */
  double estimated_scale; /* MAD scale * errorMultiplier2, as returned through the reference parameter       */
  double error_init, error_final;
  size_t num_obs;         /* observations left after the outlier test                                        */
  double Cov[36];         /* frame->Cov_                                                                     */
  int n_iter_done;        /* linear systems built and solved                                                 */
  int n_deleted;
} svo_orc_pose_opt_result;

int svo_orc_pose_optimize(double error_multiplier2, double reproj_thresh, int n_iter, const double T_f_w[7], int n,
                          const double* f /*[n][3]*/, const double* pos /*[n][3]*/, const int* level /*[n]*/,
                          uint8_t* has_point /*[n] in/out*/, svo_orc_pose_opt_result* out);
int svo_orc_point_optimize(int n_iter, double pos[3] /*in/out*/, int n_obs, const double* obs_T_f_w /*[n_obs][7]*/,
                           const double* obs_f /*[n_obs][3]*/, int* iters_done);

/* ---- next row f-3: FastDetector::detect (feature_detection.cpp:77-122) -----------------------------------
 * cv::FAST is third-party (OpenCV 4.5.4 features2d, neither source nor library under /root/reference): PARITY
 * UNPINNED, restated from its published algorithm.  vk::shiTomasiScore (vision.cpp:113-154) is pinned. */
int svo_orc_fast(const uint8_t* img, int w, int h, int threshold, int max_out, int* xs, int* ys, int* scores);
float svo_orc_shi_tomasi_score(const uint8_t* img, int cols, int rows, int u, int v);
/* px_out[n][2] level-0 pixel (integers), level_out[n], score_out[n]; capacity = number of grid cells; returns n */
int svo_orc_detect_features(const uint8_t* const* pyr, int width, int height, int n_pyr_levels, int cell_size,
                            const uint8_t* occupancy /*[cells] or NULL*/, double detection_threshold, int* px_out,
                            int* level_out, float* score_out);

/* ---- next row f-2: the cell loop of Reprojector::reprojectMap (reprojector.cpp:149-166, 180-241) ----
 * candidates of cell c: [cell_offset[c], cell_offset[c+1]) in trial order; see svo_oracle.c */
int svo_orc_reproject_cells(const svo_orc_camera* cam, int n_kf, const uint8_t* const* const* kf_pyr, const double* T_kf_w,
                            const uint8_t* const* cur_pyr, const double T_cur_w[7], int n_cells, const int* cell_offset,
                            const int* kf_slot, const double* px_ref, const double* f_ref, const int* level_ref,
                            const double* pt_pos, const uint8_t* edgelet, const double* grad, const uint8_t* deleted,
                            double* px_cur /*in/out*/, int max_fts, int n_pyr_levels, int align_max_iter, uint8_t* tried,
                            uint8_t* matched, int* search_level, int* cell_winner, size_t* n_matches, size_t* n_trials);

/* ---- Reprojector::reprojectMap on a flattened svo::Map (reprojector.cpp:72-259, map.cpp:109-131, point.cpp:101-125,
 *      matcher.cpp:156-202).  Index tables instead of the pointer graph: keyframes in Map::keyframes_ order, the features
 *      of each keyframe that have a point in fts_ order, the observations of each point in Point::obs_ order (each
 *      names the keyframe it lies in), the candidate points in MapPointCandidates::candidates_ order.  Point types are
 *      Point::PointType values (0 deleted, 1 candidate, 2 unknown, 3 good). */
typedef struct {
  int n_kf;
  const double* T_kf_w;          /* [n_kf][7] */
  const int* kf_key_point;       /* [n_kf][5]: point of key_pts_[j], -1 for NULL */
  const int* kf_ftr_offset;      /* [n_kf + 1] */
  const int* kf_ftr_point;       /* point index per keyframe feature (-1: no point) */
  int n_points;
  const double* pt_pos;          /* [n_points][3] */
  int* pt_type;                  /* in/out */
  int* pt_n_failed;              /* in/out */
  int* pt_n_succeeded;           /* in/out */
  const int* pt_obs_offset;      /* [n_points + 1] */
  const int* obs_kf;             /* keyframe index of every observation */
  const double* obs_px;          /* [..][2] */
  const double* obs_f;           /* [..][3] */
  const int* obs_level;
  const uint8_t* obs_edgelet;    /* may be NULL */
  const double* obs_grad;        /* [..][2], may be NULL */
  int n_candidates;
  const int* cand_point;
} svo_orc_map;

/* pt_unlinked [n_points] in/out: points whose references were cut (safeDeletePoint / deleted candidates).  overlap_kf /
 * overlap_count [max_n_kfs]: the reference's overlap_kfs.  feat_* [number of grid cells]: the features added to the
 * frame, in creation order. */
int svo_orc_reproject_map(const svo_orc_camera* cam, svo_orc_map* m, const uint8_t* const* const* kf_pyr,
                          const uint8_t* const* cur_pyr, const double T_cur_w[7], int grid_size, int max_fts, int max_n_kfs,
                          int n_pyr_levels, int align_max_iter, uint8_t* pt_unlinked, int* n_overlap_out, int* overlap_kf,
                          int* overlap_count, int* n_feat_out, double* feat_px, int* feat_level, int* feat_point,
                          uint8_t* feat_edgelet, double* feat_grad, size_t* n_matches_out, size_t* n_trials_out);

#ifdef __cplusplus
}
#endif
#endif
