// ref_objects.cpp -- runs the REFERENCE's compiled SparseImgAlign and Matcher member functions on
// real svo::Frame / svo::Feature / svo::Point objects (TEST INFRASTRUCTURE ONLY, built into
// oracle/_ref/libsvo_ref.so by `make -C oracle ref`, only where /root/reference is mounted).
//
// What runs, unmodified, from the reference's own translation units:
//   sparse_img_align.cpp : precomputeReferencePatches, computeResiduals, solve, update,
//                          startIteration, finishIteration (reached through the class's own vtable)
//   nlls_solver_impl.hpp : reset, optimize, optimizeGaussNewton (the header's templates)
//   matcher.cpp          : Matcher::findEpipolarMatchDirect, Matcher::findMatchDirect and everything
//                          they call (warp::*, align2D/align1D, ZMSSD, depthFromTriangulation)
//   point.cpp            : Point constructors, Point::getCloseViewObs
//
// What does NOT run: SparseImgAlign's constructor and SparseImgAlign::run, and Frame's constructor.
// All three call out-of-line cv::Mat constructors of the OpenCV core library, which this image does
// not have (only Android-ABI archives are vendored); no stand-in for that library is written.  Their
// symbols stay unresolved in the shared object (lazy binding) and are never reached.  Instead the
// harness lays the objects out by hand in zeroed storage: the fields are set to what the inline base
// constructor (nlls_solver.h:96-116) and the constructor body (sparse_img_align.cpp:29-41) assign, the
// vptr is pointed at the class's own vtable, cv::Mat headers get their public fields filled (as in
// MatView), and the dozen driver statements of run() (:51-92) are performed here through the object's
// own members.  Every arithmetic statement that produces a compared number is reference code.
#include <algorithm>
#include <cstdlib>
#include <list>
#include <memory>
#include <new>
#include <vector>

#include <svo/global.h>
#include <svo/config.h>
#include <svo/frame.h>
#include <svo/feature.h>
#include <svo/point.h>
#include <svo/matcher.h>
#include <svo/robust_cost.h>

// the cache members and the GN state of SparseImgAlign are protected; access specifiers do not change
// the layout gcc gives the class
#define protected public
#include <svo/nlls_solver.h>
#include <svo/sparse_img_align.h>
#undef protected

// Reprojector keeps its grid and reprojectCell private
#include <svo/map.h>
#define private public
#include <svo/reprojector.h>
#undef private

#include "ref_common.h"
#include "ref_frames.h"
#ifdef REF_RUN_TRACKER_BINDING
#include "frame_tracker_hip.h"      // include/svo_dropin: the binding under test (tests/test_gpu_dropin_binding.py)
#endif

using namespace refh;

// vtable of svo::SparseImgAlign, emitted in the reference's sparse_img_align.o
extern "C" char _ZTVN3svo14SparseImgAlignE[];

// (HandFrame: ref_frames.h)

static bool pt_obs_nonempty(const svo::Point* p) { return !p->obs_.empty(); }

// radtan coefficients of the camera of the next ref_sparse_img_align_run* calls of this thread (all zero: the distortion-free
// HarnessPinhole the fixtures were recorded with; otherwise a hand-laid vk::PinholeCamera: the reference's compiled world2cam)
static thread_local double g_dist[5] = {0, 0, 0, 0, 0};
extern "C" void ref_set_distortion(const double* d5) { for (int i = 0; i < 5; ++i) g_dist[i] = d5 ? d5[i] : 0.0; }

#ifdef REF_RUN_TRACKER_BINDING
// what the last ref_reproject_map_keys call of this build (the frame-tracker binding in place of the Reprojector) left
static int g_tracker_ok = 0, g_tracker_pose_optimised = 0;
static double g_tracker_pose[7] = {0}, g_tracker_sfba[4] = {0};
static double g_second[8] = {0};      // second frame: ok, SparseImgAlign's tracked count, last-frame features with a point, matches, |pose change|, refined
extern "C" void dropin_tracker_second(double* out8) { for (int i = 0; i < 8; ++i) out8[i] = g_second[i]; }
static int g_struct_n = 0;
static double g_struct_max_diff = -1.0, g_struct_moved = 0.0;
extern "C" int dropin_tracker_structure(double* max_diff, double* moved) { *max_diff = g_struct_max_diff; *moved = g_struct_moved; return g_struct_n; }
extern "C" int dropin_tracker_last(double* pose7, double* sfba4, int* pose_optimised) {
  for (int i = 0; i < 7; ++i) pose7[i] = g_tracker_pose[i];
  for (int i = 0; i < 4; ++i) sfba4[i] = g_tracker_sfba[i];
  *pose_optimised = g_tracker_pose_optimised;
  return g_tracker_ok;
}
#endif

extern "C" {

// The reference's SparseImgAlign on one frame pair (see the file header for what is hand-laid).
// Outputs: pose, n_meas_/16, H_, chi2_, iter_ and n_meas_ per level (after that level's optimize), and the
// caches as they stand after the last level: ref_patch_cache_ [n][16] f32, jacobian_cache_ [n*16][6] f64
// (column-major 6 x 16n), visible_fts_ [n].
// method: 0 GaussNewton, 1 LevenbergMarquardt; scale_estimator / weight_function: the reference's enums
// (nlls_solver.h:47-48), handed to ITS setRobustCostFunction (nlls_solver_impl.hpp:229-281, robust_cost.o).
// scale_mu_out (optional): {scale_, mu_, nu_} as the run leaves them.
int ref_sparse_img_align_run_ex(int width, int height, double fx, double fy, double cx, double cy, int n_levels,
                             const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr, int n,
                             const double* px, const double* fv, const double* pos, const uint8_t* has_point,
                             const double* T_ref_w, const double* T_cur_w_init, int max_level, int min_level,
                             int n_iter, int method, int scale_estimator, int weight_function,
                             double* T_cur_w_out, size_t* n_tracked, double* H_out, double* chi2_out,
                             int* stop_out, int* iter_per_level /*[8]*/, size_t* n_meas_per_level /*[8]*/,
                             float* ref_patch_cache_out, double* jacobian_cache_out, uint8_t* visible_out,
                             double* scale_mu_out /*[3]*/);

// Fixed work (BASELINE config C1's "30 GN iters"): the reference's compiled computeResiduals / solve / update, exactly
// n_iter evaluations per level.  Its own optimizeGaussNewton cannot do that (the error-increase exit is unconditional,
// nlls_solver_impl.hpp:62), so the loop around the three members (:35-60 without the exits) is made here.
static thread_local int g_fixed_work = 0;
int ref_sparse_img_align_run_fixed_work(int width, int height, double fx, double fy, double cx, double cy, int n_levels,
                             const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr, int n,
                             const double* px, const double* fv, const double* pos, const uint8_t* has_point,
                             const double* T_ref_w, const double* T_cur_w_init, int max_level, int min_level,
                             int n_iter, double* T_cur_w_out, size_t* n_tracked, double* H_out, double* chi2_out);

int ref_sparse_img_align_run(int width, int height, double fx, double fy, double cx, double cy, int n_levels,
                             const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr, int n,
                             const double* px, const double* fv, const double* pos, const uint8_t* has_point,
                             const double* T_ref_w, const double* T_cur_w_init, int max_level, int min_level,
                             int n_iter, double* T_cur_w_out, size_t* n_tracked, double* H_out, double* chi2_out,
                             int* stop_out, int* iter_per_level /*[8]*/, size_t* n_meas_per_level /*[8]*/,
                             float* ref_patch_cache_out, double* jacobian_cache_out, uint8_t* visible_out) {
  return ref_sparse_img_align_run_ex(width, height, fx, fy, cx, cy, n_levels, ref_pyr, cur_pyr, n, px, fv, pos, has_point, T_ref_w,
                                     T_cur_w_init, max_level, min_level, n_iter, 0, 0, 0, T_cur_w_out, n_tracked, H_out, chi2_out,
                                     stop_out, iter_per_level, n_meas_per_level, ref_patch_cache_out, jacobian_cache_out,
                                     visible_out, nullptr);
}

int ref_sparse_img_align_run_ex(int width, int height, double fx, double fy, double cx, double cy, int n_levels,
                             const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr, int n,
                             const double* px, const double* fv, const double* pos, const uint8_t* has_point,
                             const double* T_ref_w, const double* T_cur_w_init, int max_level, int min_level,
                             int n_iter, int method, int scale_estimator, int weight_function,
                             double* T_cur_w_out, size_t* n_tracked, double* H_out, double* chi2_out,
                             int* stop_out, int* iter_per_level /*[8]*/, size_t* n_meas_per_level /*[8]*/,
                             float* ref_patch_cache_out, double* jacobian_cache_out, uint8_t* visible_out,
                             double* scale_mu_out) {
  const bool fixed_work = g_fixed_work != 0;
  HarnessPinhole plain(width, height, fx, fy, cx, cy);
  void* hand = nullptr;
  bool radtan = false;
  for (int i = 0; i < 5; ++i) radtan = radtan || g_dist[i] != 0.0;
  vk::AbstractCamera* camp = radtan ? make_hand_pinhole(width, height, fx, fy, cx, cy, g_dist, &hand) : &plain;
  struct HandGuard { void* h; ~HandGuard() { if (h) free_hand_pinhole(h); } } guard = {hand};
  HandFrame ref(camp, ref_pyr, width, height, n_levels, T_ref_w);
  HandFrame cur(camp, cur_pyr, width, height, n_levels, T_cur_w_init);
  for (int i = 0; i < n; ++i) ref.add_feature(px + 2 * i, fv + 3 * i, 0, has_point[i] ? pos + 3 * i : nullptr);
  svo::FramePtr ref_frame = ref.ptr(), cur_frame = cur.ptr();

  void* storage = ::aligned_alloc(32, (sizeof(svo::SparseImgAlign) + 31) / 32 * 32);
  std::memset(storage, 0, sizeof(svo::SparseImgAlign));
  *reinterpret_cast<void**>(storage) = _ZTVN3svo14SparseImgAlignE + 2 * sizeof(void*);
  svo::SparseImgAlign* s = reinterpret_cast<svo::SparseImgAlign*>(storage);
  // vk::NLLSSolver<6,SE3>::NLLSSolver() (nlls_solver.h:96-116)
  s->have_prior_ = false;
  s->mu_init_ = 0.01f; s->mu_ = s->mu_init_;
  s->nu_init_ = 2.0; s->nu_ = s->nu_init_;
  s->n_trials_ = 0; s->n_trials_max_ = 5; s->n_meas_ = 0;
  s->stop_ = false; s->iter_ = 0;
  s->use_weights_ = false; s->scale_ = 0.0;
  // SparseImgAlign::SparseImgAlign(max_level, min_level, n_iter, GaussNewton, false, false) (:29-41)
  s->display_ = false; s->max_level_ = max_level; s->min_level_ = min_level;
  s->n_iter_ = n_iter; s->n_iter_init_ = s->n_iter_;
  s->method_ = method == 1 ? svo::SparseImgAlign::LevenbergMarquardt : svo::SparseImgAlign::GaussNewton;
  s->verbose_ = false;
  s->eps_ = 0.000001;
  // the zeroed scale_estimator_ / weight_function_ members are empty shared pointers; a caller of the reference sets a
  // robust cost through the solver's own public member
  if (scale_estimator != 0 || weight_function != 0)
    s->setRobustCostFunction((svo::SparseImgAlign::ScaleEstimatorType)scale_estimator,
                             (svo::SparseImgAlign::WeightFunctionType)weight_function);

  // SparseImgAlign::run (:51-92)
  size_t ret = 0;
  s->reset();
  float* cache = nullptr;
  if (!ref_frame->fts_.empty()) {
    s->ref_frame_ = ref_frame;
    s->cur_frame_ = cur_frame;
    cache = static_cast<float*>(std::calloc((size_t)n * 16, sizeof(float)));
    fill_mat_header(&s->ref_patch_cache_, cache, n, 16, 16 * sizeof(float), CV_32FC1);
    s->jacobian_cache_.resize(Eigen::NoChange, s->ref_patch_cache_.rows * 16);
    s->visible_fts_.resize(s->ref_patch_cache_.rows, false);
    SE3 T_cur_from_ref(cur_frame->T_f_w_ * ref_frame->T_f_w_.inverse());
    for (s->level_ = s->max_level_; s->level_ >= s->min_level_; --s->level_) {
      s->mu_ = 0.1;
      s->jacobian_cache_.setZero();
      s->have_ref_patch_cache_ = false;
      if (fixed_work) {
        for (s->iter_ = 0; s->iter_ < s->n_iter_; ++s->iter_) {
          s->H_.setZero();
          s->Jres_.setZero();
          s->n_meas_ = 0;
          const double new_chi2 = s->computeResiduals(T_cur_from_ref, true, false);
          if (!s->solve()) { s->stop_ = true; break; }
          SE3 T_new;
          s->update(T_cur_from_ref, T_new);
          T_cur_from_ref = T_new;
          s->chi2_ = new_chi2;
        }
      } else {
        s->optimize(T_cur_from_ref);
      }
      if (s->level_ < 8) { iter_per_level[s->level_] = (int)s->iter_; n_meas_per_level[s->level_] = s->n_meas_; }
    }
    cur_frame->T_f_w_ = T_cur_from_ref * ref_frame->T_f_w_;
    ret = s->n_meas_ / 16;
  }
  from_se3(cur_frame->T_f_w_, T_cur_w_out);
  *n_tracked = ret;
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) H_out[6 * i + j] = s->H_(i, j);
  *chi2_out = s->chi2_;
  *stop_out = s->stop_ ? 1 : 0;
  if (scale_mu_out) { scale_mu_out[0] = (double)s->scale_; scale_mu_out[1] = s->mu_; scale_mu_out[2] = s->nu_; }
  if (cache) {
    if (ref_patch_cache_out) std::memcpy(ref_patch_cache_out, cache, (size_t)n * 16 * sizeof(float));
    if (jacobian_cache_out) std::memcpy(jacobian_cache_out, s->jacobian_cache_.data(), (size_t)n * 16 * 6 * sizeof(double));
    if (visible_out) for (int i = 0; i < n; ++i) visible_out[i] = s->visible_fts_[i] ? 1 : 0;
  }
  // tear down by hand (no destructor of the hand-laid object runs)
  s->jacobian_cache_.resize(Eigen::NoChange, 0);
  s->visible_fts_.~vector();
  s->scale_estimator_.reset();
  s->weight_function_.reset();
  s->ref_frame_.reset();
  s->cur_frame_.reset();
  std::free(cache);
  std::free(storage);
  return 0;
}

int ref_sparse_img_align_run_fixed_work(int width, int height, double fx, double fy, double cx, double cy, int n_levels,
                             const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr, int n,
                             const double* px, const double* fv, const double* pos, const uint8_t* has_point,
                             const double* T_ref_w, const double* T_cur_w_init, int max_level, int min_level,
                             int n_iter, double* T_cur_w_out, size_t* n_tracked, double* H_out, double* chi2_out) {
  int stop = 0, iters[8] = {0};
  size_t n_meas[8] = {0};
  g_fixed_work = 1;
  const int rc = ref_sparse_img_align_run_ex(width, height, fx, fy, cx, cy, n_levels, ref_pyr, cur_pyr, n, px, fv, pos, has_point, T_ref_w,
                                             T_cur_w_init, max_level, min_level, n_iter, 0, 0, 0, T_cur_w_out, n_tracked, H_out, chi2_out,
                                             &stop, iters, n_meas, nullptr, nullptr, nullptr, nullptr);
  g_fixed_work = 0;
  return rc;
}

// Matcher::findEpipolarMatchDirect (matcher.cpp:207-355) on real frames.
int ref_find_epipolar_match_direct(int width, int height, double fx, double fy, double cx, double cy, int n_levels,
                                   const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
                                   const double* T_ref_w, const double* T_cur_w, const double* px_ref,
                                   const double* f_ref, int level_ref, double d_estimate, double d_min, double d_max,
                                   double* depth, double* px_cur, int* search_level, double* epi_length,
                                   uint8_t* patch_with_border /*100*/) {
  HarnessPinhole cam(width, height, fx, fy, cx, cy);
  HandFrame ref(&cam, ref_pyr, width, height, n_levels, T_ref_w);
  HandFrame cur(&cam, cur_pyr, width, height, n_levels, T_cur_w);
  svo::Feature* ftr = ref.add_feature(px_ref, f_ref, level_ref, nullptr);
  svo::Matcher* m = new svo::Matcher();
  std::memset(m->patch_with_border_, 0, sizeof(m->patch_with_border_));
  std::memset(m->patch_, 0, sizeof(m->patch_));
  m->px_cur_ = Eigen::Vector2d(0, 0);
  m->search_level_ = -1;
  m->epi_length_ = -1.0;
  double z = 0.0;
  const bool ok = m->findEpipolarMatchDirect(*ref.f, *cur.f, *ftr, d_estimate, d_min, d_max, z);
  *depth = z;
  px_cur[0] = m->px_cur_[0]; px_cur[1] = m->px_cur_[1];
  *search_level = m->search_level_;
  *epi_length = m->epi_length_;
  std::memcpy(patch_with_border, m->patch_with_border_, 100);
  delete m;
  return ok ? 1 : 0;
}

// Matcher::findMatchDirect (matcher.cpp:156-202): the point has one observation (the reference feature).
int ref_find_match_direct(int width, int height, double fx, double fy, double cx, double cy, int n_levels,
                          const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr, const double* T_ref_w,
                          const double* T_cur_w, const double* px_ref, const double* f_ref, int level_ref,
                          const double* pt_pos, int edgelet, const double* grad, double* px_cur, int* search_level) {
  HarnessPinhole cam(width, height, fx, fy, cx, cy);
  HandFrame ref(&cam, ref_pyr, width, height, n_levels, T_ref_w);
  HandFrame cur(&cam, cur_pyr, width, height, n_levels, T_cur_w);
  svo::Feature* ftr = ref.add_feature(px_ref, f_ref, level_ref, pt_pos);
  if (edgelet) { ftr->type = svo::Feature::EDGELET; ftr->grad = Eigen::Vector2d(grad[0], grad[1]); }
  svo::Matcher* m = new svo::Matcher();
  std::memset(m->patch_with_border_, 0, sizeof(m->patch_with_border_));
  std::memset(m->patch_, 0, sizeof(m->patch_));
  m->search_level_ = -1;
  Eigen::Vector2d p(px_cur[0], px_cur[1]);
  const bool ok = m->findMatchDirect(*ftr->point, *cur.f, p);
  px_cur[0] = p[0]; px_cur[1] = p[1];
  *search_level = m->search_level_;
  delete m;
  return ok ? 1 : 0;
}

// Point::optimize (point.cpp:130-192): one map point observed from n_obs frames (pose + bearing each).
int ref_point_optimize(int n_iter, double* pos, int n_obs, const double* obs_T_f_w, const double* obs_f) {
  HarnessPinhole cam(640, 480, 500, 500, 320, 240);
  std::vector<HandFrame*> frames;
  svo::Point* pt = new svo::Point(Eigen::Vector3d(pos[0], pos[1], pos[2]));
  const double px0[2] = {0, 0};
  for (int k = 0; k < n_obs; ++k) {
    HandFrame* hf = new HandFrame(&cam, nullptr, 640, 480, 0, obs_T_f_w + 7 * k);
    svo::Feature* ftr = hf->add_feature(px0, obs_f + 3 * k, 0, nullptr);
    ftr->point = pt;
    pt->obs_.push_back(ftr);
    frames.push_back(hf);
  }
  pt->optimize((size_t)n_iter);
  pos[0] = pt->pos_[0]; pos[1] = pt->pos_[1]; pos[2] = pt->pos_[2];
  delete pt;
  for (HandFrame* hf : frames) delete hf;
  return 0;
}

// pieces of pose_optimizer::optimizeGaussNewton that exist outside pose_optimizer.cpp (which cannot be built:
// its logging macros need the Android NDK header)
float ref_tukey_weight(float x) {
  vk::robust_cost::TukeyWeightFunction w;
  return w.value(x);
}
float ref_mad_scale(const float* errors, int n) {
  std::vector<float> v(errors, errors + n);
  vk::robust_cost::MADScaleEstimator est;
  return est.compute(v);
}
double ref_median_d(const double* data, int n) {
  std::vector<double> v(data, data + n);
  return vk::getMedian(v);
}
void ref_inverse6(const double* A, double* out) {
  Eigen::Matrix<double, 6, 6> m;
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) m(i, j) = A[6 * i + j];
  Eigen::Matrix<double, 6, 6> r = m.inverse();
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) out[6 * i + j] = r(i, j);
}
void ref_ldlt3_solve(const double* A, const double* b, double* x) {
  Eigen::Matrix3d m;
  Eigen::Vector3d v(b[0], b[1], b[2]);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = A[3 * i + j];
  Eigen::Vector3d r = m.ldlt().solve(v);
  x[0] = r[0]; x[1] = r[1]; x[2] = r[2];
}

// The cell loop of Reprojector::reprojectMap (reprojector.cpp:149-166) on a real Reprojector with a real (empty) Map:
// the candidates are put into grid_.cells in the given order, then reprojectCell (the reference's compiled member,
// with its cell.sort, Matcher::findMatchDirect, point bookkeeping and frame->addFeature) is called cell by cell exactly
// as reprojectMap does.  One map point per candidate, observed once from keyframe kf_slot[i].
// Outputs per candidate: n_failed / n_succeeded / type of its point afterwards and whether it is still in its cell;
// per new feature of the frame (in creation order): candidate index, px, level, type, grad.
int ref_reproject_cells(int width, int height, double fx, double fy, double cx, double cy, int n_levels, int n_kf,
                        const uint8_t* const* const* kf_pyr, const double* T_kf_w, const uint8_t* const* cur_pyr,
                        const double* T_cur_w, int n_cells, const int* cell_offset, const int* kf_slot, const double* px_ref,
                        const double* f_ref, const int* level_ref, const double* pt_pos, const uint8_t* edgelet,
                        const double* grad, const int* point_type, const int* n_failed_in, const int* n_succeeded_in,
                        const double* px_cur_in, int max_fts, int* n_failed_out, int* n_succeeded_out, int* type_out,
                        uint8_t* left_in_cell, int* feat_cand, double* feat_px, int* feat_level, int* feat_type,
                        double* feat_grad, size_t* n_matches, size_t* n_trials) {
  HarnessPinhole cam(width, height, fx, fy, cx, cy);
  std::vector<HandFrame*> kfs;
  for (int k = 0; k < n_kf; ++k) kfs.push_back(new HandFrame(&cam, kf_pyr[k], width, height, n_levels, T_kf_w + 7 * k));
  HandFrame cur(&cam, cur_pyr, width, height, n_levels, T_cur_w);
  svo::FramePtr frame = cur.ptr();
  const int n = cell_offset[n_cells];
  std::vector<svo::Point*> pts((size_t)n);
  svo::Map* map = new svo::Map();
  svo::Reprojector* rp = new svo::Reprojector(&cam, *map);
  if ((int)rp->grid_.cells.size() < n_cells) return -1;
  for (int c = 0; c < n_cells; ++c)
    for (int i = cell_offset[c]; i < cell_offset[c + 1]; ++i) {
      svo::Feature* ftr = kfs[kf_slot[i]]->add_feature(px_ref + 2 * i, f_ref + 3 * i, level_ref[i], pt_pos + 3 * i);
      kfs[kf_slot[i]]->points.pop_back();                   // ownership below: deleted points go to the map's trash
      if (edgelet && edgelet[i]) { ftr->type = svo::Feature::EDGELET; ftr->grad = Eigen::Vector2d(grad[2 * i], grad[2 * i + 1]); }
      svo::Point* pt = ftr->point;
      pt->type_ = (svo::Point::PointType)point_type[i];
      pt->n_failed_reproj_ = n_failed_in[i];
      pt->n_succeeded_reproj_ = n_succeeded_in[i];
      pts[i] = pt;
      Eigen::Vector2d px(px_cur_in[2 * i], px_cur_in[2 * i + 1]);
      rp->grid_.cells.at(c)->push_back(svo::Reprojector::Candidate(pt, px));
    }
  // reprojectMap :149-166 (resetGrid() zeroes the counters at :63-64)
  rp->n_matches_ = 0;
  rp->n_trials_ = 0;
  const size_t cap = (size_t)max_fts;
  for (size_t i = 0; i < rp->grid_.cells.size() && (int)i < n_cells; ++i) {
    if (rp->reprojectCell(*rp->grid_.cells.at(rp->grid_.cell_order[i]), frame)) ++rp->n_matches_;
    if (rp->n_matches_ > cap) break;
  }
  *n_matches = rp->n_matches_;
  *n_trials = rp->n_trials_;
  for (int c = 0; c < n_cells; ++c) {
    for (int i = cell_offset[c]; i < cell_offset[c + 1]; ++i) left_in_cell[i] = 0;
    for (auto& cand : *rp->grid_.cells.at(c))
      for (int i = cell_offset[c]; i < cell_offset[c + 1]; ++i)
        if (pts[i] == cand.pt) left_in_cell[i] = 1;
  }
  for (int i = 0; i < n; ++i) {
    n_failed_out[i] = pts[i]->n_failed_reproj_;
    n_succeeded_out[i] = pts[i]->n_succeeded_reproj_;
    type_out[i] = (int)pts[i]->type_;
  }
  int nf = 0;
  for (svo::Feature* ftr : cur.f->fts_) {
    int idx = -1;
    for (int i = 0; i < n; ++i) if (pts[i] == ftr->point) idx = i;
    feat_cand[nf] = idx;
    feat_px[2 * nf] = ftr->px[0]; feat_px[2 * nf + 1] = ftr->px[1];
    feat_level[nf] = ftr->level;
    feat_type[nf] = (int)ftr->type;
    feat_grad[2 * nf] = ftr->grad[0]; feat_grad[2 * nf + 1] = ftr->grad[1];
    ++nf;
  }
  // tear down: points that were not deleted are ours, deleted ones sit in the map's trash (freed by ~Map)
  for (int i = 0; i < n; ++i) if (pts[i]->type_ != svo::Point::TYPE_DELETED) delete pts[i];
  delete rp;
  delete map;
  for (HandFrame* k : kfs) delete k;
  return nf;
}

// The whole Reprojector::reprojectMap (reprojector.cpp:72-168) on a real svo::Map: keyframes with their feature lists
// and key points (chosen by the reference's own Frame::setKeyPoints), map points with several observations, point
// candidates.  Everything that runs is the reference's compiled code: Map::getCloseKeyframes, Frame::isVisible, the
// list sort, reprojectPoint, the candidate loop, reprojectCell with cell.sort / Point::getCloseViewObs /
// Matcher::findMatchDirect / Map::safeDeletePoint / MapPointCandidates::deleteCandidatePoint.
// Inputs (index tables): observation o lies in keyframe obs_kf[o] and belongs to point obs_point[o]; the observations of a
// point are appended to its obs_ list in ascending o; keyframe k's fts_ holds the features kf_ftr_obs[kf_ftr_offset[k] ..
// kf_ftr_offset[k+1]) (observation indices) in that order; candidate c is point cand_point[c] with its single
// observation cand_obs[c] (a feature that is in no fts_ list).
// Outputs: kf_key_point [n_kf][5] (point index or -1) as the reference chose them BEFORE the call; per point type / counters
// after the call and whether it is still referenced by its features; the frame's new features; overlap_kfs.
// ref_reproject_map_keys: the same with the keyframes' key features under the caller's control -- key_override [n_kf][5]: -2 keeps
// what Frame::setKeyPoints chose, -1 empties the slot, a point index puts that point's feature in this keyframe there (an
// incumbent of the frame's history that a fresh selection would not necessarily pick) -- and the key points AFTER the call in
// kf_key_after [n_kf][5]: what Map::safeDeletePoint -> Frame::removeKeyPoint -> setKeyPoints (S/map.cpp:78-88,
// S/frame.cpp:83-165) left, executed by the reference's own compiled code inside reprojectMap.  Both may be null.
int ref_reproject_map_keys(int width, int height, double fx, double fy, double cx, double cy, int n_levels, int grid_size, int max_fts,
                           int n_pyr_levels, int n_kf, const uint8_t* const* const* kf_pyr, const double* T_kf_w,
                           const uint8_t* const* cur_pyr, const double* T_cur_w, int n_points, const double* pt_pos, const int* pt_type,
                           const int* pt_n_failed, const int* pt_n_succeeded, int n_obs, const int* obs_point, const int* obs_kf,
                           const double* obs_px, const double* obs_f, const int* obs_level, const uint8_t* obs_edgelet,
                           const double* obs_grad, const int* kf_ftr_offset, const int* kf_ftr_obs, int n_candidates,
                           const int* cand_point, const int* cand_obs,
                           int* kf_key_point, int* type_out, int* n_failed_out, int* n_succeeded_out, uint8_t* unlinked_out,
                           int* n_overlap, int* overlap_kf, int* overlap_count, int* feat_point, double* feat_px, int* feat_level,
                           int* feat_type, double* feat_grad, size_t* n_matches, size_t* n_trials,
                           const int* key_override, int* kf_key_after);

int ref_reproject_map(int width, int height, double fx, double fy, double cx, double cy, int n_levels, int grid_size, int max_fts,
                      int n_pyr_levels, int n_kf, const uint8_t* const* const* kf_pyr, const double* T_kf_w,
                      const uint8_t* const* cur_pyr, const double* T_cur_w, int n_points, const double* pt_pos, const int* pt_type,
                      const int* pt_n_failed, const int* pt_n_succeeded, int n_obs, const int* obs_point, const int* obs_kf,
                      const double* obs_px, const double* obs_f, const int* obs_level, const uint8_t* obs_edgelet,
                      const double* obs_grad, const int* kf_ftr_offset, const int* kf_ftr_obs, int n_candidates,
                      const int* cand_point, const int* cand_obs,
                      int* kf_key_point, int* type_out, int* n_failed_out, int* n_succeeded_out, uint8_t* unlinked_out,
                      int* n_overlap, int* overlap_kf, int* overlap_count, int* feat_point, double* feat_px, int* feat_level,
                      int* feat_type, double* feat_grad, size_t* n_matches, size_t* n_trials) {
  return ref_reproject_map_keys(width, height, fx, fy, cx, cy, n_levels, grid_size, max_fts, n_pyr_levels, n_kf, kf_pyr, T_kf_w, cur_pyr, T_cur_w,
                                n_points, pt_pos, pt_type, pt_n_failed, pt_n_succeeded, n_obs, obs_point, obs_kf, obs_px, obs_f, obs_level,
                                obs_edgelet, obs_grad, kf_ftr_offset, kf_ftr_obs, n_candidates, cand_point, cand_obs, kf_key_point, type_out,
                                n_failed_out, n_succeeded_out, unlinked_out, n_overlap, overlap_kf, overlap_count, feat_point, feat_px,
                                feat_level, feat_type, feat_grad, n_matches, n_trials, nullptr, nullptr);
}

int ref_reproject_map_keys(int width, int height, double fx, double fy, double cx, double cy, int n_levels, int grid_size, int max_fts,
                           int n_pyr_levels, int n_kf, const uint8_t* const* const* kf_pyr, const double* T_kf_w,
                           const uint8_t* const* cur_pyr, const double* T_cur_w, int n_points, const double* pt_pos, const int* pt_type,
                           const int* pt_n_failed, const int* pt_n_succeeded, int n_obs, const int* obs_point, const int* obs_kf,
                           const double* obs_px, const double* obs_f, const int* obs_level, const uint8_t* obs_edgelet,
                           const double* obs_grad, const int* kf_ftr_offset, const int* kf_ftr_obs, int n_candidates,
                           const int* cand_point, const int* cand_obs,
                           int* kf_key_point, int* type_out, int* n_failed_out, int* n_succeeded_out, uint8_t* unlinked_out,
                           int* n_overlap, int* overlap_kf, int* overlap_count, int* feat_point, double* feat_px, int* feat_level,
                           int* feat_type, double* feat_grad, size_t* n_matches, size_t* n_trials,
                           const int* key_override, int* kf_key_after) {
  svo::Config::gridSize() = (size_t)grid_size;
  svo::Config::maxFts() = (size_t)max_fts;
  svo::Config::nPyrLevels() = (size_t)n_pyr_levels;
  HarnessPinhole cam(width, height, fx, fy, cx, cy);
  std::vector<HandFrame*> kfs;
  for (int k = 0; k < n_kf; ++k) {
    kfs.push_back(new HandFrame(&cam, kf_pyr[k], width, height, n_levels, T_kf_w + 7 * k));
    kfs.back()->f->is_keyframe_ = true;
  }
  HandFrame cur(&cam, cur_pyr, width, height, n_levels, T_cur_w);
  svo::FramePtr frame = cur.ptr();
  std::vector<svo::Point*> pts((size_t)n_points);
  for (int p = 0; p < n_points; ++p) {
    pts[p] = new svo::Point(Eigen::Vector3d(pt_pos[3 * p], pt_pos[3 * p + 1], pt_pos[3 * p + 2]));
    pts[p]->type_ = (svo::Point::PointType)pt_type[p];
    pts[p]->n_failed_reproj_ = pt_n_failed[p];
    pts[p]->n_succeeded_reproj_ = pt_n_succeeded[p];
  }
  std::vector<svo::Feature*> obs((size_t)n_obs);
  for (int o = 0; o < n_obs; ++o) {
    svo::Feature* ftr = new svo::Feature(kfs[obs_kf[o]]->f, Eigen::Vector2d(obs_px[2 * o], obs_px[2 * o + 1]),
                                         Eigen::Vector3d(obs_f[3 * o], obs_f[3 * o + 1], obs_f[3 * o + 2]), obs_level[o]);
    if (obs_edgelet && obs_edgelet[o]) { ftr->type = svo::Feature::EDGELET; ftr->grad = Eigen::Vector2d(obs_grad[2 * o], obs_grad[2 * o + 1]); }
    ftr->point = pts[obs_point[o]];
    pts[obs_point[o]]->obs_.push_back(ftr);
    pts[obs_point[o]]->n_obs_++;
    obs[o] = ftr;
  }
  std::vector<char> in_fts((size_t)n_obs, 0);
  for (int k = 0; k < n_kf; ++k) {
    for (int j = kf_ftr_offset[k]; j < kf_ftr_offset[k + 1]; ++j) { kfs[k]->f->fts_.push_back(obs[kf_ftr_obs[j]]); in_fts[kf_ftr_obs[j]] = 1; }
    kfs[k]->f->setKeyPoints();                                       // the reference's own choice (frame.cpp:79-133)
    if (key_override)
      for (int j = 0; j < 5; ++j) {
        const int want = key_override[5 * k + j];
        if (want == -2) continue;
        svo::Feature* pick = nullptr;
        if (want >= 0)
          for (svo::Feature* ftr : kfs[k]->f->fts_) if (ftr->point == pts[want]) { pick = ftr; break; }
        kfs[k]->f->key_pts_[j] = pick;
      }
  }
  for (int k = 0; k < n_kf; ++k)
    for (int j = 0; j < 5; ++j) {
      kf_key_point[5 * k + j] = -1;
      svo::Feature* kp = kfs[k]->f->key_pts_[j];
      if (kp != nullptr)
        for (int p = 0; p < n_points; ++p) if (pts[p] == kp->point) kf_key_point[5 * k + j] = p;
    }
  svo::Map* map = new svo::Map();
  for (int k = 0; k < n_kf; ++k) map->addKeyframe(kfs[k]->ptr());
  for (int c = 0; c < n_candidates; ++c)                               // (newCandidatePoint would reset the type: push the pair as it does)
    map->point_candidates_.candidates_.push_back(svo::MapPointCandidates::PointCandidate(pts[cand_point[c]], obs[cand_obs[c]]));
  std::vector<std::pair<svo::FramePtr, std::size_t> > overlap;
#ifdef REF_RUN_TRACKER_BINDING
  // The SAME scene through the frame-tracker binding (include/svo_dropin/frame_tracker_hip.h: processFrame's alignment,
  // reprojection and pose refinement as one device call) instead of the reference's Reprojector: a last frame without features
  // at the pose the fixture was recorded at (SparseImgAlign::run returns at once, :55-59), then track().  The binding flattens
  // this svo::Map, runs the chain on the GPU and applies the outcome to these objects with the reference's own functions.
  svo::Reprojector* rp = nullptr;
  svo::hip_bridge::FrameTracker* tracker_p = new svo::hip_bridge::FrameTracker(&cam, n_kf > 0 ? n_kf : 1);
  {
    HandFrame last(&cam, cur_pyr, width, height, n_levels, T_cur_w);
    svo::hip_bridge::FrameTracker& tracker = *tracker_p;
    svo::hip_bridge::FrameTracker::Outcome oc;
    std::memset(&oc, 0, sizeof(oc));
    g_tracker_ok = tracker.ok() && tracker.track(last.ptr(), frame, *map, overlap, oc) ? 1 : 0;
    *n_matches = oc.repr_n_matches;
    *n_trials = oc.repr_n_trials;
    g_tracker_pose_optimised = oc.pose_optimised ? 1 : 0;
    g_tracker_sfba[0] = (double)oc.sfba_n_edges_final; g_tracker_sfba[1] = oc.sfba_thresh; g_tracker_sfba[2] = oc.sfba_error_init;
    g_tracker_sfba[3] = oc.sfba_error_final;
    from_se3(frame->T_f_w_, g_tracker_pose);
    // FrameHandlerBase::optimizeStructure (frame_handler_base.cpp:190-210) through the binding -- the frame's points refined on
    // the device, positions written back into Point::pos_ -- against the reference's compiled Point::optimize (point.o) run on
    // the very same objects from the very same start
    std::vector<svo::Point*> fpts;
    for (svo::Feature* ftr : frame->fts_)
      if (ftr->point != nullptr && std::find(fpts.begin(), fpts.end(), ftr->point) == fpts.end()) fpts.push_back(ftr->point);
    std::vector<Eigen::Vector3d> start;
    for (svo::Point* pt : fpts) start.push_back(pt->pos_);
    g_struct_n = (int)fpts.size();
    g_struct_max_diff = -1.0;
    if (g_tracker_ok && !fpts.empty() && tracker.optimiseStructure(frame, *map, fpts.size(), 5)) {
      std::vector<Eigen::Vector3d> dev;
      for (svo::Point* pt : fpts) dev.push_back(pt->pos_);
      g_struct_max_diff = 0.0;
      g_struct_moved = 0.0;
      for (size_t i = 0; i < fpts.size(); ++i) {
        fpts[i]->pos_ = start[i];
        fpts[i]->optimize(5);                                        // the reference's own
        g_struct_max_diff = std::max(g_struct_max_diff, (fpts[i]->pos_ - dev[i]).cwiseAbs().maxCoeff());
        g_struct_moved = std::max(g_struct_moved, (fpts[i]->pos_ - start[i]).cwiseAbs().maxCoeff());
        fpts[i]->pos_ = start[i];                                    // (the outputs below are those of the tracking call)
      }
    }
  }
#else
  svo::Reprojector* rp = new svo::Reprojector(&cam, *map);
  rp->reprojectMap(frame, overlap);
  *n_matches = rp->n_matches_;
  *n_trials = rp->n_trials_;
#endif
  *n_overlap = (int)overlap.size();
  for (size_t i = 0; i < overlap.size(); ++i) {
    overlap_kf[i] = -1;
    for (int k = 0; k < n_kf; ++k) if (overlap[i].first.get() == kfs[k]->f) overlap_kf[i] = k;
    overlap_count[i] = (int)overlap[i].second;
  }
  std::vector<char> is_cand((size_t)n_points, 0), cand_left((size_t)n_points, 0);
  for (int c = 0; c < n_candidates; ++c) is_cand[cand_point[c]] = 1;
  for (auto& pc : map->point_candidates_.candidates_)
    for (int p = 0; p < n_points; ++p) if (pts[p] == pc.first) cand_left[p] = 1;
  for (int p = 0; p < n_points; ++p) {
    type_out[p] = (int)pts[p]->type_;
    n_failed_out[p] = pts[p]->n_failed_reproj_;
    n_succeeded_out[p] = pts[p]->n_succeeded_reproj_;
    // cut loose: a map point whose observation list was cleared (safeDeletePoint), a candidate that left candidates_
    unlinked_out[p] = is_cand[p] ? (cand_left[p] ? 0 : 1) : ((pt_obs_nonempty(pts[p]) ? 0 : 1));
  }
  if (kf_key_after)
    for (int k = 0; k < n_kf; ++k)
      for (int j = 0; j < 5; ++j) {
        kf_key_after[5 * k + j] = -1;
        svo::Feature* kp = kfs[k]->f->key_pts_[j];
        if (kp != nullptr && kp->point != nullptr)
          for (int p = 0; p < n_points; ++p) if (pts[p] == kp->point) kf_key_after[5 * k + j] = p;
      }
  int nf = 0;
  for (svo::Feature* ftr : cur.f->fts_) {
    int idx = -1;
    for (int p = 0; p < n_points; ++p) if (pts[p] == ftr->point) idx = p;
    feat_point[nf] = idx;
    feat_px[2 * nf] = ftr->px[0]; feat_px[2 * nf + 1] = ftr->px[1];
    feat_level[nf] = ftr->level;
    feat_type[nf] = (int)ftr->type;
    feat_grad[2 * nf] = ftr->grad[0]; feat_grad[2 * nf + 1] = ftr->grad[1];
    ++nf;
  }
#ifdef REF_RUN_TRACKER_BINDING
  // A SECOND frame through the same tracker object, with the frame just tracked as the last frame: its features and their
  // points are the alignment's reference now (the hand-over inside the binding).  The image is the same and the pose starts
  // where the last frame's ended, so the alignment must use every last-frame feature that has a point and leave the pose
  // where it is; the reprojection runs again on the map as the first call left it.
  // (The image differs from the last one in the lowest bit of every 13th pixel: with the very same image every residual is
  // exactly 0, the update is exactly 0 and SE3::exp returns the NaN translation the reference's own exp returns for a zero
  // rotation -- SURVEY 8a-11, reproduced by the device -- which is not what this check is after.)
  {
    double T_last[7];
    from_se3(frame->T_f_w_, T_last);
    std::vector<std::vector<uint8_t> > second_img((size_t)n_levels);
    std::vector<const uint8_t*> second_pyr((size_t)n_levels);
    for (int l = 0; l < n_levels; ++l) {
      const size_t npx = (size_t)(width >> l) * (size_t)(height >> l);
      second_img[l].assign(cur_pyr[l], cur_pyr[l] + npx);
      for (size_t i = (size_t)l; i < npx; i += 13) second_img[l][i] ^= 1;
      second_pyr[l] = second_img[l].data();
    }
    HandFrame second(&cam, second_pyr.data(), width, height, n_levels, T_last);
    svo::FramePtr f2 = second.ptr();
    std::vector<std::pair<svo::FramePtr, std::size_t> > ov2;
    svo::hip_bridge::FrameTracker::Outcome o2;
    std::memset(&o2, 0, sizeof(o2));
    size_t with_point = 0;
    for (svo::Feature* ftr : frame->fts_) with_point += ftr->point != nullptr ? 1 : 0;
    g_second[0] = tracker_p->track(frame, f2, *map, ov2, o2) ? 1.0 : 0.0;
    g_second[1] = (double)o2.img_align_n_tracked;
    g_second[2] = (double)with_point;
    g_second[3] = (double)o2.repr_n_matches;
    double T2[7];
    from_se3(f2->T_f_w_, T2);
    double dmax = 0.0;
    for (int i = 0; i < 7; ++i) dmax = std::max(dmax, std::fabs(T2[i] - T_last[i]));
    g_second[4] = dmax;
    g_second[5] = o2.pose_optimised ? 1.0 : 0.0;
    g_second[6] = (double)o2.repr_n_trials;
    g_second[7] = (double)ov2.size();
  }
  delete tracker_p;
#endif
  // tear down.  Candidate points and their features belong to the map (MapPointCandidates::reset / its trash delete them);
  // deleted map points sit in the map's trash; the other points are ours; keyframe features die with their HandFrame.
  delete rp;
  for (int p = 0; p < n_points; ++p)
    if (!is_cand[p] && pts[p]->type_ != svo::Point::TYPE_DELETED) delete pts[p];
  delete map;
  for (HandFrame* k : kfs) delete k;
  return nf;
}

}  // extern "C"
