// sparse_img_align_hip.cpp -- drop-in replacement of svo/sparse_img_align.cpp.
//
// Keeps the class surface of I/sparse_img_align.h:33-79 (same header, same ctor, `run`,
// `getFisherInformation`, inherited NLLSSolver knobs) and forwards the whole coarse-to-fine solve
// to libsvo_hip.so.  Compile this file INSTEAD of sparse_img_align.cpp.
//
// Differences a maintainer should know:
//  * run() flattens ref_frame->fts_ (std::list walk, sparse_img_align.cpp:116-118) into SoA arrays in
//    list order, uploads them with the two pyramids and the poses, runs, and reads back T_f_w_, n_meas_,
//    H_ and chi2_.  Host pointers are never retained.
//  * method_ (GaussNewton / LevenbergMarquardt) and the robust cost set with setRobustCostFunction go down with the call
//    (svo_hip_sia_set_option; android_svo_amd/csrc/svo_nlls.hip).  No caller of the reference enables either
//    (frame_handler_mono.cpp:186-187,331-332), and Gauss-Newton without weights is the path with the fused kernel; the
//    other branches run one launch per evaluation.  mu_, nu_ and scale_ are read back like H_ and chi2_.
//  * display_ (residual image window) is not offloaded: a warning, then the run without it.
//  * the virtual hooks computeResiduals/solve/update exist for the base class but are not called.
#include <algorithm>
#include <vector>

#include <svo/abstract_camera.h>
#include <svo/config.h>
#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/point.h>
#include <svo/robust_cost.h>
#include <svo/sparse_img_align.h>

#include "svo_hip_bridge.h"

namespace svo {

namespace {
// one context + device pyramids + solver per host thread that aligns frames
struct AlignDevice {
  hip_bridge::Context ctx;
  hip_bridge::PyramidCache ref_pyr, cur_pyr;
  svo_hip_sia* sia;
  int sia_capacity;
  AlignDevice() : ctx(0), ref_pyr(ctx.get(), 1), cur_pyr(ctx.get(), 1), sia(NULL), sia_capacity(0) {}
  ~AlignDevice() { if (sia) svo_hip_sia_destroy(sia); }
};

AlignDevice& device() {
  static thread_local AlignDevice d;
  return d;
}
}  // namespace

SparseImgAlign::SparseImgAlign(int max_level, int min_level, int n_iter, Method method, bool display, bool verbose)
    : display_(display), max_level_(max_level), min_level_(min_level) {
  n_iter_ = n_iter;
  n_iter_init_ = n_iter_;
  method_ = method;
  verbose_ = verbose;
  eps_ = 0.000001;
}

size_t SparseImgAlign::run(FramePtr ref_frame, FramePtr cur_frame) {
  reset();
  if (ref_frame->fts_.empty()) {
    SVO_WARN_STREAM("SparseImgAlign: no features to track!");
    return 0;
  }
  ref_frame_ = ref_frame;
  cur_frame_ = cur_frame;

  AlignDevice& dev = device();
  if (!dev.ctx.ok()) { hip_bridge::reportDeviceFailure(NULL, "SparseImgAlign::run"); stop_ = true; return 0; }
  const int n = (int)ref_frame->fts_.size();
  std::vector<double> px(2 * (size_t)n), f(3 * (size_t)n), pos(3 * (size_t)n, 0.0);
  std::vector<uint8_t> has_point((size_t)n, 0);
  size_t i = 0;
  for (auto it = ref_frame->fts_.begin(); it != ref_frame->fts_.end(); ++it, ++i) {
    const Feature* ftr = *it;
    px[2 * i] = ftr->px[0]; px[2 * i + 1] = ftr->px[1];
    f[3 * i] = ftr->f[0]; f[3 * i + 1] = ftr->f[1]; f[3 * i + 2] = ftr->f[2];
    if (ftr->point != NULL) {
      has_point[i] = 1;
      pos[3 * i] = ftr->point->pos_[0]; pos[3 * i + 1] = ftr->point->pos_[1]; pos[3 * i + 2] = ftr->point->pos_[2];
    }
  }
  if (!dev.sia || dev.sia_capacity < n) {
    if (dev.sia) svo_hip_sia_destroy(dev.sia);
    dev.sia = NULL;
    dev.sia_capacity = std::max(n, 2048);
    if (svo_hip_sia_create(dev.ctx.get(), 1, dev.sia_capacity, &dev.sia) != SVO_HIP_OK) { stop_ = true; return 0; }
  }
  const int rs = dev.ref_pyr.slotOf(*ref_frame);
  const int cs = dev.cur_pyr.slotOf(*cur_frame);
  if (rs < 0 || cs < 0) { stop_ = true; return 0; }

  const svo_hip_camera cam = hip_bridge::toCamera(cur_frame->cam_);
  double T_ref[7], T_cur[7];
  hip_bridge::toPose7(ref_frame->T_f_w_, T_ref);
  hip_bridge::toPose7(cur_frame->T_f_w_, T_cur);
  svo_hip_sia_params prm;
  prm.max_level = max_level_; prm.min_level = min_level_; prm.n_iter = (int)n_iter_; prm.eps = eps_;
  prm.early_stop = 1;
  svo_hip_sia_result res;
  if (display_) SVO_WARN_STREAM("SparseImgAlign: the residual image (display_) is not produced by the device path");
  // NLLSSolver's other branches, as this object is configured (I/nlls_solver.h:46-48,105-111)
  int scale_kind = SVO_HIP_SIA_SCALE_UNIT, weight_kind = SVO_HIP_SIA_WEIGHT_UNIT;
  if (use_weights_) {
    using namespace vk::robust_cost;
    if (dynamic_cast<TDistributionScaleEstimator*>(scale_estimator_.get())) scale_kind = SVO_HIP_SIA_SCALE_TDIST;
    else if (dynamic_cast<MADScaleEstimator*>(scale_estimator_.get())) scale_kind = SVO_HIP_SIA_SCALE_MAD;
    else if (dynamic_cast<NormalDistributionScaleEstimator*>(scale_estimator_.get())) scale_kind = SVO_HIP_SIA_SCALE_NORMAL;
    if (dynamic_cast<TDistributionWeightFunction*>(weight_function_.get())) weight_kind = SVO_HIP_SIA_WEIGHT_TDIST;
    else if (dynamic_cast<TukeyWeightFunction*>(weight_function_.get())) weight_kind = SVO_HIP_SIA_WEIGHT_TUKEY;
    else if (dynamic_cast<HuberWeightFunction*>(weight_function_.get())) weight_kind = SVO_HIP_SIA_WEIGHT_HUBER;
    if (scale_kind == SVO_HIP_SIA_SCALE_UNIT)            // use_weights_ with an estimator of the caller's own
      SVO_WARN_STREAM("SparseImgAlign: unknown scale estimator, running without weights");
  }
  int rc = svo_hip_sia_set_option(dev.sia, SVO_HIP_SIA_OPT_METHOD,
                                  method_ == LevenbergMarquardt ? SVO_HIP_SIA_METHOD_LEVENBERG_MARQUARDT : SVO_HIP_SIA_METHOD_GAUSS_NEWTON);
  if (rc == SVO_HIP_OK) rc = svo_hip_sia_set_option(dev.sia, SVO_HIP_SIA_OPT_SCALE_ESTIMATOR, scale_kind);
  if (rc == SVO_HIP_OK) rc = svo_hip_sia_set_option(dev.sia, SVO_HIP_SIA_OPT_WEIGHT_FUNCTION, weight_kind);
  if (rc == SVO_HIP_OK) rc = svo_hip_sia_set_frames(dev.sia, dev.ref_pyr.pyramid(), dev.cur_pyr.pyramid());
  if (rc == SVO_HIP_OK) rc = svo_hip_sia_upload_features(dev.sia, 0, n, px.data(), f.data(), pos.data(), has_point.data());
  if (rc == SVO_HIP_OK) rc = svo_hip_sia_upload_poses(dev.sia, 0, &cam, T_ref, T_cur);
  if (rc == SVO_HIP_OK) rc = svo_hip_sia_run(dev.sia, 1, &prm);
  if (rc == SVO_HIP_OK) rc = svo_hip_sia_download(dev.sia, 0, &res);
  if (rc != SVO_HIP_OK) {
    // device errors degrade to "not converged": pose untouched, 0 tracked (SURVEY 8b "Error conventions")
    hip_bridge::reportDeviceFailure(dev.ctx.get(), "SparseImgAlign::run");
    SVO_WARN_STREAM("SparseImgAlign: device path failed, pose left unchanged");
    stop_ = true;
    return 0;
  }
  cur_frame_->T_f_w_ = hip_bridge::fromPose7(res.T_cur_w);
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) H_(a, b) = res.H[a * 6 + b];
  chi2_ = res.chi2;
  stop_ = res.stop != 0;
  n_meas_ = (size_t)res.n_tracked * patch_area_;
  if (method_ == LevenbergMarquardt || scale_kind != SVO_HIP_SIA_SCALE_UNIT) {
    float scale = 0.0f;
    double mu = mu_, nu = nu_;
    if (svo_hip_sia_solver_state(dev.sia, 0, &scale, &mu, &nu) == SVO_HIP_OK) { scale_ = scale; mu_ = mu; nu_ = nu; }
  }
  return (size_t)res.n_tracked;
}

Matrix<double, 6, 6> SparseImgAlign::getFisherInformation() {
  double sigma_i_sq = 5e-4 * 255 * 255;   // image noise (sparse_img_align.cpp:96)
  Matrix<double, 6, 6> I = H_ / sigma_i_sq;
  return I;
}

// The per-evaluation hooks of vk::NLLSSolver are pure virtual in the base class; run() never
// calls optimize(), so they only have to exist.
void SparseImgAlign::precomputeReferencePatches() {}
double SparseImgAlign::computeResiduals(const SE3&, bool, bool) { return 0.0; }
int SparseImgAlign::solve() { return 0; }
void SparseImgAlign::update(const ModelType& old_model, ModelType& new_model) { new_model = old_model; }
void SparseImgAlign::startIteration() {}
void SparseImgAlign::finishIteration() {}

}  // namespace svo
