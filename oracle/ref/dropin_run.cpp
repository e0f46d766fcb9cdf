// dropin_run.cpp -- RUNS the drop-in bindings include/svo_dropin/sparse_img_align_hip.cpp, pose_optimizer_hip.cpp and
// feature_alignment_hip.h on the reference's own types
// (TEST INFRASTRUCTURE ONLY; built into oracle/_ref/libsvo_dropin_run.so by `make -C oracle dropin-run`, only where
// /root/reference is mounted; the built library travels to the GPU box and is loaded by tests/test_gpu_dropin_binding.py).
//
// What runs: the drop-in's svo::SparseImgAlign::run() and getFisherInformation() -- the translation unit a maintainer compiles
// INSTEAD of svo/sparse_img_align.cpp -- with the reference's unmodified headers (svo::Frame, Feature, Point,
// vk::PinholeCamera, the NLLSSolver base with its reset() and setRobustCostFunction) and the reference's compiled point.o,
// config.o, robust_cost.o, pinhole_camera.o behind them; the device work in android_svo_amd/csrc/libsvo_hip.so.
// What is laid out by hand (as in ref_objects.cpp / ref_camera.cpp, for the same reason: cv::Mat's constructors and
// destructor live in the OpenCV library this image does not have, and no stand-in is written): the two svo::Frame objects,
// the vk::PinholeCamera, and the SparseImgAlign object itself -- zeroed storage, the vptr pointed at the DROP-IN's vtable,
// the fields set to what the inline base constructor (nlls_solver.h:96-116) and the drop-in's constructor body assign
// (sparse_img_align_hip.cpp: its two cv::Mat members stay zeroed headers nothing touches).
#include <cmath>
#include <condition_variable>   // (I/depth_filter.h:148 relies on a transitive include of libc++; depth_filter_hip.h includes it first too)
#include <cstdlib>
#include <cstring>
#include <functional>
#include <list>
#include <memory>
#include <vector>

#include <svo/global.h>

// width_/height_ are protected, the intrinsics private const members, the solver state of SparseImgAlign protected; access
// specifiers do not change the layout gcc gives the classes
#define protected public
#define private public
#include <svo/abstract_camera.h>
#include <svo/pinhole_camera.h>
#include <svo/nlls_solver.h>
#include <svo/sparse_img_align.h>
#undef private
#undef protected

#include <svo/config.h>
#include <svo/depth_filter.h>
#include <svo/frame.h>
#include <svo/feature.h>
#include <svo/point.h>
#include <svo/pose_optimizer.h>
#include <svo/robust_cost.h>

#include "ref_common.h"
#include "ref_frames.h"
#include "feature_alignment_hip.h"   // include/svo_dropin: the batched align2D wrapper (header-only)
#include "depth_filter_hip.h"        // ... the depth filter binding's host policy + the device seed mirror it drives

using namespace refh;

// vtable of svo::SparseImgAlign, emitted in the drop-in's object
extern "C" char _ZTVN3svo14SparseImgAlignE[];

#include "ref_pinhole_hand.h"
using refh::HandPinhole;

extern "C" {

// svo::SparseImgAlign(max_level, min_level, n_iter, method, false, false) [+ setRobustCostFunction]; run(ref_frame, cur_frame)
// exactly as FrameHandlerMono::processFrame makes the call (frame_handler_mono.cpp:186-188), through the DROP-IN.
int dropin_sparse_img_align_run(int width, int height, double fx, double fy, double cx, double cy, const double* d5, int n_levels,
                                const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr, int n, const double* px,
                                const double* fv, const double* pos, const uint8_t* has_point, const double* T_ref_w,
                                const double* T_cur_w_init, int max_level, int min_level, int n_iter, int method, int scale_estimator,
                                int weight_function, double* T_cur_w_out, size_t* n_tracked, double* fisher36, double* chi2_out,
                                int* stop_out, double* scale_mu_nu /*[3]*/, int row_pad) {
  HandPinhole cam(width, height, fx, fy, cx, cy, d5);
  // row_pad > 0: the frames' cv::Mat levels are NOT continuous (step = cols + row_pad, the padding filled with noise): the
  // bridge has to compact them before the upload
  std::vector<std::vector<uint8_t> > padded;
  std::vector<const uint8_t*> ref_p(n_levels), cur_p(n_levels);
  for (int l = 0; l < n_levels; ++l) { ref_p[l] = ref_pyr[l]; cur_p[l] = cur_pyr[l]; }
  if (row_pad > 0) {
    for (int which = 0; which < 2; ++which)
      for (int l = 0; l < n_levels; ++l) {
        const int w = width >> l, h = height >> l, st = w + row_pad;
        padded.push_back(std::vector<uint8_t>((size_t)st * h));
        std::vector<uint8_t>& b = padded.back();
        const uint8_t* src = which ? cur_pyr[l] : ref_pyr[l];
        for (int y = 0; y < h; ++y) {
          std::memcpy(&b[(size_t)y * st], src + (size_t)y * w, w);
          for (int x = w; x < st; ++x) b[(size_t)y * st + x] = (uint8_t)(37 * x + 11 * y + 101 * l);
        }
        (which ? cur_p : ref_p)[l] = b.data();
      }
  }
  HandFrame ref(cam.cam, ref_p.data(), width, height, n_levels, T_ref_w, row_pad);
  HandFrame cur(cam.cam, cur_p.data(), width, height, n_levels, T_cur_w_init, row_pad);
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
  // SparseImgAlign::SparseImgAlign(max_level, min_level, n_iter, method, false, false) as the drop-in writes it
  s->display_ = false; s->max_level_ = max_level; s->min_level_ = min_level;
  s->n_iter_ = n_iter; s->n_iter_init_ = s->n_iter_;
  s->method_ = method == 1 ? svo::SparseImgAlign::LevenbergMarquardt : svo::SparseImgAlign::GaussNewton;
  s->verbose_ = false;
  s->eps_ = 0.000001;
  if (scale_estimator != 0 || weight_function != 0)
    s->setRobustCostFunction((svo::SparseImgAlign::ScaleEstimatorType)scale_estimator, (svo::SparseImgAlign::WeightFunctionType)weight_function);
  const size_t ret = s->run(ref_frame, cur_frame);                   // the drop-in
  {
    const Eigen::Matrix<double, 6, 6> I = s->getFisherInformation();
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) fisher36[6 * i + j] = I(i, j);
  }
  *chi2_out = s->getChi2();
  *stop_out = s->stop_ ? 1 : 0;
  if (scale_mu_nu) { scale_mu_nu[0] = (double)s->scale_; scale_mu_nu[1] = s->mu_; scale_mu_nu[2] = s->nu_; }
  // tear down by hand (no destructor of the hand-laid object runs)
  s->scale_estimator_.reset();
  s->weight_function_.reset();
  s->ref_frame_.reset();
  s->cur_frame_.reset();
  from_se3(cur_frame->T_f_w_, T_cur_w_out);
  *n_tracked = ret;
  std::free(storage);
  return 0;
}

// pose_optimizer::optimizeGaussNewton(reproj_thresh, n_iter, false, frame, ...) as FrameHandlerMono::processFrame calls it
// (frame_handler_mono.cpp:226-229), through the DROP-IN (pose_optimizer_hip.cpp, compiled instead of pose_optimizer.cpp): the
// frame's features and their points in, frame->T_f_w_, frame->Cov_, the nulled Feature::point of rejected observations and the
// four outputs back.
int dropin_pose_optimize(int width, int height, double fx, double fy, double cx, double cy, const double* d5, double reproj_thresh,
                         int n_iter, const double* T_f_w_in, int n, const double* fv, const double* pos, const int32_t* level,
                         uint8_t* has_point_inout, double* T_f_w_out, double* cov36, double* outputs4 /*scale, init, final, num_obs*/) {
  HandPinhole cam(width, height, fx, fy, cx, cy, d5);
  HandFrame fr(cam.cam, nullptr, width, height, 0, T_f_w_in);
  const double px0[2] = {0.0, 0.0};
  std::vector<svo::Feature*> fts;
  for (int i = 0; i < n; ++i) fts.push_back(fr.add_feature(px0, fv + 3 * i, level[i], has_point_inout[i] ? pos + 3 * i : nullptr));
  svo::FramePtr frame = fr.ptr();
  double estimated_scale = 0.0, error_init = 0.0, error_final = 0.0;
  size_t num_obs = 0;
  svo::pose_optimizer::optimizeGaussNewton(reproj_thresh, (size_t)n_iter, false, frame, estimated_scale, error_init, error_final, num_obs);
  from_se3(frame->T_f_w_, T_f_w_out);
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) cov36[6 * r + c] = frame->Cov_(r, c);
  for (int i = 0; i < n; ++i) has_point_inout[i] = fts[i]->point != nullptr ? 1 : 0;
  outputs4[0] = estimated_scale; outputs4[1] = error_init; outputs4[2] = error_final; outputs4[3] = (double)num_obs;
  return 0;
}

// feature_alignment::align2D_batch of include/svo_dropin/feature_alignment_hip.h on a real svo::Frame: n 10x10 reference
// patches against pyramid level `level` of the frame's image, px in / out in level coordinates, convergence flags out.
int dropin_align2d_batch(int width, int height, const uint8_t* img, int n, const uint8_t* ref_patch_with_border, int n_iter,
                         double* px_inout, uint8_t* converged) {
  const double d0[5] = {0, 0, 0, 0, 0}, T0[7] = {0, 0, 0, 0, 0, 0, 1};
  HandPinhole cam(width, height, 100.0, 100.0, width / 2.0, height / 2.0, d0);
  const uint8_t* levels[1] = {img};
  HandFrame fr(cam.cam, levels, width, height, 1, T0);
  static thread_local svo::hip_bridge::Context ctx(0);
  static thread_local svo::hip_bridge::PyramidCache pyr(ctx.get(), 1);
  std::vector<uint8_t> pwb(ref_patch_with_border, ref_patch_with_border + (size_t)n * 100);
  std::vector<Eigen::Vector2d> px((size_t)n);
  for (int i = 0; i < n; ++i) px[i] = Eigen::Vector2d(px_inout[2 * i], px_inout[2 * i + 1]);
  const std::vector<bool> ok = svo::feature_alignment::align2D_batch(ctx, pyr, *fr.f, 0, pwb, n_iter, px);
  for (int i = 0; i < n; ++i) { px_inout[2 * i] = px[i][0]; px_inout[2 * i + 1] = px[i][1]; converged[i] = ok[i] ? 1 : 0; }
  return ctx.ok() ? 0 : -1;
}

// The depth filter binding's work on real Seed / Feature / Frame objects: hip_bridge::DeviceSeedMirror<std::list<Seed>> driven
// with hip_bridge::DepthFilterRefHost (depth_filter_hip.h) -- what DepthFilterHip::updateSeeds does under its mutex -- for a
// keyframe with n_seeds seeds over n_frames frames.  (DepthFilterHip itself cannot be instantiated here: its base class lives in
// depth_filter.cpp, which needs the NDK's <android/log.h>; Seed's constructor too, so the seeds are laid out by hand with the
// values depth_filter.cpp:36-45 assigns.)
// Outputs: the seeds still in the list (id, a, b, mu, sigma2) in list order, the convergence callbacks in firing order
// (seed id, point xyz, sigma2).
int dropin_depth_filter_frames(int width, int height, double fx, double fy, double cx, double cy, const double* d5, int n_levels,
                               const uint8_t* const* kf_pyr, const double* T_kf_w, int n_frames, const uint8_t* const* const* cur_pyr,
                               const double* T_cur_w, int n_seeds, const double* seed_px, const double* seed_f, const int32_t* seed_level,
                               double depth_mean, double depth_min, int n_pyr_levels, double* rows_out /*[n_seeds][5]*/, int* n_rows,
                               double* conv_out /*[n_seeds][5]*/, int* n_conv) {
  svo::Config::nPyrLevels() = (size_t)n_pyr_levels;
  HandPinhole cam(width, height, fx, fy, cx, cy, d5);
  HandFrame kf(cam.cam, kf_pyr, width, height, n_levels, T_kf_w);
  kf.f->is_keyframe_ = true;
  std::list<svo::Seed> seeds;
  std::vector<svo::Feature*> fts;
  for (int i = 0; i < n_seeds; ++i) {
    svo::Feature* ftr = new svo::Feature(kf.f, Eigen::Vector2d(seed_px[2 * i], seed_px[2 * i + 1]),
                                         Eigen::Vector3d(seed_f[3 * i], seed_f[3 * i + 1], seed_f[3 * i + 2]), seed_level[i]);
    fts.push_back(ftr);
    alignas(svo::Seed) unsigned char raw[sizeof(svo::Seed)];
    std::memset(raw, 0, sizeof(raw));
    svo::Seed& s0 = *reinterpret_cast<svo::Seed*>(raw);         // Seed::Seed(ftr, depth_mean, depth_min), depth_filter.cpp:36-45
    s0.batch_id = 0; s0.id = i; s0.ftr = ftr;
    const float dmean = (float)depth_mean, dmin = (float)depth_min;      // (the constructor's parameters are floats)
    s0.a = 10; s0.b = 10;
    s0.mu = 1.0 / dmean;
    s0.z_range = 1.0 / dmin;
    s0.sigma2 = s0.z_range * s0.z_range / 36;
    seeds.push_back(s0);
  }
  struct Fired { int id; double xyz[3]; double sigma2; };
  std::vector<Fired> fired;
  std::vector<svo::Point*> made;
  svo::DepthFilter::callback_t cb = [&](svo::Point* pt, double sigma2) {
    Fired fd;
    fd.id = -1;
    for (int i = 0; i < n_seeds; ++i) if (fts[i]->point == pt) fd.id = i;
    fd.xyz[0] = pt->pos_[0]; fd.xyz[1] = pt->pos_[1]; fd.xyz[2] = pt->pos_[2];
    fd.sigma2 = sigma2;
    fired.push_back(fd);
    made.push_back(pt);
  };
  int rc = 0;
  {
    svo::hip_bridge::Context ctx(0);
    svo::hip_bridge::PyramidCache kf_cache(ctx.get(), 8), cur_cache(ctx.get(), 2);
    svo::hip_bridge::DeviceSeedMirror<std::list<svo::Seed> > mirror;
    svo::hip_bridge::DepthFilterRefHost host;
    host.kf_pyr = &kf_cache; host.cur_pyr = &cur_cache; host.detector = nullptr; host.seed_converged_cb = &cb;
    svo_hip_df_params prm;
    prm.n_pyr_levels = n_pyr_levels; prm.align_max_iter = 10; prm.max_epi_search_steps = 1000;
    prm.seed_convergence_sigma2_thresh = svo::DepthFilter::Options().seed_convergence_sigma2_thresh;   // (I/depth_filter.h:78-87: 100)
    const volatile bool halt = false;
    if (!ctx.ok()) rc = -1;
    for (int k = 0; k < n_frames && rc == 0; ++k) {
      HandFrame cur(cam.cam, cur_pyr[k], width, height, n_levels, T_cur_w + 7 * k);
      const svo::hip_bridge::SeedBatchStats st = mirror.update(host, ctx.get(), seeds, *cur.f, prm, /*Seed::batch_counter*/ 1, /*max_n_kfs*/ 3, halt);
      if (st.n_device_errors) rc = -2;
    }
    if (rc == 0 && !mirror.syncToHost()) rc = -3;
    mirror.clear();
  }
  *n_rows = 0;
  for (const svo::Seed& s1 : seeds) {
    double* r = rows_out + 5 * (*n_rows)++;
    r[0] = s1.id; r[1] = s1.a; r[2] = s1.b; r[3] = s1.mu; r[4] = s1.sigma2;
  }
  *n_conv = (int)fired.size();
  for (size_t i = 0; i < fired.size(); ++i) {
    double* r = conv_out + 5 * i;
    r[0] = fired[i].id; r[1] = fired[i].xyz[0]; r[2] = fired[i].xyz[1]; r[3] = fired[i].xyz[2]; r[4] = fired[i].sigma2;
  }
  for (svo::Point* p2 : made) delete p2;
  for (svo::Feature* f2 : fts) delete f2;
  return rc;
}

}  // extern "C"
