// pose_optimizer_hip.cpp -- drop-in replacement of svo/pose_optimizer.cpp.
//
// Keeps the free function of I/pose_optimizer.h:36-45 (same name, parameters and side effects: frame->T_f_w_,
// frame->Cov_, (*it)->point = NULL for the observations the final test rejects, the four reference outputs) and
// forwards the work to libsvo_hip.so (svo_hip_pose_optimize).  Compile this file INSTEAD of pose_optimizer.cpp.
// `verbose` only controls logging in the reference; the summary line is kept.
//
// Also provides optimizeStructureHip(): the body of FrameHandlerBase::optimizeStructure
// (frame_handler_base.cpp:190-210) with the per-point Point::optimize calls batched into one launch
// (svo_hip_point_optimize_batch).  The selection of the points (nth_element on last_structure_optim_) and the
// bookkeeping stay on the host, unchanged.
#include <algorithm>
#include <deque>
#include <vector>

#include <svo/abstract_camera.h>
#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/point.h>
#include <svo/pose_optimizer.h>

#include "pose_optimizer_hip.h"
#include "svo_hip_bridge.h"

namespace svo {

namespace {
hip_bridge::Context& refineContext() {
  static thread_local hip_bridge::Context ctx(0);
  return ctx;
}
}  // namespace

namespace pose_optimizer {

void optimizeGaussNewton(const double reproj_thresh, const size_t n_iter, const bool verbose, FramePtr& frame,
                         double& estimated_scale, double& error_init, double& error_final, size_t& num_obs) {
  hip_bridge::Context& ctx = refineContext();
  if (!ctx.ok()) { hip_bridge::reportDeviceFailure(NULL, "pose_optimizer::optimizeGaussNewton"); return; }   // nothing is touched
  const int n = (int)frame->fts_.size();
  std::vector<double> f(3 * (size_t)n), pos(3 * (size_t)n, 0.0);
  std::vector<int32_t> level((size_t)n, 0);
  std::vector<uint8_t> has_point((size_t)n, 0);
  size_t i = 0;
  for (Features::iterator it = frame->fts_.begin(); it != frame->fts_.end(); ++it, ++i) {
    const Feature* ftr = *it;
    f[3 * i] = ftr->f[0]; f[3 * i + 1] = ftr->f[1]; f[3 * i + 2] = ftr->f[2];
    level[i] = ftr->level;
    if (ftr->point != NULL) {
      has_point[i] = 1;
      pos[3 * i] = ftr->point->pos_[0]; pos[3 * i + 1] = ftr->point->pos_[1]; pos[3 * i + 2] = ftr->point->pos_[2];
    }
  }
  double T[7];
  hip_bridge::toPose7(frame->T_f_w_, T);
  svo_hip_pose_opt_result res;
  const int rc = svo_hip_pose_optimize(ctx.get(), n, T, f.data(), pos.data(), level.data(), has_point.data(),
                                       frame->cam_->errorMultiplier2(), reproj_thresh, (int)n_iter, &res);
  if (rc != SVO_HIP_OK) { hip_bridge::reportDeviceFailure(ctx.get(), "pose_optimizer::optimizeGaussNewton"); return; }
  if (!res.ran) return;                                    // pose_optimizer.cpp:61-62: returns before touching anything
  frame->T_f_w_ = hip_bridge::fromPose7(res.T_f_w);
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) frame->Cov_(r, c) = res.Cov[6 * r + c];
  i = 0;
  for (Features::iterator it = frame->fts_.begin(); it != frame->fts_.end(); ++it, ++i)
    if ((*it)->point != NULL && !has_point[i]) (*it)->point = NULL;          // :154-157
  estimated_scale = res.estimated_scale;
  error_init = res.error_init;
  error_final = res.error_final;
  num_obs = (size_t)res.num_obs;
  if (verbose) {
#ifdef ANDROID        // the logging macros are printf-style on Android and stream-style elsewhere (I/global.h:39-69)
    SVO_INFO_STREAM("n deleted obs = %d\t scale = %f\t error init = %f\t error end = %f", res.n_deleted,
                    estimated_scale, error_init, error_final);
#else
    SVO_INFO_STREAM("n deleted obs = " << res.n_deleted << "\t scale = " << estimated_scale << "\t error init = "
                    << error_init << "\t error end = " << error_final);
#endif
  }
}

}  // namespace pose_optimizer

namespace {
bool ptLastOptimComparator(Point* lhs, Point* rhs) { return (lhs->last_structure_optim_ < rhs->last_structure_optim_); }
}  // namespace

/// FrameHandlerBase::optimizeStructure (frame_handler_base.cpp:190-210) with the Point::optimize calls batched.
void optimizeStructureHip(FramePtr frame, size_t max_n_pts, int max_iter) {
  std::deque<Point*> pts;
  for (Features::iterator it = frame->fts_.begin(); it != frame->fts_.end(); ++it)
    if ((*it)->point != NULL) pts.push_back((*it)->point);
  max_n_pts = std::min(max_n_pts, pts.size());
  std::nth_element(pts.begin(), pts.begin() + max_n_pts, pts.end(), ptLastOptimComparator);
  if (max_n_pts == 0) return;
  hip_bridge::Context& ctx = refineContext();
  std::vector<double> pos, obs_T, obs_f;
  std::vector<int32_t> offset(1, 0);
  for (size_t k = 0; k < max_n_pts; ++k) {
    Point* pt = pts[k];
    pos.push_back(pt->pos_[0]); pos.push_back(pt->pos_[1]); pos.push_back(pt->pos_[2]);
    for (std::list<Feature*>::iterator it = pt->obs_.begin(); it != pt->obs_.end(); ++it) {   // list order = summation order
      double T[7];
      hip_bridge::toPose7((*it)->frame->T_f_w_, T);
      obs_T.insert(obs_T.end(), T, T + 7);
      obs_f.push_back((*it)->f[0]); obs_f.push_back((*it)->f[1]); obs_f.push_back((*it)->f[2]);
    }
    offset.push_back((int32_t)(obs_f.size() / 3));
  }
  const bool done = ctx.ok() &&
                    svo_hip_point_optimize_batch(ctx.get(), (int)max_n_pts, max_iter, pos.data(), offset.data(),
                                                 obs_T.empty() ? NULL : obs_T.data(), obs_f.empty() ? NULL : obs_f.data(),
                                                 NULL) == SVO_HIP_OK;
  if (!done) {                                              // no CPU stand-in: the points stay as they are, and the log says why
    hip_bridge::reportDeviceFailure(ctx.ok() ? ctx.get() : NULL, "optimizeStructureHip");
    return;
  }
  for (size_t k = 0; k < max_n_pts; ++k) {
    Point* pt = pts[k];
    pt->pos_ = Vector3d(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]);
    pt->last_structure_optim_ = frame->id_;
  }
}

}  // namespace svo
