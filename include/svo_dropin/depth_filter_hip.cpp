// depth_filter_hip.cpp -- see depth_filter_hip.h.
#include <algorithm>
#include <map>
#include <vector>

#include <svo/config.h>
#include <svo/feature.h>
#include <svo/feature_detection.h>
#include <svo/frame.h>
#include <svo/point.h>

#include "depth_filter_hip.h"

namespace svo {

DepthFilterHip::DepthFilterHip(feature_detection::DetectorPtr feature_detector, callback_t seed_converged_cb)
    : DepthFilter(feature_detector, seed_converged_cb), ctx_(0), kf_pyr_(ctx_.get(), 8), cur_pyr_(ctx_.get(), 2) {}

DepthFilterHip::~DepthFilterHip() { stopThread(); }

void DepthFilterHip::updateSeeds(FramePtr frame) {
  lock_t lock(seeds_mut_);
  if (!ctx_.ok() || seeds_.empty()) return;

  // age-out first (depth_filter.cpp:256-261), then bucket the survivors by reference keyframe
  std::map<Frame*, std::vector<std::list<Seed>::iterator> > by_kf;
  for (std::list<Seed>::iterator it = seeds_.begin(); it != seeds_.end();) {
    if (seeds_updating_halt_) return;
    if ((Seed::batch_counter - it->batch_id) > options_.max_n_kfs) { it = seeds_.erase(it); continue; }
    by_kf[it->ftr->frame].push_back(it);
    ++it;
  }
  const svo_hip_camera cam = hip_bridge::toCamera(frame->cam_);
  const int cur_slot = cur_pyr_.slotOf(*frame);
  if (cur_slot < 0) return;
  double T_cur[7];
  hip_bridge::toPose7(frame->T_f_w_, T_cur);
  svo_hip_df_params prm;
  prm.n_pyr_levels = (int)Config::nPyrLevels();
  prm.align_max_iter = 10;             // Matcher::Options defaults (I/matcher.h:83-91)
  prm.max_epi_search_steps = 1000;
  prm.seed_convergence_sigma2_thresh = options_.seed_convergence_sigma2_thresh;

  for (std::map<Frame*, std::vector<std::list<Seed>::iterator> >::iterator kf = by_kf.begin(); kf != by_kf.end(); ++kf) {
    // the halt flag is honoured at batch boundaries (depth_filter.cpp:253)
    if (seeds_updating_halt_) return;
    Frame* ref = kf->first;
    std::vector<std::list<Seed>::iterator>& its = kf->second;
    const int n = (int)its.size();
    const int ref_slot = kf_pyr_.slotOf(*ref);
    if (ref_slot < 0) continue;
    std::vector<double> px(2 * (size_t)n), f(3 * (size_t)n), z((size_t)n), xyz(3 * (size_t)n);
    std::vector<int32_t> level((size_t)n), status((size_t)n);
    std::vector<float> a((size_t)n), b((size_t)n), mu((size_t)n), zr((size_t)n), s2((size_t)n);
    for (int i = 0; i < n; ++i) {
      const Seed& s = *its[i];
      px[2 * i] = s.ftr->px[0]; px[2 * i + 1] = s.ftr->px[1];
      f[3 * i] = s.ftr->f[0]; f[3 * i + 1] = s.ftr->f[1]; f[3 * i + 2] = s.ftr->f[2];
      level[i] = s.ftr->level;
      a[i] = s.a; b[i] = s.b; mu[i] = s.mu; zr[i] = s.z_range; s2[i] = s.sigma2;
    }
    double T_ref[7];
    hip_bridge::toPose7(ref->T_f_w_, T_ref);
    const int rc = svo_hip_depth_filter_update(ctx_.get(), kf_pyr_.pyramid(), ref_slot, cur_pyr_.pyramid(), cur_slot, &cam,
                                               T_ref, T_cur, n, px.data(), f.data(), level.data(), a.data(), b.data(),
                                               mu.data(), zr.data(), s2.data(), &prm, status.data(), z.data(), xyz.data(),
                                               NULL, NULL);
    if (rc != SVO_HIP_OK) continue;     // device error: this batch keeps its old state
    for (int i = 0; i < n; ++i) {
      std::list<Seed>::iterator it = its[i];
      it->a = a[i]; it->b = b[i]; it->mu = mu[i]; it->sigma2 = s2[i];
      if (status[i] == SVO_HIP_SEED_CONVERGED) {
        // depth_filter.cpp:310-331: hand the new point to the candidate list, drop the seed
        Vector3d xyz_world(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        Point* point = new Point(xyz_world, it->ftr);
        it->ftr->point = point;
        seed_converged_cb_(point, it->sigma2);
        seeds_.erase(it);
      } else if (status[i] == SVO_HIP_SEED_NAN) {
        seeds_.erase(it);               // :333-337
      }
      // (the keyframe-only feature_detector_->setGridOccpuancy(px_cur) of :302-306 needs the matched
      //  pixel; it is applied by initializeSeeds through setExistingFeatures for converged points)
    }
  }
}

}  // namespace svo
