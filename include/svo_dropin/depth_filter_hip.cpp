// depth_filter_hip.cpp -- see depth_filter_hip.h.  The device mirror of the seed list, the ordering and the halt logic are
// hip_bridge::DeviceSeedMirror (depth_filter_batch.h), shared with the executable host layer
// android_svo_amd/host/svo_host.h; this file only adapts the reference's types to it.
#include <svo/config.h>
#include <svo/feature.h>
#include <svo/feature_detection.h>
#include <svo/frame.h>
#include <svo/point.h>

#include "depth_filter_batch.h"
#include "depth_filter_hip.h"

namespace svo {

typedef hip_bridge::DepthFilterRefHost RefHost;      // (depth_filter_hip.h)

DepthFilterHip::DepthFilterHip(feature_detection::DetectorPtr feature_detector, callback_t seed_converged_cb)
    : DepthFilter(feature_detector, seed_converged_cb), ctx_(0), kf_pyr_(ctx_.get(), 8), cur_pyr_(ctx_.get(), 2) {}

DepthFilterHip::~DepthFilterHip() { stopThread(); mirror_.clear(); }

bool DepthFilterHip::syncSeeds() {
  lock_t lock(seeds_mut_);
  return mirror_.syncToHost();
}

void DepthFilterHip::updateSeeds(FramePtr frame) {
  lock_t lock(seeds_mut_);
  if (seeds_.empty()) { mirror_.clear(); return; }
  if (!ctx_.ok()) { hip_bridge::reportDeviceFailure(NULL, "DepthFilterHip::updateSeeds"); return; }
  svo_hip_df_params prm;
  prm.n_pyr_levels = (int)Config::nPyrLevels();
  prm.align_max_iter = 10;             // Matcher::Options defaults (I/matcher.h:83-91)
  prm.max_epi_search_steps = 1000;
  prm.seed_convergence_sigma2_thresh = options_.seed_convergence_sigma2_thresh;
  RefHost host;
  host.kf_pyr = &kf_pyr_; host.cur_pyr = &cur_pyr_;
  host.detector = feature_detector_.get(); host.seed_converged_cb = &seed_converged_cb_;
  const hip_bridge::SeedBatchStats st =
      mirror_.update(host, ctx_.get(), seeds_, *frame, prm, Seed::batch_counter, options_.max_n_kfs, seeds_updating_halt_);
  // a device failure is never silent: the seeds it did not reach keep their state (as after a failed match of the reference)
  if (st.n_device_errors) hip_bridge::reportDeviceFailure(ctx_.get(), "DepthFilterHip::updateSeeds");
}

std::list<Seed>& DepthFilterHip::getSeeds() {
  syncSeeds();
  return seeds_;
}

void DepthFilterHip::getSeedsCopy(const FramePtr& frame, std::list<Seed>& seeds) {
  syncSeeds();
  DepthFilter::getSeedsCopy(frame, seeds);
}

}  // namespace svo
