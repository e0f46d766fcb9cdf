// depth_filter_hip.h -- svo::DepthFilterHip: the reference's DepthFilter with the per-seed work
// of updateSeeds() (I/depth_filter.h:155, S/depth_filter.cpp:237-341) done on the GPU.
//
// DepthFilter::updateSeeds is `protected virtual` in the reference precisely as an override
// point; everything else (thread, frame queue, keyframe hand-off, seed list, callback) is inherited
// unchanged.  FrameHandlerMono::initialize (frame_handler_mono.cpp:46-50) constructs this class
// instead of DepthFilter -- a one-line change described in INTEGRATION.md.
#ifndef SVO_DEPTH_FILTER_HIP_H_
#define SVO_DEPTH_FILTER_HIP_H_

#include <condition_variable>

#include <svo/depth_filter.h>

#include "depth_filter_batch.h"
#include "svo_hip_bridge.h"

namespace svo {

class DepthFilterHip : public DepthFilter {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  DepthFilterHip(feature_detection::DetectorPtr feature_detector, callback_t seed_converged_cb);
  virtual ~DepthFilterHip();

  /// Between frames the seeds' state (a, b, mu, sigma2) lives on the device; the list entries are brought up to date by
  /// this call -- make it before reading getSeeds() / getSeedsCopy() (both non-virtual in the reference; the app never
  /// calls them).  Returns false on a device error.
  bool syncSeeds();
  /// The reference's accessors (I/depth_filter.h:117-123, non-virtual) with the synchronisation in front: a caller that
  /// holds a DepthFilterHip* (FrameHandlerMono::depth_filter_ after the one-line change of INTEGRATION.md is declared
  /// DepthFilter*: cast, or call syncSeeds() first) reads the device's a / b / mu / sigma2, not the construction-time values.
  std::list<Seed>& getSeeds();
  void getSeedsCopy(const FramePtr& frame, std::list<Seed>& seeds);

 protected:
  /// One pass over all seeds against `frame`, batched per reference keyframe.
  virtual void updateSeeds(FramePtr frame);

 private:
  hip_bridge::Context ctx_;            // the depth-filter thread's own stream
  hip_bridge::PyramidCache kf_pyr_;    // keyframes that still own seeds (max_n_kfs + 1 batches alive)
  hip_bridge::PyramidCache cur_pyr_;
  hip_bridge::DeviceSeedMirror<std::list<Seed> > mirror_;   // per-keyframe seed batches resident on the device (under seeds_mut_)
};

}  // namespace svo

#endif  // SVO_DEPTH_FILTER_HIP_H_
