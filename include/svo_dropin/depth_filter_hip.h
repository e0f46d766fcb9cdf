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
#include <svo/feature.h>
#include <svo/feature_detection.h>
#include <svo/frame.h>
#include <svo/point.h>

#include "depth_filter_batch.h"
#include "svo_hip_bridge.h"

namespace svo {

namespace hip_bridge {
/// Host policy of hip_bridge::DeviceSeedMirror on the reference's data model: how a Seed's feature, its keyframe, the frames'
/// pyramids and poses are read, and what happens on the host when a seed converges (depth_filter.cpp:302-331).  What
/// DepthFilterHip::updateSeeds hands to the mirror; tests/test_gpu_dropin_binding.py runs the mirror with it on real Seed /
/// Feature / Frame objects.
struct DepthFilterRefHost {
  hip_bridge::PyramidCache* kf_pyr;
  hip_bridge::PyramidCache* cur_pyr;
  feature_detection::AbstractDetector* detector;
  DepthFilter::callback_t* seed_converged_cb;

  Frame* keyframeOf(const Seed& s) const { return s.ftr->frame; }
  void feature(const Seed& s, double px[2], double f[3], int* level) const {
    px[0] = s.ftr->px[0]; px[1] = s.ftr->px[1];
    f[0] = s.ftr->f[0]; f[1] = s.ftr->f[1]; f[2] = s.ftr->f[2];
    *level = s.ftr->level;
  }
  void pose7(const Frame& fr, double T[7]) const { hip_bridge::toPose7(fr.T_f_w_, T); }
  bool keyframeSlots(const std::vector<Frame*>& kfs, std::vector<int>& slots) {
    std::vector<const Frame*> c(kfs.begin(), kfs.end());
    return kf_pyr->acquire(c, slots);
  }
  int currentSlot(Frame& fr) { return cur_pyr->slotOf(fr); }
  svo_hip_pyramid* keyframePyramids() const { return kf_pyr->pyramid(); }
  svo_hip_pyramid* currentPyramids() const { return cur_pyr->pyramid(); }
  svo_hip_camera camera(const Frame& fr) const { return hip_bridge::toCamera(fr.cam_); }
  bool isKeyframe(const Frame& fr) const { return fr.isKeyframe(); }
  void setGridOccupancy(const double px_cur[2]) { detector->setGridOccpuancy(Vector2d(px_cur[0], px_cur[1])); }   // depth_filter.cpp:302-306
  void converged(Seed& s, const double xyz[3]) {                                                                 // :310-331
    Point* point = new Point(Vector3d(xyz[0], xyz[1], xyz[2]), s.ftr);
    s.ftr->point = point;
    (*seed_converged_cb)(point, s.sigma2);
  }
};
}  // namespace hip_bridge

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
