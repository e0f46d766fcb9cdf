// frame_tracker_hip.h -- the first half of FrameHandlerMono::processFrame (S/frame_handler_mono.cpp:171-229: sparse image
// alignment against the last frame, map reprojection, pose refinement) as ONE device call per frame
// (svo_hip_tracker_track: one stream, one synchronisation; the aligned pose, the candidates and the matches never leave
// the device between the stages, the frame's matches are the next call's reference features).
//
// The host logic -- flattening svo::Map into the tracker's index tables, applying a frame's outcome to the reference's own
// objects with the reference's own functions -- is hip_bridge::FrameTrackerT (frame_tracker_batch.h); this header
// instantiates it on the reference's types.  The same template runs on the GPU against the self-contained twins of those
// types (android_svo_amd/host/svo_host.h, tests/test_gpu_host_cpp.py).
// INTEGRATION.md shows the edit of processFrame; `make -C oracle dropin-check` compiles this file against the
// reference's headers.
#ifndef SVO_FRAME_TRACKER_HIP_H_
#define SVO_FRAME_TRACKER_HIP_H_

#include <algorithm>

#include <svo/config.h>
#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/map.h>
#include <svo/point.h>

#include "frame_tracker_batch.h"
#include "svo_hip_bridge.h"

namespace svo {
namespace hip_bridge {

/// Host policy of FrameTrackerT on the reference's data model
struct SvoTrackerHost {
  typedef svo::Frame Frame;
  typedef svo::FramePtr FramePtr;
  typedef svo::Feature Feature;
  typedef svo::Point Point;
  typedef svo::Map Map;
  typedef svo::Features FeatureList;
  typedef svo::MapPointCandidates::PointCandidateList CandidateList;
  static void pose7(const Frame& fr, double T[7]) { toPose7(fr.T_f_w_, T); }
  static void setPose(Frame& fr, const double T[7]) { fr.T_f_w_ = fromPose7(T); }
  static const uint8_t* level0(const Frame& fr, int* stride, int* cols, int* rows) {
    const cv::Mat& img = fr.img_pyr_[0];
    *stride = (int)img.step.p[0]; *cols = img.cols; *rows = img.rows;
    return img.data;
  }
  static Feature* makeFeature(Frame* fr, const double px[2], const double f[3], int level) {
    return new Feature(fr, Vector2d(px[0], px[1]), Vector3d(f[0], f[1], f[2]), level);
  }
  static void setEdgelet(Feature& ftr, const double g[2]) { ftr.type = Feature::EDGELET; ftr.grad = Vector2d(g[0], g[1]); }
  static bool isEdgelet(const Feature& ftr) { return ftr.type == Feature::EDGELET; }
  static void setCov(Frame& fr, const double cov[36]) {
    Matrix<double, 6, 6> c;
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) c(a, b) = cov[6 * a + b];
    fr.Cov_ = c;
  }
  /// svo_hip_tracker_config from svo::Config (the values processFrame's stages read)
  static svo_hip_tracker_config config(int max_keyframes) {
    svo_hip_tracker_config cfg;
    svo_hip_tracker_default_config(&cfg);
    cfg.max_keyframes = max_keyframes;
    cfg.n_levels = (int)std::max(Config::nPyrLevels(), Config::kltMaxLevel() + 1);      // frame.cpp:63
    cfg.klt_max_level = (int)Config::kltMaxLevel(); cfg.klt_min_level = (int)Config::kltMinLevel();
    cfg.grid_size = (int)Config::gridSize(); cfg.max_fts = (int)Config::maxFts(); cfg.quality_min_fts = (int)Config::qualityMinFts();
    cfg.n_pyr_levels = (int)Config::nPyrLevels();
    cfg.pose_optim_thresh = Config::poseOptimThresh(); cfg.pose_optim_num_iter = (int)Config::poseOptimNumIter();
    return cfg;
  }
};

class FrameTracker : public FrameTrackerT<SvoTrackerHost> {
 public:
  explicit FrameTracker(vk::AbstractCamera* cam, int max_keyframes = 256)
      : FrameTrackerT<SvoTrackerHost>(toCamera(cam), SvoTrackerHost::config(max_keyframes)) {}
};

/// Several FrameHandlerMono objects (cameras, or sequences replayed side by side) tracked together: one chain of launches per
/// call for all of them (svo_hip_tracker_group).  camera(c) is camera c's FrameTrackerT; trackAll replaces the three stages of
/// every camera's processFrame (INTEGRATION.md "Several cameras per call").
class FrameTrackerGroup : public FrameTrackerGroupT<SvoTrackerHost> {
 public:
  FrameTrackerGroup(vk::AbstractCamera* cam, int n_cameras, int max_keyframes = 256)
      : FrameTrackerGroupT<SvoTrackerHost>(toCamera(cam), groupConfig(max_keyframes), n_cameras) {}
 private:
  static svo_hip_tracker_config groupConfig(int max_keyframes) {
    svo_hip_tracker_config cfg = SvoTrackerHost::config(max_keyframes);
    cfg.max_items = (cfg.max_items + 15) / 16 * 16;          // a block of the batched warp stage takes 16 candidates of ONE camera
    return cfg;
  }
};

}  // namespace hip_bridge
}  // namespace svo

#endif  // SVO_FRAME_TRACKER_HIP_H_
