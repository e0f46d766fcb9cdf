// feature_detection_hip.h -- svo::feature_detection::FastDetectorHip: the reference's FastDetector
// (I/feature_detection.h:90-103, S/feature_detection.cpp:65-122) with cv::FAST, the Shi-Tomasi scores and the
// one-corner-per-cell selection done on the GPU (svo_hip_detect_features).
//
// Same base class, same detect() signature and side effects: appends `new Feature(frame, px, level)` in grid-cell
// order and resets the occupancy grid.  Construct it instead of FastDetector at frame_handler_mono.cpp:43-45
// (one line); DepthFilter::initializeSeeds and everything else is unchanged.
#ifndef SVO_FEATURE_DETECTION_HIP_H_
#define SVO_FEATURE_DETECTION_HIP_H_

#include <vector>

#include <svo/feature.h>
#include <svo/feature_detection.h>
#include <svo/frame.h>

#include "svo_hip_bridge.h"

namespace svo {
namespace feature_detection {

class FastDetectorHip : public AbstractDetector {
 public:
  FastDetectorHip(const int img_width, const int img_height, const int cell_size, const int n_pyr_levels)
      : AbstractDetector(img_width, img_height, cell_size, n_pyr_levels), ctx_(0), pyr_(ctx_.get(), 2) {}
  virtual ~FastDetectorHip() {}

  virtual void detect(Frame* frame, const ImgPyr& img_pyr, const double detection_threshold, Features& fts) {
    (void)img_pyr;                                   // == frame->img_pyr_ at the only call site (depth_filter.cpp:135)
    const size_t n_cells = grid_occupancy_.size();
    bool done = false;
    if (ctx_.ok()) {
      const int slot = pyr_.slotOf(*frame);
      std::vector<uint8_t> occ(n_cells);
      for (size_t k = 0; k < n_cells; ++k) occ[k] = grid_occupancy_[k] ? 1 : 0;
      std::vector<double> px(2 * n_cells);
      std::vector<int32_t> level(n_cells);
      int32_t n = 0;
      if (slot >= 0 && svo_hip_detect_features(ctx_.get(), pyr_.pyramid(), slot, NULL, n_pyr_levels_, cell_size_, occ.data(),
                                               detection_threshold, &n, px.data(), NULL, level.data(), NULL) == SVO_HIP_OK) {
        for (int32_t i = 0; i < n; ++i)             // Feature's constructor evaluates f = cam->cam2world(px) (any camera model)
          fts.push_back(new Feature(frame, Vector2d(px[2 * i], px[2 * i + 1]), level[i]));
        done = true;
      }
    }
    if (!done) SVO_WARN_STREAM("FastDetectorHip: device unavailable, no features detected");
    resetGrid();
  }

 private:
  hip_bridge::Context ctx_;
  hip_bridge::PyramidCache pyr_;
};

}  // namespace feature_detection
}  // namespace svo

#endif  // SVO_FEATURE_DETECTION_HIP_H_
