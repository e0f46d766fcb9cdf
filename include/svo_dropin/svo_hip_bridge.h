// svo_hip_bridge.h -- glue between the reference's C++ data model (svo::Frame / Feature /
// Point / Seed, Eigen, cv::Mat) and the C-ABI of libsvo_hip.so (include/svo_hip.h).
//
// These files are meant to be dropped into the reference tree (app/src/main/cpp/svo/) and
// compiled with the reference's own headers; they contain no device code.  See INTEGRATION.md.
#ifndef SVO_HIP_BRIDGE_H_
#define SVO_HIP_BRIDGE_H_

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <cstdio>
#include <vector>

#include <svo/abstract_camera.h>
#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/global.h>
#include <svo/pinhole_camera.h>
#include <svo/point.h>

#include "slot_table.h"
#include "svo_hip.h"

namespace svo {
namespace hip_bridge {

inline void toPose7(const SE3& T, double out[7]) {
  out[0] = T.get_translation().x; out[1] = T.get_translation().y; out[2] = T.get_translation().z;
  out[3] = T.get_rotation().x; out[4] = T.get_rotation().y; out[5] = T.get_rotation().z;
  out[6] = T.get_rotation().w;
}

inline SE3 fromPose7(const double p[7]) { return SE3(p[0], p[1], p[2], p[3], p[4], p[5], p[6]); }

/// Camera parameters of a vk::AbstractCamera: the pinhole model exposes fx..d4; any other
/// model is approximated by its error multiplier (distortion-free), as the harness camera does.
inline svo_hip_camera toCamera(const vk::AbstractCamera* cam) {
  svo_hip_camera c;
  c.width = cam->width(); c.height = cam->height();
  c.distortion = 0;
  for (int i = 0; i < 5; ++i) c.d[i] = 0.0;
  const vk::PinholeCamera* ph = dynamic_cast<const vk::PinholeCamera*>(cam);
  if (ph) {
    c.fx = ph->fx(); c.fy = ph->fy(); c.cx = ph->cx(); c.cy = ph->cy();
    c.d[0] = ph->d0(); c.d[1] = ph->d1(); c.d[2] = ph->d2(); c.d[3] = ph->d3(); c.d[4] = ph->d4();
    c.distortion = std::fabs(ph->d0()) > 0.0000001;       // pinhole_camera.cpp:27
  } else {
    // another camera class: the device knows the pinhole / radtan model only.  The intrinsics are read off the camera's own
    // projection of the unit plane (exact for any distortion-free pinhole: fl(f*1 + c) - c == f whenever both are exact in
    // f64, as half-pixel principal points and integer-ish focal lengths are), and the model is checked at two more points: a
    // camera that is not a plain pinhole is reported, loudly, instead of being tracked with the wrong projection.
    const Vector2d o = cam->world2cam(Vector2d(0.0, 0.0)), ex = cam->world2cam(Vector2d(1.0, 0.0)), ey = cam->world2cam(Vector2d(0.0, 1.0));
    c.cx = o[0]; c.cy = o[1];
    c.fx = ex[0] - o[0]; c.fy = ey[1] - o[1];
    const Vector2d p = cam->world2cam(Vector2d(-0.37, 0.23)), q = cam->world2cam(Vector2d(0.51, -0.29));
    const double err = std::max(std::max(std::fabs(p[0] - (c.fx * -0.37 + c.cx)), std::fabs(p[1] - (c.fy * 0.23 + c.cy))),
                                std::max(std::fabs(q[0] - (c.fx * 0.51 + c.cx)), std::fabs(q[1] - (c.fy * -0.29 + c.cy))));
    if (!(err < 1e-9) || ex[1] != o[1] || ey[0] != o[0]) {
      fprintf(stderr, "[svo_hip] the camera is neither a vk::PinholeCamera nor a distortion-free pinhole (projection differs by %g px): "
                      "the device path does not support this model\n", err);
      fflush(stderr);
    }
  }
  return c;
}

/// A device-side failure is never papered over with the CPU path: it is reported on stderr, loudly, and the caller degrades
/// to the reference's own "nothing found" outcome (tracking failure, no seed update) -- visible in the log and in the
/// tracking quality, never silent and never a different arithmetic.
inline void reportDeviceFailure(svo_hip_ctx* ctx, const char* where) {
  fprintf(stderr, "[svo_hip] %s FAILED: %s\n", where, ctx ? svo_hip_last_error(ctx) : "no device context (libsvo_hip.so / GPU unavailable)");
  fflush(stderr);
}

/// One context (stream) per host thread that enters the library (tracking thread,
/// depth-filter thread): SURVEY 8b "Threading".
class Context {
 public:
  explicit Context(int device = 0) : ctx_(NULL) {
    if (svo_hip_ctx_create(&ctx_, device, NULL) != SVO_HIP_OK) ctx_ = NULL;
  }
  ~Context() { if (ctx_) svo_hip_ctx_destroy(ctx_); }
  svo_hip_ctx* get() const { return ctx_; }
  bool ok() const { return ctx_ != NULL; }
 private:
  Context(const Context&);
  Context& operator=(const Context&);
  svo_hip_ctx* ctx_;
};

/// Device copy of a Frame's image pyramid, cached by frame id so a keyframe is uploaded once (slot_table.h says why a
/// call that needs several frames has to resolve them together).
class PyramidCache {
 public:
  PyramidCache(svo_hip_ctx* ctx, int capacity) : ctx_(ctx), pyr_(NULL), width_(0), height_(0), n_levels_(0), table_(capacity) {}
  ~PyramidCache() { if (pyr_) svo_hip_pyramid_destroy(pyr_); }

  /// slot holding `frame`'s pyramid, uploading it (all levels, stride == cols) if needed; -1 on error.  For ONE frame per
  /// device call; a call that reads several frames' pyramids uses acquire().
  int slotOf(const Frame& frame) {
    std::vector<const Frame*> one(1, &frame);
    std::vector<int> slots;
    return acquire(one, slots) ? slots[0] : -1;
  }
  /// Slots holding the pyramids of ALL `frames` at once (the reprojector matches against every keyframe that observes a
  /// candidate point, the depth filter against every keyframe that still owns seeds: the map is unbounded, so is their
  /// number).  Frames already resident keep their slot; the others go to slots no frame of this call uses; the cache
  /// grows (to a multiple of 16 slots) when the set does not fit.  Returns false on a device error.
  bool acquire(const std::vector<const Frame*>& frames, std::vector<int>& slots) {
    slots.assign(frames.size(), -1);
    if (frames.empty()) return true;
    std::vector<int> ids(frames.size());
    int distinct = 0;
    for (size_t k = 0; k < frames.size(); ++k) {
      ids[k] = frames[k]->id_;
      bool seen = false;
      for (size_t m = 0; m < k && !seen; ++m) seen = ids[m] == ids[k];
      if (!seen) ++distinct;
    }
    const int need = table_.capacityFor(distinct);
    // every frame of a call has the geometry of the first; a cache that holds another geometry (the camera's resolution or
    // the number of pyramid levels changed since) starts again: what it held cannot be what these frames are
    const cv::Mat& l0 = frames[0]->img_pyr_[0];
    const int n_levels = (int)frames[0]->img_pyr_.size();
    for (size_t k = 1; k < frames.size(); ++k) {
      const cv::Mat& lk = frames[k]->img_pyr_[0];
      if (lk.cols != l0.cols || lk.rows != l0.rows || (int)frames[k]->img_pyr_.size() != n_levels) return false;
    }
    if (need != table_.capacity() || !pyr_ || l0.cols != width_ || l0.rows != height_ || n_levels != n_levels_) {
      if (pyr_) { svo_hip_pyramid_destroy(pyr_); pyr_ = NULL; }
      if (svo_hip_pyramid_create(ctx_, l0.cols, l0.rows, n_levels, need, &pyr_) != SVO_HIP_OK) return false;
      width_ = l0.cols; height_ = l0.rows; n_levels_ = n_levels;
      table_.reset(need);
    }
    Uploader up = {this, &frames};
    return table_.acquire(ids, slots, up);
  }
  int capacity() const { return table_.capacity(); }
  svo_hip_pyramid* pyramid() const { return pyr_; }

 private:
  struct Uploader {
    PyramidCache* self;
    const std::vector<const Frame*>* frames;
    bool operator()(size_t k, int slot) const { return self->upload(*(*frames)[k], slot); }
  };
  bool upload(const Frame& frame, int s) {
    const int n_levels = (int)frame.img_pyr_.size();
    std::vector<const uint8_t*> levels(SVO_HIP_MAX_LEVELS, (const uint8_t*)NULL);
    std::vector<std::vector<uint8_t> > packed(n_levels);
    for (int l = 0; l < n_levels; ++l) {
      const cv::Mat& m = frame.img_pyr_[l];
      if ((int)m.step.p[0] == m.cols) {
        levels[l] = m.data;
      } else {                       // compact a padded Mat: the kernels assume stride == cols
        packed[l].resize((size_t)m.rows * m.cols);
        for (int y = 0; y < m.rows; ++y) memcpy(&packed[l][(size_t)y * m.cols], m.data + (size_t)y * m.step.p[0], m.cols);
        levels[l] = packed[l].data();
      }
    }
    if (svo_hip_pyramid_upload(pyr_, s, levels.data()) != SVO_HIP_OK) return false;
    return svo_hip_ctx_sync(ctx_) == SVO_HIP_OK;
  }

  svo_hip_ctx* ctx_;
  svo_hip_pyramid* pyr_;
  int width_, height_, n_levels_;   // geometry of pyr_
  SlotTable table_;
};

}  // namespace hip_bridge
}  // namespace svo

#endif  // SVO_HIP_BRIDGE_H_
