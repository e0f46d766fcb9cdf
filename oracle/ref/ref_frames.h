// ref_frames.h -- a svo::Frame laid out by hand (TEST INFRASTRUCTURE ONLY; used by ref_objects.cpp and dropin_run.cpp).
// Include after <svo/frame.h>, <svo/feature.h>, <svo/point.h> and ref_common.h.
#pragma once
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace refh {

static_assert(sizeof(std::vector<cv::Mat>) == 3 * sizeof(void*), "libstdc++ vector = {begin, end, end_of_storage}");

// A svo::Frame in raw storage: pose, camera, feature list and an image pyramid that aliases the caller's
// level buffers.  Frame::Frame (frame.cpp) is never called.
struct HandFrame {
  void* storage = nullptr;
  void* mats = nullptr;
  svo::Frame* f = nullptr;
  std::vector<svo::Point*> points;

  /// row_pad: extra bytes at the end of every row of the caller's level buffers (a cv::Mat that is not continuous)
  HandFrame(vk::AbstractCamera* cam, const uint8_t* const* pyr, int width, int height, int n_levels, const double* T_f_w, int row_pad = 0) {
    storage = ::aligned_alloc(32, (sizeof(svo::Frame) + 31) / 32 * 32);
    std::memset(storage, 0, sizeof(svo::Frame));
    f = reinterpret_cast<svo::Frame*>(storage);
    static int next_id = 0;            // Frame::frame_counter_ is defined in frame.cpp (not built)
    f->id_ = next_id++;
    f->timestamp_ = 0.0;
    f->cam_ = cam;
    f->T_f_w_ = to_se3(T_f_w);
    f->is_keyframe_ = false;
    new (&f->fts_) svo::Features();
    new (&f->key_pts_) std::vector<svo::Feature*>(5, nullptr);
    mats = ::aligned_alloc(32, (sizeof(cv::Mat) * n_levels + 31) / 32 * 32);
    cv::Mat* m = reinterpret_cast<cv::Mat*>(mats);
    for (int l = 0; l < n_levels; ++l)
      fill_mat_header(&m[l], const_cast<uint8_t*>(pyr[l]), height >> l, width >> l, (size_t)((width >> l) + row_pad), CV_8UC1);
    cv::Mat* rep[3] = {m, m + n_levels, m + n_levels};
    std::memcpy(static_cast<void*>(&f->img_pyr_), rep, sizeof(rep));
  }
  svo::Feature* add_feature(const double* px, const double* fv, int level, const double* pos /*or null*/) {
    svo::Feature* ftr = new svo::Feature(f, Eigen::Vector2d(px[0], px[1]), Eigen::Vector3d(fv[0], fv[1], fv[2]), level);
    if (pos) {
      svo::Point* pt = new svo::Point(Eigen::Vector3d(pos[0], pos[1], pos[2]), ftr);
      ftr->point = pt;
      points.push_back(pt);
    }
    f->fts_.push_back(ftr);
    return ftr;
  }
  svo::FramePtr ptr() { return svo::FramePtr(f, [](svo::Frame*) {}); }
  ~HandFrame() {
    for (svo::Feature* ftr : f->fts_) delete ftr;
    for (svo::Point* p : points) delete p;
    f->fts_.~list();
    f->key_pts_.~vector();
    std::free(mats);       // the cv::Mat headers own nothing
    std::free(storage);
  }
};


}  // namespace refh
