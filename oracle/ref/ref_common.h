// ref_common.h -- helpers shared by the reference harness translation units
// (TEST INFRASTRUCTURE ONLY; see ref_harness.cpp for what is and is not compiled).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include <svo/global.h>
#include <svo/SE3.h>
#include <svo/abstract_camera.h>
#include <svo/math_utils.h>

namespace refh {

struct MatView {
  alignas(cv::Mat) unsigned char storage[sizeof(cv::Mat)];
  cv::Mat& set(const uint8_t* data, int rows, int cols, int stride) {
    std::memset(storage, 0, sizeof(storage));
    cv::Mat& m = *reinterpret_cast<cv::Mat*>(storage);
    m.flags = cv::Mat::MAGIC_VAL | CV_8UC1 | (stride == cols ? cv::Mat::CONTINUOUS_FLAG : 0);
    m.dims = 2;
    m.rows = rows;
    m.cols = cols;
    m.data = const_cast<uint8_t*>(data);
    m.datastart = m.data;
    m.dataend = m.datalimit = m.data + (size_t)rows * stride;
    m.size.p = &m.rows;
    m.step.p = m.step.buf;
    m.step.buf[0] = (size_t)stride;
    m.step.buf[1] = 1;
    return m;
  }
};

// Fill the public fields of a cv::Mat that lives in zeroed raw storage (any depth); see MatView.
inline void fill_mat_header(void* storage, void* data, int rows, int cols, size_t row_bytes, int type) {
  std::memset(storage, 0, sizeof(cv::Mat));
  cv::Mat& m = *reinterpret_cast<cv::Mat*>(storage);
  const size_t esz = CV_ELEM_SIZE(type);
  m.flags = cv::Mat::MAGIC_VAL | type | (row_bytes == (size_t)cols * esz ? cv::Mat::CONTINUOUS_FLAG : 0);
  m.dims = 2;
  m.rows = rows;
  m.cols = cols;
  m.data = static_cast<uint8_t*>(data);
  m.datastart = m.data;
  m.dataend = m.datalimit = m.data + (size_t)rows * row_bytes;
  m.size.p = &m.rows;
  m.step.p = m.step.buf;
  m.step.buf[0] = row_bytes;
  m.step.buf[1] = esz;
}

// Distortion-free pinhole behind the reference's vk::AbstractCamera interface
// (the reference's PinholeCamera cannot be linked: its ctor needs OpenCV).
class HarnessPinhole : public vk::AbstractCamera {
 public:
  double fx_, fy_, cx_, cy_;
  HarnessPinhole(int w, int h, double fx, double fy, double cx, double cy)
      : vk::AbstractCamera(w, h), fx_(fx), fy_(fy), cx_(cx), cy_(cy) {}
  Eigen::Vector3d cam2world(const double& u, const double& v) const override {
    Eigen::Vector3d xyz((u - cx_) / fx_, (v - cy_) / fy_, 1.0);
    return xyz.normalized();
  }
  Eigen::Vector3d cam2world(const Eigen::Vector2d& px) const override { return cam2world(px[0], px[1]); }
  Eigen::Vector2d world2cam(const Eigen::Vector3d& xyz) const override { return world2cam(vk::project2d(xyz)); }
  Eigen::Vector2d world2cam(const Eigen::Vector2d& uv) const override {
    return Eigen::Vector2d(fx_ * uv[0] + cx_, fy_ * uv[1] + cy_);
  }
  double errorMultiplier2() const override { return std::fabs(fx_); }
  double errorMultiplier() const override { return std::fabs(4.0 * fx_ * fy_); }
};

// a vk::PinholeCamera (with its radtan coefficients) laid out by hand, behind the abstract interface (ref_camera.cpp)
vk::AbstractCamera* make_hand_pinhole(int width, int height, double fx, double fy, double cx, double cy, const double* d5, void** handle);
void free_hand_pinhole(void* handle);

inline SE3 to_se3(const double* T) { return SE3(T[0], T[1], T[2], T[3], T[4], T[5], T[6]); }
inline void from_se3(const SE3& s, double* T) {
  T[0] = s.get_translation().x; T[1] = s.get_translation().y; T[2] = s.get_translation().z;
  T[3] = s.get_rotation().x; T[4] = s.get_rotation().y; T[5] = s.get_rotation().z;
  T[6] = s.get_rotation().w;
}

}  // namespace refh
