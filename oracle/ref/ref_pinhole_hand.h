// ref_pinhole_hand.h -- a vk::PinholeCamera laid out by hand (TEST INFRASTRUCTURE ONLY; ref_camera.cpp, ref_objects.cpp,
// dropin_run.cpp).  Its constructor calls the OpenCV library (cv::Mat members, initUndistortRectifyMap), which this image does
// not have; no stand-in is written: the object lives in zeroed storage with the vptr of the class's own vtable (emitted in the
// reference's pinhole_camera.o) and the members set as the constructor's initialiser list assigns them
// (pinhole_camera.cpp:20-30; the four cv::Mat members stay zeroed headers that nothing touches).
// Include after <svo/abstract_camera.h> and <svo/pinhole_camera.h> have been included with `protected` / `private` opened.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>

extern "C" char _ZTVN2vk13PinholeCameraE[];

namespace refh {
struct HandPinhole {
  void* storage;
  vk::AbstractCamera* cam;
  HandPinhole(int width, int height, double fx, double fy, double cx, double cy, const double* d) {
    storage = ::aligned_alloc(32, (sizeof(vk::PinholeCamera) + 31) / 32 * 32);
    std::memset(storage, 0, sizeof(vk::PinholeCamera));
    *reinterpret_cast<void**>(storage) = _ZTVN2vk13PinholeCameraE + 2 * sizeof(void*);
    vk::PinholeCamera* p = reinterpret_cast<vk::PinholeCamera*>(storage);
    p->width_ = width; p->height_ = height;                                    // AbstractCamera(width, height)
    const_cast<double&>(p->fx_) = fx; const_cast<double&>(p->fy_) = fy;        // fx_(fx), fy_(fy), cx_(cx), cy_(cy)
    const_cast<double&>(p->cx_) = cx; const_cast<double&>(p->cy_) = cy;
    p->distortion_ = std::fabs(d[0]) > 0.0000001;                              // distortion_(fabs(d0) > 0.0000001)
    for (int i = 0; i < 5; ++i) p->d_[i] = d[i];
    p->use_optimization_ = false;
    p->K_ << fx, 0.0, cx, 0.0, fy, cy, 0.0, 0.0, 1.0;
    cam = p;                                                                   // used through the abstract interface
  }
  ~HandPinhole() { std::free(storage); }
};
}  // namespace refh
