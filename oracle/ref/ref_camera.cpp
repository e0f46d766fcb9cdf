// ref_camera.cpp -- runs the REFERENCE's compiled vk::PinholeCamera member functions (TEST INFRASTRUCTURE ONLY,
// built into oracle/_ref/libsvo_ref.so by `make -C oracle ref`, only where /root/reference is mounted).
//
// pinhole_camera.cpp compiles unmodified against the vendored OpenCV headers.  Its constructor, the distorted branch
// of cam2world and undistortImage call out-of-line functions of the OpenCV core / calib3d / imgproc libraries
// (cv::Mat constructors, cv::initUndistortRectifyMap, cv::undistortPoints, cv::remap), which this image does not
// have; no stand-in is written: those symbols stay unresolved in the shared object (lazy binding) and are never
// reached.  What is pure arithmetic is reference code and runs here through the class's own vtable on an object laid
// out by hand in zeroed storage (members as the constructor's initialiser list assigns them, pinhole_camera.cpp:20-30;
// the four cv::Mat members stay zeroed headers that nothing touches):
//   PinholeCamera::world2cam(const Vector3d&)   pinhole_camera.cpp:73-77  (project2d + the 2-D overload)
//   PinholeCamera::world2cam(const Vector2d&)   :79-106  pinhole and 5-coefficient radtan forward model
//   PinholeCamera::cam2world(u, v)              :44-71   distortion-free branch only
//   AbstractCamera::isInFrame (both overloads)  I/abstract_camera.h:52-70
#include <cstdlib>
#include <cstring>

#include <svo/global.h>
// width_/height_ are protected, the intrinsics private const members; access specifiers do not change the layout gcc
// gives the classes
#define protected public
#define private public
#include <svo/abstract_camera.h>
#include <svo/pinhole_camera.h>
#undef private
#undef protected


#include "ref_pinhole_hand.h"
using refh::HandPinhole;

// for the translation units that see vk::PinholeCamera with its access specifiers closed (ref_objects.cpp): a hand-laid
// camera behind the abstract interface (ref_common.h declares the two)
namespace refh {
vk::AbstractCamera* make_hand_pinhole(int width, int height, double fx, double fy, double cx, double cy, const double* d5, void** handle) {
  HandPinhole* h = new HandPinhole(width, height, fx, fy, cx, cy, d5);
  *handle = h;
  return h->cam;
}
void free_hand_pinhole(void* handle) { delete static_cast<HandPinhole*>(handle); }
}  // namespace refh

extern "C" {

// world2cam of n camera-frame points (virtual call on vk::AbstractCamera*): px[n][2]
int ref_pinhole_world2cam(int width, int height, double fx, double fy, double cx, double cy, const double* d, int n,
                          const double* xyz, double* px) {
  HandPinhole h(width, height, fx, fy, cx, cy, d);
  for (int i = 0; i < n; ++i) {
    const Eigen::Vector2d r = h.cam->world2cam(Eigen::Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    px[2 * i] = r[0]; px[2 * i + 1] = r[1];
  }
  return 0;
}

// world2cam of n unit-plane points uv[n][2]
int ref_pinhole_world2cam_uv(int width, int height, double fx, double fy, double cx, double cy, const double* d, int n,
                             const double* uv, double* px) {
  HandPinhole h(width, height, fx, fy, cx, cy, d);
  for (int i = 0; i < n; ++i) {
    const Eigen::Vector2d r = h.cam->world2cam(Eigen::Vector2d(uv[2 * i], uv[2 * i + 1]));
    px[2 * i] = r[0]; px[2 * i + 1] = r[1];
  }
  return 0;
}

// cam2world of n pixels; only for d[0] == 0 (the distorted branch needs cv::undistortPoints): returns -1 otherwise
int ref_pinhole_cam2world(int width, int height, double fx, double fy, double cx, double cy, const double* d, int n,
                          const double* px, double* f) {
  if (std::fabs(d[0]) > 0.0000001) return -1;
  HandPinhole h(width, height, fx, fy, cx, cy, d);
  for (int i = 0; i < n; ++i) {
    const Eigen::Vector3d r = h.cam->cam2world(px[2 * i], px[2 * i + 1]);
    f[3 * i] = r[0]; f[3 * i + 1] = r[1]; f[3 * i + 2] = r[2];
  }
  return 0;
}

// isInFrame(obs, boundary) and isInFrame(obs, boundary, level) of n integer pixels
int ref_camera_is_in_frame(int width, int height, int n, const int* obs, int boundary, int level, uint8_t* plain,
                           uint8_t* levelled) {
  const double d[5] = {0, 0, 0, 0, 0};
  HandPinhole h(width, height, 100.0, 100.0, width / 2.0, height / 2.0, d);
  for (int i = 0; i < n; ++i) {
    const Eigen::Vector2i o(obs[2 * i], obs[2 * i + 1]);
    plain[i] = h.cam->isInFrame(o, boundary) ? 1 : 0;
    levelled[i] = h.cam->isInFrame(o, boundary, level) ? 1 : 0;
  }
  return 0;
}

}  // extern "C"
