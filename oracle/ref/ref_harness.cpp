// ref_harness.cpp -- extern "C" entry points into the REFERENCE's own code
// (TEST INFRASTRUCTURE ONLY; built into oracle/_ref/libsvo_ref.so by oracle/Makefile,
// only in the container that has /root/reference; never shipped, never copied).
//
// What is compiled: the reference translation units feature_alignment.cpp,
// matcher.cpp, vision.cpp, point.cpp, config.cpp, robust_cost.cpp, math_utils.cpp,
// sparse_img_align.cpp, unmodified, from /root/reference, plus the reference's
// header-only code (SE3.h, SO3.h, frame.h, nlls_solver*.h, patch_score.h, vendored
// Eigen 3.4.0).  This file uses the free functions and header templates; the member
// functions that need real Frame / SparseImgAlign objects are driven by ref_objects.cpp.
//
// What is NOT compiled (unbuildable here without stand-ins): frame.cpp (OpenCV core
// library), depth_filter.cpp (needs <android/log.h>), pinhole_camera.cpp
// (cv::initUndistortRectifyMap).  No stub library and no fake header is written.
//
// cv::Mat: align2D/align1D/warpAffine/interpolateMat_8u/halfSample read an image
// only through the public fields data/rows/cols/step.  No cv::Mat can be
// constructed without the OpenCV library, so MatView below fills those public
// fields of a raw-storage cv::Mat (the real vendored header layout); no OpenCV
// function is called and none is provided.
#include <cstring>
#include <cstdint>
#include <cmath>

#include <svo/global.h>
#include <svo/SE3.h>
#include <svo/frame.h>
#include <svo/feature.h>
#include <svo/feature_alignment.h>
#include <svo/matcher.h>
#include <svo/patch_score.h>
#include <svo/vision.h>
#include <svo/nlls_solver.h>
#include <svo/abstract_camera.h>
#include <svo/math_utils.h>

#include "../svo_oracle.h"
#include "ref_common.h"

// defined (non-static) in matcher.cpp:123-136 but not declared in matcher.h
namespace svo {
bool depthFromTriangulation(const SE3& T_search_ref, const Vector3d& f_ref, const Vector3d& f_cur,
                            double& depth);
}

using namespace refh;

namespace {

// The reference's Gauss-Newton driver (nlls_solver_impl.hpp) with the residual
// body supplied by the C restatement; solve()/update() are the two statements of
// sparse_img_align.cpp:291-308 on the real Eigen LDLT and the real SE3::exp.
class DrivenAlign : public vk::NLLSSolver<6, SE3> {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  void* orc = nullptr;
  int evals = 0;
  DrivenAlign(int n_iter, double eps) {
    n_iter_ = n_iter;
    n_iter_init_ = n_iter_;
    method_ = GaussNewton;
    verbose_ = false;
    eps_ = eps;
  }
  double chi2() const { return chi2_; }
  const Eigen::Matrix<double, 6, 6>& H() const { return H_; }

 protected:
  double computeResiduals(const SE3& model, bool linearize, bool) override {
    double T[7], H[36], J[6];
    size_t nm = 0;
    from_se3(model, T);
    double r = svo_orc_sia_eval(orc, T, linearize ? 1 : 0, H, J, &nm);
    ++evals;
    n_meas_ = nm;
    if (linearize) {
      for (int a = 0; a < 6; ++a) {
        for (int b = 0; b < 6; ++b) H_(a, b) = H[a * 6 + b];
        Jres_[a] = J[a];
      }
    }
    return r;
  }
  int solve() override {
    x_ = H_.ldlt().solve(Jres_);
    if ((bool)std::isnan((double)x_[0])) return 0;
    return 1;
  }
  void update(const ModelType& old_model, ModelType& new_model) override {
    Eigen::Matrix<double, 6, 1> x = -x_;
    new_model = old_model * SE3::exp(x.data());
  }
};

}  // namespace

extern "C" {

// ---- SE3 / SO3 (SE3.h, SO3.h) --------------------------------------------
void ref_se3_mul(const double* A, const double* B, double* out) { from_se3(to_se3(A) * to_se3(B), out); }
void ref_se3_inverse(const double* A, double* out) { from_se3(to_se3(A).inverse(), out); }
void ref_se3_act(const double* A, const double* p, double* out) {
  Eigen::Vector3d r = to_se3(A) * Eigen::Vector3d(p[0], p[1], p[2]);
  out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}
void ref_se3_exp(const double* l, double* out) { from_se3(SE3::exp(l), out); }
void ref_so3_log(const double* q, double* out) {
  Point3d r = SO3(q[0], q[1], q[2], q[3]).log();
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_se3_rotation_matrix(const double* A, double* R) {
  Eigen::Matrix3d m = to_se3(A).rotation_matrix();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = m(i, j);
}

// ---- Frame::jacobian_xyz2uv (frame.h:110-132) ----------------------------
void ref_jacobian_xyz2uv(const double* xyz, double* J) {
  Eigen::Matrix<double, 2, 6> m;
  svo::Frame::jacobian_xyz2uv(Eigen::Vector3d(xyz[0], xyz[1], xyz[2]), m);
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 6; ++j) J[6 * i + j] = m(i, j);
}

// ---- Eigen LDLT solve as used by SparseImgAlign::solve --------------------
void ref_ldlt6_solve(const double* H, const double* b, double* x) {
  Eigen::Matrix<double, 6, 6> m;
  Eigen::Matrix<double, 6, 1> v;
  for (int i = 0; i < 6; ++i) { v[i] = b[i]; for (int j = 0; j < 6; ++j) m(i, j) = H[6 * i + j]; }
  Eigen::Matrix<double, 6, 1> r = m.ldlt().solve(v);
  for (int i = 0; i < 6; ++i) x[i] = r[i];
}

// ---- reference GN driver over the restated residual body ------------------
int ref_driven_sparse_align(const svo_orc_camera* cam, const uint8_t* const* ref_pyr,
                            const uint8_t* const* cur_pyr, int n, const double* px, const double* f,
                            const double* pos, const uint8_t* has_point, const double* T_ref_w,
                            const double* T_cur_w_init, int max_level, int min_level, int n_iter,
                            double eps, double* T_cur_w_out, size_t* n_tracked, double* chi2,
                            int* iters /*[8]*/, double* H_out /*[36]*/) {
  DrivenAlign* a = new DrivenAlign(n_iter, eps);
  a->reset();
  a->orc = svo_orc_sia_open(cam, ref_pyr, cur_pyr, n, px, f, pos, has_point, T_ref_w);
  SE3 T_ref = to_se3(T_ref_w), T_cur = to_se3(T_cur_w_init);
  SE3 T_cur_from_ref(T_cur * T_ref.inverse());
  for (int level = max_level; level >= min_level; --level) {
    a->mu_ = 0.1;
    svo_orc_sia_set_level(a->orc, level);
    a->evals = 0;
    a->optimize(T_cur_from_ref);
    if (iters && level < 8) iters[level] = a->evals;
  }
  from_se3(T_cur_from_ref * T_ref, T_cur_w_out);
  *n_tracked = a->n_meas_ / 16;
  *chi2 = a->chi2();
  if (H_out) for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) H_out[6 * i + j] = a->H()(i, j);
  svo_orc_sia_close(a->orc);
  delete a;
  return 0;
}

// ---- feature_alignment.cpp -------------------------------------------------
int ref_align2d(const uint8_t* img, int cols, int rows, int stride, const uint8_t* pwb,
                const uint8_t* patch, int n_iter, double* px) {
  MatView mv;
  alignas(16) uint8_t b[100];
  alignas(16) uint8_t p[64];
  std::memcpy(b, pwb, 100);
  std::memcpy(p, patch, 64);
  Eigen::Vector2d e(px[0], px[1]);
  bool ok = svo::feature_alignment::align2D(mv.set(img, rows, cols, stride), b, p, n_iter, e, false);
  px[0] = e[0]; px[1] = e[1];
  return ok ? 1 : 0;
}

int ref_align1d(const uint8_t* img, int cols, int rows, int stride, const float* dir,
                const uint8_t* pwb, const uint8_t* patch, int n_iter, double* px, double* h_inv) {
  MatView mv;
  alignas(16) uint8_t b[100];
  alignas(16) uint8_t p[64];
  std::memcpy(b, pwb, 100);
  std::memcpy(p, patch, 64);
  Eigen::Vector2d e(px[0], px[1]);
  bool ok = svo::feature_alignment::align1D(mv.set(img, rows, cols, stride), Eigen::Vector2f(dir[0], dir[1]),
                                            b, p, n_iter, e, *h_inv);
  px[0] = e[0]; px[1] = e[1];
  return ok ? 1 : 0;
}

// ---- patch_score.h ZMSSD<4> (SSE2 path on this host) -------------------------
int ref_zmssd(const uint8_t* ref_patch, const uint8_t* cur, int stride) {
  alignas(16) uint8_t p[64];
  std::memcpy(p, ref_patch, 64);
  vk::patch_score::ZMSSD<4> s(p);
  return s.computeScore(const_cast<uint8_t*>(cur), stride);
}

// ---- matcher.cpp warp:: / triangulation ---------------------------------------
void ref_get_warp_matrix_affine(int w, int h, double fx, double fy, double cx, double cy,
                                const double* px_ref, const double* f_ref, double depth_ref,
                                const double* T_cur_ref, int level_ref, double* A /*row-major*/) {
  HarnessPinhole cam(w, h, fx, fy, cx, cy);
  Eigen::Matrix2d M;
  svo::warp::getWarpMatrixAffine(cam, cam, Eigen::Vector2d(px_ref[0], px_ref[1]),
                                 Eigen::Vector3d(f_ref[0], f_ref[1], f_ref[2]), depth_ref,
                                 to_se3(T_cur_ref), level_ref, M);
  A[0] = M(0, 0); A[1] = M(0, 1); A[2] = M(1, 0); A[3] = M(1, 1);
}

int ref_get_best_search_level(const double* A, int max_level) {
  Eigen::Matrix2d M;
  M << A[0], A[1], A[2], A[3];
  return svo::warp::getBestSearchLevel(M, max_level);
}

void ref_warp_affine(const double* A, const uint8_t* img, int cols, int rows, const double* px_ref,
                     int level_ref, int search_level, int halfpatch_size, uint8_t* patch) {
  MatView mv;
  Eigen::Matrix2d M;
  M << A[0], A[1], A[2], A[3];
  svo::warp::warpAffine(M, mv.set(img, rows, cols, cols), Eigen::Vector2d(px_ref[0], px_ref[1]),
                        level_ref, search_level, halfpatch_size, patch);
}

void ref_patch_from_border(const uint8_t* pwb, uint8_t* patch) {
  svo::Matcher* m = new svo::Matcher();
  std::memcpy(m->patch_with_border_, pwb, 100);
  m->createPatchFromPatchWithBorder();
  std::memcpy(patch, m->patch_, 64);
  delete m;
}

int ref_depth_from_triangulation(const double* T_search_ref, const double* f_ref, const double* f_cur,
                                 double* depth) {
  return svo::depthFromTriangulation(to_se3(T_search_ref), Eigen::Vector3d(f_ref[0], f_ref[1], f_ref[2]),
                                     Eigen::Vector3d(f_cur[0], f_cur[1], f_cur[2]), *depth) ? 1 : 0;
}

void ref_cam2world(int w, int h, double fx, double fy, double cx, double cy, double u, double v, double* f) {
  HarnessPinhole cam(w, h, fx, fy, cx, cy);
  Eigen::Vector3d r = cam.cam2world(u, v);
  f[0] = r[0]; f[1] = r[1]; f[2] = r[2];
}

// ---- vision.h / vision.cpp ------------------------------------------------------
float ref_interpolate_8u(const uint8_t* img, int cols, int rows, float u, float v) {
  MatView mv;
  return vk::interpolateMat_8u(mv.set(img, rows, cols, cols), u, v);
}

// `in` 16-byte aligned and cols%16==0 -> the SSE2 path; otherwise the scalar path.
void ref_half_sample(const uint8_t* in, int cols, int rows, uint8_t* out) {
  MatView a, b;
  vk::halfSample(a.set(in, rows, cols, cols), b.set(out, rows / 2, cols / 2, cols / 2));
}

// vk::shiTomasiScore (vision.cpp:113-154)
float ref_shi_tomasi_score(const uint8_t* img, int cols, int rows, int u, int v) {
  MatView mv;
  return vk::shiTomasiScore(mv.set(img, rows, cols, cols), u, v);
}

}  // extern "C"
