// svo_device_math.h -- SE3/SO3, pinhole camera and 6x6 LDLT for gfx950 device code.
//
// Semantics follow the reference statement by statement (quirks included) so that
// fp64 index-critical chains (projection -> floor/round -> pixel index) round the
// same way as the CPU path; the library is built with -ffp-contract=off.
//   SE3/SO3:  I/SE3.h:35-61,153-182, I/SO3.h:468-488,523-526
//   camera:   S/pinhole_camera.cpp:44-106, I/abstract_camera.h:52-70
//   LDLT:     Eigen 3.4.0 Cholesky/LDLT.h:297-403,574-613 (as used by
//             SparseImgAlign::solve, S/sparse_img_align.cpp:291-297)
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/svo_hip.h"

#define SVO_DEV __device__ __forceinline__

namespace svo_dev {

struct Cam {
  double fx, fy, cx, cy;
  double d[5];
  int distortion;
  int width, height;
};

// wave-uniform copy of lane l's value
SVO_DEV double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

SVO_DEV void cross3(const double* a, const double* b, double* o) {
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}

// q = {x,y,z,w}.  p + w*uv + q x uv with uv = 2 (q x p)   (I/SO3.h:478-483)
SVO_DEV void so3_rotate(const double* q, const double* p, double* o) {
  double uv[3], quv[3];
  cross3(q, p, uv);
  uv[0] = uv[0] + uv[0]; uv[1] = uv[1] + uv[1]; uv[2] = uv[2] + uv[2];
  cross3(q, uv, quv);
  double x = (p[0] + q[3] * uv[0]) + quv[0];
  double y = (p[1] + q[3] * uv[1]) + quv[1];
  double z = (p[2] + q[3] * uv[2]) + quv[2];
  o[0] = x; o[1] = y; o[2] = z;
}

SVO_DEV void so3_mul(const double* a, const double* b, double* o) {
  double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

// T = {tx,ty,tz,qx,qy,qz,qw}
SVO_DEV void se3_mul(const double* A, const double* B, double* out) {
  double q[4], rt[3];
  so3_mul(A + 3, B + 3, q);
  so3_rotate(A + 3, B, rt);
  double t0 = A[0] + rt[0], t1 = A[1] + rt[1], t2 = A[2] + rt[2];
  out[0] = t0; out[1] = t1; out[2] = t2;
  out[3] = q[0]; out[4] = q[1]; out[5] = q[2]; out[6] = q[3];
}

SVO_DEV void se3_inverse(const double* T, double* out) {
  double qi[4] = {-T[3], -T[4], -T[5], T[6]};
  double rt[3];
  so3_rotate(qi, T, rt);
  out[0] = -rt[0]; out[1] = -rt[1]; out[2] = -rt[2];
  out[3] = qi[0]; out[4] = qi[1]; out[5] = qi[2]; out[6] = qi[3];
}

SVO_DEV void se3_act(const double* T, const double* p, double* out) {
  double r[3];
  so3_rotate(T + 3, p, r);
  out[0] = T[0] + r[0]; out[1] = T[1] + r[1]; out[2] = T[2] + r[2];
}

// sin and cos of a small angle by their Taylor series in Horner form (|x| <= 0.5: truncation error < 2e-23,
// rounding ~1 ulp, i.e. the same accuracy class as a libm call at a tenth of the instructions); the
// Gauss-Newton update angles are tiny, larger arguments take the library path.
SVO_DEV void sincos_small(double x, double* s, double* c) {
  if (fabs(x) <= 0.5) {
    const double z = x * x;
    // sin x = x (1 - z/6 (1 - z/20 (1 - z/42 (1 - z/72 (1 - z/110 (1 - z/156 (1 - z/210 (1 - z/272))))))))
    double ps = 1.0 - z * (1.0 / 272.0);
    ps = 1.0 - z * (1.0 / 210.0) * ps;
    ps = 1.0 - z * (1.0 / 156.0) * ps;
    ps = 1.0 - z * (1.0 / 110.0) * ps;
    ps = 1.0 - z * (1.0 / 72.0) * ps;
    ps = 1.0 - z * (1.0 / 42.0) * ps;
    ps = 1.0 - z * (1.0 / 20.0) * ps;
    ps = 1.0 - z * (1.0 / 6.0) * ps;
    *s = x * ps;
    // cos x = 1 - z/2 (1 - z/12 (1 - z/30 (1 - z/56 (1 - z/90 (1 - z/132 (1 - z/182 (1 - z/240)))))))
    double pc = 1.0 - z * (1.0 / 240.0);
    pc = 1.0 - z * (1.0 / 182.0) * pc;
    pc = 1.0 - z * (1.0 / 132.0) * pc;
    pc = 1.0 - z * (1.0 / 90.0) * pc;
    pc = 1.0 - z * (1.0 / 56.0) * pc;
    pc = 1.0 - z * (1.0 / 30.0) * pc;
    pc = 1.0 - z * (1.0 / 12.0) * pc;
    *c = 1.0 - z * 0.5 * pc;
  } else {
    *s = sin(x);
    *c = cos(x);
  }
}

// I/SE3.h:153-182.  theta == 0 gives a NaN translation, as in the reference.
SVO_DEV void se3_exp(const double* l, double* out) {
  const double p[3] = {l[0], l[1], l[2]};
  const double r[3] = {l[3], l[4], l[5]};
  double theta_sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  double theta = sqrt(theta_sq);
  double half_theta = 0.5 * theta;
  double imag_factor, real_factor;
  double sin_h, cos_h, sin_t, cos_t;
  sincos_small(half_theta, &sin_h, &cos_h);
  sincos_small(theta, &sin_t, &cos_t);
  if (theta < 1e-10) {
    double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real_factor = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    imag_factor = sin_h / theta;
    real_factor = cos_h;
  }
  double rxp[3], rxrxp[3];
  cross3(r, p, rxp);
  cross3(r, rxp, rxrxp);
  double c1 = (1 - cos_t) / theta_sq;
  double c2 = (theta - sin_t) / (theta_sq * theta);
  out[0] = (p[0] + c1 * rxp[0]) + c2 * rxrxp[0];
  out[1] = (p[1] + c1 * rxp[1]) + c2 * rxrxp[1];
  out[2] = (p[2] + c1 * rxp[2]) + c2 * rxrxp[2];
  out[3] = imag_factor * r[0];
  out[4] = imag_factor * r[1];
  out[5] = imag_factor * r[2];
  out[6] = real_factor;
}

// se3_exp for the Gauss-Newton updates of the fused SparseImgAlign kernel (every other wave of the workgroup is
// waiting for it): the same four series as sincos_small, but the quotients of I/SE3.h:153-182 are taken from the
// series themselves -- sin(h)/theta = ps_h / 2, (1 - cos t)/t^2 = pc_t / 2, (t - sin t)/t^3 = (inner sine
// series)/6 -- so there is no square root and no division on the critical path, and none of the cancellation the
// quotient forms have at small angles.  theta == 0 keeps the reference's NaN translation.
// The four series are 8 dependent steps p <- 1 - (z c_k) p each; on one lane they cost ~1.2 k cycles of latency, so
// they run on four lanes at once: lane l & 3 takes column l & 3 of the coefficient table (a leading zero pads the
// 7-step series: 1 - (z 0) 1 = 1) and the caller collects the four values with readlane.
//   column 0: qs_t (inner sine series of theta^2)   column 1: pc_t (cosine series of theta^2)
//   column 2: ps_h (sine series of (theta/2)^2)     column 3: pc_h (cosine series of (theta/2)^2)
SVO_DEV void se3_exp_series_table(double* tab /* [8][4] */) {
  const double sn[8] = {1.0 / 272.0, 1.0 / 210.0, 1.0 / 156.0, 1.0 / 110.0, 1.0 / 72.0, 1.0 / 42.0, 1.0 / 20.0, 1.0 / 6.0};
  const double cs[7] = {1.0 / 240.0, 1.0 / 182.0, 1.0 / 132.0, 1.0 / 90.0, 1.0 / 56.0, 1.0 / 30.0, 1.0 / 12.0};
  for (int k = 0; k < 8; ++k) {
    tab[k * 4 + 0] = k == 0 ? 0.0 : sn[k - 1];
    tab[k * 4 + 1] = k == 0 ? 0.0 : cs[k - 1];
    tab[k * 4 + 2] = sn[k];
    tab[k * 4 + 3] = k == 0 ? 0.0 : cs[k - 1];
  }
}
// one lane's series: c[0..7] = its column, z = theta^2 (columns 0, 1) or theta^2 / 4 (columns 2, 3)
SVO_DEV double se3_exp_series_lane(const double* c, double z) {
  double p = 1.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) p = 1.0 - z * c[k] * p;
  return p;
}
// the rest of the exponential, given the four series values (small angles only: zt <= 0.25)
SVO_DEV void se3_exp_small_finish(const double* l, double zt, double qs_t, double pc_t, double ps_h, double pc_h, double* out) {
  const double p[3] = {l[0], l[1], l[2]};
  const double r[3] = {l[3], l[4], l[5]};
  const double zh = 0.25 * zt;                                    // (theta/2)^2
  const double imag_factor = 0.5 * ps_h;                          // sin(theta/2) / theta
  const double real_factor = 1.0 - zh * 0.5 * pc_h;               // cos(theta/2)
  double c1 = 0.5 * pc_t;                                         // (1 - cos theta) / theta^2
  double c2 = (1.0 / 6.0) * qs_t;                                 // (theta - sin theta) / theta^3
  if (zt == 0.0) { c1 = __longlong_as_double(0x7ff8000000000000LL); c2 = c1; }   // 0/0 of the reference's quotients
  double rxp[3], rxrxp[3];
  cross3(r, p, rxp);
  cross3(r, rxp, rxrxp);
  out[0] = (p[0] + c1 * rxp[0]) + c2 * rxrxp[0];
  out[1] = (p[1] + c1 * rxp[1]) + c2 * rxrxp[1];
  out[2] = (p[2] + c1 * rxp[2]) + c2 * rxrxp[2];
  out[3] = imag_factor * r[0];
  out[4] = imag_factor * r[1];
  out[5] = imag_factor * r[2];
  out[6] = real_factor;
}

// row-major 3x3 (I/SO3.h:391-406)
SVO_DEV void se3_rotation_matrix(const double* T, double* m) {
  double x = T[3], y = T[4], z = T[5], w = T[6];
  double x2 = x * x, y2 = y * y, z2 = z * z;
  double xy = x * y, xz = x * z, yz = y * z;
  double wx = w * x, wy = w * y, wz = w * z;
  m[0] = 1.0 - 2.0 * (y2 + z2); m[1] = 2.0 * (xy - wz);       m[2] = 2.0 * (xz + wy);
  m[3] = 2.0 * (xy + wz);       m[4] = 1.0 - 2.0 * (x2 + z2); m[5] = 2.0 * (yz - wx);
  m[6] = 2.0 * (xz - wy);       m[7] = 2.0 * (yz + wx);       m[8] = 1.0 - 2.0 * (x2 + y2);
}

// unit-plane (u,v) -> pixel, with the radtan model when enabled (S/pinhole_camera.cpp:79-106)
SVO_DEV void world2cam_uv(const Cam& c, double u, double v, double* px) {
  if (!c.distortion) {
    px[0] = c.fx * u + c.cx;
    px[1] = c.fy * v + c.cy;
  } else {
    double x = u, y = v;
    double r2 = x * x + y * y;
    double r4 = r2 * r2;
    double r6 = r4 * r2;
    double a1 = 2 * x * y;
    double a2 = r2 + 2 * x * x;
    double a3 = r2 + 2 * y * y;
    double cdist = 1 + c.d[0] * r2 + c.d[1] * r4 + c.d[4] * r6;
    double xd = x * cdist + c.d[2] * a1 + c.d[3] * a2;
    double yd = y * cdist + c.d[2] * a3 + c.d[3] * a1;
    px[0] = xd * c.fx + c.cx;
    px[1] = yd * c.fy + c.cy;
  }
}

SVO_DEV void world2cam(const Cam& c, const double* xyz, double* px) {
  world2cam_uv(c, xyz[0] / xyz[2], xyz[1] / xyz[2], px);
}

// PinholeCamera::cam2world (S/pinhole_camera.cpp:44-71).  Distorted cameras: cv::undistortPoints on float points with
// float K / D (:54-63, :31-32) -- OpenCV 4.5.4 calib3d, third-party and absent from the reference tree (parity
// unpinned): five fixed-point iterations of the radial-tangential model in double, result stored as float.
SVO_DEV void cam2world(const Cam& c, double u, double v, double* f) {
  double x, y;
  if (!c.distortion) {
    x = (u - c.cx) / c.fx;
    y = (v - c.cy) / c.fy;
  } else {
    const double fx = (double)(float)c.fx, fy = (double)(float)c.fy, cx = (double)(float)c.cx, cy = (double)(float)c.cy;
    const double k1 = (double)(float)c.d[0], k2 = (double)(float)c.d[1], p1 = (double)(float)c.d[2],
                 p2 = (double)(float)c.d[3], k3 = (double)(float)c.d[4];
    const double uf = (double)(float)u, vf = (double)(float)v;
    const double ifx = 1. / fx, ify = 1. / fy;
    x = (uf - cx) * ifx;
    y = (vf - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
      if (icdist < 0) { x = (uf - cx) * ifx; y = (vf - cy) * ify; break; }
      const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + 0.0 * r2 + 0.0 * r2 * r2;
      const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + 0.0 * r2 + 0.0 * r2 * r2;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    x = (double)(float)x;
    y = (double)(float)y;
  }
  double z = 1.0;
  double n2 = x * x + y * y + z * z;
  if (n2 > 0.0) {
    double n = sqrt(n2);
    f[0] = x / n; f[1] = y / n; f[2] = z / n;
  } else {
    f[0] = x; f[1] = y; f[2] = z;
  }
}

SVO_DEV bool is_in_frame_level(const Cam& c, int ox, int oy, int boundary, int level) {
  return ox >= boundary && ox < c.width / (1 << level) - boundary && oy >= boundary &&
         oy < c.height / (1 << level) - boundary;
}

// S/matcher.cpp:36-60 (A row-major)
SVO_DEV void get_warp_matrix_affine(const Cam& cam, const double* px_ref, const double* f_ref, double depth_ref,
                                    const double* T_cur_ref, int level_ref, double* A) {
  const int halfpatch_size = 5;
  const double xyz_ref[3] = {f_ref[0] * depth_ref, f_ref[1] * depth_ref, f_ref[2] * depth_ref};
  double du[3], dv[3];
  const double off = (double)halfpatch_size * (1 << level_ref);
  cam2world(cam, px_ref[0] + off, px_ref[1] + 0.0 * (1 << level_ref), du);
  cam2world(cam, px_ref[0] + 0.0 * (1 << level_ref), px_ref[1] + off, dv);
  const double su = xyz_ref[2] / du[2];
  du[0] *= su; du[1] *= su; du[2] *= su;
  const double sv = xyz_ref[2] / dv[2];
  dv[0] *= sv; dv[1] *= sv; dv[2] *= sv;
  double p[3], px_cur[2], px_du[2], px_dv[2];
  se3_act(T_cur_ref, xyz_ref, p); world2cam(cam, p, px_cur);
  se3_act(T_cur_ref, du, p);      world2cam(cam, p, px_du);
  se3_act(T_cur_ref, dv, p);      world2cam(cam, p, px_dv);
  A[0] = (px_du[0] - px_cur[0]) / halfpatch_size;
  A[2] = (px_du[1] - px_cur[1]) / halfpatch_size;
  A[1] = (px_dv[0] - px_cur[0]) / halfpatch_size;
  A[3] = (px_dv[1] - px_cur[1]) / halfpatch_size;
}


// I/frame.h:110-132, 2x6 row-major
SVO_DEV void jacobian_xyz2uv(const double* p, double* J) {
  const double x = p[0], y = p[1];
  const double z_inv = 1. / p[2];
  const double z_inv_2 = z_inv * z_inv;
  J[0] = -z_inv;
  J[1] = 0.0;
  J[2] = x * z_inv_2;
  J[3] = y * J[2];
  J[4] = -(1.0 + x * J[2]);
  J[5] = y * z_inv;
  J[6] = 0.0;
  J[7] = -z_inv;
  J[8] = y * z_inv_2;
  J[9] = 1.0 + y * J[8];
  J[10] = -J[3];
  J[11] = -x * z_inv;
}

// Sums as Eigen 3.4 evaluates them inside its unrolled fixed-size triangular solves (Core/SolveTriangular.h,
// Core/Redux.h): redux_tree = binary splitting [0, n/2) + [n/2, n) of the scalar unroller; redux_packet2 = the
// SSE2 form used when both operands are contiguous (two lanes summed packet-wise, lanes added, then the remainder).
// ---- SparseImgAlign: a patch's share of the normal equations (svo_sia.hip, svo_nlls.hip)
// The 21 upper-triangle entries (row-major) of  sxx AA^T + sxy (AB^T + BA^T) + syy BB^T.
SVO_DEV void patch_hessian(const double* A, const double* B, double sxx, double sxy, double syy, double* hp) {
  int e = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) {
      hp[e] = sxx * (A[i] * A[j]) + sxy * (A[i] * B[j] + B[i] * A[j]) + syy * (B[i] * B[j]);
      ++e;
    }
}

// A = row 0, B = row 1 of Frame::jacobian_xyz2uv (I/frame.h:110-132) times fx/2^L, from {x,y,z,1/z}
SVO_DEV void patch_jacobian_rows(double x, double y, double z_inv, double jscale, double* A, double* B) {
  const double z_inv_2 = z_inv * z_inv;
  const double j02 = x * z_inv_2;
  const double j03 = y * j02;
  const double j12 = y * z_inv_2;
  A[0] = -z_inv * jscale;
  A[1] = 0.0 * jscale;
  A[2] = j02 * jscale;
  A[3] = j03 * jscale;
  A[4] = -(1.0 + x * j02) * jscale;
  A[5] = (y * z_inv) * jscale;
  B[0] = 0.0 * jscale;
  B[1] = -z_inv * jscale;
  B[2] = j12 * jscale;
  B[3] = (1.0 + y * j12) * jscale;
  B[4] = -j03 * jscale;
  B[5] = (-x * z_inv) * jscale;
}

template <int N>
SVO_DEV double redux_tree(const double* a) {
  if constexpr (N == 1) return a[0];
  else return redux_tree<N / 2>(a) + redux_tree<N - N / 2>(a + N / 2);
}
template <int N>
SVO_DEV double redux_packet2(const double* a) {
  if constexpr (N < 2) return a[0];
  else if constexpr (N == 2) return a[0] + a[1];
  else if constexpr (N == 3) return (a[0] + a[1]) + a[2];
  else if constexpr (N == 4) return (a[0] + a[2]) + (a[1] + a[3]);
  else return ((a[0] + a[2]) + (a[1] + a[3])) + a[4];
}
// d[I] -= sum_{j<I} m[I][j] d[j]  /  d[I] -= sum_{j>I} m[j][I] d[j], statically indexed
template <int N, int I>
SVO_DEV void ldlt_forward_row(const double (*m)[N], double* d) {
  if constexpr (I < N) {
    double t[I];
#pragma unroll
    for (int j = 0; j < I; ++j) t[j] = m[I][j] * d[j];
    d[I] -= redux_tree<I>(t);
    ldlt_forward_row<N, I + 1>(m, d);
  }
}
template <int N, int I>
SVO_DEV void ldlt_backward_row(const double (*m)[N], double* d) {
  if constexpr (I >= 0) {
    double t[N - 1 - I];
#pragma unroll
    for (int j = I + 1; j < N; ++j) t[j - I - 1] = m[j][I] * d[j];
    d[I] -= redux_packet2<N - 1 - I>(t);
    ldlt_backward_row<N, I - 1>(m, d);
  }
}

// Pivoted LDL^T of a symmetric NxN (lower part read) and solve, one thread (Eigen 3.4 Cholesky/LDLT.h).
template <int N>
SVO_DEV void ldlt_solve(const double* Hin, const double* b, double* x) {
  double m[N][N];
  int tr[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) m[i][j] = Hin[i * N + j];
  bool all_zero = false;
  for (int k = 0; k < N && !all_zero; ++k) {
    int big = k;
    double best = fabs(m[k][k]);
    for (int i = k + 1; i < N; ++i)
      if (fabs(m[i][i]) > best) { best = fabs(m[i][i]); big = i; }
    tr[k] = big;
    if (k != big) {
      for (int j = 0; j < k; ++j) { double t = m[k][j]; m[k][j] = m[big][j]; m[big][j] = t; }
      for (int i = big + 1; i < N; ++i) { double t = m[i][k]; m[i][k] = m[i][big]; m[i][big] = t; }
      { double t = m[k][k]; m[k][k] = m[big][big]; m[big][big] = t; }
      for (int i = k + 1; i < big; ++i) { double t = m[i][k]; m[i][k] = m[big][i]; m[big][i] = t; }
    }
    if (k > 0) {
      double temp[N];
      for (int i = 0; i < k; ++i) temp[i] = m[i][i] * m[k][i];
      double s = 0.0;
      for (int i = 0; i < k; ++i) s += m[k][i] * temp[i];
      m[k][k] -= s;
      for (int r = k + 1; r < N; ++r) {
        double a = 0.0;
        for (int i = 0; i < k; ++i) a += m[r][i] * temp[i];
        m[r][k] -= a;
      }
    }
    double akk = m[k][k];
    bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) {
      for (int j = 0; j < N; ++j) tr[j] = j;
      all_zero = true;
    } else if (valid) {
      for (int r = k + 1; r < N; ++r) m[r][k] /= akk;
    }
  }
  double d[N];
  for (int i = 0; i < N; ++i) d[i] = b[i];
  for (int k = 0; k < N; ++k)
    if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  ldlt_forward_row<N, 1>(m, d);        // L^-1 (unit lower), Eigen's summation order
  const double tol = 2.2250738585072014e-308;   // DBL_MIN: pseudo-inverse of D
  for (int i = 0; i < N; ++i) {
    if (fabs(m[i][i]) > tol) d[i] /= m[i][i]; else d[i] = 0.0;
  }
  ldlt_backward_row<N, N - 2>(m, d);   // L^-T
  for (int k = N - 1; k >= 0; --k)
    if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < N; ++i) x[i] = d[i];
}
SVO_DEV void ldlt6_solve(const double* Hin, const double* b, double* x) { ldlt_solve<6>(Hin, b, x); }

// vk::interpolateMat_8u (I/vision.h:19-36)
SVO_DEV float interpolate_8u(const uint8_t* img, int stride, float u, float v) {
  int x = (int)floorf(u);
  int y = (int)floorf(v);
  float subpix_x = u - x;
  float subpix_y = v - y;
  float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
  float w01 = (1.0f - subpix_x) * subpix_y;
  float w10 = subpix_x * (1.0f - subpix_y);
  float w11 = 1.0f - w00 - w01 - w10;
  const uint8_t* ptr = img + y * stride + x;
  return w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1];
}

// butterfly sum over `width` consecutive lanes (width = 16 or 64), all lanes get the sum
template <int WIDTH>
SVO_DEV double group_sum(double v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int WIDTH>
SVO_DEV float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int WIDTH>
SVO_DEV int group_sum(int v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- quad (4-lane) communication through DPP quad_perm: no LDS, no bpermute ------------------
template <int CTRL>
SVO_DEV int dpp_quad(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
SVO_DEV float dpp_quad(float v) { return __int_as_float(dpp_quad<CTRL>(__float_as_int(v))); }
template <int CTRL>
SVO_DEV double dpp_quad(double v) {
  const int lo = dpp_quad<CTRL>(__double2loint(v)), hi = dpp_quad<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// value of lane S (0..3) of the caller's quad
template <int S, typename T>
SVO_DEV T quad_bcast(T v) { return dpp_quad<S * 0x55>(v); }
// sum over the 4 lanes of a quad, all lanes get it:  (v0+v1)+(v2+v3) up to commutation
SVO_DEV double quad_sum(double v) { v += dpp_quad<0xB1>(v); v += dpp_quad<0x4E>(v); return v; }
SVO_DEV float quad_sum(float v) { v += dpp_quad<0xB1>(v); v += dpp_quad<0x4E>(v); return v; }

// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), every lane gets it, no LDS traffic and no index
// arithmetic: quad_perm for lanes ^1 and ^2, row_half_mirror for the other quad of the half row, row_mirror for the
// other half.  The additions pair the same partial sums as the xor butterfly, so the value is bit-identical to it.
SVO_DEV float row16_sum(float v) {
  v += dpp_quad<0xB1>(v);
  v += dpp_quad<0x4E>(v);
  v += dpp_quad<0x141>(v);               // row_half_mirror
  v += dpp_quad<0x140>(v);               // row_mirror
  return v;
}
SVO_DEV int row16_sum(int v) {
  v += dpp_quad<0xB1>(v);
  v += dpp_quad<0x4E>(v);
  v += dpp_quad<0x141>(v);
  v += dpp_quad<0x140>(v);
  return v;
}

// Sum 8 per-lane doubles over the 64 lanes of a wave with 7 long-range exchanges (lane swaps and DPP, no LDS
// traffic) instead of 48: every
// exchange step halves the number of values a lane still carries (lanes with the exchanged bit set keep the
// upper half of the values and hand over the lower half).  On return lanes 8j .. 8j+7 all hold the wave
// total of v[j].  The order of the additions is fixed, so the result is reproducible run to run.
// gfx950 lane swaps (VALU, no LDS round trip): v_permlane32_swap exchanges lanes 32..63 of the first operand with
// lanes 0..31 of the second, v_permlane16_swap the odd 16-lane rows of the first with the even rows of the second.
SVO_DEV void permlane32_swap(double& a, double& b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]);
  b = __hiloint2double((int)hi[1], (int)lo[1]);
}
SVO_DEV void permlane16_swap(double& a, double& b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]);
  b = __hiloint2double((int)hi[1], (int)lo[1]);
}

SVO_DEV double wave_reduce8(const double* v) {
  const int lane = threadIdx.x & 63;
  const bool b3 = (lane & 8) != 0;
  double a[4], c[2];
  // lanes 0..31 keep v[i] and add the other half's v[i]; lanes 32..63 keep v[i+4] and add the other half's v[i+4]:
  // after the swap the first operand holds {own v[i] | partner's v[i+4]} and the second {partner's v[i] | own v[i+4]}
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double p = v[i], q = v[i + 4];
    permlane32_swap(p, q);
    a[i] = p + q;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    double p = a[i], q = a[i + 2];
    permlane16_swap(p, q);
    c[i] = p + q;
  }
  const double keep = b3 ? c[1] : c[0], send = b3 ? c[0] : c[1];
  double r = keep + dpp_quad<0x128>(send);   // DPP row_ror:8 = the lane 8 further on in the 16-lane row (xor 8)
  r = quad_sum(r);                       // lanes xor 1, xor 2 (DPP quad_perm)
  r += dpp_quad<0x141>(r);               // DPP row_half_mirror: the other quad of the 8-lane group
  return r;
}

// Pivoted LDL^T solve of a symmetric 6x6 with every index static (registers only, no scratch):
// same algorithm and operation order as ldlt6_solve above.  MUST be called with exactly one active
// lane per wave (the callers run it on lane 0 only): the pivot index is read with readfirstlane so that
// the row/column swaps are real scalar branches instead of ~500 predicated moves.
// factorisation half: m = the in-place LDL^T (unit lower L below the diagonal, D on it), tr = the transpositions
SVO_DEV void ldlt6_factor_reg(const double* Hin, double (*m)[6], int* tr) {
  constexpr int N = 6;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) m[i][j] = Hin[i * N + j];
  bool all_zero = false;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (!all_zero) {
      int big = k;
      double best = fabs(m[k][k]);
#pragma unroll
      for (int i = k + 1; i < N; ++i)
        if (fabs(m[i][i]) > best) { best = fabs(m[i][i]); big = i; }
      big = __builtin_amdgcn_readfirstlane(big);
      tr[k] = big;
#pragma unroll
      for (int c = k + 1; c < N; ++c) {
        if (big == c) {
#pragma unroll
          for (int j = 0; j < k; ++j) { double t = m[k][j]; m[k][j] = m[c][j]; m[c][j] = t; }
#pragma unroll
          for (int i = c + 1; i < N; ++i) { double t = m[i][k]; m[i][k] = m[i][c]; m[i][c] = t; }
          { double t = m[k][k]; m[k][k] = m[c][c]; m[c][c] = t; }
#pragma unroll
          for (int i = k + 1; i < c; ++i) { double t = m[i][k]; m[i][k] = m[c][i]; m[c][i] = t; }
        }
      }
      if (k > 0) {
        double temp[N];
#pragma unroll
        for (int i = 0; i < k; ++i) temp[i] = m[i][i] * m[k][i];
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < k; ++i) s += m[k][i] * temp[i];
        m[k][k] -= s;
#pragma unroll
        for (int r = k + 1; r < N; ++r) {
          double a = 0.0;
#pragma unroll
          for (int i = 0; i < k; ++i) a += m[r][i] * temp[i];
          m[r][k] -= a;
        }
      }
      const double akk = m[k][k];
      const bool valid = fabs(akk) > 0.0;
      if (k == 0 && !valid) {
#pragma unroll
        for (int j = 0; j < N; ++j) tr[j] = j;
        all_zero = true;
      } else if (valid) {
#pragma unroll
        for (int r = k + 1; r < N; ++r) m[r][k] /= akk;
      }
    }
  }
}

// substitution half: x = P^T L^-T D^+ L^-1 P b with the factor above (Eigen LDLT.h:574-613)
SVO_DEV void ldlt6_substitute_reg(const double (*m)[6], const int* tr, const double* b, double* x) {
  constexpr int N = 6;
  double d[N];
#pragma unroll
  for (int i = 0; i < N; ++i) d[i] = b[i];
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int c = k + 1; c < N; ++c)
      if (tr[k] == c) { double t = d[k]; d[k] = d[c]; d[c] = t; }
  }
  ldlt_forward_row<N, 1>(m, d);        // L^-1 (unit lower), Eigen's summation order
  const double tol = 2.2250738585072014e-308;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (fabs(m[i][i]) > tol) d[i] /= m[i][i]; else d[i] = 0.0;
  }
  ldlt_backward_row<N, N - 2>(m, d);   // L^-T
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
#pragma unroll
    for (int c = k + 1; c < N; ++c)
      if (tr[k] == c) { double t = d[k]; d[k] = d[c]; d[c] = t; }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = d[i];
}

SVO_DEV void ldlt6_solve_reg(const double* Hin, const double* b, double* x) {
  double m[6][6];
  int tr[6];
  ldlt6_factor_reg(Hin, m, tr);
  ldlt6_substitute_reg(m, tr, b, x);
}

}  // namespace svo_dev
