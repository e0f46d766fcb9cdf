/*
 * svo_oracle.c -- CPU restatement of the SVO hot path (TEST INFRASTRUCTURE ONLY).
 * See svo_oracle.h for scope, conventions and pinning status.
 *
 * Build with:  gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math
 * (no FMA contraction: the reference is built without it on x86-64 and the
 * float/double promotion of every statement below follows the C++ source).
 *
 * Citations: "S/" = /root/reference/app/src/main/cpp/svo/, "I/" = S/include/svo/.
 */
#include "svo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* SE3 / SO3                                                                 */
/* ------------------------------------------------------------------------ */

static void cross3(const double a[3], const double b[3], double o[3]) {
  /* I/SO3.h:106-109 Point3d::cross */
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}

/* I/SO3.h:478-483: p + w*uv + q x uv with uv = 2 (q x p) */
static void so3_rotate(const double q[4], const double p[3], double o[3]) {
  double uv[3], quv[3];
  cross3(q, p, uv);
  uv[0] = uv[0] + uv[0]; uv[1] = uv[1] + uv[1]; uv[2] = uv[2] + uv[2];
  cross3(q, uv, quv);
  double x = (p[0] + q[3] * uv[0]) + quv[0];
  double y = (p[1] + q[3] * uv[1]) + quv[1];
  double z = (p[2] + q[3] * uv[2]) + quv[2];
  o[0] = x; o[1] = y; o[2] = z;
}

/* I/SO3.h:468-474 */
static void so3_mul(const double a[4], const double b[4], double o[4]) {
  double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

void svo_orc_se3_identity(double T[7]) {
  T[0] = T[1] = T[2] = 0.0; T[3] = T[4] = T[5] = 0.0; T[6] = 1.0;
}

/* I/SE3.h:46-50 */
void svo_orc_se3_mul(const double A[7], const double B[7], double out[7]) {
  double q[4], rt[3];
  so3_mul(A + 3, B + 3, q);
  so3_rotate(A + 3, B, rt);
  double t0 = A[0] + rt[0], t1 = A[1] + rt[1], t2 = A[2] + rt[2];
  out[0] = t0; out[1] = t1; out[2] = t2;
  out[3] = q[0]; out[4] = q[1]; out[5] = q[2]; out[6] = q[3];
}

/* I/SE3.h:35-38, I/SO3.h:523-526 (conjugate; assumes unit quaternion) */
void svo_orc_se3_inverse(const double T[7], double out[7]) {
  double qi[4] = {-T[3], -T[4], -T[5], T[6]};
  double rt[3];
  so3_rotate(qi, T, rt);
  out[0] = -rt[0]; out[1] = -rt[1]; out[2] = -rt[2];
  out[3] = qi[0]; out[4] = qi[1]; out[5] = qi[2]; out[6] = qi[3];
}

/* I/SE3.h:58-62 */
void svo_orc_se3_act(const double T[7], const double p[3], double out[3]) {
  double r[3];
  so3_rotate(T + 3, p, r);
  out[0] = T[0] + r[0]; out[1] = T[1] + r[1]; out[2] = T[2] + r[2];
}

/* I/SE3.h:153-182.  Quirk kept: the translation part is unguarded, theta==0
 * gives 0/0 = NaN. */
void svo_orc_se3_exp(const double l[6], double out[7]) {
  const double p[3] = {l[0], l[1], l[2]};
  const double r[3] = {l[3], l[4], l[5]};
  double theta_sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  double theta = sqrt(theta_sq);
  double half_theta = 0.5 * theta;
  double imag_factor, real_factor;
  if (theta < 1e-10) {
    double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real_factor = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    double sin_half_theta = sin(half_theta);
    imag_factor = sin_half_theta / theta;
    real_factor = cos(half_theta);
  }
  double rxp[3], rxrxp[3];
  cross3(r, p, rxp);
  cross3(r, rxp, rxrxp);
  double c1 = (1 - cos(theta)) / theta_sq;
  double c2 = (theta - sin(theta)) / (theta_sq * theta);
  out[0] = (p[0] + c1 * rxp[0]) + c2 * rxrxp[0];
  out[1] = (p[1] + c1 * rxp[1]) + c2 * rxrxp[1];
  out[2] = (p[2] + c1 * rxp[2]) + c2 * rxrxp[2];
  out[3] = imag_factor * r[0];
  out[4] = imag_factor * r[1];
  out[5] = imag_factor * r[2];
  out[6] = real_factor;
}

/* I/SO3.h:210-241 */
void svo_orc_so3_log(const double q[4], double out[3]) {
  const double NEAR_ZERO = 1e-10, PI_ = 3.14159265358979323846;
  double w = q[3];
  double squared_w = w * w;
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  double A_inv;
  if (n < NEAR_ZERO) {
    A_inv = 2. / w - 2. * (1.0 - squared_w) / (w * squared_w);
  } else if (fabs(w) < NEAR_ZERO) {
    A_inv = (w > 0) ? PI_ / n : -PI_ / n;
  } else {
    A_inv = 2 * atan(n / w) / n;
  }
  out[0] = q[0] * A_inv; out[1] = q[1] * A_inv; out[2] = q[2] * A_inv;
}

/* I/SO3.h:391-406 via I/SE3.h:216-226; row-major 3x3 */
void svo_orc_se3_rotation_matrix(const double T[7], double m[9]) {
  double x = T[3], y = T[4], z = T[5], w = T[6];
  double x2 = x * x, y2 = y * y, z2 = z * z;
  double xy = x * y, xz = x * z, yz = y * z;
  double wx = w * x, wy = w * y, wz = w * z;
  m[0] = 1.0 - 2.0 * (y2 + z2); m[1] = 2.0 * (xy - wz);       m[2] = 2.0 * (xz + wy);
  m[3] = 2.0 * (xy + wz);       m[4] = 1.0 - 2.0 * (x2 + z2); m[5] = 2.0 * (yz - wx);
  m[6] = 2.0 * (xz - wy);       m[7] = 2.0 * (yz + wx);       m[8] = 1.0 - 2.0 * (x2 + y2);
}

/* ------------------------------------------------------------------------ */
/* camera                                                                    */
/* ------------------------------------------------------------------------ */

/* S/pinhole_camera.cpp:79-106 */
void svo_orc_world2cam_uv(const svo_orc_camera* c, const double uv[2], double px[2]) {
  if (!c->distortion) {
    px[0] = c->fx * uv[0] + c->cx;
    px[1] = c->fy * uv[1] + c->cy;
  } else {
    double x = uv[0], y = uv[1];
    double r2 = x * x + y * y;
    double r4 = r2 * r2;
    double r6 = r4 * r2;
    double a1 = 2 * x * y;
    double a2 = r2 + 2 * x * x;
    double a3 = r2 + 2 * y * y;
    double cdist = 1 + c->d[0] * r2 + c->d[1] * r4 + c->d[4] * r6;
    double xd = x * cdist + c->d[2] * a1 + c->d[3] * a2;
    double yd = y * cdist + c->d[2] * a3 + c->d[3] * a1;
    px[0] = xd * c->fx + c->cx;
    px[1] = yd * c->fy + c->cy;
  }
}

/* S/pinhole_camera.cpp:73-77 + I/math_utils.h:104-107 (true division) */
void svo_orc_world2cam(const svo_orc_camera* c, const double xyz[3], double px[2]) {
  double uv[2] = {xyz[0] / xyz[2], xyz[1] / xyz[2]};
  svo_orc_world2cam_uv(c, uv, px);
}

/* S/pinhole_camera.cpp:44-71; Eigen normalized() = v / sqrt(squaredNorm) when squaredNorm > 0.
 * Distorted cameras (:54-63) go through cv::undistortPoints(src CV_32FC2, dst CV_32FC2, cvK_, cvD_) with cvK_/cvD_
 * float matrices (:31-32).  PARITY UNPINNED: OpenCV 4.5.4 calib3d (undistort.dispatch.cpp, cvUndistortPointsInternal)
 * is third-party code that is not under /root/reference; restated from its published algorithm: the point and the
 * parameters are read as floats, x0 = (u - cx)/fx, y0 = (v - cy)/fy, five fixed-point iterations
 *   icdist = 1/(1 + ((k3 r2 + k2) r2 + k1) r2),  x = (x0 - dx) icdist,  y = (y0 - dy) icdist
 * with the tangential terms dx = 2 p1 x y + p2 (r2 + 2 x^2), dy = p1 (r2 + 2 y^2) + 2 p2 x y (a negative icdist
 * keeps the undistorted start value), in double, the result stored as float. */
void svo_orc_cam2world(const svo_orc_camera* c, double u, double v, double f[3]) {
  double x, y;
  if (!c->distortion) {
    x = (u - c->cx) / c->fx;
    y = (v - c->cy) / c->fy;
  } else {
    const double fx = (double)(float)c->fx, fy = (double)(float)c->fy, cx = (double)(float)c->cx, cy = (double)(float)c->cy;
    const double k1 = (double)(float)c->d[0], k2 = (double)(float)c->d[1], p1 = (double)(float)c->d[2],
                 p2 = (double)(float)c->d[3], k3 = (double)(float)c->d[4];
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

/* I/abstract_camera.h:62-70 */
static int is_in_frame_level(const svo_orc_camera* c, int ox, int oy, int boundary, int level) {
  return ox >= boundary && ox < c->width / (1 << level) - boundary &&
         oy >= boundary && oy < c->height / (1 << level) - boundary;
}

/* I/abstract_camera.h:58-72 -- both overloads: level < 0 selects isInFrame(obs, boundary) */
int svo_orc_is_in_frame(const svo_orc_camera* c, int ox, int oy, int boundary, int level) {
  if (level < 0)
    return ox >= boundary && ox < c->width - boundary && oy >= boundary && oy < c->height - boundary;
  return is_in_frame_level(c, ox, oy, boundary, level);
}

/* ------------------------------------------------------------------------ */
/* small algebra                                                             */
/* ------------------------------------------------------------------------ */

/* I/frame.h:110-132 */
void svo_orc_jacobian_xyz2uv(const double p[3], double J[12]) {
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

/* Eigen 3.4.0 Cholesky/LDLT.h:297-403 (unblocked, lower) + :574-613 (solve). */
/* sum of n terms as Eigen's redux_novec_unroller splits it: [0, n/2) + [n/2, n) recursively */
static double redux_tree(const double* a, int n) {
  if (n == 1) return a[0];
  const int h = n / 2;
  return redux_tree(a, h) + redux_tree(a + h, n - h);
}

/* the same sum when both operands have packet access (2 doubles per SSE2 packet): the packets are added by
 * binary splitting, the two lanes of the result are added, then the scalar remainder (Core/Redux.h) */
static double redux_packet2(const double* a, int n) {
  const int np = n / 2;
  if (np == 0) return redux_tree(a, n);
  double lo[3], hi[3];
  for (int k = 0; k < np; ++k) { lo[k] = a[2 * k]; hi[k] = a[2 * k + 1]; }
  double res = redux_tree(lo, np) + redux_tree(hi, np);
  if (2 * np != n) res = res + redux_tree(a + 2 * np, n - 2 * np);
  return res;
}

static int ldlt_solve_n(const int N, const double* Hin, const double* b, double* x) {
  double m[6][6];
  int tr[6];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) m[i][j] = Hin[i * N + j];

  for (int k = 0; k < N; ++k) {
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
      double temp[6];
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
    int valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) {
      for (int j = 0; j < N; ++j) tr[j] = j;
      break;
    }
    if (valid)
      for (int r = k + 1; r < N; ++r) m[r][k] /= akk;
  }

  double d[6];
  for (int i = 0; i < N; ++i) d[i] = b[i];
  for (int k = 0; k < N; ++k)
    if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  /* Eigen solves fixed-size triangular systems with an unrolled row-wise formula (Core/SolveTriangular.h:
   * rhs(I) -= lhs.row(I).segment(..).cwiseProduct(rhs.segment(..)).sum()): the products are summed first, by the
   * unrolled redux's binary splitting (Core/Redux.h), and the sum is subtracted once. */
  for (int i = 1; i < N; ++i) {            /* L^-1 (unit lower) */
    double t[6];
    for (int j = 0; j < i; ++j) t[j] = m[i][j] * d[j];
    d[i] -= redux_tree(t, i);
  }
  const double tol = DBL_MIN;              /* pseudo-inverse of D, :595-603 */
  for (int i = 0; i < N; ++i) {
    if (fabs(m[i][i]) > tol) d[i] /= m[i][i]; else d[i] = 0.0;
  }
  for (int i = N - 2; i >= 0; --i) {       /* L^-T */
    double t[6];
    for (int j = i + 1; j < N; ++j) t[j - i - 1] = m[j][i] * d[j];
    d[i] -= redux_packet2(t, N - 1 - i);   /* column i of L is contiguous: Eigen vectorises this sum (SSE2) */
  }
  for (int k = N - 1; k >= 0; --k)
    if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < N; ++i) x[i] = d[i];
  return 1;
}

int svo_orc_ldlt6_solve(const double Hin[36], const double b[6], double x[6]) { return ldlt_solve_n(6, Hin, b, x); }
int svo_orc_ldlt3_solve(const double Ain[9], const double b[3], double x[3]) { return ldlt_solve_n(3, Ain, b, x); }

/* ------------------------------------------------------------------------ */
/* image helpers                                                             */
/* ------------------------------------------------------------------------ */

/* S/vision.cpp:89-110 */
void svo_orc_half_sample(const uint8_t* in, int w, int h, uint8_t* out) {
  int ow = w / 2, oh = h / 2;
  for (int y = 0; y < oh; ++y) {
    const uint8_t* top = in + (size_t)(2 * y) * w;
    const uint8_t* bot = top + w;
    uint8_t* p = out + (size_t)y * ow;
    for (int x = 0; x < ow; ++x)
      p[x] = (uint8_t)(((uint16_t)top[2 * x] + top[2 * x + 1] + bot[2 * x] + bot[2 * x + 1]) / 4);
  }
}

/* S/vision.cpp:20-45: vertical pavgb then horizontal pavgw: (a+b+1)>>1 twice */
void svo_orc_half_sample_sse2form(const uint8_t* in, int w, int h, uint8_t* out) {
  int ow = w / 2, oh = h / 2;
  for (int y = 0; y < oh; ++y) {
    const uint8_t* top = in + (size_t)(2 * y) * w;
    const uint8_t* bot = top + w;
    uint8_t* p = out + (size_t)y * ow;
    for (int x = 0; x < ow; ++x) {
      int v0 = (top[2 * x] + bot[2 * x] + 1) >> 1;
      int v1 = (top[2 * x + 1] + bot[2 * x + 1] + 1) >> 1;
      p[x] = (uint8_t)((v0 + v1 + 1) >> 1);
    }
  }
}

/* I/vision.h:19-36 */
float svo_orc_interpolate_8u(const uint8_t* img, int stride, float u, float v) {
  int x = (int)floor(u);
  int y = (int)floor(v);
  float subpix_x = u - x;
  float subpix_y = v - y;
  float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
  float w01 = (1.0f - subpix_x) * subpix_y;
  float w10 = subpix_x * (1.0f - subpix_y);
  float w11 = 1.0f - w00 - w01 - w10;
  const uint8_t* ptr = img + y * stride + x;
  return w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1];
}

/* ------------------------------------------------------------------------ */
/* SparseImgAlign                                                            */
/* ------------------------------------------------------------------------ */

typedef struct {
  const svo_orc_camera* cam;
  const uint8_t* const* ref_pyr;
  const uint8_t* const* cur_pyr;
  int n;
  const double *px, *f, *pos;
  const uint8_t* has_point;
  double ref_pos[3];
  int level;
  int have_ref_patch_cache;
  float* ref_patch_cache;   /* [n][16] */
  double* jac;              /* [n][16][6]  (column-major 6 x 16n in the reference) */
  uint8_t* visible;         /* [n] sticky */
  /* NLLS state, I/nlls_solver.h:51-60,96-111 */
  double H[36], Jres[6], x[6];
  double chi2;
  size_t n_meas;
  int stop;
  long n_pre, n_res;
  /* the other branches of NLLSSolver (I/nlls_solver.h:46-48,96-111): method_, robust cost, LM damping */
  int method;               /* 0 GaussNewton, 1 LevenbergMarquardt */
  int use_weights;          /* setRobustCostFunction: true unless the scale estimator is UnitScale */
  int scale_kind, weight_kind;
  float scale;              /* scale_ (0.0 until the first estimate) */
  size_t iter;              /* iter_: survives from one level's optimize() into the next one's weight-scale pass */
  double mu, nu;
  int evals;
} sia_state;

/* S/sparse_img_align.cpp:105-178 */
static void sia_precompute(sia_state* s) {
  const int border = 2 + 1;
  const int L = s->level;
  const uint8_t* ref_img = s->ref_pyr[L];
  const int cols = s->cam->width >> L, rows = s->cam->height >> L;
  const int stride = cols;
  const float scale = 1.0f / (1 << L);
  const double focal_length = fabs(s->cam->fx);          /* errorMultiplier2 */
  const double jscale = focal_length / (1 << L);
  for (int i = 0; i < s->n; ++i) {
    const float u_ref = (float)(s->px[2 * i] * scale);
    const float v_ref = (float)(s->px[2 * i + 1] * scale);
    const int u_ref_i = (int)floorf(u_ref);
    const int v_ref_i = (int)floorf(v_ref);
    if (!s->has_point[i] || u_ref_i - border < 0 || v_ref_i - border < 0 ||
        u_ref_i + border >= cols || v_ref_i + border >= rows)
      continue;
    s->visible[i] = 1;
    s->n_pre++;

    const double dxp = s->pos[3 * i] - s->ref_pos[0];
    const double dyp = s->pos[3 * i + 1] - s->ref_pos[1];
    const double dzp = s->pos[3 * i + 2] - s->ref_pos[2];
    const double depth = sqrt(dxp * dxp + dyp * dyp + dzp * dzp);
    const double xyz_ref[3] = {s->f[3 * i] * depth, s->f[3 * i + 1] * depth, s->f[3 * i + 2] * depth};
    double fj[12];
    svo_orc_jacobian_xyz2uv(xyz_ref, fj);

    const float subpix_u_ref = u_ref - u_ref_i;
    const float subpix_v_ref = v_ref - v_ref_i;
    const float w_tl = (float)((1.0 - subpix_u_ref) * (1.0 - subpix_v_ref));
    const float w_tr = (float)(subpix_u_ref * (1.0 - subpix_v_ref));
    const float w_bl = (float)((1.0 - subpix_u_ref) * subpix_v_ref);
    const float w_br = subpix_u_ref * subpix_v_ref;
    float* cache = s->ref_patch_cache + 16 * (size_t)i;
    double* jc = s->jac + 96 * (size_t)i;
    int pix = 0;
    for (int y = 0; y < 4; ++y) {
      const uint8_t* p = ref_img + (v_ref_i + y - 2) * stride + (u_ref_i - 2);
      for (int x = 0; x < 4; ++x, ++p, ++pix) {
        cache[pix] = w_tl * p[0] + w_tr * p[1] + w_bl * p[stride] + w_br * p[stride + 1];
        float dx = 0.5f * ((w_tl * p[1] + w_tr * p[2] + w_bl * p[stride + 1] + w_br * p[stride + 2]) -
                           (w_tl * p[-1] + w_tr * p[0] + w_bl * p[stride - 1] + w_br * p[stride]));
        float dy = 0.5f * ((w_tl * p[stride] + w_tr * p[1 + stride] + w_bl * p[stride * 2] + w_br * p[stride * 2 + 1]) -
                           (w_tl * p[-stride] + w_tr * p[1 - stride] + w_bl * p[0] + w_br * p[1]));
        for (int k = 0; k < 6; ++k)
          jc[6 * pix + k] = ((double)dx * fj[k] + (double)dy * fj[6 + k]) * jscale;
      }
    }
  }
  s->have_ref_patch_cache = 1;
}

/* ---- robust cost, S/robust_cost.cpp (all f32, evaluated in the order of the errors vector) */
/* TDistributionScaleEstimator::compute (:38-66), dof 5, initial sigma 5 */
static float scale_tdist(const float* e, size_t n) {
  const float dof = 5.0f;
  float initial_lamda = 1.0f / (5.0f * 5.0f);
  int num = 0;
  float lambda = initial_lamda;
  do {
    initial_lamda = lambda;
    num = 0;
    lambda = 0.0f;
    for (size_t i = 0; i < n; ++i) {
      if (isfinite(e[i])) {
        ++num;
        const float error2 = e[i] * e[i];
        lambda += error2 * ((dof + 1.0f) / (dof + initial_lamda * error2));
      }
    }
    lambda = (float)num / lambda;
  } while ((double)fabsf(lambda - initial_lamda) > 1e-3);
  return sqrtf(1.0f / lambda);
}

static int cmp_float_asc(const void* a, const void* b) {
  const float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}

/* MADScaleEstimator::compute (:70-75): 1.48 * the element nth_element leaves at floor(n/2) (I/math_utils.h:124-131) */
static float scale_mad(float* e, size_t n) {
  if (n == 0) return NAN;                                   /* (the reference reads past an empty vector here) */
  qsort(e, n, sizeof(float), cmp_float_asc);
  return 1.48f * e[n / 2];
}

/* NormalDistributionScaleEstimator::compute (:77-86): std::accumulate with an int seed truncates the running sum to
 * int at every step, the mean is an integer quotient, and the value returned is sqrt of the SUM of squares */
static float scale_normal(const float* e, size_t n) {
  if (n == 0) return NAN;                                   /* (integer division by zero in the reference) */
  int acc = 0;
  for (size_t i = 0; i < n; ++i) acc = (int)((float)acc + e[i]);
  const float mean = (float)((size_t)acc / n);
  float var = 0.0f;
  for (size_t i = 0; i < n; ++i) var += (e[i] - mean) * (e[i] - mean);
  return sqrtf(var);
}

/* WeightFunction::value (:94-160): TDist dof 5, Tukey b 8.6851, Huber k 1.345 */
static float robust_weight(int kind, float x) {
  switch (kind) {
    case 1: return (5.0f + 1.0f) / (5.0f + (x * x));
    case 2: {
      const float b_square = 8.6851f * 8.6851f;
      const float x_square = x * x;
      if (x_square <= b_square) { const float tmp = 1.0f - x_square / b_square; return tmp * tmp; }
      return 0.0f;
    }
    case 3: {
      const float t_abs = fabsf(x);
      if (t_abs < 1.345f) return 1.0f;
      return 1.345f / t_abs;
    }
    default: return 1.0f;
  }
}

/* S/sparse_img_align.cpp:184-286 */
static double sia_compute_residuals_w(sia_state* s, const double T_cur_from_ref[7], int linearize, int compute_weight_scale);
static double sia_compute_residuals(sia_state* s, const double T_cur_from_ref[7], int linearize) {
  return sia_compute_residuals_w(s, T_cur_from_ref, linearize, 0);
}
static double sia_compute_residuals_w(sia_state* s, const double T_cur_from_ref[7], int linearize, int compute_weight_scale) {
  const int L = s->level;
  const uint8_t* cur_img = s->cur_pyr[L];
  if (!s->have_ref_patch_cache) sia_precompute(s);
  const int cols = s->cam->width >> L, rows = s->cam->height >> L;
  const int stride = cols;
  const int border = 2 + 1;
  const float scale = 1.0f / (1 << L);
  float chi2 = 0.0f;
  float* errors = NULL;
  size_t n_err = 0;
  if (compute_weight_scale) errors = (float*)malloc(sizeof(float) * 16 * (size_t)(s->n > 0 ? s->n : 1));
  for (int i = 0; i < s->n; ++i) {
    if (!s->visible[i]) continue;
    const double dxp = s->pos[3 * i] - s->ref_pos[0];
    const double dyp = s->pos[3 * i + 1] - s->ref_pos[1];
    const double dzp = s->pos[3 * i + 2] - s->ref_pos[2];
    const double depth = sqrt(dxp * dxp + dyp * dyp + dzp * dzp);
    const double xyz_ref[3] = {s->f[3 * i] * depth, s->f[3 * i + 1] * depth, s->f[3 * i + 2] * depth};
    double xyz_cur[3], pxd[2];
    svo_orc_se3_act(T_cur_from_ref, xyz_ref, xyz_cur);
    svo_orc_world2cam(s->cam, xyz_cur, pxd);
    const float u_cur = (float)pxd[0] * scale;
    const float v_cur = (float)pxd[1] * scale;
    const int u_cur_i = (int)floorf(u_cur);
    const int v_cur_i = (int)floorf(v_cur);
    if (u_cur_i < 0 || v_cur_i < 0 || u_cur_i - border < 0 || v_cur_i - border < 0 ||
        u_cur_i + border >= cols || v_cur_i + border >= rows)
      continue;
    s->n_res++;
    const float subpix_u = u_cur - u_cur_i;
    const float subpix_v = v_cur - v_cur_i;
    const float w_tl = (float)((1.0 - subpix_u) * (1.0 - subpix_v));
    const float w_tr = (float)(subpix_u * (1.0 - subpix_v));
    const float w_bl = (float)((1.0 - subpix_u) * subpix_v);
    const float w_br = subpix_u * subpix_v;
    const float* cache = s->ref_patch_cache + 16 * (size_t)i;
    const double* jc = s->jac + 96 * (size_t)i;
    int pix = 0;
    for (int y = 0; y < 4; ++y) {
      const uint8_t* p = cur_img + (v_cur_i + y - 2) * stride + (u_cur_i - 2);
      for (int x = 0; x < 4; ++x, ++pix, ++p) {
        const float intensity_cur = w_tl * p[0] + w_tr * p[1] + w_bl * p[stride] + w_br * p[stride + 1];
        const float res = intensity_cur - cache[pix];
        if (compute_weight_scale) errors[n_err++] = fabsf(res);             /* :256-257 */
        float weight = 1.0f;
        if (s->use_weights) weight = robust_weight(s->weight_kind, res / s->scale);   /* :260-263 */
        chi2 += res * res * weight;
        s->n_meas++;
        if (linearize) {
          const double* J = jc + 6 * pix;
          const double w = (double)weight, r = (double)res;
          for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) s->H[a * 6 + b] += J[a] * J[b] * w;
            s->Jres[a] -= J[a] * r * w;
          }
        }
      }
    }
  }
  /* compute the weights on the first iteration (:281-283) */
  if (compute_weight_scale) {
    if (s->iter == 0) {
      switch (s->scale_kind) {
        case 1: s->scale = scale_tdist(errors, n_err); break;
        case 2: s->scale = scale_mad(errors, n_err); break;
        case 3: s->scale = scale_normal(errors, n_err); break;
        default: s->scale = 1.0f; break;
      }
    }
    free(errors);
  }
  /* float / size_t -> float division, then widened (:285) */
  return (double)(chi2 / (float)s->n_meas);
}

/* I/math_utils.h:91-102 */
static double norm_max6(const double* v) {
  double mx = -1;
  for (int i = 0; i < 6; ++i) { double a = fabs(v[i]); if (a > mx) mx = a; }
  return mx;
}

/* I/nlls_solver_impl.hpp:25-100 with S/sparse_img_align.cpp:291-308 plugged in.
 * early_stop == 0 ("fixed work"): the error-increase and |x|<=eps exits are
 * disabled so exactly n_iter evaluations run per level (a NaN solve still stops). */
static int sia_optimize_gn(sia_state* s, double model[7], int n_iter, double eps, int early_stop) {
  if (s->use_weights) sia_compute_residuals_w(s, model, 0, 1);          /* :28-29 */
  double old_model[7];
  memcpy(old_model, model, sizeof(old_model));
  int evals = 0;
  int iter;
  for (iter = 0; iter < n_iter; ++iter) {
    s->iter = (size_t)iter;
    memset(s->H, 0, sizeof(s->H));
    memset(s->Jres, 0, sizeof(s->Jres));
    s->n_meas = 0;
    double new_chi2 = sia_compute_residuals(s, model, 1);
    ++evals;
    svo_orc_ldlt6_solve(s->H, s->Jres, s->x);
    if (isnan(s->x[0])) s->stop = 1;
    if ((early_stop && iter > 0 && new_chi2 > s->chi2) || s->stop) {
      memcpy(model, old_model, sizeof(old_model));
      break;
    }
    double mx[6], dT[7], new_model[7];
    for (int k = 0; k < 6; ++k) mx[k] = -s->x[k];
    svo_orc_se3_exp(mx, dT);
    svo_orc_se3_mul(model, dT, new_model);
    memcpy(old_model, model, sizeof(old_model));
    memcpy(model, new_model, sizeof(new_model));
    s->chi2 = new_chi2;
    if (early_stop && norm_max6(s->x) <= eps) break;
  }
  s->iter = (size_t)iter;                                               /* iter_ as the loop leaves it */
  return evals;
}

/* I/nlls_solver_impl.hpp:102-227 with S/sparse_img_align.cpp:291-308 plugged in (have_prior_ false, mu_ >= 0).
 * As in the reference: n_meas_ is NOT cleared before the weight-scale pass and the first evaluation, so chi2_ of a
 * level starts as that evaluation's f32 sum over (what the previous level left + this level's counts); H_ is damped in
 * place and is what getInformationMatrix() returns afterwards; stop_ survives into the next level. */
static int sia_optimize_lm(sia_state* s, double model[7], int n_iter, double eps) {
  if (s->use_weights) sia_compute_residuals_w(s, model, 0, 1);          /* :105-106 */
  s->chi2 = sia_compute_residuals(s, model, 1);                         /* :109 */
  int evals = 1;
  const int n_trials_max = 5;
  size_t iter;
  for (iter = 0; iter < (size_t)n_iter; ++iter) {
    s->iter = iter;
    int n_trials = 0;
    double rho;
    do {
      double new_model[7];
      double new_chi2 = -1;
      memset(s->H, 0, sizeof(s->H));
      memset(s->Jres, 0, sizeof(s->Jres));
      s->n_meas = 0;
      sia_compute_residuals(s, model, 1);
      ++evals;
      for (int k = 0; k < 6; ++k) s->H[7 * k] += s->H[7 * k] * s->mu;   /* H_ += (H_.diagonal()*mu_).asDiagonal() (:150) */
      svo_orc_ldlt6_solve(s->H, s->Jres, s->x);
      if (!isnan(s->x[0])) {
        double mx[6], dT[7];
        for (int k = 0; k < 6; ++k) mx[k] = -s->x[k];
        svo_orc_se3_exp(mx, dT);
        svo_orc_se3_mul(model, dT, new_model);
        s->n_meas = 0;
        new_chi2 = sia_compute_residuals(s, new_model, 0);
        ++evals;
        rho = s->chi2 - new_chi2;
      } else {
        rho = -1;
      }
      if (rho > 0) {
        memcpy(model, new_model, sizeof(new_model));
        s->chi2 = new_chi2;
        s->stop = norm_max6(s->x) <= eps;
        const double c = 1. - pow(2 * rho - 1, 3);
        const double m = c < 2. / 3. ? c : 2. / 3.;
        s->mu *= (1. / 3. > m ? 1. / 3. : m);                            /* max(1./3., min(1.-pow(2*rho_-1,3), 2./3.)) */
        s->nu = 2.;
      } else {
        s->mu *= s->nu;
        s->nu *= 2.;
        ++n_trials;
        if (n_trials >= n_trials_max) s->stop = 1;
      }
    } while (!(rho > 0 || s->stop));
    if (s->stop) break;
  }
  s->iter = iter;
  return evals;
}

/* S/sparse_img_align.cpp:51-92 */
int svo_orc_sparse_img_align(
    const svo_orc_camera* cam, const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
    int n_feat, const double* px, const double* f, const double* pos, const uint8_t* has_point,
    const double T_ref_w[7], const double T_cur_w_init[7], const svo_orc_sia_params* prm,
    svo_orc_sia_result* out) {
  return svo_orc_sparse_img_align_ex(cam, ref_pyr, cur_pyr, n_feat, px, f, pos, has_point, T_ref_w, T_cur_w_init, prm, 0, 0, 0, out, NULL);
}

int svo_orc_sparse_img_align_ex(
    const svo_orc_camera* cam, const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
    int n_feat, const double* px, const double* f, const double* pos, const uint8_t* has_point,
    const double T_ref_w[7], const double T_cur_w_init[7], const svo_orc_sia_params* prm,
    int method, int scale_estimator, int weight_function, svo_orc_sia_result* out, float* scale_out) {
  memset(out, 0, sizeof(*out));
  memcpy(out->T_cur_w, T_cur_w_init, 7 * sizeof(double));
  out->chi2 = 1e10;
  if (n_feat <= 0) return 0;                          /* :55-59 */
  sia_state s;
  memset(&s, 0, sizeof(s));
  s.cam = cam; s.ref_pyr = ref_pyr; s.cur_pyr = cur_pyr;
  s.n = n_feat; s.px = px; s.f = f; s.pos = pos; s.has_point = has_point;
  s.ref_patch_cache = (float*)calloc((size_t)n_feat * 16, sizeof(float));
  s.jac = (double*)calloc((size_t)n_feat * 96, sizeof(double));
  s.visible = (uint8_t*)calloc((size_t)n_feat, 1);
  s.chi2 = 1e10;                                      /* reset(), nlls_solver_impl.hpp:299-309 */
  s.method = method;
  s.scale_kind = scale_estimator; s.weight_kind = weight_function;
  s.use_weights = scale_estimator != 0;               /* setRobustCostFunction (:234-262): UnitScale switches the weights off */
  s.scale = 0.0f;
  s.iter = 0;
  s.nu = 2.0;                                         /* nu_init_, restored by reset() */
  double T_ref_inv[7], T_cur_from_ref[7];
  svo_orc_se3_inverse(T_ref_w, T_ref_inv);
  s.ref_pos[0] = T_ref_inv[0]; s.ref_pos[1] = T_ref_inv[1]; s.ref_pos[2] = T_ref_inv[2];
  svo_orc_se3_mul(T_cur_w_init, T_ref_inv, T_cur_from_ref);
  for (int L = prm->max_level; L >= prm->min_level; --L) {
    s.level = L;
    memset(s.jac, 0, (size_t)n_feat * 96 * sizeof(double));    /* :76 */
    s.have_ref_patch_cache = 0;
    s.mu = 0.1;                                                /* :74 */
    int ev = method == 1 ? sia_optimize_lm(&s, T_cur_from_ref, prm->n_iter, prm->eps)
                         : sia_optimize_gn(&s, T_cur_from_ref, prm->n_iter, prm->eps, prm->early_stop);
    if (L < SVO_ORACLE_MAX_LEVELS) out->iters[L] = ev;
  }
  svo_orc_se3_mul(T_cur_from_ref, T_ref_w, out->T_cur_w);      /* :89 */
  out->n_tracked = s.n_meas / 16;
  memcpy(out->H, s.H, sizeof(s.H));
  memcpy(out->Jres, s.Jres, sizeof(s.Jres));
  out->chi2 = s.chi2;
  out->stop = s.stop;
  out->n_precompute_patches = s.n_pre;
  out->n_residual_patches = s.n_res;
  if (scale_out) *scale_out = s.scale;
  free(s.ref_patch_cache); free(s.jac); free(s.visible);
  return 0;
}

/* Step-wise access to the restated residual body, so that the reference's own
 * NLLSSolver driver (oracle/ref/ref_harness.cpp) can be run on top of it. */
void* svo_orc_sia_open(const svo_orc_camera* cam, const uint8_t* const* ref_pyr,
                       const uint8_t* const* cur_pyr, int n_feat, const double* px,
                       const double* f, const double* pos, const uint8_t* has_point,
                       const double T_ref_w[7]) {
  sia_state* s = (sia_state*)calloc(1, sizeof(sia_state));
  s->cam = cam; s->ref_pyr = ref_pyr; s->cur_pyr = cur_pyr;
  s->n = n_feat; s->px = px; s->f = f; s->pos = pos; s->has_point = has_point;
  s->ref_patch_cache = (float*)calloc((size_t)n_feat * 16, sizeof(float));
  s->jac = (double*)calloc((size_t)n_feat * 96, sizeof(double));
  s->visible = (uint8_t*)calloc((size_t)n_feat, 1);
  s->chi2 = 1e10;
  double T_ref_inv[7];
  svo_orc_se3_inverse(T_ref_w, T_ref_inv);
  s->ref_pos[0] = T_ref_inv[0]; s->ref_pos[1] = T_ref_inv[1]; s->ref_pos[2] = T_ref_inv[2];
  return s;
}

void svo_orc_sia_set_level(void* h, int level) {
  sia_state* s = (sia_state*)h;
  s->level = level;
  memset(s->jac, 0, (size_t)s->n * 96 * sizeof(double));
  s->have_ref_patch_cache = 0;
}

double svo_orc_sia_eval(void* h, const double T_cur_from_ref[7], int linearize, double H[36],
                        double Jres[6], size_t* n_meas) {
  sia_state* s = (sia_state*)h;
  memset(s->H, 0, sizeof(s->H));
  memset(s->Jres, 0, sizeof(s->Jres));
  s->n_meas = 0;
  double r = sia_compute_residuals(s, T_cur_from_ref, linearize);
  if (H) memcpy(H, s->H, sizeof(s->H));
  if (Jres) memcpy(Jres, s->Jres, sizeof(s->Jres));
  if (n_meas) *n_meas = s->n_meas;
  return r;
}

void svo_orc_sia_close(void* h) {
  sia_state* s = (sia_state*)h;
  if (!s) return;
  free(s->ref_patch_cache); free(s->jac); free(s->visible); free(s);
}

int svo_orc_sia_single_eval(
    const svo_orc_camera* cam, const uint8_t* ref_img, const uint8_t* cur_img, int level,
    int n_feat, const double* px, const double* f, const double* pos, const uint8_t* has_point,
    const double T_ref_w[7], const double T_cur_from_ref[7], double out28[28], long* n_meas,
    float* ref_patch_cache, double* jac_cache, uint8_t* visible) {
  const uint8_t* rp[SVO_ORACLE_MAX_LEVELS] = {0};
  const uint8_t* cp[SVO_ORACLE_MAX_LEVELS] = {0};
  if (level < 0 || level >= SVO_ORACLE_MAX_LEVELS) return -1;
  rp[level] = ref_img; cp[level] = cur_img;
  sia_state s;
  memset(&s, 0, sizeof(s));
  s.cam = cam; s.ref_pyr = rp; s.cur_pyr = cp;
  s.n = n_feat; s.px = px; s.f = f; s.pos = pos; s.has_point = has_point;
  s.ref_patch_cache = (float*)calloc((size_t)n_feat * 16, sizeof(float));
  s.jac = (double*)calloc((size_t)n_feat * 96, sizeof(double));
  s.visible = (uint8_t*)calloc((size_t)n_feat, 1);
  double T_ref_inv[7];
  svo_orc_se3_inverse(T_ref_w, T_ref_inv);
  s.ref_pos[0] = T_ref_inv[0]; s.ref_pos[1] = T_ref_inv[1]; s.ref_pos[2] = T_ref_inv[2];
  s.level = level;
  double mean_chi2 = sia_compute_residuals(&s, T_cur_from_ref, 1);
  int k = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) out28[k++] = s.H[a * 6 + b];
  for (int a = 0; a < 6; ++a) out28[k++] = s.Jres[a];
  out28[27] = mean_chi2;   /* (double)(float chi2 sum / (float)n_meas), as computeResiduals returns */
  *n_meas = (long)s.n_meas;
  if (ref_patch_cache) memcpy(ref_patch_cache, s.ref_patch_cache, (size_t)n_feat * 16 * sizeof(float));
  if (jac_cache) memcpy(jac_cache, s.jac, (size_t)n_feat * 96 * sizeof(double));
  if (visible) memcpy(visible, s.visible, (size_t)n_feat);
  free(s.ref_patch_cache); free(s.jac); free(s.visible);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* feature_alignment                                                         */
/* ------------------------------------------------------------------------ */

/* Eigen Matrix3f::inverse() (LU/InverseImpl.h, Size==3: cofactors / det) */
static void inverse3f(const float m[9], float inv[9]) {
  /* cofactor_3x3<i,j>: m(i1,j1)*m(i2,j2) - m(i1,j2)*m(i2,j1), i1=(i+1)%3 ... */
#define M(r, c) m[(r) * 3 + (c)]
#define COF(i, j) (M(((i) + 1) % 3, ((j) + 1) % 3) * M(((i) + 2) % 3, ((j) + 2) % 3) - \
                   M(((i) + 1) % 3, ((j) + 2) % 3) * M(((i) + 2) % 3, ((j) + 1) % 3))
  float c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
  /* Eigen's unrolled 3-term redux splits in halves: a0 + (a1 + a2) */
  float det = c00 * M(0, 0) + (c10 * M(1, 0) + c20 * M(2, 0));   /* cofactors_col0 . col(0) */
  float invdet = 1.0f / det;
  /* result.row(0) = cofactors_col0 * invdet; result(i,j) = cofactor(j,i) * invdet */
  inv[0] = c00 * invdet; inv[1] = c10 * invdet; inv[2] = c20 * invdet;
  inv[3] = COF(0, 1) * invdet; inv[4] = COF(1, 1) * invdet; inv[5] = COF(2, 1) * invdet;
  inv[6] = COF(0, 2) * invdet; inv[7] = COF(1, 2) * invdet; inv[8] = COF(2, 2) * invdet;
#undef COF
#undef M
}

/* S/feature_alignment.cpp:154-282 (scalar path :167-281) */
int svo_orc_align2d(const uint8_t* cur_img, int cols, int rows, int cur_step,
                    const uint8_t* ref_patch_with_border, const uint8_t* ref_patch,
                    int n_iter, double px_inout[2], int* iters_done) {
  const int halfpatch = 4, patch_size = 8;
  int converged = 0;
  float dxs[64], dys[64];
  float H[9] = {0};
  const int ref_step = patch_size + 2;
  int k = 0;
  for (int y = 0; y < patch_size; ++y) {
    const uint8_t* it = ref_patch_with_border + (y + 1) * ref_step + 1;
    for (int x = 0; x < patch_size; ++x, ++it, ++k) {
      float J[3];
      J[0] = (float)(0.5 * (it[1] - it[-1]));
      J[1] = (float)(0.5 * (it[ref_step] - it[-ref_step]));
      J[2] = 1;
      dxs[k] = J[0]; dys[k] = J[1];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) H[a * 3 + b] += J[a] * J[b];
    }
  }
  float Hinv[9];
  inverse3f(H, Hinv);
  float mean_diff = 0;
  float u = (float)px_inout[0];
  float v = (float)px_inout[1];
  const float min_update_squared = (float)(0.5 * 0.5);
  float update[3] = {0, 0, 0};
  int it_count = 0;
  for (int iter = 0; iter < n_iter; ++iter) {
    int u_r = (int)floor(u);
    int v_r = (int)floor(v);
    if (u_r < halfpatch || v_r < halfpatch || u_r >= cols - halfpatch || v_r >= rows - halfpatch)
      break;
    if (isnan(u) || isnan(v)) { if (iters_done) *iters_done = it_count; return 0; }
    ++it_count;
    float subpix_x = u - u_r;
    float subpix_y = v - v_r;
    float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
    float wTR = (float)(subpix_x * (1.0 - subpix_y));
    float wBL = (float)((1.0 - subpix_x) * subpix_y);
    float wBR = subpix_x * subpix_y;
    float Jres[3] = {0, 0, 0};
    int r = 0;
    for (int y = 0; y < patch_size; ++y) {
      const uint8_t* it = cur_img + (v_r + y - halfpatch) * cur_step + u_r - halfpatch;
      for (int x = 0; x < patch_size; ++x, ++it, ++r) {
        float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
        float res = search_pixel - ref_patch[r] + mean_diff;
        Jres[0] -= res * dxs[r];
        Jres[1] -= res * dys[r];
        Jres[2] -= res;
      }
    }
    for (int a = 0; a < 3; ++a)
      update[a] = Hinv[a * 3 + 0] * Jres[0] + (Hinv[a * 3 + 1] * Jres[1] + Hinv[a * 3 + 2] * Jres[2]);   /* Eigen 3-term redux order */
    u += update[0];
    v += update[1];
    mean_diff += update[2];
    if (update[0] * update[0] + update[1] * update[1] < min_update_squared) {
      converged = 1;
      break;
    }
  }
  px_inout[0] = u; px_inout[1] = v;
  if (iters_done) *iters_done = it_count;
  return converged;
}

/* S/feature_alignment.cpp:35-152 */
int svo_orc_align1d(const uint8_t* cur_img, int cols, int rows, int cur_step, const float dir[2],
                    const uint8_t* ref_patch_with_border, const uint8_t* ref_patch, int n_iter,
                    double px_inout[2], double* h_inv, int* iters_done) {
  const int halfpatch = 4, patch_size = 8;
  int converged = 0;
  float dv[64];
  float H[4] = {0, 0, 0, 0};
  const int ref_step = patch_size + 2;
  int k = 0;
  for (int y = 0; y < patch_size; ++y) {
    const uint8_t* it = ref_patch_with_border + (y + 1) * ref_step + 1;
    for (int x = 0; x < patch_size; ++x, ++it, ++k) {
      float J[2];
      J[0] = (float)(0.5 * (dir[0] * (it[1] - it[-1]) + dir[1] * (it[ref_step] - it[-ref_step])));
      J[1] = 1;
      dv[k] = J[0];
      H[0] += J[0] * J[0]; H[1] += J[0] * J[1]; H[2] += J[1] * J[0]; H[3] += J[1] * J[1];
    }
  }
  *h_inv = 1.0 / H[0] * patch_size * patch_size;
  /* Eigen 2x2 inverse: invdet = 1/det; adjugate * invdet */
  float det = H[0] * H[3] - H[2] * H[1];
  float invdet = 1.0f / det;
  float Hinv[4] = {H[3] * invdet, -H[1] * invdet, -H[2] * invdet, H[0] * invdet};
  float mean_diff = 0;
  float u = (float)px_inout[0];
  float v = (float)px_inout[1];
  const float min_update_squared = (float)(0.03 * 0.03);
  float chi2 = 0;
  float update[2] = {0, 0};
  int it_count = 0;
  for (int iter = 0; iter < n_iter; ++iter) {
    int u_r = (int)floor(u);
    int v_r = (int)floor(v);
    if (u_r < halfpatch || v_r < halfpatch || u_r >= cols - halfpatch || v_r >= rows - halfpatch)
      break;
    if (isnan(u) || isnan(v)) { if (iters_done) *iters_done = it_count; return 0; }
    ++it_count;
    float subpix_x = u - u_r;
    float subpix_y = v - v_r;
    float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
    float wTR = (float)(subpix_x * (1.0 - subpix_y));
    float wBL = (float)((1.0 - subpix_x) * subpix_y);
    float wBR = subpix_x * subpix_y;
    float new_chi2 = 0.0f;
    float Jres[2] = {0, 0};
    int r = 0;
    for (int y = 0; y < patch_size; ++y) {
      const uint8_t* it = cur_img + (v_r + y - halfpatch) * cur_step + u_r - halfpatch;
      for (int x = 0; x < patch_size; ++x, ++it, ++r) {
        float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
        float res = search_pixel - ref_patch[r] + mean_diff;
        Jres[0] -= res * dv[r];
        Jres[1] -= res;
        new_chi2 += res * res;
      }
    }
    if (iter > 0 && new_chi2 > chi2) {
      u -= update[0];
      v -= update[1];
      break;
    }
    chi2 = new_chi2;
    update[0] = Hinv[0] * Jres[0] + Hinv[1] * Jres[1];
    update[1] = Hinv[2] * Jres[0] + Hinv[3] * Jres[1];
    u += update[0] * dir[0];
    v += update[0] * dir[1];
    mean_diff += update[1];
    if (update[0] * update[0] + update[1] * update[1] < min_update_squared) {
      converged = 1;
      break;
    }
  }
  px_inout[0] = u; px_inout[1] = v;
  if (iters_done) *iters_done = it_count;
  return converged;
}

/* ------------------------------------------------------------------------ */
/* matcher pieces                                                            */
/* ------------------------------------------------------------------------ */

/* S/matcher.cpp:36-60 */
void svo_orc_get_warp_matrix_affine(
    const svo_orc_camera* cam_ref, const svo_orc_camera* cam_cur, const double px_ref[2],
    const double f_ref[3], double depth_ref, const double T_cur_ref[7], int level_ref,
    double A[4]) {
  const int halfpatch_size = 5;
  const double xyz_ref[3] = {f_ref[0] * depth_ref, f_ref[1] * depth_ref, f_ref[2] * depth_ref};
  double du[3], dv[3];
  const double off = (double)halfpatch_size * (1 << level_ref);
  svo_orc_cam2world(cam_ref, px_ref[0] + off, px_ref[1] + 0.0 * (1 << level_ref), du);
  svo_orc_cam2world(cam_ref, px_ref[0] + 0.0 * (1 << level_ref), px_ref[1] + off, dv);
  double su = xyz_ref[2] / du[2];
  du[0] *= su; du[1] *= su; du[2] *= su;
  double sv = xyz_ref[2] / dv[2];
  dv[0] *= sv; dv[1] *= sv; dv[2] *= sv;
  double p[3], px_cur[2], px_du[2], px_dv[2];
  svo_orc_se3_act(T_cur_ref, xyz_ref, p); svo_orc_world2cam(cam_cur, p, px_cur);
  svo_orc_se3_act(T_cur_ref, du, p);      svo_orc_world2cam(cam_cur, p, px_du);
  svo_orc_se3_act(T_cur_ref, dv, p);      svo_orc_world2cam(cam_cur, p, px_dv);
  A[0] = (px_du[0] - px_cur[0]) / halfpatch_size;   /* col 0 */
  A[2] = (px_du[1] - px_cur[1]) / halfpatch_size;
  A[1] = (px_dv[0] - px_cur[0]) / halfpatch_size;   /* col 1 */
  A[3] = (px_dv[1] - px_cur[1]) / halfpatch_size;
}

/* S/matcher.cpp:65-78 */
int svo_orc_get_best_search_level(const double A[4], int max_level) {
  int search_level = 0;
  double D = A[0] * A[3] - A[2] * A[1];
  while (D > 3.0 && search_level < max_level) {
    search_level += 1;
    D *= 0.25;
  }
  return search_level;
}

/* S/matcher.cpp:83-116 */
int svo_orc_warp_affine(const double A[4], const uint8_t* img_ref, int cols, int rows,
                        const double px_ref[2], int level_ref, int search_level,
                        int halfpatch_size, uint8_t* patch) {
  const int patch_size = halfpatch_size * 2;
  /* Eigen 2x2 inverse (double), then cast<float> */
  const double det = A[0] * A[3] - A[2] * A[1];
  const double invdet = 1.0 / det;
  const float a00 = (float)(A[3] * invdet), a01 = (float)(-A[1] * invdet);
  const float a10 = (float)(-A[2] * invdet), a11 = (float)(A[0] * invdet);
  if (isnan(a00)) return 0;
  const float prx = (float)px_ref[0] / (1 << level_ref);
  const float pry = (float)px_ref[1] / (1 << level_ref);
  uint8_t* pp = patch;
  for (int y = 0; y < patch_size; ++y) {
    for (int x = 0; x < patch_size; ++x, ++pp) {
      float ppx = (float)(x - halfpatch_size);
      float ppy = (float)(y - halfpatch_size);
      ppx *= (1 << search_level);
      ppy *= (1 << search_level);
      const float qx = (a00 * ppx + a01 * ppy) + prx;
      const float qy = (a10 * ppx + a11 * ppy) + pry;
      if (qx < 0 || qy < 0 || qx >= cols - 1 || qy >= rows - 1)
        *pp = 0;
      else
        *pp = (uint8_t)svo_orc_interpolate_8u(img_ref, cols, qx, qy);
    }
  }
  return 1;
}

/* S/matcher.cpp:138-147 */
void svo_orc_patch_from_border(const uint8_t* pwb, uint8_t* patch) {
  for (int y = 1; y < 9; ++y)
    for (int x = 0; x < 8; ++x) patch[(y - 1) * 8 + x] = pwb[y * 10 + 1 + x];
}

/* I/patch_score.h:40-63,199-219 (integer exact; SSE2 path gives the same sums) */
int svo_orc_zmssd(const uint8_t* ref_patch, const uint8_t* cur, int stride) {
  uint32_t sumA = 0, sumAA = 0, sumB = 0, sumBB = 0, sumAB = 0;
  for (int r = 0; r < 64; ++r) { uint32_t n = ref_patch[r]; sumA += n; sumAA += n * n; }
  for (int y = 0, r = 0; y < 8; ++y) {
    const uint8_t* p = cur + y * stride;
    for (int x = 0; x < 8; ++x, ++r) {
      const uint32_t c = p[x];
      sumB += c; sumBB += c * c; sumAB += c * ref_patch[r];
    }
  }
  const int sA = (int)sumA, sAA = (int)sumAA, sB = (int)sumB, sBB = (int)sumBB, sAB = (int)sumAB;
  return sAA - 2 * sAB + sBB - (sA * sA - 2 * sA * sB + sB * sB) / 64;
}

/* S/matcher.cpp:123-136 */
int svo_orc_depth_from_triangulation(const double T_search_ref[7], const double f_ref[3],
                                     const double f_cur[3], double* depth) {
  double R[9], t[3] = {T_search_ref[0], T_search_ref[1], T_search_ref[2]};
  svo_orc_se3_rotation_matrix(T_search_ref, R);
  double a0[3], a1[3] = {f_cur[0], f_cur[1], f_cur[2]};
  for (int i = 0; i < 3; ++i) a0[i] = (R[3 * i] * f_ref[0] + R[3 * i + 1] * f_ref[1]) + R[3 * i + 2] * f_ref[2];
  const double m00 = (a0[0] * a0[0] + a0[1] * a0[1]) + a0[2] * a0[2];
  const double m01 = (a0[0] * a1[0] + a0[1] * a1[1]) + a0[2] * a1[2];
  const double m10 = m01;
  const double m11 = (a1[0] * a1[0] + a1[1] * a1[1]) + a1[2] * a1[2];
  const double det = m00 * m11 - m10 * m01;
  if (det < 0.000001) return 0;
  const double invdet = 1.0 / det;
  /* -(AtA^-1) */
  const double n00 = -(m11 * invdet), n01 = -(-m01 * invdet);
  /* row 0 of (-(AtA^-1)) * A^T, then dot t */
  double r0[3];
  for (int k = 0; k < 3; ++k) r0[k] = n00 * a0[k] + n01 * a1[k];
  const double d0 = (r0[0] * t[0] + r0[1] * t[1]) + r0[2] * t[2];
  *depth = fabs(d0);
  return 1;
}

/* S/matcher.cpp:207-355 */
int svo_orc_find_epipolar_match_direct(
    const svo_orc_camera* cam, const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
    const double T_cur_ref[7], const double px_ref[2], const double f_ref[3], int level_ref,
    double d_estimate, double d_min, double d_max, int n_pyr_levels, int align_max_iter,
    int max_epi_search_steps, svo_orc_epi_result* o) {
  memset(o, 0, sizeof(*o));
  const int halfpatch = 4, patch_size = 8;
  const int zmssd_threshold = 2000 * 64;
  int zmssd_best = zmssd_threshold;
  double uv_best[2] = {0, 0};

  double pa[3], pb[3], tmp[3];
  tmp[0] = f_ref[0] * d_min; tmp[1] = f_ref[1] * d_min; tmp[2] = f_ref[2] * d_min;
  svo_orc_se3_act(T_cur_ref, tmp, pa);
  tmp[0] = f_ref[0] * d_max; tmp[1] = f_ref[1] * d_max; tmp[2] = f_ref[2] * d_max;
  svo_orc_se3_act(T_cur_ref, tmp, pb);
  const double A[2] = {pa[0] / pa[2], pa[1] / pa[2]};
  const double B[2] = {pb[0] / pb[2], pb[1] / pb[2]};
  const double epi_dir[2] = {A[0] - B[0], A[1] - B[1]};

  double Acr[4];
  svo_orc_get_warp_matrix_affine(cam, cam, px_ref, f_ref, d_estimate, T_cur_ref, level_ref, Acr);
  const int search_level = svo_orc_get_best_search_level(Acr, n_pyr_levels - 1);
  o->search_level = search_level;

  double px_A[2], px_B[2];
  svo_orc_world2cam_uv(cam, A, px_A);
  svo_orc_world2cam_uv(cam, B, px_B);
  {
    const double ex = px_A[0] - px_B[0], ey = px_A[1] - px_B[1];
    o->epi_length = sqrt(ex * ex + ey * ey) / (1 << search_level);
  }
  const double epi_length = o->epi_length;

  uint8_t* pwb = o->patch_with_border;
  uint8_t patch[64];
  svo_orc_warp_affine(Acr, ref_pyr[level_ref], cam->width >> level_ref, cam->height >> level_ref,
                      px_ref, level_ref, search_level, halfpatch + 1, pwb);
  svo_orc_patch_from_border(pwb, patch);

  const int ccols = cam->width >> search_level, crows = cam->height >> search_level;
  const uint8_t* cur_img = cur_pyr[search_level];

  if (epi_length < 2.0) {
    o->path = 0;
    o->px_cur[0] = (px_A[0] + px_B[0]) / 2.0;
    o->px_cur[1] = (px_A[1] + px_B[1]) / 2.0;
    double px_scaled[2] = {o->px_cur[0] / (1 << search_level), o->px_cur[1] / (1 << search_level)};
    int res = svo_orc_align2d(cur_img, ccols, crows, ccols, pwb, patch, align_max_iter, px_scaled,
                              &o->n_align_iters);
    if (res) {
      o->px_cur[0] = px_scaled[0] * (1 << search_level);
      o->px_cur[1] = px_scaled[1] * (1 << search_level);
      double fc[3];
      svo_orc_cam2world(cam, o->px_cur[0], o->px_cur[1], fc);
      if (svo_orc_depth_from_triangulation(T_cur_ref, f_ref, fc, &o->depth)) { o->ok = 1; return 1; }
    }
    return 0;
  }

  size_t n_steps = (size_t)(epi_length / 0.7);
  const double step[2] = {epi_dir[0] / n_steps, epi_dir[1] / n_steps};
  if (n_steps > (size_t)max_epi_search_steps) { o->path = 2; return 0; }
  o->path = 1;

  double uv[2] = {B[0] - step[0], B[1] - step[1]};
  int last_x = 0, last_y = 0;
  ++n_steps;
  for (size_t i = 0; i < n_steps; ++i, uv[0] += step[0], uv[1] += step[1]) {
    double px[2];
    svo_orc_world2cam_uv(cam, uv, px);
    const int pxi_x = (int)(px[0] / (1 << search_level) + 0.5);
    const int pxi_y = (int)(px[1] / (1 << search_level) + 0.5);
    if (pxi_x == last_x && pxi_y == last_y) continue;
    last_x = pxi_x; last_y = pxi_y;
    if (!is_in_frame_level(cam, pxi_x, pxi_y, patch_size, search_level)) continue;
    const uint8_t* cur_patch_ptr = cur_img + (pxi_y - halfpatch) * ccols + (pxi_x - halfpatch);
    const int zmssd = svo_orc_zmssd(patch, cur_patch_ptr, ccols);
    o->n_zmssd++;
    if (zmssd < zmssd_best) { zmssd_best = zmssd; uv_best[0] = uv[0]; uv_best[1] = uv[1]; }
  }

  if (zmssd_best < zmssd_threshold) {
    svo_orc_world2cam_uv(cam, uv_best, o->px_cur);
    double px_scaled[2] = {o->px_cur[0] / (1 << search_level), o->px_cur[1] / (1 << search_level)};
    int res = svo_orc_align2d(cur_img, ccols, crows, ccols, pwb, patch, align_max_iter, px_scaled,
                              &o->n_align_iters);
    if (res) {
      o->px_cur[0] = px_scaled[0] * (1 << search_level);
      o->px_cur[1] = px_scaled[1] * (1 << search_level);
      double fc[3];
      svo_orc_cam2world(cam, o->px_cur[0], o->px_cur[1], fc);
      if (svo_orc_depth_from_triangulation(T_cur_ref, f_ref, fc, &o->depth)) { o->ok = 1; return 1; }
    }
    return 0;
  }
  return 0;
}

/* S/matcher.cpp:156-202 without Point::getCloseViewObs (host bookkeeping: the caller passes the
 * reference feature it selected).  Returns the success flag; px_cur in/out (level-0 pixels). */
int svo_orc_find_match_direct(const svo_orc_camera* cam, const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
                              const double T_ref_w[7], const double T_cur_w[7], const double px_ref[2],
                              const double f_ref[3], int level_ref, const double pt_pos[3], int edgelet,
                              const double grad[2], int n_pyr_levels, int align_max_iter, double px_cur[2],
                              int* search_level_out) {
  /* isInFrame(px.cast<int>()/(1<<level), halfpatch_size_+2, level) (:164-166) */
  const int ox = (int)px_ref[0] / (1 << level_ref), oy = (int)px_ref[1] / (1 << level_ref);
  if (!is_in_frame_level(cam, ox, oy, 4 + 2, level_ref)) return 0;
  double T_ref_inv[7], T_cur_ref[7];
  svo_orc_se3_inverse(T_ref_w, T_ref_inv);
  svo_orc_se3_mul(T_cur_w, T_ref_inv, T_cur_ref);
  const double dx = T_ref_inv[0] - pt_pos[0], dy = T_ref_inv[1] - pt_pos[1], dz = T_ref_inv[2] - pt_pos[2];
  const double depth = sqrt(dx * dx + dy * dy + dz * dz);          /* (ref_frame.pos() - pt.pos_).norm() */
  double A[4];
  svo_orc_get_warp_matrix_affine(cam, cam, px_ref, f_ref, depth, T_cur_ref, level_ref, A);
  const int search_level = svo_orc_get_best_search_level(A, n_pyr_levels - 1);
  if (search_level_out) *search_level_out = search_level;
  uint8_t pwb[100], patch[64];
  memset(pwb, 0, sizeof(pwb));
  svo_orc_warp_affine(A, ref_pyr[level_ref], cam->width >> level_ref, cam->height >> level_ref, px_ref, level_ref,
                      search_level, 5, pwb);
  svo_orc_patch_from_border(pwb, patch);
  double px_scaled[2] = {px_cur[0] / (1 << search_level), px_cur[1] / (1 << search_level)};
  const int ccols = cam->width >> search_level, crows = cam->height >> search_level;
  int success;
  if (edgelet) {
    /* dir_cur = (A_cur_ref * grad).normalized(), cast to float (:187-191) */
    double d0 = A[0] * grad[0] + A[1] * grad[1], d1 = A[2] * grad[0] + A[3] * grad[1];
    const double n2 = d0 * d0 + d1 * d1;
    if (n2 > 0.0) { const double nn = sqrt(n2); d0 = d0 / nn; d1 = d1 / nn; }
    const float dir[2] = {(float)d0, (float)d1};
    double h_inv;
    success = svo_orc_align1d(cur_pyr[search_level], ccols, crows, ccols, dir, pwb, patch, align_max_iter, px_scaled,
                              &h_inv, NULL);
  } else {
    success = svo_orc_align2d(cur_pyr[search_level], ccols, crows, ccols, pwb, patch, align_max_iter, px_scaled, NULL);
  }
  px_cur[0] = px_scaled[0] * (1 << search_level);
  px_cur[1] = px_scaled[1] * (1 << search_level);
  return success;
}

/* ------------------------------------------------------------------------ */
/* depth filter                                                              */
/* ------------------------------------------------------------------------ */

/* S/depth_filter.cpp:36-45 */
void svo_orc_seed_init(svo_orc_seed* s, float depth_mean, float depth_min) {
  s->a = 10; s->b = 10;
  s->mu = (float)(1.0 / depth_mean);
  s->z_range = (float)(1.0 / depth_min);
  s->sigma2 = s->z_range * s->z_range / 36;
}

/* S/depth_filter.cpp:359-363.  Quirk kept: the constant is sqrt(2), not sqrt(2 pi). */
static double normal_pdf(double x, double mean, double std_dev) {
  static const double SQRT_2_PI = 1.41421356237309505;
  double exponent = -0.5 * pow((x - mean) / std_dev, 2);
  return (1 / (std_dev * SQRT_2_PI)) * exp(exponent);
}

/* S/depth_filter.cpp:368-391; promotions follow each C++ statement. */
void svo_orc_update_seed(const float x, const float tau2, svo_orc_seed* seed) {
  float norm_scale = sqrtf(seed->sigma2 + tau2);
  if (isnan(norm_scale)) return;
  float s2 = (float)(1. / (1. / seed->sigma2 + 1. / tau2));
  float m = s2 * (seed->mu / seed->sigma2 + x / tau2);
  float C1 = (float)(seed->a / (seed->a + seed->b) * normal_pdf(x, seed->mu, norm_scale));
  float C2 = (float)(seed->b / (seed->a + seed->b) * 1. / seed->z_range);
  float normalization_constant = C1 + C2;
  C1 /= normalization_constant;
  C2 /= normalization_constant;
  float f = (float)(C1 * (seed->a + 1.) / (seed->a + seed->b + 1.) + C2 * seed->a / (seed->a + seed->b + 1.));
  float e = (float)(C1 * (seed->a + 1.) * (seed->a + 2.) / ((seed->a + seed->b + 1.) * (seed->a + seed->b + 2.)) +
                    C2 * seed->a * (seed->a + 1.0f) / ((seed->a + seed->b + 1.0f) * (seed->a + seed->b + 2.0f)));
  float mu_new = C1 * m + C2 * seed->mu;
  seed->sigma2 = C1 * (s2 + m * m) + C2 * (seed->sigma2 + seed->mu * seed->mu) - mu_new * mu_new;
  seed->mu = mu_new;
  seed->a = (e - f) / (f - e / f);
  seed->b = seed->a * (1.0f - f) / f;
}

/* S/depth_filter.cpp:396-416; PI = 3.14159265 (I/global.h:92) */
double svo_orc_compute_tau(const double T_ref_cur[7], const double f[3], double z,
                           double px_error_angle) {
  const double PI_SVO = 3.14159265;
  const double t[3] = {T_ref_cur[0], T_ref_cur[1], T_ref_cur[2]};
  const double a[3] = {f[0] * z - t[0], f[1] * z - t[1], f[2] * z - t[2]};
  double t_norm = sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
  double a_norm = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
  double alpha = acos(((f[0] * t[0] + f[1] * t[1]) + f[2] * t[2]) / t_norm);
  double beta = acos(((a[0] * -t[0] + a[1] * -t[1]) + a[2] * -t[2]) / (t_norm * a_norm));
  double beta_plus = beta + px_error_angle;
  double gamma_plus = PI_SVO - alpha - beta_plus;
  double z_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
  return z_plus - z;
}

/* S/depth_filter.cpp:237-341 (per-seed body; list/ageing/halt bookkeeping is host-side) */
int svo_orc_update_seeds_ex(
    const svo_orc_camera* cam, const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
    const double T_ref_w[7], const double T_cur_w[7], int n_seeds, const double* px,
    const double* f, const int* level, float* a, float* b, float* mu, const float* z_range,
    float* sigma2, int n_pyr_levels, int align_max_iter, int max_epi_search_steps,
    double convergence_sigma2_thresh, int* status, double* z_out, double* xyz_world,
    int* n_zmssd, int* n_align_iters, double* px_cur_out, int* search_level_out) {
  const double focal_length = fabs(cam->fx);
  const double px_noise = 1.0;
  const double px_error_angle = atan(px_noise / (2.0 * focal_length)) * 2.0;
  double T_cur_inv[7], T_ref_cur[7], T_cur_ref_chk[7], T_ref_inv[7], T_cur_ref[7];
  svo_orc_se3_inverse(T_cur_w, T_cur_inv);
  svo_orc_se3_mul(T_ref_w, T_cur_inv, T_ref_cur);          /* :264 */
  svo_orc_se3_inverse(T_ref_cur, T_cur_ref_chk);           /* T_ref_cur.inverse(), :265 */
  svo_orc_se3_inverse(T_ref_w, T_ref_inv);
  svo_orc_se3_mul(T_cur_w, T_ref_inv, T_cur_ref);          /* matcher.cpp:216 */
  for (int i = 0; i < n_seeds; ++i) {
    const double* fi = f + 3 * i;
    if (z_out) z_out[i] = 0.0;
    if (n_zmssd) n_zmssd[i] = 0;
    if (n_align_iters) n_align_iters[i] = 0;
    if (px_cur_out) { px_cur_out[2 * i] = NAN; px_cur_out[2 * i + 1] = NAN; }
    if (search_level_out) search_level_out[i] = -1;
    const double inv_mu = 1.0 / mu[i];
    const double pf[3] = {inv_mu * fi[0], inv_mu * fi[1], inv_mu * fi[2]};
    double xyz_f[3];
    svo_orc_se3_act(T_cur_ref_chk, pf, xyz_f);
    if (xyz_f[2] < 0.0) { status[i] = SVO_SEED_BEHIND; continue; }
    double pc[2];
    svo_orc_world2cam(cam, xyz_f, pc);
    const int ox = (int)pc[0], oy = (int)pc[1];
    if (!(ox >= 0 && ox < cam->width && oy >= 0 && oy < cam->height)) {
      status[i] = SVO_SEED_NOT_IN_FRAME; continue;
    }
    float z_inv_min = mu[i] + sqrtf(sigma2[i]);
    const float z_inv_lo = mu[i] - sqrtf(sigma2[i]);
    const float z_inv_max = (z_inv_lo < 0.00000001f) ? 0.00000001f : z_inv_lo;   /* std::max */
    svo_orc_epi_result er;
    int ok = svo_orc_find_epipolar_match_direct(
        cam, ref_pyr, cur_pyr, T_cur_ref, px + 2 * i, fi, level[i], 1.0 / mu[i],
        1.0 / z_inv_min, 1.0 / z_inv_max, n_pyr_levels, align_max_iter, max_epi_search_steps, &er);
    if (n_zmssd) n_zmssd[i] = er.n_zmssd;
    if (n_align_iters) n_align_iters[i] = er.n_align_iters;
    if (search_level_out) search_level_out[i] = er.search_level;       /* matcher_.search_level_ */
    if (!ok) { b[i] += 1.0f; status[i] = SVO_SEED_NO_MATCH; continue; }
    if (px_cur_out) { px_cur_out[2 * i] = er.px_cur[0]; px_cur_out[2 * i + 1] = er.px_cur[1]; }   /* matcher_.px_cur_, :302-306 */
    const double z = er.depth;
    if (z_out) z_out[i] = z;
    double tau = svo_orc_compute_tau(T_ref_cur, fi, z, px_error_angle);
    double zmt = z - tau;
    double tau_inverse = 0.5 * (1.0 / (0.0000001 < zmt ? zmt : 0.0000001) - 1.0 / (z + tau));
    svo_orc_seed sd = {a[i], b[i], mu[i], z_range[i], sigma2[i]};
    svo_orc_update_seed((float)(1. / z), (float)(tau_inverse * tau_inverse), &sd);
    a[i] = sd.a; b[i] = sd.b; mu[i] = sd.mu; sigma2[i] = sd.sigma2;
    if ((double)sqrtf(sd.sigma2) < sd.z_range / convergence_sigma2_thresh) {
      status[i] = SVO_SEED_CONVERGED;
      if (xyz_world) {
        const double im = 1.0 / sd.mu;
        const double pfw[3] = {fi[0] * im, fi[1] * im, fi[2] * im};
        svo_orc_se3_act(T_ref_inv, pfw, xyz_world + 3 * i);
      }
    } else if (isnan(z_inv_min)) {
      status[i] = SVO_SEED_NAN;
    } else {
      status[i] = SVO_SEED_UPDATED;
    }
  }
  return 0;
}

int svo_orc_update_seeds(
    const svo_orc_camera* cam, const uint8_t* const* ref_pyr, const uint8_t* const* cur_pyr,
    const double T_ref_w[7], const double T_cur_w[7], int n_seeds, const double* px,
    const double* f, const int* level, float* a, float* b, float* mu, const float* z_range,
    float* sigma2, int n_pyr_levels, int align_max_iter, int max_epi_search_steps,
    double convergence_sigma2_thresh, int* status, double* z_out, double* xyz_world,
    int* n_zmssd, int* n_align_iters) {
  return svo_orc_update_seeds_ex(cam, ref_pyr, cur_pyr, T_ref_w, T_cur_w, n_seeds, px, f, level, a, b, mu, z_range, sigma2,
                                 n_pyr_levels, align_max_iter, max_epi_search_steps, convergence_sigma2_thresh, status,
                                 z_out, xyz_world, n_zmssd, n_align_iters, NULL, NULL);
}

/* ------------------------------------------------------------------------ */
/* next rows f-4: motion-only pose refinement and structure refinement       */
/* ------------------------------------------------------------------------ */

/* k-th smallest (k = floor(n/2)) as vk::getMedian does with nth_element (I/math_utils.h:124-131):
 * the value does not depend on the selection algorithm. */
static int cmp_float(const void* a, const void* b) {
  float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}
static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}
float svo_orc_median_f(const float* v, int n) {
  float* t = (float*)malloc(sizeof(float) * (size_t)n);
  memcpy(t, v, sizeof(float) * (size_t)n);
  qsort(t, (size_t)n, sizeof(float), cmp_float);
  float r = t[n / 2];
  free(t);
  return r;
}
static double median_d(const double* v, int n) {
  double* t = (double*)malloc(sizeof(double) * (size_t)n);
  memcpy(t, v, sizeof(double) * (size_t)n);
  qsort(t, (size_t)n, sizeof(double), cmp_double);
  double r = t[n / 2];
  free(t);
  return r;
}

/* vk::robust_cost::TukeyWeightFunction::value with DEFAULT_B = 8.6851f (S/robust_cost.cpp:87-106) */
float svo_orc_tukey_weight(float x) {
  const float b = 8.6851f;
  const float b_square = b * b;
  const float x_square = x * x;
  if (x_square <= b_square) {
    const float tmp = 1.0f - x_square / b_square;
    return tmp * tmp;
  }
  return 0.0f;
}

/* Eigen 3.4.0 Matrix<double,6,6>::inverse() = partialPivLu().inverse(): unblocked right-looking LU with
 * partial pivoting (LU/PartialPivLU.h:379-425), then P, unit-lower and upper substitution on the identity.
 * Eigen runs the substitutions through its blocked matrix kernel, so only a tolerance is claimed. */
void svo_orc_inverse6(const double Ain[36], double out[36]) {
  enum { N = 6 };
  double lu[N][N];
  int piv[N];
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) lu[i][j] = Ain[i * N + j];
  for (int k = 0; k < N; ++k) {
    int big = k;
    double best = fabs(lu[k][k]);
    for (int i = k + 1; i < N; ++i) if (fabs(lu[i][k]) > best) { best = fabs(lu[i][k]); big = i; }
    piv[k] = big;
    if (best != 0.0) {
      if (big != k) for (int j = 0; j < N; ++j) { double t = lu[k][j]; lu[k][j] = lu[big][j]; lu[big][j] = t; }
      for (int i = k + 1; i < N; ++i) lu[i][k] /= lu[k][k];
    }
    for (int i = k + 1; i < N; ++i)
      for (int j = k + 1; j < N; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
  }
  for (int c = 0; c < N; ++c) {
    double d[N];
    for (int i = 0; i < N; ++i) d[i] = (i == c) ? 1.0 : 0.0;
    for (int k = 0; k < N; ++k) if (piv[k] != k) { double t = d[k]; d[k] = d[piv[k]]; d[piv[k]] = t; }
    for (int i = 0; i < N; ++i) { double s = d[i]; for (int j = 0; j < i; ++j) s -= lu[i][j] * d[j]; d[i] = s; }
    for (int i = N - 1; i >= 0; --i) {
      double s = d[i];
      for (int j = i + 1; j < N; ++j) s -= lu[i][j] * d[j];
      d[i] = s / lu[i][i];
    }
    for (int i = 0; i < N; ++i) out[i * N + c] = d[i];
  }
}

/* pose_optimizer::optimizeGaussNewton, S/pose_optimizer.cpp:31-181 (verbose output left out).
 * em = frame->cam_->errorMultiplier2().  has_point[i] != 0 <=> (*it)->point != NULL; cleared for the
 * observations the final outlier test removes (:150-157). */
int svo_orc_pose_optimize(double em, double reproj_thresh, int n_iter, const double T_f_w_in[7], int n,
                          const double* f, const double* pos, const int* level, uint8_t* has_point,
                          svo_orc_pose_opt_result* out) {
  memset(out, 0, sizeof(*out));
  double T[7], T_old[7];
  memcpy(T, T_f_w_in, sizeof(T));
  memcpy(T_old, T_f_w_in, sizeof(T));                                   /* :45 */
  memcpy(out->T_f_w, T, sizeof(T));
  double chi2 = 0.0;
  double A[36], b[6];
  /* :51-60 scale of the error for robust estimation */
  float* errors = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  int n_err = 0;
  for (int i = 0; i < n; ++i) {
    if (!has_point[i]) continue;
    double xyz[3];
    svo_orc_se3_act(T, pos + 3 * i, xyz);
    const double* fi = f + 3 * i;
    double e0 = fi[0] / fi[2] - xyz[0] / xyz[2];
    double e1 = fi[1] / fi[2] - xyz[1] / xyz[2];
    const double s = 1.0 / (1 << level[i]);
    e0 *= s; e1 *= s;
    errors[n_err++] = (float)sqrt(e0 * e0 + e1 * e1);
  }
  if (n_err == 0) { free(errors); return 0; }                              /* :61-62 */
  out->ran = 1;
  double estimated_scale = (double)(1.48f * svo_orc_median_f(errors, n_err));   /* MADScaleEstimator, robust_cost.cpp:67-74 */
  free(errors);
  size_t num_obs = (size_t)n_err;
  double* chi2_init = (double*)malloc(sizeof(double) * num_obs);
  double* chi2_final = (double*)malloc(sizeof(double) * num_obs);
  int n_init = 0, n_final = 0;
  double scale = estimated_scale;
  memset(A, 0, sizeof(A));
  for (int iter = 0; iter < n_iter; ++iter) {
    if (iter == 5) scale = 0.85 / em;                                      /* :74-75 */
    memset(b, 0, sizeof(b));
    memset(A, 0, sizeof(A));
    double new_chi2 = 0.0;
    for (int i = 0; i < n; ++i) {
      if (!has_point[i]) continue;
      double xyz[3], J[12];
      svo_orc_se3_act(T, pos + 3 * i, xyz);
      svo_orc_jacobian_xyz2uv(xyz, J);
      const double* fi = f + 3 * i;
      double e0 = fi[0] / fi[2] - xyz[0] / xyz[2];
      double e1 = fi[1] / fi[2] - xyz[1] / xyz[2];
      const double sqrt_inv_cov = 1.0 / (1 << level[i]);
      e0 *= sqrt_inv_cov; e1 *= sqrt_inv_cov;
      const double sq = e0 * e0 + e1 * e1;
      if (iter == 0) chi2_init[n_init++] = sq;
      for (int k = 0; k < 12; ++k) J[k] *= sqrt_inv_cov;
      const double weight = (double)svo_orc_tukey_weight((float)(sqrt(sq) / scale));
      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) A[r * 6 + c] += (J[r] * J[c] + J[6 + r] * J[6 + c]) * weight;   /* J^T J w */
        b[r] -= (J[r] * e0 + J[6 + r] * e1) * weight;                                                /* J^T e w */
      }
      new_chi2 += sq * weight;
    }
    double dT[6];
    svo_orc_ldlt6_solve(A, b, dT);
    out->n_iter_done = iter + 1;
    if ((iter > 0 && new_chi2 > chi2 * 1.2) || isnan(dT[0])) {           /* :106-116 */
      memcpy(T, T_old, sizeof(T));
      break;
    }
    double E[7], Tn[7];
    svo_orc_se3_exp(dT, E);
    svo_orc_se3_mul(E, T, Tn);                                             /* SE3::exp(dT) * T_f_w (:120) */
    memcpy(T_old, T, sizeof(T));
    memcpy(T, Tn, sizeof(T));
    chi2 = new_chi2;
    double mx = -1;
    for (int k = 0; k < 6; ++k) { double a = fabs(dT[k]); if (a > mx) mx = a; }
    if (mx <= 0.0000000001) break;                                         /* EPS, global.h:91 */
  }
  /* :141 covariance = (A em^2)^-1 */
  {
    double As[36];
    const double em2 = pow(em, 2);
    for (int k = 0; k < 36; ++k) As[k] = A[k] * em2;
    svo_orc_inverse6(As, out->Cov);
  }
  /* :144-159 remove measurements with too large reprojection error */
  const double thresh = reproj_thresh / em;
  int n_deleted = 0;
  for (int i = 0; i < n; ++i) {
    if (!has_point[i]) continue;
    double xyz[3];
    svo_orc_se3_act(T, pos + 3 * i, xyz);
    const double* fi = f + 3 * i;
    double e0 = fi[0] / fi[2] - xyz[0] / xyz[2];
    double e1 = fi[1] / fi[2] - xyz[1] / xyz[2];
    const double s = 1.0 / (1 << level[i]);
    e0 *= s; e1 *= s;
    const double sq = e0 * e0 + e1 * e1;
    chi2_final[n_final++] = sq;
    if (sqrt(sq) > thresh) { has_point[i] = 0; ++n_deleted; }
  }
  out->error_init = n_init ? sqrt(median_d(chi2_init, n_init)) * em : 0.0;
  out->error_final = n_final ? sqrt(median_d(chi2_final, n_final)) * em : 0.0;
  out->estimated_scale = estimated_scale * em;
  out->num_obs = num_obs - (size_t)n_deleted;
  out->n_deleted = n_deleted;
  memcpy(out->T_f_w, T, sizeof(T));
  free(chi2_init); free(chi2_final);
  return 0;
}

/* Point::jacobian_xyz2uv, I/point.h:83-97: -[1/z 0 -x/z^2; 0 1/z -y/z^2] * R_f_w.
 * Eigen evaluates this 2x3 product a column (2 doubles = one SSE2 packet) at a time, accumulating over the
 * inner index: (a0 + a1) + a2. */
static void point_jacobian(const double p[3], const double R[9], double J[6]) {
  const double z_inv = 1.0 / p[2];
  const double z_inv_sq = z_inv * z_inv;
  const double j[6] = {-(z_inv), -(0.0), -(-p[0] * z_inv_sq), -(0.0), -(z_inv), -(-p[1] * z_inv_sq)};
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c)
      J[r * 3 + c] = (j[r * 3 + 0] * R[0 * 3 + c] + j[r * 3 + 1] * R[1 * 3 + c]) + j[r * 3 + 2] * R[2 * 3 + c];
}

/* Point::optimize, S/point.cpp:130-192.  obs k: pose of the observing frame T_f_w[k] and bearing f[k]. */
int svo_orc_point_optimize(int n_iter, double pos[3], int n_obs, const double* obs_T_f_w, const double* obs_f,
                           int* iters_done) {
  double old_point[3] = {pos[0], pos[1], pos[2]};
  double chi2 = 0.0;
  int done = 0;
  for (int i = 0; i < n_iter; ++i) {
    double A[9] = {0}, b[3] = {0};
    double new_chi2 = 0.0;
    for (int k = 0; k < n_obs; ++k) {
      const double* T = obs_T_f_w + 7 * k;
      const double* fk = obs_f + 3 * k;
      double p[3], R[9], J[6];
      svo_orc_se3_act(T, pos, p);
      svo_orc_se3_rotation_matrix(T, R);
      point_jacobian(p, R, J);
      const double e0 = fk[0] / fk[2] - p[0] / p[2];
      const double e1 = fk[1] / fk[2] - p[1] / p[2];
      new_chi2 += e0 * e0 + e1 * e1;
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) A[r * 3 + c] += J[r] * J[c] + J[3 + r] * J[3 + c];
        b[r] -= J[r] * e0 + J[3 + r] * e1;
      }
    }
    double dp[3];
    svo_orc_ldlt3_solve(A, b, dp);
    done = i + 1;
    if ((i > 0 && new_chi2 > chi2) || isnan(dp[0])) {
      pos[0] = old_point[0]; pos[1] = old_point[1]; pos[2] = old_point[2];
      break;
    }
    old_point[0] = pos[0]; old_point[1] = pos[1]; old_point[2] = pos[2];
    pos[0] += dp[0]; pos[1] += dp[1]; pos[2] += dp[2];
    chi2 = new_chi2;
    double mx = -1;
    for (int k = 0; k < 3; ++k) { double a = fabs(dp[k]); if (a > mx) mx = a; }
    if (mx <= 0.0000000001) break;
  }
  if (iters_done) *iters_done = done;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* next row f-3: FastDetector::detect (feature_detection.cpp:77-122)         */
/* ------------------------------------------------------------------------ */

/* the 16-pixel Bresenham circle of radius 3, in OpenCV's order (features2d/src/fast_score.cpp, makeOffsets) */
static const int kFastDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int kFastDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* FAST-9/16 decision and score of one pixel.  PARITY UNPINNED: this is cv::FAST(img, kp, threshold, true) of
 * OpenCV 4.5.4 (features2d/src/fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>), a third-party dependency whose
 * source and library are not present under /root/reference; restated from its published algorithm:
 *   corner  <=>  9 contiguous circle pixels all darker than v - t or all brighter than v + t (strict);
 *   score   =    max(t, max over the 16 arcs of min(v - x), max over the arcs of min(x - v)) - 1, stored as u8. */
static int fast_pixel(const uint8_t* img, int stride, int x, int y, int t, int* score) {
  const int v = img[y * stride + x];
  int d[25];
  for (int k = 0; k < 25; ++k) d[k] = v - img[(y + kFastDy[k & 15]) * stride + x + kFastDx[k & 15]];
  int corner = 0;
  for (int pol = 0; pol < 2 && !corner; ++pol) {
    int count = 0;
    for (int k = 0; k < 25; ++k) {
      const int hit = pol == 0 ? (d[k] > t) : (d[k] < -t);
      if (hit) { if (++count > 8) { corner = 1; break; } } else count = 0;
    }
  }
  if (!corner) return 0;
  int a0 = t;
  for (int k = 0; k < 16; ++k) {
    int mn = d[k], mx = d[k];
    for (int j = 1; j < 9; ++j) { if (d[k + j] < mn) mn = d[k + j]; if (d[k + j] > mx) mx = d[k + j]; }
    if (mn > a0) a0 = mn;
    if (-mx > a0) a0 = -mx;
  }
  *score = (uint8_t)(a0 - 1);
  return 1;
}

/* cv::FAST with non-maximum suppression: key points in row-major order; returns their number */
int svo_orc_fast(const uint8_t* img, int w, int h, int threshold, int max_out, int* xs, int* ys, int* scores) {
  uint8_t* sc = (uint8_t*)calloc((size_t)w * h, 1);
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) {
      int s = 0;
      if (fast_pixel(img, w, x, y, threshold, &s)) sc[y * w + x] = (uint8_t)s;   /* a corner always scores >= t - 1 */
    }
  int n = 0;
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) {
      int s = 0;
      if (!fast_pixel(img, w, x, y, threshold, &s)) continue;
      const uint8_t* p = sc + y * w + x;
      if (s > p[1] && s > p[-1] && s > p[-w - 1] && s > p[-w] && s > p[-w + 1] && s > p[w - 1] && s > p[w] && s > p[w + 1]) {
        if (n < max_out) { xs[n] = x; ys[n] = y; scores[n] = s; }
        ++n;
      }
    }
  free(sc);
  return n;
}

/* vk::shiTomasiScore, S/vision.cpp:113-154 */
float svo_orc_shi_tomasi_score(const uint8_t* img, int cols, int rows, int u, int v) {
  float dXX = 0.0, dYY = 0.0, dXY = 0.0;
  const int halfbox_size = 4;
  const int box_size = 2 * halfbox_size;
  const int box_area = box_size * box_size;
  const int x_min = u - halfbox_size, x_max = u + halfbox_size;
  const int y_min = v - halfbox_size, y_max = v + halfbox_size;
  if (x_min < 1 || x_max >= cols - 1 || y_min < 1 || y_max >= rows - 1) return 0.0;
  const int stride = cols;
  for (int y = y_min; y < y_max; ++y) {
    const uint8_t* ptr_left = img + stride * y + x_min - 1;
    const uint8_t* ptr_right = img + stride * y + x_min + 1;
    const uint8_t* ptr_top = img + stride * (y - 1) + x_min;
    const uint8_t* ptr_bottom = img + stride * (y + 1) + x_min;
    for (int x = 0; x < box_size; ++x, ++ptr_left, ++ptr_right, ++ptr_top, ++ptr_bottom) {
      float dx = *ptr_right - *ptr_left;
      float dy = *ptr_bottom - *ptr_top;
      dXX += dx * dx;
      dYY += dy * dy;
      dXY += dx * dy;
    }
  }
  dXX = dXX / (2.0 * box_area);
  dYY = dYY / (2.0 * box_area);
  dXY = dXY / (2.0 * box_area);
  /* C++ overload resolution picks sqrt(float): the parenthesis is evaluated in f32, only the 0.5 factor is double */
  const float tr = dXX + dYY;
  const float root = sqrtf(tr * tr - 4 * (dXX * dYY - dXY * dXY));
  return (float)(0.5 * (tr - root));
}

/* FastDetector::detect, S/feature_detection.cpp:77-122: per pyramid level FAST corners, one corner per grid cell
 * (best Shi-Tomasi score, strictly above detection_threshold), cells flagged in `occupancy` skipped.
 * Outputs in cell order: px (level-0 coordinates), level, score.  The glue is PARITY UNPINNED (detect() cannot run
 * without cv::FAST); vk::shiTomasiScore is pinned. */
int svo_orc_detect_features(const uint8_t* const* pyr, int width, int height, int n_pyr_levels, int cell_size,
                            const uint8_t* occupancy, double detection_threshold, int* px_out, int* level_out,
                            float* score_out) {
  const int gc = (int)ceil((double)width / cell_size), gr = (int)ceil((double)height / cell_size);
  const int n_cells = gc * gr;
  int* cx = (int*)calloc((size_t)n_cells, sizeof(int));
  int* cy = (int*)calloc((size_t)n_cells, sizeof(int));
  int* cl = (int*)calloc((size_t)n_cells, sizeof(int));
  float* cs = (float*)malloc(sizeof(float) * (size_t)n_cells);
  for (int k = 0; k < n_cells; ++k) cs[k] = (float)detection_threshold;
  for (int L = 0; L < n_pyr_levels; ++L) {
    const int scale = 1 << L;
    const int w = width >> L, h = height >> L;
    const int cap = w * h;
    int* xs = (int*)malloc(sizeof(int) * (size_t)cap);
    int* ys = (int*)malloc(sizeof(int) * (size_t)cap);
    int* ss = (int*)malloc(sizeof(int) * (size_t)cap);
    const int n = svo_orc_fast(pyr[L], w, h, 10, cap, xs, ys, ss);
    for (int i = 0; i < n; ++i) {
      const float fx = (float)xs[i], fy = (float)ys[i];
      const int k = (int)((fy * scale) / cell_size) * gc + (int)((fx * scale) / cell_size);
      if (occupancy && occupancy[k]) continue;
      const float score = svo_orc_shi_tomasi_score(pyr[L], w, h, xs[i], ys[i]);
      if (score > cs[k]) { cx[k] = (int)(fx * scale); cy[k] = (int)(fy * scale); cs[k] = score; cl[k] = L; }
    }
    free(xs); free(ys); free(ss);
  }
  int n_out = 0;
  for (int k = 0; k < n_cells; ++k)
    if ((double)cs[k] > detection_threshold) {
      px_out[2 * n_out] = cx[k]; px_out[2 * n_out + 1] = cy[k];
      level_out[n_out] = cl[k];
      score_out[n_out] = cs[k];
      ++n_out;
    }
  free(cx); free(cy); free(cl); free(cs);
  return n_out;
}


/* ------------------------------------------------------------------------ */
/* next row f-2: the cell loop of Reprojector::reprojectMap                   */
/* ------------------------------------------------------------------------ */

/* Reprojector::reprojectMap, the loop over the grid cells (S/reprojector.cpp:149-166) with
 * Reprojector::reprojectCell (:180-241), on candidates that are already bucketed per cell and sorted by point
 * quality (cell.sort(pointQualityComparator), :183).  Serial semantics: the candidates of a cell are tried in
 * order until one matches (at most one feature per cell); candidates behind the winner are not touched; the loop
 * stops once n_matches exceeds max_fts.  deleted[i] <=> it->pt->type_ == TYPE_DELETED (counted as a trial, :190-194).
 * Point bookkeeping (n_failed_reproj_, n_succeeded_reproj_, type changes, :202-215) is the caller's: tried[] and
 * matched[] tell it which candidates were visited and with what result. */
int svo_orc_reproject_cells(const svo_orc_camera* cam, int n_kf, const uint8_t* const* const* kf_pyr, const double* T_kf_w,
                            const uint8_t* const* cur_pyr, const double T_cur_w[7], int n_cells, const int* cell_offset,
                            const int* kf_slot, const double* px_ref, const double* f_ref, const int* level_ref,
                            const double* pt_pos, const uint8_t* edgelet, const double* grad, const uint8_t* deleted,
                            double* px_cur, int max_fts, int n_pyr_levels, int align_max_iter, uint8_t* tried,
                            uint8_t* matched, int* search_level, int* cell_winner, size_t* n_matches_out,
                            size_t* n_trials_out) {
  (void)n_kf;
  size_t n_matches = 0, n_trials = 0;
  const int n_cand = cell_offset[n_cells];
  memset(tried, 0, (size_t)n_cand);
  memset(matched, 0, (size_t)n_cand);
  for (int c = 0; c < n_cells; ++c) cell_winner[c] = -1;
  for (int c = 0; c < n_cells; ++c) {
    for (int i = cell_offset[c]; i < cell_offset[c + 1]; ++i) {
      ++n_trials;
      tried[i] = 1;
      if (deleted[i]) continue;
      const double g[2] = {grad ? grad[2 * i] : 1.0, grad ? grad[2 * i + 1] : 0.0};
      int sl = 0;
      const int k = kf_slot[i];
      const int ok = svo_orc_find_match_direct(cam, kf_pyr[k], cur_pyr, T_kf_w + 7 * k, T_cur_w, px_ref + 2 * i, f_ref + 3 * i,
                                               level_ref[i], pt_pos + 3 * i, edgelet ? edgelet[i] : 0, g, n_pyr_levels,
                                               align_max_iter, px_cur + 2 * i, &sl);
      if (search_level) search_level[i] = sl;
      if (!ok) continue;
      matched[i] = 1;
      cell_winner[c] = i;
      ++n_matches;
      break;                                  /* maximum one point per cell (:238-239) */
    }
    if (n_matches > (size_t)max_fts) break;    /* :164-165 */
  }
  *n_matches_out = n_matches;
  *n_trials_out = n_trials;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Reprojector::reprojectMap on a flattened svo::Map (S/reprojector.cpp:72-168 with reprojectPoint :246-259,
 * reprojectCell :180-241, Map::getCloseKeyframes S/map.cpp:109-131, Frame::isVisible S/frame.cpp:162-172,
 * Point::getCloseViewObs S/point.cpp:101-125, Matcher::findMatchDirect S/matcher.cpp:156-202).
 * The pointer graph is given as index tables (svo_orc_map); what the reference changes on its objects is changed in
 * the tables: pt_type / pt_n_failed / pt_n_succeeded in place, pt_unlinked[p] = 1 for a point that went through
 * Map::safeDeletePoint (its keyframe features now have point == NULL) or whose candidate was erased.
 * New features of the frame are returned in creation order. */
/* ------------------------------------------------------------------------ */

/* S/frame.cpp:162-172 */
static int frame_is_visible(const svo_orc_camera* cam, const double T_f_w[7], const double xyz_w[3]) {
  double xyz_f[3], px[2];
  svo_orc_se3_act(T_f_w, xyz_w, xyz_f);
  if (xyz_f[2] < 0.0) return 0;
  svo_orc_world2cam(cam, xyz_f, px);
  return px[0] >= 0.0 && px[1] >= 0.0 && px[0] < cam->width && px[1] < cam->height;
}

/* S/point.cpp:101-125: index of the observation with the closest view (first maximum of the cosine above 0; the
 * first observation when none is above 0); returns whether its cosine reaches 0.5 */
static int point_close_view_obs(const svo_orc_map* m, int p, const double framepos[3], int* obs_out) {
  const double* pos = m->pt_pos + 3 * (size_t)p;
  double od[3] = {framepos[0] - pos[0], framepos[1] - pos[1], framepos[2] - pos[2]};
  {
    const double n2 = od[0] * od[0] + od[1] * od[1] + od[2] * od[2];
    if (n2 > 0.0) { const double nn = sqrt(n2); od[0] = od[0] / nn; od[1] = od[1] / nn; od[2] = od[2] / nn; }
  }
  int min_it = m->pt_obs_offset[p];
  double min_cos_angle = 0;
  for (int o = m->pt_obs_offset[p]; o < m->pt_obs_offset[p + 1]; ++o) {
    double Tinv[7];
    svo_orc_se3_inverse(m->T_kf_w + 7 * (size_t)m->obs_kf[o], Tinv);          /* (*it)->frame->pos() */
    double d[3] = {Tinv[0] - pos[0], Tinv[1] - pos[1], Tinv[2] - pos[2]};
    const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (n2 > 0.0) { const double nn = sqrt(n2); d[0] = d[0] / nn; d[1] = d[1] / nn; d[2] = d[2] / nn; }
    const double cos_angle = od[0] * d[0] + od[1] * d[1] + od[2] * d[2];
    if (cos_angle > min_cos_angle) { min_cos_angle = cos_angle; min_it = o; }
  }
  *obs_out = min_it;
  return !(min_cos_angle < 0.5);
}

typedef struct { int point; double px[2]; int seq; } orc_cand;

static int orc_cand_cmp(const void* a, const void* b) {          /* cell.sort(pointQualityComparator): stable, by type descending */
  const orc_cand* x = (const orc_cand*)a; const orc_cand* y = (const orc_cand*)b;
  return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

int svo_orc_reproject_map(const svo_orc_camera* cam, svo_orc_map* m, const uint8_t* const* const* kf_pyr,
                          const uint8_t* const* cur_pyr, const double T_cur_w[7], int grid_size, int max_fts, int max_n_kfs,
                          int n_pyr_levels, int align_max_iter, uint8_t* pt_unlinked, int* n_overlap_out, int* overlap_kf,
                          int* overlap_count, int* n_feat_out, double* feat_px, int* feat_level, int* feat_point,
                          uint8_t* feat_edgelet, double* feat_grad, size_t* n_matches_out, size_t* n_trials_out) {
  const int cols = (int)ceil((double)cam->width / grid_size), rows = (int)ceil((double)cam->height / grid_size);   /* :46-47 */
  const int n_cells = cols * rows;
  /* ---- Map::getCloseKeyframes (map.cpp:109-131), then close_kfs.sort by distance (:82-83; std::list::sort is stable) */
  int* close = (int*)malloc(sizeof(int) * (size_t)(m->n_kf > 0 ? m->n_kf : 1));
  double* dist = (double*)malloc(sizeof(double) * (size_t)(m->n_kf > 0 ? m->n_kf : 1));
  int n_close = 0;
  for (int k = 0; k < m->n_kf; ++k) {
    for (int j = 0; j < 5; ++j) {
      const int p = m->kf_key_point[5 * k + j];
      if (p < 0) continue;
      if (frame_is_visible(cam, T_cur_w, m->pt_pos + 3 * (size_t)p)) {
        const double* tk = m->T_kf_w + 7 * (size_t)k;
        const double dx = T_cur_w[0] - tk[0], dy = T_cur_w[1] - tk[1], dz = T_cur_w[2] - tk[2];
        close[n_close] = k;
        dist[n_close] = sqrt(dx * dx + dy * dy + dz * dz);
        ++n_close;
        break;
      }
    }
  }
  for (int i = 1; i < n_close; ++i) {                                   /* stable insertion sort */
    const int ck = close[i]; const double cd = dist[i];
    int j = i - 1;
    while (j >= 0 && cd < dist[j]) { close[j + 1] = close[j]; dist[j + 1] = dist[j]; --j; }
    close[j + 1] = ck; dist[j + 1] = cd;
  }
  /* ---- candidates per cell, in push_back order */
  size_t cap = (size_t)m->n_candidates + 1;
  for (int i = 0; i < n_close && i < max_n_kfs; ++i) cap += (size_t)(m->kf_ftr_offset[close[i] + 1] - m->kf_ftr_offset[close[i]]);
  orc_cand* cand = (orc_cand*)malloc(sizeof(orc_cand) * cap);
  int* cand_cell = (int*)malloc(sizeof(int) * cap);
  int n_cand = 0;
  uint8_t* projected = (uint8_t*)calloc((size_t)(m->n_points > 0 ? m->n_points : 1), 1);   /* last_projected_kf_id_ == frame->id_ */
  int n_overlap = 0;
  for (int i = 0; i < n_close && i < max_n_kfs; ++i) {                   /* :88-114 */
    const int k = close[i];
    overlap_kf[n_overlap] = k; overlap_count[n_overlap] = 0;
    for (int j = m->kf_ftr_offset[k]; j < m->kf_ftr_offset[k + 1]; ++j) {
      const int p = m->kf_ftr_point[j];
      if (p < 0 || pt_unlinked[p]) continue;                             /* (*it_ftr)->point == NULL */
      if (projected[p]) continue;
      projected[p] = 1;
      double xyz_f[3], px[2];                                            /* reprojectPoint (:246-259) */
      svo_orc_se3_act(T_cur_w, m->pt_pos + 3 * (size_t)p, xyz_f);
      svo_orc_world2cam(cam, xyz_f, px);
      if (svo_orc_is_in_frame(cam, (int)px[0], (int)px[1], 8, -1)) {
        cand_cell[n_cand] = (int)(px[1] / grid_size) * cols + (int)(px[0] / grid_size);
        cand[n_cand].point = p; cand[n_cand].px[0] = px[0]; cand[n_cand].px[1] = px[1];
        ++n_cand;
        overlap_count[n_overlap]++;
      }
    }
    ++n_overlap;
  }
  *n_overlap_out = n_overlap;
  for (int c = 0; c < m->n_candidates; ++c) {                            /* :118-138 */
    const int p = m->cand_point[c];
    if (p < 0 || pt_unlinked[p]) continue;                               /* erased from candidates_ earlier */
    double xyz_f[3], px[2];
    svo_orc_se3_act(T_cur_w, m->pt_pos + 3 * (size_t)p, xyz_f);
    svo_orc_world2cam(cam, xyz_f, px);
    if (svo_orc_is_in_frame(cam, (int)px[0], (int)px[1], 8, -1)) {
      cand_cell[n_cand] = (int)(px[1] / grid_size) * cols + (int)(px[0] / grid_size);
      cand[n_cand].point = p; cand[n_cand].px[0] = px[0]; cand[n_cand].px[1] = px[1];
      ++n_cand;
    } else {
      m->pt_n_failed[p] += 3;
      if (m->pt_n_failed[p] > 30) { m->pt_type[p] = 0 /* TYPE_DELETED */; pt_unlinked[p] = 1; }   /* deleteCandidate + erase */
    }
  }
  /* ---- the cell loop (:149-166) */
  double T_cur_inv[7];
  svo_orc_se3_inverse(T_cur_w, T_cur_inv);                               /* cur_frame.pos() */
  size_t n_matches = 0, n_trials = 0;
  int n_feat = 0;
  orc_cand* cell = (orc_cand*)malloc(sizeof(orc_cand) * (size_t)(n_cand > 0 ? n_cand : 1));
  for (int c = 0; c < n_cells; ++c) {
    /* the cell's list in push_back order, then cell.sort(pointQualityComparator) (:183): stable, higher type first */
    int nc = 0;
    for (int t = 3; t >= 0; --t)
      for (int i = 0; i < n_cand; ++i)
        if (cand_cell[i] == c && m->pt_type[cand[i].point] == t) { cell[nc] = cand[i]; cell[nc].seq = nc; ++nc; }
    qsort(cell, (size_t)nc, sizeof(orc_cand), orc_cand_cmp);             /* (already ordered: keeps the intent explicit) */
    int found = 0;
    for (int i = 0; i < nc && !found; ++i) {
      ++n_trials;
      const int p = cell[i].point;
      if (m->pt_type[p] == 0) continue;                                  /* TYPE_DELETED (:190-194) */
      int o = -1;
      int ok = point_close_view_obs(m, p, T_cur_inv, &o);
      double px_cur[2] = {cell[i].px[0], cell[i].px[1]};
      int sl = 0;
      double A[4] = {0, 0, 0, 0};
      if (ok) {
        const int k = m->obs_kf[o];
        const double g[2] = {m->obs_grad ? m->obs_grad[2 * (size_t)o] : 1.0, m->obs_grad ? m->obs_grad[2 * (size_t)o + 1] : 0.0};
        const int edge = m->obs_edgelet ? m->obs_edgelet[o] : 0;
        ok = svo_orc_find_match_direct(cam, kf_pyr[k], cur_pyr, m->T_kf_w + 7 * (size_t)k, T_cur_w, m->obs_px + 2 * (size_t)o,
                                       m->obs_f + 3 * (size_t)o, m->obs_level[o], m->pt_pos + 3 * (size_t)p, edge, g, n_pyr_levels,
                                       align_max_iter, px_cur, &sl);
        if (ok && edge) {                                                /* A_cur_ref_ of the successful call, for the new feature's grad */
          double T_ref_inv[7], T_cur_ref[7];
          svo_orc_se3_inverse(m->T_kf_w + 7 * (size_t)k, T_ref_inv);
          svo_orc_se3_mul(T_cur_w, T_ref_inv, T_cur_ref);
          const double* pp = m->pt_pos + 3 * (size_t)p;
          const double dx = T_ref_inv[0] - pp[0], dy = T_ref_inv[1] - pp[1], dz = T_ref_inv[2] - pp[2];
          svo_orc_get_warp_matrix_affine(cam, cam, m->obs_px + 2 * (size_t)o, m->obs_f + 3 * (size_t)o, sqrt(dx * dx + dy * dy + dz * dz),
                                         T_cur_ref, m->obs_level[o], A);
        }
      }
      if (!ok) {                                                         /* :202-209 */
        m->pt_n_failed[p]++;
        if (m->pt_type[p] == 2 /* UNKNOWN */ && m->pt_n_failed[p] > 15) { m->pt_type[p] = 0; pt_unlinked[p] = 1; }    /* safeDeletePoint */
        if (m->pt_type[p] == 1 /* CANDIDATE */ && m->pt_n_failed[p] > 30) { m->pt_type[p] = 0; pt_unlinked[p] = 1; }  /* deleteCandidatePoint */
        continue;
      }
      m->pt_n_succeeded[p]++;                                            /* :211-214 */
      if (m->pt_type[p] == 2 && m->pt_n_succeeded[p] > 10) m->pt_type[p] = 3;
      feat_px[2 * n_feat] = px_cur[0]; feat_px[2 * n_feat + 1] = px_cur[1];   /* new Feature(frame, it->px, search_level_) (:217) */
      feat_level[n_feat] = sl;
      feat_point[n_feat] = p;
      feat_edgelet[n_feat] = 0;
      feat_grad[2 * n_feat] = 1.0; feat_grad[2 * n_feat + 1] = 0.0;
      if (m->obs_edgelet && m->obs_edgelet[o]) {                         /* :224-229 */
        double g0 = A[0] * m->obs_grad[2 * (size_t)o] + A[1] * m->obs_grad[2 * (size_t)o + 1];
        double g1 = A[2] * m->obs_grad[2 * (size_t)o] + A[3] * m->obs_grad[2 * (size_t)o + 1];
        const double n2 = g0 * g0 + g1 * g1;
        if (n2 > 0.0) { const double nn = sqrt(n2); g0 = g0 / nn; g1 = g1 / nn; }
        feat_edgelet[n_feat] = 1;
        feat_grad[2 * n_feat] = g0; feat_grad[2 * n_feat + 1] = g1;
      }
      ++n_feat;
      found = 1;
    }
    if (found) ++n_matches;
    if (n_matches > (size_t)max_fts) break;                              /* :164-165 */
  }
  *n_feat_out = n_feat;
  *n_matches_out = n_matches;
  *n_trials_out = n_trials;
  free(cell); free(projected); free(cand_cell); free(cand); free(dist); free(close);
  return 0;
}
