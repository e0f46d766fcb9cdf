// svo_point_refine.h -- Point::optimize (S/point.cpp:130-192) of one map point as a device function: the batch kernel of
// svo_refine.hip (observations in CSR arrays) and the tracking chain's structure step (observations read straight from the
// map tables, svo_track.hip) run the same code, so both give the reference's bits.
#pragma once
#include "svo_device_math.h"

namespace svo_dev {

// Point::jacobian_xyz2uv (I/point.h:83-97): -[1/z 0 -x/z^2; 0 1/z -y/z^2] * R_f_w, inner sums (a0 + a1) + a2
SVO_DEV void point_jacobian(const double* p, const double* R, double* J) {
  const double z_inv = 1.0 / p[2];
  const double z_inv_sq = z_inv * z_inv;
  const double j[6] = {-(z_inv), -(0.0), -(-p[0] * z_inv_sq), -(0.0), -(z_inv), -(-p[1] * z_inv_sq)};
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      J[r * 3 + c] = (j[r * 3 + 0] * R[0 * 3 + c] + j[r * 3 + 1] * R[1 * 3 + c]) + j[r * 3 + 2] * R[2 * 3 + c];
}

// Gauss-Newton over the observations o0 .. o1-1 in list order; load_obs(k, T[7], f[3]) gives observation k's frame pose
// (world -> frame) and bearing.  P: the point, updated in place.  Returns the number of iterations taken.
template <class LoadObs>
SVO_DEV int point_refine_one(double* P, int o0, int o1, int n_iter, LoadObs load_obs) {
  double old_point[3] = {P[0], P[1], P[2]};
  double chi2 = 0.0;
  int done = 0;
  for (int i = 0; i < n_iter; ++i) {
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};
    double new_chi2 = 0.0;
    for (int k = o0; k < o1; ++k) {
      double T[7], fo[3], q[3], R[9], J[6];
      load_obs(k, T, fo);
      const double fx = fo[0], fy = fo[1], fz = fo[2];
      se3_act(T, P, q);
      se3_rotation_matrix(T, R);
      point_jacobian(q, R, J);
      const double e0 = fx / fz - q[0] / q[2];
      const double e1 = fy / fz - q[1] / q[2];
      new_chi2 += e0 * e0 + e1 * e1;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) A[r * 3 + c] += J[r] * J[c] + J[3 + r] * J[3 + c];
        bb[r] -= J[r] * e0 + J[3 + r] * e1;
      }
    }
    double dp[3];
    ldlt_solve<3>(A, bb, dp);
    done = i + 1;
    if ((i > 0 && new_chi2 > chi2) || dp[0] != dp[0]) {
      P[0] = old_point[0]; P[1] = old_point[1]; P[2] = old_point[2];
      break;
    }
    old_point[0] = P[0]; old_point[1] = P[1]; old_point[2] = P[2];
    P[0] += dp[0]; P[1] += dp[1]; P[2] += dp[2];
    chi2 = new_chi2;
    double mx = -1;
    for (int k = 0; k < 3; ++k) { const double a = fabs(dp[k]); if (a > mx) mx = a; }
    if (mx <= 0.0000000001) break;
  }
  return done;
}

}  // namespace svo_dev
