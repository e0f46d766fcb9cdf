// svo_refine.hip -- next rows of the hot path: the two small Gauss-Newton refinements that follow
// SparseImgAlign and the reprojection matching in FrameHandlerMono::processFrame.
//
//   pose_refine_kernel   pose_optimizer::optimizeGaussNewton   S/pose_optimizer.cpp:31-181
//   point_refine_kernel  Point::optimize                       S/point.cpp:130-192
//   ldlt6_batch_kernel   Eigen LDLT 6x6 solve as used by both solvers (test utility: device == Eigen bit for bit)
//
// pose_refine_kernel: one workgroup per frame calls svo_pose::pose_refine_block (svo_pose_refine.h).
#include "svo_internal.h"
#include "svo_pose_refine.h"

namespace {

using namespace svo_dev;

using namespace svo_pose;

__global__ __launch_bounds__(PR_THREADS) void pose_refine_kernel(
    int max_n, const int* __restrict__ n_feat, const double* __restrict__ T_in, const double* __restrict__ f,
    const double* __restrict__ pos, const int* __restrict__ level, uint8_t* __restrict__ has_point, double em,
    double reproj_thresh, int n_iter, float* __restrict__ err_ws, double* __restrict__ sq_init_ws,
    double* __restrict__ sq_final_ws, PoseOptOut* __restrict__ out) {
  pose_refine_block(blockIdx.x, max_n, n_feat, T_in, f, pos, level, has_point, em, reproj_thresh, n_iter, err_ws, sq_init_ws, sq_final_ws, out);
}

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

// Point::optimize (S/point.cpp:130-192): thread per map point, observations in CSR form, list order kept
__global__ void point_refine_kernel(int n_points, int n_iter, double* __restrict__ pos, const int* __restrict__ obs_offset,
                                    const double* __restrict__ obs_T, const double* __restrict__ obs_f,
                                    int* __restrict__ iters_out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_points) return;
  double P[3] = {pos[3 * p], pos[3 * p + 1], pos[3 * p + 2]};
  double old_point[3] = {P[0], P[1], P[2]};
  double chi2 = 0.0;
  int done = 0;
  const int o0 = obs_offset[p], o1 = obs_offset[p + 1];
  for (int i = 0; i < n_iter; ++i) {
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};
    double new_chi2 = 0.0;
    for (int k = o0; k < o1; ++k) {
      double T[7], q[3], R[9], J[6];
      for (int t = 0; t < 7; ++t) T[t] = obs_T[7 * k + t];
      const double fx = obs_f[3 * k], fy = obs_f[3 * k + 1], fz = obs_f[3 * k + 2];
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
  pos[3 * p] = P[0]; pos[3 * p + 1] = P[1]; pos[3 * p + 2] = P[2];
  if (iters_out) iters_out[p] = done;
}

// one system per wave (lane 0 only: ldlt6_solve_reg's contract)
__global__ void ldlt6_batch_kernel(int n, const double* __restrict__ H, const double* __restrict__ b, double* __restrict__ x) {
  const int i = blockIdx.x;
  if (i >= n || threadIdx.x != 0) return;
  double Hm[36], bv[6], xv[6];
  for (int k = 0; k < 36; ++k) Hm[k] = H[36 * (size_t)i + k];
  for (int k = 0; k < 6; ++k) bv[k] = b[6 * (size_t)i + k];
  ldlt6_solve_reg(Hm, bv, xv);
  for (int k = 0; k < 6; ++k) x[6 * (size_t)i + k] = xv[k];
}

}  // namespace

extern "C" {

int svo_hip_pose_optimize_batch_dev(svo_hip_ctx* ctx, int batch, int max_n, const int32_t* n_feat_dev,
                                    const double* T_f_w_dev, const double* f_dev, const double* pos_dev,
                                    const int32_t* level_dev, uint8_t* has_point_dev, double error_multiplier2,
                                    double reproj_thresh, int n_iter, svo_hip_pose_opt_result* results_dev) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, batch > 0 && max_n > 0 && n_iter >= 0);
  SVO_REQUIRE(ctx, n_feat_dev && T_f_w_dev && f_dev && pos_dev && level_dev && has_point_dev && results_dev);
  SVO_REQUIRE(ctx, error_multiplier2 > 0.0);
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  const size_t per = (size_t)batch * max_n;
  void* ws = nullptr;
  const int rc = svo_ctx_scratch(ctx, per * (sizeof(float) + 2 * sizeof(double)) + 64, &ws);
  if (rc != SVO_HIP_OK) return rc;
  double* sq_init = reinterpret_cast<double*>(ws);
  double* sq_final = sq_init + per;
  float* err = reinterpret_cast<float*>(sq_final + per);
  hipLaunchKernelGGL(pose_refine_kernel, dim3(batch), dim3(PR_THREADS), 0, ctx->stream, max_n, n_feat_dev, T_f_w_dev, f_dev,
                     pos_dev, level_dev, has_point_dev, error_multiplier2, reproj_thresh, n_iter, err, sq_init, sq_final,
                     reinterpret_cast<PoseOptOut*>(results_dev));
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

int svo_hip_pose_optimize(svo_hip_ctx* ctx, int n, const double T_f_w[7], const double* f, const double* pos,
                          const int32_t* level, uint8_t* has_point, double error_multiplier2, double reproj_thresh,
                          int n_iter, svo_hip_pose_opt_result* result) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n >= 0 && T_f_w && result && (n == 0 || (f && pos && level && has_point)));
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  const int max_n = n > 0 ? n : 1;
  // one layout on both sides -- [T][f][pos][level][n][result][has_point] -- gathered in page-locked memory: one transfer in,
  // one out ([result][has_point]) instead of six and two
  const size_t o_T = 0, o_f = o_T + 7 * sizeof(double), o_p = o_f + 3 * sizeof(double) * max_n,
               o_l = o_p + 3 * sizeof(double) * max_n, o_n = o_l + sizeof(int32_t) * max_n,
               o_r = (o_n + sizeof(int32_t) + 7) & ~(size_t)7, o_h = o_r + sizeof(svo_hip_pose_opt_result),
               bytes = o_h + (size_t)max_n + 64;
  char* d = nullptr;
  char* hs = nullptr;
  {
    const int rc_st = svo_ctx_staging(ctx, bytes, &d);
    if (rc_st != SVO_HIP_OK) return rc_st;
    const int rc_hs = svo_ctx_host_staging(ctx, bytes, &hs);
    if (rc_hs != SVO_HIP_OK) return rc_hs;
  }
  memcpy(hs + o_T, T_f_w, 7 * sizeof(double));
  if (n > 0) {
    memcpy(hs + o_f, f, 3 * sizeof(double) * n); memcpy(hs + o_p, pos, 3 * sizeof(double) * n);
    memcpy(hs + o_l, level, sizeof(int32_t) * n); memcpy(hs + o_h, has_point, (size_t)n);
  }
  const int32_t n32 = n;
  memcpy(hs + o_n, &n32, sizeof(int32_t));
  int rc = SVO_HIP_OK;
  hipError_t e = hipMemcpyAsync(d, hs, o_h + (size_t)max_n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    rc = svo_hip_pose_optimize_batch_dev(ctx, 1, max_n, reinterpret_cast<int32_t*>(d + o_n), reinterpret_cast<double*>(d + o_T),
                                         reinterpret_cast<double*>(d + o_f), reinterpret_cast<double*>(d + o_p),
                                         reinterpret_cast<int32_t*>(d + o_l), reinterpret_cast<uint8_t*>(d + o_h), error_multiplier2,
                                         reproj_thresh, n_iter, reinterpret_cast<svo_hip_pose_opt_result*>(d + o_r));
    if (rc == SVO_HIP_OK) {
      e = hipMemcpyAsync(hs + o_r, d + o_r, sizeof(svo_hip_pose_opt_result) + (size_t)max_n, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e == hipSuccess) {
        memcpy(result, hs + o_r, sizeof(*result));
        if (n > 0) memcpy(has_point, hs + o_h, (size_t)n);
      }
    }
  }
  if (e != hipSuccess) return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_pose_optimize", hipGetErrorString(e));
  return rc;
}

int svo_hip_point_optimize_batch_dev(svo_hip_ctx* ctx, int n_points, int n_iter, double* pos_dev,
                                     const int32_t* obs_offset_dev, const double* obs_T_f_w_dev,
                                     const double* obs_f_dev, int32_t* iters_dev) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n_points >= 0 && n_iter >= 0);
  if (n_points == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, pos_dev && obs_offset_dev && obs_T_f_w_dev && obs_f_dev);
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(point_refine_kernel, dim3((n_points + 63) / 64), dim3(64), 0, ctx->stream, n_points, n_iter, pos_dev,
                     obs_offset_dev, obs_T_f_w_dev, obs_f_dev, iters_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

int svo_hip_point_optimize_batch(svo_hip_ctx* ctx, int n_points, int n_iter, double* pos, const int32_t* obs_offset,
                                 const double* obs_T_f_w, const double* obs_f, int32_t* iters) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n_points >= 0 && n_iter >= 0);
  if (n_points == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, pos && obs_offset);
  const int m = obs_offset[n_points];
  SVO_REQUIRE(ctx, m >= 0 && (m == 0 || (obs_T_f_w && obs_f)));
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  const size_t mm = m > 0 ? (size_t)m : 1;
  // [T][f][offsets][pos][iterations]: gathered in page-locked memory, one transfer in, one out ([pos][iterations])
  const size_t o_T = 0, o_f = o_T + sizeof(double) * 7 * mm, o_o = o_f + sizeof(double) * 3 * mm,
               o_p = (o_o + sizeof(int32_t) * ((size_t)n_points + 1) + 7) & ~(size_t)7, o_i = o_p + sizeof(double) * 3 * n_points,
               bytes = o_i + sizeof(int32_t) * (size_t)n_points + 64;
  char* d = nullptr;
  char* hs = nullptr;
  {
    const int rc_st = svo_ctx_staging(ctx, bytes, &d);
    if (rc_st != SVO_HIP_OK) return rc_st;
    const int rc_hs = svo_ctx_host_staging(ctx, bytes, &hs);
    if (rc_hs != SVO_HIP_OK) return rc_hs;
  }
  if (m > 0) { memcpy(hs + o_T, obs_T_f_w, sizeof(double) * 7 * m); memcpy(hs + o_f, obs_f, sizeof(double) * 3 * m); }
  memcpy(hs + o_o, obs_offset, sizeof(int32_t) * ((size_t)n_points + 1));
  memcpy(hs + o_p, pos, sizeof(double) * 3 * n_points);
  hipError_t e = hipMemcpyAsync(d, hs, o_i, hipMemcpyHostToDevice, ctx->stream);
  int rc = SVO_HIP_OK;
  if (e == hipSuccess) {
    rc = svo_hip_point_optimize_batch_dev(ctx, n_points, n_iter, reinterpret_cast<double*>(d + o_p), reinterpret_cast<int32_t*>(d + o_o),
                                          reinterpret_cast<double*>(d + o_T), reinterpret_cast<double*>(d + o_f),
                                          reinterpret_cast<int32_t*>(d + o_i));
    if (rc == SVO_HIP_OK) {
      e = hipMemcpyAsync(hs + o_p, d + o_p, bytes - 64 - o_p, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e == hipSuccess) {
        memcpy(pos, hs + o_p, sizeof(double) * 3 * n_points);
        if (iters) memcpy(iters, hs + o_i, sizeof(int32_t) * (size_t)n_points);
      }
    }
  }
  if (e != hipSuccess) return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_point_optimize_batch", hipGetErrorString(e));
  return rc;
}

int svo_hip_ldlt6_solve_batch(svo_hip_ctx* ctx, int n, const double* H, const double* b, double* x) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n >= 0 && (n == 0 || (H && b && x)));
  if (n == 0) return SVO_HIP_OK;
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  char* dc = nullptr;
  {
    const int rc_st = svo_ctx_staging(ctx, sizeof(double) * 48 * (size_t)n, &dc);
    if (rc_st != SVO_HIP_OK) return rc_st;
  }
  double* d = reinterpret_cast<double*>(dc);
  double *dH = d, *db = d + 36 * (size_t)n, *dx = db + 6 * (size_t)n;
  hipError_t e = hipMemcpyAsync(dH, H, sizeof(double) * 36 * n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(db, b, sizeof(double) * 6 * n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(ldlt6_batch_kernel, dim3(n), dim3(64), 0, ctx->stream, n, dH, db, dx);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(x, dx, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_ldlt6_solve_batch", hipGetErrorString(e));
  return SVO_HIP_OK;
}

}  // extern "C"
