// svo_depth.hip -- align2D batch, DepthFilter::updateSeed / computeTau batches and the
// full per-seed DepthFilter::updateSeeds body (visibility, epipolar ZMSSD search, align2D,
// triangulation, tau, Bayes update, convergence test) on gfx950.
//
// Reference: S/depth_filter.cpp:237-416, S/matcher.cpp:36-147,207-355,
// I/patch_score.h:40-220, S/feature_alignment.cpp:154-282.
//
// Mapping: three stages (thread-per-seed geometry, wave-per-seed pixel work, thread-per-seed update);
// the fp64 geometry of a seed is evaluated once, 64 seeds per wave instruction;
// the 10x10 warped reference patch is built by lanes 0..99 (two passes) into LDS; the
// epipolar search assigns one candidate position per lane (64 candidates per pass), each
// lane computing a whole 8x8 ZMSSD with packed u8 dot products against the LDS patch;
// the best candidate is an order-preserving wave arg-min (first minimum wins, exactly the
// serial `zmssd < zmssd_best` scan); the sub-pixel refinement is the wave-per-patch
// align2D of svo_align_device.h.
//
// Exactness: all integer/index work (candidate pixel, search level, warped patch bytes,
// ZMSSD, arg-min) is bit-exact against the CPU path: the fp64 chains that feed integer
// conversions use the same operation order (built with -ffp-contract=off), and the
// epipolar abscissa uv_i is produced by the same repeated addition uv += step as
// S/matcher.cpp:299 (each lane replays its prefix).
#include "svo_align_device.h"
#include <cstddef>
#include <vector>

#include "svo_internal.h"
#include "svo_match_device.h"

using namespace svo_dev;

namespace {

constexpr int ZMSSD_THRESHOLD = 2000 * 64;     // I/patch_score.h:46

// ---- align2D / align1D over n patches: one DPP quad per patch, 16 patches per single-wave workgroup ------------
// The [n][100] (and optional [n][64]) patch arrays are read with coalesced word loads into LDS and handed to the
// lanes from there: lane q of a patch takes the ten words of border rows 2q .. 2q+3 (svo_align_device.h).
// One wave per workgroup: a wave leaves as soon as its own 16 patches are done and its slot is refilled at once
// (with 4-wave workgroups the next workgroup waited for the slowest of 64 patches: measured average residency 2.0 of
// 4 waves per SIMD at 200 000 patches).
constexpr int ALIGN_BLOCK = 64;
constexpr int ALIGN_PATCHES = ALIGN_BLOCK / ALIGN_LANES_PER_PATCH;     // patches per workgroup
constexpr int ALIGN_STAGE_ROUNDS = (ALIGN_PATCHES * 25 + ALIGN_BLOCK - 1) / ALIGN_BLOCK;
constexpr int ALIGN_STAGE_WORDS = ALIGN_STAGE_ROUNDS * ALIGN_BLOCK;   // >= ALIGN_PATCHES * 25: no bounds test while staging

SVO_DEV void stage_patch_words(const uint8_t* __restrict__ pwb, const uint8_t* __restrict__ ref_patch, int first, int n,
                               uint32_t* s_words, QuadPatch& qp) {
  const int tid = threadIdx.x;
  const int pl = tid >> 2, q = tid & 3;
  const int n_here = min(ALIGN_PATCHES, n - first);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(pwb + (size_t)first * 100);
  // unconditional loads at a clamped index: all of a thread's loads are in flight together (words beyond the last
  // patch are never used)
  const int last = n_here * 25 - 1;
  uint32_t stage[ALIGN_STAGE_ROUNDS];
#pragma unroll
  for (int k = 0; k < ALIGN_STAGE_ROUNDS; ++k) stage[k] = src[min(k * ALIGN_BLOCK + tid, last)];      // all in flight together
#pragma unroll
  for (int k = 0; k < ALIGN_STAGE_ROUNDS; ++k) s_words[k * ALIGN_BLOCK + tid] = stage[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 10; ++k) qp.b[k] = s_words[pl * 25 + 5 * q + k];
  if (ref_patch) {
    __syncthreads();
    const uint32_t* src2 = reinterpret_cast<const uint32_t*>(ref_patch + (size_t)first * 64);
    const int last2 = n_here * 16 - 1;
#pragma unroll
    for (int k = 0; k < ALIGN_PATCHES * 16 / ALIGN_BLOCK; ++k) {
      const int w = k * ALIGN_BLOCK + tid;
      s_words[w + (w >> 4)] = src2[min(w, last2)];                 // row stride 17 words: conflict-free reads
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) qp.p[k] = s_words[pl * 17 + 4 * q + k];
  } else {
    patch_from_border(qp);
  }
}

__global__ __launch_bounds__(ALIGN_BLOCK) void align2d_kernel(const uint8_t* __restrict__ img, int cols, int rows, int n,
                                                              const uint8_t* __restrict__ pwb,
                                                              const uint8_t* __restrict__ ref_patch, int n_iter,
                                                              double* __restrict__ px, uint8_t* __restrict__ converged,
                                                              int32_t* __restrict__ iters) {
  __shared__ uint32_t s_words[ALIGN_STAGE_WORDS];
  const int first = blockIdx.x * ALIGN_PATCHES;
  const int w = first + (threadIdx.x >> 2);
  const bool have = w < n;
  QuadPatch qp;
  stage_patch_words(pwb, ref_patch, first, n, s_words, qp);
  double u = 0.0, v = 0.0;
  if (have) { u = px[2 * (size_t)w]; v = px[2 * (size_t)w + 1]; }
  int it = 0;
  const bool ok = align2d_quad(img, cols, rows, cols, qp, n_iter, have, &u, &v, &it);
  if (have && (threadIdx.x & 3) == 0) {
    px[2 * (size_t)w] = u;
    px[2 * (size_t)w + 1] = v;
    converged[w] = ok ? 1 : 0;
    if (iters) iters[w] = it;
  }
}

// align1D over n patches (edgelets / Matcher::Options::align_1d; secondary)
__global__ __launch_bounds__(ALIGN_BLOCK) void align1d_kernel(const uint8_t* __restrict__ img, int cols, int rows, int n,
                                                              const uint8_t* __restrict__ pwb,
                                                              const uint8_t* __restrict__ ref_patch,
                                                              const float* __restrict__ dir, int n_iter,
                                                              double* __restrict__ px, uint8_t* __restrict__ converged,
                                                              double* __restrict__ h_inv, int32_t* __restrict__ iters) {
  __shared__ uint32_t s_words[ALIGN_STAGE_WORDS];
  const int first = blockIdx.x * ALIGN_PATCHES;
  const int w = first + (threadIdx.x >> 2);
  const bool have = w < n;
  QuadPatch qp;
  stage_patch_words(pwb, ref_patch, first, n, s_words, qp);
  double u = 0.0, v = 0.0, hi = 0.0;
  float d0 = 0.0f, d1 = 0.0f;
  if (have) { u = px[2 * (size_t)w]; v = px[2 * (size_t)w + 1]; d0 = dir[2 * (size_t)w]; d1 = dir[2 * (size_t)w + 1]; }
  int it = 0;
  const bool ok = align1d_quad(img, cols, rows, cols, d0, d1, qp, n_iter, have, &u, &v, &hi, &it);
  if (have && (threadIdx.x & 3) == 0) {
    px[2 * (size_t)w] = u;
    px[2 * (size_t)w + 1] = v;
    converged[w] = ok ? 1 : 0;
    if (h_inv) h_inv[w] = hi;
    if (iters) iters[w] = it;
  }
}

// ---- DepthFilter::updateSeed (S/depth_filter.cpp:359-391) -------------------------------------
struct SeedState { float a, b, mu, z_range, sigma2; };

SVO_DEV double normal_pdf_quirk(double x, double mean, double std_dev) {
  const double SQRT_2_PI = 1.41421356237309505;       // sqrt(2), as in the reference (:360)
  const double q = (x - mean) / std_dev;
  const double exponent = -0.5 * (q * q);              // pow(q, 2)
  return (1 / (std_dev * SQRT_2_PI)) * exp(exponent);
}

SVO_DEV void update_seed(const float x, const float tau2, SeedState* seed) {
  const float norm_scale = sqrtf(seed->sigma2 + tau2);
  if (norm_scale != norm_scale) return;
  const float s2 = (float)(1. / (1. / seed->sigma2 + 1. / tau2));
  const float m = s2 * (seed->mu / seed->sigma2 + x / tau2);
  float C1 = (float)(seed->a / (seed->a + seed->b) * normal_pdf_quirk(x, seed->mu, norm_scale));
  float C2 = (float)(seed->b / (seed->a + seed->b) * 1. / seed->z_range);
  const float normalization_constant = C1 + C2;
  C1 /= normalization_constant;
  C2 /= normalization_constant;
  const float f = (float)(C1 * (seed->a + 1.) / (seed->a + seed->b + 1.) + C2 * seed->a / (seed->a + seed->b + 1.));
  const float e = (float)(C1 * (seed->a + 1.) * (seed->a + 2.) / ((seed->a + seed->b + 1.) * (seed->a + seed->b + 2.)) +
                          C2 * seed->a * (seed->a + 1.0f) / ((seed->a + seed->b + 1.0f) * (seed->a + seed->b + 2.0f)));
  const float mu_new = C1 * m + C2 * seed->mu;
  seed->sigma2 = C1 * (s2 + m * m) + C2 * (seed->sigma2 + seed->mu * seed->mu) - mu_new * mu_new;
  seed->mu = mu_new;
  seed->a = (e - f) / (f - e / f);
  seed->b = seed->a * (1.0f - f) / f;
}

// 44 B/seed of HBM traffic (20 B state + 8 B measurement read, 16 B written): purely HBM-bound
__global__ void update_seed_kernel(int n, const float* __restrict__ x, const float* __restrict__ tau2,
                                   float* __restrict__ a, float* __restrict__ b, float* __restrict__ mu,
                                   const float* __restrict__ z_range, float* __restrict__ sigma2) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    SeedState s = {a[i], b[i], mu[i], z_range[i], sigma2[i]};
    update_seed(x[i], tau2[i], &s);
    a[i] = s.a; b[i] = s.b; mu[i] = s.mu; sigma2[i] = s.sigma2;
  }
}

// S/depth_filter.cpp:396-416; PI = 3.14159265 (I/global.h:92)
SVO_DEV double compute_tau(const double* t, const double* f, double z, double px_error_angle) {
  const double PI_SVO = 3.14159265;
  const double a[3] = {f[0] * z - t[0], f[1] * z - t[1], f[2] * z - t[2]};
  const double t_norm = sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
  const double a_norm = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
  const double alpha = acos(((f[0] * t[0] + f[1] * t[1]) + f[2] * t[2]) / t_norm);
  const double beta = acos(((a[0] * -t[0] + a[1] * -t[1]) + a[2] * -t[2]) / (t_norm * a_norm));
  const double beta_plus = beta + px_error_angle;
  const double gamma_plus = PI_SVO - alpha - beta_plus;
  const double z_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
  return z_plus - z;
}

struct Vec3 { double v[3]; };

__global__ void compute_tau_kernel(int n, Vec3 t, const double* __restrict__ f, const double* __restrict__ z,
                                   double px_error_angle, double* __restrict__ tau) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double fi[3] = {f[3 * (size_t)i], f[3 * (size_t)i + 1], f[3 * (size_t)i + 2]};
    tau[i] = compute_tau(t.v, fi, z[i], px_error_angle);
  }
}

// ---- the camera model exactly as every kernel evaluates it, over a batch (parity tests against the reference's
// ---- compiled vk::PinholeCamera, tests/golden/camera_ref.npz) -------------------------------------------------
__global__ void camera_batch_kernel(Cam cam, int n, const double* __restrict__ xyz, const double* __restrict__ uv,
                                    const double* __restrict__ px, const int32_t* __restrict__ obs, int boundary, int level,
                                    double* __restrict__ px_of_xyz, double* __restrict__ px_of_uv,
                                    double* __restrict__ f_of_px, uint8_t* __restrict__ in_frame) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (xyz && px_of_xyz) {
    const double p[3] = {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
    world2cam(cam, p, px_of_xyz + 2 * (size_t)i);
  }
  if (uv && px_of_uv) world2cam_uv(cam, uv[2 * (size_t)i], uv[2 * (size_t)i + 1], px_of_uv + 2 * (size_t)i);
  if (px && f_of_px) cam2world(cam, px[2 * (size_t)i], px[2 * (size_t)i + 1], f_of_px + 3 * (size_t)i);
  if (obs && in_frame) {
    const int ox = obs[2 * (size_t)i], oy = obs[2 * (size_t)i + 1];
    in_frame[i] = level < 0 ? (ox >= boundary && ox < cam.width - boundary && oy >= boundary && oy < cam.height - boundary)
                            : is_in_frame_level(cam, ox, oy, boundary, level);
  }
}

// ---- matcher pieces ---------------------------------------------------------------------------
// (get_warp_matrix_affine: svo_device_math.h)

// S/matcher.cpp:123-136
SVO_DEV bool depth_from_triangulation(const double* T_search_ref, const double* f_ref, const double* f_cur,
                                      double* depth) {
  double R[9];
  const double t[3] = {T_search_ref[0], T_search_ref[1], T_search_ref[2]};
  se3_rotation_matrix(T_search_ref, R);
  double a0[3];
  const double a1[3] = {f_cur[0], f_cur[1], f_cur[2]};
  for (int i = 0; i < 3; ++i) a0[i] = (R[3 * i] * f_ref[0] + R[3 * i + 1] * f_ref[1]) + R[3 * i + 2] * f_ref[2];
  const double m00 = (a0[0] * a0[0] + a0[1] * a0[1]) + a0[2] * a0[2];
  const double m01 = (a0[0] * a1[0] + a0[1] * a1[1]) + a0[2] * a1[2];
  const double m11 = (a1[0] * a1[0] + a1[1] * a1[1]) + a1[2] * a1[2];
  const double det = m00 * m11 - m01 * m01;
  if (det < 0.000001) return false;
  const double invdet = 1.0 / det;
  const double n00 = -(m11 * invdet), n01 = -(-m01 * invdet);
  double r0[3];
  for (int k = 0; k < 3; ++k) r0[k] = n00 * a0[k] + n01 * a1[k];
  *depth = fabs((r0[0] * t[0] + r0[1] * t[1]) + r0[2] * t[2]);
  return true;
}

struct DfFrame {
  Cam cam;
  double T_ref_cur[7];       // it->ftr->frame->T_f_w_ * frame->T_f_w_.inverse()   (depth_filter.cpp:264)
  double T_cur_ref_vis[7];   // T_ref_cur.inverse()                                (:265)
  double T_cur_ref[7];       // cur.T_f_w_ * ref.T_f_w_.inverse()                  (matcher.cpp:216)
  double T_ref_inv[7];       // ref.T_f_w_.inverse()                                (:314)
  double px_error_angle;
  size_t ref_level_off[SVO_HIP_MAX_LEVELS];
  size_t cur_level_off[SVO_HIP_MAX_LEVELS];
  int n_pyr_levels, align_max_iter, max_epi_search_steps;
  int keep_px_on_failure;    // findMatchDirect writes px_scaled back even when align fails (matcher.cpp:200)
  double conv_thresh;
};

// 8x8 ZMSSD of the LDS reference patch against the image patch whose top-left is `p`
// (I/patch_score.h:186-219: integer exact).
SVO_DEV int zmssd_8x8(const uint8_t* __restrict__ p, int stride, const uint32_t* patch_words, int sumA, int sumAA) {
  uint32_t sumB = 0, sumBB = 0, sumAB = 0;
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const uint8_t* row = p + y * stride;
    uint32_t w0, w1;
    __builtin_memcpy(&w0, row, 4);
    __builtin_memcpy(&w1, row + 4, 4);
    const uint32_t a0 = patch_words[2 * y], a1 = patch_words[2 * y + 1];
    sumB = __builtin_amdgcn_udot4(w0, 0x01010101u, sumB, false);
    sumB = __builtin_amdgcn_udot4(w1, 0x01010101u, sumB, false);
    sumBB = __builtin_amdgcn_udot4(w0, w0, sumBB, false);
    sumBB = __builtin_amdgcn_udot4(w1, w1, sumBB, false);
    sumAB = __builtin_amdgcn_udot4(w0, a0, sumAB, false);
    sumAB = __builtin_amdgcn_udot4(w1, a1, sumAB, false);
  }
  const int sB = (int)sumB, sBB = (int)sumBB, sAB = (int)sumAB;
  return sumAA - 2 * sAB + sBB - (sumA * sumA - 2 * sumA * sB + sB * sB) / 64;
}

// ---- DepthFilter::updateSeeds as three stages -------------------------------------------------
// G  thread per seed : visibility, epipolar segment, affine warp, search level, search plan  -> SeedRec
// S  wave per seed   : warp the 10x10 patch (LDS), ZMSSD search along the epipolar line, align2D -> SeedRec
// F  thread per seed : triangulation, computeTau, updateSeed, convergence                     -> seed arrays
// The fp64 geometry of a seed runs once (64 seeds per wave instruction) instead of on every lane of
// the seed's wave; only the pixel work is wave-per-seed.
// (SeedRec: svo_match_device.h)

// EXPLICIT_DEPTH = false: the per-seed body of DepthFilter::updateSeeds (visibility test, depth interval from mu/sigma2);
// EXPLICIT_DEPTH = true: Matcher::findEpipolarMatchDirect as a caller would use it directly, with d_estimate / d_min /
// d_max given per item (dep[3][n]) and no visibility test.
// (the per-seed body; `rec_out` = where this seed's record goes, `ref_slot` = its reference keyframe's pyramid slot for
// the search stage when one pass covers several keyframes -- 0 otherwise)
template <bool EXPLICIT_DEPTH>
SVO_DEV void df_geometry_seed(const Cam& cam, const double* T_cur_ref_vis, const double* T_cur_ref_, int n_pyr_levels,
                              int max_epi_search_steps, int ref_slot, int i, int n, const double* __restrict__ px,
                              const double* __restrict__ f, const int32_t* __restrict__ level, const float* __restrict__ smu,
                              const float* __restrict__ ssigma2, const double* __restrict__ dep, double* __restrict__ epi_len_out,
                              SeedRec* __restrict__ rec_out, const uint8_t* __restrict__ alive) {
  SeedRec rc;
  rc.uv0[0] = rc.uv0[1] = rc.step[0] = rc.step[1] = 0.0;
  rc.a00 = rc.a01 = rc.a10 = rc.a11 = rc.prx = rc.pry = 0.0f;
  rc.n_steps = 0; rc.search_level = 0; rc.path = -1; rc.status = SVO_HIP_SEED_NO_MATCH; rc.warp_nan = 0;
  rc.matched = 0; rc.n_zmssd = 0; rc.n_align = 0; rc.pad = ref_slot;
  const double fi[3] = {f[3 * (size_t)i], f[3 * (size_t)i + 1], f[3 * (size_t)i + 2]};
  const double px_ref[2] = {px[2 * (size_t)i], px[2 * (size_t)i + 1]};
  const int level_ref = level[i];
  bool live = true;
  if (alive && !alive[i]) { rc.status = SVO_HIP_SEED_ERASED; live = false; }     // erased from the list (seed batches)
  double d_estimate, d_min, d_max;
  if (EXPLICIT_DEPTH) {
    d_estimate = dep[i]; d_min = dep[(size_t)n + i]; d_max = dep[2 * (size_t)n + i];
    rc.z_inv_min = 0.0f;
  } else {
    const float mu = smu[i], sigma2 = ssigma2[i];
    // ---- visibility in the current frame (depth_filter.cpp:264-275)
    const double inv_mu = 1.0 / mu;
    const double pf[3] = {inv_mu * fi[0], inv_mu * fi[1], inv_mu * fi[2]};
    double xyz_f[3];
    se3_act(T_cur_ref_vis, pf, xyz_f);
    if (live && xyz_f[2] < 0.0) { rc.status = SVO_HIP_SEED_BEHIND; live = false; }
    if (live) {
      double pc[2];
      world2cam(cam, xyz_f, pc);
      const int ox = (int)pc[0], oy = (int)pc[1];
      if (!(ox >= 0 && ox < cam.width && oy >= 0 && oy < cam.height)) { rc.status = SVO_HIP_SEED_NOT_IN_FRAME; live = false; }
    }
    const float z_inv_min = mu + sqrtf(sigma2);
    rc.z_inv_min = z_inv_min;
    const float z_inv_lo = mu - sqrtf(sigma2);
    const float z_inv_max = (z_inv_lo < 0.00000001f) ? 0.00000001f : z_inv_lo;
    d_estimate = 1.0 / mu; d_min = 1.0 / z_inv_min; d_max = 1.0 / z_inv_max;
  }
  if (live) {
    // ---- Matcher::findEpipolarMatchDirect up to the search plan (matcher.cpp:216-296)
    const double* T_cur_ref = T_cur_ref_;
    double pa[3], pb[3], tmp[3];
    tmp[0] = fi[0] * d_min; tmp[1] = fi[1] * d_min; tmp[2] = fi[2] * d_min;
    se3_act(T_cur_ref, tmp, pa);
    tmp[0] = fi[0] * d_max; tmp[1] = fi[1] * d_max; tmp[2] = fi[2] * d_max;
    se3_act(T_cur_ref, tmp, pb);
    const double Aep[2] = {pa[0] / pa[2], pa[1] / pa[2]};
    const double Bep[2] = {pb[0] / pb[2], pb[1] / pb[2]};
    const double epi_dir[2] = {Aep[0] - Bep[0], Aep[1] - Bep[1]};
    double Acr[4];
    get_warp_matrix_affine(cam, px_ref, fi, d_estimate, T_cur_ref, level_ref, Acr);
    int search_level = 0;
    {
      double D = Acr[0] * Acr[3] - Acr[2] * Acr[1];
      while (D > 3.0 && search_level < n_pyr_levels - 1) { search_level += 1; D *= 0.25; }
    }
    rc.search_level = search_level;
    double px_A[2], px_B[2];
    world2cam_uv(cam, Aep[0], Aep[1], px_A);
    world2cam_uv(cam, Bep[0], Bep[1], px_B);
    double epi_length;
    {
      const double ex = px_A[0] - px_B[0], ey = px_A[1] - px_B[1];
      epi_length = sqrt(ex * ex + ey * ey) / (1 << search_level);
    }
    {
      // warp::warpAffine prologue (matcher.cpp:92-102)
      const double det = Acr[0] * Acr[3] - Acr[2] * Acr[1];
      const double invdet = 1.0 / det;
      rc.a00 = (float)(Acr[3] * invdet); rc.a01 = (float)(-Acr[1] * invdet);
      rc.a10 = (float)(-Acr[2] * invdet); rc.a11 = (float)(Acr[0] * invdet);
      rc.warp_nan = rc.a00 != rc.a00;
      rc.prx = (float)px_ref[0] / (1 << level_ref);
      rc.pry = (float)px_ref[1] / (1 << level_ref);
    }
    if (epi_len_out) epi_len_out[i] = epi_length;
    if (epi_length < 2.0) {
      rc.path = 0;
      rc.uv0[0] = (px_A[0] + px_B[0]) / 2.0;
      rc.uv0[1] = (px_A[1] + px_B[1]) / 2.0;
    } else {
      const size_t n_steps = (size_t)(epi_length / 0.7);
      if (n_steps > (size_t)max_epi_search_steps) {
        rc.path = 2;
      } else {
        rc.path = 1;
        rc.n_steps = (int)n_steps;
        rc.step[0] = epi_dir[0] / n_steps; rc.step[1] = epi_dir[1] / n_steps;
        rc.uv0[0] = Bep[0] - rc.step[0]; rc.uv0[1] = Bep[1] - rc.step[1];
      }
    }
  }
  *rec_out = rc;
}

template <bool EXPLICIT_DEPTH>
__global__ __launch_bounds__(256) void df_geometry_kernel(
    DfFrame fr, int n, const double* __restrict__ px, const double* __restrict__ f, const int32_t* __restrict__ level,
    const float* __restrict__ smu, const float* __restrict__ ssigma2, const double* __restrict__ dep,
    double* __restrict__ epi_len_out, SeedRec* __restrict__ recs, const uint8_t* __restrict__ alive = nullptr,
    int* __restrict__ ev_hist = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (ev_hist && i < 8) ev_hist[i] = 0;                   // status histogram of the pass (filled by the finalize stage)
  if (i >= n) return;
  df_geometry_seed<EXPLICIT_DEPTH>(fr.cam, fr.T_cur_ref_vis, fr.T_cur_ref, fr.n_pyr_levels, fr.max_epi_search_steps, 0, i, n, px, f,
                                   level, smu, ssigma2, dep, epi_len_out, recs + i, alive);
}

// ---- image windows in LDS -------------------------------------------------------------------------------------
// Both pixel phases of the search kernel gather bytes around one place of an image: the 100 bilinear samples of the
// affine warp, and the up to 16 overlapping 8x8 candidate patches of one chunk of the epipolar walk.  Read straight
// from memory that is 14 + 16 scattered load instructions per lane with ~40 cache-line accesses each, and the kernel
// is bound by L1 address processing (PMC: 0.8 line accesses per cycle and CU, TA busy 69 %).  Instead the 16 lanes of
// a seed copy the bounding box of what they need -- at most WIN_ROWS rows of WIN_PITCH bytes, one or two unaligned
// 16-byte loads per row, one row (two if the box is taller than 16) per lane -- into LDS and gather from there.  A box
// that does not fit (a strongly scaled warp, a segment steeper than the chunk allows) falls back to the direct loads.
constexpr int WIN_PITCH = 32;
constexpr int WIN_ROWS = 32;
constexpr int WIN_BYTES = WIN_ROWS * WIN_PITCH;

// lane cl (0..15) of a seed copies rows cl and cl + 16 of the h x WIN_PITCH box whose top-left byte is `src`.
// (Round 4 tried two lanes per row -- neighbouring lanes reading the two 16-byte halves of one row, four passes of eight
// rows, so that the address unit sees half as many cache lines per load: the kernel ran 23 % (100 k seeds) and 43 % (1 M
// seeds) SLOWER, same-box A/B; four conditional single-load blocks instead of two double-load ones.  Kept as it was.)
SVO_DEV void win_fill(const uint8_t* __restrict__ src, int pitch, int h, int cl, uint8_t* win) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int r = cl + 16 * half;
    if (r < h) {
      const uint8_t* p = src + (size_t)r * pitch;
      uint4 a, b;
      __builtin_memcpy(&a, p, 16);                       // unaligned global_load_dwordx4
      __builtin_memcpy(&b, p + 16, 16);
      *reinterpret_cast<uint4*>(win + r * WIN_PITCH) = a;
      *reinterpret_cast<uint4*>(win + r * WIN_PITCH + 16) = b;
    }
  }
}

// min / max over the 16 lanes of a DPP row, every lane gets it (the reductions of row16_sum with another operator)
SVO_DEV int row16_min(int v) {
  v = min(v, dpp_quad<0xB1>(v)); v = min(v, dpp_quad<0x4E>(v)); v = min(v, dpp_quad<0x141>(v)); v = min(v, dpp_quad<0x140>(v));
  return v;
}
SVO_DEV int row16_max(int v) {
  v = max(v, dpp_quad<0xB1>(v)); v = max(v, dpp_quad<0x4E>(v)); v = max(v, dpp_quad<0x141>(v)); v = max(v, dpp_quad<0x140>(v));
  return v;
}

// zmssd_8x8 with the candidate's 8x8 patch taken from an LDS window: `row0` = window row of the patch's first row,
// ox = byte offset of its first column (any value 0 .. WIN_PITCH - 8; dword-aligned reads + v_alignbyte)
SVO_DEV int zmssd_8x8_win(const uint8_t* row0, int ox, const uint32_t* patch_words, int sumA, int sumAA) {
  uint32_t sumB = 0, sumBB = 0, sumAB = 0;
  const int sh = ox & 3;
  const uint32_t* base = reinterpret_cast<const uint32_t*>(row0 + (ox & ~3));
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const uint32_t* r = base + y * (WIN_PITCH / 4);
    const uint32_t d0 = r[0], d1 = r[1], d2 = r[2];       // d2 may belong to the next row (or the pad): unused when sh == 0
    const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
    const uint32_t a0 = patch_words[2 * y], a1 = patch_words[2 * y + 1];
    sumB = __builtin_amdgcn_udot4(w0, 0x01010101u, sumB, false);
    sumB = __builtin_amdgcn_udot4(w1, 0x01010101u, sumB, false);
    sumBB = __builtin_amdgcn_udot4(w0, w0, sumBB, false);
    sumBB = __builtin_amdgcn_udot4(w1, w1, sumBB, false);
    sumAB = __builtin_amdgcn_udot4(w0, a0, sumAB, false);
    sumAB = __builtin_amdgcn_udot4(w1, a1, sumAB, false);
  }
  const int sB = (int)sumB, sBB = (int)sumBB, sAB = (int)sumAB;
  return sumAA - 2 * sAB + sBB - (sumA * sumA - 2 * sumA * sB + sB * sB) / 64;
}

// Four seeds per wave, 16 per 256-thread block.  Three phases per wave:
//   A  one seed at a time, all 64 lanes: warp::warpAffine of the 10x10 reference patch into LDS (lanes = pixels)
//   B  the four seeds at once, 16 lanes each: ZMSSD search along the epipolar line (lane = candidate step; the
//      usual segment has ~11 steps, so a whole wave per seed left 5/6 of the lanes idle)
//   C  the four seeds at once, 16 lanes each: align2D / align1D (lane = 4 pixels of the 8x8 patch)
constexpr int SEEDS_PER_WAVE = 4;
constexpr int SEEDS_PER_BLOCK = 16;

// (the body: workgroup `block` of 256 threads, seeds [16 * block, 16 * block + 16); no block-level barrier inside -- every wave
// works on its own four seeds)
SVO_DEV void df_search_block(const DfFrame& fr, const uint8_t* __restrict__ ref_base, size_t ref_pyr_bytes,
                             const uint8_t* __restrict__ cur_pyr, int n, const int32_t* __restrict__ level,
                             SeedRec* __restrict__ recs, uint32_t* __restrict__ pwb_t, int n_pad, int block) {
  __shared__ __attribute__((aligned(16))) uint8_t s_pwb[SEEDS_PER_BLOCK][112];
  __shared__ __attribute__((aligned(16))) uint32_t s_patch[SEEDS_PER_BLOCK][16];
  __shared__ double s_px[SEEDS_PER_BLOCK][2];
  __shared__ int s_do[SEEDS_PER_BLOCK], s_nz[SEEDS_PER_BLOCK];
  __shared__ __attribute__((aligned(16))) uint8_t s_win[SEEDS_PER_BLOCK * WIN_BYTES + 16];   // image windows (+ pad: see zmssd_8x8_win)
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform -> SGPRs
  const int lane = threadIdx.x & 63;
  const int i0 = (block * 4 + wib) * SEEDS_PER_WAVE;
  if (i0 >= n) return;                     // wave-uniform; no block-level barrier is used below
  const Cam cam = fr.cam;
  // the wave's four records (384 contiguous bytes) and levels go to LDS in one round trip: every phase below reads
  // its per-seed parameters from there instead of paying a memory latency per dependent field access
  __shared__ __attribute__((aligned(16))) SeedRec s_rec[SEEDS_PER_BLOCK];
  __shared__ int s_level[SEEDS_PER_BLOCK];
  {
    const int n_here = n - i0 < SEEDS_PER_WAVE ? n - i0 : SEEDS_PER_WAVE;
    static_assert(sizeof(SeedRec) == 96, "six 16-byte words per record");
    const uint4* src = reinterpret_cast<const uint4*>(recs + i0);
    uint4* dst = reinterpret_cast<uint4*>(s_rec + wib * SEEDS_PER_WAVE);
    if (lane < 6 * n_here) dst[lane] = src[lane];
    if (lane >= 32 && lane < 32 + n_here) s_level[wib * SEEDS_PER_WAVE + lane - 32] = level[i0 + lane - 32];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

  // ---------------- phase A: warp the reference patches ----------------
  bool any_search = false;
  for (int sidx = 0; sidx < SEEDS_PER_WAVE; ++sidx) {
    const int i = i0 + sidx;
    if (i >= n) break;                                   // wave-uniform
    const int slot = wib * SEEDS_PER_WAVE + sidx;
    const SeedRec* rp = s_rec + slot;
    const int path = rp->path;
    if (lane == 0) { s_do[slot] = (path == 0 || path == 3) ? 1 : 0; s_nz[slot] = 0; s_px[slot][0] = rp->uv0[0]; s_px[slot][1] = rp->uv0[1]; }
    any_search |= path == 1;
  }
  {
    // 16 lanes per seed, 7 samples per lane: the seed's warp parameters are read once per lane
    const int g = lane >> 4, cl = lane & 15;
    const int i = i0 + g;
    const bool have = i < n;
    const SeedRec* rp = s_rec + wib * SEEDS_PER_WAVE + (have ? g : 0);
    const int path = have ? rp->path : -1;
    if (path == 0 || path == 1 || path == 3) {
      const uint8_t* ref_pyr = ref_base + (size_t)rp->pad * ref_pyr_bytes;   // pad = reference keyframe slot
      const int level_ref = s_level[wib * SEEDS_PER_WAVE + g];
      const int search_level = rp->search_level;
      // warp::warpAffine of the 10x10 reference patch (matcher.cpp:83-116)
      const int rcols = cam.width >> level_ref, rrows = cam.height >> level_ref;
      const uint8_t* img_ref = ref_pyr + fr.ref_level_off[level_ref];
      const float a00 = rp->a00, a01 = rp->a01, a10 = rp->a10, a11 = rp->a11, prx = rp->prx, pry = rp->pry;
      const bool warp_nan = rp->warp_nan != 0;
      const float lscale = (float)(1 << search_level);
      uint8_t* pwb = s_pwb[wib * SEEDS_PER_WAVE + g];
      // The source footprint of the patch: the four corners bound every sample (each f32 operation of the affine map is
      // monotone in the patch coordinate), the samples that are interpolated lie in [0, rcols-1) x [0, rrows-1).
      uint8_t* win = s_win + (wib * SEEDS_PER_WAVE + g) * WIN_BYTES;
      int wx0 = 0, wy0 = 0;
      bool use_win = false;
      {
        float xmin = 0, xmax = 0, ymin = 0, ymax = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float ppx = (float)((c & 1) ? 4 : -5) * lscale, ppy = (float)((c & 2) ? 4 : -5) * lscale;
          const float qx = (a00 * ppx + a01 * ppy) + prx, qy = (a10 * ppx + a11 * ppy) + pry;
          xmin = c == 0 ? qx : fminf(xmin, qx); xmax = c == 0 ? qx : fmaxf(xmax, qx);
          ymin = c == 0 ? qy : fminf(ymin, qy); ymax = c == 0 ? qy : fmaxf(ymax, qy);
        }
        // (comparisons with a NaN are false: a NaN box, like a NaN warp, keeps the direct path, which zeroes the patch)
        if (!warp_nan && xmin > -1e6f && ymin > -1e6f && xmax < 1e6f && ymax < 1e6f) {
          wx0 = max((int)floorf(xmin), 0); wy0 = max((int)floorf(ymin), 0);
          const int wx1 = min((int)floorf(xmax), rcols - 2) + 1, wy1 = min((int)floorf(ymax), rrows - 2) + 1;
          const int ww = wx1 - wx0 + 1, wh = wy1 - wy0 + 1;
          if (ww > 0 && wh > 0 && ww <= WIN_PITCH && wh <= WIN_ROWS) {
            use_win = true;
            win_fill(img_ref + (size_t)wy0 * rcols + wx0, rcols, wh, cl, win);
          }
        }
      }
      // Two passes so that the 7 x 2 loads of a lane are in flight together (one memory latency instead of seven):
      // first every sample's address and weights -- a sample that is not interpolated reads offset 0 of the level,
      // which is always valid memory -- then the arithmetic of vk::interpolateMat_8u (I/vision.h:19-36) in its order.
      float w00[7], w01[7], w10[7], w11[7];
      unsigned r0[7], r1[7];
      bool ok[7];
      int woff[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int k = cl + 16 * j;
        const int yy = k / 10, xx = k - yy * 10;
        float ppx = (float)(xx - 5), ppy = (float)(yy - 5);
        ppx *= lscale;
        ppy *= lscale;
        const float qx = (a00 * ppx + a01 * ppy) + prx;
        const float qy = (a10 * ppx + a11 * ppy) + pry;
        // the reference keeps the previous seed's patch when the inverse warp is NaN and reads out of bounds when
        // qx/qy are NaN (inf inverse of a singular A); such samples are 0 here
        ok[j] = k < 100 && !warp_nan && qx >= 0 && qy >= 0 && qx < rcols - 1 && qy < rrows - 1;
        const float u = ok[j] ? qx : 0.0f, v = ok[j] ? qy : 0.0f;
        const int x = (int)floorf(u), y = (int)floorf(v);
        const float subpix_x = u - x, subpix_y = v - y;
        w00[j] = (1.0f - subpix_x) * (1.0f - subpix_y);
        w01[j] = (1.0f - subpix_x) * subpix_y;
        w10[j] = subpix_x * (1.0f - subpix_y);
        w11[j] = 1.0f - w00[j] - w01[j] - w10[j];
        woff[j] = ok[j] ? (y - wy0) * WIN_PITCH + (x - wx0) : 0;
        if (!use_win) {
          const uint8_t* ptr = img_ref + y * rcols + x;
          typedef unsigned short __attribute__((aligned(1))) u16u;
          r0[j] = *reinterpret_cast<const u16u*>(ptr);            // the two neighbours of a row in one unaligned load
          r1[j] = *reinterpret_cast<const u16u*>(ptr + rcols);
        }
      }
      if (use_win) {
        // the window of this seed was written by its own 16 lanes: order the LDS writes before the gathers
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      }
      if (use_win) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const uint8_t* q = win + woff[j];
          r0[j] = (unsigned)q[0] | ((unsigned)q[1] << 8);
          r1[j] = (unsigned)q[WIN_PITCH] | ((unsigned)q[WIN_PITCH + 1] << 8);
        }
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int k = cl + 16 * j;
        const float val = w00[j] * (float)(r0[j] & 0xff) + w01[j] * (float)(r1[j] & 0xff) + w10[j] * (float)(r0[j] >> 8) +
                          w11[j] * (float)(r1[j] >> 8);
        if (k < 100) pwb[k] = ok[j] ? (uint8_t)val : (uint8_t)0;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // createPatchFromPatchWithBorder (matcher.cpp:138-147): lane = pixel of the 8x8 patch
#pragma unroll
  for (int sidx = 0; sidx < SEEDS_PER_WAVE; ++sidx) {
    const int slot = wib * SEEDS_PER_WAVE + sidx;
    reinterpret_cast<uint8_t*>(s_patch[slot])[lane] = s_pwb[slot][((lane >> 3) + 1) * 10 + (lane & 7) + 1];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

  // ---------------- phase B: epipolar ZMSSD search, 16 lanes per seed ----------------
  if (any_search) {                                      // wave-uniform
    const int g = lane >> 4, cl = lane & 15, gbase = lane & 48;
    const int gi = i0 + g;
    const int slot = wib * SEEDS_PER_WAVE + g;
    const bool have = gi < n;
    const SeedRec* rp = s_rec + wib * SEEDS_PER_WAVE + (have ? g : 0);
    const bool live = have && rp->path == 1;
    const int search_level = rp->search_level;
    const int ccols = cam.width >> search_level;
    const double inv_level = 1.0 / (1 << search_level);
    const uint8_t* cur_img = cur_pyr + fr.cur_level_off[search_level];
    const uint32_t* patch_words = s_patch[slot];
    const double step[2] = {rp->step[0], rp->step[1]};
    const double uv0[2] = {rp->uv0[0], rp->uv0[1]};
    const int n_steps = live ? rp->n_steps + 1 : 0;      // ++n_steps (matcher.cpp:297)
    int n_max = n_steps;
    n_max = max(n_max, __shfl_xor(n_max, 16, 64));
    n_max = max(n_max, __shfl_xor(n_max, 32, 64));       // wave-uniform trip count
    // reference patch statistics (ZMSSD ctor, patch_score.h:49-61): 4 pixels per lane
    int sumA, sumAA;
    {
      const uint32_t w = patch_words[cl];
      sumA = row16_sum((int)__builtin_amdgcn_udot4(w, 0x01010101u, 0u, false));
      sumAA = row16_sum((int)__builtin_amdgcn_udot4(w, w, 0u, false));
    }
    unsigned long long best_key = ~0ull;
    int prev_tail_x = 0, prev_tail_y = 0;                // last_checked_pxi entering the chunk
    double uv_tail[2] = {uv0[0], uv0[1]};                // exact uv of step index chunk0 (slow path only)
    bool chain_exact = true;                             // uv_tail is current (no fast chunk has been taken yet)
    int n_zmssd = 0;
    for (int chunk0 = 0; chunk0 < n_max; chunk0 += 16) {
      const int idx = chunk0 + cl;
      const bool in_range = idx < n_steps;
      // The serial loop produces uv_i by repeated addition (matcher.cpp:299).  uv0 + i*step differs from it by
      // < 1e-13 relative, i.e. < 1e-10 px: if every candidate of the chunk is farther than 1e-7 px from a
      // rounding boundary of (int)(px/2^L + 0.5) the integer pixel is the same and the chain is not needed.
      double uv[2] = {uv0[0] + (double)idx * step[0], uv0[1] + (double)idx * step[1]};
      double pxs[2];
      world2cam_uv(cam, uv[0], uv[1], pxs);
      double tx = pxs[0] * inv_level + 0.5, ty = pxs[1] * inv_level + 0.5;       // division by 2^L, exact
      const bool risky = in_range && !(fabs(tx - rint(tx)) > 1e-7 && fabs(ty - rint(ty)) > 1e-7);
      const bool group_risky = ((__ballot(risky) >> gbase) & 0xffffull) != 0ull;      // uniform within the 16 lanes
      if (group_risky) {
        // exact replay of the chain for this chunk: bring the tail to chunk0, then lane l adds `step` l times
        if (!chain_exact) {
          uv_tail[0] = uv0[0]; uv_tail[1] = uv0[1];
          for (int k = 0; k < chunk0; ++k) { uv_tail[0] += step[0]; uv_tail[1] += step[1]; }
        }
        uv[0] = uv_tail[0]; uv[1] = uv_tail[1];
        for (int k = 0; k < 15; ++k) {
          if (k < cl) { uv[0] += step[0]; uv[1] += step[1]; }
        }
        uv_tail[0] = __shfl(uv[0], gbase + 15, 64) + step[0];
        uv_tail[1] = __shfl(uv[1], gbase + 15, 64) + step[1];
        chain_exact = true;
        world2cam_uv(cam, uv[0], uv[1], pxs);
        tx = pxs[0] * inv_level + 0.5; ty = pxs[1] * inv_level + 0.5;
      } else {
        chain_exact = false;
      }
      const int pxi_x = (int)tx, pxi_y = (int)ty;
      // dedupe against the previous step (matcher.cpp:306-308): it never changes the arg-min, only the
      // count of evaluations, which we keep for the work counters
      int prev_x = __shfl_up(pxi_x, 1, 64), prev_y = __shfl_up(pxi_y, 1, 64);
      if (cl == 0) { prev_x = prev_tail_x; prev_y = prev_tail_y; }
      const bool dup = (pxi_x == prev_x && pxi_y == prev_y);
      const bool inframe = is_in_frame_level(cam, pxi_x, pxi_y, 8, search_level);
      int score = 0x7fffffff;
      const bool eval = in_range && !dup && inframe;
      // bounding box of the chunk's candidate patches (uniform within the 16 lanes): if it fits, one window fill
      // replaces the 16 row loads of every candidate
      const int bx0 = row16_min(eval ? pxi_x : 0x7fffffff), bx1 = row16_max(eval ? pxi_x : -0x7fffffff);
      const int by0 = row16_min(eval ? pxi_y : 0x7fffffff), by1 = row16_max(eval ? pxi_y : -0x7fffffff);
      const bool any_eval = bx1 >= bx0;
      const bool fits = any_eval && bx1 - bx0 + 8 <= WIN_PITCH && by1 - by0 + 8 <= WIN_ROWS;
      uint8_t* win = s_win + slot * WIN_BYTES;
      if (chunk0 > 0) {                                    // wave-uniform: the window of the previous chunk has been read
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      if (fits) win_fill(cur_img + (size_t)(by0 - 4) * ccols + (bx0 - 4), ccols, by1 - by0 + 8, cl, win);
      if (eval) {
        if (fits) {
          score = zmssd_8x8_win(win + (pxi_y - by0) * WIN_PITCH, pxi_x - bx0, patch_words, sumA, sumAA);
        } else {
          const uint8_t* cp = cur_img + (pxi_y - 4) * ccols + (pxi_x - 4);
          score = zmssd_8x8(cp, ccols, patch_words, sumA, sumAA);
        }
      }
      n_zmssd += __popcll((__ballot(eval) >> gbase) & 0xffffull);
      if (eval && score < ZMSSD_THRESHOLD) {
        const unsigned long long key = ((unsigned long long)(unsigned)score << 32) | (unsigned long long)(unsigned)idx;
        if (key < best_key) best_key = key;
      }
      prev_tail_x = __shfl(pxi_x, gbase + 15, 64); prev_tail_y = __shfl(pxi_y, gbase + 15, 64);
    }
    // arg-min over the 16 lanes on (score, index): the smallest score, earliest index on ties
    unsigned long long k = best_key;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(k, o, 64);
      k = other < k ? other : k;
    }
    if (live) {
      double px_best[2] = {uv0[0], uv0[1]};
      const bool found = k != ~0ull;
      if (found) {
        // uv_best: replay the serial chain up to the winning index (uniform within the 16 lanes)
        const int best_idx = (int)(k & 0xffffffffull);
        double bu = uv0[0], bv = uv0[1];
        for (int t = 0; t < best_idx; ++t) { bu += step[0]; bv += step[1]; }
        world2cam_uv(cam, bu, bv, px_best);
      }
      if (cl == 0) { s_do[slot] = found ? 1 : 0; s_nz[slot] = n_zmssd; s_px[slot][0] = px_best[0]; s_px[slot][1] = px_best[1]; }
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

  // ---------------- hand-over to the alignment stage ----------------
  // the warped 10x10 patch goes to memory word-transposed (word w of seed i at pwb_t[w * n_pad + i]: the lane-per-seed
  // alignment kernel reads it with fully coalesced loads), the start pixel and the "align this seed" flag into the record
  {
    const int g = lane >> 4, cl = lane & 15;
    const int gi = i0 + g;
    const int slot = wib * SEEDS_PER_WAVE + g;
    const bool have = gi < n;
    const SeedRec* rp = s_rec + wib * SEEDS_PER_WAVE + (have ? g : 0);
    const int path = have ? rp->path : -1;
    const bool live = path == 0 || path == 1 || path == 3;
    const bool do_align = live && s_do[slot] != 0;
    if (do_align) {
      const uint32_t* words = reinterpret_cast<const uint32_t*>(s_pwb[slot]);
      pwb_t[(size_t)cl * n_pad + gi] = words[cl];
      if (cl < 9) pwb_t[(size_t)(cl + 16) * n_pad + gi] = words[cl + 16];
    }
    if (live && cl == 0) {
      SeedRec* rp_out = recs + gi;
      rp_out->matched = do_align ? 2 : 0;                // 2 = "to be aligned" (df_align_kernel settles it to 0 / 1)
      if (path != 3) { rp_out->step[0] = s_px[slot][0]; rp_out->step[1] = s_px[slot][1]; }   // path 3 keeps its direction there
      rp_out->n_zmssd = s_nz[slot]; rp_out->n_align = 0;
    }
  }
}

__global__ __launch_bounds__(256, 7) void df_search_kernel(
    DfFrame fr, const uint8_t* __restrict__ ref_base, size_t ref_pyr_bytes, const uint8_t* __restrict__ cur_pyr, int n,
    const int32_t* __restrict__ level, SeedRec* __restrict__ recs, uint32_t* __restrict__ pwb_t, int n_pad,
    const int* __restrict__ n_dev = nullptr) {
  if (n_dev) { const int nd = *n_dev; n = nd < n ? nd : n; }       // item count left by an earlier kernel of the stream (svo_track.hip)
  df_search_block(fr, ref_base, ref_pyr_bytes, cur_pyr, n, level, recs, pwb_t, n_pad, blockIdx.x);
}

// Sub-pixel refinement of every seed the search stage flagged: one quad per seed, the reference's serial pixel order
// (svo_align_device.h), so `converged`, the refined pixel and the iteration count equal the CPU path's bit for bit.
// ONE_D = false refines the corner features with align2D; ONE_D = true the EDGELET reference features of
// findMatchDirect with align1D (matcher.cpp:183-191; the depth filter never produces them).
// (the body: this lane is lane q = threadIdx.x & 3 of the quad of seed i; `have` = the seed exists)
template <bool ONE_D>
SVO_DEV void df_align_quad(const DfFrame& fr, const uint8_t* __restrict__ cur_pyr, int n_pad, const uint32_t* __restrict__ pwb_t,
                           SeedRec* __restrict__ recs, int i, bool have) {
  const int q = threadIdx.x & 3;
  if (__builtin_amdgcn_ballot_w64(have) == 0ull) return;
  SeedRec* rp = recs + (have ? i : 0);
  // ONE trip to memory for everything the quad needs: lane q takes one 16-byte word of the seed's record -- {step}, {search
  // level, path, status, warp_nan}, {matched, counters}, {uv0} -- and its ten words of the warped patch, all asked for at
  // once; the fields then go round the quad by DPP.  (Round 3 read `path` and `matched` first, the other fields and the
  // patch words -- only where the seed is to be aligned -- after them: two dependent trips to HBM at the head of every
  // wave, in a stage whose set-up part is a third of its time: 79 of 264 us at 1 M seeds, tools/df_iter_probe.py.)
  static_assert(offsetof(SeedRec, step) == 16 && offsetof(SeedRec, search_level) == 64 && offsetof(SeedRec, path) == 68 &&
                offsetof(SeedRec, matched) == 80 && offsetof(SeedRec, uv0) == 0, "record words read below");
  const int word = q == 0 ? 1 : (q == 1 ? 4 : (q == 2 ? 5 : 0));
  const uint4 mine_w = reinterpret_cast<const uint4*>(rp)[word];
  QuadPatch qp;
#pragma unroll
  for (int k = 0; k < 10; ++k) qp.b[k] = pwb_t[(size_t)(5 * q + k) * n_pad + (have ? i : 0)];     // (read-only scratch: any content is valid)
  const int search_level_r = quad_bcast<1>((int)mine_w.x);
  const int path = have ? quad_bcast<1>((int)mine_w.y) : -1;
  const int matched_r = quad_bcast<2>((int)mine_w.x);
  const bool mine = ONE_D ? path == 3 : (path == 0 || path == 1);
  const bool do_align = mine && matched_r == 2;
  const double step0 = __hiloint2double(quad_bcast<0>((int)mine_w.y), quad_bcast<0>((int)mine_w.x));
  const double step1 = __hiloint2double(quad_bcast<0>((int)mine_w.w), quad_bcast<0>((int)mine_w.z));
  const double uv00 = __hiloint2double(quad_bcast<3>((int)mine_w.y), quad_bcast<3>((int)mine_w.x));
  const double uv01 = __hiloint2double(quad_bcast<3>((int)mine_w.w), quad_bcast<3>((int)mine_w.z));
  if (__builtin_amdgcn_ballot_w64(do_align) == 0ull) return;      // wave-uniform
  const int search_level = do_align ? search_level_r : 0;
  const int ccols = fr.cam.width >> search_level, crows = fr.cam.height >> search_level;
  const uint8_t* cur_img = cur_pyr + fr.cur_level_off[search_level];
  patch_from_border(qp);
  double px_cur[2] = {0.0, 0.0};
  float dir0 = 0.0f, dir1 = 0.0f;
  if (do_align) {
    if (ONE_D) { px_cur[0] = uv00; px_cur[1] = uv01; dir0 = (float)step0; dir1 = (float)step1; }
    else { px_cur[0] = step0; px_cur[1] = step1; }
  }
  const double inv_scale = 1.0 / (1 << search_level);    // exact: a power of two
  double us = px_cur[0] * inv_scale, vs = px_cur[1] * inv_scale;
  int n_align = 0;
  bool res;
  if (ONE_D) {
    double h_inv;
    res = align1d_quad(cur_img, ccols, crows, ccols, dir0, dir1, qp, fr.align_max_iter, do_align, &us, &vs, &h_inv, &n_align);
  } else {
    res = align2d_quad(cur_img, ccols, crows, ccols, qp, fr.align_max_iter, do_align, &us, &vs, &n_align);
  }
  if (do_align && q == 0) {
    if (res || fr.keep_px_on_failure) {
      px_cur[0] = us * (1 << search_level);
      px_cur[1] = vs * (1 << search_level);
    }
    rp->matched = res ? 1 : 0;
    rp->step[0] = px_cur[0]; rp->step[1] = px_cur[1];
    rp->n_align = n_align;
  }
}

template <bool ONE_D>
__global__ __launch_bounds__(ALIGN_BLOCK) void df_align_kernel(DfFrame fr, const uint8_t* __restrict__ cur_pyr, int n, int n_pad,
                                                               const uint32_t* __restrict__ pwb_t, SeedRec* __restrict__ recs,
                                                               const int* __restrict__ n_dev = nullptr) {
  if (n_dev) { const int nd = *n_dev; n = nd < n ? nd : n; }
  const int i = blockIdx.x * ALIGN_PATCHES + (threadIdx.x >> 2);
  df_align_quad<ONE_D>(fr, cur_pyr, n_pad, pwb_t, recs, i, i < n);
}

// ---- the warp / search and alignment stages over the candidates of SEVERAL cameras (svo_hip_tracker_group): camera c's
// records are [c * cap, c * cap + n_c) of one record array (cap a multiple of the 16 items a block takes, so a block never
// straddles two cameras), n_c = counters[c * counter_stride] as the camera's planning kernel left it, its current image is
// slot c of the frame pyramid batch; the reference keyframe of an item is the slot in its record, as always.
__global__ __launch_bounds__(256, 7) void df_search_cams_kernel(
    DfFrame fr, const uint8_t* __restrict__ ref_base, size_t ref_pyr_bytes, const uint8_t* __restrict__ cur_base, size_t cur_pyr_bytes,
    int cap, const int* __restrict__ counters, int counter_stride, const int32_t* __restrict__ level, SeedRec* __restrict__ recs,
    uint32_t* __restrict__ pwb_t, int n_pad) {
  const int c = (int)(((long long)blockIdx.x * SEEDS_PER_BLOCK) / cap);
  int n_c = counters[(size_t)c * counter_stride];
  n_c = n_c < cap ? n_c : cap;
  df_search_block(fr, ref_base, ref_pyr_bytes, cur_base + (size_t)c * cur_pyr_bytes, c * cap + n_c, level, recs, pwb_t, n_pad, blockIdx.x);
}

template <bool ONE_D>
__global__ __launch_bounds__(ALIGN_BLOCK) void df_align_cams_kernel(DfFrame fr, const uint8_t* __restrict__ cur_base, size_t cur_pyr_bytes, int cap,
                                                                    const int* __restrict__ counters, int counter_stride, int n_pad,
                                                                    const uint32_t* __restrict__ pwb_t, SeedRec* __restrict__ recs) {
  const int i = blockIdx.x * ALIGN_PATCHES + (threadIdx.x >> 2);
  const int c = (int)(((long long)blockIdx.x * ALIGN_PATCHES) / cap);
  int n_c = counters[(size_t)c * counter_stride];
  n_c = n_c < cap ? n_c : cap;
  df_align_quad<ONE_D>(fr, cur_base + (size_t)c * cur_pyr_bytes, n_pad, pwb_t, recs, i, i < c * cap + n_c);
}

// EVENTS (device-resident seed batches, svo_hip_seed_batch_*): the kernel also counts, per block, the seeds whose outcome
// the HOST has to hear about -- converged and NaN seeds (callback / erase, depth_filter.cpp:310-337) and, on keyframes
// (report_updated), every updated seed (its px_cur marks the detector grid, :302-306) -- clears their `alive` flag where
// the reference erases them, and adds the block's status histogram to ev_hist[8] (slot status + 1; slot 0 = erased).
//
// (the per-seed body: seed i of the arrays, its record `rc`; returns the seed's status)
SVO_DEV int df_finalize_seed(const Cam& cam, const double* T_cur_ref, const double* T_ref_cur, const double* T_ref_inv,
                             double px_error_angle, double conv_thresh, int i, const SeedRec& rc, const double* __restrict__ f,
                             float* __restrict__ sa, float* __restrict__ sb, float* __restrict__ smu,
                             const float* __restrict__ sz_range, float* __restrict__ ssigma2, int32_t* __restrict__ status,
                             double* __restrict__ z_out, double* __restrict__ xyz_world, int32_t* __restrict__ n_zmssd_out,
                             int32_t* __restrict__ n_align_out, double* __restrict__ px_cur_out,
                             int32_t* __restrict__ search_level_out) {
  int st = rc.status;
  {
    double z = 0.0;
    if (rc.path >= 0) {
      const double fi[3] = {f[3 * (size_t)i], f[3 * (size_t)i + 1], f[3 * (size_t)i + 2]};
      SeedState seed = {sa[i], sb[i], smu[i], sz_range[i], ssigma2[i]};
      bool matched = false;
      if (rc.matched) {
        double fc[3];
        cam2world(cam, rc.step[0], rc.step[1], fc);
        matched = depth_from_triangulation(T_cur_ref, fi, fc, &z);
      }
      if (!matched) {
        seed.b += 1.0f;                                   // depth_filter.cpp:286
        st = SVO_HIP_SEED_NO_MATCH;
        z = 0.0;
      } else {
        // ---- computeTau + updateSeed + convergence (depth_filter.cpp:294-337)
        const double tau = compute_tau(T_ref_cur, fi, z, px_error_angle);
        const double zmt = z - tau;
        const double tau_inverse = 0.5 * (1.0 / (0.0000001 < zmt ? zmt : 0.0000001) - 1.0 / (z + tau));
        update_seed((float)(1. / z), (float)(tau_inverse * tau_inverse), &seed);
        if ((double)sqrtf(seed.sigma2) < seed.z_range / conv_thresh) {
          st = SVO_HIP_SEED_CONVERGED;
          if (xyz_world) {
            const double im = 1.0 / seed.mu;
            const double pfw[3] = {fi[0] * im, fi[1] * im, fi[2] * im};
            double xw[3];
            se3_act(T_ref_inv, pfw, xw);
            xyz_world[3 * (size_t)i] = xw[0]; xyz_world[3 * (size_t)i + 1] = xw[1]; xyz_world[3 * (size_t)i + 2] = xw[2];
          }
        } else if (rc.z_inv_min != rc.z_inv_min) {
          st = SVO_HIP_SEED_NAN;
        } else {
          st = SVO_HIP_SEED_UPDATED;
        }
      }
      sa[i] = seed.a; sb[i] = seed.b; smu[i] = seed.mu; ssigma2[i] = seed.sigma2;
    }
    status[i] = st;
    if (z_out) z_out[i] = z;
    if (n_zmssd_out) n_zmssd_out[i] = rc.n_zmssd;
    if (n_align_out) n_align_out[i] = rc.n_align;
    // Matcher::px_cur_ / search_level_ of this seed's findEpipolarMatchDirect call (matcher.cpp:345-346): what
    // DepthFilter::updateSeeds hands to feature_detector_->setGridOccpuancy on keyframes (depth_filter.cpp:302-306)
    if (px_cur_out) {
      const bool updated = st >= SVO_HIP_SEED_UPDATED;
      const double qnan = __longlong_as_double(0x7ff8000000000000ll);
      px_cur_out[2 * (size_t)i] = updated ? rc.step[0] : qnan;
      px_cur_out[2 * (size_t)i + 1] = updated ? rc.step[1] : qnan;
    }
    if (search_level_out) search_level_out[i] = rc.path >= 0 ? rc.search_level : -1;
  }
  return st;
}

// (the block's part of the EVENTS bookkeeping: every thread of the 256-thread block calls it; `block` = the block's index
// within the batch)
SVO_DEV void df_finalize_events(int st, int i, int n, int block, uint8_t* __restrict__ alive, int report_updated,
                                int* __restrict__ ev_block_count, int* __restrict__ ev_hist) {
  const bool gone = st == SVO_HIP_SEED_CONVERGED || st == SVO_HIP_SEED_NAN;      // the reference erases these (:330, :336)
  const bool ev = gone || (report_updated && st >= SVO_HIP_SEED_UPDATED);
  if (gone) alive[i] = 0;
  __shared__ int s_ev[4];
  __shared__ int s_hist[8];
  if (threadIdx.x < 8) s_hist[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long m = __ballot(ev);
  if ((threadIdx.x & 63) == 0) s_ev[threadIdx.x >> 6] = __popcll(m);
  if (i < n) atomicAdd(&s_hist[st + 1], 1);
  __syncthreads();
  if (threadIdx.x == 0) ev_block_count[block] = s_ev[0] + s_ev[1] + s_ev[2] + s_ev[3];
  if (threadIdx.x < 8 && s_hist[threadIdx.x]) atomicAdd(&ev_hist[threadIdx.x], s_hist[threadIdx.x]);
}

template <bool EVENTS>
__global__ __launch_bounds__(256) void df_finalize_kernel(
    DfFrame fr, int n, const double* __restrict__ f, const SeedRec* __restrict__ recs, float* __restrict__ sa,
    float* __restrict__ sb, float* __restrict__ smu, const float* __restrict__ sz_range, float* __restrict__ ssigma2,
    int32_t* __restrict__ status, double* __restrict__ z_out, double* __restrict__ xyz_world,
    int32_t* __restrict__ n_zmssd_out, int32_t* __restrict__ n_align_out, double* __restrict__ px_cur_out,
    int32_t* __restrict__ search_level_out, uint8_t* __restrict__ alive = nullptr, int report_updated = 0,
    int* __restrict__ ev_block_count = nullptr, int* __restrict__ ev_hist = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int st = SVO_HIP_SEED_ERASED;
  if (i < n) {
    const SeedRec rc = recs[i];                            // (a copy: the whole record is asked for at once)
    st = df_finalize_seed(fr.cam, fr.T_cur_ref, fr.T_ref_cur, fr.T_ref_inv, fr.px_error_angle, fr.conv_thresh, i, rc, f, sa, sb,
                          smu, sz_range, ssigma2, status, z_out, xyz_world, n_zmssd_out, n_align_out, px_cur_out, search_level_out);
  }
  if (EVENTS) df_finalize_events(st, i, n, blockIdx.x, alive, report_updated, ev_block_count, ev_hist);
}

// Matcher::findEpipolarMatchDirect called directly (explicit depth interval): the tail of the function,
// matcher.cpp:340-354 -- cam2world of the refined pixel, depthFromTriangulation.
__global__ __launch_bounds__(256) void epi_finalize_kernel(
    DfFrame fr, int n, const double* __restrict__ f, const SeedRec* __restrict__ recs, uint8_t* __restrict__ ok_out,
    double* __restrict__ depth_out, double* __restrict__ px_cur_out, int32_t* __restrict__ search_level_out,
    int32_t* __restrict__ n_zmssd_out, int32_t* __restrict__ n_align_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SeedRec rc = recs[i];
  bool ok = false;
  double z = 0.0;
  if (rc.path >= 0 && rc.matched) {
    const double fi[3] = {f[3 * (size_t)i], f[3 * (size_t)i + 1], f[3 * (size_t)i + 2]};
    double fc[3];
    cam2world(fr.cam, rc.step[0], rc.step[1], fc);
    ok = depth_from_triangulation(fr.T_cur_ref, fi, fc, &z);
  }
  ok_out[i] = ok ? 1 : 0;
  if (depth_out) depth_out[i] = ok ? z : 0.0;
  if (px_cur_out) {
    const bool have = rc.path >= 0 && rc.matched;        // px_cur_ is assigned when the alignment succeeded (:345)
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    px_cur_out[2 * (size_t)i] = have ? rc.step[0] : qnan;
    px_cur_out[2 * (size_t)i + 1] = have ? rc.step[1] : qnan;
  }
  if (search_level_out) search_level_out[i] = rc.path >= 0 ? rc.search_level : -1;
  if (n_zmssd_out) n_zmssd_out[i] = rc.n_zmssd;
  if (n_align_out) n_align_out[i] = rc.n_align;
}

// ---- Matcher::findMatchDirect over n (map point, reference feature) pairs (S/matcher.cpp:156-202) ----
// (MdFrame, md_geometry_item: svo_match_device.h)
// thread per item: frame test, depth, affine warp, search level -> SeedRec (path 0 = align2D, 3 = align1D)
__global__ __launch_bounds__(256) void md_geometry_kernel(
    MdFrame fr, int n, const double* __restrict__ T_ref_w /*[n_kf][7]*/, const int32_t* __restrict__ kf_slot,
    const double* __restrict__ px_ref, const double* __restrict__ f_ref, const int32_t* __restrict__ level,
    const double* __restrict__ pt_pos, const uint8_t* __restrict__ edgelet, const double* __restrict__ grad,
    const double* __restrict__ px_cur, SeedRec* __restrict__ recs, const double* __restrict__ T_cur_w_dev,
    const int* __restrict__ n_dev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // the item count and the frame pose may come from earlier kernels of the same stream; the launch then covers the
  // capacity and the items past the count become records no later stage touches
  if (n_dev && i >= *n_dev) { recs[i] = md_dead_record(); return; }
  if (T_cur_w_dev) {
#pragma unroll
    for (int k = 0; k < 7; ++k) fr.T_cur_w[k] = T_cur_w_dev[k];
  }
  const double g[2] = {grad ? grad[2 * (size_t)i] : 1.0, grad ? grad[2 * (size_t)i + 1] : 0.0};
  recs[i] = md_geometry_item(fr, T_ref_w, kf_slot[i], level[i], px_ref + 2 * (size_t)i, f_ref + 3 * (size_t)i, pt_pos + 3 * (size_t)i,
                             edgelet && edgelet[i], g, px_cur + 2 * (size_t)i);
}

__global__ void md_finalize_kernel(int n, const SeedRec* __restrict__ recs, double* __restrict__ px_cur,
                                   uint8_t* __restrict__ success, int32_t* __restrict__ search_level) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SeedRec& rc = recs[i];
  if (rc.path >= 0) {            // the frame test passed: px_cur is always rewritten (matcher.cpp:200)
    px_cur[2 * (size_t)i] = rc.step[0];
    px_cur[2 * (size_t)i + 1] = rc.step[1];
  }
  success[i] = rc.path >= 0 && rc.matched;
  if (search_level) search_level[i] = rc.search_level;
}

// the alignment stage over records [0, n)
template <bool ONE_D>
void launch_df_align(svo_hip_ctx* ctx, const DfFrame& fr, const uint8_t* cur_img, int n, int n_pad, const uint32_t* pwb_t, SeedRec* recs,
                     const int* n_dev = nullptr) {
  hipLaunchKernelGGL(df_align_kernel<ONE_D>, dim3((n + ALIGN_PATCHES - 1) / ALIGN_PATCHES), dim3(ALIGN_BLOCK), 0, ctx->stream, fr, cur_img, n,
                     n_pad, pwb_t, recs, n_dev);
}

int grid_for(int n, int block) {
  long long g = ((long long)n + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

// ---- ordered compaction of the converged seeds into packed records (the payload of the multi-GPU gather) ----
// record = {seed id, mu, sigma2, x, y, z} as f64; order = ascending seed index (the order of the reference's callbacks)
namespace {
__global__ __launch_bounds__(256) void conv_count_kernel(int n, const int32_t* __restrict__ status, int* __restrict__ block_count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool c = i < n && status[i] == SVO_HIP_SEED_CONVERGED;
  const unsigned long long m = __ballot(c);
  __shared__ int s_w[4];
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive scan of the block counts in place, one workgroup; total -> *count.  Wave-level scans by lane shuffles and one
// pass over the 16 wave totals per 1024 counts (round 3: a Hillis-Steele scan in LDS, 20 barriers per 1024 counts: 8.3 us
// for the 3 907 blocks of 1 M seeds)
__global__ __launch_bounds__(1024) void conv_scan_kernel(int n_blocks, int* __restrict__ block_count, int* __restrict__ count) {
  __shared__ int s_part[16];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n_blocks; base += 1024) {
    const int k = base + threadIdx.x;
    const int v = k < n_blocks ? block_count[k] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += t; }
    if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += s_part[w];
    const int carry = s_carry;
    if (k < n_blocks) block_count[k] = carry + wave_off + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = s_carry;
}

__global__ __launch_bounds__(256) void conv_scatter_kernel(int n, long long id_offset, const int32_t* __restrict__ status,
                                                           const float* __restrict__ mu, const float* __restrict__ sigma2,
                                                           const double* __restrict__ xyz, const int* __restrict__ block_offset,
                                                           double* __restrict__ records) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool c = i < n && status[i] == SVO_HIP_SEED_CONVERGED;
  const unsigned long long m = __ballot(c);
  __shared__ int s_w[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_w[wave] = __popcll(m);
  __syncthreads();
  if (!c) return;
  int off = block_offset[blockIdx.x];
  for (int w = 0; w < wave; ++w) off += s_w[w];
  off += __popcll(m & ((1ull << lane) - 1ull));
  double* r = records + 6 * (size_t)off;
  r[0] = (double)(id_offset + i); r[1] = (double)mu[i]; r[2] = (double)sigma2[i];
  r[3] = xyz[3 * (size_t)i]; r[4] = xyz[3 * (size_t)i + 1]; r[5] = xyz[3 * (size_t)i + 2];
}
// up to `cap` of the packed records and the clamped count into this rank's block of the gather buffers
__global__ void records_clamp_copy_kernel(const double* __restrict__ rec, const int* __restrict__ count, int cap,
                                          double* __restrict__ dst, int32_t* __restrict__ dst_count) {
  const int n = min(*count, cap);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 6 * n; i += gridDim.x * blockDim.x) dst[i] = rec[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) *dst_count = *count;      // the true count: > cap tells the caller about the overflow
}
}  // namespace

extern "C" {

int svo_hip_align2d_batch_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* cur, int slot, int level, int n,
                              const uint8_t* pwb_dev, const uint8_t* ref_patch_dev, int n_iter, double* px_dev,
                              uint8_t* converged_dev, int32_t* iters_dev) {
  if (!ctx || !cur) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, slot >= 0 && slot < cur->batch && level >= 0 && level < cur->n_levels);
  SVO_REQUIRE(ctx, n >= 0 && n_iter >= 0);
  if (n == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, pwb_dev && px_dev && converged_dev);
  // ref_patch_dev == NULL: the 8x8 patch is the interior of the bordered one (createPatchFromPatchWithBorder,
  // matcher.cpp:138-147), which is how every caller of the reference fills it
  SVO_REQUIRE(ctx, ((uintptr_t)pwb_dev & 3) == 0 && ((uintptr_t)ref_patch_dev & 3) == 0);
  const uint8_t* img = cur->base + (size_t)slot * cur->pyr_bytes + cur->level_offset[level];
  hipLaunchKernelGGL(align2d_kernel, dim3((n + ALIGN_PATCHES - 1) / ALIGN_PATCHES), dim3(ALIGN_BLOCK), 0, ctx->stream, img,
                     cur->width >> level, cur->height >> level, n, pwb_dev, ref_patch_dev, n_iter, px_dev, converged_dev,
                     iters_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

int svo_hip_align1d_batch_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* cur, int slot, int level, int n,
                              const uint8_t* pwb_dev, const float* dir_dev, int n_iter, double* px_dev,
                              uint8_t* converged_dev, double* h_inv_dev, int32_t* iters_dev) {
  if (!ctx || !cur) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, slot >= 0 && slot < cur->batch && level >= 0 && level < cur->n_levels);
  SVO_REQUIRE(ctx, n >= 0 && n_iter >= 0);
  if (n == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, pwb_dev && dir_dev && px_dev && converged_dev);
  const uint8_t* img = cur->base + (size_t)slot * cur->pyr_bytes + cur->level_offset[level];
  SVO_REQUIRE(ctx, ((uintptr_t)pwb_dev & 3) == 0);
  hipLaunchKernelGGL(align1d_kernel, dim3((n + ALIGN_PATCHES - 1) / ALIGN_PATCHES), dim3(ALIGN_BLOCK), 0, ctx->stream, img,
                     cur->width >> level, cur->height >> level, n, pwb_dev, (const uint8_t*)nullptr, dir_dev, n_iter, px_dev,
                     converged_dev, h_inv_dev, iters_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

int svo_hip_align2d_batch(svo_hip_ctx* ctx, const svo_hip_pyramid* cur, int slot, int level, int n,
                          const uint8_t* pwb, const uint8_t* ref_patch, int n_iter, double* px, uint8_t* converged,
                          int32_t* iters) {
  if (!ctx || !cur) return SVO_HIP_ERR_INVALID;
  if (n == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, n > 0 && pwb && px && converged);
  // one device block from the context's staging area: px(16) pwb(100) patch(64) iters(4) converged(1) per patch
  const size_t N = (size_t)n;
  const size_t o_px = 0, o_pwb = o_px + 16 * N, o_rp = o_pwb + 100 * N, o_it = o_rp + 64 * N, o_cv = o_it + 4 * N,
               total = o_cv + N;
  char* d = nullptr;
  int rc = svo_ctx_staging(ctx, total, &d);
  if (rc == SVO_HIP_OK) rc = svo_hip_memcpy_h2d(ctx, d + o_pwb, pwb, 100 * N);
  if (rc == SVO_HIP_OK && ref_patch) rc = svo_hip_memcpy_h2d(ctx, d + o_rp, ref_patch, 64 * N);
  if (rc == SVO_HIP_OK) rc = svo_hip_memcpy_h2d(ctx, d + o_px, px, 16 * N);
  if (rc == SVO_HIP_OK)
    rc = svo_hip_align2d_batch_dev(ctx, cur, slot, level, n, (const uint8_t*)(d + o_pwb),
                                   ref_patch ? (const uint8_t*)(d + o_rp) : nullptr, n_iter, (double*)(d + o_px),
                                   (uint8_t*)(d + o_cv), (int32_t*)(d + o_it));
  if (rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, px, d + o_px, 16 * N);
  if (rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, converged, d + o_cv, N);
  if (rc == SVO_HIP_OK && iters) rc = svo_hip_memcpy_d2h(ctx, iters, d + o_it, 4 * N);
  return rc;
}

int svo_hip_camera_batch(svo_hip_ctx* ctx, const svo_hip_camera* cam, int n, const double* xyz, const double* uv,
                         const double* px, const int32_t* obs, int boundary, int level, double* px_of_xyz,
                         double* px_of_uv, double* f_of_px, uint8_t* in_frame) {
  if (!ctx || !cam) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n >= 0 && boundary >= 0 && level < SVO_HIP_MAX_LEVELS);
  if (n == 0) return SVO_HIP_OK;
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = (size_t)n;
  // one staging block: xyz(24) uv(16) px(16) px_of_xyz(16) px_of_uv(16) f_of_px(24) obs(8) in_frame(1)
  const size_t o_xyz = 0, o_uv = o_xyz + 24 * N, o_px = o_uv + 16 * N, o_a = o_px + 16 * N, o_b = o_a + 16 * N,
               o_f = o_b + 16 * N, o_obs = o_f + 24 * N, o_in = o_obs + 8 * N, total = o_in + N;
  char* d = nullptr;
  int rc = svo_ctx_staging(ctx, total, &d);
  if (rc != SVO_HIP_OK) return rc;
  if (xyz && rc == SVO_HIP_OK) rc = svo_hip_memcpy_h2d(ctx, d + o_xyz, xyz, 24 * N);
  if (uv && rc == SVO_HIP_OK) rc = svo_hip_memcpy_h2d(ctx, d + o_uv, uv, 16 * N);
  if (px && rc == SVO_HIP_OK) rc = svo_hip_memcpy_h2d(ctx, d + o_px, px, 16 * N);
  if (obs && rc == SVO_HIP_OK) rc = svo_hip_memcpy_h2d(ctx, d + o_obs, obs, 8 * N);
  if (rc != SVO_HIP_OK) return rc;
  hipLaunchKernelGGL(camera_batch_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, svo_make_cam(*cam), n,
                     xyz ? (const double*)(d + o_xyz) : nullptr, uv ? (const double*)(d + o_uv) : nullptr,
                     px ? (const double*)(d + o_px) : nullptr, obs ? (const int32_t*)(d + o_obs) : nullptr, boundary, level,
                     px_of_xyz ? (double*)(d + o_a) : nullptr, px_of_uv ? (double*)(d + o_b) : nullptr,
                     f_of_px ? (double*)(d + o_f) : nullptr, in_frame ? (uint8_t*)(d + o_in) : nullptr);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  if (xyz && px_of_xyz && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, px_of_xyz, d + o_a, 16 * N);
  if (uv && px_of_uv && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, px_of_uv, d + o_b, 16 * N);
  if (px && f_of_px && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, f_of_px, d + o_f, 24 * N);
  if (obs && in_frame && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, in_frame, d + o_in, N);
  return rc;
}

int svo_hip_update_seed_batch_dev(svo_hip_ctx* ctx, int n, const float* x, const float* tau2, float* a, float* b,
                                  float* mu, const float* z_range, float* sigma2) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n >= 0);
  if (n == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, x && tau2 && a && b && mu && z_range && sigma2);
  hipLaunchKernelGGL(update_seed_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, n, x, tau2, a, b, mu,
                     z_range, sigma2);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

int svo_hip_compute_tau_batch_dev(svo_hip_ctx* ctx, int n, const double T_ref_cur[7], const double* f,
                                  const double* z, double px_error_angle, double* tau) {
  if (!ctx || !T_ref_cur) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n >= 0);
  if (n == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, f && z && tau);
  Vec3 t = {{T_ref_cur[0], T_ref_cur[1], T_ref_cur[2]}};
  hipLaunchKernelGGL(compute_tau_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, n, t, f, z,
                     px_error_angle, tau);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

// host-side SE3 helpers (same arithmetic as the device ones; this TU is built with -ffp-contract=off)
static void h_cross(const double* a, const double* b, double* o) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static void h_rot(const double* q, const double* p, double* o) {
  double uv[3], quv[3];
  h_cross(q, p, uv);
  uv[0] = uv[0] + uv[0]; uv[1] = uv[1] + uv[1]; uv[2] = uv[2] + uv[2];
  h_cross(q, uv, quv);
  double x = (p[0] + q[3] * uv[0]) + quv[0], y = (p[1] + q[3] * uv[1]) + quv[1], z = (p[2] + q[3] * uv[2]) + quv[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void h_inv(const double* T, double* o) {
  double qi[4] = {-T[3], -T[4], -T[5], T[6]}, rt[3];
  h_rot(qi, T, rt);
  o[0] = -rt[0]; o[1] = -rt[1]; o[2] = -rt[2]; o[3] = qi[0]; o[4] = qi[1]; o[5] = qi[2]; o[6] = qi[3];
}
static void h_mul(const double* A, const double* B, double* o) {
  const double* a = A + 3; const double* b = B + 3;
  double q[4] = {a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2],
                 a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]};
  double rt[3];
  h_rot(a, B, rt);
  double t0 = A[0] + rt[0], t1 = A[1] + rt[1], t2 = A[2] + rt[2];
  o[0] = t0; o[1] = t1; o[2] = t2; o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
}

// frame-level constants of one updateSeeds / findEpipolarMatchDirect batch
static void df_make_frame(const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, const svo_hip_camera* cam,
                          const double T_ref_w[7], const double T_cur_w[7], const svo_hip_df_params* prm, DfFrame& fr) {
  memset(&fr, 0, sizeof(fr));
  fr.cam = svo_make_cam(*cam);
  double T_cur_inv[7];
  h_inv(T_cur_w, T_cur_inv);
  h_mul(T_ref_w, T_cur_inv, fr.T_ref_cur);
  h_inv(fr.T_ref_cur, fr.T_cur_ref_vis);
  h_inv(T_ref_w, fr.T_ref_inv);
  h_mul(T_cur_w, fr.T_ref_inv, fr.T_cur_ref);
  const double focal_length = fabs(cam->fx);
  fr.px_error_angle = atan(1.0 / (2.0 * focal_length)) * 2.0;       // depth_filter.cpp:245-247
  for (int l = 0; l < ref->n_levels; ++l) fr.ref_level_off[l] = ref->level_offset[l];
  for (int l = 0; l < cur->n_levels; ++l) fr.cur_level_off[l] = cur->level_offset[l];
  fr.n_pyr_levels = prm->n_pyr_levels;
  fr.align_max_iter = prm->align_max_iter;
  fr.max_epi_search_steps = prm->max_epi_search_steps;
  fr.conv_thresh = prm->seed_convergence_sigma2_thresh;
}

// per-seed records between the stages + the word-transposed warped patches: grow-only scratch owned by the context
static int df_scratch(svo_hip_ctx* ctx, int n, SeedRec** recs, uint32_t** pwb_t, int* n_pad) {
  *n_pad = (n + 63) / 64 * 64;
  const size_t rec_bytes = ((size_t)n * sizeof(SeedRec) + 255) & ~(size_t)255;
  const size_t need = rec_bytes + (size_t)25 * *n_pad * sizeof(uint32_t);
  void* ws = nullptr;
  const int rc = svo_ctx_scratch(ctx, need, &ws);
  if (rc != SVO_HIP_OK) return rc;
  *recs = (SeedRec*)ws;
  *pwb_t = reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + rec_bytes);
  return SVO_HIP_OK;
}

}  // extern "C"

// events of a device-resident seed batch (see svo_hip_seed_batch_* below): where the finalize stage leaves its counts
struct DfEvents {
  uint8_t* alive = nullptr;          // [n] in/out: 0 = erased from the list; cleared for seeds that converge / turn NaN
  int report_updated = 0;
  int* block_count = nullptr;        // [(n + 255) / 256]
  int* hist = nullptr;               // [8]
};

// One DepthFilter::updateSeeds pass over n seeds in device SoA arrays: geometry -> search -> align -> finalize on the
// context stream (shared by svo_hip_depth_filter_update_dev and the seed batches).
static int df_run_pass(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, int ref_slot, const svo_hip_pyramid* cur, int cur_slot,
                       const svo_hip_camera* cam, const double T_ref_w[7], const double T_cur_w[7], int n, const double* px,
                       const double* f, const int32_t* level, float* a, float* b, float* mu, const float* z_range, float* sigma2,
                       const svo_hip_df_params* prm, int32_t* status, double* z, double* xyz_world, int32_t* n_zmssd,
                       int32_t* n_align_iters, double* px_cur, int32_t* search_level, const DfEvents* ev) {
  if (!ctx || !ref || !cur || !cam || !T_ref_w || !T_cur_w || !prm) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, ref_slot >= 0 && ref_slot < ref->batch && cur_slot >= 0 && cur_slot < cur->batch);
  SVO_REQUIRE(ctx, ref->width == cam->width && ref->height == cam->height);
  SVO_REQUIRE(ctx, cur->width == cam->width && cur->height == cam->height);
  SVO_REQUIRE(ctx, prm->n_pyr_levels >= 1 && prm->n_pyr_levels <= ref->n_levels && prm->n_pyr_levels <= cur->n_levels);
  SVO_REQUIRE(ctx, prm->align_max_iter >= 0 && prm->max_epi_search_steps >= 0);
  SVO_REQUIRE(ctx, n >= 0);
  if (n == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, px && f && level && a && b && mu && z_range && sigma2 && status);
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  DfFrame fr;
  df_make_frame(ref, cur, cam, T_ref_w, T_cur_w, prm, fr);
  SeedRec* recs = nullptr;
  uint32_t* pwb_t = nullptr;
  int n_pad = 0;
  {
    const int rc = df_scratch(ctx, n, &recs, &pwb_t, &n_pad);
    if (rc != SVO_HIP_OK) return rc;
  }
  const uint8_t* ref_img = ref->base + (size_t)ref_slot * ref->pyr_bytes;
  const uint8_t* cur_img = cur->base + (size_t)cur_slot * cur->pyr_bytes;
  const bool prof = ctx->df_profile;                     // diagnostic: events between the stages (svo_hip_df_set_profiling)
  auto stamp = [&](int k) { if (prof) (void)hipEventRecord(ctx->df_ev[k], ctx->stream); };
  stamp(0);
  hipLaunchKernelGGL(df_geometry_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fr, n, px, f, level, mu, sigma2,
                     (const double*)nullptr, (double*)nullptr, recs, ev ? (const uint8_t*)ev->alive : (const uint8_t*)nullptr,
                     ev ? ev->hist : (int*)nullptr);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  stamp(1);
  hipLaunchKernelGGL(df_search_kernel, dim3((n + SEEDS_PER_BLOCK - 1) / SEEDS_PER_BLOCK), dim3(256), 0, ctx->stream, fr, ref_img, (size_t)0, cur_img, n, level, recs, pwb_t, n_pad);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  stamp(2);
  launch_df_align<false>(ctx, fr, cur_img, n, n_pad, pwb_t, recs);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  stamp(3);
  struct Last { const bool on; svo_hip_ctx* c; ~Last() { if (on) { (void)hipEventRecord(c->df_ev[4], c->stream); c->df_ev_recorded = true; } } } last{prof, ctx};
  if (ev) {
    hipLaunchKernelGGL(df_finalize_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fr, n, f, recs, a, b, mu, z_range,
                       sigma2, status, z, xyz_world, n_zmssd, n_align_iters, px_cur, search_level, ev->alive, ev->report_updated,
                       ev->block_count, ev->hist);
  } else {
    hipLaunchKernelGGL(df_finalize_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fr, n, f, recs, a, b, mu, z_range,
                       sigma2, status, z, xyz_world, n_zmssd, n_align_iters, px_cur, search_level);
  }
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

extern "C" {

// Diagnostic: time the four stages of every following depth-filter pass of this context with HIP events on its stream
// (geometry, search, align, finalize); svo_hip_df_get_profile waits for the last pass and returns their durations.
int svo_hip_df_set_profiling(svo_hip_ctx* ctx, int enable) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  if (enable) for (hipEvent_t& e : ctx->df_ev) if (!e) SVO_CHECK_HIP(ctx, hipEventCreate(&e));
  ctx->df_profile = enable != 0;
  ctx->df_ev_recorded = false;
  return SVO_HIP_OK;
}

int svo_hip_df_get_profile(svo_hip_ctx* ctx, double stage_us[4]) {
  if (!ctx || !stage_us) return SVO_HIP_ERR_INVALID;
  if (!ctx->df_ev_recorded) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_df_get_profile", "no profiled pass yet");
  SVO_CHECK_HIP(ctx, hipEventSynchronize(ctx->df_ev[4]));
  for (int k = 0; k < 4; ++k) {
    float ms = 0.0f;
    SVO_CHECK_HIP(ctx, hipEventElapsedTime(&ms, ctx->df_ev[k], ctx->df_ev[k + 1]));
    stage_us[k] = 1e3 * (double)ms;
  }
  return SVO_HIP_OK;
}

int svo_hip_depth_filter_update_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, int ref_slot,
                                    const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam,
                                    const double T_ref_w[7], const double T_cur_w[7], int n, const double* px,
                                    const double* f, const int32_t* level, float* a, float* b, float* mu,
                                    const float* z_range, float* sigma2, const svo_hip_df_params* prm,
                                    int32_t* status, double* z, double* xyz_world, int32_t* n_zmssd,
                                    int32_t* n_align_iters, double* px_cur, int32_t* search_level) {
  return df_run_pass(ctx, ref, ref_slot, cur, cur_slot, cam, T_ref_w, T_cur_w, n, px, f, level, a, b, mu, z_range, sigma2, prm, status, z,
                     xyz_world, n_zmssd, n_align_iters, px_cur, search_level, nullptr);
}

int svo_hip_epipolar_match_batch_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, int ref_slot,
                                     const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam,
                                     const double T_ref_w[7], const double T_cur_w[7], int n, const double* px,
                                     const double* f, const int32_t* level, const double* depth_est_min_max,
                                     const svo_hip_df_params* prm, uint8_t* ok, double* depth, double* px_cur,
                                     int32_t* search_level, double* epi_length, int32_t* n_zmssd,
                                     int32_t* n_align_iters) {
  if (!ctx || !ref || !cur || !cam || !T_ref_w || !T_cur_w || !prm) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, ref_slot >= 0 && ref_slot < ref->batch && cur_slot >= 0 && cur_slot < cur->batch);
  SVO_REQUIRE(ctx, ref->width == cam->width && ref->height == cam->height);
  SVO_REQUIRE(ctx, cur->width == cam->width && cur->height == cam->height);
  SVO_REQUIRE(ctx, prm->n_pyr_levels >= 1 && prm->n_pyr_levels <= ref->n_levels && prm->n_pyr_levels <= cur->n_levels);
  SVO_REQUIRE(ctx, prm->align_max_iter >= 0 && prm->max_epi_search_steps >= 0);
  SVO_REQUIRE(ctx, n >= 0);
  if (n == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, px && f && level && depth_est_min_max && ok);
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  DfFrame fr;
  df_make_frame(ref, cur, cam, T_ref_w, T_cur_w, prm, fr);
  SeedRec* recs = nullptr;
  uint32_t* pwb_t = nullptr;
  int n_pad = 0;
  {
    const int rc = df_scratch(ctx, n, &recs, &pwb_t, &n_pad);
    if (rc != SVO_HIP_OK) return rc;
  }
  const uint8_t* ref_img = ref->base + (size_t)ref_slot * ref->pyr_bytes;
  const uint8_t* cur_img = cur->base + (size_t)cur_slot * cur->pyr_bytes;
  hipLaunchKernelGGL(df_geometry_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fr, n, px, f, level,
                     (const float*)nullptr, (const float*)nullptr, depth_est_min_max, epi_length, recs);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(df_search_kernel, dim3((n + SEEDS_PER_BLOCK - 1) / SEEDS_PER_BLOCK), dim3(256), 0, ctx->stream, fr, ref_img, (size_t)0, cur_img, n, level, recs, pwb_t, n_pad);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  launch_df_align<false>(ctx, fr, cur_img, n, n_pad, pwb_t, recs);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(epi_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fr, n, f, recs, ok, depth, px_cur,
                     search_level, n_zmssd, n_align_iters);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

}  // extern "C"

// The middle stages of the findMatchDirect pipeline for callers that form the records themselves (svo_track.hip): scratch
// for n_cap records + word-transposed patches, then warp / align2D (/ align1D) over records [0, min(n_cap, *n_dev)).
int svo_match_scratch(svo_hip_ctx* ctx, int n_cap, svo_dev::SeedRec** recs, uint32_t** pwb_t, int* n_pad) {
  return df_scratch(ctx, n_cap, recs, pwb_t, n_pad);
}

int svo_match_stages(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam,
                     int n_cap, const int* n_dev, const int32_t* level_ref_dev, svo_dev::SeedRec* recs, uint32_t* pwb_t, int n_pad,
                     int n_pyr_levels, int align_max_iter, bool edgelets) {
  if (!ctx || !ref || !cur || !cam || !recs || !pwb_t || !level_ref_dev) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n_cap > 0 && cur_slot >= 0 && cur_slot < cur->batch);
  DfFrame fr;
  memset(&fr, 0, sizeof(fr));
  fr.cam = svo_make_cam(*cam);
  for (int l = 0; l < ref->n_levels; ++l) fr.ref_level_off[l] = ref->level_offset[l];
  for (int l = 0; l < cur->n_levels; ++l) fr.cur_level_off[l] = cur->level_offset[l];
  fr.n_pyr_levels = n_pyr_levels; fr.align_max_iter = align_max_iter; fr.keep_px_on_failure = 1;
  const uint8_t* cur_img = cur->base + (size_t)cur_slot * cur->pyr_bytes;
  hipLaunchKernelGGL(df_search_kernel, dim3((n_cap + SEEDS_PER_BLOCK - 1) / SEEDS_PER_BLOCK), dim3(256), 0, ctx->stream, fr, ref->base, ref->pyr_bytes,
                     cur_img, n_cap, level_ref_dev, recs, pwb_t, n_pad, n_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  launch_df_align<false>(ctx, fr, cur_img, n_cap, n_pad, pwb_t, recs, n_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  if (edgelets) {
    launch_df_align<true>(ctx, fr, cur_img, n_cap, n_pad, pwb_t, recs, n_dev);
    SVO_CHECK_HIP(ctx, hipGetLastError());
  }
  return SVO_HIP_OK;
}

// the same stages for n_cams cameras in one set of launches (svo_hip_tracker_group_track): see df_search_cams_kernel
int svo_match_stages_cams(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, const svo_hip_camera* cam, int n_cams,
                          int cap, const int* counters_dev, int counter_stride, const int32_t* level_ref_dev, svo_dev::SeedRec* recs,
                          uint32_t* pwb_t, int n_pad, int n_pyr_levels, int align_max_iter, bool edgelets) {
  if (!ctx || !ref || !cur || !cam || !recs || !pwb_t || !level_ref_dev || !counters_dev) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n_cams >= 1 && n_cams <= cur->batch && cap > 0 && cap % SEEDS_PER_BLOCK == 0 && cap % ALIGN_PATCHES == 0 && counter_stride >= 1);
  SVO_REQUIRE(ctx, n_pad >= n_cams * cap);
  DfFrame fr;
  memset(&fr, 0, sizeof(fr));
  fr.cam = svo_make_cam(*cam);
  for (int l = 0; l < ref->n_levels; ++l) fr.ref_level_off[l] = ref->level_offset[l];
  for (int l = 0; l < cur->n_levels; ++l) fr.cur_level_off[l] = cur->level_offset[l];
  fr.n_pyr_levels = n_pyr_levels; fr.align_max_iter = align_max_iter; fr.keep_px_on_failure = 1;
  const int n_all = n_cams * cap;
  hipLaunchKernelGGL(df_search_cams_kernel, dim3(n_all / SEEDS_PER_BLOCK), dim3(256), 0, ctx->stream, fr, ref->base, ref->pyr_bytes, cur->base,
                     cur->pyr_bytes, cap, counters_dev, counter_stride, level_ref_dev, recs, pwb_t, n_pad);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(df_align_cams_kernel<false>, dim3(n_all / ALIGN_PATCHES), dim3(ALIGN_BLOCK), 0, ctx->stream, fr, cur->base, cur->pyr_bytes, cap,
                     counters_dev, counter_stride, n_pad, pwb_t, recs);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  if (edgelets) {
    hipLaunchKernelGGL(df_align_cams_kernel<true>, dim3(n_all / ALIGN_PATCHES), dim3(ALIGN_BLOCK), 0, ctx->stream, fr, cur->base, cur->pyr_bytes, cap,
                       counters_dev, counter_stride, n_pad, pwb_t, recs);
    SVO_CHECK_HIP(ctx, hipGetLastError());
  }
  return SVO_HIP_OK;
}

// svo_hip_match_direct_batch_dev with, optionally, the pose of the current frame and the number of items left on the
// device by earlier kernels of the stream (T_cur_w_dev / n_dev non-null: the tracking chain of svo_track.hip; n is then
// the capacity the launches cover)
int svo_match_direct_internal(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, int cur_slot,
                              const svo_hip_camera* cam, int n_kf, const double* T_ref_w_dev, const double* T_cur_w,
                              const double* T_cur_w_dev, int n, const int* n_dev, const int32_t* kf_slot_dev,
                              const double* px_ref_dev, const double* f_ref_dev, const int32_t* level_ref_dev,
                              const double* pt_pos_dev, const uint8_t* edgelet_dev, const double* grad_dev, int n_pyr_levels,
                              int align_max_iter, double* px_cur_dev, uint8_t* success_dev, int32_t* search_level_dev) {
  if (!ctx || !ref || !cur || !cam || (!T_cur_w && !T_cur_w_dev)) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, cur_slot >= 0 && cur_slot < cur->batch && n_kf >= 1 && n_kf <= ref->batch);
  SVO_REQUIRE(ctx, ref->width == cam->width && ref->height == cam->height && cur->width == cam->width && cur->height == cam->height);
  SVO_REQUIRE(ctx, n_pyr_levels >= 1 && n_pyr_levels <= ref->n_levels && n_pyr_levels <= cur->n_levels && align_max_iter >= 0);
  SVO_REQUIRE(ctx, n >= 0);
  if (n == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, T_ref_w_dev && kf_slot_dev && px_ref_dev && f_ref_dev && level_ref_dev && pt_pos_dev && px_cur_dev && success_dev);
  SVO_REQUIRE(ctx, !edgelet_dev || grad_dev);
  SeedRec* recs = nullptr;
  uint32_t* pwb_t = nullptr;
  int n_pad = 0;
  {
    const int rc = df_scratch(ctx, n, &recs, &pwb_t, &n_pad);
    if (rc != SVO_HIP_OK) return rc;
  }
  MdFrame mf;
  mf.cam = svo_make_cam(*cam);
  if (T_cur_w) memcpy(mf.T_cur_w, T_cur_w, sizeof(double) * 7);
  else memset(mf.T_cur_w, 0, sizeof(double) * 7);
  mf.n_pyr_levels = n_pyr_levels;
  mf.n_kf = n_kf; mf.n_ref_levels = ref->n_levels; mf.slot_base = 0;
  DfFrame fr;
  memset(&fr, 0, sizeof(fr));
  fr.cam = mf.cam;
  for (int l = 0; l < ref->n_levels; ++l) fr.ref_level_off[l] = ref->level_offset[l];
  for (int l = 0; l < cur->n_levels; ++l) fr.cur_level_off[l] = cur->level_offset[l];
  fr.n_pyr_levels = n_pyr_levels; fr.align_max_iter = align_max_iter; fr.keep_px_on_failure = 1;
  hipLaunchKernelGGL(md_geometry_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, mf, n, T_ref_w_dev, kf_slot_dev,
                     px_ref_dev, f_ref_dev, level_ref_dev, pt_pos_dev, edgelet_dev, grad_dev, px_cur_dev, recs, T_cur_w_dev, n_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(df_search_kernel, dim3((n + SEEDS_PER_BLOCK - 1) / SEEDS_PER_BLOCK), dim3(256), 0, ctx->stream, fr, ref->base, ref->pyr_bytes,
                     cur->base + (size_t)cur_slot * cur->pyr_bytes, n, level_ref_dev, recs, pwb_t, n_pad);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  launch_df_align<false>(ctx, fr, cur->base + (size_t)cur_slot * cur->pyr_bytes, n, n_pad, pwb_t, recs);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  if (edgelet_dev) {
    launch_df_align<true>(ctx, fr, cur->base + (size_t)cur_slot * cur->pyr_bytes, n, n_pad, pwb_t, recs);
    SVO_CHECK_HIP(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(md_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, recs, px_cur_dev, success_dev,
                     search_level_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

extern "C" {

int svo_hip_match_direct_batch_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, int cur_slot,
                                   const svo_hip_camera* cam, int n_kf, const double* T_ref_w_dev,
                                   const double T_cur_w[7], int n, const int32_t* kf_slot_dev, const double* px_ref_dev,
                                   const double* f_ref_dev, const int32_t* level_ref_dev, const double* pt_pos_dev,
                                   const uint8_t* edgelet_dev, const double* grad_dev, int n_pyr_levels,
                                   int align_max_iter, double* px_cur_dev, uint8_t* success_dev,
                                   int32_t* search_level_dev) {
  if (!T_cur_w) return SVO_HIP_ERR_INVALID;
  return svo_match_direct_internal(ctx, ref, cur, cur_slot, cam, n_kf, T_ref_w_dev, T_cur_w, nullptr, n, nullptr, kf_slot_dev, px_ref_dev,
                                   f_ref_dev, level_ref_dev, pt_pos_dev, edgelet_dev, grad_dev, n_pyr_levels, align_max_iter, px_cur_dev,
                                   success_dev, search_level_dev);
}

// host-buffer convenience form: copies the SoA arrays in, runs, copies the results out, synchronises
int svo_hip_depth_filter_update(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, int ref_slot, const svo_hip_pyramid* cur,
                                int cur_slot, const svo_hip_camera* cam, const double T_ref_w[7], const double T_cur_w[7],
                                int n, const double* px, const double* f, const int32_t* level, float* a, float* b,
                                float* mu, const float* z_range, float* sigma2, const svo_hip_df_params* prm,
                                int32_t* status, double* z, double* xyz_world, int32_t* n_zmssd, int32_t* n_align_iters,
                                double* px_cur, int32_t* search_level) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n >= 0);
  if (n == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, px && f && level && a && b && mu && z_range && sigma2 && status);
  const size_t N = (size_t)n;
  // One staging layout on both sides, the pageable arguments gathered in page-locked memory: one transfer each way.
  // inputs  : px(16) f(24) level(4) z_range(4) | in/out: a b mu sigma2 (4x4) | outputs: z(8) xyz(24) px_cur(16) status nz na sl (4x4)
  const size_t o_px = 0, o_f = o_px + 16 * N, o_z = o_f + 24 * N, o_xyz = o_z + 8 * N, o_pc = o_xyz + 24 * N,
               o_lvl = o_pc + 16 * N, o_zr = o_lvl + 4 * N, o_a = o_zr + 4 * N, o_b = o_a + 4 * N, o_mu = o_b + 4 * N,
               o_s2 = o_mu + 4 * N, o_st = o_s2 + 4 * N, o_nz = o_st + 4 * N, o_na = o_nz + 4 * N, o_sl = o_na + 4 * N,
               total = o_sl + 4 * N;
  char* blk = nullptr;
  char* hs = nullptr;
  int rc = svo_ctx_staging(ctx, total, &blk);
  if (rc != SVO_HIP_OK) return rc;
  rc = svo_ctx_host_staging(ctx, total, &hs);
  if (rc != SVO_HIP_OK) return rc;
  uint8_t* d = (uint8_t*)blk;
  memcpy(hs + o_px, px, 16 * N); memcpy(hs + o_f, f, 24 * N); memcpy(hs + o_lvl, level, 4 * N);
  memcpy(hs + o_zr, z_range, 4 * N); memcpy(hs + o_a, a, 4 * N); memcpy(hs + o_b, b, 4 * N);
  memcpy(hs + o_mu, mu, 4 * N); memcpy(hs + o_s2, sigma2, 4 * N);
  // two uploads (the output block in the middle does not cross the link): [px f] and [level .. sigma2]
  hipError_t e = hipMemcpyAsync(d + o_px, hs + o_px, 40 * N, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d + o_lvl, hs + o_lvl, o_st - o_lvl, hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_depth_filter_update", hipGetErrorString(e));
  rc = svo_hip_depth_filter_update_dev(ctx, ref, ref_slot, cur, cur_slot, cam, T_ref_w, T_cur_w, n, (double*)(d + o_px),
                                       (double*)(d + o_f), (int32_t*)(d + o_lvl), (float*)(d + o_a), (float*)(d + o_b),
                                       (float*)(d + o_mu), (float*)(d + o_zr), (float*)(d + o_s2), prm,
                                       (int32_t*)(d + o_st), (double*)(d + o_z), (double*)(d + o_xyz),
                                       (int32_t*)(d + o_nz), (int32_t*)(d + o_na), (double*)(d + o_pc), (int32_t*)(d + o_sl));
  if (rc != SVO_HIP_OK) return rc;
  e = hipMemcpyAsync(hs + o_z, d + o_z, o_lvl - o_z, hipMemcpyDeviceToHost, ctx->stream);                 // z xyz px_cur
  if (e == hipSuccess) e = hipMemcpyAsync(hs + o_a, d + o_a, total - o_a, hipMemcpyDeviceToHost, ctx->stream);   // a .. search level
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_depth_filter_update", hipGetErrorString(e));
  memcpy(a, hs + o_a, 4 * N); memcpy(b, hs + o_b, 4 * N); memcpy(mu, hs + o_mu, 4 * N); memcpy(sigma2, hs + o_s2, 4 * N);
  memcpy(status, hs + o_st, 4 * N);
  if (z) memcpy(z, hs + o_z, 8 * N);
  if (xyz_world) memcpy(xyz_world, hs + o_xyz, 24 * N);
  if (px_cur) memcpy(px_cur, hs + o_pc, 16 * N);
  if (n_zmssd) memcpy(n_zmssd, hs + o_nz, 4 * N);
  if (n_align_iters) memcpy(n_align_iters, hs + o_na, 4 * N);
  if (search_level) memcpy(search_level, hs + o_sl, 4 * N);
  return SVO_HIP_OK;
}

int svo_hip_seed_compact_converged_dev(svo_hip_ctx* ctx, int n, long long id_offset, const int32_t* status_dev,
                                       const float* mu_dev, const float* sigma2_dev, const double* xyz_world_dev,
                                       double* records_dev, int32_t* count_dev) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n >= 0 && count_dev);
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) {
    SVO_CHECK_HIP(ctx, hipMemsetAsync(count_dev, 0, sizeof(int32_t), ctx->stream));
    return SVO_HIP_OK;
  }
  SVO_REQUIRE(ctx, status_dev && mu_dev && sigma2_dev && xyz_world_dev && records_dev);
  const int n_blocks = (n + 255) / 256;
  void* ws = nullptr;
  const int rc = svo_ctx_scratch(ctx, sizeof(int) * (size_t)n_blocks + 64, &ws);
  if (rc != SVO_HIP_OK) return rc;
  int* block_count = reinterpret_cast<int*>(ws);
  hipLaunchKernelGGL(conv_count_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream, n, status_dev, block_count);
  hipLaunchKernelGGL(conv_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, n_blocks, block_count, reinterpret_cast<int*>(count_dev));
  hipLaunchKernelGGL(conv_scatter_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream, n, id_offset, status_dev, mu_dev, sigma2_dev,
                     xyz_world_dev, block_count, records_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

// The exchange step of the seed-sharded depth filter (BASELINE config C4): this rank's converged seeds are
// packed on the device in seed order, clamped to `cap` records, and all-gathered together with the per-rank counts:
// records_all[world][cap][6] f64, counts_all[world] i32 (a count above cap = that rank had more: raise cap).
// Two collectives of fixed size, no host round trip in between.
int svo_hip_seed_gather_converged_dev(svo_hip_ctx* ctx, svo_hip_comm* comm, int n, long long id_offset, const int32_t* status_dev,
                                      const float* mu_dev, const float* sigma2_dev, const double* xyz_world_dev, int cap,
                                      double* records_all_dev, int32_t* counts_all_dev) {
  if (!ctx || !comm) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n >= 0 && cap >= 1 && records_all_dev && counts_all_dev);
  int rank = 0, world = 1;
  svo_hip_comm_info(comm, &rank, &world, nullptr);
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  // scratch: [block counts of the compaction][n records][count]
  const int n_blocks = (n + 255) / 256;
  const size_t o_rec = ((sizeof(int) * (size_t)n_blocks + 64) + 255) & ~(size_t)255;
  const size_t o_cnt = o_rec + (size_t)(n > 0 ? n : 1) * 6 * sizeof(double);
  void* ws = nullptr;
  int rc = svo_ctx_scratch(ctx, o_cnt + 64, &ws);
  if (rc != SVO_HIP_OK) return rc;
  // (svo_hip_seed_compact_converged_dev takes its block counts from the start of the same scratch area)
  double* rec = reinterpret_cast<double*>(static_cast<char*>(ws) + o_rec);
  int32_t* cnt = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + o_cnt);
  rc = svo_hip_seed_compact_converged_dev(ctx, n, id_offset, status_dev, mu_dev, sigma2_dev, xyz_world_dev, rec, cnt);
  if (rc != SVO_HIP_OK) return rc;
  hipLaunchKernelGGL(records_clamp_copy_kernel, dim3(64), dim3(256), 0, ctx->stream, rec, reinterpret_cast<const int*>(cnt), cap,
                     records_all_dev + (size_t)rank * cap * 6, counts_all_dev + rank);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  rc = svo_comm_all_gather(comm, records_all_dev, (size_t)cap * 6 * sizeof(double));
  if (rc != SVO_HIP_OK) return rc;
  return svo_comm_all_gather(comm, counts_all_dev, sizeof(int32_t));
}

// The cell loop of Reprojector::reprojectMap (reprojector.cpp:149-166, 180-241) over candidates bucketed per cell in
// trial order: every live candidate is matched in ONE batch on the device, then the serial policy (first success per
// cell wins, later candidates untouched, stop once n_matches exceeds max_fts) is replayed on the host over the
// results.  findMatchDirect is a pure function of its candidate, so this equals the sequential evaluation.
int svo_hip_reproject_cells(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, int cur_slot,
                            const svo_hip_camera* cam, int n_kf, const double* T_kf_w, const double T_cur_w[7], int n_cells,
                            const int32_t* cell_offset, const int32_t* kf_slot, const double* px_ref, const double* f_ref,
                            const int32_t* level_ref, const double* pt_pos, const uint8_t* edgelet, const double* grad,
                            const uint8_t* deleted, double* px_cur, int max_fts, int n_pyr_levels, int align_max_iter,
                            uint8_t* tried, uint8_t* matched, int32_t* search_level, int32_t* cell_winner,
                            uint64_t* n_matches_out, uint64_t* n_trials_out) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n_cells >= 0 && cell_offset && tried && matched && cell_winner && n_matches_out && n_trials_out);
  const int n = cell_offset[n_cells];
  SVO_REQUIRE(ctx, n >= 0 && (n == 0 || (kf_slot && px_ref && f_ref && level_ref && pt_pos && deleted && px_cur && T_kf_w)));
  SVO_REQUIRE(ctx, !edgelet || grad);
  for (int c = 0; c < n_cells; ++c) cell_winner[c] = -1;
  *n_matches_out = 0; *n_trials_out = 0;
  if (n == 0) return SVO_HIP_OK;
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = (size_t)n;
  // one device block: px_ref(16) f(24) pos(24) px_cur(16) grad(16) T_kf(56*n_kf) kf_slot(4) level(4) sl(4) edgelet(1) ok(1)
  // one staging layout on both sides: [inputs ...][px_cur (in/out)][search level][matched]; the pageable arguments are
  // gathered in page-locked memory and cross the link in one transfer each way (nine + three transfers of ~10 us before)
  const size_t o_pr = 0, o_f = o_pr + 16 * N, o_pos = o_f + 24 * N, o_g = o_pos + 24 * N, o_T = o_g + 16 * N,
               o_k = o_T + 56 * (size_t)n_kf, o_l = o_k + 4 * N, o_e = o_l + 4 * N, o_pc = (o_e + N + 15) & ~(size_t)15,
               o_sl = o_pc + 16 * N, o_ok = o_sl + 4 * N, total = o_ok + N + 64;
  char* d = nullptr;
  char* hs = nullptr;
  {
    const int rc_st = svo_ctx_staging(ctx, total, &d);
    if (rc_st != SVO_HIP_OK) return rc_st;
    const int rc_hs = svo_ctx_host_staging(ctx, total, &hs);
    if (rc_hs != SVO_HIP_OK) return rc_hs;
  }
  auto put = [&](size_t off, const void* src, size_t bytes) { if (src) memcpy(hs + off, src, bytes); };
  put(o_pr, px_ref, 16 * N); put(o_f, f_ref, 24 * N); put(o_pos, pt_pos, 24 * N); put(o_g, grad, 16 * N);
  put(o_T, T_kf_w, 56 * (size_t)n_kf); put(o_k, kf_slot, 4 * N); put(o_l, level_ref, 4 * N); put(o_e, edgelet, N);
  put(o_pc, px_cur, 16 * N);
  hipError_t e = hipMemcpyAsync(d, hs, o_pc + 16 * N, hipMemcpyHostToDevice, ctx->stream);
  int rc = SVO_HIP_OK;
  const uint8_t* ok = reinterpret_cast<const uint8_t*>(hs + o_ok);
  const int32_t* sl = reinterpret_cast<const int32_t*>(hs + o_sl);
  if (e == hipSuccess) {
    rc = svo_hip_match_direct_batch_dev(ctx, ref, cur, cur_slot, cam, n_kf, reinterpret_cast<const double*>(d + o_T), T_cur_w, n,
                                        reinterpret_cast<const int32_t*>(d + o_k), reinterpret_cast<const double*>(d + o_pr),
                                        reinterpret_cast<const double*>(d + o_f), reinterpret_cast<const int32_t*>(d + o_l),
                                        reinterpret_cast<const double*>(d + o_pos),
                                        edgelet ? reinterpret_cast<const uint8_t*>(d + o_e) : nullptr,
                                        edgelet ? reinterpret_cast<const double*>(d + o_g) : nullptr, n_pyr_levels, align_max_iter,
                                        reinterpret_cast<double*>(d + o_pc), reinterpret_cast<uint8_t*>(d + o_ok),
                                        reinterpret_cast<int32_t*>(d + o_sl));
    if (rc == SVO_HIP_OK) {
      e = hipMemcpyAsync(hs + o_pc, d + o_pc, o_ok + N - o_pc, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
  }
  if (e != hipSuccess) return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_reproject_cells", hipGetErrorString(e));
  if (rc != SVO_HIP_OK) return rc;
  // ---- the serial policy (reprojector.cpp:149-166 with reprojectCell :180-241)
  memset(tried, 0, N);
  memset(matched, 0, N);
  uint64_t n_matches = 0, n_trials = 0;
  for (int c = 0; c < n_cells; ++c) {
    for (int i = cell_offset[c]; i < cell_offset[c + 1]; ++i) {
      ++n_trials;
      tried[i] = 1;
      if (deleted[i]) continue;                              // TYPE_DELETED: erased from the cell (:190-194)
      // findMatchDirect ran for this candidate: its px_cur is rewritten (matcher.cpp:200); candidates the serial
      // loop never reaches keep the caller's values
      memcpy(px_cur + 2 * (size_t)i, hs + o_pc + 16 * (size_t)i, 16);
      if (search_level) search_level[i] = sl[i];
      if (!ok[i]) continue;                                  // the caller counts the failure on the point (:202-209)
      matched[i] = 1;
      cell_winner[c] = i;
      ++n_matches;
      break;                                                 // maximum one point per cell (:238-239)
    }
    if (n_matches > (uint64_t)max_fts) break;                // :164-165
  }
  *n_matches_out = n_matches;
  *n_trials_out = n_trials;
  return SVO_HIP_OK;
}

}  // extern "C"

// ---- device-resident seed batches (include/svo_hip.h: svo_hip_seed_batch_*) -----------------------------------------
namespace {

struct EvHeader { int32_t n_events; int32_t counts[7]; };

// exclusive scan of the per-block event counts (one workgroup of 1024) + the header of the page-locked event block
SVO_DEV void ev_scan_block(int n_blocks, int* __restrict__ block_count, const int* __restrict__ hist,
                           EvHeader* __restrict__ header /* host memory */, int* __restrict__ n_events_dev) {
  __shared__ int s_part[1024];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n_blocks; base += 1024) {
    const int k = base + threadIdx.x;
    const int v = k < n_blocks ? block_count[k] : 0;
    int incl = v;                                              // wave-level inclusive scan, then the 16 wave totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += t; }
    if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += s_part[w];
    const int carry = s_carry;
    if (k < n_blocks) block_count[k] = carry + wave_off + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) { header->n_events = s_carry; *n_events_dev = s_carry; }
  if (threadIdx.x < 7) header->counts[threadIdx.x] = hist[threadIdx.x];
}

__global__ __launch_bounds__(1024) void ev_scan_kernel(int n_blocks, int* __restrict__ block_count, const int* __restrict__ hist,
                                                       EvHeader* __restrict__ header, int* __restrict__ n_events_dev) {
  ev_scan_block(n_blocks, block_count, hist, header, n_events_dev);
}

// The events, ascending seed index (56 B records).  A handful of them -- every frame but a keyframe's -- goes straight into
// the page-locked host block (no copy, no second wait); more than EV_DIRECT_MAX are packed in device memory and fetched
// with one transfer after the wait (scattered 56-byte stores over the link run at a few GB/s: 100 k events took 1.4 ms).
constexpr int EV_DIRECT_MAX = 512;
SVO_DEV void ev_scatter_block(int block, int n, int report_updated, const int32_t* __restrict__ status,
                              const float* __restrict__ mu, const float* __restrict__ sigma2, const double* __restrict__ xyz,
                              const double* __restrict__ px_cur, const int* __restrict__ block_offset,
                              const int* __restrict__ n_events_dev, svo_hip_seed_event* __restrict__ events_host,
                              svo_hip_seed_event* __restrict__ events_dev) {
  svo_hip_seed_event* events = *n_events_dev <= EV_DIRECT_MAX ? events_host : events_dev;
  const int i = block * 256 + threadIdx.x;
  const int st = i < n ? status[i] : SVO_HIP_SEED_ERASED;
  const bool ev = st == SVO_HIP_SEED_CONVERGED || st == SVO_HIP_SEED_NAN || (report_updated && st >= SVO_HIP_SEED_UPDATED);
  const unsigned long long m = __ballot(ev);
  __shared__ int s_w[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_w[wave] = __popcll(m);
  __syncthreads();
  if (!ev) return;
  int off = block_offset[block];
  for (int w = 0; w < wave; ++w) off += s_w[w];
  off += __popcll(m & ((1ull << lane) - 1ull));
  svo_hip_seed_event e;
  e.index = i; e.status = st; e.mu = mu[i]; e.sigma2 = sigma2[i];
  const bool conv = st == SVO_HIP_SEED_CONVERGED;
  e.xyz_world[0] = conv ? xyz[3 * (size_t)i] : 0.0; e.xyz_world[1] = conv ? xyz[3 * (size_t)i + 1] : 0.0; e.xyz_world[2] = conv ? xyz[3 * (size_t)i + 2] : 0.0;
  e.px_cur[0] = px_cur[2 * (size_t)i]; e.px_cur[1] = px_cur[2 * (size_t)i + 1];
  events[off] = e;
}

__global__ __launch_bounds__(256) void ev_scatter_kernel(int n, int report_updated, const int32_t* __restrict__ status,
                                                         const float* __restrict__ mu, const float* __restrict__ sigma2,
                                                         const double* __restrict__ xyz, const double* __restrict__ px_cur,
                                                         const int* __restrict__ block_offset, const int* __restrict__ n_events_dev,
                                                         svo_hip_seed_event* __restrict__ events_host,
                                                         svo_hip_seed_event* __restrict__ events_dev) {
  ev_scatter_block(blockIdx.x, n, report_updated, status, mu, sigma2, xyz, px_cur, block_offset, n_events_dev, events_host, events_dev);
}

// ---- one launch set for SEVERAL batches (svo_hip_seed_batch_update_group_async) ------------------------------------------
// A frame of the reference's depth filter updates the seeds of a few keyframes, a few hundred each: per batch that is six
// launches of a handful of workgroups -- launch-bound (4 x 500 seeds: 105 us, profiles/r04_df_realistic_sizes.txt).  Here the
// batches of a frame go through the stages together: batch j owns the blocks [first_block, first_block + n_blocks) of the
// thread-per-seed stages (a block never straddles two batches, so the batch -- its arrays, its keyframe's transforms -- is
// block-uniform and lives in scalar registers), its records sit at first_block * 256 of the pass's scratch, the tail of its
// last block is filled with inert records, and the pixel stages (search, align) run over the concatenation unchanged: a
// record carries its keyframe's pyramid slot (SeedRec::pad).
constexpr int DF_GROUP_MAX = 8;
struct DfJob {
  double T_ref_cur[7], T_cur_ref_vis[7], T_cur_ref[7], T_ref_inv[7];     // as DfFrame's, of this batch's keyframe
  const double *px, *f;
  const int32_t* level;
  float *a, *b, *mu;
  const float* z_range;
  float* sigma2;
  int32_t* status;
  double *xyz, *px_cur;
  uint8_t* alive;
  int *block_count, *hist;
  EvHeader* header;
  svo_hip_seed_event *events_host, *events_dev;
  int n, n_blocks, first_block, ref_slot;
};
struct DfGroup {
  DfJob job[DF_GROUP_MAX];
  int n_jobs, report_updated;
};
static_assert(sizeof(DfFrame) + sizeof(DfGroup) + 64 <= 4096, "kernel arguments of the grouped stages");

SVO_DEV int df_job_of_block(const DfGroup& g, int block) {
  int j = 0;
#pragma unroll
  for (int k = 1; k < DF_GROUP_MAX; ++k) if (k < g.n_jobs && block >= g.job[k].first_block) j = k;
  return j;
}

__global__ __launch_bounds__(256) void df_geometry_group_kernel(DfFrame fr, DfGroup g, SeedRec* __restrict__ recs,
                                                                int32_t* __restrict__ level_cat) {
  const DfJob& J = g.job[df_job_of_block(g, blockIdx.x)];
  const int i = (blockIdx.x - J.first_block) * 256 + threadIdx.x;       // seed of its batch
  const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;              // its record in the pass
  if (i < 8) J.hist[i] = 0;
  if (i < J.n) {
    level_cat[r] = J.level[i];
    df_geometry_seed<false>(fr.cam, J.T_cur_ref_vis, J.T_cur_ref, fr.n_pyr_levels, fr.max_epi_search_steps, J.ref_slot, i, J.n, J.px, J.f,
                            J.level, J.mu, J.sigma2, nullptr, nullptr, recs + r, J.alive);
  } else {                                                 // the tail of the batch's last block: records no stage touches
    SeedRec rc = md_dead_record();
    rc.status = SVO_HIP_SEED_ERASED;
    recs[r] = rc;
    level_cat[r] = 0;
  }
}

__global__ __launch_bounds__(256) void df_finalize_group_kernel(DfFrame fr, DfGroup g, const SeedRec* __restrict__ recs) {
  const DfJob& J = g.job[df_job_of_block(g, blockIdx.x)];
  const int block = blockIdx.x - J.first_block;
  const int i = block * 256 + threadIdx.x;
  int st = SVO_HIP_SEED_ERASED;
  if (i < J.n) {
    const SeedRec rc = recs[(size_t)blockIdx.x * 256 + threadIdx.x];
    st = df_finalize_seed(fr.cam, J.T_cur_ref, J.T_ref_cur, J.T_ref_inv, fr.px_error_angle, fr.conv_thresh, i, rc, J.f, J.a, J.b, J.mu,
                          J.z_range, J.sigma2, J.status, nullptr, J.xyz, nullptr, nullptr, J.px_cur, nullptr);
  }
  df_finalize_events(st, i, J.n, block, J.alive, g.report_updated, J.block_count, J.hist);
}

__global__ __launch_bounds__(1024) void ev_scan_group_kernel(DfGroup g) {
  const DfJob& J = g.job[blockIdx.x];
  ev_scan_block(J.n_blocks, J.block_count, J.hist, J.header, J.hist + 8);
}

__global__ __launch_bounds__(256) void ev_scatter_group_kernel(DfGroup g) {
  const DfJob& J = g.job[df_job_of_block(g, blockIdx.x)];
  ev_scatter_block(blockIdx.x - J.first_block, J.n, g.report_updated, J.status, J.mu, J.sigma2, J.xyz, J.px_cur, J.block_count, J.hist + 8,
                   J.events_host, J.events_dev);
}

// ---- small passes: two launches ------------------------------------------------------------------------------------------
// Below a few thousand seeds every stage is a handful of workgroups whose time is its launch and a chain of dependent memory
// trips (4 keyframes x 500 seeds: geometry 11, search 10, align 18, finalize 12 us with events between them), six launches a
// frame.  Here ONE wave takes its four seeds through all four stages -- geometry on lanes 0-3, the search stage as it is (it
// never had a block-level barrier: a wave works on its own four seeds), alignment on lanes 0-15 (a quad per seed), the update
// on lanes 0-3 -- handing the records and warped patches from stage to stage through the same scratch as the large pass
// (same wave, same CU: a workgroup-scope fence orders them), and a second launch of one workgroup per batch does what the
// finalize stage's bookkeeping, ev_scan and ev_scatter do: outcome counts, `alive`, the scan and the ordered events.
// The lanes that idle through the f64 stages cost nothing here: the chip is empty.  Same device functions as the large pass,
// so every per-seed result is bit-identical (tests/test_gpu_seed_batch.py).
constexpr int DF_SMALL_MAX = 16384;                       // upper limit of svo_hip_df_set_small_pass_limit
__global__ __launch_bounds__(256) void df_small_pass_kernel(DfFrame fr, DfGroup g, const uint8_t* __restrict__ ref_base,
                                                            size_t ref_pyr_bytes, const uint8_t* __restrict__ cur_pyr, int n_cat,
                                                            SeedRec* __restrict__ recs, uint32_t* __restrict__ pwb_t,
                                                            int32_t* __restrict__ level_cat, int n_pad) {
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + wib) * SEEDS_PER_WAVE;   // the wave's first record
  if (r0 >= n_cat) return;                                 // wave-uniform; no block-level barrier below
  const DfJob& J = g.job[df_job_of_block(g, blockIdx.x >> 4)];          // 16 of these workgroups per 256-seed block
  const int i0 = r0 - J.first_block * 256;                 // ... and its first seed within its batch
  // ---- geometry
  if (lane < SEEDS_PER_WAVE) {
    const int i = i0 + lane;
    const size_t r = (size_t)r0 + lane;
    if (i < J.n) {
      level_cat[r] = J.level[i];
      df_geometry_seed<false>(fr.cam, J.T_cur_ref_vis, J.T_cur_ref, fr.n_pyr_levels, fr.max_epi_search_steps, J.ref_slot, i, J.n, J.px, J.f,
                              J.level, J.mu, J.sigma2, nullptr, nullptr, recs + r, J.alive);
    } else {                                               // the tail of a batch's last block
      SeedRec rc = md_dead_record();
      rc.status = SVO_HIP_SEED_ERASED;
      recs[r] = rc;
      level_cat[r] = 0;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // ---- warp + epipolar search
  df_search_block(fr, ref_base, ref_pyr_bytes, cur_pyr, n_cat, level_cat, recs, pwb_t, n_pad, blockIdx.x);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // ---- align2D
  {
    const int r = r0 + (lane >> 2);
    df_align_quad<false>(fr, cur_pyr, n_pad, pwb_t, recs, r, lane < 4 * SEEDS_PER_WAVE && r < n_cat);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // ---- triangulation, computeTau, updateSeed
  if (lane < SEEDS_PER_WAVE && i0 + lane < J.n) {
    const SeedRec rc = recs[(size_t)r0 + lane];
    (void)df_finalize_seed(fr.cam, J.T_cur_ref, J.T_ref_cur, J.T_ref_inv, fr.px_error_angle, fr.conv_thresh, i0 + lane, rc, J.f, J.a, J.b,
                           J.mu, J.z_range, J.sigma2, J.status, nullptr, J.xyz, nullptr, nullptr, J.px_cur, nullptr);
  }
}

// one workgroup of 1024 per batch: what df_finalize_events + ev_scan_block + ev_scatter_block do, for a batch of at most
// DF_SMALL_MAX seeds
__global__ __launch_bounds__(1024) void ev_small_kernel(DfGroup g) {
  const DfJob& J = g.job[blockIdx.x];
  const int n = J.n, report_updated = g.report_updated;
  __shared__ int s_cnt[DF_SMALL_MAX / 256], s_off[DF_SMALL_MAX / 256], s_hist[8], s_w[16], s_total;
  if (threadIdx.x < DF_SMALL_MAX / 256) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x < 8) s_hist[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 1024) {
    const int st = J.status[i];
    const bool gone = st == SVO_HIP_SEED_CONVERGED || st == SVO_HIP_SEED_NAN;
    if (gone) J.alive[i] = 0;
    atomicAdd(&s_hist[st + 1], 1);
    if (gone || (report_updated && st >= SVO_HIP_SEED_UPDATED)) atomicAdd(&s_cnt[i >> 8], 1);
  }
  __syncthreads();
  if (threadIdx.x < 64) {                                  // exclusive scan of the (at most 64) block counts
    const int v = threadIdx.x < J.n_blocks ? s_cnt[threadIdx.x] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((int)threadIdx.x >= o) incl += t; }
    s_off[threadIdx.x] = incl - v;
    if (threadIdx.x == 63) s_total = incl;
  }
  __syncthreads();
  const int total = s_total;
  if (threadIdx.x == 0) { J.header->n_events = total; J.hist[8] = total; }
  if (threadIdx.x < 7) J.header->counts[threadIdx.x] = s_hist[threadIdx.x];
  svo_hip_seed_event* events = total <= EV_DIRECT_MAX ? J.events_host : J.events_dev;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int st = i < n ? J.status[i] : SVO_HIP_SEED_ERASED;
    const bool ev = st == SVO_HIP_SEED_CONVERGED || st == SVO_HIP_SEED_NAN || (report_updated && st >= SVO_HIP_SEED_UPDATED);
    const unsigned long long m = __ballot(ev);
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    if (ev) {
      int off = s_off[i >> 8];
      for (int w = wave & ~3; w < wave; ++w) off += s_w[w];          // the waves of this 256-seed block before mine
      off += __popcll(m & ((1ull << lane) - 1ull));
      svo_hip_seed_event e;
      e.index = i; e.status = st; e.mu = J.mu[i]; e.sigma2 = J.sigma2[i];
      const bool conv = st == SVO_HIP_SEED_CONVERGED;
      e.xyz_world[0] = conv ? J.xyz[3 * (size_t)i] : 0.0; e.xyz_world[1] = conv ? J.xyz[3 * (size_t)i + 1] : 0.0; e.xyz_world[2] = conv ? J.xyz[3 * (size_t)i + 2] : 0.0;
      e.px_cur[0] = J.px_cur[2 * (size_t)i]; e.px_cur[1] = J.px_cur[2 * (size_t)i + 1];
      events[off] = e;
    }
    __syncthreads();
  }
}

}  // namespace

struct svo_hip_seed_batch {
  svo_hip_ctx* ctx = nullptr;
  int n = 0, n_alive = 0, n_blocks = 0;
  char* dev = nullptr;                  // one allocation: the arrays below
  double *px = nullptr, *f = nullptr, *xyz = nullptr, *px_cur = nullptr;
  int32_t *level = nullptr, *status = nullptr;
  float *a = nullptr, *b = nullptr, *mu = nullptr, *z_range = nullptr, *sigma2 = nullptr;
  uint8_t* alive = nullptr;
  int *block_count = nullptr, *hist = nullptr;
  svo_hip_seed_event* events_dev = nullptr;    // where the pass packs its events when there are many
  char* host = nullptr;                 // page-locked, mapped: EvHeader + n events, written by the kernels
  char* host_dev = nullptr;             // its device address
  bool pending = false;                 // a pass is enqueued and not collected yet
  int report_updated = 0;
  svo_seed_block blk;                   // the pool block behind dev / host (returned to the context on destroy)
};

static size_t sb_align(size_t v) { return (v + 255) & ~(size_t)255; }
static constexpr size_t kEvHeaderBytes = 64;
static_assert(sizeof(svo_hip_seed_event) == 56, "event record layout (include/svo_hip.h)");

// Capacity classes of the context's seed-batch pool: powers of two from 256 seeds (the drop-in's batches are <= 4096 seeds).
static int sb_capacity_class(int n) {
  int c = 256;
  while (c < n && c < (1 << 30)) c <<= 1;
  return c;
}

// A block for a batch of n seeds: a free block of n's capacity class, else a new one sized for the class (so that the next
// keyframe of about this size reuses it).  dev_need / host_need are the bytes a batch of n seeds lays out; both grow with n.
static int sb_take_block(svo_hip_ctx* ctx, int n, size_t dev_need, size_t host_need, svo_seed_block* out) {
  const int cap = sb_capacity_class(n);
  for (size_t i = 0; i < ctx->seed_pool.size(); ++i) {
    const svo_seed_block& b = ctx->seed_pool[i];
    if (b.cap == cap && b.dev_bytes >= dev_need && b.host_bytes >= host_need) {
      *out = b;
      ctx->seed_pool[i] = ctx->seed_pool.back();
      ctx->seed_pool.pop_back();
      ++ctx->seed_blocks_in_use;
      return SVO_HIP_OK;
    }
  }
  svo_seed_block b;
  b.cap = cap;
  // the layout of `cap` seeds bounds that of any n <= cap: per seed 16+24+4*6 uploaded, 24+16+4+1 outputs, one event record;
  // 15 sections rounded up to 256 bytes each, the block counts and the histogram
  const size_t C = (size_t)cap;
  b.dev_bytes = C * (16 + 24 + 4 * 6 + 24 + 16 + 4 + 1 + sizeof(svo_hip_seed_event)) + 4 * ((C + 255) / 256) + 64 + 16 * 256;
  if (b.dev_bytes < dev_need) b.dev_bytes = dev_need;
  b.host_bytes = kEvHeaderBytes + C * sizeof(svo_hip_seed_event);
  if (b.host_bytes < host_need) b.host_bytes = host_need;
  if (hipMalloc((void**)&b.dev, b.dev_bytes) != hipSuccess) return svo_fail(ctx, SVO_HIP_ERR_NOMEM, "svo_hip_seed_batch_create", "hipMalloc failed");
  ++ctx->n_allocs;
  if (hipHostMalloc((void**)&b.host, b.host_bytes, hipHostMallocMapped) != hipSuccess) {
    (void)hipFree(b.dev);
    return svo_fail(ctx, SVO_HIP_ERR_NOMEM, "svo_hip_seed_batch_create", "hipHostMalloc failed");
  }
  ++ctx->n_allocs;
  if (hipHostGetDevicePointer((void**)&b.host_dev, b.host, 0) != hipSuccess) {
    (void)hipFree(b.dev); (void)hipHostFree(b.host);
    return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_seed_batch_create", "hipHostGetDevicePointer failed");
  }
  *out = b;
  ++ctx->seed_blocks_in_use;
  return SVO_HIP_OK;
}


extern "C" {

int svo_hip_seed_batch_create(svo_hip_ctx* ctx, int n, const double* px, const double* f, const int32_t* level, const float* a,
                              const float* b, const float* mu, const float* z_range, const float* sigma2, svo_hip_seed_batch** out) {
  if (!ctx || !out) return SVO_HIP_ERR_INVALID;
  *out = nullptr;
  SVO_REQUIRE(ctx, n > 0 && px && f && level && a && b && mu && z_range && sigma2);
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  svo_hip_seed_batch* sb = new (std::nothrow) svo_hip_seed_batch;
  if (!sb) return svo_fail(ctx, SVO_HIP_ERR_NOMEM, "svo_hip_seed_batch_create", "out of host memory");
  sb->ctx = ctx; sb->n = n; sb->n_alive = n; sb->n_blocks = (n + 255) / 256;
  const size_t N = (size_t)n;
  // layout: uploaded part first (px f level a b mu z_range sigma2), then outputs and bookkeeping
  size_t o = 0;
  const size_t o_px = o; o = sb_align(o + 16 * N);
  const size_t o_f = o; o = sb_align(o + 24 * N);
  const size_t o_lvl = o; o = sb_align(o + 4 * N);
  const size_t o_a = o; o = sb_align(o + 4 * N);
  const size_t o_b = o; o = sb_align(o + 4 * N);
  const size_t o_mu = o; o = sb_align(o + 4 * N);
  const size_t o_zr = o; o = sb_align(o + 4 * N);
  const size_t o_s2 = o; o = sb_align(o + 4 * N);
  const size_t upload_bytes = o;
  const size_t o_xyz = o; o = sb_align(o + 24 * N);
  const size_t o_pc = o; o = sb_align(o + 16 * N);
  const size_t o_st = o; o = sb_align(o + 4 * N);
  const size_t o_al = o; o = sb_align(o + N);
  const size_t o_bc = o; o = sb_align(o + 4 * (size_t)sb->n_blocks);
  const size_t o_h = o; o = sb_align(o + 64);
  const size_t o_ev = o; o = sb_align(o + N * sizeof(svo_hip_seed_event));
  int rc = sb_take_block(ctx, n, o, kEvHeaderBytes + N * sizeof(svo_hip_seed_event), &sb->blk);
  sb->dev = sb->blk.dev; sb->host = sb->blk.host; sb->host_dev = sb->blk.host_dev;
  if (rc == SVO_HIP_OK) {
    char* d = sb->dev;
    sb->px = (double*)(d + o_px); sb->f = (double*)(d + o_f); sb->level = (int32_t*)(d + o_lvl);
    sb->a = (float*)(d + o_a); sb->b = (float*)(d + o_b); sb->mu = (float*)(d + o_mu); sb->z_range = (float*)(d + o_zr);
    sb->sigma2 = (float*)(d + o_s2); sb->xyz = (double*)(d + o_xyz); sb->px_cur = (double*)(d + o_pc);
    sb->status = (int32_t*)(d + o_st); sb->alive = (uint8_t*)(d + o_al); sb->block_count = (int*)(d + o_bc); sb->hist = (int*)(d + o_h);
    sb->events_dev = (svo_hip_seed_event*)(d + o_ev);
    memset(sb->host, 0, kEvHeaderBytes);
    // the uploaded arrays gathered in the context's page-locked staging area: one transfer
    char* hs = nullptr;
    rc = svo_ctx_host_staging(ctx, upload_bytes, &hs);
    if (rc == SVO_HIP_OK) {
      memcpy(hs + o_px, px, 16 * N); memcpy(hs + o_f, f, 24 * N); memcpy(hs + o_lvl, level, 4 * N); memcpy(hs + o_a, a, 4 * N);
      memcpy(hs + o_b, b, 4 * N); memcpy(hs + o_mu, mu, 4 * N); memcpy(hs + o_zr, z_range, 4 * N); memcpy(hs + o_s2, sigma2, 4 * N);
      hipError_t e = hipMemcpyAsync(d, hs, upload_bytes, hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipMemsetAsync(sb->alive, 1, N, ctx->stream);
      if (e == hipSuccess) e = hipMemsetAsync(sb->status, 0xff, 4 * N, ctx->stream);       // SVO_HIP_SEED_ERASED until the first pass
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);                          // the staging area is free again
      if (e != hipSuccess) rc = svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_seed_batch_create", hipGetErrorString(e));
    }
  }
  if (rc != SVO_HIP_OK) { svo_hip_seed_batch_destroy(sb); return rc; }
  *out = sb;
  return SVO_HIP_OK;
}

int svo_hip_seed_batch_destroy(svo_hip_seed_batch* sb) {
  if (!sb) return SVO_HIP_OK;
  if (sb->ctx) { (void)hipSetDevice(sb->ctx->device); if (sb->pending) (void)hipStreamSynchronize(sb->ctx->stream); }
  // the block goes back to the context's pool: no hipFree (a device-wide synchronisation) at keyframe rate
#ifndef SVO_NO_SEED_POOL          // (the A/B build of tools/ab_churn.sh frees every block: what rounds 3-4 did)
  if (sb->ctx && sb->blk.dev && sb->blk.host) { sb->ctx->seed_pool.push_back(sb->blk); --sb->ctx->seed_blocks_in_use; }
  else
#else
  if (sb->ctx && sb->blk.dev) { --sb->ctx->seed_blocks_in_use; sb->ctx->n_frees += 2; }
#endif
  {
    if (sb->blk.dev) (void)hipFree(sb->blk.dev);
    if (sb->blk.host) (void)hipHostFree(sb->blk.host);
  }
  delete sb;
  return SVO_HIP_OK;
}

int svo_hip_seed_batch_size(const svo_hip_seed_batch* sb, int* n, int* n_alive) {
  if (!sb) return SVO_HIP_ERR_INVALID;
  if (n) *n = sb->n;
  if (n_alive) *n_alive = sb->n_alive;
  return SVO_HIP_OK;
}

static int sb_enqueue_jobs(svo_hip_ctx* ctx, int n_jobs, svo_hip_seed_batch* const* batches, const svo_hip_pyramid* ref, const int* ref_slots,
                           const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam, const double* T_ref_w,
                           const double T_cur_w[7], const svo_hip_df_params* prm, int report_updated);
static int sb_check_pass_args(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam,
                              const svo_hip_df_params* prm);

int svo_hip_seed_batch_update_async(svo_hip_seed_batch* sb, const svo_hip_pyramid* ref, int ref_slot, const svo_hip_pyramid* cur,
                                    int cur_slot, const svo_hip_camera* cam, const double T_ref_w[7], const double T_cur_w[7],
                                    const svo_hip_df_params* prm, int report_updated) {
  if (!sb) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = sb->ctx;
  if (sb->pending) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_seed_batch_update_async", "the previous pass has not been collected");
  if (sb->n_blocks * 256 <= ctx->df_small_max) {           // a small batch: the two-launch form (sb_enqueue_jobs)
    if (!ref || !cur || !cam || !T_ref_w || !T_cur_w || !prm) return SVO_HIP_ERR_INVALID;
    SVO_REQUIRE(ctx, ref_slot >= 0 && ref_slot < ref->batch);
    const int rc = sb_check_pass_args(ctx, ref, cur, cur_slot, cam, prm);
    if (rc != SVO_HIP_OK) return rc;
    SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    return sb_enqueue_jobs(ctx, 1, &sb, ref, &ref_slot, cur, cur_slot, cam, T_ref_w, T_cur_w, prm, report_updated);
  }
  DfEvents ev;
  ev.alive = sb->alive; ev.report_updated = report_updated ? 1 : 0; ev.block_count = sb->block_count; ev.hist = sb->hist;
  const int rc = df_run_pass(ctx, ref, ref_slot, cur, cur_slot, cam, T_ref_w, T_cur_w, sb->n, sb->px, sb->f, sb->level, sb->a, sb->b, sb->mu,
                             sb->z_range, sb->sigma2, prm, sb->status, nullptr, sb->xyz, nullptr, nullptr, sb->px_cur, nullptr, &ev);
  if (rc != SVO_HIP_OK) return rc;
  hipLaunchKernelGGL(ev_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, sb->n_blocks, sb->block_count, sb->hist,
                     reinterpret_cast<EvHeader*>(sb->host_dev), sb->hist + 8);
  hipLaunchKernelGGL(ev_scatter_kernel, dim3(sb->n_blocks), dim3(256), 0, ctx->stream, sb->n, ev.report_updated, sb->status, sb->mu,
                     sb->sigma2, sb->xyz, sb->px_cur, sb->block_count, sb->hist + 8,
                     reinterpret_cast<svo_hip_seed_event*>(sb->host_dev + kEvHeaderBytes), sb->events_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  sb->pending = true;
  sb->report_updated = ev.report_updated;
  return SVO_HIP_OK;
}

// one launch set over at most DF_GROUP_MAX batches (arguments checked by the callers)
static int sb_enqueue_jobs(svo_hip_ctx* ctx, int n_jobs, svo_hip_seed_batch* const* batches, const svo_hip_pyramid* ref, const int* ref_slots,
                           const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam, const double* T_ref_w,
                           const double T_cur_w[7], const svo_hip_df_params* prm, int report_updated) {
  const uint8_t* cur_img = cur->base + (size_t)cur_slot * cur->pyr_bytes;
  const bool prof = ctx->df_profile;
  auto stamp = [&](int k) { if (prof) (void)hipEventRecord(ctx->df_ev[k], ctx->stream); };
  DfFrame fr;
  df_make_frame(ref, cur, cam, T_ref_w, T_cur_w, prm, fr);
  DfGroup g;
  memset(&g, 0, sizeof(g));
  g.n_jobs = n_jobs;
  g.report_updated = report_updated ? 1 : 0;
  int n_blocks = 0, n_cat = 0;
  for (int j = 0; j < n_jobs; ++j) {
    svo_hip_seed_batch* sb = batches[j];
    DfJob& J = g.job[j];
    DfFrame fj;
    df_make_frame(ref, cur, cam, T_ref_w + 7 * (size_t)j, T_cur_w, prm, fj);
    memcpy(J.T_ref_cur, fj.T_ref_cur, sizeof(J.T_ref_cur)); memcpy(J.T_cur_ref_vis, fj.T_cur_ref_vis, sizeof(J.T_cur_ref_vis));
    memcpy(J.T_cur_ref, fj.T_cur_ref, sizeof(J.T_cur_ref)); memcpy(J.T_ref_inv, fj.T_ref_inv, sizeof(J.T_ref_inv));
    J.px = sb->px; J.f = sb->f; J.level = sb->level; J.a = sb->a; J.b = sb->b; J.mu = sb->mu; J.z_range = sb->z_range; J.sigma2 = sb->sigma2;
    J.status = sb->status; J.xyz = sb->xyz; J.px_cur = sb->px_cur; J.alive = sb->alive; J.block_count = sb->block_count; J.hist = sb->hist;
    J.header = reinterpret_cast<EvHeader*>(sb->host_dev);
    J.events_host = reinterpret_cast<svo_hip_seed_event*>(sb->host_dev + kEvHeaderBytes);
    J.events_dev = sb->events_dev;
    J.n = sb->n; J.n_blocks = sb->n_blocks; J.first_block = n_blocks; J.ref_slot = ref_slots[j];
    n_cat = n_blocks * 256 + sb->n;
    n_blocks += sb->n_blocks;
  }
  // scratch of the pass: records (every block of the thread-per-seed stages whole), the transposed patches, the levels
  const int n_rec = n_blocks * 256, n_pad = n_rec;
  const size_t rec_bytes = (size_t)n_rec * sizeof(SeedRec), pwb_bytes = (size_t)25 * n_pad * sizeof(uint32_t);
  void* ws = nullptr;
  {
    const int rc = svo_ctx_scratch(ctx, rec_bytes + pwb_bytes + (size_t)n_rec * sizeof(int32_t), &ws);
    if (rc != SVO_HIP_OK) return rc;
  }
  SeedRec* recs = (SeedRec*)ws;
  uint32_t* pwb_t = reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + rec_bytes);
  int32_t* level_cat = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + rec_bytes + pwb_bytes);
  stamp(0);
  if (n_rec <= ctx->df_small_max) {
    // two launches: a wave takes its four seeds through every stage, then one workgroup per batch does the events
    hipLaunchKernelGGL(df_small_pass_kernel, dim3((n_cat + SEEDS_PER_BLOCK - 1) / SEEDS_PER_BLOCK), dim3(256), 0, ctx->stream, fr, g,
                       ref->base, ref->pyr_bytes, cur_img, n_cat, recs, pwb_t, level_cat, n_pad);
    stamp(1); stamp(2); stamp(3);
    if (prof) { (void)hipEventRecord(ctx->df_ev[4], ctx->stream); ctx->df_ev_recorded = true; }
    hipLaunchKernelGGL(ev_small_kernel, dim3(n_jobs), dim3(1024), 0, ctx->stream, g);
  } else {
    hipLaunchKernelGGL(df_geometry_group_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream, fr, g, recs, level_cat);
    stamp(1);
    hipLaunchKernelGGL(df_search_kernel, dim3((n_cat + SEEDS_PER_BLOCK - 1) / SEEDS_PER_BLOCK), dim3(256), 0, ctx->stream, fr, ref->base,
                       ref->pyr_bytes, cur_img, n_cat, (const int32_t*)level_cat, recs, pwb_t, n_pad);
    stamp(2);
    launch_df_align<false>(ctx, fr, cur_img, n_cat, n_pad, pwb_t, recs);
    stamp(3);
    hipLaunchKernelGGL(df_finalize_group_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream, fr, g, (const SeedRec*)recs);
    if (prof) { (void)hipEventRecord(ctx->df_ev[4], ctx->stream); ctx->df_ev_recorded = true; }
    hipLaunchKernelGGL(ev_scan_group_kernel, dim3(n_jobs), dim3(1024), 0, ctx->stream, g);
    hipLaunchKernelGGL(ev_scatter_group_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream, g);
  }
  SVO_CHECK_HIP(ctx, hipGetLastError());
  for (int j = 0; j < n_jobs; ++j) { batches[j]->pending = true; batches[j]->report_updated = g.report_updated; }
  return SVO_HIP_OK;
}

// the argument checks df_run_pass makes, for the entry points that build their launches themselves
static int sb_check_pass_args(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam,
                              const svo_hip_df_params* prm) {
  SVO_REQUIRE(ctx, cur_slot >= 0 && cur_slot < cur->batch);
  SVO_REQUIRE(ctx, ref->width == cam->width && ref->height == cam->height && cur->width == cam->width && cur->height == cam->height);
  SVO_REQUIRE(ctx, prm->n_pyr_levels >= 1 && prm->n_pyr_levels <= ref->n_levels && prm->n_pyr_levels <= cur->n_levels);
  SVO_REQUIRE(ctx, prm->align_max_iter >= 0 && prm->max_epi_search_steps >= 0);
  return SVO_HIP_OK;
}

int svo_hip_seed_batch_update_group_async(int n_batches, svo_hip_seed_batch* const* batches, const svo_hip_pyramid* ref,
                                          const int* ref_slots, const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam,
                                          const double* T_ref_w, const double T_cur_w[7], const svo_hip_df_params* prm,
                                          int report_updated) {
  if (n_batches < 1 || !batches || !batches[0]) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = batches[0]->ctx;
  if (!ref || !ref_slots || !cur || !cam || !T_ref_w || !T_cur_w || !prm) return svo_fail(ctx, SVO_HIP_ERR_INVALID, "svo_hip_seed_batch_update_group_async", "null argument");
  if (n_batches == 1)
    return svo_hip_seed_batch_update_async(batches[0], ref, ref_slots[0], cur, cur_slot, cam, T_ref_w, T_cur_w, prm, report_updated);
  // every batch's arguments are checked before anything is enqueued; up to DF_GROUP_MAX batches per set of launches
  for (int k = 0; k < n_batches; ++k) {
    svo_hip_seed_batch* sb = batches[k];
    SVO_REQUIRE(ctx, sb && sb->ctx == ctx);
    if (sb->pending) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_seed_batch_update_group_async", "the previous pass has not been collected");
    for (int m = 0; m < k; ++m) SVO_REQUIRE(ctx, batches[m] != sb);
    SVO_REQUIRE(ctx, ref_slots[k] >= 0 && ref_slots[k] < ref->batch);
  }
  {
    const int rc = sb_check_pass_args(ctx, ref, cur, cur_slot, cam, prm);
    if (rc != SVO_HIP_OK) return rc;
  }
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  for (int first = 0; first < n_batches; first += DF_GROUP_MAX) {
    const int n_jobs = n_batches - first < DF_GROUP_MAX ? n_batches - first : DF_GROUP_MAX;
    const int rc = sb_enqueue_jobs(ctx, n_jobs, batches + first, ref, ref_slots + first, cur, cur_slot, cam, T_ref_w + 7 * (size_t)first, T_cur_w,
                                   prm, report_updated);
    if (rc != SVO_HIP_OK) return rc;
  }
  return SVO_HIP_OK;
}

int svo_hip_df_set_small_pass_limit(svo_hip_ctx* ctx, int max_seeds) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, max_seeds >= 0 && max_seeds <= DF_SMALL_MAX);
  ctx->df_small_max = max_seeds;
  return SVO_HIP_OK;
}

int svo_hip_seed_batch_pending(const svo_hip_seed_batch* sb) { return sb && sb->pending ? 1 : 0; }

int svo_hip_seed_batch_collect(svo_hip_seed_batch* sb, const svo_hip_seed_event** events, int* n_events, int32_t status_counts[7]) {
  if (!sb) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = sb->ctx;
  if (!sb->pending) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_seed_batch_collect", "no pass has been enqueued");
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  sb->pending = false;
  const EvHeader* h = reinterpret_cast<const EvHeader*>(sb->host);
  if (h->n_events > EV_DIRECT_MAX) {                      // many events (a keyframe): packed on the device, one transfer
    SVO_CHECK_HIP(ctx, hipMemcpyAsync(sb->host + kEvHeaderBytes, sb->events_dev, (size_t)h->n_events * sizeof(svo_hip_seed_event),
                                      hipMemcpyDeviceToHost, ctx->stream));
    SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (events) *events = reinterpret_cast<const svo_hip_seed_event*>(sb->host + kEvHeaderBytes);
  if (n_events) *n_events = h->n_events;
  if (status_counts) for (int k = 0; k < 7; ++k) status_counts[k] = h->counts[k];
  sb->n_alive -= h->counts[SVO_HIP_SEED_CONVERGED + 1] + h->counts[SVO_HIP_SEED_NAN + 1];
  return SVO_HIP_OK;
}

int svo_hip_seed_batch_erase(svo_hip_seed_batch* sb, int n, const int32_t* indices) {
  if (!sb) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = sb->ctx;
  SVO_REQUIRE(ctx, n >= 0 && (n == 0 || indices));
  if (sb->pending) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_seed_batch_erase", "collect the enqueued pass first");
  if (n == 0) return SVO_HIP_OK;
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  // the alive flags come down, are edited and go back: erasures are rare (removeKeyframe) and this keeps n_alive exact
  std::vector<uint8_t> al((size_t)sb->n);
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(al.data(), sb->alive, (size_t)sb->n, hipMemcpyDeviceToHost, ctx->stream));
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < n; ++k) {
    SVO_REQUIRE(ctx, indices[k] >= 0 && indices[k] < sb->n);
    if (al[(size_t)indices[k]]) { al[(size_t)indices[k]] = 0; --sb->n_alive; }
  }
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(sb->alive, al.data(), (size_t)sb->n, hipMemcpyHostToDevice, ctx->stream));
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_HIP_OK;
}

int svo_hip_seed_batch_download(svo_hip_seed_batch* sb, float* a, float* b, float* mu, float* sigma2, uint8_t* alive) {
  if (!sb) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = sb->ctx;
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = (size_t)sb->n;
  if (a) SVO_CHECK_HIP(ctx, hipMemcpyAsync(a, sb->a, 4 * N, hipMemcpyDeviceToHost, ctx->stream));
  if (b) SVO_CHECK_HIP(ctx, hipMemcpyAsync(b, sb->b, 4 * N, hipMemcpyDeviceToHost, ctx->stream));
  if (mu) SVO_CHECK_HIP(ctx, hipMemcpyAsync(mu, sb->mu, 4 * N, hipMemcpyDeviceToHost, ctx->stream));
  if (sigma2) SVO_CHECK_HIP(ctx, hipMemcpyAsync(sigma2, sb->sigma2, 4 * N, hipMemcpyDeviceToHost, ctx->stream));
  if (alive) SVO_CHECK_HIP(ctx, hipMemcpyAsync(alive, sb->alive, N, hipMemcpyDeviceToHost, ctx->stream));
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_HIP_OK;
}

int svo_hip_seed_batch_arrays(svo_hip_seed_batch* sb, float** a_dev, float** b_dev, float** mu_dev, float** sigma2_dev, int32_t** status_dev) {
  if (!sb) return SVO_HIP_ERR_INVALID;
  if (a_dev) *a_dev = sb->a;
  if (b_dev) *b_dev = sb->b;
  if (mu_dev) *mu_dev = sb->mu;
  if (sigma2_dev) *sigma2_dev = sb->sigma2;
  if (status_dev) *status_dev = sb->status;
  return SVO_HIP_OK;
}

}  // extern "C"
