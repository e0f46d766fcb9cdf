// svo_refine.hip -- next rows of the hot path: the two small Gauss-Newton refinements that follow
// SparseImgAlign and the reprojection matching in FrameHandlerMono::processFrame.
//
//   pose_refine_kernel   pose_optimizer::optimizeGaussNewton   S/pose_optimizer.cpp:31-181
//   point_refine_kernel  Point::optimize                       S/point.cpp:130-192
//   ldlt6_batch_kernel   Eigen LDLT 6x6 solve as used by both solvers (test utility: device == Eigen bit for bit)
//
// pose_refine_kernel: one workgroup (256 threads) per frame runs the whole refinement in one launch -- MAD scale
// (exact k-th element by radix selection on the float bit pattern), <= n_iter robust Gauss-Newton steps (thread per
// observation, fp64 2x6 Jacobians, Tukey weights in f32 as the reference, transposing wave reduction of the 28
// sums, one-lane pivoted LDL^T + SE3::exp), covariance (one-lane partial-pivot LU inverse), outlier test, and the
// two medians of the squared errors (radix selection on the f64 bit pattern).  Sums over observations are taken
// in a fixed tree order instead of the reference's list order: results agree to rounding, run to run bit-equal.
#include "svo_internal.h"
#include "svo_point_refine.h"

namespace {

using namespace svo_dev;

constexpr int PR_THREADS = 256;
constexpr int PR_WAVES = PR_THREADS / 64;
constexpr int PR_CACHED = 8;                    // feature slots per thread kept in registers (frames up to 2048 features)

struct PoseOptOut {               // == svo_hip_pose_opt_result
  int ran;
  int n_iter_done;
  int n_deleted;
  int pad_;
  unsigned long long num_obs;
  double T_f_w[7];
  double estimated_scale, error_init, error_final;
  double Cov[36];
};
static_assert(sizeof(PoseOptOut) == sizeof(svo_hip_pose_opt_result), "layout of svo_hip_pose_opt_result");

// TukeyWeightFunction::value with DEFAULT_B (S/robust_cost.cpp:87-106)
SVO_DEV float tukey_weight(float x) {
  const float b = 8.6851f;
  const float b_square = b * b;
  const float x_square = x * x;
  if (x_square <= b_square) {
    const float tmp = 1.0f - x_square / b_square;
    return tmp * tmp;
  }
  return 0.0f;
}

// reprojection error on the unit plane, scaled by 1/2^level (:54-57, :92-95, :151-153)
SVO_DEV void unit_plane_error(const double* T, const double* f, const double* pos, int level, double* e, double* xyz) {
  se3_act(T, pos, xyz);
  e[0] = f[0] / f[2] - xyz[0] / xyz[2];
  e[1] = f[1] / f[2] - xyz[1] / xyz[2];
  const double s = 1.0 / (1 << level);
  e[0] *= s; e[1] *= s;
}

// k-th smallest (0-based) of the keys for_each_key enumerates (every thread its own), all threads of the block take
// part and get the key.
// Keys are bit patterns of non-negative IEEE numbers (monotonic as unsigned integers).  hist: 256 ints of LDS.
template <typename KeyT, typename ForEachKey>
__device__ KeyT block_radix_select(unsigned k, ForEachKey for_each_key, int* hist, KeyT* s_prefix, unsigned* s_k) {
  constexpr int BITS = 8 * (int)sizeof(KeyT);
  if (threadIdx.x == 0) { *s_prefix = 0; *s_k = k; }
  for (int shift = BITS - 8; shift >= 0; shift -= 8) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const KeyT prefix = *s_prefix;
    for_each_key([&](KeyT key) {
      const bool match = (shift == BITS - 8) ? true : ((key >> (shift + 8)) == prefix);
      if (match) atomicAdd(&hist[(int)((key >> shift) & 0xff)], 1);
    });
    __syncthreads();
    if (threadIdx.x < 64) {                    // wave 0: 4 bins per lane, exclusive scan, pick the digit
      const int lane = threadIdx.x;
      const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const int mine = h0 + h1 + h2 + h3;
      int incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      const unsigned before = (unsigned)(incl - mine);
      const unsigned kk = *s_k;
      if (kk >= before && kk < before + (unsigned)mine) {          // exactly one lane
        unsigned r = kk - before;
        int d = 4 * lane;
        if (r >= (unsigned)h0) { r -= h0; ++d; if (r >= (unsigned)h1) { r -= h1; ++d; if (r >= (unsigned)h2) { r -= h2; ++d; } } }
        *s_prefix = (KeyT)((prefix << 8) | (KeyT)d);
        *s_k = r;
      }
    }
    __syncthreads();
  }
  return *s_prefix;
}

// The same k-th smallest key for a frame with at most one observation per thread (n <= PR_THREADS: what a tracked frame
// has -- a few hundred matches), for NS key sets at once (same k, same threads holding keys): the keys go to LDS and
// every key is ranked against all of them -- it is the answer when (number of keys below it) <= k < (number of keys not
// above it); equal keys all qualify and carry the same value.  A frame of <= 128 (<= 64) keys splits the scan of a key
// over 2 (4) threads, the LDS reads are issued eight keys ahead of their use: two barriers and ~2 k cycles instead of
// four or eight histogram passes of three barriers each.
// has_key / key[NS]: this thread's keys (a thread without one passes has_key = false).
template <int NS>
__device__ void block_rank_select(unsigned k, bool has_key, const unsigned long long* key, int n,
                                  unsigned long long (*s_keys)[PR_THREADS], unsigned (*s_cnt)[PR_THREADS],
                                  unsigned long long* s_out, unsigned long long* out) {
  const unsigned long long none = ~0ull;                   // above every key (bit patterns of non-negative numbers)
  const int t = threadIdx.x;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    s_keys[s][t] = has_key ? key[s] : none;
    s_cnt[2 * s][t] = 0; s_cnt[2 * s + 1][t] = 0;
  }
  __syncthreads();
  const int P = n <= PR_THREADS / 4 ? 4 : (n <= PR_THREADS / 2 ? 2 : 1);
  const int per = PR_THREADS / P;
  const int ki = t & (per - 1), part = t / per;
  const int n8 = (n + 7) & ~7;
  const int chunk = ((n8 / P) + 7) & ~7;
  const int j0 = part * chunk, j1 = (j0 + chunk < n8) ? j0 + chunk : n8;
  unsigned long long mine[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) mine[s] = s_keys[s][ki];
  if (ki < n) {
    unsigned lt[NS], le[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { lt[s] = 0; le[s] = 0; }
    for (int j = j0; j < j1; j += 8) {
      ulonglong2 o[NS][4];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[s][e] = *reinterpret_cast<const ulonglong2*>(&s_keys[s][j + 2 * e]);
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lt[s] += (o[s][e].x < mine[s] ? 1u : 0u) + (o[s][e].y < mine[s] ? 1u : 0u);
          le[s] += (o[s][e].x <= mine[s] ? 1u : 0u) + (o[s][e].y <= mine[s] ? 1u : 0u);
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) { atomicAdd(&s_cnt[2 * s][ki], lt[s]); atomicAdd(&s_cnt[2 * s + 1][ki], le[s]); }
  }
  __syncthreads();
  if (part == 0 && ki < n) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (mine[s] != none && s_cnt[2 * s][ki] <= k && k < s_cnt[2 * s + 1][ki]) s_out[s] = mine[s];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NS; ++s) out[s] = s_out[s];
}

// Matrix<double,6,6>::inverse() = partialPivLu().inverse() (Eigen LU/PartialPivLU.h:379-425), column `col` of it:
// the columns are independent substitutions behind a shared factorisation, so six lanes of a wave take one each (all of
// them factor the same matrix -- the pivots are wave-uniform and read with readfirstlane, which keeps every index static
// and the matrix in registers).  Same arithmetic per column as the serial loop over the columns.
SVO_DEV void inverse6_column(const double* Ain, int col, double* d) {
  constexpr int N = 6;
  double lu[N][N];
  int piv[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) lu[i][j] = Ain[i * N + j];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    int big = k;
    double best = fabs(lu[k][k]);
#pragma unroll
    for (int i = k + 1; i < N; ++i)
      if (fabs(lu[i][k]) > best) { best = fabs(lu[i][k]); big = i; }
    big = __builtin_amdgcn_readfirstlane(big);
    piv[k] = big;
    if (best != 0.0) {
#pragma unroll
      for (int c = k + 1; c < N; ++c) {
        if (big == c) {
#pragma unroll
          for (int j = 0; j < N; ++j) { const double t = lu[k][j]; lu[k][j] = lu[c][j]; lu[c][j] = t; }
        }
      }
#pragma unroll
      for (int i = k + 1; i < N; ++i) lu[i][k] /= lu[k][k];
    }
#pragma unroll
    for (int i = k + 1; i < N; ++i)
#pragma unroll
      for (int j = k + 1; j < N; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
  }
#pragma unroll
  for (int i = 0; i < N; ++i) d[i] = (i == col) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int c = k + 1; c < N; ++c)
      if (piv[k] == c) { const double t = d[k]; d[k] = d[c]; d[c] = t; }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double acc = d[i];
#pragma unroll
    for (int j = 0; j < i; ++j) acc -= lu[i][j] * d[j];
    d[i] = acc;
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double acc = d[i];
#pragma unroll
    for (int j = i + 1; j < N; ++j) acc -= lu[i][j] * d[j];
    d[i] = acc / lu[i][i];
  }
}

// se3_exp_series_table (svo_device_math.h), [step][column]
__device__ __constant__ double kExpSeries[32] = {
    0.0,         0.0,         1.0 / 272.0, 0.0,
    1.0 / 272.0, 1.0 / 240.0, 1.0 / 210.0, 1.0 / 240.0,
    1.0 / 210.0, 1.0 / 182.0, 1.0 / 156.0, 1.0 / 182.0,
    1.0 / 156.0, 1.0 / 132.0, 1.0 / 110.0, 1.0 / 132.0,
    1.0 / 110.0, 1.0 / 90.0,  1.0 / 72.0,  1.0 / 90.0,
    1.0 / 72.0,  1.0 / 56.0,  1.0 / 42.0,  1.0 / 56.0,
    1.0 / 42.0,  1.0 / 30.0,  1.0 / 20.0,  1.0 / 30.0,
    1.0 / 20.0,  1.0 / 12.0,  1.0 / 6.0,   1.0 / 12.0};
// large update angles (theta^2 > 0.25): the library path of SE3::exp
__device__ __noinline__ void se3_exp_cold(const double* l, double* out) { se3_exp(l, out); }

__global__ __launch_bounds__(PR_THREADS) void pose_refine_kernel(
    int max_n, const int* __restrict__ n_feat, const double* __restrict__ T_in, const double* __restrict__ f,
    const double* __restrict__ pos, const int* __restrict__ level, uint8_t* __restrict__ has_point, double em,
    double reproj_thresh, int n_iter, float* __restrict__ err_ws, double* __restrict__ sq_init_ws,
    double* __restrict__ sq_final_ws, PoseOptOut* __restrict__ out, int n_feat_stride = 1, int T_stride = 7) {
  __shared__ int hist[256];
  __shared__ __attribute__((aligned(16))) unsigned long long s_keys[2][PR_THREADS];
  __shared__ unsigned s_cnt[4][PR_THREADS];
  __shared__ unsigned long long s_sel[2];
  __shared__ double s_Ac[36], s_cov[36];
  __shared__ unsigned s_k;
  __shared__ unsigned s_pref32;
  __shared__ unsigned long long s_pref64;
  __shared__ double red[PR_WAVES][32];
  __shared__ double s_S[32];
  __shared__ double s_T[7], s_Told[7];
  __shared__ double s_chi2, s_scale;
  __shared__ int s_done, s_iters;
  __shared__ unsigned s_count;

  const int b = blockIdx.x;
  const int n = n_feat[(size_t)b * n_feat_stride];      // (strides: a tracker group reads the counts / poses where its other kernels left them)
  const size_t base = (size_t)b * max_n;
  const double* fb = f + 3 * base;
  const double* pb = pos + 3 * base;
  const int* lb = level + base;
  uint8_t* hb = has_point + base;
  float* err = err_ws + base;
  double* sq_init = sq_init_ws + base;
  double* sq_final = sq_final_ws + base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // The observations are read 13 times (scale, 10 Gauss-Newton steps, outlier test, ...): the first PR_CACHED ones of
  // every thread stay in registers -- with one workgroup per frame a pass over them from memory is a chain of
  // dependent latencies, not bandwidth -- and only what lies beyond (frames above 2048 features) is re-read.
  double cf[PR_CACHED][3], cp[PR_CACHED][3];
  int cl[PR_CACHED];
  bool ch[PR_CACHED], ch0[PR_CACHED];                       // has a point now / had one on entry
  float cerr[PR_CACHED];                                     // keys of the three medians
  double csqi[PR_CACHED], csqf[PR_CACHED];
#pragma unroll
  for (int j = 0; j < PR_CACHED; ++j) {
    const int i = threadIdx.x + PR_THREADS * j;
    const bool in = i < n;
    const int ii = in ? i : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { cf[j][c] = in ? fb[3 * ii + c] : 0.0; cp[j][c] = in ? pb[3 * ii + c] : 0.0; }
    cl[j] = in ? lb[ii] : 0;
    ch[j] = in && hb[ii] != 0;
    ch0[j] = ch[j];
    cerr[j] = 0.0f; csqi[j] = 0.0; csqf[j] = -1.0;
  }
  // body(i, j, f, pos, level, has_point&) for every feature slot of this thread; j = register slot, -1 beyond
  auto for_each_obs = [&](auto&& body) {
#pragma unroll
    for (int j = 0; j < PR_CACHED; ++j) {
      const int i = threadIdx.x + PR_THREADS * j;
      if (i < n) body(i, j, cf[j], cp[j], cl[j], ch[j]);
    }
    for (int i = threadIdx.x + PR_THREADS * PR_CACHED; i < n; i += PR_THREADS) {
      bool hp = hb[i] != 0;
      body(i, -1, fb + 3 * i, pb + 3 * i, lb[i], hp);
    }
  };

  if (threadIdx.x == 0) {
    for (int i = 0; i < 7; ++i) { s_T[i] = T_in[(size_t)T_stride * b + i]; s_Told[i] = s_T[i]; }            // :45
    s_chi2 = 0.0; s_done = 0; s_iters = 0; s_count = 0;
  }
  if (lane >= 28 && lane < 32) red[wave][lane] = 0.0;
  __syncthreads();

  // ---- scale of the error for the robust weights (:51-66)
  {
    double T[7];
    for (int i = 0; i < 7; ++i) T[i] = s_T[i];
    unsigned mine = 0;
    for_each_obs([&](int i, int j, const double* fo, const double* po, int lv, bool& hp) {
      if (!hp) return;
      double e[2], xyz[3];
      unit_plane_error(T, fo, po, lv, e, xyz);
      const float ef = (float)sqrt(e[0] * e[0] + e[1] * e[1]);
      if (j >= 0) cerr[j < 0 ? 0 : j] = ef; else err[i] = ef;
      ++mine;
    });
    if (mine) atomicAdd(&s_count, mine);
  }
  __syncthreads();
  const unsigned n_obs = s_count;
  PoseOptOut& o = out[b];
  if (n_obs == 0) {                                                                            // :61-62
    if (threadIdx.x == 0) {
      o.ran = 0; o.n_iter_done = 0; o.n_deleted = 0; o.num_obs = 0;
      for (int i = 0; i < 7; ++i) o.T_f_w[i] = s_T[i];
      o.estimated_scale = 0.0; o.error_init = 0.0; o.error_final = 0.0;
      for (int i = 0; i < 36; ++i) o.Cov[i] = 0.0;
    }
    return;
  }
  const bool small = n <= PR_THREADS;                        // block-uniform: one observation per thread at most
  unsigned long long med_small = 0;
  if (small) {
    const unsigned long long key = (unsigned long long)__float_as_uint(cerr[0]);
    block_rank_select<1>(n_obs / 2, ch[0], &key, n, s_keys, s_cnt, s_sel, &med_small);
  }
  const unsigned med_bits = small
      ? (unsigned)med_small
      : block_radix_select<unsigned>(
      n_obs / 2, [&](auto&& emit) {
#pragma unroll
        for (int j = 0; j < PR_CACHED; ++j) if (ch[j]) emit(__float_as_uint(cerr[j]));
        for (int i = threadIdx.x + PR_THREADS * PR_CACHED; i < n; i += PR_THREADS) if (hb[i]) emit(__float_as_uint(err[i]));
      }, hist, &s_pref32, &s_k);
  const double estimated_scale = (double)(1.48f * __uint_as_float(med_bits));                  // MADScaleEstimator
  if (threadIdx.x == 0) { s_scale = estimated_scale; s_count = 0; }
  __syncthreads();

  int tri_r = 0, tri_c = 0;                                  // lane < 21: its entry of the upper triangle, row-major
  {
    int kk = lane < 21 ? lane : 0;
    while (kk >= 6 - tri_r) { kk -= 6 - tri_r; ++tri_r; }
    tri_c = tri_r + kk;
  }
  // Cov_ = (A em^2)^-1 (:141) of the step's A, on six lanes of the last wave while wave 0 solves the step (the waves
  // without the solve wait at the barrier anyway): what the last step leaves in s_cov is reported.
  auto covariance_of = [&](bool from_partials) {
    if (lane < 21) {
      double v = 0.0;
      if (from_partials)
        for (int w = 0; w < PR_WAVES; ++w) v += red[w][lane];
      s_Ac[tri_r * 6 + tri_c] = v; s_Ac[tri_c * 6 + tri_r] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < 6) {
      double As[36], col[6];
      const double em2 = em * em;                            // pow(em, 2)
#pragma unroll
      for (int k = 0; k < 36; ++k) As[k] = s_Ac[k] * em2;
      inverse6_column(As, lane, col);
#pragma unroll
      for (int i = 0; i < 6; ++i) s_cov[i * 6 + lane] = col[i];
    }
  };

  // ---- robust Gauss-Newton (:70-138)
  for (int iter = 0; iter < n_iter; ++iter) {
    if (s_done) break;                                       // block-uniform
    double T[7];
    for (int i = 0; i < 7; ++i) T[i] = s_T[i];
    const double scale = (iter >= 5) ? 0.85 / em : s_scale;  // :74-75 (the overwrite at iteration 5 stays)
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
    for_each_obs([&](int i, int j, const double* fo, const double* po, int lv, bool& hp) {
      if (!hp) return;
      double e[2], xyz[3];
      unit_plane_error(T, fo, po, lv, e, xyz);
      const double sqrt_inv_cov = 1.0 / (1 << lv);
      const double sq = e[0] * e[0] + e[1] * e[1];
      if (iter == 0) { if (j >= 0) csqi[j < 0 ? 0 : j] = sq; else sq_init[i] = sq; }
      // Frame::jacobian_xyz2uv (I/frame.h:110-132) times sqrt_inv_cov
      const double x = xyz[0], y = xyz[1];
      const double z_inv = 1. / xyz[2];
      const double z_inv_2 = z_inv * z_inv;
      double J0[6], J1[6];
      J0[0] = -z_inv; J0[1] = 0.0; J0[2] = x * z_inv_2; J0[3] = y * J0[2]; J0[4] = -(1.0 + x * J0[2]); J0[5] = y * z_inv;
      J1[0] = 0.0; J1[1] = -z_inv; J1[2] = y * z_inv_2; J1[3] = 1.0 + y * J1[2]; J1[4] = -J0[3]; J1[5] = -x * z_inv;
#pragma unroll
      for (int k = 0; k < 6; ++k) { J0[k] *= sqrt_inv_cov; J1[k] *= sqrt_inv_cov; }
      const double weight = (double)tukey_weight((float)(sqrt(sq) / scale));
      int kk = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) acc[kk++] += (J0[r] * J0[c] + J1[r] * J1[c]) * weight;          // A += J^T J w
#pragma unroll
      for (int r = 0; r < 6; ++r) acc[21 + r] -= (J0[r] * e[0] + J1[r] * e[1]) * weight;              // b -= J^T e w
      acc[27] += sq * weight;
    });
    // 28 sums: four transposing wave reductions of 8 (lanes 8j..8j+7 get value j), then the waves in fixed order
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double v8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v8[k] = (8 * g + k < 28) ? acc[8 * g + k] : 0.0;
      const double t = wave_reduce8(v8);
      if ((lane & 7) == 0 && 8 * g + (lane >> 3) < 28) red[wave][8 * g + (lane >> 3)] = t;
    }
    // wave 0 asks for what the step needs and is known already before the barrier (it runs the serial part alone)
    double coef[8], Tc[7];
    double chi2_old = 0.0;
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) coef[k] = kExpSeries[k * 4 + (lane & 3)];
#pragma unroll
      for (int i = 0; i < 7; ++i) Tc[i] = s_T[i];
      chi2_old = s_chi2;
    }
    __syncthreads();
    if (wave == PR_WAVES - 1) covariance_of(true);
    if (wave == 0) {
      double v = 0.0;                                          // the waves in fixed order, one sum per lane
      if (lane < 28)
        for (int w = 0; w < PR_WAVES; ++w) v += red[w][lane];
      s_S[lane < 28 ? lane : 28] = v;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      double dT0[6] = {0, 0, 0, 0, 0, 0};
      if (lane == 0) {
        double S[27];
        for (int k = 0; k < 27; ++k) S[k] = s_S[k];
        double A[36], bb[6];
        int kk = 0;
        for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { A[r * 6 + c] = S[kk]; A[c * 6 + r] = S[kk]; ++kk; }
        for (int r = 0; r < 6; ++r) bb[r] = S[21 + r];
        ldlt6_solve_reg(A, bb, dT0);
      }
      double dT[6];                                            // wave-uniform copies
#pragma unroll
      for (int k = 0; k < 6; ++k) dT[k] = readlane_f64(dT0[k], 0);
      const double new_chi2 = readlane_f64(v, 27);
      // the four series of SE3::exp on lanes 0..3 (svo_device_math.h)
      const double zt = dT[3] * dT[3] + dT[4] * dT[4] + dT[5] * dT[5];
      const double ser = se3_exp_series_lane(coef, (lane & 2) ? 0.25 * zt : zt);
      const double qs_t = readlane_f64(ser, 0), pc_t = readlane_f64(ser, 1), ps_h = readlane_f64(ser, 2), pc_h = readlane_f64(ser, 3);
      if (lane == 0) {
        s_iters = iter + 1;
        if ((iter > 0 && new_chi2 > chi2_old * 1.2) || dT[0] != dT[0]) {                         // :106-116
          for (int i = 0; i < 7; ++i) s_T[i] = s_Told[i];
          s_done = 1;
        } else {
          double E[7], Tn[7];
          if (zt <= 0.25) se3_exp_small_finish(dT, zt, qs_t, pc_t, ps_h, pc_h, E);
          else se3_exp_cold(dT, E);
          se3_mul(E, Tc, Tn);                                                                     // exp(dT) * T_f_w (:120)
          for (int i = 0; i < 7; ++i) { s_Told[i] = Tc[i]; s_T[i] = Tn[i]; }
          s_chi2 = new_chi2;
          double mx = -1;
          for (int k = 0; k < 6; ++k) { const double a = fabs(dT[k]); if (a > mx) mx = a; }
          if (mx <= 0.0000000001) s_done = 1;                                                     // EPS
        }
      }
    }
    __syncthreads();
  }

  // ---- covariance (:141), outlier test (:144-159), medians (:161-166)
  if (n_iter <= 0) {                                         // no step ran: the inverse of the zero matrix, as there
    if (wave == PR_WAVES - 1) covariance_of(false);
    __syncthreads();
  }
  if (threadIdx.x < 36) o.Cov[threadIdx.x] = s_cov[threadIdx.x];
  {
    double T[7];
    for (int i = 0; i < 7; ++i) T[i] = s_T[i];
    const double thresh = reproj_thresh / em;
    unsigned deleted = 0;
    for_each_obs([&](int i, int j, const double* fo, const double* po, int lv, bool& hp) {
      if (!hp) { if (j < 0) sq_final[i] = -1.0; return; }    // negative: not an observation
      double e[2], xyz[3];
      unit_plane_error(T, fo, po, lv, e, xyz);
      const double sq = e[0] * e[0] + e[1] * e[1];
      if (j >= 0) csqf[j < 0 ? 0 : j] = sq; else sq_final[i] = sq;
      if (sqrt(sq) > thresh) { hb[i] = 0; hp = false; ++deleted; }
    });
    if (deleted) atomicAdd(&s_count, deleted);
  }
  // the observations of the init/final vectors are those that had a point when the function was entered:
  // sq_final >= 0 marks them (has_point was just cleared for the outliers)
  unsigned long long mi = 0, mf = 0;
  if (small) {                                               // both medians in one ranking pass (its barriers cover s_count)
    const unsigned long long keys[2] = {(unsigned long long)__double_as_longlong(csqi[0]), (unsigned long long)__double_as_longlong(csqf[0])};
    unsigned long long med[2];
    block_rank_select<2>(n_obs / 2, ch0[0], keys, n, s_keys, s_cnt, s_sel, med);
    mi = med[0]; mf = med[1];
  } else {
    __syncthreads();
    mi = block_radix_select<unsigned long long>(
        n_obs / 2, [&](auto&& emit) {
#pragma unroll
          for (int j = 0; j < PR_CACHED; ++j) if (ch0[j]) emit((unsigned long long)__double_as_longlong(csqi[j]));
          for (int i = threadIdx.x + PR_THREADS * PR_CACHED; i < n; i += PR_THREADS)
            if (sq_final[i] >= 0.0) emit((unsigned long long)__double_as_longlong(sq_init[i]));
        }, hist, &s_pref64, &s_k);
    mf = block_radix_select<unsigned long long>(
        n_obs / 2, [&](auto&& emit) {
#pragma unroll
          for (int j = 0; j < PR_CACHED; ++j) if (ch0[j]) emit((unsigned long long)__double_as_longlong(csqf[j]));
          for (int i = threadIdx.x + PR_THREADS * PR_CACHED; i < n; i += PR_THREADS)
            if (sq_final[i] >= 0.0) emit((unsigned long long)__double_as_longlong(sq_final[i]));
        }, hist, &s_pref64, &s_k);
  }
  const unsigned n_deleted = s_count;
  if (threadIdx.x == 0) {
    o.ran = 1;
    o.n_iter_done = s_iters;
    o.n_deleted = (int)n_deleted;
    o.num_obs = (unsigned long long)n_obs - n_deleted;
    for (int i = 0; i < 7; ++i) o.T_f_w[i] = s_T[i];
    o.estimated_scale = estimated_scale * em;
    o.error_init = (n_iter > 0) ? sqrt(__longlong_as_double((long long)mi)) * em : 0.0;       // empty vector when no iteration ran
    o.error_final = sqrt(__longlong_as_double((long long)mf)) * em;
  }
}

// Point::optimize (S/point.cpp:130-192): thread per map point, observations in CSR form, list order kept (svo_point_refine.h)
__global__ void point_refine_kernel(int n_points, int n_iter, double* __restrict__ pos, const int* __restrict__ obs_offset,
                                    const double* __restrict__ obs_T, const double* __restrict__ obs_f,
                                    int* __restrict__ iters_out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_points) return;
  double P[3] = {pos[3 * p], pos[3 * p + 1], pos[3 * p + 2]};
  const int done = point_refine_one(P, obs_offset[p], obs_offset[p + 1], n_iter, [&](int k, double* T, double* fo) {
    for (int t = 0; t < 7; ++t) T[t] = obs_T[7 * k + t];
    fo[0] = obs_f[3 * k]; fo[1] = obs_f[3 * k + 1]; fo[2] = obs_f[3 * k + 2];
  });
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

// the batch entry with the counts n_feat_stride ints and the poses T_stride doubles apart (svo_track.hip: a tracker group's
// cameras keep them in their counter blocks and solver records)
int svo_pose_optimize_batch_strided(svo_hip_ctx* ctx, int batch, int max_n, const int32_t* n_feat_dev, int n_feat_stride,
                                    const double* T_f_w_dev, int T_stride, const double* f_dev, const double* pos_dev,
                                    const int32_t* level_dev, uint8_t* has_point_dev, double error_multiplier2,
                                    double reproj_thresh, int n_iter, svo_hip_pose_opt_result* results_dev) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, batch > 0 && max_n > 0 && n_iter >= 0 && n_feat_stride >= 1 && T_stride >= 7);
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
                     reinterpret_cast<PoseOptOut*>(results_dev), n_feat_stride, T_stride);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}


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
