// svo_nlls.hip -- the branches of vk::NLLSSolver<6, SE3> that svo::SparseImgAlign inherits and no caller of the reference
// switches on (frame_handler_mono.cpp:186-187,331-332): optimizeLevenbergMarquardt (I/nlls_solver_impl.hpp:102-227) and
// the robust cost (setRobustCostFunction :229-281, S/robust_cost.cpp, S/sparse_img_align.cpp:256-263,281-283).
// svo_hip_sia_run hands the solve over to svo_nlls_run when svo_hip_sia_set_option has selected either.
//
// They run on the streaming solver of svo_sia.hip -- its precomputeReferencePatches kernel, its caches, its per-frame state
// record -- with three things of their own:
//   * nlls_residual_robust_kernel: computeResiduals with a weight per pixel (or its weight-scale pass, which leaves the
//     absolute residuals of every pixel in the order of the reference's `errors` vector).  With weights the Hessian of a
//     patch is no longer a per-level constant: sum_px w J J^T = mxx AA^T + mxy (AB^T + BA^T) + myy BB^T with the weighted
//     moments mxx = sum w dx^2 ... formed per evaluation.  It also leaves res*res*weight of every pixel in memory: the
//     control step adds them into ONE f32 in the reference's order (chi2 += res*res*weight, :266), so chi2 -- on which
//     Levenberg-Marquardt accepts or rejects a trial, often by a few units in the last place -- is the reference's f32 sum
//     bit for bit whenever the poses are (the Gauss-Newton hot path sums chi2 per patch and then in f64: faster, and an
//     error-increase exit can flip there).
//   * nlls_scale_kernel: the three scale estimators, in f32 and in the reference's order of operations (the sequential f32
//     sums of TDist and Normal on one lane out of LDS; MAD by a radix select over the bit patterns), so scale_ is the
//     reference's bit for bit.
//   * nlls_solve_kernel: the control step of one frame after an evaluation -- Gauss-Newton (gn_control_step, plus the
//     bookkeeping of iter_ the scale pass depends on) or Levenberg-Marquardt.  LM costs ONE evaluation per trial where the
//     reference makes two: the sums of the evaluation at a pose that is then accepted are that pose's linearisation
//     (computeResiduals is a pure function of the pose), so they are kept and damped again for the next trial.
// No product path besides this file knows about them; the fused kernel and the tracking chain are Gauss-Newton only.
#include <cstdlib>
#include <new>

#include "svo_internal.h"
#include "svo_ordered_sum.h"

using namespace svo_dev;

namespace {

constexpr int RED = SVO_HIP_REDUCE_DOUBLES;
constexpr int TILE = 64;
constexpr int MAX_CHUNKS = 64;                // block rows per frame in the solver's partial buffer (svo_sia.hip)
constexpr int N_TRIALS_MAX = 5;             // n_trials_max_ (I/nlls_solver.h:107)
constexpr int BURST = 6;                    // evaluations launched between two looks at the frames' state
constexpr int SEQ_PATCHES = 128;            // patches per LDS stage of a sequential f32 sum (8 KiB)

// state of one frame beside FrameState
struct NllsExt {
  double mu, nu;                 // mu_ (0.1 at every level, S/sparse_img_align.cpp:74), nu_ (2 after reset() and after a success)
  double curH[21], curJ[6];      // LM: sums of the evaluation at the accepted pose (the linearisation of every trial from it)
  unsigned long long cur_n;
  unsigned long long iter_ref;   // iter_ exactly as the reference's loops leave it (decides the next level's scale estimate)
  int lm_iter;                   // LM: iter_ of the running level
  int n_trials;
  int phase;                     // LM: 0 = the next sums are the level's first evaluation (:109)
  float scale;                   // scale_
};

// WeightFunction::value (S/robust_cost.cpp:94-160): TDist dof 5, Tukey b 8.6851, Huber k 1.345; all f32
SVO_DEV float robust_weight(int kind, float x) {
  switch (kind) {
    case SVO_HIP_SIA_WEIGHT_TDIST: return (5.0f + 1.0f) / (5.0f + (x * x));
    case SVO_HIP_SIA_WEIGHT_TUKEY: {
      const float b_square = 8.6851f * 8.6851f;
      const float x_square = x * x;
      if (x_square <= b_square) { const float tmp = 1.0f - x_square / b_square; return tmp * tmp; }
      return 0.0f;
    }
    case SVO_HIP_SIA_WEIGHT_HUBER: {
      const float t_abs = fabsf(x);
      if (t_abs < 1.345f) return 1.0f;
      return 1.345f / t_abs;
    }
    default: return 1.0f;
  }
}

__global__ void nlls_begin_kernel(NllsExt* __restrict__ ext, int n_slots) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_slots) return;
  NllsExt& e = ext[b];
  e.mu = 0.01f; e.nu = 2.0;                 // mu_init_ (a float literal there), nu_init_: reset() (:299-309)
  for (int k = 0; k < 21; ++k) e.curH[k] = 0.0;
  for (int k = 0; k < 6; ++k) e.curJ[k] = 0.0;
  e.cur_n = 0; e.iter_ref = 0; e.lm_iter = 0; e.n_trials = 0; e.phase = 0;
  e.scale = 0.0f;
}

// after svo_hip_sia_level_begin: mu_ = 0.1 (S/sparse_img_align.cpp:74); a Gauss-Newton loop of zero iterations leaves iter_ at 0
__global__ void nlls_level_begin_kernel(NllsExt* __restrict__ ext, int n_slots, int n_iter, int method) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_slots) return;
  NllsExt& e = ext[b];
  e.mu = 0.1; e.phase = 0; e.n_trials = 0; e.lm_iter = 0;
  if (method == SVO_HIP_SIA_METHOD_GAUSS_NEWTON && n_iter <= 0) e.iter_ref = 0;
}

// computeResiduals (S/sparse_img_align.cpp:184-286) with use_weights_, one lane per patch, a wave per tile of 64.
// grid = (chunks, n_slots), block = 256: wave w of block c owns tiles (4c + w) * tpw ... + tpw - 1, whatever the frame's
// size and whoever else is in the launch, so a frame's sums -- its block rows added in block order, rows past its last
// tile being zeros -- do not depend on the batch it is solved in (tpw is 1 for solvers of up to 16384 features).
// Output: a row of RED doubles per block (21 H, 6 Jres, chi2, n_meas) in the layout of the unweighted kernel's.
// SCALE_PASS: the call with compute_weight_scale (:28-29 / :105-106): no sums, but |res| of every pixel
// (errors.push_back(fabsf(res)), :256-257) and whether the patch was inside the image.
template <bool SCALE_PASS>
__global__ __launch_bounds__(256) void nlls_residual_robust_kernel(
    const FrameConst* __restrict__ fc, const FrameState* __restrict__ st, const NllsExt* __restrict__ ext,
    const uint8_t* __restrict__ cur_level, size_t pyr_bytes, int cols, int rows, int level, int max_n, int chunks, int tpw,
    const float4* __restrict__ ref_cache, const float4* __restrict__ dxc, const float4* __restrict__ dyc,
    const double4* __restrict__ xyz4, const uint8_t* __restrict__ flags, int weight_kind, double* __restrict__ partial,
    float* __restrict__ errs, float* __restrict__ terms, uint8_t* __restrict__ err_ok) {
  const int b = blockIdx.y;
  const int chunk = blockIdx.x;
  if (st[b].level_done) return;
  const FrameConst& c = fc[b];
  const Cam cam = c.cam;
  double T[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) T[k] = st[b].model[k];
  const float wscale = ext[b].scale;
  const int n = c.n_feat;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int border = 3;
  const float scale = 1.0f / (1 << level);
  const int stride = cols;
  const uint8_t* img = cur_level + (size_t)b * pyr_bytes;
  const double jscale = fabs(cam.fx) / (1 << level);

  double accH[21], accJ[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 21; ++e) accH[e] = 0.0;
  double acc_chi = 0.0;
  unsigned acc_n = 0;

  for (int t = 0; t < tpw; ++t) {
    const int i = ((chunk * 4 + wave) * tpw + t) * TILE + lane;
    if (i >= n) continue;
    const size_t fo = (size_t)b * max_n + i;
    const uint8_t fl = flags[fo];
    bool ok = false;
    float w_tl = 0, w_tr = 0, w_bl = 0, w_br = 0;
    int off = 0;
    const double4 X = xyz4[fo];
    if (fl & F_VISIBLE) {                                   // (:212-213)
      const double xyz_ref[3] = {X.x, X.y, X.z};
      double xyz_cur[3], pxd[2];
      se3_act(T, xyz_ref, xyz_cur);
      world2cam(cam, xyz_cur, pxd);
      const float u_cur = (float)pxd[0] * scale;
      const float v_cur = (float)pxd[1] * scale;
      const int u_cur_i = (int)floorf(u_cur);
      const int v_cur_i = (int)floorf(v_cur);
      // (a NaN projection compares false everywhere in the reference and would read out of bounds there; outside here)
      ok = (u_cur_i >= 0 && v_cur_i >= 0 && u_cur_i - border >= 0 && v_cur_i - border >= 0 && u_cur_i < cols - border &&
            v_cur_i < rows - border) && u_cur == u_cur && v_cur == v_cur;
      const float subpix_u = u_cur - u_cur_i;
      const float subpix_v = v_cur - v_cur_i;
      w_tl = (float)((1.0 - subpix_u) * (1.0 - subpix_v));
      w_tr = (float)(subpix_u * (1.0 - subpix_v));
      w_bl = (float)((1.0 - subpix_u) * subpix_v);
      w_br = subpix_u * subpix_v;
      off = ok ? (v_cur_i - 2) * stride + (u_cur_i - 2) : 0;
    }
    err_ok[fo] = ok ? 1 : 0;
    if (!ok) continue;
    const uint8_t* p0 = img + off;
    float chi_p = 0.0f;
    double sdx = 0.0, sdy = 0.0, mxx = 0.0, mxy = 0.0, myy = 0.0;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const float4 rc4 = ref_cache[fo * 4 + y];
      const float4 gx4 = SCALE_PASS ? rc4 : dxc[fo * 4 + y];
      const float4 gy4 = SCALE_PASS ? rc4 : dyc[fo * 4 + y];
      const float rcv[4] = {rc4.x, rc4.y, rc4.z, rc4.w};
      const float gxv[4] = {gx4.x, gx4.y, gx4.z, gx4.w};
      const float gyv[4] = {gy4.x, gy4.y, gy4.z, gy4.w};
      const uint8_t* p = p0 + y * stride;
      float r0[5], r1[5];
#pragma unroll
      for (int x = 0; x < 5; ++x) { r0[x] = (float)p[x]; r1[x] = (float)p[stride + x]; }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float inten = w_tl * r0[x] + w_tr * r0[x + 1] + w_bl * r1[x] + w_br * r1[x + 1];       // (:246)
        const float res = inten - rcv[x];
        if (SCALE_PASS) {
          errs[fo * 16 + y * 4 + x] = fabsf(res);
        } else {
          const float weight = robust_weight(weight_kind, res / wscale);                              // (:260-263)
          terms[fo * 16 + y * 4 + x] = res * res * weight;                                            // chi2 += ... (:266), added up in order by the control step
          chi_p += res * res * weight;
          const double dres = (double)res, dw = (double)weight;
          const double ddx = (double)gxv[x], ddy = (double)gyv[x];
          sdx += (ddx * dres) * dw;                                                                   // Jres_ -= J*res*weight (:273)
          sdy += (ddy * dres) * dw;
          mxx += (ddx * ddx) * dw; mxy += (ddx * ddy) * dw; myy += (ddy * ddy) * dw;                  // H_ += J*J^T*weight (:272)
        }
      }
    }
    if (SCALE_PASS) continue;
    acc_chi += (double)chi_p;
    acc_n += 16;
    if (fl & F_JVALID) {                    // the patch has a non-zero Jacobian block at this level
      double A[6], B[6], hp[21];
      patch_jacobian_rows(X.x, X.y, X.w, jscale, A, B);
      patch_hessian(A, B, mxx, mxy, myy, hp);
#pragma unroll
      for (int e = 0; e < 21; ++e) accH[e] += hp[e];
#pragma unroll
      for (int k = 0; k < 6; ++k) accJ[k] -= A[k] * sdx + B[k] * sdy;
    }
  }
  if (SCALE_PASS) return;

  // ---- the wave's lane sums, then the 4 waves through LDS (fixed order)
  double mine = 0.0;
#pragma unroll
  for (int e = 0; e < 21; ++e) {
    const double t = group_sum<64>(accH[e]);
    if (lane == e) mine = t;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double t = group_sum<64>(accJ[k]);
    if (lane == 21 + k) mine = t;
  }
  {
    const double t = group_sum<64>(acc_chi);
    if (lane == 27) mine = t;
    const int tn = group_sum<64>((int)acc_n);
    if (lane == 28) mine = (double)tn;
  }
  __shared__ double red[4][32];
  if (lane < 32) red[wave][lane] = lane < 29 ? mine : 0.0;
  __syncthreads();
  if (threadIdx.x < 32)
    partial[((size_t)b * chunks + chunk) * RED + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- a sequential recurrence over the errors of a frame, in the order of the reference's vector (patches in list
// order, those that were inside the image only; 16 pixels each).  The whole block stages SEQ_PATCHES patches in LDS,
// thread 0 runs the recurrence over them.  Every thread returns the final value.  (The plain f32 sums -- chi2, TDist's
// lambda, Normal's variance -- do not need one lane: svo_ordered_sum.h.)
struct SeqIntAdd {                               // std::accumulate(begin, end, 0): the int seed truncates every partial sum (:80)
  int acc = 0;
  SVO_DEV void step(float v) { acc = (int)((float)acc + v); }
  SVO_DEV float value() const { return __int_as_float(acc); }   // (bit cast: the caller wants the int)
};

template <typename Op>
SVO_DEV float block_sequential(const float* __restrict__ vals, const uint8_t* __restrict__ ok, int n, float* s_stage, uint8_t* s_ok,
                               float* s_out) {
  Op op;
  for (int base = 0; base < n; base += SEQ_PATCHES) {
    const int m = min(SEQ_PATCHES, n - base);
    for (int k = threadIdx.x; k < m * 16; k += blockDim.x) s_stage[k] = vals[(size_t)base * 16 + k];
    for (int k = threadIdx.x; k < m; k += blockDim.x) s_ok[k] = ok[base + k];
    __syncthreads();
    if (threadIdx.x == 0) {
      // the next patch's 16 values are read from LDS while the chain of 16 dependent operations of this one runs
      const float4* q = reinterpret_cast<const float4*>(s_stage);
      float4 n0 = q[0], n1 = q[1], n2 = q[2], n3 = q[3];
      for (int i = 0; i < m; ++i) {
        const float4 v0 = n0, v1 = n1, v2 = n2, v3 = n3;
        const int nx = i + 1 < m ? i + 1 : i;
        n0 = q[4 * nx]; n1 = q[4 * nx + 1]; n2 = q[4 * nx + 2]; n3 = q[4 * nx + 3];
        if (!s_ok[i]) continue;
        op.step(v0.x); op.step(v0.y); op.step(v0.z); op.step(v0.w);
        op.step(v1.x); op.step(v1.y); op.step(v1.z); op.step(v1.w);
        op.step(v2.x); op.step(v2.y); op.step(v2.z); op.step(v2.w);
        op.step(v3.x); op.step(v3.y); op.step(v3.z); op.step(v3.w);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *s_out = op.value();
  __syncthreads();
  const float r = *s_out;
  __syncthreads();
  return r;
}

// The end of the weight-scale call: n_meas_ has grown by its pixels (n_meas_++ at :267 runs in that call too, and nobody
// clears n_meas_ before it), and scale_ = scale_estimator_->compute(errors) for the frames whose iter_ is 0 (:281-283).
// grid = n_slots, block = OS_THREADS.  errs / terms: [slot][max_n][16] f32.
__global__ __launch_bounds__(OS_THREADS) void nlls_scale_kernel(const FrameConst* __restrict__ fc, FrameState* __restrict__ st,
                                                         NllsExt* __restrict__ ext, const float* __restrict__ errs_all,
                                                         const uint8_t* __restrict__ ok_all, float* __restrict__ terms_all, int max_n,
                                                         int scale_kind) {
  const int b = blockIdx.x;
  if (st[b].level_done) return;                                    // block-uniform
  const int n = fc[b].n_feat;
  const float* errs = errs_all + (size_t)b * max_n * 16;
  const uint8_t* ok = ok_all + (size_t)b * max_n;
  float* terms = terms_all + (size_t)b * max_n * 16;
  __shared__ __attribute__((aligned(16))) float s_stage[SEQ_PATCHES * 16];
  __shared__ uint8_t s_ok[SEQ_PATCHES];
  __shared__ float s_out;
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_sel[2];
  __shared__ int s_count;
  __shared__ OsShared s_os;
  const int tid = threadIdx.x;
  if (tid == 0) s_count = 0;
  __syncthreads();
  {
    int mine = 0;
    for (int i = tid; i < n; i += blockDim.x) mine += ok[i] ? 1 : 0;
    if (mine) atomicAdd(&s_count, mine);
  }
  __syncthreads();
  const int n_err = 16 * s_count;                                  // errors.size()
  if (tid == 0) st[b].n_meas += (unsigned long long)n_err;
  if (ext[b].iter_ref != 0) return;                                // block-uniform: if(compute_weight_scale && iter_ == 0)
  float result;
  if (scale_kind == SVO_HIP_SIA_SCALE_TDIST) {
    // TDistributionScaleEstimator::compute (S/robust_cost.cpp:38-66): dof 5, initial sigma 5; every error is finite
    float lambda = 1.0f / (5.0f * 5.0f);
    for (int round = 0; round < 100000; ++round) {                 // (the reference's loop has no bound; it settles in a handful)
      const float initial_lamda = lambda;
      for (int k = tid; k < n * 16; k += blockDim.x) {
        if (!ok[k >> 4]) continue;
        const float error2 = errs[k] * errs[k];
        terms[k] = error2 * ((5.0f + 1.0f) / (5.0f + initial_lamda * error2));
      }
      __syncthreads();
      const float sum = os_block_sum(terms, ok, n * 16, s_os);      // lambda += ..., one f32, in order (svo_ordered_sum.h)
      lambda = (float)n_err / sum;
      if (!((double)fabsf(lambda - initial_lamda) > 1e-3)) break;  // block-uniform: every thread holds the same values
    }
    result = sqrtf(1.0f / lambda);
  } else if (scale_kind == SVO_HIP_SIA_SCALE_MAD) {
    // MADScaleEstimator::compute (:70-75): 1.48 * the element nth_element leaves at size/2 (I/math_utils.h:124-131), i.e.
    // the value of rank size/2.  Non-negative floats order like their bit patterns: a radix select, 8 bits per pass.
    if (n_err == 0) {
      result = __int_as_float(0x7fc00000);                         // (the reference dereferences end() of an empty vector)
    } else {
      unsigned prefix = 0, rank = (unsigned)(n_err / 2);
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        for (int k = tid; k < n * 16; k += blockDim.x) {
          if (!ok[k >> 4]) continue;
          const unsigned bits = (unsigned)__float_as_int(errs[k]);
          if (pass == 0 || (bits >> (shift + 8)) == prefix) atomicAdd(&s_hist[(bits >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
          unsigned cum = 0, bin = 0;
          for (; bin < 255; ++bin) {
            if (cum + s_hist[bin] > rank) break;
            cum += s_hist[bin];
          }
          s_sel[0] = (prefix << 8) | bin;
          s_sel[1] = rank - cum;
        }
        __syncthreads();
        prefix = s_sel[0]; rank = s_sel[1];
        __syncthreads();
      }
      result = 1.48f * __int_as_float((int)prefix);
    }
  } else if (scale_kind == SVO_HIP_SIA_SCALE_NORMAL) {
    // NormalDistributionScaleEstimator::compute (:77-86): the int-seeded accumulate, an integer mean, sqrt of the SUM of squares
    if (n_err == 0) {
      result = __int_as_float(0x7fc00000);                         // (an integer division by zero in the reference)
    } else {
      const int acc = __float_as_int(block_sequential<SeqIntAdd>(errs, ok, n, s_stage, s_ok, &s_out));
      const float mean = (float)((unsigned long long)acc / (unsigned long long)n_err);
      for (int k = tid; k < n * 16; k += blockDim.x) {
        if (!ok[k >> 4]) continue;
        terms[k] = (errs[k] - mean) * (errs[k] - mean);
      }
      __syncthreads();
      result = sqrtf(os_block_sum(terms, ok, n * 16, s_os));
    }
  } else {
    result = 1.0f;                                                 // UnitScaleEstimator
  }
  if (tid == 0) ext[b].scale = result;
}

// x^3 rounded once: the product formed exactly in two doubles (what a correctly rounded pow(x, 3) returns)
SVO_DEV double cube_rn(double x) {
  const double p = x * x, pe = __builtin_fma(x, x, -p);
  const double r = p * x, re = __builtin_fma(p, x, -r);
  return r + (re + pe * x);
}

SVO_DEV double norm_max6(const double* x) {
  double m = -1;
#pragma unroll
  for (int i = 0; i < 6; ++i) { const double a = fabs(x[i]); if (a > m) m = a; }
  return m;
}

// Gauss-Newton with a robust cost: gn_control_step, plus iter_ as the reference's loop leaves it (:35-99: a rollback and
// the |x| <= eps exit break at the running index, the natural end leaves n_iter_)
SVO_DEV void gn_ext_control_step(FrameState& s, NllsExt& e, const double* r, int level, int n_iter, double eps, int early_stop) {
  const int it = s.iter;
  gn_control_step(s, r, level, n_iter, eps, early_stop);
  if (!s.level_done) return;
  if (s.iter == it) { e.iter_ref = (unsigned long long)it; return; }          // rolled back / stop_
  const bool eps_exit = early_stop && norm_max6(s.x) <= eps;
  e.iter_ref = (unsigned long long)(eps_exit ? it : n_iter);
}

// optimizeLevenbergMarquardt (I/nlls_solver_impl.hpp:102-227), one step of its state machine per evaluation.
// r: the sums of the evaluation at s.model -- the level's first (e.phase 0) or that of a trial pose.  s.old_model is the
// accepted pose, s.model the one evaluated next.
SVO_DEV void lm_control_step(FrameState& s, NllsExt& e, const double* r, int level, int n_iter, double eps) {
  const double chi2_sum = r[27];
  const unsigned long long n_eval = (unsigned long long)(r[28] + 0.5);
  s.n_res += n_eval / 16;
  s.iters[level] += 1;
  bool accepted_sums = false;
  if (e.phase == 0) {
    // chi2_ = computeResiduals(model, true, false) (:109).  n_meas_ is not cleared first: it still holds what the level
    // before left, plus the weight-scale pass of this level (:105-106, added by nlls_scale_kernel), so this chi2_ is the
    // f32 sum divided by that total.
    const unsigned long long n_acc = s.n_meas + n_eval;
    s.chi2 = (double)((float)chi2_sum / (float)n_acc);
    s.n_meas = n_acc;
    e.phase = 1; e.lm_iter = 0; e.n_trials = 0;
    accepted_sums = true;
    if (n_iter <= 0) { e.iter_ref = 0; s.level_done = 1; }
  } else {
    // new_chi2 = computeResiduals(new_model, false, false); rho_ = chi2_ - new_chi2 (:163-165)
    const double new_chi2 = (double)((float)chi2_sum / (float)n_eval);
    s.n_meas = n_eval;
    const double rho = s.chi2 - new_chi2;
    if (rho > 0) {                                                 // (:177-195)
#pragma unroll
      for (int i = 0; i < 7; ++i) s.old_model[i] = s.model[i];
      s.chi2 = new_chi2;
      s.stop = norm_max6(s.x) <= eps ? 1 : 0;
      const double c = 1. - cube_rn(2 * rho - 1);
      const double m = c < 2. / 3. ? c : 2. / 3.;
      e.mu *= (1. / 3. > m ? 1. / 3. : m);
      e.nu = 2.;
      accepted_sums = true;
      if (s.stop) {
        e.iter_ref = (unsigned long long)e.lm_iter; s.level_done = 1;          // if (stop_) break (:222-223)
      } else {
        e.lm_iter += 1;
        e.n_trials = 0;
        if (e.lm_iter >= n_iter) { e.iter_ref = (unsigned long long)n_iter; s.level_done = 1; }
      }
    } else {                                                       // (:196-214)
      e.mu *= e.nu;
      e.nu *= 2.;
      e.n_trials += 1;
      if (e.n_trials >= N_TRIALS_MAX) s.stop = 1;
      if (s.stop) { e.iter_ref = (unsigned long long)e.lm_iter; s.level_done = 1; }
    }
  }
  if (accepted_sums) {
#pragma unroll
    for (int k = 0; k < 21; ++k) e.curH[k] = r[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) e.curJ[k] = r[21 + k];
    e.cur_n = n_eval;
  }
  // the next trial from the accepted pose (:141-161); a singular system is a failed trial without an evaluation (:167-174)
  while (!s.level_done) {
    double H[36], Jres[6], x[6];
    {
      int k = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) { H[i * 6 + j] = e.curH[k]; H[j * 6 + i] = e.curH[k]; ++k; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) { H[7 * i] += H[7 * i] * e.mu; Jres[i] = e.curJ[i]; }   // H_ += (H_.diagonal()*mu_).asDiagonal() (:150)
#pragma unroll
    for (int i = 0; i < 36; ++i) s.H[i] = H[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) s.Jres[i] = Jres[i];
    ldlt6_solve_reg(H, Jres, x);
#pragma unroll
    for (int i = 0; i < 6; ++i) s.x[i] = x[i];
    if (x[0] == x[0]) {
      double mx[6], dT[7], cur[7], nm[7];
#pragma unroll
      for (int i = 0; i < 6; ++i) mx[i] = -x[i];
#pragma unroll
      for (int i = 0; i < 7; ++i) cur[i] = s.old_model[i];
      se3_exp(mx, dT);
      se3_mul(cur, dT, nm);                                        // T_new = T_old * exp(-x) (S/sparse_img_align.cpp:307)
#pragma unroll
      for (int i = 0; i < 7; ++i) s.model[i] = nm[i];
      return;                                                      // ... evaluated by the next launch
    }
    s.n_meas = e.cur_n;                                            // (n_meas_ of the linearisation)
    e.mu *= e.nu;
    e.nu *= 2.;
    e.n_trials += 1;
    if (e.n_trials >= N_TRIALS_MAX) s.stop = 1;
    if (s.stop) { e.iter_ref = (unsigned long long)e.lm_iter; s.level_done = 1; }
  }
  // the level is over: the solver's pose is the accepted one
#pragma unroll
  for (int i = 0; i < 7; ++i) s.model[i] = s.old_model[i];
}

// One workgroup per frame: lanes 0..28 add the frame's block rows in block order; the block adds the frame's
// res*res*weight into one f32 in the reference's order (float chi2 ... chi2 += res*res*weight, :207,266); thread 0 takes the
// control step.
__global__ __launch_bounds__(OS_THREADS) void nlls_solve_kernel(const FrameConst* __restrict__ fc, FrameState* __restrict__ st, NllsExt* __restrict__ ext,
                                                         const double* __restrict__ partial, const float* __restrict__ terms_all,
                                                         const uint8_t* __restrict__ ok_all, int max_n, int chunks, int n_slots, int level,
                                                         int n_iter, double eps, int early_stop, int method) {
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  if (b >= n_slots || st[b].level_done) return;                    // block-uniform
  __shared__ double r[RED];
  __shared__ OsShared s_os;
  if (tid < RED) {
    double v = 0.0;
    const double* p = partial + (size_t)b * chunks * RED + tid;
    for (int c = 0; c < chunks; ++c) v += p[(size_t)c * RED];
    r[tid] = v;
  }
  const float chi2 = os_block_sum(terms_all + (size_t)b * max_n * 16, ok_all + (size_t)b * max_n, fc[b].n_feat * 16, s_os);
  __syncthreads();
  if (tid != 0) return;
  r[27] = (double)chi2;                                            // (the control steps divide (float)r[27] by (float)n_meas_, :285)
  if (method == SVO_HIP_SIA_METHOD_LEVENBERG_MARQUARDT) lm_control_step(st[b], ext[b], r, level, n_iter, eps);
  else gn_ext_control_step(st[b], ext[b], r, level, n_iter, eps, early_stop);
}

__global__ void nlls_pending_kernel(const FrameState* __restrict__ st, int n_slots, int* __restrict__ pending) {
  __shared__ int s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  int mine = 0;
  for (int b = threadIdx.x; b < n_slots; b += blockDim.x) mine += st[b].level_done ? 0 : 1;
  if (mine) atomicAdd(&s_n, mine);
  __syncthreads();
  if (threadIdx.x == 0) *pending = s_n;
}

}  // namespace

struct svo_nlls_ext {
  svo_hip_ctx* ctx = nullptr;
  NllsExt* ext = nullptr;
  float* errs = nullptr;
  float* terms = nullptr;
  uint8_t* err_ok = nullptr;
  int* pending_host = nullptr;     // page-locked, device-mapped
  int* pending_dev = nullptr;
  int cap_slots = 0;
  size_t cap_patches = 0;
  int ran_slots = 0;               // frames of the last completed run (what svo_hip_sia_solver_state may read)
};

void svo_nlls_free(svo_nlls_ext* e) {
  if (!e) return;
  if (e->ext) (void)hipFree(e->ext);
  if (e->errs) (void)hipFree(e->errs);
  if (e->terms) (void)hipFree(e->terms);
  if (e->err_ok) (void)hipFree(e->err_ok);
  if (e->pending_host) (void)hipHostFree(e->pending_host);
  delete e;
}

namespace {

int nlls_reserve(svo_hip_ctx* ctx, svo_nlls_ext** slot, int n_slots, int max_n) {
  if (!*slot) {
    *slot = new (std::nothrow) svo_nlls_ext();
    if (!*slot) return SVO_HIP_ERR_NOMEM;
    (*slot)->ctx = ctx;
  }
  if (!(*slot)->pending_dev) {                               // (both pointers or neither: a failure half way leaves neither)
    void* h = nullptr;
    SVO_CHECK_HIP(ctx, hipHostMalloc(&h, sizeof(int), hipHostMallocMapped));
    void* d = nullptr;
    const hipError_t err = hipHostGetDevicePointer(&d, h, 0);
    if (err != hipSuccess || !d) {
      (void)hipHostFree(h);
      SVO_CHECK_HIP(ctx, err != hipSuccess ? err : hipErrorInvalidValue);
    }
    ++ctx->n_allocs;
    (*slot)->pending_host = (int*)h;
    (*slot)->pending_dev = (int*)d;
  }
  svo_nlls_ext* e = *slot;
  if (e->cap_slots < n_slots) {
    SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (e->ext) { (void)hipFree(e->ext); ++ctx->n_frees; }
    e->ext = nullptr; e->cap_slots = 0;
    void* p = nullptr;
    SVO_CHECK_HIP(ctx, hipMalloc(&p, sizeof(NllsExt) * (size_t)n_slots));
    ++ctx->n_allocs;
    e->ext = (NllsExt*)p;
    e->cap_slots = n_slots;
  }
  const size_t patches = (size_t)n_slots * max_n;
  if (e->cap_patches < patches) {
    SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (e->errs) { (void)hipFree(e->errs); ++ctx->n_frees; }
    if (e->terms) { (void)hipFree(e->terms); ++ctx->n_frees; }
    if (e->err_ok) { (void)hipFree(e->err_ok); ++ctx->n_frees; }
    e->errs = e->terms = nullptr; e->err_ok = nullptr; e->cap_patches = 0;
    void* p = nullptr;
    SVO_CHECK_HIP(ctx, hipMalloc(&p, patches * 16 * sizeof(float)));
    ++ctx->n_allocs;
    e->errs = (float*)p;
    SVO_CHECK_HIP(ctx, hipMalloc(&p, patches * 16 * sizeof(float)));
    ++ctx->n_allocs;
    e->terms = (float*)p;
    SVO_CHECK_HIP(ctx, hipMalloc(&p, patches));
    ++ctx->n_allocs;
    e->err_ok = (uint8_t*)p;
    e->cap_patches = patches;
  }
  return SVO_HIP_OK;
}

template <bool SCALE_PASS>
int launch_robust(const svo_sia_view& v, svo_nlls_ext* e, int weight_kind, int chunks, int tpw) {
  hipLaunchKernelGGL(nlls_residual_robust_kernel<SCALE_PASS>, dim3(chunks, v.n_slots), dim3(256), 0, v.ctx->stream, v.fc, v.st, e->ext,
                     v.cur_level, v.pyr_bytes, v.cols, v.rows, v.level, v.max_n, chunks, tpw, v.ref_cache, v.dxc, v.dyc, v.xyz4, v.flags,
                     weight_kind, v.partial, e->errs, e->terms, e->err_ok);
  SVO_CHECK_HIP(v.ctx, hipGetLastError());
  return SVO_HIP_OK;
}

}  // namespace

// SparseImgAlign::run (S/sparse_img_align.cpp:51-92) with method_ / the robust cost as set on the solver.
int svo_nlls_run(svo_hip_sia* s, int n_slots, const svo_hip_sia_params* prm, int method, int scale_estimator, int weight_function) {
  int rc = svo_hip_sia_begin(s, n_slots, prm);            // validates; geometry and the solver's reset()
  if (rc != SVO_HIP_OK) return rc;
  const bool weights = scale_estimator != SVO_HIP_SIA_SCALE_UNIT;
  const bool lm = method == SVO_HIP_SIA_METHOD_LEVENBERG_MARQUARDT;
  svo_nlls_ext** slot = svo_sia_nlls_slot(s);
  if (*slot) (*slot)->ran_slots = 0;
  svo_sia_view v;
  svo_hip_ctx* ctx = nullptr;
  int chunks = 1, tpw = 1;
  for (int level = prm->max_level; level >= prm->min_level; --level) {
    if ((rc = svo_hip_sia_level_begin(s, level)) != SVO_HIP_OK) return rc;
    if ((rc = svo_sia_view_get(s, &v)) != SVO_HIP_OK) return rc;
    ctx = v.ctx;
    {
      // the block <-> tile map of nlls_residual_robust_kernel: a function of the solver's capacity alone
      const int tiles = (v.max_n + TILE - 1) / TILE;
      chunks = (tiles + 3) / 4 < MAX_CHUNKS ? (tiles + 3) / 4 : MAX_CHUNKS;
      tpw = (tiles + 4 * chunks - 1) / (4 * chunks);
    }
    if (level == prm->max_level) {
      if ((rc = nlls_reserve(ctx, slot, n_slots, v.max_n)) != SVO_HIP_OK) return rc;
      hipLaunchKernelGGL(nlls_begin_kernel, dim3((n_slots + 63) / 64), dim3(64), 0, ctx->stream, (*slot)->ext, n_slots);
      SVO_CHECK_HIP(ctx, hipGetLastError());
    }
    svo_nlls_ext* e = *slot;
    hipLaunchKernelGGL(nlls_level_begin_kernel, dim3((n_slots + 63) / 64), dim3(64), 0, ctx->stream, e->ext, n_slots, prm->n_iter, method);
    SVO_CHECK_HIP(ctx, hipGetLastError());
    if (weights) {                                          // computeResiduals(model, false, true) (:28-29 / :105-106)
      if ((rc = launch_robust<true>(v, e, weight_function, chunks, tpw)) != SVO_HIP_OK) return rc;
      hipLaunchKernelGGL(nlls_scale_kernel, dim3(n_slots), dim3(OS_THREADS), 0, ctx->stream, v.fc, v.st, e->ext, e->errs, e->err_ok, e->terms,
                         v.max_n, scale_estimator);
      SVO_CHECK_HIP(ctx, hipGetLastError());
    }
    auto evaluate_and_step = [&]() -> int {
      // (without a robust cost the weight function is Unit: the same kernel, every weight 1)
      int r2 = launch_robust<false>(v, e, weights ? weight_function : SVO_HIP_SIA_WEIGHT_UNIT, chunks, tpw);
      if (r2 != SVO_HIP_OK) return r2;
      hipLaunchKernelGGL(nlls_solve_kernel, dim3(n_slots), dim3(OS_THREADS), 0, ctx->stream, v.fc, v.st, e->ext, v.partial, e->terms, e->err_ok, v.max_n,
                         chunks, n_slots, level, prm->n_iter, prm->eps, prm->early_stop, method);
      SVO_CHECK_HIP(ctx, hipGetLastError());
      return SVO_HIP_OK;
    };
    // Gauss-Newton: at most n_iter evaluations.  Levenberg-Marquardt: one for chi2_, then one per trial, at most
    // N_TRIALS_MAX per outer iteration.  How many a frame needs is data dependent (a handful, normally): they are launched
    // in rounds of BURST and the host looks at the number of unfinished frames between rounds -- cheaper than launching the
    // worst case (launches of finished frames return at once but still cost a few microseconds each).
    const long budget = lm ? 1 + (long)N_TRIALS_MAX * prm->n_iter : (long)prm->n_iter;
    long done = 0;
    while (done < budget) {
      const long m = BURST < budget - done ? BURST : budget - done;
      for (long k = 0; k < m; ++k)
        if ((rc = evaluate_and_step()) != SVO_HIP_OK) return rc;
      done += m;
      if (done >= budget) break;
      hipLaunchKernelGGL(nlls_pending_kernel, dim3(1), dim3(256), 0, ctx->stream, v.st, n_slots, e->pending_dev);
      SVO_CHECK_HIP(ctx, hipGetLastError());
      SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (*(volatile int*)e->pending_host == 0) break;
    }
  }
  rc = svo_hip_sia_finish(s);
  if (*slot) (*slot)->ran_slots = rc == SVO_HIP_OK ? n_slots : 0;
  return rc;
}

int svo_nlls_scale(svo_hip_sia* s, int slot, float* scale, double* mu, double* nu) {
  svo_nlls_ext** p = svo_sia_nlls_slot(s);
  if (!p || !*p || slot < 0 || slot >= (*p)->ran_slots) return SVO_HIP_ERR_STATE;
  NllsExt h;
  int rc = svo_hip_memcpy_d2h((*p)->ctx, &h, (*p)->ext + slot, sizeof(h));      // (waits for the context's stream)
  if (rc != SVO_HIP_OK) return rc;
  if (scale) *scale = h.scale;
  if (mu) *mu = h.mu;
  if (nu) *nu = h.nu;
  return SVO_HIP_OK;
}

extern "C" int svo_hip_sia_solver_state(svo_hip_sia* sia, int slot, float* scale, double* mu, double* nu) {
  if (!sia) return SVO_HIP_ERR_INVALID;
  return svo_nlls_scale(sia, slot, scale, mu, nu);
}

namespace {
__global__ __launch_bounds__(OS_THREADS) void ordered_sum_kernel(const float* __restrict__ vals, int n, float* __restrict__ out) {
  __shared__ OsShared s_os;
  const float r = os_block_sum(vals, nullptr, n, s_os);
  if (threadIdx.x == 0) *out = r;
}
}  // namespace

extern "C" int svo_hip_ordered_sum_f32_dev(svo_hip_ctx* ctx, const float* vals_dev, size_t n, float* out_dev) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, (vals_dev || !n) && out_dev && n <= (size_t)INT_MAX);
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(OS_THREADS), 0, ctx->stream, vals_dev, (int)n, out_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}
