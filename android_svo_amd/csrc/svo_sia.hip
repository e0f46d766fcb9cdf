// svo_sia.hip -- batched sparse image alignment (svo::SparseImgAlign) on gfx950.
//
// Replaces, per frame pair, SparseImgAlign::run (S/sparse_img_align.cpp:51-92) with
// precomputeReferencePatches (:105-178), computeResiduals (:184-286), solve/update
// (:291-308) driven by NLLSSolver::optimizeGaussNewton (I/nlls_solver_impl.hpp:25-100).
//
// Two implementations of svo_hip_sia_run with the same semantics:
//   (F) the fused kernel (second half of this file, the default): one workgroup per frame pair runs the whole
//       coarse-to-fine solve in one launch with the interpolated reference patches in LDS / L2-resident memory;
//   (S) the streaming kernels described next: one launch per Gauss-Newton evaluation; used by the step-wise entry
//       points (multi-GPU all-reduce of the normal equations) and for frames with more than 2816 features.
//
// Design of (S) (MI355X-first, not the reference's loop structure):
//   * B independent frame pairs are solved at once; the whole coarse-to-fine loop is a
//     fixed sequence of kernels on one stream with the Gauss-Newton control state
//     (model, rollback copy, chi2, stop, iteration) resident in HBM -- no host round trip;
//     data-dependent early exits become per-frame "done" flags that later kernels test.
//   * Work unit = a TILE of 64 patches per wavefront.  A lane plays two roles:
//       lane-per-patch  (lane 4g+r owns patch 16r+g of the tile): fp64 projection of the
//                       patch, bilinear weights, and the per-patch part of the normal
//                       equations -- 64 distinct patches per instruction, nothing redundant;
//       lane-per-pixel-row (4 sub-passes s=0..3; the quad 4g..4g+3 works on patch 16s+g,
//                       lane r on pixel row r): two unaligned 8-byte image row reads, three
//                       16-byte cache reads (1 KiB contiguous per wave instruction), 4 pixels
//                       of f32 image math per lane.
//     The owner of patch 16s+g is lane s of the same quad, so every exchange between the two
//     roles is a DPP quad_perm (no LDS, no ds_bpermute).
//   * Inverse-compositional structure: the per-pixel Jacobian is J = dx*A + dy*B with A,B the
//     rows of the 2x6 projection Jacobian of the patch times fx/2^L, hence
//       sum_px J J^T = sxx AA^T + sxy (AB^T+BA^T) + syy BB^T,   sum_px J r = A sum(dx r) + B sum(dy r).
//     We keep 3 f32 per pixel (ref, dx, dy), {x,y,z,1/z} and {sxx,sxy,syy} per patch instead of the
//     reference's 768 B/patch fp64 Jacobian cache.  The 21 distinct entries of each tile's
//     sum_patches H_patch are stored once per level; every evaluation adds that row (lane e adds
//     entry e) and takes out, one at a time, the linearised patches that are outside the current
//     image at this evaluation -- H is therefore the sum over the patches visible now, as
//     computeResiduals builds it.
//   * fp64 for H/Jres and their partial sums; residual/chi2 in f32 per pixel row, widened per
//     patch.  Reductions are in fixed order (bitwise reproducible run to run, no atomics).
#include <cstdlib>
#include <cstring>
#include <vector>

#include "svo_internal.h"

using namespace svo_dev;

namespace {

constexpr int PATCH_AREA = 16;
constexpr int RED = SVO_HIP_REDUCE_DOUBLES;     // 32
constexpr int MAX_CHUNKS = 64;
constexpr int TILE = 64;                        // patches per wavefront pass
constexpr int TILE_ROW = 24;                    // doubles per tile-H row (21 used)
// (F_VISIBLE, F_JVALID: svo_internal.h)
constexpr uint8_t F_GONE = 4;
constexpr uint8_t F_HASPOINT = 8;              // fused kernel: the feature has a map point (sticky)                   // fused kernel: outside the current image at the previous evaluation

// (FrameConst, FrameState: svo_internal.h -- the tracking chain of svo_track.hip writes the one and reads the other on the device)
struct LevelGeom {
  int cols, rows;
  size_t ref_off, cur_off;   // byte offset of the level inside a pyramid
};

struct Shard { int rank, world; };

// upper-triangle (row-major) index -> (i, j)
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
__device__ __constant__ int8_t kTriI[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
// the same tables for compile-time indices (unrolled loops)
constexpr int kTriIc[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
constexpr int kTriJc[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
__device__ __constant__ int8_t kTriJ[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};

SVO_DEV void shard_range(int n, Shard sh, int* lo, int* hi) {
  *lo = (int)(((long long)n * sh.rank) / sh.world);
  *hi = (int)(((long long)n * (sh.rank + 1)) / sh.world);
}

// byte k (0..7) of an 8-byte little-endian row segment, as float
SVO_DEV float byte_f(uint2 w, int k) {
  const unsigned v = k < 4 ? w.x : w.y;
  return (float)((v >> (8 * (k & 3))) & 0xffu);
}

SVO_DEV uint2 load_row8(const uint8_t* p) {
  uint2 w;
  __builtin_memcpy(&w, p, 8);      // unaligned global_load_dwordx2
  return w;
}

// (patch_hessian, patch_jacobian_rows: svo_device_math.h)

__global__ void sia_begin_kernel(const FrameConst* __restrict__ fc, FrameState* __restrict__ st, int n_slots) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_slots) return;
  FrameState& s = st[b];
  double Tinv[7], T[7];
  se3_inverse(fc[b].T_ref_w, Tinv);
  se3_mul(fc[b].T_cur_w_init, Tinv, T);            // sparse_img_align.cpp:69
  for (int i = 0; i < 7; ++i) { s.model[i] = T[i]; s.old_model[i] = T[i]; s.T_cur_w[i] = fc[b].T_cur_w_init[i]; }
  s.chi2 = 1e10;                                   // reset(), nlls_solver_impl.hpp:299-309
  for (int i = 0; i < 36; ++i) s.H[i] = 0.0;
  for (int i = 0; i < 6; ++i) { s.Jres[i] = 0.0; s.x[i] = 0.0; }
  s.n_meas = 0; s.n_pre = 0; s.n_res = 0;
  s.stop = 0; s.iter = 0; s.level_done = 0;
  s.empty = fc[b].n_feat <= 0;
  for (int i = 0; i < SVO_HIP_MAX_LEVELS; ++i) s.iters[i] = 0;
}

// xyz_ref = f * |pos - ref_pos| and 1/z for every feature: level independent
// (sparse_img_align.cpp:131-135,220-221).  One thread per patch.
__global__ __launch_bounds__(256) void sia_geometry_kernel(
    const FrameConst* __restrict__ fc, int max_n, const double* __restrict__ f, const double* __restrict__ pos,
    double4* __restrict__ xyz4) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const FrameConst& c = fc[b];
  if (i >= c.n_feat) return;
  const size_t fi = (size_t)b * max_n + i;
  const double dxp = pos[3 * fi] - c.ref_pos[0];
  const double dyp = pos[3 * fi + 1] - c.ref_pos[1];
  const double dzp = pos[3 * fi + 2] - c.ref_pos[2];
  const double depth = sqrt(dxp * dxp + dyp * dyp + dzp * dzp);
  double4 o;
  o.x = f[3 * fi] * depth; o.y = f[3 * fi + 1] * depth; o.z = f[3 * fi + 2] * depth;
  o.w = 1. / o.z;
  xyz4[fi] = o;
}

// One level's precomputeReferencePatches.  grid = (ceil(max_n/256), n_slots), block = 256:
// wave w of block x owns tile 4x+w of the frame's shard.
// Algorithmic bytes per patch: 49 B footprint + 64 B feature in, 64 B + 768 B caches out;
// physical: 16 x 8 B row reads + 17 B feature/flag in, 192 B pixel caches + 32 B sums (+ 3 B tile row) out.
__global__ __launch_bounds__(256) void sia_precompute_kernel(
    const FrameConst* __restrict__ fc, FrameState* __restrict__ st, const uint8_t* __restrict__ ref_base,
    size_t pyr_bytes, LevelGeom g, int level, int max_n, int max_tiles, Shard sh,
    const double* __restrict__ px, const uint8_t* __restrict__ has_point, const double4* __restrict__ xyz4,
    float4* __restrict__ ref_cache, float4* __restrict__ dxc, float4* __restrict__ dyc,
    double4* __restrict__ sxyz, double* __restrict__ tile_h, uint8_t* __restrict__ flags,
    unsigned int* __restrict__ n_pre_count) {
  const int b = blockIdx.y;
  const FrameConst& c = fc[b];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // per-level solver reset: optimizeGaussNewton saves the model for rollback (:33) and
    // restarts iter_; stop_ and chi2_ deliberately persist
    FrameState& s = st[b];
    for (int k = 0; k < 7; ++k) s.old_model[k] = s.model[k];
    s.iter = 0;
    s.level_done = s.empty;
  }
  int lo, hi;
  shard_range(c.n_feat, sh, &lo, &hi);
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> SGPRs
  const int tile_base = lo + tile * TILE;
  if (tile_base >= hi) return;                       // wave-uniform
  const int q = lane >> 2, r = lane & 3;
  const int stride = g.cols;
  const uint8_t* img = ref_base + (size_t)b * pyr_bytes + g.ref_off;

  // ---- lane-per-patch: visibility in the reference image and bilinear weights (:121-142)
  const int i_own = tile_base + 16 * r + q;
  const size_t fo = (size_t)b * max_n + i_own;
  bool valid = false;
  float w_tl = 0, w_tr = 0, w_bl = 0, w_br = 0;
  int off = 0;
  if (i_own < hi) {
    const int border = 3;
    const float scale = 1.0f / (1 << level);
    const float u_ref = (float)(px[2 * fo] * scale);
    const float v_ref = (float)(px[2 * fo + 1] * scale);
    const int u_ref_i = (int)floorf(u_ref);
    const int v_ref_i = (int)floorf(v_ref);
    valid = has_point[fo] && !(u_ref_i - border < 0 || v_ref_i - border < 0 || u_ref_i >= g.cols - border ||
                               v_ref_i >= g.rows - border);
    const float subpix_u = u_ref - u_ref_i;
    const float subpix_v = v_ref - v_ref_i;
    w_tl = (float)((1.0 - subpix_u) * (1.0 - subpix_v));
    w_tr = (float)(subpix_u * (1.0 - subpix_v));
    w_bl = (float)((1.0 - subpix_u) * subpix_v);
    w_br = subpix_u * subpix_v;
    off = (v_ref_i - 3) * stride + (u_ref_i - 3);    // top-left of the 8-byte-wide, 7-row footprint
    const uint8_t fl = flags[fo];
    // visible_fts_ is only ever set (:128); the Jacobian block is zero unless recomputed now (:76)
    flags[fo] = valid ? (uint8_t)(F_VISIBLE | F_JVALID) : (uint8_t)(fl & F_VISIBLE);
  }
  {
    const unsigned long long m = __ballot(valid);
    if (lane == 0 && m) atomicAdd(&n_pre_count[b], (unsigned)__popcll(m));
  }

  // ---- lane-per-pixel-row: quad q works on patch 16s+q in sub-pass s, lane r on row r
  double sxx = 0.0, sxy = 0.0, syy = 0.0;
#define SVO_PRE_SUBPASS(S)                                                                              \
  {                                                                                                     \
    const bool v_s = quad_bcast<S>((int)valid) != 0;                                                    \
    if (v_s) {                                                                                          \
      const float a_tl = quad_bcast<S>(w_tl), a_tr = quad_bcast<S>(w_tr);                               \
      const float a_bl = quad_bcast<S>(w_bl), a_br = quad_bcast<S>(w_br);                               \
      const uint8_t* p = img + quad_bcast<S>(off) + r * stride;                                         \
      const uint2 Rm = load_row8(p), R0 = load_row8(p + stride), R1 = load_row8(p + 2 * stride),        \
                  R2 = load_row8(p + 3 * stride);                                                       \
      float val[4], dx[4], dy[4];                                                                       \
      double pxx = 0.0, pxy = 0.0, pyy = 0.0;                                                           \
      _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                                   \
        const float m_1 = byte_f(Rm, x + 1), m_2 = byte_f(Rm, x + 2);                                   \
        const float c_0 = byte_f(R0, x), c_1 = byte_f(R0, x + 1), c_2 = byte_f(R0, x + 2), c_3 = byte_f(R0, x + 3); \
        const float d_0 = byte_f(R1, x), d_1 = byte_f(R1, x + 1), d_2 = byte_f(R1, x + 2), d_3 = byte_f(R1, x + 3); \
        const float e_1 = byte_f(R2, x + 1), e_2 = byte_f(R2, x + 2);                                   \
        val[x] = a_tl * c_1 + a_tr * c_2 + a_bl * d_1 + a_br * d_2;                                     \
        dx[x] = 0.5f * ((a_tl * c_2 + a_tr * c_3 + a_bl * d_2 + a_br * d_3) -                          \
                        (a_tl * c_0 + a_tr * c_1 + a_bl * d_0 + a_br * d_1));                          \
        dy[x] = 0.5f * ((a_tl * d_1 + a_tr * d_2 + a_bl * e_1 + a_br * e_2) -                          \
                        (a_tl * m_1 + a_tr * m_2 + a_bl * c_1 + a_br * c_2));                          \
        const double ddx = (double)dx[x], ddy = (double)dy[x];                                          \
        pxx += ddx * ddx; pxy += ddx * ddy; pyy += ddy * ddy;                                           \
      }                                                                                                 \
      const size_t o4 = ((size_t)b * max_n + tile_base + 16 * S + q) * 4 + r;                          \
      ref_cache[o4] = make_float4(val[0], val[1], val[2], val[3]);                                      \
      dxc[o4] = make_float4(dx[0], dx[1], dx[2], dx[3]);                                                \
      dyc[o4] = make_float4(dy[0], dy[1], dy[2], dy[3]);                                                \
      pxx = quad_sum(pxx); pxy = quad_sum(pxy); pyy = quad_sum(pyy);                                    \
      if (r == S) { sxx = pxx; sxy = pxy; syy = pyy; }                                                  \
    }                                                                                                   \
  }
  SVO_PRE_SUBPASS(0) SVO_PRE_SUBPASS(1) SVO_PRE_SUBPASS(2) SVO_PRE_SUBPASS(3)
#undef SVO_PRE_SUBPASS

  // ---- lane-per-patch: per-patch sums and the tile's Hessian row
  double hp[21];
#pragma unroll
  for (int e = 0; e < 21; ++e) hp[e] = 0.0;
  if (valid) {
    sxyz[fo] = make_double4(sxx, sxy, syy, 0.0);
    const double4 X = xyz4[fo];
    double A[6], B[6];
    patch_jacobian_rows(X.x, X.y, X.w, fabs(c.cam.fx) / (1 << level), A, B);   // errorMultiplier2()/2^L (:113,172)
    patch_hessian(A, B, sxx, sxy, syy, hp);
  }
  double mine = 0.0;
#pragma unroll
  for (int e = 0; e < 21; ++e) {
    const double t = group_sum<64>(hp[e]);
    if (lane == e) mine = t;
  }
  if (lane < TILE_ROW) tile_h[((size_t)b * max_tiles + tile) * TILE_ROW + lane] = lane < 21 ? mine : 0.0;
}

// (defined with the fused kernel below)
template <bool EXACT_ROWS>
SVO_DEV double fused_tile_row_body(double x, double y, double z_inv, double jscale, double sxx, double sxy, double syy, bool contributes,
                                   int lane);

// (gn_control_step: svo_internal.h)

// The control step at the head of an evaluation launch (see sia_residual_kernel), called by the first wave of the block
// and kept out of line: the 6x6 solve must not shape the register allocation of the streaming loop (inlined it took the
// kernel from 127 to 165 VGPRs and the state copy into 640 B of scratch per lane).  The state is stepped in LDS.
__device__ __noinline__ void head_control_step(const FrameState* in, FrameState* out, const double* partial_rows, int chunks, int level, int n_iter,
                                               double eps, int early_stop, double* s_model, int* s_level_done) {
  __shared__ FrameState s_state;
  __shared__ double s_r[RED];
  const int lane = threadIdx.x;
  {
    // 64 lanes copy the record and the sums in (8-byte words)
    static_assert(sizeof(FrameState) % 8 == 0, "FrameState is copied as 8-byte words");
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(in);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&s_state);
    for (int i = lane; i < (int)(sizeof(FrameState) / 8); i += 64) dst[i] = src[i];
    if (lane < RED) {                 // the frame's sums: its block partials in block order (as sia_solve_kernel<true> adds them)
      double v = 0.0;
      for (int c = 0; c < chunks; ++c) v += partial_rows[(size_t)c * RED + lane];
      s_r[lane] = v;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane == 0) {
    if (!s_state.level_done) gn_control_step(s_state, s_r, level, n_iter, eps, early_stop);
    for (int k = 0; k < 7; ++k) s_model[k] = s_state.model[k];
    *s_level_done = s_state.level_done;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (out) {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&s_state);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(out);
    for (int i = lane; i < (int)(sizeof(FrameState) / 8); i += 64) dst[i] = src[i];
  }
}

// what the FUSED_STEP instance of sia_residual_kernel does at its head (pending == null: nothing)
struct SiaStepFusion {
  const double* pending;         // block partials of the previous evaluation, all-reduced over the ranks: take its control step
  FrameState* st_out;            // ... and write the stepped state here (block 0 of the frame)
  int n_iter, early_stop;
  double eps;
};

// One computeResiduals(linearize=true) evaluation for every live frame.
// grid = (chunks, n_slots), block = 256; wave w of chunk c walks tiles c*tpc + w, +4, ...
// Output: one partial row of RED doubles per block.
// FUSED_STEP: the instance svo_hip_sia_run_sharded launches (control step of the previous evaluation at the head); the
// plain instance -- svo_hip_sia_run's streaming path, the HBM-bound Jacobian pass, and the step-wise entry points --
// carries none of that code (the out-of-line head costs the kernel a wave of occupancy).
template <bool FUSED_STEP>
__global__ __launch_bounds__(256) void sia_residual_kernel(
    const FrameConst* __restrict__ fc, const FrameState* __restrict__ st, const uint8_t* __restrict__ cur_base,
    size_t pyr_bytes, LevelGeom g, int level, int max_n, int max_tiles, int chunks, Shard sh,
    const float4* __restrict__ ref_cache, const float4* __restrict__ dxc, const float4* __restrict__ dyc,
    const double4* __restrict__ sxyz, const double4* __restrict__ xyz4, const double* __restrict__ tile_h,
    const uint8_t* __restrict__ flags, double* __restrict__ partial, SiaStepFusion fu) {
  const int b = blockIdx.y;
  const int chunk = blockIdx.x;
  double* out_row = partial + ((size_t)b * chunks + chunk) * RED;
  __shared__ double s_model[7];
  __shared__ int s_level_done;
  // Patch-sharded solve (svo_hip_sia_run_sharded): the control step of the PREVIOUS evaluation -- its sums have been
  // all-reduced into fu.pending since -- is taken at the head of this launch by every block of the frame (identical
  // inputs, identical result), and block 0 writes the stepped state to the other state buffer, which nobody reads
  // during this launch.  One launch per Gauss-Newton step instead of three.
  if (FUSED_STEP && fu.pending) {
    if (threadIdx.x < 64) head_control_step(st + b, chunk == 0 ? fu.st_out + b : nullptr, fu.pending + (size_t)b * chunks * RED, chunks, level, fu.n_iter,
                                            fu.eps, fu.early_stop, s_model, &s_level_done);
  }
  const bool stepped = FUSED_STEP && fu.pending;             // block-uniform
  if (stepped) __syncthreads();
  const int level_done = stepped ? s_level_done : st[b].level_done;
  if (level_done) {                  // this frame's GN loop already exited at this level: it contributes zeros to the exchange
    if (FUSED_STEP && threadIdx.x < RED) out_row[threadIdx.x] = 0.0;
    return;
  }
  const FrameConst& c = fc[b];
  const Cam cam = c.cam;
  double T[7];                       // block-uniform: scalar registers (a scalar load, or the stepped model out of LDS)
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const double v = stepped ? s_model[k] : st[b].model[k];
    T[k] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  }

  int lo, hi;
  shard_range(c.n_feat, sh, &lo, &hi);
  const int n_tiles = (hi - lo + TILE - 1) / TILE;
  const int tpc = (n_tiles + chunks - 1) / chunks;
  const int t_end = min(n_tiles, (chunk + 1) * tpc);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> SGPRs
  const int q = lane >> 2, r = lane & 3;
  const int border = 3;
  const float scale = 1.0f / (1 << level);
  const int stride = g.cols;
  const uint8_t* img = cur_base + (size_t)b * pyr_bytes + g.cur_off;
  const double jscale = fabs(cam.fx) / (1 << level);

  double accH = 0.0;                       // lane e < 21 accumulates H entry e
  double accJ[6] = {0, 0, 0, 0, 0, 0};     // lane-per-patch partial sums of Jres
  double acc_chi = 0.0;
  unsigned acc_n = 0;

  for (int tile = chunk * tpc + wave; tile < t_end; tile += 4) {
    const int tile_base = lo + tile * TILE;
    // every load that does not depend on the projection is issued first, unconditionally, so that
    // the HBM/MALL latencies of a tile overlap (slots past n_feat are allocated, their values unused)
    const size_t o4 = ((size_t)b * max_n + tile_base + q) * 4 + r;
    float4 rc[4], gx[4], gy[4];
#pragma unroll
    for (int S = 0; S < 4; ++S) {
      rc[S] = ref_cache[o4 + 64 * S];
      gx[S] = dxc[o4 + 64 * S];
      gy[S] = dyc[o4 + 64 * S];
    }
    // ---- lane-per-patch: project the patch into the current image (:220-236)
    const int i_own = tile_base + 16 * r + q;
    const size_t fo = (size_t)b * max_n + i_own;
    bool ok = false, jvalid = false;
    float w_tl = 0, w_tr = 0, w_bl = 0, w_br = 0;
    int off = 0;
    const double4 X = xyz4[fo];
    if (i_own < hi) {
      const uint8_t fl = flags[fo];
      jvalid = (fl & F_JVALID) != 0;
      if (fl & F_VISIBLE) {
        const double xyz_ref[3] = {X.x, X.y, X.z};
        double xyz_cur[3], pxd[2];
        se3_act(T, xyz_ref, xyz_cur);
        world2cam(cam, xyz_cur, pxd);
        const float u_cur = (float)pxd[0] * scale;
        const float v_cur = (float)pxd[1] * scale;
        const int u_cur_i = (int)floorf(u_cur);
        const int v_cur_i = (int)floorf(v_cur);
        // NaN projections compare false everywhere in the reference and would read out of bounds
        // there; here they are outside the image.
        ok = (u_cur_i >= 0 && v_cur_i >= 0 && u_cur_i - border >= 0 && v_cur_i - border >= 0 &&
              u_cur_i < g.cols - border && v_cur_i < g.rows - border) && u_cur == u_cur && v_cur == v_cur;   // (no u_cur_i + border: the conversion saturates)
        const float subpix_u = u_cur - u_cur_i;
        const float subpix_v = v_cur - v_cur_i;
        w_tl = (float)((1.0 - subpix_u) * (1.0 - subpix_v));
        w_tr = (float)(subpix_u * (1.0 - subpix_v));
        w_bl = (float)((1.0 - subpix_u) * subpix_v);
        w_br = subpix_u * subpix_v;
        off = ok ? (v_cur_i - 2) * stride + (u_cur_i - 2) : 0;
      }
    }

    // ---- lane-per-pixel-row: residuals of patch 16s+q, row r (:238-279).
    // All loads of the four sub-passes are issued up front and unconditionally (a patch that is
    // not ok reads row r of the image / a cache row of its own slot: valid memory, result unused)
    // so their latencies overlap instead of queueing behind one another.
    double sdx = 0.0, sdy = 0.0;
    float chi = 0.0f;
    {
      uint2 R0[4], R1[4];
#define SVO_ROWS(S)                                                                   \
      {                                                                               \
        const uint8_t* p = img + quad_bcast<S>(off) + r * stride;                     \
        R0[S] = load_row8(p); R1[S] = load_row8(p + stride);                          \
      }
      SVO_ROWS(0) SVO_ROWS(1) SVO_ROWS(2) SVO_ROWS(3)
#undef SVO_ROWS
#define SVO_RES_SUBPASS(S)                                                                              \
      {                                                                                                 \
        const float a_tl = quad_bcast<S>(w_tl), a_tr = quad_bcast<S>(w_tr);                             \
        const float a_bl = quad_bcast<S>(w_bl), a_br = quad_bcast<S>(w_br);                             \
        const float rcv[4] = {rc[S].x, rc[S].y, rc[S].z, rc[S].w};                                      \
        const float gxv[4] = {gx[S].x, gx[S].y, gx[S].z, gx[S].w};                                      \
        const float gyv[4] = {gy[S].x, gy[S].y, gy[S].z, gy[S].w};                                      \
        double px_ = 0.0, py_ = 0.0;                                                                    \
        float pc = 0.0f;                                                                                \
        _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                                 \
          const float inten = a_tl * byte_f(R0[S], x) + a_tr * byte_f(R0[S], x + 1) +                   \
                              a_bl * byte_f(R1[S], x) + a_br * byte_f(R1[S], x + 1);                    \
          const float res = inten - rcv[x];                                                             \
          pc += res * res;                                                                              \
          const double dres = (double)res;                                                              \
          px_ += (double)gxv[x] * dres;                                                                 \
          py_ += (double)gyv[x] * dres;                                                                 \
        }                                                                                               \
        px_ = quad_sum(px_); py_ = quad_sum(py_); pc = quad_sum(pc);                                    \
        if (r == S) { sdx = px_; sdy = py_; chi = pc; }                                                 \
      }
      SVO_RES_SUBPASS(0) SVO_RES_SUBPASS(1) SVO_RES_SUBPASS(2) SVO_RES_SUBPASS(3)
#undef SVO_RES_SUBPASS
    }

    // ---- lane-per-patch: normal equations
    const bool lin = ok && jvalid;          // the patch carries a non-zero Jacobian block
    if (ok) { acc_chi += (double)chi; acc_n += 16; }
    if (lin) {
      double A[6], B[6];
      patch_jacobian_rows(X.x, X.y, X.w, jscale, A, B);
#pragma unroll
      for (int k = 0; k < 6; ++k) accJ[k] -= A[k] * sdx + B[k] * sdy;          // Jres_ -= J*res (:273)
    }
    // H: add the tile's precomputed row (lane e adds entry e), then take out the linearised patches that are outside the
    // current image at this evaluation -- all of them at once (fused_tile_row: every lane its own point and gradient
    // sums, one wave reduction; walking them one after the other cost ~500 cycles per patch, and a coarse level can
    // have most of a tile outside): H is the sum over the patches visible now, as in the reference.
    if (lane < 21) accH += tile_h[((size_t)b * max_tiles + tile) * TILE_ROW + lane];
    const bool gone_lane = jvalid && !ok;
    if (__ballot(gone_lane) != 0ull) {                            // wave-uniform, normally not taken
      const double4 S4 = sxyz[(size_t)b * max_n + (gone_lane ? tile_base + 16 * (lane & 3) + (lane >> 2) : 0)];   // (always a valid address)
      const double out_row = fused_tile_row_body<true>(X.x, X.y, X.w, jscale, S4.x, S4.y, S4.z, gone_lane, lane);   // (inlined here: 127 VGPRs either way)
      if (lane < 21) accH -= out_row;
    }
  }

  // ---- wave reduction of the lane-per-patch sums, then the 4 waves through LDS
  double mine = accH;                       // lanes 0..20
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
    out_row[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Sum the block partials of each frame in fixed order -> reduce buffer [n_slots][RED] (the step-wise entry points: what a
// caller exchanges between svo_hip_sia_accumulate and svo_hip_sia_solve_update).
__global__ void sia_sum_partials_kernel(const FrameState* __restrict__ st, const double* __restrict__ partial,
                                        int chunks, double* __restrict__ reduce, int n_slots) {
  const int b = blockIdx.x;
  const int t = threadIdx.x;       // 32 threads
  if (b >= n_slots) return;
  if (st[b].level_done) { reduce[(size_t)b * RED + t] = 0.0; return; }
  double v = 0.0;
  const double* p = partial + (size_t)b * chunks * RED + t;
  for (int c = 0; c < chunks; ++c) v += p[(size_t)c * RED];
  reduce[(size_t)b * RED + t] = v;
}

// One Gauss-Newton control step per frame, one wave per frame: lanes 0..28 gather the frame's sums (from the block
// partials, or from the all-reduced buffer), lane 0 runs gn_control_step.  st_out != st: the state is read from st and
// the stepped state written to st_out (the end-of-level step of the sharded solve, which brings the state back to its
// home buffer); frames that are done are copied through.
template <bool FROM_PARTIALS>
__global__ __launch_bounds__(64) void sia_solve_kernel(const FrameState* __restrict__ st, FrameState* __restrict__ st_out, const double* __restrict__ src,
                                                       int chunks, int n_slots, int level, int n_iter, double eps,
                                                       int early_stop) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  if (b >= n_slots) return;
  const bool done = st[b].level_done != 0;
  if (done && st_out == st) return;
  __shared__ double r[RED];
  if (!done && lane < RED) {
    double v = 0.0;
    if (FROM_PARTIALS) {
      const double* p = src + (size_t)b * chunks * RED + lane;
      for (int c = 0; c < chunks; ++c) v += p[(size_t)c * RED];
    } else {
      v = src[(size_t)b * RED + lane];
    }
    r[lane] = v;
  }
  __syncthreads();
  if (lane != 0) return;
  if (st_out == st) { gn_control_step(st_out[b], r, level, n_iter, eps, early_stop); return; }
  FrameState loc = st[b];
  if (!done) gn_control_step(loc, r, level, n_iter, eps, early_stop);
  st_out[b] = loc;
}

__global__ void sia_finish_kernel(const FrameConst* __restrict__ fc, FrameState* __restrict__ st,
                                  const unsigned int* __restrict__ n_pre_count, int n_slots) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_slots) return;
  FrameState& s = st[b];
  double T[7];
  se3_mul(s.model, fc[b].T_ref_w, T);                         // :89
  for (int i = 0; i < 7; ++i) s.T_cur_w[i] = s.empty ? fc[b].T_cur_w_init[i] : T[i];
  s.n_pre = n_pre_count[b];
}

// =================================================================================================
// Fused path: the whole coarse-to-fine solve of one frame pair in ONE workgroup (512 threads, one CU),
// one launch per run.
//   * a wave owns tiles of 64 patches, a lane one patch of a tile; the lane walks the 16 pixels itself, so there is no
//     cross-lane traffic inside a tile;
//   * per level a lane interpolates its patch once: the 32 values W whose differences are reference value, dx and dy
//     of the 16 pixels (bit-identical to the reference's cache).  W (128 B per patch) stays in LDS for two tiles of
//     every wave and in L2 / Infinity-Cache resident memory for the others; {x,y,z,1/z}, the flags and the wave's
//     tile-H entries stay in VGPRs; {sxx,sxy,syy} go to memory once per level and are only re-read when the set of
//     patches outside the current image changes;
//   * the loop over a wave's tiles is software-pipelined by one tile (projection, image rows, W from memory);
//   * 8 waves (2 per SIMD, 256 VGPRs per lane) measured fastest: 16 waves (128 VGPRs) 77 k frames/s against 98 k at the
//     time, 12 waves 85 k, 4 waves 68 k.  Packed f32 (v_pk_mul_f32 / v_pk_add_f32) was measured too: CDNA4's SIMD
//     already issues a plain wave64 f32 op in 2 cycles, the packed forms take two passes (18 % slower);
//   * the two waves of a SIMD are balanced with s_setprio (see tile_of in the kernel);
//   * the Gauss-Newton state lives in LDS; the solve runs between two barriers: H^-1 by columns on six lanes when H
//     changed, a 6x6 matrix-vector product otherwise, SE3::exp and the pose product on one lane.
// Data-dependent exits are real `break`s here, so a converged frame costs nothing further.
// Semantics are those of the streaming kernels above (same 64-patch tiles, lane-ordered sums per wave,
// then a fixed-order sum over the waves).
// =================================================================================================
constexpr int FUSED_THREADS = 512;
constexpr int FUSED_WAVES = FUSED_THREADS / 64;
constexpr int FUSED_MAX_TILES = 44;               // 2816 patches: 6 tiles on an older wave, 5 on a younger one
constexpr int FUSED_WC_BYTES = TILE * 128;        // interpolated reference patches of one tile
constexpr int FUSED_EXTRA_TILES = 3;              // 8-wave shape: at most this many LDS tiles beyond two per wave
constexpr int FUSED_MAX_TPW = 6;                  // the older wave's share of a SIMD's 11 tiles
constexpr int FUSED_EXACT_ROW_BELOW = 16;         // frames with fewer patches: H rows entry by entry (fused_tile_row_exact)
// instances of a kernel shape: the plain one, the one whose workgroups may take the exact rows, the fast arithmetic
constexpr int FUSED_PLAIN = 0, FUSED_EXACT_ROWS = 1, FUSED_FAST = 2, FUSED_M32 = 3;

struct FusedLevels {
  int cols[SVO_HIP_MAX_LEVELS], rows[SVO_HIP_MAX_LEVELS];
  unsigned long long ref_off[SVO_HIP_MAX_LEVELS], cur_off[SVO_HIP_MAX_LEVELS];
};

struct FusedParams { int max_level, min_level, n_iter, early_stop; double eps; int moments_f32; };

// interpolated value at footprint row pair (R, Rn) and byte column k, with the reference's operation order
SVO_DEV float interp_at(uint2 R, uint2 Rn, int k, float a_tl, float a_tr, float a_bl, float a_br) {
  return a_tl * byte_f(R, k) + a_tr * byte_f(R, k + 1) + a_bl * byte_f(Rn, k) + a_br * byte_f(Rn, k + 1);
}
// the same sum contracted as a compiler with -ffp-contract=on forms it (a product, then three fused multiply-adds: 4
// operations instead of 7): SVO_HIP_SIA_ARITH_FAST only
SVO_DEV float interp_at_contracted(uint2 R, uint2 Rn, int k, float a_tl, float a_tr, float a_bl, float a_br) {
  return __builtin_fmaf(a_br, byte_f(Rn, k + 1), __builtin_fmaf(a_bl, byte_f(Rn, k), __builtin_fmaf(a_tl, byte_f(R, k), a_tr * byte_f(R, k + 1))));
}

struct LppGeom {
  bool ok;
  float w_tl, w_tr, w_bl, w_br;
  int off;
};

// what the hot loop keeps of the camera in scalar registers; the distortion coefficients stay in memory and are
// only fetched (wave-uniform scalar loads) by a distorted camera
struct LeanCam {
  double fx, fy, cx, cy;
  const double* d;      // null: pinhole without distortion
};

// x/z and y/z for the projection of the hot loop.  The compiler's expansion of an fp64 division is v_div_scale (both
// operands), v_rcp_f64, two Newton steps on the reciprocal, the quotient with one residual correction (v_div_fmas) and
// v_div_fixup for the special operands; the scaling only acts when an exponent is near the end of the range.  Here
// the refined reciprocal is formed ONCE for both quotients and the same residual correction is applied: for a
// denominator whose exponent is in the middle 1800 binades these are, operation for operation, the roundings of the
// expansion.  Checked on the device against the compiler's divisions (tools/probes/div_probe.hip,
// profiles/r02_probe_div_probe.txt, 10^9 random quotients per range): bit-identical for |z| in [2^-900, 2^900) and
// numerators in [2^-960, 2^960) whenever the quotient is a normal number.  A numerator outside that range can cost
// the last bit (the residual turns subnormal) -- of a quotient below 2^-60 or above 2^60, which projects onto the
// principal point or outside every image either way, as does a subnormal, infinite or NaN quotient.  Any other
// denominator in the wave takes the two plain divisions.
SVO_DEV void div2_by(double x, double y, double z, double& qx, double& qy) {
  const unsigned ez = (unsigned)__double2hiint(z) & 0x7ff00000u;                  // biased exponent << 20
  const bool mid = ez - (123u << 20) < (1800u << 20);                             // 2^-900 <= |z| < 2^900 (no 0, inf, NaN, subnormal)
  if (__ballot(!mid) == 0ull) {                                                   // wave-uniform
    double r = __builtin_amdgcn_rcp(z);
    double e = __builtin_fma(-z, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-z, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q0 = x * r, p0 = y * r;
    qx = __builtin_fma(__builtin_fma(-z, q0, x), r, q0);
    qy = __builtin_fma(__builtin_fma(-z, p0, y), r, p0);
  } else {
    qx = x / z;
    qy = y / z;
  }
}

// projection of one patch into the current level image (:220-236); the weights come back halved (see the caller)
SVO_DEV LppGeom lpp_project(const double* T, const LeanCam& cam, const double4& X, bool visible, float scale, int cols,
                            int rows, int stride) {
  LppGeom g;
  // (no `if (visible)` around this: a patch that is not visible carries finite coordinates, and a branch costs more
  // than the work it would skip in the rare wave without a single visible patch)
  {
    const int border = 3;
    const double xyz_ref[3] = {X.x, X.y, X.z};
    double xyz_cur[3], pxd[2];
    se3_act(T, xyz_ref, xyz_cur);
    double un, vn;
    div2_by(xyz_cur[0], xyz_cur[1], xyz_cur[2], un, vn);
    if (__builtin_expect(cam.d == nullptr, 1)) {
      pxd[0] = cam.fx * un + cam.cx;
      pxd[1] = cam.fy * vn + cam.cy;
    } else {
      Cam full;
      full.fx = cam.fx; full.fy = cam.fy; full.cx = cam.cx; full.cy = cam.cy;
      full.distortion = 1; full.width = full.height = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i) full.d[i] = cam.d[i];
      world2cam_uv(full, un, vn, pxd);
    }
    const float u_cur = (float)pxd[0] * scale;
    const float v_cur = (float)pxd[1] * scale;
    const int u_cur_i = (int)floorf(u_cur);
    const int v_cur_i = (int)floorf(v_cur);
    g.ok = visible && (u_cur_i >= 0 && v_cur_i >= 0 && u_cur_i - border >= 0 && v_cur_i - border >= 0 &&
            u_cur_i < cols - border && v_cur_i < rows - border) && u_cur == u_cur && v_cur == v_cur;       // (no u_cur_i + border: the conversion saturates)
    const float subpix_u = u_cur - u_cur_i;
    const float subpix_v = v_cur - v_cur_i;
    // (float)((1.0 - su) * (1.0 - sv)) of the reference, in f32: inside the image u_cur >= 3, so su = u_cur - floor(u_cur)
    // is a multiple of 2^-22 below 1, 1 - su is exact in f32 (<= 22 significant bits) and the product of two such
    // numbers (<= 44 bits, exact in f64) is rounded to f32 once on either path.  Outside the image the weights are not used.
    const float ou = 1.0f - subpix_u, ov = 1.0f - subpix_v;
    g.w_tl = 0.5f * (ou * ov);
    g.w_tr = 0.5f * (subpix_u * ov);
    g.w_bl = 0.5f * (ou * subpix_v);
    g.w_br = 0.5f * (subpix_u * subpix_v);
    g.off = g.ok ? (v_cur_i - 2) * stride + (u_cur_i - 2) : 0;
  }
  return g;
}

// which of a wave's TPW tiles keep their interpolated patches in LDS (CK of them) and which in memory, in the order
// they are processed: alternating, LDS first, so that every tile read from memory has a predecessor whose computation
// covers the latency of its prefetch
template <int TPW, int CK>
struct FusedPlan {
  static constexpr bool in_lds(int p) {
    int l = CK, g = TPW - CK;
    bool last_l = false, pick_l = false;
    for (int i = 0; i <= p; ++i) {
      pick_l = l > 0 && (g == 0 || !last_l);
      if (pick_l) --l; else --g;
      last_l = pick_l;
    }
    return pick_l;
  }
  static constexpr int lds_slot(int p) {       // number of LDS tiles before position p
    int c = 0;
    for (int i = 0; i < p; ++i) c += in_lds(i) ? 1 : 0;
    return c;
  }
};

// NW = 8: one workgroup (one frame pair) per CU, two waves of it on every SIMD.  NW = 4: one wave per SIMD and 256
// VGPRs, so that TWO workgroups -- two independent frame pairs -- share a CU: while one of them is in the one-lane
// phase between its barriers the other one has the SIMDs to itself (used when a launch holds at least two pairs per
// CU).
SVO_DEV int wave_of_thread() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// Cold parts of the fused kernel, kept out of line: they run a handful of times per frame pair, and inlined their
// register demand (H[36] + m[6][6] in registers, the library sine / cosine) shapes the allocation of the whole kernel.
//
// H changed: lane 0 factors it (pivoted LDL^T, I/nlls_solver_impl.hpp:35-99 -> Eigen LDLT), then lanes 0..5 each push
// one unit vector through the substitution -- six right-hand sides in the time of one -- and keep the columns of H^-1.
// Every evaluation after that is a 6x6 matrix-vector product on six lanes instead of a serial substitution with six
// divisions.  Called by a whole wave; the arrays live in LDS.
__device__ __noinline__ void fused_refactor_cold(const double* s_Hc, double* s_fac, int* s_ftr, int* s_fac_valid, double* s_inv, int lane) {
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane == 0) {
    double H[36], m[6][6];
    int tr[6];
    int kk = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) { H[i * 6 + j] = s_Hc[kk]; H[j * 6 + i] = s_Hc[kk]; ++kk; }
    ldlt6_factor_reg(H, m, tr);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      s_ftr[i] = tr[i];
#pragma unroll
      for (int j = 0; j <= i; ++j) s_fac[i * (i + 1) / 2 + j] = m[i][j];
    }
    *s_fac_valid = 1;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < 6) {
    double m[6][6], e[6], col[6];
    int tr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      tr[i] = __builtin_amdgcn_readfirstlane(s_ftr[i]);
      e[i] = lane == i ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j <= i; ++j) m[i][j] = s_fac[i * (i + 1) / 2 + j];
    }
    ldlt6_substitute_reg(m, tr, e, col);
#pragma unroll
    for (int j = 0; j < 6; ++j) s_inv[lane * 6 + j] = col[j];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// large update angles (theta^2 > 0.25): the library path of SE3::exp
__device__ __noinline__ void se3_exp_cold(const double* l, double* out) { se3_exp(l, out); }

// The Hessian row of a set of patches of one tile: lane e (< 21) returns entry e -- (i, j) of the upper triangle in
// row-major order -- of the sum, over the lanes with `contributes`, of the patch's J^T J summed over its pixels
// = sxx A A^T + sxy (A B^T + B A^T) + syy B B^T.  Used for a tile's whole row when the reference patches of a level are
// formed (contributes = valid) and for the part to take out of it when the set of patches outside the current image has
// changed (contributes = outside): both with the same arithmetic per patch and the same reduction, so that a tile whose
// patches are all outside is left with a row of exact zeros.
// Factored: entry (i, j) = A_i P_j + B_i Q_j with P = sxx A + sxy B, Q = sxy A + syy B (A_1 and B_0 are structural
// zeros), 54 fp64 operations instead of 210; a lane that does not contribute has its sums and its point replaced
// before the products.  EXACT_ROWS: entry by entry (see sia_fused_kernel).
// The 21 wave sums go through three transposing reductions (7 long-range exchanges each) instead of 21 butterflies;
// after reduction c3 the eight lanes 8j..8j+7 all hold the total of entry 8 c3 + j: lane 8j + c3 keeps it, and ONE lane
// exchange at the end (instead of one per reduction) brings entry e from lane 8 (e & 7) + (e >> 3).
template <bool EXACT_ROWS>
SVO_DEV double fused_tile_row_body(double x, double y, double z_inv, double jscale, double sxx, double sxy, double syy, bool contributes,
                                   int lane) {
  const double gxx = contributes ? sxx : 0.0, gxy = contributes ? sxy : 0.0, gyy = contributes ? syy : 0.0;
  double A[6], B[6], P[6], Q[6];
  if (EXACT_ROWS) {
    patch_jacobian_rows(x, y, z_inv, jscale, A, B);
  } else {
    patch_jacobian_rows(contributes ? x : 0.0, contributes ? y : 0.0, contributes ? z_inv : 1.0, jscale, A, B);
    P[0] = gxx * A[0]; Q[0] = gxy * A[0];
    P[1] = gxy * B[1]; Q[1] = gyy * B[1];
#pragma unroll
    for (int j = 2; j < 6; ++j) { P[j] = __builtin_fma(gxx, A[j], gxy * B[j]); Q[j] = __builtin_fma(gxy, A[j], gyy * B[j]); }
  }
  double kept = 0.0;
#pragma unroll
  for (int c3 = 0; c3 < 3; ++c3) {
    double h[8];
#pragma unroll
    for (int q8 = 0; q8 < 8; ++q8) {
      const int e = 8 * c3 + q8;
      h[q8] = 0.0;
      if (e < 21) {
        const int i = kTriIc[e], j = kTriJc[e];
        if (EXACT_ROWS) {
          const double he = sxx * (A[i] * A[j]) + sxy * (A[i] * B[j] + B[i] * A[j]) + syy * (B[i] * B[j]);
          h[q8] = contributes ? he : 0.0;             // a select, not a branch per entry
        } else {
          h[q8] = i == 0 ? A[0] * P[j] : i == 1 ? B[1] * Q[j] : __builtin_fma(A[i], P[j], B[i] * Q[j]);
        }
      }
    }
    const double r = wave_reduce8(h);                  // lanes 8j..8j+7: total of entry 8*c3 + j
    if ((lane & 7) == c3) kept = r;
  }
  return __shfl(kept, 8 * (lane & 7) + (lane < 24 ? lane >> 3 : 0), 64);
}

// The entry-by-entry form runs once per level and tile (and on the rare corrections), for the handful of frames with
// fewer than FUSED_EXACT_ROW_BELOW patches: out of line, so that its ~250 live fp64 values do not shape the register
// allocation of the evaluation loop (inlined, the <8,1,1,true> instance spilled 30 VGPRs inside that loop and ran three
// times slower per evaluation than <8,1,1,false>: 1.58 ms for ONE 5-patch frame in fixed-work mode).
__device__ __noinline__ double fused_tile_row_exact(double x, double y, double z_inv, double jscale, double sxx, double sxy, double syy,
                                                    bool contributes, int lane) {
  return fused_tile_row_body<true>(x, y, z_inv, jscale, sxx, sxy, syy, contributes, lane);
}

template <bool EXACT_ROWS>
SVO_DEV double fused_tile_row(double x, double y, double z_inv, double jscale, double sxx, double sxy, double syy, bool contributes,
                              int lane) {
  if (EXACT_ROWS) return fused_tile_row_exact(x, y, z_inv, jscale, sxx, sxy, syy, contributes, lane);
  return fused_tile_row_body<false>(x, y, z_inv, jscale, sxx, sxy, syy, contributes, lane);
}

// Reference patch of the fused kernel, shared with the kernel that dumps it for the parity test (sia_fused_patch_dump_kernel):
// the reference's (float)((1.0 - su) * (1.0 - sv)) etc. in f32 (exact for a position inside the image, see lpp_project),
// HALVED like the weights of the evaluation: every W is exactly half the reference's interpolated value -- what the
// evaluation wants to read -- and a difference of two of them is its 0.5f * (a - b) gradient.
SVO_DEV void fused_ref_weights(float u_ref, float v_ref, int u_ref_i, int v_ref_i, float* w) {
  const float su = u_ref - u_ref_i, sv = v_ref - v_ref_i;
  const float ou = 1.0f - su, ov = 1.0f - sv;
  w[0] = 0.5f * (ou * ov);
  w[1] = 0.5f * (su * ov);
  w[2] = 0.5f * (ou * sv);
  w[3] = 0.5f * (su * sv);
}

// the 32 interpolated values of a patch (6 x 6 without the corners) from its seven 8-byte footprint rows.  (The fused
// kernel keeps this two-line loop inline: called as a function there, the same code spilled 43 instead of 20 VGPRs.)
SVO_DEV void fused_interp_W(const uint2* F, float w_tl, float w_tr, float w_bl, float w_br, float (*W)[6]) {
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int c2 = 0; c2 < 6; ++c2)
      W[j][c2] = ((j == 0 || j == 5) && (c2 == 0 || c2 == 5)) ? 0.0f : interp_at(F[j], F[j + 1], c2, w_tl, w_tr, w_bl, w_br);
}

// EXACT_ROWS: workgroups whose frame has fewer than FUSED_EXACT_ROW_BELOW patches form the per-tile Hessian rows entry
// by entry as sxx (A_i A_j) + sxy (A_i B_j + B_i A_j) + syy (B_i B_j) -- the form the kernel used for every frame until
// round 2 -- instead of factored (see fused_tile_row_body).  There H can be rank-deficient (one patch: rank 2), the
// reference's own solve then wanders off on rounding noise -- in the reference-derived test of a one-patch frame until
// the patch leaves the image and the pose turns NaN -- and the exact form happens to follow it
// (tests/test_gpu_parity.py::test_batch_ragged_and_empty) where the factored one returned a finite pose.  The launcher
// picks the EXACT_ROWS instance of a shape for a batch that holds such a frame; every other frame of that batch still
// takes the factored rows, so its result does not depend on the company it is launched in.
template <int NW, int TPW, int CK, int VARIANT>
__global__ __launch_bounds__(NW * 64, 2) void sia_fused_kernel(
    const FrameConst* __restrict__ fc, FrameState* __restrict__ st, const uint8_t* __restrict__ ref_base,
    const uint8_t* __restrict__ cur_base, size_t pyr_bytes, FusedLevels lv, int max_n, const double* __restrict__ px,
    const double* __restrict__ f, const double* __restrict__ pos, const uint8_t* __restrict__ has_point,
    double4* __restrict__ sxyz, double* __restrict__ tile_h, float4* __restrict__ wmem, int max_tiles, FusedParams prm,
    int tiles_young, int n_extra) {
  using Plan = FusedPlan<TPW, CK>;
  // the LDS left over holds one more tile -- the first one the plan keeps in memory -- for the first n_extra waves
  // (kernels whose waves own five or six tiles are register-bound: for them the extra LDS slot and the deferred
  // outside-the-image correction below cost more in spills than they save -- measured at 2500 patches: 2.02 ms
  // without either, 2.36 ms with the slot, 2.98 ms with both)
  constexpr bool LEAN = TPW >= 5;
  constexpr int P_EXTRA = (!LEAN && CK > 0 && TPW > CK) ? 1 : -1;   // plan order is L G L G ...: position 1
  const bool extra_lds = wave_of_thread() < n_extra;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // The interpolated reference patch of every feature -- the 32 values W from which reference value, dx and dy of its
  // 16 pixels are differences (128 B, halved, see below) -- is formed once per level.  CK tiles of every wave keep
  // theirs in LDS ([slot][8][64] float4, slot = position * 8 + wave), the others in memory ([frame][tile][8][64]
  // float4, L2 / Infinity-Cache resident: 8 coalesced 1-KiB loads per tile and evaluation, issued one tile ahead).
  float4* wc = reinterpret_cast<float4*>(smem);
  __shared__ double red[NW][29];                              // 21 H + 6 Jres + chi2 + n_meas per wave
  __shared__ double s_th[NW * TPW * 21];                      // per-tile H rows (lane e keeps entry e)
  __shared__ double s_x[8];
  __shared__ double s_last[TPW >= 5 ? 29 : 1];               // register-bound shapes only (see LEAN below)
  __shared__ double s_Hc[21], s_fac[21], s_inv[36];   // H of the previous evaluation, its LDL^T factor (lower triangle), H^-1 by columns
  __shared__ int s_ftr[6], s_fac_valid;
  __shared__ double s_model[8], s_old[8];
  __shared__ double s_chi2;
  __shared__ int s_done, s_stop, s_iter;
  __shared__ unsigned s_npre;
  __shared__ double s_nres, s_nmeas;             // exact integer counts, carried as doubles
  __shared__ int s_iters[SVO_HIP_MAX_LEVELS];
#ifdef SVO_STAMPS
  const long long k_m0 = __builtin_amdgcn_s_memtime(), k_r0 = __builtin_amdgcn_s_memrealtime();   // core clock / constant 100 MHz
  __shared__ long long s_stamp[10];
  __shared__ long long s_lev[SVO_HIP_MAX_LEVELS];      // whole evaluations (evaluation + barriers + solve) per level, wave 1
  if (threadIdx.x < SVO_HIP_MAX_LEVELS) s_lev[threadIdx.x] = 0;
  __shared__ unsigned s_cnt[4];        // re-factorisations of H, tile rows corrected, patches outside the image at those corrections
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
  __shared__ long long s_wst[16];
  if (threadIdx.x < 16) s_wst[threadIdx.x] = 0;
  if (threadIdx.x == 64) { s_stamp[0] = s_stamp[1] = s_stamp[2] = 0; s_stamp[9] = 0; }
  if (threadIdx.x == 0) { for (int i = 3; i < 9; ++i) s_stamp[i] = 0; }
#endif

  const int b = blockIdx.x;
  const FrameConst& c = fc[b];
  LeanCam cam;
  cam.fx = c.cam.fx; cam.fy = c.cam.fy; cam.cx = c.cam.cx; cam.cy = c.cam.cy;
  cam.d = c.cam.distortion ? c.cam.d : nullptr;
  const int n = c.n_feat;
  // block-uniform: a frame with a handful of patches takes the entry-by-entry Hessian rows (out of line: fused_tile_row_exact).
  // Only the instances launched for a batch that holds such a frame carry the branch (EXACT_ROWS): it costs the evaluation
  // loop registers -- 36 instead of 20 spilled VGPRs in <8,4,2>, -1.4 % fixed work, -3 % with the reference's exits.
  constexpr bool EXACT_ROWS = VARIANT == FUSED_EXACT_ROWS;
  constexpr bool FAST = VARIANT == FUSED_FAST;
  // SVO_HIP_SIA_ARITH_MOMENTS_F32: the reference's residuals and chi2, the two gradient moments of a patch summed in f32.
  // The EXACT_ROWS instance (a batch that holds a frame of a handful of patches) takes the level from a kernel argument, so
  // that a frame's result does not depend on the company it is launched in.
  const bool M32 = VARIANT == FUSED_M32 || (VARIANT == FUSED_EXACT_ROWS && prm.moments_f32 != 0);
  const bool exact_rows = EXACT_ROWS && n < FUSED_EXACT_ROW_BELOW;
  const int n_tiles = (n + TILE - 1) / TILE;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> SGPRs
  const bool empty = n <= 0;
  const int tri_i = kTriI[lane < 21 ? lane : 0], tri_j = kTriJ[lane < 21 ? lane : 0];
  // Tile ownership.  Waves w and w+4 share SIMD w&3, and the SIMD's arbiter favours the older one (w < 4): it runs
  // at the pace of a wave that is alone on its SIMD (one instruction per ~5.4 cycles, whatever the type: tools/probes/valu_rate_probe.hip), the younger
  // one gets the issue cycles that are left (measured: 57 % of that pace) and finishes an equal share 40 % later,
  // alone on a half-used SIMD.  The tiles of a SIMD (s, s+4, s+8, ...) are split evenly, the older wave takes the
  // first TPW of them and the younger one the tiles_young that follow; during the first 5/8 of its tiles (to half a tile) the younger
  // wave raises its priority (s_setprio), so that each wave is the favoured one for about half of the evaluation and both
  // reach the barrier together (measured per-wave cycles per evaluation: 15.0 k / 21.5 k before, 18.2 k / 19.7 k after).
  const int simd = wave & 3;
  const int my_tiles = (NW == 4 || wave < 4) ? TPW : tiles_young;       // wave-uniform
  const int first_j = (NW == 4 || wave < 4) ? 0 : TPW;
  // favoured for 5/8 of its tiles, to half a tile: 2.5 of 4 measured best (2 of 4: -1.2 %, 3 of 4: -1.4 %)
  const int favoured_tiles = (5 * my_tiles) / 8;                        // whole tiles at raised priority
  const int favoured_half = (5 * my_tiles) % 8 >= 4 ? favoured_tiles : -1;   // then the first half of this one
  auto tile_of = [&](int k) -> int { return k < my_tiles ? simd + 4 * (first_j + k) : n_tiles; };
  // per-tile H row of this wave (lane e keeps entry e): in LDS, read once per tile and evaluation
  auto th_get = [&](int k) -> double { return lane < 21 ? s_th[(wave * TPW + k) * 21 + lane] : 0.0; };
  auto th_set = [&](int k, double v) { if (lane < 21) s_th[(wave * TPW + k) * 21 + lane] = v; };

  if (threadIdx.x == 0) {
    double Tinv[7], T[7];
    se3_inverse(c.T_ref_w, Tinv);
    se3_mul(c.T_cur_w_init, Tinv, T);                        // sparse_img_align.cpp:69
    for (int i = 0; i < 7; ++i) { s_model[i] = T[i]; s_old[i] = T[i]; }
    s_chi2 = 1e10;                                           // reset(), nlls_solver_impl.hpp:299-309
    s_stop = 0; s_done = 0; s_iter = 0; s_npre = 0; s_nres = 0; s_nmeas = 0; s_fac_valid = 0;
    for (int i = 0; i < SVO_HIP_MAX_LEVELS; ++i) s_iters[i] = 0;
    for (int i = 0; i < 8; ++i) s_x[i] = 0.0;
  }

  double v_last = 0.0;                                       // wave 0, lanes 0..28: sums of the last evaluation
  if (LEAN && threadIdx.x < 29) s_last[threadIdx.x] = 0.0;
  // ---- lane-per-patch persistent state of the wave's tiles
  double4 X[TPW];
  uint8_t fl[TPW];
  float pxf[TPW][2];
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    const int tile = tile_of(k);
    const int i_own = tile * TILE + lane;
    X[k] = make_double4(0, 0, 1, 1);
    th_set(k, 0.0);
    fl[k] = 0;
    pxf[k][0] = pxf[k][1] = 0.0f;
    if (tile < n_tiles && i_own < n) {
      const size_t fi = (size_t)b * max_n + i_own;
      if (has_point[fi]) fl[k] = F_HASPOINT;
      pxf[k][0] = (float)px[2 * fi]; pxf[k][1] = (float)px[2 * fi + 1];
      const double dxp = pos[3 * fi] - c.ref_pos[0];
      const double dyp = pos[3 * fi + 1] - c.ref_pos[1];
      const double dzp = pos[3 * fi + 2] - c.ref_pos[2];
      const double depth = sqrt(dxp * dxp + dyp * dyp + dzp * dzp);
      X[k].x = f[3 * fi] * depth; X[k].y = f[3 * fi + 1] * depth; X[k].z = f[3 * fi + 2] * depth;
      X[k].w = 1. / X[k].z;
    }
  }
  __syncthreads();

  for (int level = prm.max_level; level >= prm.min_level; --level) {
    const int cols = lv.cols[level], rows = lv.rows[level], stride = cols;
    const uint8_t* ref_img = ref_base + (size_t)b * pyr_bytes + lv.ref_off[level];
    const uint8_t* cur_img = cur_base + (size_t)b * pyr_bytes + lv.cur_off[level];
    const float scale = 1.0f / (1 << level);
    const double jscale = fabs(cam.fx) / (1 << level);
    const int border = 3;
    if (threadIdx.x == 0) {
      for (int i = 0; i < 7; ++i) s_old[i] = s_model[i];     // rollback copy (:33)
      s_iter = 0;
      s_done = empty ? 1 : 0;
    }

    // ================= precomputeReferencePatches for the wave's tiles =================
#ifdef SVO_STAMPS
    const long long tp0 = __builtin_amdgcn_s_memtime();
#endif
    // Pass 1: where every patch of the wave's tiles sits in the reference level (the feature loads of all tiles are
    // independent of each other and go out together).  Pass 2: footprint rows one tile ahead of the arithmetic.
    int pre_off[TPW];
    float pre_w[TPW][4];
    bool pre_valid[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int tile = tile_of(k);
      const int i_own = tile * TILE + lane;
      const bool have = tile < n_tiles && i_own < n;
      // (float)(px * 2^-L) == (float)(px) * 2^-L: the level-0 position in f32 is kept, no feature loads per level
      const float u_ref = pxf[k][0] * scale;
      const float v_ref = pxf[k][1] * scale;
      const int u_ref_i = (int)floorf(u_ref);
      const int v_ref_i = (int)floorf(v_ref);
      const bool valid = have && (fl[k] & F_HASPOINT) != 0 && !(u_ref_i - border < 0 || v_ref_i - border < 0 || u_ref_i >= cols - border ||
                                                    v_ref_i >= rows - border);
      fused_ref_weights(u_ref, v_ref, u_ref_i, v_ref_i, pre_w[k]);
      pre_off[k] = valid ? (v_ref_i - 3) * stride + (u_ref_i - 3) : 0;   // offset 0: always valid memory
      pre_valid[k] = valid;
      // visible_fts_ is only ever set (:128); the Jacobian block is zero unless recomputed now (:76)
      if (have) fl[k] = valid ? (uint8_t)(F_VISIBLE | F_JVALID | F_HASPOINT) : (uint8_t)(fl[k] & (F_VISIBLE | F_HASPOINT));
    }
    uint2 Fq[2][7];
#pragma unroll
    for (int j = 0; j < 7; ++j) Fq[0][j] = load_row8(ref_img + pre_off[0] + j * stride);
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      // the younger wave of a SIMD is favoured for the first 5/8 of its tiles here as in the evaluation (see tile_of):
      // without it the older waves sit at the barrier below while the younger ones crawl
      if (NW == 8 && wave >= 4) { if (8 * k < 5 * my_tiles) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
      // (opaque: the store addresses derived from the tile number are formed here, per level, instead of being
      // hoisted out of the level loop into registers that then spill)
      int tile = tile_of(k);
      asm volatile("" : "+s"(tile));
      if (k + 1 < TPW) {
#pragma unroll
        for (int j = 0; j < 7; ++j) Fq[(k + 1) & 1][j] = load_row8(ref_img + pre_off[k + 1] + j * stride);
      }
      if (tile >= n_tiles) continue;                         // wave-uniform
      const int i_own = tile * TILE + lane;
      const bool valid = pre_valid[k];
      const float w_tl = pre_w[k][0], w_tr = pre_w[k][1], w_bl = pre_w[k][2], w_br = pre_w[k][3];
#ifdef SVO_STAMPS
      const long long tq0 = __builtin_amdgcn_s_memtime();
#endif
      {
        const unsigned long long m = __ballot(valid);
        if (lane == 0 && m) atomicAdd(&s_npre, (unsigned)__popcll(m));
      }
      double sxx = 0.0, sxy = 0.0, syy = 0.0;
      {
        // lane-per-patch: the lane interpolates its patch's 7 footprint rows and sums the 16 pixels.  A patch
        // that is not valid at this level keeps the values of the level before (the reference's stale cache row).
        float W[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int c2 = 0; c2 < 6; ++c2)
            W[j][c2] = ((j == 0 || j == 5) && (c2 == 0 || c2 == 5)) ? 0.0f : interp_at(Fq[k & 1][j], Fq[k & 1][j + 1], c2, w_tl, w_tr, w_bl, w_br);
        if (valid) {
          // in the order the evaluation reads them: rows 0 and 5 without their corners
          float4* dst = Plan::in_lds(k) ? wc + (size_t)((Plan::lds_slot(k) * NW + wave) * 8) * TILE + lane
                        : (k == P_EXTRA && extra_lds) ? wc + (size_t)((CK * NW + wave) * 8) * TILE + lane
                                                      : wmem + ((size_t)b * max_tiles + tile) * 8 * TILE + lane;
          float q[32];
          int e = 0;
#pragma unroll
          for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int c2 = 0; c2 < 6; ++c2)
              if (!((j == 0 || j == 5) && (c2 == 0 || c2 == 5))) q[e++] = W[j][c2];
#pragma unroll
          for (int c4 = 0; c4 < 8; ++c4) dst[c4 * TILE] = make_float4(q[4 * c4], q[4 * c4 + 1], q[4 * c4 + 2], q[4 * c4 + 3]);
        }
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float dxv = W[y + 1][x + 2] - W[y + 1][x];                       // = 0.5f * (a - b) of the full values
            const float dyv = W[y + 2][x + 1] - W[y][x + 1];
            const double ddx = (double)dxv, ddy = (double)dyv;
            // (the kernel's own sums: fused)
            sxx = __builtin_fma(ddx, ddx, sxx); sxy = __builtin_fma(ddx, ddy, sxy); syy = __builtin_fma(ddy, ddy, syy);
          }
      }
#ifdef SVO_STAMPS
      const long long tq1 = __builtin_amdgcn_s_memtime();
#endif
      if (valid) sxyz[(size_t)b * max_n + i_own] = make_double4(sxx, sxy, syy, 0.0);   // only re-read when a patch leaves the image
      // the tile's Hessian row: lane e keeps entry e (fused_tile_row)
      {
        const double mine = exact_rows ? fused_tile_row_exact(X[k].x, X[k].y, X[k].w, jscale, sxx, sxy, syy, valid, lane)
                                       : fused_tile_row_body<false>(X[k].x, X[k].y, X[k].w, jscale, sxx, sxy, syy, valid, lane);
        th_set(k, mine);
        // the untouched row goes to memory: it is only needed again when the set of patches outside the image changes
        if (lane < 21) tile_h[((size_t)b * max_tiles + tile) * TILE_ROW + lane] = mine;
      }
#ifdef SVO_STAMPS
      if (threadIdx.x == 64) { const long long tq2 = __builtin_amdgcn_s_memtime(); s_stamp[6] += tq1 - tq0; s_stamp[7] += tq2 - tq1; }
#endif
    }
    if (NW == 8) __builtin_amdgcn_s_setprio(0);
#ifdef SVO_STAMPS
    const long long tp1 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();      // s_done / s_old of this level, footprints (own wave) in place
#ifdef SVO_STAMPS
    if (threadIdx.x == 64) s_stamp[9] += __builtin_amdgcn_s_memtime() - tp0;
    if (threadIdx.x == 0) s_stamp[5] += 0 * (tp1 - tp0);
#endif

    // ================= Gauss-Newton loop of this level =================
    // Image rows of the wave's first tile, asked for again as soon as an evaluation is through with its tiles: they
    // arrive while the wave sits in the reduction, the barriers and the one-lane solve.  The next evaluation takes
    // them from the registers if the new pose moved none of the tile's patches to another pixel (the usual case
    // after the first evaluations of a level) and otherwise loads as before.
    uint2 pf[5];
    int pf_off = -1;
#pragma unroll
    for (int j = 0; j < 5; ++j) pf[j] = make_uint2(0u, 0u);
    for (int iter = 0; iter < prm.n_iter; ++iter) {
      if (s_done) break;                                     // block-uniform (read after a barrier)
#ifdef SVO_STAMPS
      const long long t0 = __builtin_amdgcn_s_memtime();
#endif
      double T[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) {          // block-uniform: keep the model in scalar registers
        const double v = s_model[i];
        T[i] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                                __builtin_amdgcn_readfirstlane(__double2loint(v)));
      }
      double accH = 0.0;
      unsigned gone_changed = 0;                              // bit k: tile k's outside-the-image set changed
      double accJ[6] = {0, 0, 0, 0, 0, 0};
      double acc_chi = 0.0;
      unsigned acc_n = 0;
      // Software pipeline over the wave's tiles: while tile k is computed, the projection of tile k+1 is done, its five
      // current-image rows are on their way (L1 / L2) and so are its interpolated patches if they live in memory.
      // Loads return in order and nothing here is conditional, so the compiler's wait for tile k's rows leaves exactly
      // the next tile's loads in flight.  (A tile that does not exist projects nothing: its rows are read from offset
      // 0 of the image and its patches from this tile's address range, both always allocated.)
      float4 Wn[8];                                           // patches of the next tile, on their way from memory
      LppGeom gq[2];
      uint2 Crq[2][5];
      gq[0] = lpp_project(T, cam, X[0], (fl[0] & F_VISIBLE) != 0, scale, cols, rows, stride);
      const int off0 = gq[0].off;
      if (__ballot(off0 != pf_off) == 0ull) {                 // wave-uniform
#pragma unroll
        for (int j = 0; j < 5; ++j) Crq[0][j] = pf[j];
      } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) Crq[0][j] = load_row8(cur_img + off0 + j * stride);
      }
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        const int tile = tile_of(k);
        // the younger wave of a SIMD is favoured by the arbiter during its first tiles (see tile_of)
        if (NW == 8 && wave >= 4) { if (k < favoured_tiles || k == favoured_half) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }

        // this tile's interpolated patches: from LDS, or what was asked for while the previous tile was computed
        float4 Wq[8];
        if (Plan::in_lds(k)) {
          const float4* src = wc + (size_t)((Plan::lds_slot(k) * NW + wave) * 8) * TILE + lane;
#pragma unroll
          for (int c4 = 0; c4 < 8; ++c4) Wq[c4] = src[c4 * TILE];
        } else if (k == P_EXTRA && extra_lds) {                // wave-uniform: this wave's extra LDS tile
          const float4* src = wc + (size_t)((CK * NW + wave) * 8) * TILE + lane;
#pragma unroll
          for (int c4 = 0; c4 < 8; ++c4) Wq[c4] = src[c4 * TILE];
        } else {
#pragma unroll
          for (int c4 = 0; c4 < 8; ++c4) Wq[c4] = Wn[c4];
        }
        if (k + 1 < TPW) {
          gq[(k + 1) & 1] = lpp_project(T, cam, X[k + 1], (fl[k + 1] & F_VISIBLE) != 0, scale, cols, rows, stride);
#pragma unroll
          for (int j = 0; j < 5; ++j) Crq[(k + 1) & 1][j] = load_row8(cur_img + gq[(k + 1) & 1].off + j * stride);
          if (!Plan::in_lds(k + 1)) {
            const int tile_n = tile_of(k + 1) < n_tiles ? tile_of(k + 1) : (tile < n_tiles ? tile : 0);
            const float4* src = wmem + ((size_t)b * max_tiles + tile_n) * 8 * TILE + lane;
            // (a wave that keeps this tile in LDS asks for the same 1 KiB eight times instead: no branch, no traffic)
            const int c4_stride = (k + 1 == P_EXTRA && extra_lds) ? 0 : TILE;
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) Wn[c4] = src[c4 * c4_stride];
          }
        }
        if (tile >= n_tiles) continue;                       // wave-uniform: the tiles that follow do not exist either
        // ---- projection into the current image (:220-236): done one tile ahead
        const LppGeom g = gq[k & 1];
        const bool ok = g.ok;
        const bool jvalid = (fl[k] & F_JVALID) != 0;
        // All interpolation weights of the evaluation are HALVED: a power-of-two scale commutes with every rounding, so
        // W below is exactly half the reference's interpolated value, a difference of two W is its 0.5*(a - b)
        // gradient, res is half the residual, chi a quarter of the patch's chi2 and sdx/sdy half the sums -- the 32
        // multiplications by 0.5 disappear and the factors 2 and 4 are applied once per evaluation after the wave
        // reduction (exact).

        // ---- residuals (:238-279): the lane walks the 16 pixels of its own patch; no cross-lane traffic at all.
        // ref value / dx / dy of every pixel are differences of the 32 interpolated values formed once per level.
        double sdx = 0.0, sdy = 0.0;
        float sdxf = 0.0f, sdyf = 0.0f;
        float chi = 0.0f;
        {
          uint2 Cr[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) Cr[j] = Crq[k & 1][j];
          float W[6][6];
          {
            int e = 0;
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
              for (int c2 = 0; c2 < 6; ++c2) {
                if ((j == 0 || j == 5) && (c2 == 0 || c2 == 5)) { W[j][c2] = 0.0f; continue; }
                const float4 t4 = Wq[e >> 2];
                W[j][c2] = (e & 3) == 0 ? t4.x : (e & 3) == 1 ? t4.y : (e & 3) == 2 ? t4.z : t4.w;
                ++e;
              }
          }
#pragma unroll
          for (int y = 0; y < 4; ++y) {
            if (NW == 8 && y == 2 && k == favoured_half && wave >= 4) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const float refv = W[y + 1][x + 1];                                  // half the reference value
              const float dxv = W[y + 1][x + 2] - W[y + 1][x];                     // = 0.5f * (a - b) of the full values
              const float dyv = W[y + 2][x + 1] - W[y][x + 1];
              if (FAST) {
                // SVO_HIP_SIA_ARITH_FAST: contracted interpolation, the patch's chi2 and gradient moments summed in f32
                // (16 terms; the f64 form spends three conversions and two f64 operations per pixel on them)
                const float inten = interp_at_contracted(Cr[y], Cr[y + 1], x, g.w_tl, g.w_tr, g.w_bl, g.w_br);
                const float res = inten - refv;
                chi = __builtin_fmaf(res, res, chi);
                sdxf = __builtin_fmaf(dxv, res, sdxf);
                sdyf = __builtin_fmaf(dyv, res, sdyf);
                continue;
              }
              const float inten = interp_at(Cr[y], Cr[y + 1], x, g.w_tl, g.w_tr, g.w_bl, g.w_br);
              const float res = inten - refv;                                      // half the residual
              chi += res * res;
              if (M32) {
                // the 16 products dx * res, dy * res of a patch accumulated in f32 (fused: one rounding per term) instead of
                // exactly in f64: three conversions and two f64 operations per pixel less; residuals and chi2 untouched
                sdxf = __builtin_fmaf(dxv, res, sdxf);
                sdyf = __builtin_fmaf(dyv, res, sdyf);
                continue;
              }
              const double dres = (double)res;
              // f32 x f32 is exact in f64 (48-bit product): the fused form rounds exactly like mul + add
              sdx = __builtin_fma((double)dxv, dres, sdx);
              sdy = __builtin_fma((double)dyv, dres, sdy);
            }
          }
        }
        // ---- normal equations
        if (FAST || M32) { sdx = (double)sdxf; sdy = (double)sdyf; }
        const bool lin = ok && jvalid;
        if (ok) { acc_chi += (double)chi; acc_n += 16; }
        if (lin) {
          double zi = X[k].w;
          asm volatile("" : "+v"(zi));                       // opaque: nothing derived from it is kept in registers across evaluations
          // Jres_ -= J res (:273) with J = dx A + dy B summed over the patch: A sdx + B sdy written out in the
          // normalised coordinates u = x/z, v = y/z; signs and fx/2^L are applied once after the reduction
          // (this is the kernel's own summation, not an operation order of the reference: fused where it can be)
          const double u = X[k].x * zi, v = X[k].y * zi;
          const double a = __builtin_fma(u, sdx, v * sdy);
          accJ[0] = __builtin_fma(zi, sdx, accJ[0]);
          accJ[1] = __builtin_fma(zi, sdy, accJ[1]);
          accJ[2] = __builtin_fma(zi, a, accJ[2]);
          accJ[3] += __builtin_fma(v, a, sdy);
          accJ[4] += __builtin_fma(u, a, sdx);
          accJ[5] += __builtin_fma(v, sdx, -(u * sdy));
        }
        // H of the tile = its per-level row minus the patches that are outside the current image now (:76, :226-229).
        // The stored row has the patches that were outside at the previous evaluation already taken out; that set
        // changes rarely (after convergence it does not change at all).  Here only the change is noted: the
        // correction itself reads memory, and a load inside this loop -- even in a branch that is almost never taken
        // -- makes the compiler drain the loads in flight at every tile.
        const bool out_now = jvalid && !ok;
        if (!LEAN) {
          // no branch here (a taken scalar branch costs a wave ~20 cycles, a not-taken one ~10; tools/probes/branch_probe.hip):
          // the flag is rewritten by every lane and the change goes into a scalar bit
          const bool was_out = (fl[k] & F_GONE) != 0;
          gone_changed |= __ballot(out_now != was_out) != 0ull ? 1u << k : 0u;
          fl[k] = (uint8_t)((fl[k] & ~F_GONE) | (out_now ? F_GONE : 0));
        }
        const unsigned long long gone_now = LEAN ? __ballot(out_now) : 0ull;
        const unsigned long long gone_prev = LEAN ? __ballot((fl[k] & F_GONE) != 0) : 0ull;
        if (LEAN && gone_now != gone_prev) {                   // wave-uniform
          gone_changed |= 1u << k;
          fl[k] = (uint8_t)((fl[k] & ~F_GONE) | (out_now ? F_GONE : 0));
          if (LEAN) {                                          // register-bound shapes: correct the row right here
            const int tile_base = tile * TILE;
            double t = 0.0;
            if (lane < 21) t = tile_h[((size_t)b * max_tiles + tile) * TILE_ROW + lane];
            unsigned long long gone = gone_now;
            while (gone) {
              const int src = __ffsll((long long)gone) - 1;
              gone &= gone - 1;
              const double gx_ = __shfl(X[k].x, src, 64), gy_ = __shfl(X[k].y, src, 64), gzi = __shfl(X[k].w, src, 64);
              const double4 G4 = sxyz[(size_t)b * max_n + tile_base + src];
              const double g_xx = G4.x, g_xy = G4.y, g_yy = G4.z;
              double A[6], B[6];
              patch_jacobian_rows(gx_, gy_, gzi, jscale, A, B);
              double Ai = A[0], Aj = A[0], Bi = B[0], Bj = B[0];
#pragma unroll
              for (int kk = 1; kk < 6; ++kk) {
                if (tri_i == kk) { Ai = A[kk]; Bi = B[kk]; }
                if (tri_j == kk) { Aj = A[kk]; Bj = B[kk]; }
              }
              const double h = g_xx * (Ai * Aj) + g_xy * (Ai * Bj + Bi * Aj) + g_yy * (Bi * Bj);
              t -= h;
            }
            th_set(k, t);
          }
        }
        if (LEAN) { const double thk = th_get(k); if (lane < 21) accH += thk; }
      }

      __builtin_amdgcn_s_setprio(0);
      // rows of the tiles whose outside-the-image set changed: the tile's whole row minus the row of the patches that are
      // outside now, formed by all of them at once (every lane has its own point, reads its own gradient sums and the
      // wave reduces: ~350 instructions per tile whatever the number of patches; walking the patches one after the
      // other -- a dependent load and a rank update each -- cost ~500 cycles per patch, and a coarse level can have 45
      // of a tile's 64 outside: the slowest of 64 scenes spent 9 % of its time there)
      if (!LEAN && gone_changed) {                             // wave-uniform
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          if (!((gone_changed >> k) & 1u)) continue;
          const int tile = tile_of(k);
          const int i_own = tile * TILE + lane;
          const bool gone_lane = (fl[k] & F_GONE) != 0;
#ifdef SVO_STAMPS
          { const unsigned n_out = (unsigned)__popcll(__ballot(gone_lane)); if (lane == 0) { atomicAdd(&s_cnt[1], 1u); atomicAdd(&s_cnt[2], n_out); } }
#endif
          double t = 0.0;
          if (lane < 21) t = tile_h[((size_t)b * max_tiles + tile) * TILE_ROW + lane];
          const double4 G4 = sxyz[(size_t)b * max_n + (gone_lane ? i_own : 0)];            // (always a valid address)
          const double out_row = exact_rows ? fused_tile_row_exact(X[k].x, X[k].y, X[k].w, jscale, G4.x, G4.y, G4.z, gone_lane, lane)
                                            : fused_tile_row_body<false>(X[k].x, X[k].y, X[k].w, jscale, G4.x, G4.y, G4.z, gone_lane, lane);
          th_set(k, t - out_row);
        }
      }
      if (!LEAN) {
#pragma unroll
        for (int k = 0; k < TPW; ++k) {                        // rows of tiles that do not exist are zero
          const double thk = th_get(k);
          if (lane < 21) accH += thk;
        }
      }

      pf_off = off0;
#pragma unroll
      for (int j = 0; j < 5; ++j) pf[j] = load_row8(cur_img + off0 + j * stride);
      // ---- wave reduction, then the waves in fixed order, then the solve on one lane
      {
        // lanes 8j..8j+7 receive the wave total of value j: 0..5 = Jres moments (sign and fx/2^L applied here),
        // 6 = chi2, 7 = number of measurements; red[wave][0..20] = H, [21..26] = Jres, [27] = chi2, [28] = n_meas
        double v8[8];
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) v8[kk] = accJ[kk];
        v8[6] = acc_chi; v8[7] = (double)acc_n;
        const double t = wave_reduce8(v8);
        const int j = lane >> 3;
        // undo the halved weights: Jres moments x2, chi2 x4 (exact)
        const double sgn = (j == 0 || j == 1 || j == 4) ? 2.0 * jscale : (j < 6 ? -2.0 * jscale : (j == 6 ? 4.0 : 1.0));
        if (lane < 21) red[wave][lane] = accH;
        if ((lane & 7) == 0) red[wave][21 + j] = t * sgn;
      }
#ifdef SVO_STAMPS
      const long long t1 = __builtin_amdgcn_s_memtime();
#endif
      // Wave 0 runs the serial part between the two barriers, alone on the CU, and a lone wave issues one instruction
      // every ~5 cycles whatever it is: the serial part is as long as its instruction count.  So everything it needs
      // that is known already -- what the previous solve left behind -- is asked for BEFORE the first barrier (wave 0
      // is one of the older waves and gets there early), and everything nobody waits for (counters, chi2, the
      // rollback copy of the pose) is written AFTER the second one.
      double chi2_old = 0.0, nres_old = 0.0, hc_prev = 0.0;
      double coef[8], inv_row[6], cur[7];
      int it = 0, stop_old = 0, iters_l = 0, fac_valid = 0;
      if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) cur[i] = s_model[i];        // (= T, which lives in scalar registers that are needed elsewhere by now)
        chi2_old = s_chi2; nres_old = s_nres; hc_prev = s_Hc[lane < 21 ? lane : 0];   // H of the previous evaluation
        it = s_iter; stop_old = s_stop; iters_l = s_iters[level]; fac_valid = s_fac_valid;
#pragma unroll
        for (int k = 0; k < 8; ++k) coef[k] = kExpSeries[k * 4 + (lane & 3)];      // this lane's column of the exp series table
#pragma unroll
        for (int j = 0; j < 6; ++j) inv_row[j] = s_inv[(lane < 6 ? lane : 0) * 6 + j];   // lanes 0..5: their row of H^-1
      }
      __syncthreads();
#ifdef SVO_STAMPS
      const long long t2 = __builtin_amdgcn_s_memtime();
#endif
      double v = 0.0;                       // wave 0, lanes 0..28: sums over the waves
      double x[6] = {0, 0, 0, 0, 0, 0};
      double new_chi2 = 0.0;
      bool rollback = false;
      if (wave == 0) {
        {
          double r[NW];
#pragma unroll
          for (int w = 0; w < NW; ++w) r[w] = red[w][lane < 29 ? lane : 28];       // all partials asked for at once
#pragma unroll
          for (int w = 0; w < NW; ++w) v += r[w];                                  // the waves in fixed order
        }
        // H_ / Jres_ of the last evaluation are reported at the end: carried in a register, or -- by the register-bound
        // shapes, which have the LDS to spare -- kept in LDS
        if (!LEAN) v_last = v;
        else if (lane < 29) s_last[lane] = v;
        // H is the sum of the per-level tile rows minus the patches outside the image at this evaluation: as long as
        // that set does not change it is bit for bit the H of the previous evaluation, and what was derived from it
        // is reused (a deterministic function of H, so the result is the same number)
        const bool h_same = lane >= 21 || __double_as_longlong(v) == __double_as_longlong(hc_prev);
        const bool reuse = fac_valid != 0 && __ballot(!h_same) == 0ull;            // wave-uniform
        // lanes 21..26 hold Jres: wave-uniform copies, no trip through LDS
        double Jres[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) Jres[i] = readlane_f64(v, 21 + i);
#ifdef SVO_STAMPS
        const long long q0 = __builtin_amdgcn_s_memtime();
        if (lane == 0) s_stamp[5] += q0 - t2;
#endif
        if (!reuse) {
#ifdef SVO_STAMPS
          if (lane == 0) atomicAdd(&s_cnt[0], 1u);
#endif
          // H changed: H^-1 is formed again (cold, out of line)
          if (lane < 21) s_Hc[lane] = v;
          fused_refactor_cold(s_Hc, s_fac, s_ftr, &s_fac_valid, s_inv, lane);
#pragma unroll
          for (int j = 0; j < 6; ++j) inv_row[j] = s_inv[(lane < 6 ? lane : 0) * 6 + j];
        }
        // x = H^-1 Jres, one component per lane (H^-1 is symmetric: lane i uses the column it computed as row i)
        double xi = inv_row[0] * Jres[0];
#pragma unroll
        for (int j = 1; j < 6; ++j) xi += inv_row[j] * Jres[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] = readlane_f64(xi, i);
        // the four series of exp(-x) on lanes 0..3 (theta^2 is the same for x and -x), collected wave-uniform
        const double zt = x[3] * x[3] + x[4] * x[4] + x[5] * x[5];
        const double ser = se3_exp_series_lane(coef, (lane & 2) ? 0.25 * zt : zt);
        const double qs_t = readlane_f64(ser, 0), pc_t = readlane_f64(ser, 1), ps_h = readlane_f64(ser, 2), pc_h = readlane_f64(ser, 3);
#ifdef SVO_STAMPS
        const long long q1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) s_stamp[3] += q1 - q0;
#endif
        // solve()/update() of S/sparse_img_align.cpp:291-308 inside the loop of I/nlls_solver_impl.hpp:35-99
        const bool stop_now = stop_old != 0 || x[0] != x[0];                       // NaN -> stop_ (:52-59)
        bool worse = false;
        if (prm.early_stop) {                                                      // the reference's exits need chi2 now
          new_chi2 = (double)((float)readlane_f64(v, 27) / (float)readlane_f64(v, 28));   // (:285)
          worse = it > 0 && new_chi2 > chi2_old;
        }
        rollback = worse || stop_now;                                              // wave-uniform
        if (lane == 0) {
          if (rollback) {
#pragma unroll
            for (int i = 0; i < 7; ++i) s_model[i] = s_old[i];                     // rollback (:72)
            s_done = 1;
            if (stop_now) s_stop = 1;
          } else {
            double mx[6], dT[7], nm[7];
#pragma unroll
            for (int i = 0; i < 6; ++i) mx[i] = -x[i];
#ifdef SVO_STAMPS
            const long long q2 = __builtin_amdgcn_s_memtime();
#endif
            if (zt <= 0.25) se3_exp_small_finish(mx, zt, qs_t, pc_t, ps_h, pc_h, dT);
            else se3_exp_cold(mx, dT);                                             // large angles: the library path
            se3_mul(cur, dT, nm);                                                  // T_new = T_old * exp(-x) (:307)
#ifdef SVO_STAMPS
            const long long q3 = __builtin_amdgcn_s_memtime();
            s_stamp[4] += q3 - q2;
#endif
#pragma unroll
            for (int i = 0; i < 7; ++i) s_model[i] = nm[i];
            bool done = it + 1 >= prm.n_iter;
            if (prm.early_stop) {                                                  // :97-98
              double mxn = -1;
#pragma unroll
              for (int i = 0; i < 6; ++i) { double a = fabs(x[i]); if (a > mxn) mxn = a; }
              if (mxn <= prm.eps) done = true;
            }
            if (done) s_done = 1;
          }
        }
#ifdef SVO_STAMPS
        if (lane == 0) s_stamp[8] += __builtin_amdgcn_s_memtime() - t2;
#endif
      }
      __syncthreads();
      if (wave == 0) {
        // what only wave 0 itself reads again (and thread 0 at the end): the other waves are already evaluating
        const double n_meas_d = readlane_f64(v, 28);       // exact integer counts (multiples of 16) carried as doubles
        if (!prm.early_stop) new_chi2 = (double)((float)readlane_f64(v, 27) / (float)n_meas_d);   // (:285)
        if (lane == 0) {
          s_nmeas = n_meas_d;
          s_nres = nres_old + n_meas_d * 0.0625;
          s_iters[level] = iters_l + 1;
#pragma unroll
          for (int i = 0; i < 6; ++i) s_x[i] = x[i];
          if (!rollback) {
#pragma unroll
            for (int i = 0; i < 7; ++i) s_old[i] = cur[i];
            s_chi2 = new_chi2;
            s_iter = it + 1;
          }
        }
      }
#ifdef SVO_STAMPS
      if (threadIdx.x == 64) { const long long t3 = __builtin_amdgcn_s_memtime(); s_lev[level] += t3 - t0; s_stamp[0] += t1 - t0; s_stamp[1] += t2 - t1; s_stamp[2] += t3 - t2; }
      if (lane == 0) { s_wst[wave] += t1 - t0; s_wst[8 + wave] += t2 - t1; }
#endif
    }
    __syncthreads();
  }

  if (!LEAN && wave == 0 && lane < 27) {                     // H_ (upper triangle, mirrored) and Jres_ of the last evaluation
    FrameState& s = st[b];
    if (lane < 21) { s.H[tri_i * 6 + tri_j] = v_last; s.H[tri_j * 6 + tri_i] = v_last; }
    else s.Jres[lane - 21] = v_last;
  }
  if (threadIdx.x == 0) {
    FrameState& s = st[b];
    double T[7], m[7];
    for (int i = 0; i < 7; ++i) { m[i] = s_model[i]; s.model[i] = m[i]; s.old_model[i] = s_old[i]; }
    se3_mul(m, c.T_ref_w, T);                                                    // :89
    for (int i = 0; i < 7; ++i) s.T_cur_w[i] = empty ? c.T_cur_w_init[i] : T[i];
    s.chi2 = s_chi2; s.stop = s_stop; s.iter = s_iter; s.level_done = 1; s.empty = empty;
    s.n_meas = (unsigned long long)(s_nmeas + 0.5); s.n_res = (unsigned long long)(s_nres + 0.5); s.n_pre = s_npre;
    {
      for (int i = 0; i < 6; ++i) s.x[i] = s_x[i];
      if (LEAN) {
        int kk = 0;
        for (int i = 0; i < 6; ++i)
          for (int j = i; j < 6; ++j) { s.H[i * 6 + j] = s_last[kk]; s.H[j * 6 + i] = s_last[kk]; ++kk; }
        for (int i = 0; i < 6; ++i) s.Jres[i] = s_last[21 + i];
      }
#ifdef SVO_STAMPS
      for (int i = 0; i < 16; ++i) s.H[i] = (double)s_wst[i];
      for (int i = 5; i < 10; ++i) s.H[16 + i - 5] = (double)s_stamp[i];
      s.x[5] = (double)(__builtin_amdgcn_s_memtime() - k_m0) / (double)(__builtin_amdgcn_s_memrealtime() - k_r0) * 0.1;   // GHz over the kernel
      s.chi2 = (double)(__builtin_amdgcn_s_memtime() - k_m0);
      s.H[30] = (double)s_cnt[0]; s.H[31] = (double)s_cnt[1]; s.H[32] = (double)s_cnt[2];
      for (int i = 0; i < 5; ++i) s.H[21 + i] = (double)s_lev[i];                                                              // cycles of this workgroup
      s.x[0] = (double)s_stamp[0]; s.x[1] = (double)s_stamp[1]; s.x[2] = (double)s_stamp[2]; s.x[3] = (double)s_stamp[3]; s.x[4] = (double)s_stamp[4];
#endif
    }
    for (int i = 0; i < SVO_HIP_MAX_LEVELS; ++i) s.iters[i] = s_iters[i];
  }
}

// Parity test support: the reference patches exactly as sia_fused_kernel forms them at `level` (same feature position in
// f32, same weights, same interpolation: the two device functions above), written in the layout of the streaming kernels'
// caches -- reference value = 2 W (exact), dx / dy = differences of W -- so that svo_hip_sia_download_caches hands them to
// the test.  One thread per patch.
__global__ __launch_bounds__(256) void sia_fused_patch_dump_kernel(const FrameConst* __restrict__ fc, int slot, const uint8_t* __restrict__ ref_base,
                                                                   size_t pyr_bytes, LevelGeom g, int level, int max_n, const double* __restrict__ px,
                                                                   const uint8_t* __restrict__ has_point, float4* __restrict__ ref_cache,
                                                                   float4* __restrict__ dxc, float4* __restrict__ dyc, uint8_t* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= fc[slot].n_feat) return;
  const size_t fo = (size_t)slot * max_n + i;
  const int border = 3, stride = g.cols;
  const float scale = 1.0f / (1 << level);
  const float u_ref = (float)px[2 * fo] * scale, v_ref = (float)px[2 * fo + 1] * scale;
  const int u_ref_i = (int)floorf(u_ref), v_ref_i = (int)floorf(v_ref);
  const bool valid = has_point[fo] != 0 && !(u_ref_i - border < 0 || v_ref_i - border < 0 || u_ref_i >= g.cols - border || v_ref_i >= g.rows - border);
  flags[fo] = valid ? 1 : 0;
  if (!valid) return;
  float w[4];
  fused_ref_weights(u_ref, v_ref, u_ref_i, v_ref_i, w);
  const uint8_t* img = ref_base + (size_t)slot * pyr_bytes + g.ref_off + (v_ref_i - 3) * stride + (u_ref_i - 3);
  uint2 F[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) F[j] = load_row8(img + j * stride);
  float W[6][6];
  fused_interp_W(F, w[0], w[1], w[2], w[3], W);
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    float v[4], dx[4], dy[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      v[x] = 2.0f * W[y + 1][x + 1];
      dx[x] = W[y + 1][x + 2] - W[y + 1][x];
      dy[x] = W[y + 2][x + 1] - W[y][x + 1];
    }
    ref_cache[fo * 4 + y] = make_float4(v[0], v[1], v[2], v[3]);
    dxc[fo * 4 + y] = make_float4(dx[0], dx[1], dx[2], dx[3]);
    dyc[fo * 4 + y] = make_float4(dy[0], dy[1], dy[2], dy[3]);
  }
}

// Tracking chain (svo_track.hip): slot 0 of a solver filled from device arrays -- the previous frame's features and the
// point table -- so that SparseImgAlign::run(last_frame, new_frame) with new_frame->T_f_w_ = last_frame->T_f_w_
// (frame_handler_mono.cpp:175,186-188) starts without a host round trip.  One workgroup.
__global__ __launch_bounds__(256) void sia_gather_kernel(FrameConst* __restrict__ fc, Cam cam, const int* __restrict__ n_feat_dev, int max_n,
                                                         const double* __restrict__ T_last_w, const double* __restrict__ px_in,
                                                         const double* __restrict__ f_in, const int32_t* __restrict__ point_in,
                                                         const double* __restrict__ pt_pos, double* __restrict__ px, double* __restrict__ f,
                                                         double* __restrict__ pos, uint8_t* __restrict__ has_point) {
  int n = *n_feat_dev;
  if (n > max_n) n = max_n;
  if (threadIdx.x == 0) {
    FrameConst c;
    c.cam = cam;
    double Tinv[7];
    se3_inverse(T_last_w, Tinv);                                     // Frame::pos() (I/frame.h:105)
    for (int i = 0; i < 7; ++i) { c.T_ref_w[i] = T_last_w[i]; c.T_cur_w_init[i] = T_last_w[i]; }
    c.ref_pos[0] = Tinv[0]; c.ref_pos[1] = Tinv[1]; c.ref_pos[2] = Tinv[2];
    c.n_feat = n; c.pad = 0;
    fc[0] = c;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int p = point_in[i];
    px[2 * i] = px_in[2 * i]; px[2 * i + 1] = px_in[2 * i + 1];
    f[3 * i] = f_in[3 * i]; f[3 * i + 1] = f_in[3 * i + 1]; f[3 * i + 2] = f_in[3 * i + 2];
    has_point[i] = p >= 0 ? 1 : 0;                                   // point == NULL features are skipped (sparse_img_align.cpp:118)
    pos[3 * i] = p >= 0 ? pt_pos[3 * (size_t)p] : 0.0;
    pos[3 * i + 1] = p >= 0 ? pt_pos[3 * (size_t)p + 1] : 0.0;
    pos[3 * i + 2] = p >= 0 ? pt_pos[3 * (size_t)p + 2] : 1.0;
  }
}

}  // namespace

struct svo_hip_sia {
  svo_hip_ctx* ctx = nullptr;
  int batch = 0, max_n = 0;
  const svo_hip_pyramid* ref = nullptr;
  const svo_hip_pyramid* cur = nullptr;
  // device buffers
  FrameConst* fc = nullptr;
  FrameState* st = nullptr;                 // the frames' Gauss-Newton state (home buffer: results are read from here)
  FrameState* st_alt = nullptr;             // second buffer of the sharded solve (see sharded_level)
  double *px = nullptr, *f = nullptr, *pos = nullptr;
  uint8_t *has_point = nullptr, *visible = nullptr;
  float4 *ref_cache = nullptr, *dxc = nullptr, *dyc = nullptr;   // [batch][max_n][4 rows] x float4
  double4 *sxyz = nullptr, *xyz4 = nullptr;                      // {sxx,sxy,syy,-}, {x,y,z,1/z} per patch
  double* tile_h = nullptr;                                      // [batch][max_tiles][TILE_ROW]
  float4* wmem = nullptr;                                        // fused kernel: [batch][max_tiles][8][TILE] interpolated patches
  int max_tiles = 0;
  double* partial = nullptr;
  double* partial_alt = nullptr;   // sharded solve: a launch reads the exchanged partials of the previous evaluation from one buffer and writes its own into the other
  double* reduce_own = nullptr;
  double* reduce = nullptr;
  unsigned int* n_pre_count = nullptr;
  // host mirrors
  FrameConst* h_fc = nullptr;
  bool fc_dirty = true;
  int shard_rank = 0, shard_world = 1;
  // tuning / diagnostic switches of this object (svo_hip_sia_set_option); 0 / -1 = automatic
  int opt_mode = 0, opt_waves = 0, opt_chunks = 0, opt_extra_lds = -1, opt_old_tiles = 0, opt_arith = SVO_HIP_SIA_ARITH_EXACT;
  // NLLSSolver's other branches (svo_nlls.hip): method_, setRobustCostFunction
  int opt_method = SVO_HIP_SIA_METHOD_GAUSS_NEWTON, opt_scale = SVO_HIP_SIA_SCALE_UNIT, opt_weight = SVO_HIP_SIA_WEIGHT_UNIT;
  int opt_chi2 = SVO_HIP_SIA_CHI2_PER_PATCH;
  svo_nlls_ext* nlls = nullptr;
  // stepwise state
  svo_hip_sia_params prm{};
  int n_slots = 0, level = -1, chunks = 1;
  bool begun = false;
  // optional HIP-event timing of the two heavy kernels, on the context stream
  int last_mode = 0;          // 0 = streaming kernels, 1 = fused one-workgroup-per-frame kernel
  bool profiling = false;
  std::vector<hipEvent_t> ev_res, ev_pre;     // start/stop pairs
  size_t ev_res_used = 0, ev_pre_used = 0;
  // svo_hip_sia_run_sharded with graph replay: one instantiated graph per pyramid level (level_begin + n_iter x
  // {accumulate, all-reduce, solve_update}), valid for the configuration in graph_key
  bool sharded_graph = false;
  hipGraphExec_t level_graph[SVO_HIP_MAX_LEVELS] = {nullptr};
  struct GraphKey {
    unsigned long long comm_id = 0;            // svo_comm_id: never reused, unlike the communicator's address
    int n_slots = 0, n_iter = 0, early_stop = 0, max_level = 0, min_level = 0, rank = 0, world = 0, chunks = 0;
    double eps = 0.0;
    const void *ref_base = nullptr, *cur_base = nullptr, *reduce = nullptr;
    int width = 0, height = 0;
    size_t pyr_bytes = 0;
    bool operator==(const GraphKey& o) const {     // field by field: the struct has padding
      return comm_id == o.comm_id && n_slots == o.n_slots && n_iter == o.n_iter && early_stop == o.early_stop &&
             max_level == o.max_level && min_level == o.min_level && rank == o.rank && world == o.world && chunks == o.chunks &&
             eps == o.eps && ref_base == o.ref_base && cur_base == o.cur_base && reduce == o.reduce && width == o.width &&
             height == o.height && pyr_bytes == o.pyr_bytes;
    }
  } graph_key;
};

namespace {

void drop_level_graphs(svo_hip_sia* s) {
  for (int l = 0; l < SVO_HIP_MAX_LEVELS; ++l)
    if (s->level_graph[l]) { (void)hipGraphExecDestroy(s->level_graph[l]); s->level_graph[l] = nullptr; }
}

template <typename T>
int dev_alloc(svo_hip_ctx* ctx, T** p, size_t count) {
  void* d = nullptr;
  int rc = svo_hip_malloc(ctx, &d, count * sizeof(T));
  *p = (T*)d;
  return rc;
}

// blocks per frame for the residual kernel: aim at >= ~1024 blocks (4096 waves) on the chip while
// giving every wave as many tiles as possible (the per-wave reduction is amortised over them)
int pick_chunks(const svo_hip_sia* s, int n_slots, int max_n) {
  int tiles = (max_n + TILE - 1) / TILE;
  int cap = (tiles + 3) / 4;                 // at least one tile per wave
  int c = s->opt_chunks > 0 ? s->opt_chunks : (1024 + n_slots - 1) / n_slots;
  if (c > cap) c = cap;
  if (c > MAX_CHUNKS) c = MAX_CHUNKS;
  if (c < 1) c = 1;
  return c;
}

// next start/stop event pair of a pool (grown on demand); nullptr when profiling is off
hipEvent_t* next_events(svo_hip_sia* s, std::vector<hipEvent_t>& pool, size_t& used) {
  if (!s->profiling) return nullptr;
  if (used + 2 > pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return nullptr;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return nullptr; }
    pool.push_back(a); pool.push_back(b);
  }
  hipEvent_t* e = &pool[used];
  used += 2;
  return e;
}

int flush_fc(svo_hip_sia* s) {
  if (!s->fc_dirty) return SVO_HIP_OK;
  SVO_CHECK_HIP(s->ctx, hipMemcpyAsync(s->fc, s->h_fc, sizeof(FrameConst) * s->batch, hipMemcpyHostToDevice, s->ctx->stream));
  s->fc_dirty = false;
  return SVO_HIP_OK;
}

// One launch of the fused kernel over slots [0, n_launch).
template <int NW, int TPW, int CK, int VARIANT>
int launch_fused_x(svo_hip_sia* s, int n_launch, const svo_hip_sia_params* prm, size_t lds_bytes, int tiles_young, int n_extra = 0) {
  svo_hip_ctx* ctx = s->ctx;
  // > 64 KiB of dynamic LDS has to be allowed explicitly (per device: set it on every launch, it is cheap)
  SVO_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&sia_fused_kernel<NW, TPW, CK, VARIANT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  FusedLevels lv;
  memset(&lv, 0, sizeof(lv));
  for (int l = 0; l < s->ref->n_levels; ++l) {
    lv.cols[l] = s->ref->width >> l; lv.rows[l] = s->ref->height >> l;
    lv.ref_off[l] = s->ref->level_offset[l]; lv.cur_off[l] = s->cur->level_offset[l];
  }
  FusedParams fp;
  fp.max_level = prm->max_level; fp.min_level = prm->min_level; fp.n_iter = prm->n_iter;
  fp.early_stop = prm->early_stop; fp.eps = prm->eps;
  fp.moments_f32 = s->opt_arith == SVO_HIP_SIA_ARITH_MOMENTS_F32 ? 1 : 0;
  hipLaunchKernelGGL((sia_fused_kernel<NW, TPW, CK, VARIANT>), dim3(n_launch), dim3(NW * 64), lds_bytes, ctx->stream, s->fc, s->st,
                     s->ref->base, s->cur->base, s->ref->pyr_bytes, lv, s->max_n, s->px, s->f, s->pos, s->has_point, s->sxyz, s->tile_h,
                     s->wmem, s->max_tiles, fp, tiles_young, n_extra);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

// LDS tiles (8 KiB) the 8-wave shape can hold beyond ck per wave: what 160 KiB leave next to the kernel's static LDS
template <int TPW, int CK>
int fused_extra_tiles_t() {
  static int cached = -1;
  if (cached < 0) {
    hipFuncAttributes at;
    cached = 0;
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&sia_fused_kernel<8, TPW, CK, FUSED_PLAIN>)) == hipSuccess) {
      const long free_b = 160L * 1024 - (long)at.sharedSizeBytes - (long)FUSED_WAVES * CK * FUSED_WC_BYTES;
      cached = free_b > 0 ? (int)(free_b / FUSED_WC_BYTES) : 0;
      if (cached > FUSED_EXTRA_TILES) cached = FUSED_EXTRA_TILES;
    }
  }
  return cached;
}
int fused_extra_tiles(int tpw, int ck) {
  (void)ck;
  switch (tpw) {
    case 3: return fused_extra_tiles_t<3, 2>();
    case 4: return fused_extra_tiles_t<4, 2>();
    default: return 0;
  }
}

// tiles per SIMD of the largest frame of a launch -> the older wave's share of them (see tile_of in the kernel)
int fused_old_share(const svo_hip_sia* s, int max_n, int* per_simd_out) {
  const int tiles = max_n > 0 ? (max_n + TILE - 1) / TILE : 1;
  const int per_simd = (tiles + 3) / 4;
  int old_share = (per_simd + 1) / 2;
  const int ovr = s->opt_old_tiles;                                      // diagnostic override
  if (ovr > 0 && ovr >= (per_simd + 1) / 2 && ovr <= per_simd && ovr <= FUSED_MAX_TPW) old_share = ovr;
  static_assert(FUSED_MAX_TPW * 2 * 4 >= FUSED_MAX_TILES, "dispatch covers every tile count");
  *per_simd_out = per_simd;
  return old_share < 1 ? 1 : old_share;
}

// The shape of the fused kernel for a launch of n_launch pairs whose largest frame has max_n patches.
template <int VARIANT>
int launch_fused_shape(svo_hip_sia* s, int n_launch, int max_n, const svo_hip_sia_params* prm) {
  svo_hip_ctx* ctx = s->ctx;
  int per_simd = 1;
  const int tpw = fused_old_share(s, max_n, &per_simd);
  // Two frame pairs per CU (4-wave workgroups) when the launch has at least two pairs for every CU and a wave can hold
  // a quarter of the tiles: the one-lane solve phase of one pair then overlaps the evaluation of the other.
  // (measured with 512 pairs per launch: 200 patches 715 k against 444 k frames/s, 500: 549 k / 393 k, 1000: 314 k /
  // 277 k; at 2000 patches a wave would own 8 tiles and their {x,y,z,1/z} no longer fit in 256 VGPRs: 97 k / 163 k)
  bool four = n_launch >= 2 * ctx->n_cu && per_simd <= 4;
  if (s->opt_waves) four = s->opt_waves == 4 && per_simd <= 4;          // diagnostic override: 4 or 8
  if (four) {
    const size_t lds4 = (size_t)4 * (per_simd < 2 ? 1 : 2) * FUSED_WC_BYTES;
    switch (per_simd) {                                                // tiles per wave (3 runs as 4 with an empty slot)
      case 1: return launch_fused_x<4, 1, 1, VARIANT>(s, n_launch, prm, lds4, 0);
      case 2: return launch_fused_x<4, 2, 2, VARIANT>(s, n_launch, prm, lds4, 0);
      case 3:
      case 4: return launch_fused_x<4, 4, 2, VARIANT>(s, n_launch, prm, lds4, 0);
      default: break;
    }
  }
  const int ty = per_simd - tpw;             // the younger wave's share of a SIMD's tiles
  SVO_REQUIRE(ctx, ty >= 0 && ty <= tpw);
  // two tiles per wave keep their interpolated patches in LDS (8 waves x 2 x 8 KiB), the others in memory
  const int ck = tpw < 2 ? tpw : 2;
  int n_extra = tpw > ck ? fused_extra_tiles(tpw, ck) : 0;
  if (s->opt_extra_lds >= 0) n_extra = tpw > ck ? s->opt_extra_lds : 0;   // diagnostic override
  if (n_extra < 0 || n_extra > (tpw > ck ? fused_extra_tiles(tpw, ck) : 0)) n_extra = 0;
  const size_t lds = (size_t)(FUSED_WAVES * ck + n_extra) * FUSED_WC_BYTES;
  switch (tpw) {                 // tiles of an older wave
    case 1: return launch_fused_x<8, 1, 1, VARIANT>(s, n_launch, prm, lds, ty);
    case 2: return launch_fused_x<8, 2, 2, VARIANT>(s, n_launch, prm, lds, ty);
    case 3: return launch_fused_x<8, 3, 2, VARIANT>(s, n_launch, prm, lds, ty, n_extra);
    case 4: return launch_fused_x<8, 4, 2, VARIANT>(s, n_launch, prm, lds, ty, n_extra);
    case 5: return launch_fused_x<8, 5, 2, VARIANT>(s, n_launch, prm, lds, ty, n_extra);
    case 6: return launch_fused_x<8, 6, 2, VARIANT>(s, n_launch, prm, lds, ty, n_extra);
    default: break;
  }
  return svo_fail(ctx, SVO_HIP_ERR_INVALID, "fused SparseImgAlign", "unsupported tiles-per-wave");
}

// setRobustCostFunction switches use_weights_ on for every scale estimator but UnitScale (nlls_solver_impl.hpp:234-262)
// (... and plain Gauss-Newton takes the same driver when chi2 is asked for in the reference's summation order)
bool nlls_branches(const svo_hip_sia* s) {
  return s->opt_method != SVO_HIP_SIA_METHOD_GAUSS_NEWTON || s->opt_scale != SVO_HIP_SIA_SCALE_UNIT || s->opt_chi2 != SVO_HIP_SIA_CHI2_PER_PATCH;
}

// The fused kernel handles frames of at most FUSED_MAX_TILES tiles that are not patch-sharded.
bool fused_applies(const svo_hip_sia* s, int n_slots) {
  if (s->opt_mode == SVO_HIP_SIA_MODE_STREAM) return false;
  if (s->shard_world != 1) return false;
  int max_n = 0;
  for (int i = 0; i < n_slots; ++i) max_n = s->h_fc[i].n_feat > max_n ? s->h_fc[i].n_feat : max_n;
  return (max_n + TILE - 1) / TILE <= FUSED_MAX_TILES;
}

// A batch that holds a frame with a handful of patches (fewer than FUSED_EXACT_ROW_BELOW) is launched with the instance
// whose workgroups choose the form of their Hessian rows by their own patch count (EXACT_ROWS, see sia_fused_kernel);
// every other batch with the instance that has no such branch.  (Two alternatives were measured on 256 C1 pairs of which
// one has 5 or 12 patches -- tools/mixed_batch_bench.py: the whole batch in the all-exact instance of round 2, which spills
// 469 VGPRs; and a second launch for the tiny slots on a side stream, which overlaps the main launch but runs 3.5 times
// slower beside it than alone -- 1.58 ms against 0.45 ms, two different kernels evicting each other from the
// instruction cache their CUs share -- and so made the step 31-37 % longer.  The per-workgroup branch costs a mixed
// batch 1.4 % in fixed-work mode and 3 % with the reference's exits, and a pure batch nothing.)
int run_fused(svo_hip_sia* s, int n_slots, const svo_hip_sia_params* prm) {
  svo_hip_ctx* ctx = s->ctx;
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  SVO_REQUIRE(ctx, s->ref && s->cur);
  SVO_REQUIRE(ctx, n_slots > 0 && n_slots <= s->batch);
  SVO_REQUIRE(ctx, prm->min_level >= 0 && prm->max_level >= prm->min_level && prm->max_level < s->ref->n_levels);
  SVO_REQUIRE(ctx, prm->n_iter >= 0);
  SVO_REQUIRE(ctx, s->ref->pyr_bytes == s->cur->pyr_bytes);
  int rc = flush_fc(s);
  if (rc != SVO_HIP_OK) return rc;
  s->begun = false;
  s->last_mode = 1;
  bool tiny = false;
  int max_n = 0;
  for (int i = 0; i < n_slots; ++i) {
    const int n = s->h_fc[i].n_feat;
    tiny = tiny || (n > 0 && n < FUSED_EXACT_ROW_BELOW);
    if (n > max_n) max_n = n;
  }
  hipEvent_t* ev = next_events(s, s->ev_res, s->ev_res_used);
  if (ev) (void)hipEventRecord(ev[0], ctx->stream);
  // (the fast arithmetic is an option of the plain instance: a batch with a tiny frame runs with exact or f32 moments, as set)
  rc = tiny ? launch_fused_shape<FUSED_EXACT_ROWS>(s, n_slots, max_n, prm)
            : s->opt_arith == SVO_HIP_SIA_ARITH_FAST ? launch_fused_shape<FUSED_FAST>(s, n_slots, max_n, prm)
            : s->opt_arith == SVO_HIP_SIA_ARITH_MOMENTS_F32 ? launch_fused_shape<FUSED_M32>(s, n_slots, max_n, prm)
                                                            : launch_fused_shape<FUSED_PLAIN>(s, n_slots, max_n, prm);
  if (ev) (void)hipEventRecord(ev[1], ctx->stream);
  return rc;
}

}  // namespace

// (slot: the solver slot of the caller -- 0 for a lone tracker, the camera's index inside a tracker group)
int svo_sia_prepare_from_device(svo_hip_sia* s, int slot, const svo_hip_camera* cam, int n_feat_host, const int* n_feat_dev,
                                const double* T_last_w_dev, const double* px_dev, const double* f_dev, const int32_t* point_dev,
                                const double* pt_pos_dev) {
  if (!s || !cam) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  SVO_REQUIRE(ctx, slot >= 0 && slot < s->batch);
  SVO_REQUIRE(ctx, n_feat_host >= 0 && n_feat_host <= s->max_n && n_feat_dev && T_last_w_dev && px_dev && f_dev && point_dev && pt_pos_dev);
  const size_t o = (size_t)slot * s->max_n;
  hipLaunchKernelGGL(sia_gather_kernel, dim3(1), dim3(256), 0, ctx->stream, s->fc + slot, svo_make_cam(*cam), n_feat_dev, s->max_n, T_last_w_dev,
                     px_dev, f_dev, point_dev, pt_pos_dev, s->px + 2 * o, s->f + 3 * o, s->pos + 3 * o, s->has_point + o);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  // the host mirror only steers the launch (kernel shape by the feature count); the device record is authoritative
  s->h_fc[slot].n_feat = n_feat_host;
  s->h_fc[slot].cam = svo_make_cam(*cam);
  s->fc_dirty = false;
  return SVO_HIP_OK;
}

const svo_dev::FrameState* svo_sia_state_dev(const svo_hip_sia* s, int slot) { return s ? s->st + slot : nullptr; }

// a slot's input arrays, for a caller whose own kernel fills them (the hand-over kernel of svo_track.hip writes the next
// frame's reference features there directly); svo_sia_note_device_slot then tells the host mirror what the device holds
int svo_sia_slot_arrays(svo_hip_sia* s, int slot, svo_dev::FrameConst** fc, double** px, double** f, double** pos, uint8_t** has_point, int* max_n) {
  if (!s || slot < 0 || slot >= s->batch) return SVO_HIP_ERR_INVALID;
  const size_t o = (size_t)slot * s->max_n;
  *fc = s->fc + slot; *px = s->px + 2 * o; *f = s->f + 3 * o; *pos = s->pos + 3 * o; *has_point = s->has_point + o; *max_n = s->max_n;
  return SVO_HIP_OK;
}

int svo_sia_note_device_slot(svo_hip_sia* s, int slot, const svo_hip_camera* cam, int n_feat_host) {
  if (!s || !cam) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(s->ctx, slot >= 0 && slot < s->batch && n_feat_host >= 0 && n_feat_host <= s->max_n);
  s->h_fc[slot].n_feat = n_feat_host;
  s->h_fc[slot].cam = svo_make_cam(*cam);
  s->fc_dirty = false;
  return SVO_HIP_OK;
}

extern "C" {

int svo_hip_sia_create(svo_hip_ctx* ctx, int batch, int max_features, svo_hip_sia** out) {
  if (!ctx || !out) return SVO_HIP_ERR_INVALID;
  *out = nullptr;
  SVO_REQUIRE(ctx, batch > 0 && max_features > 0);
  svo_hip_sia* s = new (std::nothrow) svo_hip_sia();
  if (!s) return SVO_HIP_ERR_NOMEM;
  s->ctx = ctx; s->batch = batch; s->max_n = max_features;
  const size_t bn = (size_t)batch * max_features;
  int rc = SVO_HIP_OK;
  auto A = [&](int r) { if (rc == SVO_HIP_OK) rc = r; };
  A(dev_alloc(ctx, &s->fc, batch)); A(dev_alloc(ctx, &s->st, batch)); A(dev_alloc(ctx, &s->st_alt, batch));
  A(dev_alloc(ctx, &s->px, bn * 2)); A(dev_alloc(ctx, &s->f, bn * 3)); A(dev_alloc(ctx, &s->pos, bn * 3));
  A(dev_alloc(ctx, &s->has_point, bn)); A(dev_alloc(ctx, &s->visible, bn));
  s->max_tiles = (max_features + TILE - 1) / TILE + 1;
  A(dev_alloc(ctx, &s->ref_cache, bn * 4 + 256)); A(dev_alloc(ctx, &s->dxc, bn * 4 + 256)); A(dev_alloc(ctx, &s->dyc, bn * 4 + 256));
  A(dev_alloc(ctx, &s->sxyz, bn + 64)); A(dev_alloc(ctx, &s->xyz4, bn + 64));
  A(dev_alloc(ctx, &s->tile_h, (size_t)batch * s->max_tiles * TILE_ROW));
  A(dev_alloc(ctx, &s->wmem, (size_t)batch * s->max_tiles * 8 * TILE));
  A(dev_alloc(ctx, &s->partial, (size_t)batch * MAX_CHUNKS * RED));
  A(dev_alloc(ctx, &s->partial_alt, (size_t)batch * MAX_CHUNKS * RED));
  A(dev_alloc(ctx, &s->reduce_own, (size_t)batch * RED));
  A(dev_alloc(ctx, &s->n_pre_count, batch));
  s->h_fc = new (std::nothrow) FrameConst[batch];
  if (rc != SVO_HIP_OK || !s->h_fc) { svo_hip_sia_destroy(s); return rc != SVO_HIP_OK ? rc : SVO_HIP_ERR_NOMEM; }
  memset(s->h_fc, 0, sizeof(FrameConst) * batch);
  s->reduce = s->reduce_own;
  (void)hipMemsetAsync(s->has_point, 0, bn, ctx->stream);
  (void)hipMemsetAsync(s->ref_cache, 0, bn * 16 * sizeof(float), ctx->stream);
  (void)hipMemsetAsync(s->dxc, 0, bn * 16 * sizeof(float), ctx->stream);
  (void)hipMemsetAsync(s->dyc, 0, bn * 16 * sizeof(float), ctx->stream);
  (void)hipMemsetAsync(s->sxyz, 0, bn * sizeof(double4), ctx->stream);
  (void)hipMemsetAsync(s->xyz4, 0, bn * sizeof(double4), ctx->stream);
  (void)hipMemsetAsync(s->st, 0, sizeof(FrameState) * batch, ctx->stream);
  (void)hipMemsetAsync(s->st_alt, 0, sizeof(FrameState) * batch, ctx->stream);
  *out = s;
  return SVO_HIP_OK;
}

int svo_hip_sia_destroy(svo_hip_sia* s) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  (void)hipStreamSynchronize(ctx->stream);
  drop_level_graphs(s);
  void* ptrs[] = {s->st_alt, s->fc, s->st, s->px, s->f, s->pos, s->has_point, s->visible, s->ref_cache, s->dxc, s->dyc,
                  s->sxyz, s->xyz4, s->tile_h, s->wmem, s->partial, s->partial_alt, s->reduce_own, s->n_pre_count};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (s->nlls) svo_nlls_free(s->nlls);
  for (hipEvent_t e : s->ev_res) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->ev_pre) (void)hipEventDestroy(e);
  delete[] s->h_fc;
  delete s;
  return SVO_HIP_OK;
}

int svo_hip_sia_set_frames(svo_hip_sia* s, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur) {
  if (!s || !ref || !cur) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  SVO_REQUIRE(ctx, ref->width == cur->width && ref->height == cur->height && ref->n_levels == cur->n_levels);
  SVO_REQUIRE(ctx, ref->batch >= s->batch && cur->batch >= s->batch);
  s->ref = ref; s->cur = cur;
  return SVO_HIP_OK;
}

int svo_hip_sia_upload_features(svo_hip_sia* s, int slot, int n, const double* px, const double* f,
                                const double* pos, const uint8_t* has_point) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  SVO_REQUIRE(ctx, slot >= 0 && slot < s->batch && n >= 0 && n <= s->max_n);
  SVO_REQUIRE(ctx, n == 0 || (px && f && pos && has_point));
  const size_t o = (size_t)slot * s->max_n;
  if (n > 0) {
    SVO_CHECK_HIP(ctx, hipMemcpyAsync(s->px + o * 2, px, sizeof(double) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    SVO_CHECK_HIP(ctx, hipMemcpyAsync(s->f + o * 3, f, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
    SVO_CHECK_HIP(ctx, hipMemcpyAsync(s->pos + o * 3, pos, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
    SVO_CHECK_HIP(ctx, hipMemcpyAsync(s->has_point + o, has_point, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  }
  s->h_fc[slot].n_feat = n;
  s->fc_dirty = true;
  return SVO_HIP_OK;
}

int svo_hip_sia_upload_poses(svo_hip_sia* s, int slot, const svo_hip_camera* cam, const double T_ref_w[7],
                             const double T_cur_w_init[7]) {
  if (!s || !cam || !T_ref_w || !T_cur_w_init) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  SVO_REQUIRE(ctx, slot >= 0 && slot < s->batch);
  FrameConst& c = s->h_fc[slot];
  c.cam = svo_make_cam(*cam);
  memcpy(c.T_ref_w, T_ref_w, sizeof(double) * 7);
  memcpy(c.T_cur_w_init, T_cur_w_init, sizeof(double) * 7);
  // Frame::pos() = T_f_w_.inverse().translation (I/frame.h:103): same arithmetic as the device
  // se3_inverse, done on the host in plain doubles (no contraction: see Makefile flags)
  {
    const double q[4] = {-T_ref_w[3], -T_ref_w[4], -T_ref_w[5], T_ref_w[6]};
    const double* p = T_ref_w;
    double uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2], q[0] * p[1] - q[1] * p[0]};
    uv[0] = uv[0] + uv[0]; uv[1] = uv[1] + uv[1]; uv[2] = uv[2] + uv[2];
    const double quv[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    c.ref_pos[0] = -((p[0] + q[3] * uv[0]) + quv[0]);
    c.ref_pos[1] = -((p[1] + q[3] * uv[1]) + quv[1]);
    c.ref_pos[2] = -((p[2] + q[3] * uv[2]) + quv[2]);
  }
  s->fc_dirty = true;
  return SVO_HIP_OK;
}

int svo_hip_sia_set_shard(svo_hip_sia* s, int rank, int world) {
  if (!s) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(s->ctx, world >= 1 && rank >= 0 && rank < world);
  s->shard_rank = rank; s->shard_world = world;
  return SVO_HIP_OK;
}

int svo_hip_sia_begin(svo_hip_sia* s, int n_slots, const svo_hip_sia_params* prm) {
  if (!s || !prm) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  SVO_REQUIRE(ctx, s->ref && s->cur);
  SVO_REQUIRE(ctx, n_slots > 0 && n_slots <= s->batch);
  SVO_REQUIRE(ctx, prm->min_level >= 0 && prm->max_level >= prm->min_level && prm->max_level < s->ref->n_levels);
  SVO_REQUIRE(ctx, prm->n_iter >= 0);
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  s->last_mode = 0;             // the step-wise entry points always run the streaming kernels
  int rc = flush_fc(s);
  if (rc != SVO_HIP_OK) return rc;
  s->prm = *prm; s->n_slots = n_slots; s->level = -1; s->begun = true;
  s->chunks = pick_chunks(s, n_slots, s->max_n);
  SVO_CHECK_HIP(ctx, hipMemsetAsync(s->visible, 0, (size_t)n_slots * s->max_n, ctx->stream));
  SVO_CHECK_HIP(ctx, hipMemsetAsync(s->n_pre_count, 0, sizeof(unsigned) * n_slots, ctx->stream));
  hipLaunchKernelGGL(sia_begin_kernel, dim3((n_slots + 63) / 64), dim3(64), 0, ctx->stream, s->fc, s->st, n_slots);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(sia_geometry_kernel, dim3((s->max_n + 255) / 256, n_slots), dim3(256), 0, ctx->stream, s->fc,
                     s->max_n, s->f, s->pos, s->xyz4);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

int svo_hip_sia_level_begin(svo_hip_sia* s, int level) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  if (!s->begun) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_sia_level_begin", "begin not called");
  SVO_REQUIRE(ctx, level >= 0 && level < s->ref->n_levels);
  s->level = level;
  LevelGeom g;
  g.cols = s->ref->width >> level; g.rows = s->ref->height >> level;
  g.ref_off = s->ref->level_offset[level]; g.cur_off = s->cur->level_offset[level];
  dim3 grid((s->max_n + 255) / 256, s->n_slots), block(256);
  const Shard sh = {s->shard_rank, s->shard_world};
  hipEvent_t* ev = next_events(s, s->ev_pre, s->ev_pre_used);
  if (ev) (void)hipEventRecord(ev[0], ctx->stream);
  hipLaunchKernelGGL(sia_precompute_kernel, grid, block, 0, ctx->stream, s->fc, s->st, s->ref->base,
                     s->ref->pyr_bytes, g, level, s->max_n, s->max_tiles, sh, s->px, s->has_point, s->xyz4,
                     s->ref_cache, s->dxc, s->dyc, s->sxyz, s->tile_h, s->visible, s->n_pre_count);
  if (ev) (void)hipEventRecord(ev[1], ctx->stream);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

// One evaluation.  st_in / st_out: the control step of the previous evaluation, whose block partials have been all-reduced
// in place since, is taken at the head of the launch (state read from st_in, stepped state written to st_out).
// pending / out_partial (sharded solve): the exchanged block partials of the previous evaluation and where this launch
// writes its own -- two DIFFERENT buffers: a block that is through writes its row while another block of the frame may still
// be reading the previous evaluation's rows at its head.
static int launch_residual(svo_hip_sia* s, const FrameState* st_in = nullptr, FrameState* st_out = nullptr, const double* pending = nullptr,
                           double* out_partial = nullptr) {
  if (!out_partial) out_partial = s->partial;
  svo_hip_ctx* ctx = s->ctx;
  const int level = s->level;
  LevelGeom g;
  g.cols = s->ref->width >> level; g.rows = s->ref->height >> level;
  g.ref_off = s->ref->level_offset[level]; g.cur_off = s->cur->level_offset[level];
  dim3 grid(s->chunks, s->n_slots), block(256);
  hipEvent_t* ev = next_events(s, s->ev_res, s->ev_res_used);
  if (ev) (void)hipEventRecord(ev[0], ctx->stream);
  const Shard sh = {s->shard_rank, s->shard_world};
  SiaStepFusion fu;
  memset(&fu, 0, sizeof(fu));
  if (st_out) { fu.pending = pending; fu.st_out = st_out; fu.n_iter = s->prm.n_iter; fu.early_stop = s->prm.early_stop; fu.eps = s->prm.eps; }
  if (st_in)
    hipLaunchKernelGGL(sia_residual_kernel<true>, grid, block, 0, ctx->stream, s->fc, st_in, s->cur->base, s->cur->pyr_bytes,
                       g, level, s->max_n, s->max_tiles, s->chunks, sh, s->ref_cache, s->dxc, s->dyc, s->sxyz, s->xyz4,
                       s->tile_h, s->visible, out_partial, fu);
  else
    hipLaunchKernelGGL(sia_residual_kernel<false>, grid, block, 0, ctx->stream, s->fc, s->st, s->cur->base, s->cur->pyr_bytes,
                       g, level, s->max_n, s->max_tiles, s->chunks, sh, s->ref_cache, s->dxc, s->dyc, s->sxyz, s->xyz4,
                       s->tile_h, s->visible, out_partial, fu);
  if (ev) (void)hipEventRecord(ev[1], ctx->stream);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

// the control step as a launch of its own: from the block partials (single device) or from the reduce buffer; the state is
// stepped in place in the home buffer unless st_in names the buffer it currently lives in
static int launch_solve(svo_hip_sia* s, bool from_partials, const FrameState* st_in = nullptr, const double* partials = nullptr) {
  svo_hip_ctx* ctx = s->ctx;
  const FrameState* in = st_in ? st_in : s->st;
  if (from_partials)
    hipLaunchKernelGGL(sia_solve_kernel<true>, dim3(s->n_slots), dim3(64), 0, ctx->stream, in, s->st, partials ? partials : s->partial, s->chunks,
                       s->n_slots, s->level, s->prm.n_iter, s->prm.eps, s->prm.early_stop);
  else
    hipLaunchKernelGGL(sia_solve_kernel<false>, dim3(s->n_slots), dim3(64), 0, ctx->stream, in, s->st, s->reduce, s->chunks,
                       s->n_slots, s->level, s->prm.n_iter, s->prm.eps, s->prm.early_stop);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

int svo_hip_sia_accumulate(svo_hip_sia* s) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  if (nlls_branches(s)) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_sia_accumulate", "Levenberg-Marquardt / robust weights / reference-order chi2 run through svo_hip_sia_run only");
  if (!s->begun || s->level < 0) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_sia_accumulate", "level_begin not called");
  int rc = launch_residual(s);
  if (rc != SVO_HIP_OK) return rc;
  hipLaunchKernelGGL(sia_sum_partials_kernel, dim3(s->n_slots), dim3(RED), 0, ctx->stream, s->st, s->partial,
                     s->chunks, s->reduce, s->n_slots);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

int svo_hip_sia_solve_update(svo_hip_sia* s) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  if (!s->begun || s->level < 0) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_sia_solve_update", "level_begin not called");
  return launch_solve(s, false);
}

int svo_hip_sia_finish(svo_hip_sia* s) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  if (!s->begun) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_sia_finish", "begin not called");
  hipLaunchKernelGGL(sia_finish_kernel, dim3((s->n_slots + 63) / 64), dim3(64), 0, ctx->stream, s->fc, s->st,
                     s->n_pre_count, s->n_slots);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  s->begun = false;
  return SVO_HIP_OK;
}

int svo_hip_sia_run(svo_hip_sia* s, int n_slots, const svo_hip_sia_params* prm) {
  if (!s || !prm) return SVO_HIP_ERR_INVALID;
  if (s->shard_world != 1)     // (svo_hip_sia_run_sharded / the step-wise entry points are the sharded forms)
    return svo_fail(s->ctx, SVO_HIP_ERR_STATE, "svo_hip_sia_run", "a patch shard is set on this solver: the whole solve needs the all-reduce of svo_hip_sia_run_sharded");
  // Levenberg-Marquardt, a robust cost (I/nlls_solver.h:46-48), chi2 in the reference's order: their own driver over the streaming kernels
  if (nlls_branches(s)) return svo_nlls_run(s, n_slots, prm, s->opt_method, s->opt_scale, s->opt_weight);
  if (n_slots > 0 && n_slots <= s->batch && fused_applies(s, n_slots)) return run_fused(s, n_slots, prm);
  s->last_mode = 0;
  int rc = svo_hip_sia_begin(s, n_slots, prm);
  if (rc != SVO_HIP_OK) return rc;
  for (int level = prm->max_level; level >= prm->min_level; --level) {
    if ((rc = svo_hip_sia_level_begin(s, level)) != SVO_HIP_OK) return rc;
    for (int it = 0; it < prm->n_iter; ++it) {
      // single device: the solve gathers the block partials itself (no separate sum kernel)
      if ((rc = launch_residual(s)) != SVO_HIP_OK) return rc;
      if ((rc = launch_solve(s, true)) != SVO_HIP_OK) return rc;
    }
  }
  return svo_hip_sia_finish(s);
}

// BASELINE config C3's variant behind the C-ABI: every frame's patches are split contiguously over the ranks of
// `comm`, every rank reduces its shard to SVO_HIP_REDUCE_DOUBLES doubles per frame, ONE batched all-reduce of
// n_slots x 32 doubles per Gauss-Newton step (RCCL over xGMI, enqueued on the context stream between the two kernels:
// no host round trip), then every rank runs the identical solve on identical sums -- identical decisions, no further
// exchange.  Finished frames contribute zeros (the kernels skip them), so the fixed launch sequence keeps the
// reference's early-exit semantics.
static int sharded_level(svo_hip_sia* s, svo_hip_comm* comm, int n_slots, const svo_hip_sia_params* prm, int level) {
  int rc = svo_hip_sia_level_begin(s, level);
  if (rc != SVO_HIP_OK) return rc;
  // Per Gauss-Newton step ONE launch and ONE collective.  What the ranks exchange is the evaluation kernel's own output,
  // the block partials (n_slots x chunks rows of 32 doubles, all-reduced in place: at these sizes the collective is
  // latency, not bandwidth), and the control step of an evaluation is taken at the HEAD of the next launch: every block
  // of a frame adds the frame's rows in block order and redoes the 6x6 solve on identical numbers, block 0 writes the
  // stepped state to the buffer nobody reads during the launch, and the two state buffers change roles.  Nothing is
  // passed between blocks inside a launch (forming the frame rows in the kernel's tail needed device-scope fences
  // between blocks on different XCDs -- whose L2s are not coherent -- and cost more than the launch it saved).  The step
  // of the level's last evaluation is a launch of its own and brings the state back to the home buffer.
  // The exchange adds the ranks first and the blocks second; the step-wise entry points (block rows summed, then
  // exchanged) round differently in the last bits.  Every rank still ends with the same bits.
  const FrameState* in = s->st;
  FrameState* out = s->st_alt;
  const size_t n_exchange = (size_t)n_slots * s->chunks * SVO_HIP_REDUCE_DOUBLES;
  const double* pending = nullptr;           // exchanged partials of the previous evaluation
  double* mine = s->partial;                 // where this evaluation's partials go (the two buffers change roles per step)
  for (int it = 0; it < prm->n_iter; ++it) {
    rc = launch_residual(s, in, it == 0 ? nullptr : out, pending, mine);
    if (it > 0) { const FrameState* t = in; in = out; out = const_cast<FrameState*>(t); }
    if (rc != SVO_HIP_OK) return rc;
    if ((rc = svo_comm_all_reduce_sum_f64(comm, mine, n_exchange)) != SVO_HIP_OK) return rc;
    pending = mine;
    mine = mine == s->partial ? s->partial_alt : s->partial;
  }
  if (prm->n_iter > 0 && (rc = launch_solve(s, true, in, pending)) != SVO_HIP_OK) return rc;
  return SVO_HIP_OK;
}

static int run_sharded_levels(svo_hip_sia* s, svo_hip_comm* comm, int n_slots, const svo_hip_sia_params* prm, int rank, int world, int kind) {
  svo_hip_ctx* ctx = s->ctx;
  int rc = svo_hip_sia_begin(s, n_slots, prm);
  if (rc != SVO_HIP_OK) return rc;
  // Graph replay (opt-in, RCCL transport only: the host-staged transport blocks the host and cannot be captured): the
  // launch sequence of a level does not depend on the data -- finished frames are skipped inside the kernels -- so it
  // is captured once per configuration and replayed; what it saves is the host's launch time (3 kernels + 1 collective
  // per Gauss-Newton step), not device time.
  const bool graph = s->sharded_graph && kind == 0 && !s->profiling;
  if (graph) {
    // everything the captured nodes have baked in: the communicator (by its never-reused id, not its address), the
    // launch geometry, the buffers and the pyramids
    svo_hip_sia::GraphKey key;
    key.comm_id = svo_comm_id(comm);
    key.n_slots = n_slots; key.n_iter = prm->n_iter; key.early_stop = prm->early_stop; key.max_level = prm->max_level;
    key.min_level = prm->min_level; key.rank = rank; key.world = world; key.chunks = s->chunks; key.eps = prm->eps;
    key.ref_base = s->ref->base; key.cur_base = s->cur->base; key.reduce = s->reduce;
    key.width = s->ref->width; key.height = s->ref->height; key.pyr_bytes = s->ref->pyr_bytes;
    if (!(key == s->graph_key)) { drop_level_graphs(s); s->graph_key = key; }
  }
  for (int level = prm->max_level; level >= prm->min_level; --level) {
    if (!graph) {
      if ((rc = sharded_level(s, comm, n_slots, prm, level)) != SVO_HIP_OK) return rc;
      continue;
    }
    if (!s->level_graph[level]) {
      hipGraph_t g = nullptr;
      SVO_CHECK_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
      rc = sharded_level(s, comm, n_slots, prm, level);
      const hipError_t e = hipStreamEndCapture(ctx->stream, &g);
      if (rc != SVO_HIP_OK || e != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        return rc != SVO_HIP_OK ? rc : svo_fail(ctx, SVO_HIP_ERR_DEVICE, "hipStreamEndCapture", hipGetErrorString(e));
      }
      const hipError_t ei = hipGraphInstantiate(&s->level_graph[level], g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (ei != hipSuccess) { s->level_graph[level] = nullptr; return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "hipGraphInstantiate", hipGetErrorString(ei)); }
    }
    s->level = level;
    SVO_CHECK_HIP(ctx, hipGraphLaunch(s->level_graph[level], ctx->stream));
  }
  return svo_hip_sia_finish(s);
}

int svo_hip_sia_run_sharded(svo_hip_sia* s, svo_hip_comm* comm, int n_slots, const svo_hip_sia_params* prm) {
  if (!s || !comm || !prm) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  // the collective is enqueued on the communicator's stream, the kernels on the solver's: they must be the same one
  SVO_REQUIRE(ctx, svo_comm_ctx(comm) == s->ctx);
  if (nlls_branches(s)) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_sia_run_sharded", "Levenberg-Marquardt / robust weights / reference-order chi2 run through svo_hip_sia_run only");
  int rank = 0, world = 1, kind = 0;
  svo_hip_comm_info(comm, &rank, &world, &kind);
  // the shard of this call only: whatever svo_hip_sia_set_shard left on the object is back in place afterwards, on the
  // error paths too (a later svo_hip_sia_run must not see a shard it never asked for)
  const int old_rank = s->shard_rank, old_world = s->shard_world;
  int rc = svo_hip_sia_set_shard(s, rank, world);
  if (rc == SVO_HIP_OK) rc = run_sharded_levels(s, comm, n_slots, prm, rank, world, kind);
  s->shard_rank = old_rank; s->shard_world = old_world;
  return rc;
}

int svo_hip_sia_set_option(svo_hip_sia* s, int option, int value) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  switch (option) {
    case SVO_HIP_SIA_OPT_MODE: SVO_REQUIRE(ctx, value == SVO_HIP_SIA_MODE_AUTO || value == SVO_HIP_SIA_MODE_STREAM); s->opt_mode = value; break;
    case SVO_HIP_SIA_OPT_WAVES: SVO_REQUIRE(ctx, value == 0 || value == 4 || value == 8); s->opt_waves = value; break;
    case SVO_HIP_SIA_OPT_CHUNKS: SVO_REQUIRE(ctx, value >= 0 && value <= MAX_CHUNKS); s->opt_chunks = value; break;
    case SVO_HIP_SIA_OPT_EXTRA_LDS: SVO_REQUIRE(ctx, value >= -1 && value <= FUSED_EXTRA_TILES); s->opt_extra_lds = value; break;
    case SVO_HIP_SIA_OPT_OLD_TILES: SVO_REQUIRE(ctx, value >= 0 && value <= FUSED_MAX_TPW); s->opt_old_tiles = value; break;
    case SVO_HIP_SIA_OPT_ARITH:
      SVO_REQUIRE(ctx, value == SVO_HIP_SIA_ARITH_EXACT || value == SVO_HIP_SIA_ARITH_FAST || value == SVO_HIP_SIA_ARITH_MOMENTS_F32);
      s->opt_arith = value;
      break;
    case SVO_HIP_SIA_OPT_METHOD:
      SVO_REQUIRE(ctx, value == SVO_HIP_SIA_METHOD_GAUSS_NEWTON || value == SVO_HIP_SIA_METHOD_LEVENBERG_MARQUARDT);
      s->opt_method = value;
      break;
    case SVO_HIP_SIA_OPT_SCALE_ESTIMATOR: SVO_REQUIRE(ctx, value >= SVO_HIP_SIA_SCALE_UNIT && value <= SVO_HIP_SIA_SCALE_NORMAL); s->opt_scale = value; break;
    case SVO_HIP_SIA_OPT_WEIGHT_FUNCTION: SVO_REQUIRE(ctx, value >= SVO_HIP_SIA_WEIGHT_UNIT && value <= SVO_HIP_SIA_WEIGHT_HUBER); s->opt_weight = value; break;
    case SVO_HIP_SIA_OPT_CHI2: SVO_REQUIRE(ctx, value == SVO_HIP_SIA_CHI2_PER_PATCH || value == SVO_HIP_SIA_CHI2_REFERENCE_ORDER); s->opt_chi2 = value; break;
    default: return svo_fail(ctx, SVO_HIP_ERR_INVALID, "svo_hip_sia_set_option", "unknown option");
  }
  return SVO_HIP_OK;
}

int svo_hip_sia_set_sharded_graph(svo_hip_sia* s, int enable) {
  if (!s) return SVO_HIP_ERR_INVALID;
  s->sharded_graph = enable != 0;
  if (!enable) drop_level_graphs(s);
  return SVO_HIP_OK;
}

#ifdef SVO_STAMPS
int svo_hip_sia_debug_x(svo_hip_sia* s, int slot, double* x6) {
  FrameState st;
  int rc = svo_hip_memcpy_d2h(s->ctx, &st, s->st + slot, sizeof(FrameState));
  for (int i = 0; i < 6; ++i) x6[i] = st.x[i];
  for (int i = 0; i < 21; ++i) x6[6 + i] = st.H[i];      // caller passes 27 doubles
  return rc;
}
#endif

int svo_hip_sia_last_run_mode(svo_hip_sia* s, int* mode) {
  if (!s || !mode) return SVO_HIP_ERR_INVALID;
  *mode = s->last_mode;
  return SVO_HIP_OK;
}

int svo_hip_sia_set_profiling(svo_hip_sia* s, int enable) {
  if (!s) return SVO_HIP_ERR_INVALID;
  s->profiling = enable != 0;
  s->ev_res_used = 0; s->ev_pre_used = 0;
  return SVO_HIP_OK;
}

int svo_hip_sia_get_profile(svo_hip_sia* s, double* residual_ms, uint64_t* residual_launches, double* precompute_ms,
                            uint64_t* precompute_launches) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  auto total = [&](std::vector<hipEvent_t>& pool, size_t used, double* ms, uint64_t* cnt) {
    double t = 0.0;
    for (size_t i = 0; i + 1 < used; i += 2) {
      float e = 0.f;
      if (hipEventElapsedTime(&e, pool[i], pool[i + 1]) == hipSuccess) t += e;
    }
    if (ms) *ms = t;
    if (cnt) *cnt = used / 2;
  };
  total(s->ev_res, s->ev_res_used, residual_ms, residual_launches);
  total(s->ev_pre, s->ev_pre_used, precompute_ms, precompute_launches);
  s->ev_res_used = 0; s->ev_pre_used = 0;
  return SVO_HIP_OK;
}

int svo_hip_sia_reduce_buffer(svo_hip_sia* s, void** dev_ptr, size_t* n_doubles) {
  if (!s || !dev_ptr) return SVO_HIP_ERR_INVALID;
  *dev_ptr = s->reduce;
  if (n_doubles) *n_doubles = (size_t)s->batch * RED;
  return SVO_HIP_OK;
}

int svo_hip_sia_set_reduce_buffer(svo_hip_sia* s, void* dev_ptr) {
  if (!s) return SVO_HIP_ERR_INVALID;
  s->reduce = dev_ptr ? (double*)dev_ptr : s->reduce_own;
  return SVO_HIP_OK;
}

static void fill_result(const FrameState& st, svo_hip_sia_result* out) {
  memcpy(out->T_cur_w, st.T_cur_w, sizeof(double) * 7);
  out->n_tracked = st.n_meas / PATCH_AREA;
  memcpy(out->H, st.H, sizeof(double) * 36);
  out->chi2 = st.chi2;
  out->stop = st.stop;
  for (int i = 0; i < SVO_HIP_MAX_LEVELS; ++i) out->iters[i] = st.iters[i];
  out->n_precompute_patches = st.n_pre;
  out->n_residual_patches = st.n_res;
}

int svo_hip_sia_download(svo_hip_sia* s, int slot, svo_hip_sia_result* out) {
  if (!s || !out) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  SVO_REQUIRE(ctx, slot >= 0 && slot < s->batch);
  FrameState st;
  int rc = svo_hip_memcpy_d2h(ctx, &st, s->st + slot, sizeof(FrameState));
  if (rc != SVO_HIP_OK) return rc;
  fill_result(st, out);
  return SVO_HIP_OK;
}

int svo_hip_sia_download_all(svo_hip_sia* s, int n_slots, svo_hip_sia_result* out) {
  if (!s || !out) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  SVO_REQUIRE(ctx, n_slots > 0 && n_slots <= s->batch);
  FrameState* h = new (std::nothrow) FrameState[n_slots];
  if (!h) return SVO_HIP_ERR_NOMEM;
  int rc = svo_hip_memcpy_d2h(ctx, h, s->st, sizeof(FrameState) * n_slots);
  if (rc == SVO_HIP_OK)
    for (int i = 0; i < n_slots; ++i) fill_result(h[i], out + i);
  delete[] h;
  return rc;
}

int svo_hip_sia_download_fused_patches(svo_hip_sia* s, int slot, int level, float* ref_patch, float* dx, float* dy, uint8_t* valid) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  SVO_REQUIRE(ctx, slot >= 0 && slot < s->batch && s->ref && level >= 0 && level < s->ref->n_levels);
  // the dump goes through the streaming path's cache arrays: not in the middle of a step-wise solve that is using them
  if (s->begun)
    return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_sia_download_fused_patches", "a step-wise solve is in progress (svo_hip_sia_finish it first)");
  int rc = flush_fc(s);
  if (rc != SVO_HIP_OK) return rc;
  const int n = s->h_fc[slot].n_feat;
  if (n <= 0) return SVO_HIP_OK;
  s->last_mode = 1;          // ... and what svo_hip_sia_download_caches would return from now on is this dump, not a run's caches: stale
  LevelGeom g;
  g.cols = s->ref->width >> level; g.rows = s->ref->height >> level;
  g.ref_off = s->ref->level_offset[level]; g.cur_off = 0;
  hipLaunchKernelGGL(sia_fused_patch_dump_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, s->fc, slot, s->ref->base, s->ref->pyr_bytes, g,
                     level, s->max_n, s->px, s->has_point, s->ref_cache, s->dxc, s->dyc, s->visible);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  const size_t o = (size_t)slot * s->max_n;
  if (ref_patch && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, ref_patch, s->ref_cache + o * 4, sizeof(float) * 16 * n);
  if (dx && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, dx, s->dxc + o * 4, sizeof(float) * 16 * n);
  if (dy && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, dy, s->dyc + o * 4, sizeof(float) * 16 * n);
  if (valid && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, valid, s->visible + o, (size_t)n);
  return rc;
}

int svo_hip_sia_download_caches(svo_hip_sia* s, int slot, float* ref_patch, float* dx, float* dy, uint8_t* visible) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  SVO_REQUIRE(ctx, slot >= 0 && slot < s->batch);
  // the per-pixel caches exist only in the streaming implementation: after a fused run they would be stale
  if (s->last_mode == 1)
    return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_sia_download_caches",
                    "the last run used the fused kernel, which keeps no per-pixel caches in memory (use the step-wise entry points or SVO_HIP_SIA_OPT_MODE = SVO_HIP_SIA_MODE_STREAM)");
  const size_t o = (size_t)slot * s->max_n;
  const int n = s->h_fc[slot].n_feat;
  int rc = SVO_HIP_OK;
  if (ref_patch && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, ref_patch, s->ref_cache + o * 4, sizeof(float) * 16 * n);
  if (dx && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, dx, s->dxc + o * 4, sizeof(float) * 16 * n);
  if (dy && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, dy, s->dyc + o * 4, sizeof(float) * 16 * n);
  if (visible && rc == SVO_HIP_OK) {
    rc = svo_hip_memcpy_d2h(ctx, visible, s->visible + o, (size_t)n);
    for (int i = 0; i < n; ++i) visible[i] &= 1;     // bit 0 = visible_fts_, bit 1 is internal
  }
  return rc;
}

}  // extern "C"

// svo_nlls.hip works on the streaming solver's buffers (see svo_internal.h)
int svo_sia_view_get(svo_hip_sia* s, svo_sia_view* v) {
  if (!s || !v || !s->begun || s->level < 0) return SVO_HIP_ERR_STATE;
  v->ctx = s->ctx; v->batch = s->batch; v->max_n = s->max_n; v->n_slots = s->n_slots; v->level = s->level; v->chunks = s->chunks;
  v->cols = s->cur->width >> s->level; v->rows = s->cur->height >> s->level;
  v->fc = s->fc; v->st = s->st;
  v->ref_cache = s->ref_cache; v->dxc = s->dxc; v->dyc = s->dyc; v->xyz4 = s->xyz4; v->flags = s->visible; v->partial = s->partial;
  v->cur_level = s->cur->base + s->cur->level_offset[s->level];
  v->pyr_bytes = s->cur->pyr_bytes;
  return SVO_HIP_OK;
}

svo_nlls_ext** svo_sia_nlls_slot(svo_hip_sia* s) { return &s->nlls; }
