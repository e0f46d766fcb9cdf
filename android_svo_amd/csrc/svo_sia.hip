// svo_sia.hip -- batched sparse image alignment (svo::SparseImgAlign) on gfx950.
//
// Replaces, per frame pair, SparseImgAlign::run (S/sparse_img_align.cpp:51-92) with
// precomputeReferencePatches (:105-178), computeResiduals (:184-286), solve/update
// (:291-308) driven by NLLSSolver::optimizeGaussNewton (I/nlls_solver_impl.hpp:25-100).
//
// Design (MI355X-first, not the reference's loop structure):
//   * B independent frame pairs are solved at once; the whole coarse-to-fine loop is a
//     fixed sequence of kernels on one stream with the Gauss-Newton control state
//     (model, rollback copy, chi2, stop, iteration) resident in HBM -- no host round trip,
//     data-dependent early exits become per-frame "done" flags that later kernels test.
//   * 16 lanes own one 4x4 patch (lane = pixel): a wave64 handles 4 patches, cache rows
//     are 64 B per patch so every cache access of a wave is one contiguous 256 B segment.
//   * Inverse-compositional structure is exploited: the per-pixel Jacobian is
//     J = dx*A + dy*B with A,B = rows of the 2x6 projection Jacobian (per patch) times
//     fx/2^L, so  sum_px J J^T = sxx AA^T + sxy (AB^T+BA^T) + syy BB^T  with per-patch
//     constants and  sum_px J r = A sum(dx r) + B sum(dy r).  Instead of streaming the
//     reference's 768 B/patch fp64 Jacobian cache every iteration we keep 3 f32 per pixel
//     (ref, dx, dy) + a 128 B per-patch record, and only two 16-lane reductions per patch
//     are needed per iteration.  H/Jres accumulate in fp64; chi2 like the reference in f32
//     per patch then fp64 across patches.
//   * Reductions: 16-lane xor-shuffles inside the patch, then per-lane fp64 accumulators
//     over the block's patches, one cross-wave LDS step, one partial row per block; the
//     per-frame sum over blocks is done in fixed order (bitwise reproducible, no atomics).
#include <vector>

#include "svo_internal.h"

using namespace svo_dev;

namespace {

constexpr int PATCH_AREA = 16;
constexpr int RED = SVO_HIP_REDUCE_DOUBLES;     // 32
constexpr int MAX_CHUNKS = 128;

// per-frame constants
struct FrameConst {
  Cam cam;
  double T_ref_w[7];
  double T_cur_w_init[7];
  double ref_pos[3];
  int n_feat;
  int pad;
};

// per-frame Gauss-Newton state (I/nlls_solver.h:51-60,96-111)
struct FrameState {
  double model[7];        // T_cur_from_ref
  double old_model[7];
  double chi2;            // chi2_
  double H[36];
  double Jres[6];
  double x[6];
  unsigned long long n_meas;
  unsigned long long n_pre, n_res;
  int stop;               // stop_ (persists across levels)
  int iter;               // iter_ of the current level
  int level_done;         // the level's GN loop has exited
  int empty;              // ref frame has no features: run() returns 0 and leaves the pose alone (:55-59)
  int iters[SVO_HIP_MAX_LEVELS];
  double T_cur_w[7];      // result
};

struct LevelGeom {
  int cols, rows;
  size_t ref_off, cur_off;   // byte offset of the level inside a pyramid
};

// upper-triangle (row-major) index -> (i, j)
__device__ __constant__ int8_t kTriI[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
__device__ __constant__ int8_t kTriJ[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};

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

// One level's precomputeReferencePatches.  grid = (ceil(max_n/16), n_slots), block = 256.
// Algorithmic bytes per patch: 49 B footprint + 16+24+24 B feature -> 64 B + 768 B caches;
// physical: 7x7 u8 gather + 65 B feature -> 192 B pixel caches + 128 B record + 32 B xyz.
__global__ __launch_bounds__(256) void sia_precompute_kernel(
    const FrameConst* __restrict__ fc, FrameState* __restrict__ st, const uint8_t* __restrict__ ref_base,
    size_t pyr_bytes, LevelGeom g, int level, int max_n, int shard_rank, int shard_world,
    const double* __restrict__ px, const double* __restrict__ f, const double* __restrict__ pos,
    const uint8_t* __restrict__ has_point, float* __restrict__ ref_cache, float* __restrict__ dxc,
    float* __restrict__ dyc, double* __restrict__ rec, double* __restrict__ xyz, uint8_t* __restrict__ visible,
    unsigned int* __restrict__ n_pre_count) {
  const int b = blockIdx.y;
  const int lane16 = threadIdx.x & 15;
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
  const FrameConst& c = fc[b];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // per-level solver reset: optimizeGaussNewton saves the model for rollback (:33) and
    // restarts iter_; stop_ and chi2_ deliberately persist
    FrameState& s = st[b];
    for (int k = 0; k < 7; ++k) s.old_model[k] = s.model[k];
    s.iter = 0;
    s.level_done = s.empty;
  }
  const int n = c.n_feat;
  if (i >= n) return;
  const int lo = (int)(((long long)n * shard_rank) / shard_world);
  const int hi = (int)(((long long)n * (shard_rank + 1)) / shard_world);
  if (i < lo || i >= hi) return;

  const size_t fi = (size_t)b * max_n + i;
  const int border = 3;
  const float scale = 1.0f / (1 << level);
  const float u_ref = (float)(px[2 * fi] * scale);
  const float v_ref = (float)(px[2 * fi + 1] * scale);
  const int u_ref_i = (int)floorf(u_ref);
  const int v_ref_i = (int)floorf(v_ref);
  double* r = rec + fi * 16;
  if (!has_point[fi] || u_ref_i - border < 0 || v_ref_i - border < 0 || u_ref_i + border >= g.cols ||
      v_ref_i + border >= g.rows) {
    r[lane16] = 0.0;              // jacobian_cache_.setZero() (:76); ref patch cache and visibility stay as they were
    return;
  }
  if (lane16 == 0) {
    visible[fi] = 1;              // sticky across levels (:67,128)
    atomicAdd(&n_pre_count[b], 1u);
  }

  const double dxp = pos[3 * fi] - c.ref_pos[0];
  const double dyp = pos[3 * fi + 1] - c.ref_pos[1];
  const double dzp = pos[3 * fi + 2] - c.ref_pos[2];
  const double depth = sqrt(dxp * dxp + dyp * dyp + dzp * dzp);
  const double xyz_ref[3] = {f[3 * fi] * depth, f[3 * fi + 1] * depth, f[3 * fi + 2] * depth};
  double fj[12];
  jacobian_xyz2uv(xyz_ref, fj);

  const float subpix_u = u_ref - u_ref_i;
  const float subpix_v = v_ref - v_ref_i;
  const float w_tl = (float)((1.0 - subpix_u) * (1.0 - subpix_v));
  const float w_tr = (float)(subpix_u * (1.0 - subpix_v));
  const float w_bl = (float)((1.0 - subpix_u) * subpix_v);
  const float w_br = subpix_u * subpix_v;

  const int stride = g.cols;
  const uint8_t* img = ref_base + (size_t)b * pyr_bytes + g.ref_off;
  const int y = lane16 >> 2, x = lane16 & 3;
  const uint8_t* p = img + (v_ref_i + y - 2) * stride + (u_ref_i - 2) + x;
  // 3x3 neighbourhood + one more row/column: 12 distinct taps
  const float p_m10 = p[-stride], p_m11 = p[1 - stride];
  const float p_0m1 = p[-1], p_00 = p[0], p_01 = p[1], p_02 = p[2];
  const float p_1m1 = p[stride - 1], p_10 = p[stride], p_11 = p[stride + 1], p_12 = p[stride + 2];
  const float p_20 = p[2 * stride], p_21 = p[2 * stride + 1];
  const float val = w_tl * p_00 + w_tr * p_01 + w_bl * p_10 + w_br * p_11;
  const float dx = 0.5f * ((w_tl * p_01 + w_tr * p_02 + w_bl * p_11 + w_br * p_12) -
                           (w_tl * p_0m1 + w_tr * p_00 + w_bl * p_1m1 + w_br * p_10));
  const float dy = 0.5f * ((w_tl * p_10 + w_tr * p_11 + w_bl * p_20 + w_br * p_21) -
                           (w_tl * p_m10 + w_tr * p_m11 + w_bl * p_00 + w_br * p_01));
  ref_cache[fi * 16 + lane16] = val;
  dxc[fi * 16 + lane16] = dx;
  dyc[fi * 16 + lane16] = dy;

  const double ddx = (double)dx, ddy = (double)dy;
  const double sxx = group_sum<16>(ddx * ddx);
  const double sxy = group_sum<16>(ddx * ddy);
  const double syy = group_sum<16>(ddy * ddy);
  const double jscale = fabs(c.cam.fx) / (1 << level);   // errorMultiplier2() / 2^L (:113,172-173)
  double out;
  if (lane16 < 12) out = fj[lane16] * jscale;            // A = row 0, B = row 1 of the 2x6 Jacobian
  else if (lane16 == 12) out = sxx;
  else if (lane16 == 13) out = sxy;
  else if (lane16 == 14) out = syy;
  else out = 0.0;
  r[lane16] = out;
  if (lane16 < 3) xyz[fi * 4 + lane16] = xyz_ref[lane16];
}

// One computeResiduals(linearize=true) evaluation for every live frame.
// grid = (chunks, n_slots), block = 256 (16 patches per pass).  Output: one partial row of RED
// doubles per block.
__global__ __launch_bounds__(256) void sia_residual_kernel(
    const FrameConst* __restrict__ fc, const FrameState* __restrict__ st, const uint8_t* __restrict__ cur_base,
    size_t pyr_bytes, LevelGeom g, int level, int max_n, int chunks, int shard_rank, int shard_world,
    const float* __restrict__ ref_cache, const float* __restrict__ dxc, const float* __restrict__ dyc,
    const double* __restrict__ rec, const double* __restrict__ xyz, const uint8_t* __restrict__ visible,
    double* __restrict__ partial) {
  const int b = blockIdx.y;
  const int chunk = blockIdx.x;
  const FrameState& s = st[b];
  double* out_row = partial + ((size_t)b * chunks + chunk) * RED;
  if (s.level_done) return;          // this frame's GN loop already exited at this level
  const FrameConst& c = fc[b];
  const Cam cam = c.cam;
  double T[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) T[k] = s.model[k];

  const int n = c.n_feat;
  const int lo = (int)(((long long)n * shard_rank) / shard_world);
  const int hi = (int)(((long long)n * (shard_rank + 1)) / shard_world);
  const int per = (((hi - lo) + chunks - 1) / chunks + 15) & ~15;
  const int c_lo = lo + chunk * per;
  const int c_hi = min(hi, c_lo + per);

  const int lane16 = threadIdx.x & 15;
  const int grp = threadIdx.x >> 4;
  const int border = 3;
  const float scale = 1.0f / (1 << level);
  const int stride = g.cols;
  const uint8_t* img = cur_base + (size_t)b * pyr_bytes + g.cur_off;
  const int py = lane16 >> 2, pxx = lane16 & 3;

  // this lane's two reduction items: e0 = lane16 (H entries 0..15), e1 = 16 + lane16:
  //   16..20 -> H entries, 21..26 -> Jres[0..5], 27 -> chi2 sum, 28 -> n_meas
  const int e1 = 16 + lane16;
  const int i0 = kTriI[lane16], j0 = kTriJ[lane16];
  const int i1 = e1 < 21 ? kTriI[e1] : 0, j1 = e1 < 21 ? kTriJ[e1] : 0;
  const int jr = e1 - 21;            // Jres index when 0 <= jr < 6
  double acc0 = 0.0, acc1 = 0.0;

  for (int base = c_lo; base < c_hi; base += 16) {
    const int i = base + grp;
    if (i >= c_hi) continue;
    const size_t fi = (size_t)b * max_n + i;
    if (!visible[fi]) continue;
    const double xyz_ref[3] = {xyz[fi * 4], xyz[fi * 4 + 1], xyz[fi * 4 + 2]};
    double xyz_cur[3], pxd[2];
    se3_act(T, xyz_ref, xyz_cur);
    world2cam(cam, xyz_cur, pxd);
    const float u_cur = (float)pxd[0] * scale;
    const float v_cur = (float)pxd[1] * scale;
    const int u_cur_i = (int)floorf(u_cur);
    const int v_cur_i = (int)floorf(v_cur);
    // NaN projections compare false everywhere in the reference and would read out of bounds
    // there; here they are treated as outside the image.
    if (!(u_cur_i >= 0 && v_cur_i >= 0 && u_cur_i - border >= 0 && v_cur_i - border >= 0 &&
          u_cur_i + border < g.cols && v_cur_i + border < g.rows) || u_cur != u_cur || v_cur != v_cur)
      continue;
    const float subpix_u = u_cur - u_cur_i;
    const float subpix_v = v_cur - v_cur_i;
    const float w_tl = (float)((1.0 - subpix_u) * (1.0 - subpix_v));
    const float w_tr = (float)(subpix_u * (1.0 - subpix_v));
    const float w_bl = (float)((1.0 - subpix_u) * subpix_v);
    const float w_br = subpix_u * subpix_v;
    const uint8_t* p = img + (v_cur_i + py - 2) * stride + (u_cur_i - 2) + pxx;
    const float intensity = w_tl * p[0] + w_tr * p[1] + w_bl * p[stride] + w_br * p[stride + 1];
    const float res = intensity - ref_cache[fi * 16 + lane16];
    const double dres = (double)res;
    const float chi2p = group_sum<16>(res * res);
    const double sdx = group_sum<16>((double)dxc[fi * 16 + lane16] * dres);
    const double sdy = group_sum<16>((double)dyc[fi * 16 + lane16] * dres);

    const double* r = rec + fi * 16;
    const double sxx = r[12], sxy = r[13], syy = r[14];
    {
      const double Ai = r[i0], Aj = r[j0], Bi = r[6 + i0], Bj = r[6 + j0];
      acc0 += sxx * (Ai * Aj) + sxy * (Ai * Bj + Bi * Aj) + syy * (Bi * Bj);
    }
    if (e1 < 21) {
      const double Ai = r[i1], Aj = r[j1], Bi = r[6 + i1], Bj = r[6 + j1];
      acc1 += sxx * (Ai * Aj) + sxy * (Ai * Bj + Bi * Aj) + syy * (Bi * Bj);
    } else if (jr < 6) {
      acc1 -= r[jr] * sdx + r[6 + jr] * sdy;           // Jres_ -= J*res (:273)
    } else if (e1 == 27) {
      acc1 += (double)chi2p;
    } else if (e1 == 28) {
      acc1 += 16.0;
    }
  }

  // reduce the 16 patch groups of the block: 4 groups per wave by shuffles, then 4 waves via LDS
  acc0 += __shfl_xor(acc0, 16, 64); acc0 += __shfl_xor(acc0, 32, 64);
  acc1 += __shfl_xor(acc1, 16, 64); acc1 += __shfl_xor(acc1, 32, 64);
  __shared__ double red[4][32];
  const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
  if (wl < 16) { red[wave][wl] = acc0; red[wave][16 + wl] = acc1; }
  __syncthreads();
  if (threadIdx.x < 32) {
    double v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (threadIdx.x >= 29) v = 0.0;
    out_row[threadIdx.x] = v;
  }
}

// Sum the block partials of each frame in fixed order -> reduce buffer [n_slots][RED].
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

// One Gauss-Newton control step per frame (one thread per frame):
// I/nlls_solver_impl.hpp:35-99 with solve()/update() of S/sparse_img_align.cpp:291-308.
__global__ void sia_solve_kernel(FrameState* __restrict__ st, const double* __restrict__ reduce, int n_slots,
                                 int level, int n_iter, double eps, int early_stop) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_slots) return;
  FrameState& s = st[b];
  if (s.level_done) return;
  const double* r = reduce + (size_t)b * RED;
  double H[36], Jres[6], x[6];
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) { H[i * 6 + j] = r[k]; H[j * 6 + i] = r[k]; ++k; }
  for (int i = 0; i < 6; ++i) Jres[i] = r[21 + i];
  const double chi2_sum = r[27];
  const unsigned long long n_meas = (unsigned long long)(r[28] + 0.5);
  // computeResiduals returns float chi2 / size_t n_meas evaluated in float (:285)
  const double new_chi2 = (double)((float)chi2_sum / (float)n_meas);
  for (int i = 0; i < 36; ++i) s.H[i] = H[i];
  for (int i = 0; i < 6; ++i) s.Jres[i] = Jres[i];
  s.n_meas = n_meas;
  s.n_res += n_meas / 16;
  s.iters[level] += 1;

  ldlt6_solve(H, Jres, x);
  for (int i = 0; i < 6; ++i) s.x[i] = x[i];
  if (x[0] != x[0]) s.stop = 1;                               // NaN -> stop_ (:52-59)
  const int iter = s.iter;
  if ((early_stop && iter > 0 && new_chi2 > s.chi2) || s.stop) {
    for (int i = 0; i < 7; ++i) s.model[i] = s.old_model[i];  // rollback (:72)
    s.level_done = 1;
    return;
  }
  double mx[6], dT[7], nm[7];
  for (int i = 0; i < 6; ++i) mx[i] = -x[i];
  se3_exp(mx, dT);
  se3_mul(s.model, dT, nm);                                   // T_new = T_old * exp(-x) (:307)
  for (int i = 0; i < 7; ++i) { s.old_model[i] = s.model[i]; s.model[i] = nm[i]; }
  s.chi2 = new_chi2;
  double mxn = -1;
  for (int i = 0; i < 6; ++i) { double a = fabs(x[i]); if (a > mxn) mxn = a; }
  if (early_stop && mxn <= eps) s.level_done = 1;             // :97-98
  s.iter = iter + 1;
  if (s.iter >= n_iter) s.level_done = 1;
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

}  // namespace

struct svo_hip_sia {
  svo_hip_ctx* ctx = nullptr;
  int batch = 0, max_n = 0;
  const svo_hip_pyramid* ref = nullptr;
  const svo_hip_pyramid* cur = nullptr;
  // device buffers
  FrameConst* fc = nullptr;
  FrameState* st = nullptr;
  double *px = nullptr, *f = nullptr, *pos = nullptr;
  uint8_t *has_point = nullptr, *visible = nullptr;
  float *ref_cache = nullptr, *dxc = nullptr, *dyc = nullptr;
  double *rec = nullptr, *xyz = nullptr;
  double* partial = nullptr;
  double* reduce_own = nullptr;
  double* reduce = nullptr;
  unsigned int* n_pre_count = nullptr;
  // host mirrors
  FrameConst* h_fc = nullptr;
  bool fc_dirty = true;
  int shard_rank = 0, shard_world = 1;
  // stepwise state
  svo_hip_sia_params prm{};
  int n_slots = 0, level = -1, chunks = 1;
  bool begun = false;
  // optional HIP-event timing of the two heavy kernels, on the context stream
  bool profiling = false;
  std::vector<hipEvent_t> ev_res, ev_pre;     // start/stop pairs
  size_t ev_res_used = 0, ev_pre_used = 0;
};

namespace {

template <typename T>
int dev_alloc(svo_hip_ctx* ctx, T** p, size_t count) {
  void* d = nullptr;
  int rc = svo_hip_malloc(ctx, &d, count * sizeof(T));
  *p = (T*)d;
  return rc;
}

int pick_chunks(int n_slots, int max_n) {
  int c = (2048 + n_slots - 1) / n_slots;
  int cap = (max_n + 15) / 16;
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

}  // namespace

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
  A(dev_alloc(ctx, &s->fc, batch)); A(dev_alloc(ctx, &s->st, batch));
  A(dev_alloc(ctx, &s->px, bn * 2)); A(dev_alloc(ctx, &s->f, bn * 3)); A(dev_alloc(ctx, &s->pos, bn * 3));
  A(dev_alloc(ctx, &s->has_point, bn)); A(dev_alloc(ctx, &s->visible, bn));
  A(dev_alloc(ctx, &s->ref_cache, bn * 16)); A(dev_alloc(ctx, &s->dxc, bn * 16)); A(dev_alloc(ctx, &s->dyc, bn * 16));
  A(dev_alloc(ctx, &s->rec, bn * 16)); A(dev_alloc(ctx, &s->xyz, bn * 4));
  A(dev_alloc(ctx, &s->partial, (size_t)batch * MAX_CHUNKS * RED));
  A(dev_alloc(ctx, &s->reduce_own, (size_t)batch * RED));
  A(dev_alloc(ctx, &s->n_pre_count, batch));
  s->h_fc = new (std::nothrow) FrameConst[batch];
  if (rc != SVO_HIP_OK || !s->h_fc) { svo_hip_sia_destroy(s); return rc != SVO_HIP_OK ? rc : SVO_HIP_ERR_NOMEM; }
  memset(s->h_fc, 0, sizeof(FrameConst) * batch);
  s->reduce = s->reduce_own;
  (void)hipMemsetAsync(s->has_point, 0, bn, ctx->stream);
  (void)hipMemsetAsync(s->ref_cache, 0, bn * 16 * sizeof(float), ctx->stream);
  (void)hipMemsetAsync(s->st, 0, sizeof(FrameState) * batch, ctx->stream);
  *out = s;
  return SVO_HIP_OK;
}

int svo_hip_sia_destroy(svo_hip_sia* s) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  (void)hipStreamSynchronize(ctx->stream);
  void* ptrs[] = {s->fc, s->st, s->px, s->f, s->pos, s->has_point, s->visible, s->ref_cache, s->dxc, s->dyc,
                  s->rec, s->xyz, s->partial, s->reduce_own, s->n_pre_count};
  for (void* p : ptrs) if (p) (void)hipFree(p);
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
  int rc = flush_fc(s);
  if (rc != SVO_HIP_OK) return rc;
  s->prm = *prm; s->n_slots = n_slots; s->level = -1; s->begun = true;
  s->chunks = pick_chunks(n_slots, s->max_n);
  SVO_CHECK_HIP(ctx, hipMemsetAsync(s->visible, 0, (size_t)n_slots * s->max_n, ctx->stream));
  SVO_CHECK_HIP(ctx, hipMemsetAsync(s->n_pre_count, 0, sizeof(unsigned) * n_slots, ctx->stream));
  hipLaunchKernelGGL(sia_begin_kernel, dim3((n_slots + 63) / 64), dim3(64), 0, ctx->stream, s->fc, s->st, n_slots);
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
  dim3 grid((s->max_n + 15) / 16, s->n_slots), block(256);
  hipEvent_t* ev = next_events(s, s->ev_pre, s->ev_pre_used);
  if (ev) (void)hipEventRecord(ev[0], ctx->stream);
  hipLaunchKernelGGL(sia_precompute_kernel, grid, block, 0, ctx->stream, s->fc, s->st, s->ref->base,
                     s->ref->pyr_bytes, g, level, s->max_n, s->shard_rank, s->shard_world, s->px, s->f, s->pos,
                     s->has_point, s->ref_cache, s->dxc, s->dyc, s->rec, s->xyz, s->visible, s->n_pre_count);
  if (ev) (void)hipEventRecord(ev[1], ctx->stream);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

int svo_hip_sia_accumulate(svo_hip_sia* s) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  if (!s->begun || s->level < 0) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_sia_accumulate", "level_begin not called");
  const int level = s->level;
  LevelGeom g;
  g.cols = s->ref->width >> level; g.rows = s->ref->height >> level;
  g.ref_off = s->ref->level_offset[level]; g.cur_off = s->cur->level_offset[level];
  dim3 grid(s->chunks, s->n_slots), block(256);
  hipEvent_t* ev = next_events(s, s->ev_res, s->ev_res_used);
  if (ev) (void)hipEventRecord(ev[0], ctx->stream);
  hipLaunchKernelGGL(sia_residual_kernel, grid, block, 0, ctx->stream, s->fc, s->st, s->cur->base, s->cur->pyr_bytes,
                     g, level, s->max_n, s->chunks, s->shard_rank, s->shard_world, s->ref_cache, s->dxc, s->dyc,
                     s->rec, s->xyz, s->visible, s->partial);
  if (ev) (void)hipEventRecord(ev[1], ctx->stream);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(sia_sum_partials_kernel, dim3(s->n_slots), dim3(RED), 0, ctx->stream, s->st, s->partial,
                     s->chunks, s->reduce, s->n_slots);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

int svo_hip_sia_solve_update(svo_hip_sia* s) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  if (!s->begun || s->level < 0) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_sia_solve_update", "level_begin not called");
  hipLaunchKernelGGL(sia_solve_kernel, dim3((s->n_slots + 63) / 64), dim3(64), 0, ctx->stream, s->st, s->reduce,
                     s->n_slots, s->level, s->prm.n_iter, s->prm.eps, s->prm.early_stop);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
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
  int rc = svo_hip_sia_begin(s, n_slots, prm);
  if (rc != SVO_HIP_OK) return rc;
  for (int level = prm->max_level; level >= prm->min_level; --level) {
    if ((rc = svo_hip_sia_level_begin(s, level)) != SVO_HIP_OK) return rc;
    for (int it = 0; it < prm->n_iter; ++it) {
      if ((rc = svo_hip_sia_accumulate(s)) != SVO_HIP_OK) return rc;
      if ((rc = svo_hip_sia_solve_update(s)) != SVO_HIP_OK) return rc;
    }
  }
  return svo_hip_sia_finish(s);
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

int svo_hip_sia_download_caches(svo_hip_sia* s, int slot, float* ref_patch, float* dx, float* dy, uint8_t* visible) {
  if (!s) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = s->ctx;
  SVO_REQUIRE(ctx, slot >= 0 && slot < s->batch);
  const size_t o = (size_t)slot * s->max_n;
  const int n = s->h_fc[slot].n_feat;
  int rc = SVO_HIP_OK;
  if (ref_patch && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, ref_patch, s->ref_cache + o * 16, sizeof(float) * 16 * n);
  if (dx && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, dx, s->dxc + o * 16, sizeof(float) * 16 * n);
  if (dy && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, dy, s->dyc + o * 16, sizeof(float) * 16 * n);
  if (visible && rc == SVO_HIP_OK) rc = svo_hip_memcpy_d2h(ctx, visible, s->visible + o, (size_t)n);
  return rc;
}

}  // extern "C"
