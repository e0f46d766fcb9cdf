// svo_internal.h -- host-side objects behind the opaque C-ABI handles.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/svo_hip.h"
#include "svo_device_math.h"

// One block of the context's seed-batch pool (svo_depth.hip): the device arrays and the page-locked, device-mapped event
// block of a seed batch of up to `cap` seeds.
struct svo_seed_block {
  char* dev = nullptr;
  char* host = nullptr;
  char* host_dev = nullptr;
  size_t dev_bytes = 0, host_bytes = 0;
  int cap = 0;
};

struct svo_hip_ctx {
  int device = 0;
  int n_cu = 0;                     // compute units of the device
  hipStream_t stream = nullptr;
  bool own_stream = false;
  void* scratch = nullptr;          // grow-only device workspace (depth-filter stage records)
  size_t scratch_bytes = 0;
  void* staging = nullptr;          // grow-only device staging area of the host-buffer convenience entry points
  size_t staging_bytes = 0;
  void* host_staging = nullptr;     // grow-only page-locked host mirror of it: one transfer each way per call
  size_t host_staging_bytes = 0;
  // svo_hip_pyramid_upload leaves ONE transfer out of the host staging area in flight when it returns (it does not synchronise):
  // the next user of the area waits for this event first
  hipEvent_t host_staging_ev = nullptr;
  bool host_staging_in_flight = false;
  // stage timing of the depth-filter pass (svo_hip_df_set_profiling): events around geometry / search / align / finalize
  bool df_profile = false;
  int df_small_max = 8192;             // passes of at most this many seed records take the two-launch form (svo_depth.hip)
  bool df_ev_recorded = false;
  hipEvent_t df_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // Seed batches are created and destroyed at keyframe rate on the depth-filter thread (S/depth_filter.cpp:129-151,256-261);
  // hipFree synchronises the whole DEVICE, i.e. it would stall the tracking thread's stream too (SURVEY 8(b) "Threading": no
  // implicit device synchronisation).  So a destroyed batch returns its block here and a new one takes a free block of its
  // capacity class (powers of two from 256 seeds); blocks are freed with the context only.
  std::vector<svo_seed_block> seed_pool;      // free blocks
  int seed_blocks_in_use = 0;
  // allocator calls (hipMalloc / hipHostMalloc, hipFree / hipHostFree) made on behalf of this context after its creation:
  // seed-batch blocks, scratch, staging areas.  Flat once the working set has been seen (svo_hip_ctx_info).
  unsigned long long n_allocs = 0, n_frees = 0;
  char err[512] = {0};
};

struct svo_hip_pyramid {
  svo_hip_ctx* ctx = nullptr;
  int width = 0, height = 0, n_levels = 0, batch = 0;
  size_t level_offset[SVO_HIP_MAX_LEVELS + 1] = {0};   // byte offset of each level inside one pyramid
  size_t pyr_bytes = 0;                                // bytes of one pyramid (16-byte aligned levels)
  uint8_t* base = nullptr;                             // device
};

inline int svo_fail(svo_hip_ctx* ctx, int code, const char* what, const char* detail) {
  if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, detail ? detail : "");
  return code;
}

#define SVO_CHECK_HIP(ctx, expr)                                                        \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) return svo_fail((ctx), SVO_HIP_ERR_DEVICE, #expr, hipGetErrorString(e_)); \
  } while (0)

#define SVO_REQUIRE(ctx, cond)                                                          \
  do {                                                                                  \
    if (!(cond)) return svo_fail((ctx), SVO_HIP_ERR_INVALID, "invalid argument", #cond); \
  } while (0)

// grow-only device workspace of the context (stage records, selection buffers); contents are only valid
// within one entry point
inline int svo_ctx_scratch(svo_hip_ctx* ctx, size_t need, void** out) {
  if (ctx->scratch_bytes < need) {
    if (ctx->scratch) {
      SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->scratch);
      ++ctx->n_frees;
      ctx->scratch = nullptr;
      ctx->scratch_bytes = 0;
    }
    void* p = nullptr;
    SVO_CHECK_HIP(ctx, hipMalloc(&p, need + need / 4));
    ++ctx->n_allocs;
    ctx->scratch = p;
    ctx->scratch_bytes = need + need / 4;
  }
  *out = ctx->scratch;
  return SVO_HIP_OK;
}

// grow-only device staging area of the host-buffer entry points (they copy in, run, copy out and synchronise, so
// the area is free again when they return; a context is used by one host thread at a time)
inline int svo_ctx_staging(svo_hip_ctx* ctx, size_t need, char** out) {
  if (ctx->staging_bytes < need) {
    if (ctx->staging) {
      SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->staging);
      ++ctx->n_frees;
      ctx->staging = nullptr;
      ctx->staging_bytes = 0;
    }
    void* p = nullptr;
    SVO_CHECK_HIP(ctx, hipMalloc(&p, need + need / 4));
    ++ctx->n_allocs;
    ctx->staging = p;
    ctx->staging_bytes = need + need / 4;
  }
  *out = static_cast<char*>(ctx->staging);
  return SVO_HIP_OK;
}

// page-locked host area of the same kind: the entry points gather their (pageable) arguments here and move them with
// one transfer instead of one ~10 us transfer per array
inline int svo_ctx_host_staging(svo_hip_ctx* ctx, size_t need, char** out) {
  if (ctx->host_staging_in_flight) {          // an asynchronous pyramid upload still reads the area
    SVO_CHECK_HIP(ctx, hipEventSynchronize(ctx->host_staging_ev));
    ctx->host_staging_in_flight = false;
  }
  if (ctx->host_staging_bytes < need) {
    if (ctx->host_staging) {
      SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipHostFree(ctx->host_staging);
      ++ctx->n_frees;
      ctx->host_staging = nullptr;
      ctx->host_staging_bytes = 0;
    }
    void* p = nullptr;
    SVO_CHECK_HIP(ctx, hipHostMalloc(&p, need + need / 4, hipHostMallocDefault));
    ++ctx->n_allocs;
    ctx->host_staging = p;
    ctx->host_staging_bytes = need + need / 4;
  }
  *out = static_cast<char*>(ctx->host_staging);
  return SVO_HIP_OK;
}

int svo_pyramid_build_levels(svo_hip_pyramid* pyr, int first_slot, int n_slots, const uint8_t* level0_mapped, size_t mapped_stride);    // levels 1.. from level 0 (svo_ctx.hip)

inline svo_dev::Cam svo_make_cam(const svo_hip_camera& c) {
  svo_dev::Cam d;
  d.fx = c.fx; d.fy = c.fy; d.cx = c.cx; d.cy = c.cy;
  for (int i = 0; i < 5; ++i) d.d[i] = c.d[i];
  d.distortion = c.distortion;
  d.width = c.width; d.height = c.height;
  return d;
}

// ---- SparseImgAlign solver records shared between svo_sia.hip and svo_track.hip
namespace svo_dev {
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
constexpr uint8_t F_VISIBLE = 1;                // visible_fts_ (sticky across levels)
constexpr uint8_t F_JVALID = 2;                 // jacobian_cache_ column block non-zero at this level

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

// One Gauss-Newton control step of one frame from its sums r[0..28] (21 H, 6 Jres, chi2, n_meas):
// I/nlls_solver_impl.hpp:35-99 with solve()/update() of S/sparse_img_align.cpp:291-308.  One thread.
SVO_DEV void gn_control_step(FrameState& s, const double* r, int level, int n_iter, double eps, int early_stop) {
  double H[36], Jres[6], x[6];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) { H[i * 6 + j] = r[k]; H[j * 6 + i] = r[k]; ++k; }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) Jres[i] = r[21 + i];
  const double chi2_sum = r[27];
  const unsigned long long n_meas = (unsigned long long)(r[28] + 0.5);
  // computeResiduals returns float chi2 / size_t n_meas evaluated in float (:285)
  const double new_chi2 = (double)((float)chi2_sum / (float)n_meas);
#pragma unroll
  for (int i = 0; i < 36; ++i) s.H[i] = H[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) s.Jres[i] = Jres[i];
  s.n_meas = n_meas;
  s.n_res += n_meas / 16;
  s.iters[level] += 1;

  ldlt6_solve_reg(H, Jres, x);
#pragma unroll
  for (int i = 0; i < 6; ++i) s.x[i] = x[i];
  if (x[0] != x[0]) s.stop = 1;                               // NaN -> stop_ (:52-59)
  const int iter = s.iter;
  if ((early_stop && iter > 0 && new_chi2 > s.chi2) || s.stop) {
    for (int i = 0; i < 7; ++i) s.model[i] = s.old_model[i];  // rollback (:72)
    s.level_done = 1;
    return;
  }
  double mx[6], dT[7], nm[7], cur[7];
#pragma unroll
  for (int i = 0; i < 6; ++i) mx[i] = -x[i];
#pragma unroll
  for (int i = 0; i < 7; ++i) cur[i] = s.model[i];
  se3_exp(mx, dT);
  se3_mul(cur, dT, nm);                                       // T_new = T_old * exp(-x) (:307)
#pragma unroll
  for (int i = 0; i < 7; ++i) { s.old_model[i] = cur[i]; s.model[i] = nm[i]; }
  s.chi2 = new_chi2;
  double mxn = -1;
#pragma unroll
  for (int i = 0; i < 6; ++i) { double a = fabs(x[i]); if (a > mxn) mxn = a; }
  int done = 0;
  if (early_stop && mxn <= eps) done = 1;                     // :97-98
  s.iter = iter + 1;
  if (iter + 1 >= n_iter) done = 1;
  if (done) s.level_done = 1;
}
}  // namespace svo_dev

// internal entries of svo_sia.hip for the device-resident tracking chain (svo_track.hip): a slot of the solver is filled
// by a kernel from device arrays -- the previous frame's features (pixel, bearing, map point index or -1) and the point
// table -- with ref_frame->T_f_w_ = cur_frame->T_f_w_ = *T_last_w_dev (frame_handler_mono.cpp:175); n_feat_host is the
// feature count the host knows from the previous frame's result (it chooses the kernel shape; the kernel itself reads
// the count the device wrote).  The solve leaves its result in the device record svo_sia_state_dev returns.
struct svo_hip_sia;
// (slot: the solver slot of the caller -- 0 for a lone tracker, the camera's index inside a tracker group)
int svo_sia_prepare_from_device(svo_hip_sia* s, int slot, const svo_hip_camera* cam, int n_feat_host, const int* n_feat_dev,
                                const double* T_last_w_dev, const double* px_dev, const double* f_dev, const int32_t* point_dev,
                                const double* pt_pos_dev);
const svo_dev::FrameState* svo_sia_state_dev(const svo_hip_sia* s, int slot);
int svo_sia_slot_arrays(svo_hip_sia* s, int slot, svo_dev::FrameConst** fc, double** px, double** f, double** pos, uint8_t** has_point, int* max_n);
int svo_sia_note_device_slot(svo_hip_sia* s, int slot, const svo_hip_camera* cam, int n_feat_host);

// svo_depth.hip: Matcher::findMatchDirect over n items (svo_hip_match_direct_batch_dev) with, optionally, the current
// frame's pose and the item count read from device memory
int svo_match_direct_internal(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, int cur_slot,
                              const svo_hip_camera* cam, int n_kf, const double* T_ref_w_dev, const double* T_cur_w,
                              const double* T_cur_w_dev, int n, const int* n_dev, const int32_t* kf_slot_dev,
                              const double* px_ref_dev, const double* f_ref_dev, const int32_t* level_ref_dev,
                              const double* pt_pos_dev, const uint8_t* edgelet_dev, const double* grad_dev, int n_pyr_levels,
                              int align_max_iter, double* px_cur_dev, uint8_t* success_dev, int32_t* search_level_dev);

// ---- svo_sia.hip <-> svo_nlls.hip: the solver's other NLLSSolver branches (Levenberg-Marquardt, robust weights) run
// on the streaming solver's buffers.  The view is valid between svo_hip_sia_level_begin and the next call that changes
// the solver.
struct svo_sia_view {
  svo_hip_ctx* ctx;
  int batch, max_n, n_slots, level, chunks;
  int cols, rows;                       // of the current level
  svo_dev::FrameConst* fc;
  svo_dev::FrameState* st;
  const float4 *ref_cache, *dxc, *dyc;  // [slot][max_n][4 rows] x float4: reference patch, dx, dy of the level
  const double4* xyz4;                  // {x, y, z, 1/z} of the feature in the reference frame
  const uint8_t* flags;                 // svo_dev::F_VISIBLE | F_JVALID per patch
  double* partial;                      // [slot][chunks][SVO_HIP_REDUCE_DOUBLES] block sums of an evaluation
  const uint8_t* cur_level;             // level image of slot 0 of the current-frame pyramid; slot b is + b * pyr_bytes
  size_t pyr_bytes;
};
struct svo_nlls_ext;                    // device state of the extra branches, owned by the solver (svo_nlls.hip)
int svo_sia_view_get(svo_hip_sia* s, svo_sia_view* v);
svo_nlls_ext** svo_sia_nlls_slot(svo_hip_sia* s);
int svo_nlls_run(svo_hip_sia* s, int n_slots, const svo_hip_sia_params* prm, int method, int scale_estimator, int weight_function);
int svo_nlls_scale(svo_hip_sia* s, int slot, float* scale, double* mu, double* nu);
void svo_nlls_free(svo_nlls_ext* e);

namespace svo_dev { struct SeedRec; }
int svo_match_scratch(svo_hip_ctx* ctx, int n_cap, svo_dev::SeedRec** recs, uint32_t** pwb_t, int* n_pad);
int svo_match_stages(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam,
                     int n_cap, const int* n_dev, const int32_t* level_ref_dev, svo_dev::SeedRec* recs, uint32_t* pwb_t, int n_pad,
                     int n_pyr_levels, int align_max_iter, bool edgelets);
int svo_match_stages_cams(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, const svo_hip_camera* cam, int n_cams,
                          int cap, const int* counters_dev, int counter_stride, const int32_t* level_ref_dev, svo_dev::SeedRec* recs,
                          uint32_t* pwb_t, int n_pad, int n_pyr_levels, int align_max_iter, bool edgelets);
// svo_refine.hip: svo_hip_pose_optimize_batch_dev with the counts / poses of the batch n_feat_stride ints / T_stride doubles apart
int svo_pose_optimize_batch_strided(svo_hip_ctx* ctx, int batch, int max_n, const int32_t* n_feat_dev, int n_feat_stride,
                                    const double* T_f_w_dev, int T_stride, const double* f_dev, const double* pos_dev,
                                    const int32_t* level_dev, uint8_t* has_point_dev, double error_multiplier2,
                                    double reproj_thresh, int n_iter, svo_hip_pose_opt_result* results_dev);

// ---- multi-GPU exchange (svo_comm.hip): RCCL over xGMI, or a host-staged shared-memory transport for bring-up / tests.
// Both all-reduce in place on the communicator's context stream and give every rank bitwise the same result.
struct svo_hip_comm;
int svo_comm_all_reduce_sum_f64(svo_hip_comm* c, double* dev, size_t count);
int svo_comm_all_gather(svo_hip_comm* c, void* recv_dev, size_t bytes_per_rank);     // rank r's block at recv_dev + r * bytes, in place
unsigned long long svo_comm_id(const svo_hip_comm* c);        // unique per communicator object (never reused within a process)
svo_hip_ctx* svo_comm_ctx(const svo_hip_comm* c);
extern "C" int svo_hip_comm_info(const svo_hip_comm* c, int* rank, int* world, int* kind);

