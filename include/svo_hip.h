/*
 * svo_hip.h -- C-ABI of libsvo_hip.so: the MI355X (gfx950) implementation of SVO's
 * data-parallel hot path (sparse image alignment, align2D, depth-filter update).
 *
 * This is the drop-in boundary: plain C types, pointers and sizes only.
 * The reference has no FFI layer -- its "operator API" is the public C++ surface of
 * svo::SparseImgAlign / svo::feature_alignment / svo::DepthFilter; the host-side C++
 * classes that keep those signatures and forward here live in include/svo_dropin/ and are
 * described in INTEGRATION.md.  Each entry point cites the reference interface it replaces
 * (paths relative to /root/reference/app/src/main/cpp/svo/, "I/" = include/svo/).
 *
 * Conventions
 *   - every function returns an int status (SVO_HIP_OK == 0, negative on error); nothing
 *     throws across the boundary, device errors never abort the calling thread;
 *   - SE3 poses are double[7] = {tx,ty,tz,qx,qy,qz,qw} (I/SE3.h, I/SO3.h member order);
 *   - images are 8-bit, row-major, stride == cols; level l of a pyramid is (w>>l) x (h>>l);
 *   - a context owns one HIP stream; every call on a context is enqueued on that stream and
 *     is asynchronous unless stated otherwise.  Contexts share no mutable state, so the
 *     tracking thread and the depth-filter thread of the reference (SURVEY 8b "Threading")
 *     each use their own context concurrently;
 *   - pointers named *_dev are DEVICE pointers (from svo_hip_malloc or any HIP allocator,
 *     e.g. a torch tensor's data_ptr); all others are host pointers that are copied in
 *     before the call returns and never retained.
 */
#ifndef SVO_HIP_H_
#define SVO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVO_HIP_OK 0
#define SVO_HIP_ERR_INVALID (-1)    /* bad argument (null, out of range, shape mismatch) */
#define SVO_HIP_ERR_DEVICE (-2)     /* a HIP runtime call failed: see svo_hip_last_error */
#define SVO_HIP_ERR_NOMEM (-3)
#define SVO_HIP_ERR_STATE (-4)      /* call sequence violated */

#define SVO_HIP_MAX_LEVELS 8
#define SVO_HIP_REDUCE_DOUBLES 32   /* per frame: 21 H (upper, row-major) + 6 Jres + chi2 sum + n_meas + pad */

typedef struct svo_hip_ctx svo_hip_ctx;
typedef struct svo_hip_pyramid svo_hip_pyramid;
typedef struct svo_hip_sia svo_hip_sia;

/* vk::PinholeCamera forward model (pinhole_camera.cpp:73-106): distortion != 0 enables the
 * 5-coefficient radtan model d = {k1,k2,p1,p2,k3} in world2cam.  cam2world (needed by the
 * epipolar matcher) is implemented for distortion == 0 only. */
typedef struct {
  int width, height;
  double fx, fy, cx, cy;
  double d[5];
  int distortion;
} svo_hip_camera;

/* ---- context / memory ------------------------------------------------------------------ */
/* stream: an existing hipStream_t to enqueue on (e.g. torch's current stream), or NULL to
 * create a private non-blocking stream. */
int svo_hip_ctx_create(svo_hip_ctx** out, int device, void* stream);
int svo_hip_ctx_destroy(svo_hip_ctx* ctx);
int svo_hip_ctx_sync(svo_hip_ctx* ctx);            /* hipStreamSynchronize on the context stream */
/* Memory bookkeeping of a context.  The library takes device / page-locked memory for a context's grow-only work areas
 * and for the blocks of its seed-batch pool (svo_hip_seed_batch_create / _destroy recycle blocks: the reference creates and
 * drops a keyframe's seeds at keyframe rate on the depth-filter thread, S/depth_filter.cpp:129-151,256-261, and hipFree
 * would synchronise the whole device under the tracking thread).  `allocator_calls` counts hipMalloc + hipHostMalloc calls,
 * `free_calls` hipFree + hipHostFree: both stay flat once the context has seen its working set (asserted over 20 keyframes
 * in tests/test_gpu_seed_batch.py).  svo_hip_ctx_trim returns the pool's free blocks to the driver (synchronises). */
typedef struct svo_hip_ctx_stats {
  unsigned long long allocator_calls, free_calls;
  unsigned long long seed_pool_free_device_bytes, seed_pool_free_host_bytes, scratch_bytes, staging_bytes;
  int seed_blocks_in_use, seed_blocks_free;
} svo_hip_ctx_stats;
int svo_hip_ctx_info(svo_hip_ctx* ctx, svo_hip_ctx_stats* out);
int svo_hip_ctx_trim(svo_hip_ctx* ctx);
void* svo_hip_ctx_stream(svo_hip_ctx* ctx);
const char* svo_hip_last_error(svo_hip_ctx* ctx);  /* text of the last failure on this context */
const char* svo_hip_version(void);
int svo_hip_device_count(int* count);

int svo_hip_malloc(svo_hip_ctx* ctx, void** dev_ptr, size_t bytes);
int svo_hip_free(svo_hip_ctx* ctx, void* dev_ptr);
/* page-locked host memory for image / feature staging: uploads from it are asynchronous DMA at link rate, uploads
 * from pageable memory are staged by the runtime (measured: profiles/r01_hbm_probe.json) */
int svo_hip_malloc_host(svo_hip_ctx* ctx, void** host_ptr, size_t bytes);
int svo_hip_free_host(svo_hip_ctx* ctx, void* host_ptr);
int svo_hip_memcpy_h2d(svo_hip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);   /* async */
int svo_hip_memcpy_d2h(svo_hip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);   /* synchronises */
int svo_hip_copy_d2d(svo_hip_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);        /* async */
int svo_hip_memset(svo_hip_ctx* ctx, void* dst_dev, int value, size_t bytes);

/* The camera model exactly as every kernel evaluates it, over n host items (vk::PinholeCamera::world2cam for
 * camera-frame points xyz[n][3] and for unit-plane points uv[n][2], pinhole_camera.cpp:73-106; ::cam2world for
 * pixels px[n][2], :44-71; vk::AbstractCamera::isInFrame(obs[n][2], boundary) when level < 0, else
 * isInFrame(obs, boundary, level), I/abstract_camera.h:58-72).  Each input/output pair may be NULL.  Synchronises. */
int svo_hip_camera_batch(svo_hip_ctx* ctx, const svo_hip_camera* cam, int n, const double* xyz, const double* uv,
                         const double* px, const int32_t* obs, int boundary, int level, double* px_of_xyz,
                         double* px_of_uv, double* f_of_px, uint8_t* in_frame);

/* ---- image pyramids resident in HBM (Frame::img_pyr_, frame.cpp:63,186-195) ------------ */
/* A batch of `batch` pyramids of identical geometry, one contiguous allocation:
 * slot s, level l starts at s*pyr_bytes + level_offset[l]. */
int svo_hip_pyramid_create(svo_hip_ctx* ctx, int width, int height, int n_levels, int batch,
                           svo_hip_pyramid** out);
int svo_hip_pyramid_destroy(svo_hip_pyramid* pyr);
/* copy all levels from host (levels[l] has (w>>l)*(h>>l) bytes) */
int svo_hip_pyramid_upload(svo_hip_pyramid* pyr, int slot, const uint8_t* const* levels);
/* copy level 0 only and build levels 1.. on the device with the truncating 2x2 mean
 * (vk::halfSample scalar/NEON form, vision.cpp:49-67,89-110; frame_utils::createImgPyramid,
 * frame.cpp:186-195).  The x86 SSE2 form of the reference rounds differently (vision.cpp:33-37). */
int svo_hip_pyramid_upload_level0_and_build(svo_hip_pyramid* pyr, int slot, const uint8_t* level0);
int svo_hip_pyramid_download_level(svo_hip_pyramid* pyr, int slot, int level, uint8_t* out_host);
/* n_slots pyramids in the device layout (each pyr_bytes long, level l at svo_hip_pyramid_level_offset) in ONE
 * transfer: from page-locked memory this runs at link rate, where per-level uploads are latency-bound */
int svo_hip_pyramid_level_offset(const svo_hip_pyramid* pyr, int level, size_t* offset);
int svo_hip_pyramid_upload_packed(svo_hip_pyramid* pyr, int first_slot, int n_slots, const uint8_t* packed);
/* n_slots level-0 images (back to back on the host) in one strided transfer, then the coarser levels of all of them
 * on the device: 4 launches for the whole batch, and only level 0 (75 % of the pyramid bytes) crosses the link */
int svo_hip_pyramid_upload_level0_batch_and_build(svo_hip_pyramid* pyr, int first_slot, int n_slots,
                                                  const uint8_t* level0_packed);
int svo_hip_pyramid_info(const svo_hip_pyramid* pyr, int* width, int* height, int* n_levels, int* batch,
                         size_t* pyr_bytes, void** base_dev);

/* ---- SparseImgAlign (I/sparse_img_align.h:33-79, sparse_img_align.cpp:51-308,
 *      I/nlls_solver_impl.hpp:25-100) ------------------------------------------------------ */
typedef struct {
  int max_level, min_level;   /* SparseImgAlign ctor args n_levels / min_level (:29-41)        */
  int n_iter;                 /* n_iter_ (:35)                                                 */
  double eps;                 /* eps_ = 1e-6 (:40)                                             */
  int early_stop;             /* 1: reference Gauss-Newton exits (error increase, |x|<=eps);
                                 0: fixed work, exactly n_iter evaluations per level           */
} svo_hip_sia_params;

typedef struct {
  double T_cur_w[7];          /* cur_frame_->T_f_w_ after run() (:89)                          */
  uint64_t n_tracked;         /* return value of run(): n_meas_/patch_area_ (:91)              */
  double H[36];               /* H_ of the last evaluation: getInformationMatrix()             */
  double chi2;                /* getChi2()                                                      */
  int stop;                   /* stop_                                                          */
  int iters[SVO_HIP_MAX_LEVELS];   /* residual evaluations per pyramid level                    */
  uint64_t n_precompute_patches;   /* algorithmic work counters: sum over levels    */
  uint64_t n_residual_patches;     /* and over evaluations of patches actually accumulated      */
} svo_hip_sia_result;

/* A solver for `batch` independent frame pairs with up to max_features features each. */
int svo_hip_sia_create(svo_hip_ctx* ctx, int batch, int max_features, svo_hip_sia** out);
int svo_hip_sia_destroy(svo_hip_sia* sia);
/* slot s of the solver aligns ref->slot s against cur->slot s (pyramids must outlive the solver use) */
int svo_hip_sia_set_frames(svo_hip_sia* sia, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur);
/* Flattened ref_frame->fts_ in list order (sparse_img_align.cpp:116-118): level-0 pixel px[n][2],
 * unit bearing f[n][3], world position of the feature's point pos[n][3], has_point[n] (point != NULL). */
int svo_hip_sia_upload_features(svo_hip_sia* sia, int slot, int n, const double* px, const double* f,
                                const double* pos, const uint8_t* has_point);
/* ref_frame->T_f_w_, the initial cur_frame->T_f_w_ and the camera of the pair */
int svo_hip_sia_upload_poses(svo_hip_sia* sia, int slot, const svo_hip_camera* cam,
                             const double T_ref_w[7], const double T_cur_w_init[7]);
/* Only patches [n*rank/world, n*(rank+1)/world) of every frame are evaluated by this process;
 * the per-frame sums must then be all-reduced across ranks between accumulate and solve_update (step-wise entry
 * points).  svo_hip_sia_run returns SVO_HIP_ERR_STATE while world != 1: a whole solve over one shard without the
 * exchange would be a different problem. */
int svo_hip_sia_set_shard(svo_hip_sia* sia, int rank, int world);

/* ---- multi-GPU exchange: the reference has none; these entries let the C++ host shard the path ----
 * A communicator binds a context (its stream) to a rank of a group.  Transports:
 *   RCCL over xGMI: either the library creates the communicator from a 128-byte ncclUniqueId that rank 0 obtained
 *     with svo_hip_comm_unique_id and the application distributed (MPI, a file, a socket ...), or it adopts an existing
 *     ncclComm_t (svo_hip_comm_from_nccl; the application keeps ownership).  librccl is resolved at run time.
 *   host-staged exchange through a POSIX shared-memory segment `name` ("/..."; unique per group): for ranks that
 *     share one device or have no peer path (bring-up, tests).  slot_bytes = largest message of one rank.  Blocks the
 *     calling thread at every exchange.
 * Both give every rank bitwise the same all-reduce result, so the ranks' Gauss-Newton decisions stay in lock-step. */
typedef struct svo_hip_comm svo_hip_comm;
int svo_hip_comm_unique_id(void* id128);
int svo_hip_comm_create_rccl(svo_hip_ctx* ctx, const void* id128, int rank, int world, svo_hip_comm** out);
int svo_hip_comm_from_nccl(svo_hip_ctx* ctx, void* nccl_comm, int rank, int world, svo_hip_comm** out);
int svo_hip_comm_create_shm(svo_hip_ctx* ctx, const char* name, int rank, int world, size_t slot_bytes, svo_hip_comm** out);
int svo_hip_comm_destroy(svo_hip_comm* comm);
int svo_hip_comm_info(const svo_hip_comm* comm, int* rank, int* world, int* kind /* 0 RCCL, 1 shared memory */);
/* the rank count the transport itself reports (ncclCommCount / the segment header), not the creation argument */
int svo_hip_comm_count(const svo_hip_comm* comm, int* count);

/* SparseImgAlign::run with every frame's patches split over the ranks of `comm` (BASELINE config C3's variant): each
 * rank evaluates patches [n*rank/world, n*(rank+1)/world) of every slot, ONE all-reduce of n_slots x
 * SVO_HIP_REDUCE_DOUBLES doubles per Gauss-Newton step on the context stream, then the identical solve on every rank.
 * Every rank must hold the same features, poses and pyramids and call this with the same arguments; all ranks end
 * with the same result (svo_hip_sia_download).  comm must have been created on the solver's context (one stream for
 * kernels and collective).  The shard is in force for this call only: a shard set with svo_hip_sia_set_shard is back
 * in place when it returns. */
int svo_hip_sia_run_sharded(svo_hip_sia* sia, svo_hip_comm* comm, int n_slots, const svo_hip_sia_params* prm);
/* Opt-in: replay the per-level launch sequence of svo_hip_sia_run_sharded (level_begin + n_iter x {accumulate,
 * all-reduce, solve_update}) from one HIP graph per pyramid level, captured at the first call of a configuration (RCCL
 * transport only).  Saves host launch time, not device time. */
int svo_hip_sia_set_sharded_graph(svo_hip_sia* sia, int enable);

/* run(): the whole coarse-to-fine solve for slots [0, n_slots), enqueued on the stream with no
 * host round trip (the default, Gauss-Newton; with SVO_HIP_SIA_OPT_METHOD / _SCALE_ESTIMATOR / _CHI2 set the call looks
 * at the frames' state between rounds of evaluations and so waits for the device: see those options).  Poses restart
 * from the uploaded initial poses on every call. */
int svo_hip_sia_run(svo_hip_sia* sia, int n_slots, const svo_hip_sia_params* prm);
/* blocks until the stream is idle, then copies the result of one slot */
int svo_hip_sia_download(svo_hip_sia* sia, int slot, svo_hip_sia_result* out);
/* all slots at once (out[n_slots]) */
int svo_hip_sia_download_all(svo_hip_sia* sia, int n_slots, svo_hip_sia_result* out);

/* The same solve in steps, for callers that exchange the normal equations between devices:
 *   begin; for level = max..min { level_begin(level); n_iter x { accumulate; [all-reduce]; solve_update } } finish
 * accumulate leaves SVO_HIP_REDUCE_DOUBLES doubles per slot in the reduce buffer. */
int svo_hip_sia_begin(svo_hip_sia* sia, int n_slots, const svo_hip_sia_params* prm);
int svo_hip_sia_level_begin(svo_hip_sia* sia, int level);
int svo_hip_sia_accumulate(svo_hip_sia* sia);
int svo_hip_sia_solve_update(svo_hip_sia* sia);
int svo_hip_sia_finish(svo_hip_sia* sia);
/* device address of the reduce buffer ([batch][SVO_HIP_REDUCE_DOUBLES] doubles) */
int svo_hip_sia_reduce_buffer(svo_hip_sia* sia, void** dev_ptr, size_t* n_doubles);
/* use a caller-owned device buffer instead (e.g. a torch tensor that RCCL all-reduces in place) */
int svo_hip_sia_set_reduce_buffer(svo_hip_sia* sia, void* dev_ptr);
/* Tuning / diagnostic switches of ONE solver object (the library reads no environment variable and keeps no
 * process-wide switch: two host threads with two solvers do not see each other's settings).  0 = automatic unless
 * stated otherwise. */
#define SVO_HIP_SIA_OPT_MODE 0          /* SVO_HIP_SIA_MODE_AUTO, or SVO_HIP_SIA_MODE_STREAM: svo_hip_sia_run always uses the streaming kernels */
#define SVO_HIP_SIA_OPT_WAVES 1         /* fused kernel: waves per frame pair, 0 (automatic), 4 or 8 */
#define SVO_HIP_SIA_OPT_CHUNKS 2        /* streaming residual kernel: workgroups per frame, 0 (automatic) .. 64 */
#define SVO_HIP_SIA_OPT_EXTRA_LDS 3     /* fused kernel: waves with a third tile in LDS, -1 (automatic) .. 3 */
#define SVO_HIP_SIA_OPT_OLD_TILES 4     /* fused kernel: tiles of the older wave of a SIMD, 0 (automatic) .. 6 */
#define SVO_HIP_SIA_OPT_ARITH 5         /* fused kernel: SVO_HIP_SIA_ARITH_EXACT (default: the reference's arithmetic), _MOMENTS_F32 or _FAST (opt-in) */
#define SVO_HIP_SIA_OPT_METHOD 6          /* NLLSSolver::method_ (I/nlls_solver.h:46): SVO_HIP_SIA_METHOD_GAUSS_NEWTON (default) or _LEVENBERG_MARQUARDT */
#define SVO_HIP_SIA_OPT_SCALE_ESTIMATOR 7 /* setRobustCostFunction's first argument (:47): SVO_HIP_SIA_SCALE_UNIT (default: no weights), _TDIST, _MAD, _NORMAL */
#define SVO_HIP_SIA_OPT_WEIGHT_FUNCTION 8 /* ... and its second (:48): SVO_HIP_SIA_WEIGHT_UNIT (default), _TDIST, _TUKEY, _HUBER */
#define SVO_HIP_SIA_OPT_CHI2 9            /* SVO_HIP_SIA_CHI2_PER_PATCH (default) or _REFERENCE_ORDER, see below */
#define SVO_HIP_SIA_MODE_AUTO 0
#define SVO_HIP_SIA_MODE_STREAM 1
/* Arithmetic levels of the fused kernel.  At every level the image math is the reference's f32, pixel choice, projection,
 * H_ (formed once per level from the patches' gradient sums), normal equations, solve and the Gauss-Newton control flow are
 * unchanged: H_, the tracked-patch count and -- on every scene tested -- the iteration counts are bit for bit the same.
 * EXACT (the default; round 4 shipped MOMENTS_F32 as the default, round 5 took that back: every caller that sets nothing --
 *   the drop-in SparseImgAlign::run binding included -- gets the reference's arithmetic): the reference's arithmetic
 *   statement by statement -- uncontracted f32 interpolation, residual products and Jacobian moments in f64
 *   (sparse_img_align.cpp:238-279): with a fixed evaluation count poses agree with the CPU path to ~1e-13 ... 6e-9 rad.
 * MOMENTS_F32 (opt-in, pays only in launches of thousands of frame pairs): EXACT's residuals and chi2; the 16 products
 *   dx*res and dy*res of a patch are
 *   accumulated in f32 (one fused rounding per term) and widened once per patch, instead of exactly in f64 -- three
 *   conversions and two f64 operations per pixel less, the solve runs ~7 % faster.  Poses agree with the CPU path to
 *   ~3e-8 rad / 6e-8 m with a fixed evaluation count (tests assert 1e-7 against EXACT) and, with the reference's own exits
 *   -- where the f32 chi2 summation order already decides -- exactly as well as EXACT does (5e-8 rad); the reference's own
 *   translation units built for arm64 move by 1.4e-8 rad.  north_star allows 1e-4 rad / 1e-3 m.
 * FAST: additionally the bilinear sum contracted (one product + three fused multiply-adds) and chi2 summed with fused
 *   multiply-adds: ~10 % faster than EXACT, poses ~2e-8 rad from the CPU path.
 * Applies to the fused kernel (svo_hip_sia_run); the streaming / step-wise / sharded paths and svo_hip_tracker's own
 * solver always use EXACT. */
#define SVO_HIP_SIA_ARITH_EXACT 0
#define SVO_HIP_SIA_ARITH_FAST 1
#define SVO_HIP_SIA_ARITH_MOMENTS_F32 2
/* The other branches of vk::NLLSSolver<6,SE3> that SparseImgAlign inherits (I/nlls_solver.h:46-48; no caller in the
 * reference enables them -- frame_handler_mono.cpp:186-187,331-332 construct with GaussNewton and set no robust cost).
 * The values are the reference's enumerators.
 *   METHOD _LEVENBERG_MARQUARDT: optimizeLevenbergMarquardt (I/nlls_solver_impl.hpp:102-227) as SparseImgAlign::run drives
 *     it (mu_ = 0.1 at every level, S/sparse_img_align.cpp:74), with what the reference does around it: n_meas_ is not
 *     cleared before a level's first evaluation (so chi2_ starts low on every level but the first), stop_ survives into the
 *     next level, H_ is the DAMPED matrix of the last trial, params.early_stop is ignored and n_iter bounds the outer
 *     iterations (up to 5 trials each).  result.iters counts evaluations; a trial costs one, not the reference's two: the
 *     sums of the evaluation at an accepted pose are kept and are the next linearisation.
 *   SCALE_ESTIMATOR / WEIGHT_FUNCTION: setRobustCostFunction (:229-281).  _SCALE_UNIT switches the weights off whatever
 *     the weight function (as there).  The scale is estimated at a level's first pose when iter_ of the level before ended
 *     at 0, i.e. normally once per run (S/sparse_img_align.cpp:281-283); the estimators are evaluated in f32 in the order
 *     of the reference's errors vector (TDist and Normal sum sequentially; MAD selects the element nth_element leaves at
 *     size/2; Normal reproduces std::accumulate's int seed), so scale_ is the reference's bit for bit.
 * Both run over the streaming kernels (svo_hip_sia_last_run_mode 0), one launch pair per evaluation, and svo_hip_sia_run
 * looks at the frames' state after every sixth evaluation of a level (the number of trials is data dependent): unlike the
 * default path the call waits for the device.  The step-wise and sharded entry points
 * refuse a solver that has either set (SVO_HIP_ERR_STATE).  chi2 is added up in the reference's order on these paths
 * (SVO_HIP_SIA_OPT_CHI2 below): Levenberg-Marquardt accepts or rejects a trial on the sign of a chi2 difference that is
 * often a few units in the last place. */
/* CHI2: how the residuals' squares are added up.  The reference adds every pixel of every patch, in list order, into ONE
 * f32 (float chi2 ... chi2 += res*res*weight, S/sparse_img_align.cpp:207,266) and decides on that sum whether an iteration
 * increased the error (I/nlls_solver_impl.hpp:62).  _PER_PATCH (default; the fused kernel, the streaming and sharded paths,
 * the tracking chain): f32 within a patch, f64 over the patches -- a better sum, and the only one that parallelises; when two
 * successive values differ by a few units in the last place of the reference's f32 sum, the error-increase exit can fall
 * one iteration earlier or later than in the reference (poses then differ by < 2e-5 rad / 5e-5 m on the scenes tested).
 * _REFERENCE_ORDER: the squares go to memory and are added in the reference's order: chi2_ is the reference's bit
 * for bit and every exit falls where the reference's does.  Costs ~25 us per evaluation at 200 patches (one lane) and ~60 us
 * at 2000 (a workgroup: svo_hip_ordered_sum_f32_dev below) on top of the streaming kernels (svo_hip_sia_last_run_mode 0);
 * Levenberg-Marquardt and the robust costs always use it. */
#define SVO_HIP_SIA_CHI2_PER_PATCH 0
#define SVO_HIP_SIA_CHI2_REFERENCE_ORDER 1
#define SVO_HIP_SIA_METHOD_GAUSS_NEWTON 0
#define SVO_HIP_SIA_METHOD_LEVENBERG_MARQUARDT 1
#define SVO_HIP_SIA_SCALE_UNIT 0
#define SVO_HIP_SIA_SCALE_TDIST 1
#define SVO_HIP_SIA_SCALE_MAD 2
#define SVO_HIP_SIA_SCALE_NORMAL 3
#define SVO_HIP_SIA_WEIGHT_UNIT 0
#define SVO_HIP_SIA_WEIGHT_TDIST 1
#define SVO_HIP_SIA_WEIGHT_TUKEY 2
#define SVO_HIP_SIA_WEIGHT_HUBER 3
int svo_hip_sia_set_option(svo_hip_sia* sia, int option, int value);
/* The f32 sum of n values in index order, rounded exactly as `float s = 0; for (k...) s += x[k];` rounds it -- the form
 * of the reference's chi2 (S/sparse_img_align.cpp:207,266) and of its scale estimators' sums (S/robust_cost.cpp:53-60,83-85)
 * -- by one 1024-thread workgroup instead of a chain of n dependent additions (android_svo_amd/csrc/svo_ordered_sum.h: inside
 * a binade a non-negative addition depends on its predecessors through the parity of the running sum only; sequences of up
 * to 4096 values, and any stretch with a negative, infinite or NaN value, are added by one lane).  1.0 ns per value for long
 * sequences, 62 us for 32 000 values, against 5 ns per value for the one-lane chain (tools/ordered_sum_probe.py).  Enqueued on the context's stream.  What
 * SVO_HIP_SIA_CHI2_REFERENCE_ORDER, Levenberg-Marquardt and the robust costs use. */
int svo_hip_ordered_sum_f32_dev(svo_hip_ctx* ctx, const float* vals_dev, size_t n, float* out_dev);
/* scale_, mu_ and nu_ of one slot as the last svo_hip_sia_run with Levenberg-Marquardt or a robust cost left them
 * (blocks until the stream is idle; SVO_HIP_ERR_STATE when no such run has been made or `slot` was not one of its frames) */
int svo_hip_sia_solver_state(svo_hip_sia* sia, int slot, float* scale, double* mu, double* nu);

/* Which implementation the last svo_hip_sia_run used: 1 = the fused kernel (one workgroup per frame pair,
 * interpolated reference patches in LDS / L2-resident memory, whole coarse-to-fine loop in one launch; chosen
 * when every frame has at most 2816 features and no patch shard is set; launches with at least two frame pairs
 * per compute unit and at most 1024 features per frame use its 4-wave shape, two pairs per compute unit),
 * 0 = the streaming kernels (one launch per Gauss-Newton evaluation; always used by the step-wise entry
 * points).  svo_hip_sia_set_option(SVO_HIP_SIA_OPT_MODE, SVO_HIP_SIA_MODE_STREAM) forces 0.  In a batch that holds
 * frames with fewer than 16 patches those frames take the entry-by-entry Hessian rows a rank-deficient system needs
 * (chosen per workgroup); the other frames' results are bit for bit what they are without such company.  A frame's
 * result does depend on the kernel shape, i.e. on the largest frame of the batch and on the number of pairs. */
int svo_hip_sia_last_run_mode(svo_hip_sia* sia, int* mode);
/* Optional timing of the two heavy kernels with HIP events recorded on the context stream around
 * each launch (precompute: one per level; residual: one per Gauss-Newton evaluation; in fused mode the single
 * launch of a run is reported in the `residual` slot).
 * get_profile synchronises, returns the summed device time and launch counts since the last call
 * and resets the counters. */
int svo_hip_sia_set_profiling(svo_hip_sia* sia, int enable);
int svo_hip_sia_get_profile(svo_hip_sia* sia, double* residual_ms, uint64_t* residual_launches,
                            double* precompute_ms, uint64_t* precompute_launches);
/* cached reference patches / per-patch Jacobian records of one slot, for kernel-level tests (filled by the
 * streaming kernels, i.e. after the step-wise entry points or a run in SVO_HIP_SIA_MODE_STREAM):
 * ref_patch[n][16] f32, dx[n][16] f32, dy[n][16] f32, visible[n] u8 (any may be NULL).  Returns SVO_HIP_ERR_STATE when
 * the last svo_hip_sia_run used the fused kernel (it keeps no per-pixel caches in memory). */
int svo_hip_sia_download_caches(svo_hip_sia* sia, int slot, float* ref_patch, float* dx, float* dy,
                                uint8_t* visible);
/* The same quantities as the FUSED kernel forms them at one pyramid level (it keeps them in LDS / L2-resident memory as
 * 32 interpolated values per patch and never writes per-pixel caches): recomputed by a kernel that shares the fused
 * kernel's device functions for the feature position, the weights and the interpolation, for kernel-level tests against
 * precomputeReferencePatches (sparse_img_align.cpp:144-175).  valid[n]: patches inside the level image with a point;
 * rows of other patches are not written.  Overwrites the streaming kernels' cache arrays of the slot. */
int svo_hip_sia_download_fused_patches(svo_hip_sia* sia, int slot, int level, float* ref_patch, float* dx, float* dy,
                                       uint8_t* valid);

/* ---- feature_alignment::align2D (I/feature_alignment.h:40-47, feature_alignment.cpp:154-282) */
/* n independent 8x8 patches refined on level `level` of cur->slot: ref_patch_with_border
 * [n][100] u8, ref_patch [n][64] u8 (NULL: the interior of the bordered patch, which is how every caller of the
 * reference fills it, matcher.cpp:138-147), px [n][2] f64 in/out (level coordinates), converged [n] u8,
 * iters [n] i32 (iterations executed; may be NULL).  All pointers are device pointers; the patch arrays must be
 * 4-byte aligned (the reference's are 16-byte aligned members of Matcher).
 * One lane per patch walks the 64 pixels in the reference's order: `converged` and px equal the CPU path's bit for
 * bit. */
int svo_hip_align2d_batch_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* cur, int slot, int level, int n,
                              const uint8_t* ref_patch_with_border_dev, const uint8_t* ref_patch_dev,
                              int n_iter, double* px_dev, uint8_t* converged_dev, int32_t* iters_dev);
/* host-buffer convenience form (copies in, runs, copies out, synchronises) */
int svo_hip_align2d_batch(svo_hip_ctx* ctx, const svo_hip_pyramid* cur, int slot, int level, int n,
                          const uint8_t* ref_patch_with_border, const uint8_t* ref_patch, int n_iter,
                          double* px, uint8_t* converged, int32_t* iters);

/* feature_alignment::align1D (I/feature_alignment.h:30-38, feature_alignment.cpp:35-152): n patches that may only
 * move along dir[n][2] (f32, e.g. the epipolar direction); h_inv[n] f64 out (may be NULL).  Device pointers. */
int svo_hip_align1d_batch_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* cur, int slot, int level, int n,
                              const uint8_t* ref_patch_with_border_dev, const float* dir_dev, int n_iter,
                              double* px_dev, uint8_t* converged_dev, double* h_inv_dev, int32_t* iters_dev);

/* ---- Matcher::findMatchDirect over a batch (I/matcher.h:108-111, matcher.cpp:156-202) ----------
 * n (map point, reference feature) pairs against frame cur->cur_slot: frame test of the reference feature,
 * depth = |ref_pos - pt_pos|, warp::getWarpMatrixAffine, getBestSearchLevel, warpAffine, then align2D
 * (align1D along A*grad for EDGELET reference features).  Point::getCloseViewObs and the reprojector's
 * first-success-per-cell policy stay on the host: the caller passes the reference feature it chose.
 * ref holds the n_kf keyframe pyramids (slot k <-> T_ref_w_dev[k][7]); per item: kf_slot, px_ref[2], f_ref[3],
 * level_ref, pt_pos[3], optional edgelet flag + grad[2]; px_cur[n][2] in (estimate) / out, success[n] u8,
 * search_level[n] (may be NULL).  Device pointers except T_cur_w. */
int svo_hip_match_direct_batch_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur,
                                   int cur_slot, const svo_hip_camera* cam, int n_kf, const double* T_ref_w_dev,
                                   const double T_cur_w[7], int n, const int32_t* kf_slot_dev,
                                   const double* px_ref_dev, const double* f_ref_dev, const int32_t* level_ref_dev,
                                   const double* pt_pos_dev, const uint8_t* edgelet_dev, const double* grad_dev,
                                   int n_pyr_levels, int align_max_iter, double* px_cur_dev, uint8_t* success_dev,
                                   int32_t* search_level_dev);

/* The cell loop of Reprojector::reprojectMap (I/reprojector.h:57-59, reprojector.cpp:149-166 with reprojectCell
 * :180-241) on host buffers: candidates bucketed per grid cell in trial order (cell.sort(pointQualityComparator)
 * applied by the caller); candidates of cell c = [cell_offset[c], cell_offset[c+1]).  All candidates are matched in
 * one batch, then the reference's serial policy is replayed (first success per cell wins, candidates behind the
 * winner untouched, stop once n_matches exceeds max_fts = Config::maxFts()).  deleted[i] <=> the point's type is
 * TYPE_DELETED.  Outputs: tried[i] / matched[i] (what the caller needs for the point bookkeeping of :202-215),
 * px_cur[i] and search_level[i] of the visited candidates, cell_winner[c] (candidate index or -1), n_matches_,
 * n_trials_. */
int svo_hip_reproject_cells(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, const svo_hip_pyramid* cur, int cur_slot,
                            const svo_hip_camera* cam, int n_kf, const double* T_kf_w, const double T_cur_w[7], int n_cells,
                            const int32_t* cell_offset, const int32_t* kf_slot, const double* px_ref, const double* f_ref,
                            const int32_t* level_ref, const double* pt_pos, const uint8_t* edgelet, const double* grad,
                            const uint8_t* deleted, double* px_cur, int max_fts, int n_pyr_levels, int align_max_iter,
                            uint8_t* tried, uint8_t* matched, int32_t* search_level, int32_t* cell_winner,
                            uint64_t* n_matches, uint64_t* n_trials);

/* ---- DepthFilter (I/depth_filter.h:36-166, depth_filter.cpp:237-416; matcher.cpp:207-355) -- */
/* static DepthFilter::updateSeed over n seeds (depth_filter.cpp:368-391): SoA device arrays */
int svo_hip_update_seed_batch_dev(svo_hip_ctx* ctx, int n, const float* x_dev, const float* tau2_dev,
                                  float* a_dev, float* b_dev, float* mu_dev, const float* z_range_dev,
                                  float* sigma2_dev);
/* static DepthFilter::computeTau over n (f, z) pairs sharing one T_ref_cur (depth_filter.cpp:396-416) */
int svo_hip_compute_tau_batch_dev(svo_hip_ctx* ctx, int n, const double T_ref_cur[7], const double* f_dev,
                                  const double* z_dev, double px_error_angle, double* tau_dev);

typedef struct {
  int n_pyr_levels;                  /* Config::nPyrLevels() (config.cpp:59: 3)                   */
  int align_max_iter;                /* Matcher::Options::align_max_iter (I/matcher.h:86: 10)      */
  int max_epi_search_steps;          /* Matcher::Options::max_epi_search_steps (I/matcher.h:88: 1000) */
  double seed_convergence_sigma2_thresh;   /* DepthFilter::Options (I/depth_filter.h:85: 100.0)    */
} svo_hip_df_params;

/* per-seed outcome of one updateSeeds pass (depth_filter.cpp:250-340) */
#define SVO_HIP_SEED_BEHIND 0
#define SVO_HIP_SEED_NOT_IN_FRAME 1
#define SVO_HIP_SEED_NO_MATCH 2
#define SVO_HIP_SEED_UPDATED 3
#define SVO_HIP_SEED_CONVERGED 4
#define SVO_HIP_SEED_NAN 5
#define SVO_HIP_SEED_ERASED (-1)   /* seed batches only: the seed had left the list before this pass (nothing was computed) */

/* DepthFilter::updateSeeds for n seeds created in keyframe ref->ref_slot, measured in frame
 * cur->cur_slot: visibility test, Matcher::findEpipolarMatchDirect (epipolar ZMSSD search +
 * align2D + triangulation), computeTau, updateSeed, convergence test.  SoA device arrays:
 * px[n][2] f64, f[n][3] f64, level[n] i32 (Feature px/f/level), a,b,mu,sigma2 in/out f32, z_range f32;
 * outputs status[n] i32, z[n] f64, xyz_world[n][3] f64 (valid when converged), the work
 * counters n_zmssd[n], n_align_iters[n] i32, and the matcher's public results of the seed's call:
 * px_cur[n][2] f64 = Matcher::px_cur_ (level-0 pixel of the match, matcher.cpp:345; valid where status >=
 * SVO_HIP_SEED_UPDATED, NaN elsewhere) -- what updateSeeds passes to feature_detector_->setGridOccpuancy on
 * keyframes (depth_filter.cpp:302-306) -- and search_level[n] i32 = Matcher::search_level_ (-1 for seeds that
 * never reached the matcher).  Any output except status may be NULL. */
int svo_hip_depth_filter_update_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, int ref_slot,
                                    const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam,
                                    const double T_ref_w[7], const double T_cur_w[7], int n,
                                    const double* px_dev, const double* f_dev, const int32_t* level_dev,
                                    float* a_dev, float* b_dev, float* mu_dev, const float* z_range_dev,
                                    float* sigma2_dev, const svo_hip_df_params* prm, int32_t* status_dev,
                                    double* z_dev, double* xyz_world_dev, int32_t* n_zmssd_dev,
                                    int32_t* n_align_iters_dev, double* px_cur_dev, int32_t* search_level_dev);

/* Matcher::findEpipolarMatchDirect over n reference features (I/matcher.h:113-121, matcher.cpp:207-355) with the
 * depth interval given by the caller: depth_est_min_max[3][n] f64 = d_estimate[n], d_min[n], d_max[n].  Outputs:
 * ok[n] u8 (the return value), depth[n] f64 (the `depth` out-parameter), and the public members the call leaves
 * behind: px_cur[n][2] (px_cur_; NaN when the alignment failed), search_level[n] (search_level_), epi_length[n]
 * (epi_length_), plus the work counters.  Any output except ok may be NULL.  Device pointers. */
int svo_hip_epipolar_match_batch_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, int ref_slot,
                                     const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam,
                                     const double T_ref_w[7], const double T_cur_w[7], int n, const double* px_dev,
                                     const double* f_dev, const int32_t* level_dev,
                                     const double* depth_est_min_max_dev, const svo_hip_df_params* prm,
                                     uint8_t* ok_dev, double* depth_dev, double* px_cur_dev, int32_t* search_level_dev,
                                     double* epi_length_dev, int32_t* n_zmssd_dev, int32_t* n_align_iters_dev);

/* Converged seeds of a batch as packed records {seed id = id_offset + index, mu, sigma2, x, y, z} (f64[6]), in seed
 * order, plus their number: the payload of the multi-GPU gather of SURVEY 8e (and of the seed_converged callbacks,
 * depth_filter.cpp:313-329).  records_dev must hold n records. */
int svo_hip_seed_compact_converged_dev(svo_hip_ctx* ctx, int n, long long id_offset, const int32_t* status_dev,
                                       const float* mu_dev, const float* sigma2_dev, const double* xyz_world_dev,
                                       double* records_dev, int32_t* count_dev);

/* The exchange step of the seed-sharded depth filter (BASELINE config C4): this rank's converged seeds (packed as
 * above, clamped to `cap` records) and every other rank's, plus the per-rank counts: records_all[world][cap][6] f64,
 * counts_all[world] i32 (a count above cap means that rank had more: call again with a larger cap).  Two fixed-size
 * all-gathers on the context stream. */
int svo_hip_seed_gather_converged_dev(svo_hip_ctx* ctx, svo_hip_comm* comm, int n, long long id_offset,
                                      const int32_t* status_dev, const float* mu_dev, const float* sigma2_dev,
                                      const double* xyz_world_dev, int cap, double* records_all_dev,
                                      int32_t* counts_all_dev);

/* diagnostic: HIP events around the four stages (geometry, search, align, finalize) of every following depth-filter pass
 * of this context; svo_hip_df_get_profile waits for the last pass and returns its stage durations in microseconds */
int svo_hip_df_set_profiling(svo_hip_ctx* ctx, int enable);
int svo_hip_df_get_profile(svo_hip_ctx* ctx, double stage_us[4]);

/* ---- device-resident seeds: the seeds DepthFilter::initializeSeeds creates for one keyframe (depth_filter.cpp:129-151)
 * live on the device from their creation to their end.  The host uploads a batch ONCE (Feature px / f / level and the
 * Seed constructor's a, b, mu, z_range, sigma2); every frame only the two poses go down (svo_hip_seed_batch_update_async)
 * and only the seeds whose outcome the host must act on come back (svo_hip_seed_batch_collect): seeds that converged
 * (:310-331: new Point at xyz_world, seed_converged_cb(point, sigma2), erase) or turned NaN (:333-337: erase) -- the
 * device stops updating them, as the list erase does -- and, when report_updated is set (the frame is a keyframe,
 * :302-306), every updated seed with its Matcher::px_cur_ for feature_detector_->setGridOccpuancy.  Events are ordered
 * by seed index = list order, so callbacks happen in the reference's order.  Seeds the HOST erases (removeKeyframe,
 * age-out of a part of a batch) are reported with svo_hip_seed_batch_erase.  svo_hip_seed_batch_download returns the
 * current state of every seed for getSeeds() / getSeedsCopy().  One batch belongs to one context (one host thread). */
typedef struct svo_hip_seed_batch svo_hip_seed_batch;
typedef struct {
  int32_t index;            /* position in the batch = creation order of the keyframe's seeds */
  int32_t status;           /* SVO_HIP_SEED_CONVERGED, SVO_HIP_SEED_NAN or (report_updated) SVO_HIP_SEED_UPDATED */
  float mu, sigma2;         /* Seed::mu / Seed::sigma2 after the update */
  double xyz_world[3];      /* converged seeds: ref.T_f_w_.inverse() * (f / mu) (:313) */
  double px_cur[2];         /* Matcher::px_cur_ (level-0 pixel in the current frame) */
} svo_hip_seed_event;
int svo_hip_seed_batch_create(svo_hip_ctx* ctx, int n, const double* px, const double* f, const int32_t* level,
                              const float* a, const float* b, const float* mu, const float* z_range, const float* sigma2,
                              svo_hip_seed_batch** out);                     /* host arrays, uploaded once */
int svo_hip_seed_batch_destroy(svo_hip_seed_batch* batch);
int svo_hip_seed_batch_size(const svo_hip_seed_batch* batch, int* n, int* n_alive);     /* n_alive: as of the last collect */
/* enqueue one updateSeeds pass of the batch against cur->cur_slot on the context stream (returns at once) */
int svo_hip_seed_batch_update_async(svo_hip_seed_batch* batch, const svo_hip_pyramid* ref, int ref_slot,
                                    const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam,
                                    const double T_ref_w[7], const double T_cur_w[7], const svo_hip_df_params* prm,
                                    int report_updated);
/* The same pass over SEVERAL batches of one context -- the seeds of every keyframe a frame updates (the loop of
 * depth_filter.cpp:248-340 walks them all) -- as ONE set of launches: batch k against keyframe slot ref_slots[k] of `ref`
 * with pose T_ref_w[7 * k .. 7 * k + 6].  Results per batch are bit-identical to n_batches calls of
 * svo_hip_seed_batch_update_async; what changes is the cost when the batches are small (a few hundred seeds each is what
 * the reference's detector yields per keyframe): one set of launches per frame -- two for passes of a few thousand seeds,
 * six beyond (svo_hip_df_set_small_pass_limit) -- instead of one set per keyframe.  The arguments of every batch are
 * checked before anything is enqueued.  Up to 8 batches go down per set of launches: with at most 8 the call updates all
 * batches or none; with more, a device failure in a later set (out of memory for its work area, a launch error) returns
 * the error while the earlier sets are already enqueued -- svo_hip_seed_batch_pending tells which batches then have a
 * pass to collect (DeviceSeedMirror collects exactly those). */
int svo_hip_seed_batch_update_group_async(int n_batches, svo_hip_seed_batch* const* batches, const svo_hip_pyramid* ref,
                                          const int* ref_slots, const svo_hip_pyramid* cur, int cur_slot,
                                          const svo_hip_camera* cam, const double* T_ref_w /* [n_batches][7] */,
                                          const double T_cur_w[7], const svo_hip_df_params* prm, int report_updated);
/* diagnostic / tuning: passes of at most max_seeds seed records (sum over the batches of a call, each rounded up to 256)
 * run as TWO launches -- one wave takes its four seeds through geometry, search, alignment and update, then one workgroup
 * per batch writes the events -- instead of six; results are bit-identical.  0 switches the form off; at most 16384;
 * the default is 8192 (DESIGN.md section 8 has the crossover measurement). */
int svo_hip_df_set_small_pass_limit(svo_hip_ctx* ctx, int max_seeds);
/* 1 while a pass of the batch is enqueued and not collected yet, else 0 (also for NULL) */
int svo_hip_seed_batch_pending(const svo_hip_seed_batch* batch);
/* wait for the pass; *events points into page-locked memory owned by the batch (valid until its next update_async);
 * status_counts[7]: seeds per outcome of this pass, slot = status + 1 (slot 0 = SVO_HIP_SEED_ERASED ... slot 6 =
 * SVO_HIP_SEED_NAN).  Either output may be NULL. */
int svo_hip_seed_batch_collect(svo_hip_seed_batch* batch, const svo_hip_seed_event** events, int* n_events,
                               int32_t status_counts[7]);
int svo_hip_seed_batch_erase(svo_hip_seed_batch* batch, int n, const int32_t* indices);      /* host array */
/* state of every seed of the batch, erased ones included (alive[i] = 0); any output may be NULL; synchronises */
int svo_hip_seed_batch_download(svo_hip_seed_batch* batch, float* a, float* b, float* mu, float* sigma2, uint8_t* alive);
/* the batch's device arrays (tests, benchmarks): a, b, mu, sigma2 [n] f32, status [n] i32 of the last pass */
int svo_hip_seed_batch_arrays(svo_hip_seed_batch* batch, float** a_dev, float** b_dev, float** mu_dev, float** sigma2_dev,
                              int32_t** status_dev);

/* host-buffer convenience form of the above (copies in, runs, copies out, synchronises) */
int svo_hip_depth_filter_update(svo_hip_ctx* ctx, const svo_hip_pyramid* ref, int ref_slot,
                                const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam,
                                const double T_ref_w[7], const double T_cur_w[7], int n, const double* px,
                                const double* f, const int32_t* level, float* a, float* b, float* mu,
                                const float* z_range, float* sigma2, const svo_hip_df_params* prm,
                                int32_t* status, double* z, double* xyz_world, int32_t* n_zmssd,
                                int32_t* n_align_iters, double* px_cur, int32_t* search_level);

/* ---- next rows: the two small Gauss-Newton refinements of FrameHandlerMono::processFrame ------
 * pose_optimizer::optimizeGaussNewton (I/pose_optimizer.h:36-45, pose_optimizer.cpp:31-181; caller
 * frame_handler_mono.cpp:226-229): motion-only refinement of frame->T_f_w_ over the features that have a point:
 * MAD scale, <= n_iter Tukey-weighted Gauss-Newton steps on the unit plane, covariance, outlier test.
 * Per frame b of the batch: n_feat[b] observations at [b][0..max_n): f[3] unit bearing (Feature::f), pos[3]
 * (Point::pos_), level (Feature::level), has_point u8 in/out (cleared where the reference sets
 * (*it)->point = NULL, pose_optimizer.cpp:154-157).  error_multiplier2 = cam->errorMultiplier2(). */
typedef struct {
  int32_t ran;             /* 0: no observation had a point -- the reference returns before touching anything */
  int32_t n_iter_done;     /* linear systems built and solved */
  int32_t n_deleted;       /* observations removed by the final test */
  int32_t pad_;
  uint64_t num_obs;        /* observations left (the reference's num_obs) */
  double T_f_w[7];         /* refined pose (rolled back when the last step was rejected) */
  double estimated_scale;  /* as returned through the reference's parameter (MAD scale * errorMultiplier2) */
  double error_init, error_final;
  double Cov[36];          /* frame->Cov_ */
} svo_hip_pose_opt_result;

int svo_hip_pose_optimize_batch_dev(svo_hip_ctx* ctx, int batch, int max_n, const int32_t* n_feat_dev,
                                    const double* T_f_w_dev, const double* f_dev, const double* pos_dev,
                                    const int32_t* level_dev, uint8_t* has_point_dev, double error_multiplier2,
                                    double reproj_thresh, int n_iter, svo_hip_pose_opt_result* results_dev);
/* host-buffer convenience form for one frame (copies in, runs, copies out, synchronises) */
int svo_hip_pose_optimize(svo_hip_ctx* ctx, int n, const double T_f_w[7], const double* f, const double* pos,
                          const int32_t* level, uint8_t* has_point, double error_multiplier2, double reproj_thresh,
                          int n_iter, svo_hip_pose_opt_result* result);

/* Point::optimize (I/point.h:74, point.cpp:130-192; caller FrameHandlerBase::optimizeStructure,
 * frame_handler_base.cpp:190-210) for n_points map points: pos[n][3] in/out; the observations of point p are
 * obs[obs_offset[p] .. obs_offset[p+1]) in the order of Point::obs_: pose of the observing frame (T_f_w[7]) and
 * the feature's bearing f[3].  iters (optional): linear systems solved per point. */
int svo_hip_point_optimize_batch_dev(svo_hip_ctx* ctx, int n_points, int n_iter, double* pos_dev,
                                     const int32_t* obs_offset_dev, const double* obs_T_f_w_dev,
                                     const double* obs_f_dev, int32_t* iters_dev);

/* host-buffer convenience form (copies in, runs, copies out, synchronises) */
int svo_hip_point_optimize_batch(svo_hip_ctx* ctx, int n_points, int n_iter, double* pos, const int32_t* obs_offset,
                                 const double* obs_T_f_w, const double* obs_f, int32_t* iters);

/* ---- one tracked frame on one stream: FrameHandlerMono::processFrame up to the pose refinement ------------------
 *      (frame_handler_mono.cpp:171-229; SURVEY 8f-1 / f-2 / f-4 chained behind a-1)
 *   new_frame->T_f_w_ = last_frame->T_f_w_ (:175) -> SparseImgAlign(kltMaxLevel, kltMinLevel, 30).run(last, new) (:186-188)
 *   -> Reprojector::reprojectMap(new_frame, overlap_kfs) (:203; reprojector.cpp:72-259 with Map::getCloseKeyframes,
 *   Point::getCloseViewObs, Matcher::findMatchDirect) -> pose_optimizer::optimizeGaussNewton (:226-229) -> last_frame_ =
 *   new_frame_ (:91), enqueued as one chain of kernels with ONE synchronisation at the end: the pose SparseImgAlign leaves is
 *   read by the later stages on the device, the frame's matches become the next call's reference features without
 *   leaving it, and only level 0 of the new image crosses the link (from page-locked staging).
 * The map the reprojector walks is a set of index tables uploaded when the map changes (svo_hip_tracker_set_map);
 * keyframe pyramids live in a batch owned by the tracker.  The counters the reprojector keeps on the points are advanced
 * on the device and returned with every frame; when it DELETES a point (reprojector.cpp:126-133,202-209) the pointer
 * graph changes (Map::safeDeletePoint, S/map.cpp:78-88: every observation lets go of the point, a keyframe that loses a key
 * feature chooses its key features again, Frame::removeKeyPoint): the device tables follow -- the point is unlinked there and
 * the affected keyframes re-select before the next frame with the reference's rule -- and the result says map_changed so that
 * the host applies the same deletions to its own objects; a new upload of the map is NOT needed for that.
 * Host code keeps what it keeps in the reference: keyframe selection, Map / Point / Feature objects, optimizeStructure
 * (svo_hip_point_optimize_batch + svo_hip_tracker_update_point_positions), relocalisation (svo_hip_tracker_set_last_frame). */
typedef struct svo_hip_tracker svo_hip_tracker;

typedef struct {
  /* capacities */
  int max_keyframes, max_points, max_obs, max_kf_features, max_candidates;
  int max_items;               /* candidates over all grid cells of one frame */
  int max_frame_features;      /* features of one frame (<= 2816) */
  int n_levels;                /* pyramid levels of a frame: max(Config::nPyrLevels(), Config::kltMaxLevel() + 1) (frame.cpp:63) */
  /* the values processFrame reads from svo::Config / the objects' Options */
  int klt_max_level, klt_min_level, sia_n_iter;    /* config.cpp:62-63; frame_handler_mono.cpp:186-187 */
  double sia_eps;                                  /* sparse_img_align.cpp:40 */
  int grid_size, max_fts, quality_min_fts;         /* Config::gridSize(), maxFts(), qualityMinFts() */
  int reproj_max_n_kfs;                            /* Reprojector::Options::max_n_kfs (<= 16) */
  int n_pyr_levels, align_max_iter;                /* Config::nPyrLevels(), Matcher::Options::align_max_iter */
  double pose_optim_thresh;                        /* Config::poseOptimThresh() */
  int pose_optim_num_iter;                         /* Config::poseOptimNumIter() */
} svo_hip_tracker_config;

/* the reference's defaults (config.cpp:56-84) and roomy capacities */
int svo_hip_tracker_default_config(svo_hip_tracker_config* cfg);

/* svo::Map as index tables (host arrays, copied in).  Keyframes in Map::keyframes_ order; kf_slot[k] = slot of keyframe
 * k's pyramid in the tracker's keyframe batch; kf_key_point[k][5] = point index of Frame::key_pts_[j]->point or -1;
 * kf_ftr_point = the point of every keyframe feature that has one, keyframe by keyframe in fts_ order
 * ([kf_ftr_offset[k], kf_ftr_offset[k+1])).  Points: Point::pos_, type_ (0 deleted, 1 candidate, 2 unknown, 3 good),
 * n_failed_reproj_, n_succeeded_reproj_, and their observations in Point::obs_ order ([pt_obs_offset[p],
 * pt_obs_offset[p+1])): keyframe INDEX the feature lies in, Feature::px / f / level, EDGELET flag and grad (both may be
 * NULL: corners).  cand_point: the points of MapPointCandidates::candidates_ in list order (their single observation is
 * the seed's feature). */
typedef struct {
  int n_kf;
  const int32_t* kf_slot;
  const double* T_kf_w;
  const int32_t* kf_key_point;
  const int32_t* kf_ftr_offset;
  const int32_t* kf_ftr_point;
  int n_points;
  const double* pt_pos;
  const int32_t* pt_type;
  const int32_t* pt_n_failed;
  const int32_t* pt_n_succeeded;
  const int32_t* pt_obs_offset;
  const int32_t* obs_kf;
  const double* obs_px;
  const double* obs_f;
  const int32_t* obs_level;
  const uint8_t* obs_edgelet;
  const double* obs_grad;
  int n_candidates;
  const int32_t* cand_point;
} svo_hip_tracker_map;

typedef struct {
  double T_f_w[7];              /* new_frame_->T_f_w_ as processFrame leaves it for last_frame_: the refined pose; the LAST
                                   frame's pose when the reprojector matched fewer than quality_min_fts points (:208-215) */
  double T_f_w_sia[7];          /* ... after SparseImgAlign::run (:188) */
  uint64_t sia_n_tracked;       /* img_align_n_tracked */
  int32_t sia_iters[SVO_HIP_MAX_LEVELS];
  int32_t sia_stop;
  int32_t n_features;           /* new_frame_->fts_.size(): features the reprojector added, in creation order */
  uint64_t n_matches, n_trials; /* reprojector_.n_matches_, n_trials_ */
  int32_t n_overlap;            /* overlap_kfs_.size() */
  int32_t map_changed;          /* 1: a point or candidate was deleted (its type is TYPE_DELETED in pt_type): apply Map::safeDeletePoint /
                                   deleteCandidatePoint to the host's objects; the device tables have followed already */
  int32_t overlap_kf[16];       /* overlap_kfs_[i].first as keyframe index */
  int32_t overlap_count[16];    /* overlap_kfs_[i].second */
  int32_t n_candidates;         /* candidates in all cells */
  int32_t items_overflow;       /* 1: more than max_items candidates (the excess was dropped: raise max_items) */
  svo_hip_pose_opt_result pose; /* sfba_*: ran == 0 when the refinement was not reached */
} svo_hip_track_result;

int svo_hip_tracker_create(svo_hip_ctx* ctx, const svo_hip_camera* cam, const svo_hip_tracker_config* cfg, svo_hip_tracker** out);
int svo_hip_tracker_destroy(svo_hip_tracker* trk);
/* grid geometry (Reprojector::initializeGrid, reprojector.cpp:44-55) and the keyframe pyramid batch (borrowed) */
int svo_hip_tracker_info(const svo_hip_tracker* trk, int* n_cells, int* grid_cols, int* grid_rows, svo_hip_pyramid** keyframe_pyramids);
/* keyframe images: level 0 from the host (pyramid built on the device), or the pyramid of the last tracked frame
 * (FrameHandlerMono: new_frame_->setKeyframe(); map_.addKeyframe(new_frame_), :284-330) */
int svo_hip_tracker_upload_keyframe(svo_hip_tracker* trk, int slot, const uint8_t* level0);
int svo_hip_tracker_keyframe_from_last_frame(svo_hip_tracker* trk, int slot);
/* The map as index tables (every index is checked here, on the host).  The last frame's features refer to map points by index: a
 * map with the same point numbering may be set between two tracked frames; if the new map has fewer points than the largest
 * index the last frame uses, the last frame is forgotten and svo_hip_tracker_set_last_frame has to follow.  A map that is
 * refused (an index out of range, a table above its capacity) changes nothing. */
int svo_hip_tracker_set_map(svo_hip_tracker* trk, const svo_hip_tracker_map* map);
/* the keyframes' key points as the device holds them ([n_kf][5] point indices, -1 = none): what was uploaded, advanced by the
 * re-selections that followed deletions (parity tests) */
int svo_hip_tracker_download_key_points(svo_hip_tracker* trk, int32_t* kf_key_point);
/* Point::pos_ of n points after FrameHandlerBase::optimizeStructure (frame_handler_base.cpp:190-210) */
int svo_hip_tracker_update_point_positions(svo_hip_tracker* trk, int n, const int32_t* point, const double* pos);
/* FrameHandlerBase::optimizeStructure (frame_handler_base.cpp:190-210) without the observations leaving the device: the host
 * selects the points as the reference does (std::nth_element by Point::last_structure_optim_) and passes their indices
 * (n <= 64, no duplicates); Point::optimize(n_iter) runs on each over the observations the map tables hold (keyframe pose +
 * bearing in Point::obs_ order), and the new positions go into the device's point table, into the solver's copy of the last
 * frame's points and to pos_out[n][3] (iters_out[n]: iterations taken, may be NULL).  One launch, one synchronisation;
 * replaces svo_hip_point_optimize_batch + svo_hip_tracker_update_point_positions between two tracked frames. */
int svo_hip_tracker_optimize_structure(svo_hip_tracker* trk, int n, const int32_t* point, int n_iter, double* pos_out, int32_t* iters_out);
/* last_frame_ from the host (after initialisation / relocalisation): its image (level0, or NULL = the keyframe pyramid in
 * kf_slot), pose and features: px[n][2], f[n][3], point[n] (index or -1) */
int svo_hip_tracker_set_last_frame(svo_hip_tracker* trk, const uint8_t* level0, int kf_slot, const double T_f_w[7], int n,
                                   const double* px, const double* f, const int32_t* point);
/* The tracker's own image buffer: width x height bytes of page-locked, device-mapped host memory the first kernel of a
 * frame reads level 0 from.  svo_hip_tracker_track copies the caller's image there; a caller whose camera frames can land
 * in this buffer (a cv::Mat header over it, Frame::img_pyr_[0] of the new frame) passes the buffer itself as level0 and
 * saves that copy (about 10 us of a 640x480 frame).  The buffer belongs to the tracker, is free again when
 * svo_hip_tracker_track returns, and lives until svo_hip_tracker_destroy. */
int svo_hip_tracker_image_buffer(svo_hip_tracker* trk, uint8_t** buffer);
/* One frame.  Outputs (host, any may be NULL except result): the new frame's features in creation order -- px[n][2],
 * f[n][3], level[n], point[n] (-1 where the pose refinement dropped the observation, pose_optimizer.cpp:154-157),
 * edgelet[n], grad[n][2], capacity max_frame_features -- and the point counters after the frame (capacity n_points of
 * the map).  Synchronises once. */
int svo_hip_tracker_track(svo_hip_tracker* trk, const uint8_t* level0, svo_hip_track_result* result, double* feat_px,
                          double* feat_f, int32_t* feat_level, int32_t* feat_point, uint8_t* feat_edgelet, double* feat_grad,
                          int32_t* pt_type, int32_t* pt_n_failed, int32_t* pt_n_succeeded);

/* ---- several cameras: FrameHandlerMono::processFrame (frame_handler_mono.cpp:171-229) of N independent cameras per call.
 * north_star's "concurrent frame pairs" for the whole per-frame chain: N svo::FrameHandlerMono objects (N cameras, or N
 * sequences replayed side by side) each track one frame at a time; with N trackers on N host threads the HIP runtime's launch
 * path serialises them (measured: 5.5 k frames/s for one camera, 20 k for 4, 21-26 k for 8-16, whatever the number of hardware
 * queues: profiles/r05_cameras_threads.txt).  A group creates N trackers with one camera model and one configuration that
 * share the SparseImgAlign solver (camera c = slot c), the frame and keyframe pyramid batches and the per-camera arrays of
 * the batched stages, and tracks one frame of EVERY camera with ONE chain of launches: every kernel of the chain takes one
 * workgroup (or one slice of its grid) per camera, the cameras' arguments come from a table in device memory.
 * A camera's handle (svo_hip_tracker_group_camera) takes every svo_hip_tracker_* call except _track and _destroy: its map,
 * its last frame, its keyframe slots (0 .. max_keyframes - 1, its own), structure optimisation, image buffer.  Outcomes are
 * bit for bit those of the same camera tracked by a lone svo_hip_tracker (tests/test_gpu_tracker_group.py), provided the
 * SparseImgAlign kernel shape is the same: it is chosen by the largest feature count among the cameras' last frames
 * (svo_hip_sia_last_run_mode).  cfg->max_items must be a multiple of 16 when n_cameras > 1. */
typedef struct svo_hip_tracker_group svo_hip_tracker_group;
int svo_hip_tracker_group_create(svo_hip_ctx* ctx, const svo_hip_camera* cam, const svo_hip_tracker_config* cfg, int n_cameras,
                                 svo_hip_tracker_group** out);
int svo_hip_tracker_group_destroy(svo_hip_tracker_group* group);
int svo_hip_tracker_group_camera(svo_hip_tracker_group* group, int index, svo_hip_tracker** camera);
/* One frame of every camera: level0[c] = camera c's new image (its svo_hip_tracker_image_buffer: no copy); results[n_cameras]
 * (may be NULL).  Synchronises once, for all cameras.  The cameras' features and point counters are read afterwards with
 * svo_hip_tracker_last_result on the handles that need them. */
int svo_hip_tracker_group_track(svo_hip_tracker_group* group, const uint8_t* const* level0, svo_hip_track_result* results);
/* The outcome of the camera's last tracked frame again (same outputs as svo_hip_tracker_track; any may be NULL): a copy from
 * the camera's page-locked result block, no device call. */
int svo_hip_tracker_last_result(svo_hip_tracker* trk, svo_hip_track_result* result, double* feat_px, double* feat_f,
                                int32_t* feat_level, int32_t* feat_point, uint8_t* feat_edgelet, double* feat_grad,
                                int32_t* pt_type, int32_t* pt_n_failed, int32_t* pt_n_succeeded);

/* The 6x6 pivoted LDL^T solve both Gauss-Newton solvers use (x = H.ldlt().solve(b), Eigen 3.4 semantics incl. the
 * pseudo-inverse of D), n systems from host buffers: exposed so that the parity tests can show it is bit-identical
 * to Eigen's result. */
int svo_hip_ldlt6_solve_batch(svo_hip_ctx* ctx, int n, const double* H /*[n][36]*/, const double* b /*[n][6]*/,
                              double* x /*[n][6]*/);

/* ---- next row: the producer of depth-filter seeds --------------------------------------------
 * FastDetector::detect (I/feature_detection.h:90-103, feature_detection.cpp:77-122; called by
 * DepthFilter::initializeSeeds, depth_filter.cpp:129-151): cv::FAST(img, kp, 10, true) on the first n_pyr_levels
 * levels of pyramid slot `slot`, one corner per grid cell (cell_size level-0 pixels) chosen by vk::shiTomasiScore,
 * strictly above detection_threshold; cells flagged in occupancy[grid_rows*grid_cols] (setExistingFeatures) are
 * skipped.  Outputs in cell order (capacity = number of cells): px[n][2] level-0 pixel, optional f[n][3] =
 * cam->cam2world(px) (distortion-free cameras only), level[n], optional score[n]; *n_out = n. */
int svo_hip_detect_grid(int width, int height, int cell_size, int* grid_cols, int* grid_rows);
int svo_hip_detect_features_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* pyr, int slot, const svo_hip_camera* cam,
                                int n_pyr_levels, int cell_size, const uint8_t* occupancy_dev,
                                double detection_threshold, int32_t* n_out_dev, double* px_dev, double* f_dev,
                                int32_t* level_dev, float* score_dev);
/* host-buffer convenience form (copies in, runs, copies out, synchronises) */
int svo_hip_detect_features(svo_hip_ctx* ctx, const svo_hip_pyramid* pyr, int slot, const svo_hip_camera* cam,
                            int n_pyr_levels, int cell_size, const uint8_t* occupancy, double detection_threshold,
                            int32_t* n_out, double* px, double* f, int32_t* level, float* score);
/* Seed::Seed for n new seeds (depth_filter.cpp:36-45): a = b = 10, mu = 1/depth_mean, z_range = 1/depth_min,
 * sigma2 = z_range^2/36 */
int svo_hip_seed_init_batch_dev(svo_hip_ctx* ctx, int n, double depth_mean, double depth_min, float* a_dev, float* b_dev,
                                float* mu_dev, float* z_range_dev, float* sigma2_dev);

#ifdef __cplusplus
}
#endif
#endif /* SVO_HIP_H_ */
