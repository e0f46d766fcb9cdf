// svo_detect.hip -- next row of the hot path: the producer of depth-filter seeds.
//
//   FastDetector::detect       S/feature_detection.cpp:77-122   (cv::FAST 9/16 + non-maximum suppression per pyramid
//                                                                level, one corner per grid cell by Shi-Tomasi score)
//   vk::shiTomasiScore         S/vision.cpp:113-154
//   Seed::Seed                 S/depth_filter.cpp:36-45          (DepthFilter::initializeSeeds, :129-151)
//
// Byte/integer work, HBM/L2-bound: thread per pixel for the FAST decision and score (16 circle pixels through the
// cache, 9-contiguous test on two 16-bit masks), thread per pixel for the 3x3 suppression + Shi-Tomasi + the
// per-cell maximum (one 64-bit atomicMax per surviving corner: score bits above, reversed scan order below, so the
// winner is the one the reference's sequential loop keeps), one workgroup for the ordered compaction.
// Everything is integer-exact or exact in f32 (the Shi-Tomasi sums are integers below 2^24), so the result does
// not depend on the order of evaluation.
//
// cv::FAST is OpenCV 4.5.4 code that is not under /root/reference: PARITY UNPINNED for the corner decision / score
// (restated from the published algorithm, checked against the CPU restatement); vk::shiTomasiScore is pinned.
#include "svo_internal.h"

namespace {

using namespace svo_dev;

__constant__ int kDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__constant__ int kDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

SVO_DEV bool has_arc9(unsigned m) {            // 9 contiguous set bits on the 16-bit circle
  const unsigned M = m | (m << 16);
  const unsigned m2 = M & (M >> 1);            // runs of 2
  const unsigned m4 = m2 & (m2 >> 2);          // runs of 4
  const unsigned m8 = m4 & (m4 >> 4);          // runs of 8
  return ((m8 & (M >> 8)) & 0xffffu) != 0;     // runs of 9
}

// FAST-9/16 decision and score (u8, 0 = no corner) of every pixel of one level
__global__ void fast_score_kernel(const uint8_t* __restrict__ img, int w, int h, int t, uint8_t* __restrict__ score) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  uint8_t out = 0;
  if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) {
    const int v = img[y * w + x];
    int d[16];
    unsigned dark = 0, bright = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      d[k] = v - (int)img[(y + kDy[k]) * w + x + kDx[k]];
      dark |= (d[k] > t) ? (1u << k) : 0u;
      bright |= (d[k] < -t) ? (1u << k) : 0u;
    }
    if (has_arc9(dark) || has_arc9(bright)) {
      int a0 = t;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        int mn = d[k], mx = d[k];
#pragma unroll
        for (int j = 1; j < 9; ++j) { const int e = d[(k + j) & 15]; mn = min(mn, e); mx = max(mx, e); }
        a0 = max(a0, max(mn, -mx));
      }
      out = (uint8_t)(a0 - 1);
    }
  }
  score[y * w + x] = out;
}

SVO_DEV float shi_tomasi(const uint8_t* img, int cols, int rows, int u, int v) {
  const int x_min = u - 4, x_max = u + 4, y_min = v - 4, y_max = v + 4;
  if (x_min < 1 || x_max >= cols - 1 || y_min < 1 || y_max >= rows - 1) return 0.0f;
  int sxx = 0, syy = 0, sxy = 0;                 // exact: the reference's f32 sums are integers below 2^24
  for (int y = y_min; y < y_max; ++y) {
    const uint8_t* row = img + cols * y + x_min;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int dx = (int)row[x + 1] - (int)row[x - 1];
      const int dy = (int)row[x + cols] - (int)row[x - cols];
      sxx += dx * dx; syy += dy * dy; sxy += dx * dy;
    }
  }
  float dXX = (float)sxx, dYY = (float)syy, dXY = (float)sxy;
  dXX = (float)(dXX / (2.0 * 64));
  dYY = (float)(dYY / (2.0 * 64));
  dXY = (float)(dXY / (2.0 * 64));
  // C++ overload resolution picks sqrt(float): the parenthesis is evaluated in f32, only the 0.5 factor is double
  const float tr = dXX + dYY;
  // correctly rounded f32 square root: the f64 root rounded once more (2 * 24 + 2 <= 53 bits, so no double-rounding error)
  const float root = (float)sqrt((double)(tr * tr - 4 * (dXX * dYY - dXY * dXY)));
  return (float)(0.5 * (tr - root));
}

__global__ void cell_init_kernel(unsigned long long* __restrict__ cells, int n_cells, float threshold) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_cells) cells[k] = ((unsigned long long)__float_as_uint(threshold) << 32) | 0xffffffffull;
}

// 3x3 suppression (strictly greater than the 8 neighbours), grid cell, occupancy, Shi-Tomasi, per-cell maximum
__global__ void fast_nms_grid_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ score, int w, int h,
                                     int level, int cell_size, int grid_cols, const uint8_t* __restrict__ occupancy,
                                     float threshold, unsigned long long* __restrict__ cells) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x < 3 || x >= w - 3 || y < 3 || y >= h - 3) return;
  const uint8_t* p = score + y * w + x;
  const int s = p[0];
  if (s == 0) return;
  if (!(s > p[1] && s > p[-1] && s > p[-w - 1] && s > p[-w] && s > p[-w + 1] && s > p[w - 1] && s > p[w] && s > p[w + 1])) return;
  const int scale = 1 << level;
  const int k = ((y * scale) / cell_size) * grid_cols + (x * scale) / cell_size;
  if (occupancy && occupancy[k]) return;
  const float st = shi_tomasi(img, w, h, x, y);
  if (!(st > threshold)) return;    // cannot beat the cell's initial score (:80); also keeps the key's float bits non-negative
  const unsigned order = ((unsigned)level << 24) | (unsigned)(y * w + x);          // the reference's loop order
  const unsigned long long key = ((unsigned long long)__float_as_uint(st) << 32) | (unsigned long long)(0xffffffffu - order);
  atomicMax(&cells[k], key);        // larger score wins; equal scores: the one met first (:105 is a strict >)
}

struct LevelDims { int w[SVO_HIP_MAX_LEVELS]; };

// ordered compaction of the winning corners (cell order, as the reference's for_each over `corners`), one workgroup
__global__ __launch_bounds__(256) void detect_compact_kernel(const unsigned long long* __restrict__ cells, int n_cells,
                                                             float threshold, LevelDims dims, Cam cam, int want_f,
                                                             int* __restrict__ n_out, double* __restrict__ px_out,
                                                             double* __restrict__ f_out, int* __restrict__ level_out,
                                                             float* __restrict__ score_out) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int k0 = 0; k0 < n_cells; k0 += 256) {
    const int k = k0 + threadIdx.x;
    unsigned long long key = 0;
    bool keep = false;
    if (k < n_cells) {
      key = cells[k];
      keep = (double)__uint_as_float((unsigned)(key >> 32)) > (double)threshold;            // :115
    }
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (keep) {
      const int i = off + before;
      const unsigned order = 0xffffffffu - (unsigned)(key & 0xffffffffull);
      const int level = (int)(order >> 24);
      const int idx = (int)(order & 0xffffffu);
      const int w = dims.w[level];
      const int scale = 1 << level;
      const double u = (double)((idx % w) * scale), v = (double)((idx / w) * scale);
      px_out[2 * i] = u; px_out[2 * i + 1] = v;
      level_out[i] = level;
      if (score_out) score_out[i] = __uint_as_float((unsigned)(key >> 32));
      if (want_f) cam2world(cam, u, v, f_out + 3 * i);                                       // Feature::Feature (I/feature.h:47-55)
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = s_base;
}

// Seed::Seed (S/depth_filter.cpp:36-45): a = b = 10, mu = 1/depth_mean, z_range = 1/depth_min, sigma2 = z_range^2/36
__global__ void seed_init_kernel(int n, float depth_mean, float depth_min, float* __restrict__ a, float* __restrict__ b,
                                 float* __restrict__ mu, float* __restrict__ z_range, float* __restrict__ sigma2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float m = (float)(1.0 / depth_mean);
  const float zr = (float)(1.0 / depth_min);
  a[i] = 10; b[i] = 10; mu[i] = m; z_range[i] = zr;
  sigma2[i] = zr * zr / 36;
}

}  // namespace

extern "C" {

int svo_hip_detect_grid(int width, int height, int cell_size, int* grid_cols, int* grid_rows) {
  if (width <= 0 || height <= 0 || cell_size <= 0 || !grid_cols || !grid_rows) return SVO_HIP_ERR_INVALID;
  *grid_cols = (width + cell_size - 1) / cell_size;            // ceil(width / cell_size), feature_detection.cpp:31-32
  *grid_rows = (height + cell_size - 1) / cell_size;
  return SVO_HIP_OK;
}

int svo_hip_detect_features_dev(svo_hip_ctx* ctx, const svo_hip_pyramid* pyr, int slot, const svo_hip_camera* cam,
                                int n_pyr_levels, int cell_size, const uint8_t* occupancy_dev,
                                double detection_threshold, int32_t* n_out_dev, double* px_dev, double* f_dev,
                                int32_t* level_dev, float* score_dev) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, pyr && slot >= 0 && slot < pyr->batch);
  SVO_REQUIRE(ctx, n_pyr_levels > 0 && n_pyr_levels <= pyr->n_levels && cell_size > 0);
  SVO_REQUIRE(ctx, n_out_dev && px_dev && level_dev);
  SVO_REQUIRE(ctx, !f_dev || cam);
  SVO_REQUIRE(ctx, (size_t)pyr->width * pyr->height < (1u << 24));
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  int gc = 0, gr = 0;
  svo_hip_detect_grid(pyr->width, pyr->height, cell_size, &gc, &gr);
  const int n_cells = gc * gr;
  size_t score_bytes = 0;
  for (int l = 0; l < n_pyr_levels; ++l) score_bytes += ((size_t)(pyr->width >> l) * (pyr->height >> l) + 15) & ~(size_t)15;
  void* ws = nullptr;
  const int rc = svo_ctx_scratch(ctx, sizeof(unsigned long long) * n_cells + score_bytes + 64, &ws);
  if (rc != SVO_HIP_OK) return rc;
  unsigned long long* cells = reinterpret_cast<unsigned long long*>(ws);
  uint8_t* score = reinterpret_cast<uint8_t*>(cells + n_cells);
  const float thr = (float)detection_threshold;                 // Corner::score is a float (I/feature_detection.h:34)
  hipLaunchKernelGGL(cell_init_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, ctx->stream, cells, n_cells, thr);
  const uint8_t* base = pyr->base + (size_t)slot * pyr->pyr_bytes;
  LevelDims dims;
  memset(&dims, 0, sizeof(dims));
  uint8_t* sc = score;
  for (int l = 0; l < n_pyr_levels; ++l) {
    const int w = pyr->width >> l, h = pyr->height >> l;
    dims.w[l] = w;
    const uint8_t* img = base + pyr->level_offset[l];
    const dim3 blk(64, 4), grd((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(fast_score_kernel, grd, blk, 0, ctx->stream, img, w, h, 10, sc);       // cv::FAST(img, kp, 10, true) (:93-96)
    hipLaunchKernelGGL(fast_nms_grid_kernel, grd, blk, 0, ctx->stream, img, sc, w, h, l, cell_size, gc, occupancy_dev, thr, cells);
    sc += ((size_t)w * h + 15) & ~(size_t)15;
  }
  Cam c;
  memset(&c, 0, sizeof(c));
  if (cam) c = svo_make_cam(*cam);
  hipLaunchKernelGGL(detect_compact_kernel, dim3(1), dim3(256), 0, ctx->stream, cells, n_cells, thr, dims, c, f_dev ? 1 : 0,
                     n_out_dev, px_dev, f_dev, level_dev, score_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

int svo_hip_detect_features(svo_hip_ctx* ctx, const svo_hip_pyramid* pyr, int slot, const svo_hip_camera* cam,
                            int n_pyr_levels, int cell_size, const uint8_t* occupancy, double detection_threshold,
                            int32_t* n_out, double* px, double* f, int32_t* level, float* score) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, pyr && n_out && px && level && cell_size > 0);
  int gc = 0, gr = 0;
  svo_hip_detect_grid(pyr->width, pyr->height, cell_size, &gc, &gr);
  const size_t nc = (size_t)gc * gr;
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  char* d = nullptr;
  char* hs = nullptr;
  {
    const size_t bytes = nc * (5 * sizeof(double) + sizeof(int32_t) + sizeof(float) + 1) + 64;
    const int rc_st = svo_ctx_staging(ctx, bytes, &d);
    if (rc_st != SVO_HIP_OK) return rc_st;
    const int rc_hs = svo_ctx_host_staging(ctx, bytes, &hs);
    if (rc_hs != SVO_HIP_OK) return rc_hs;
  }
  double* dpx = reinterpret_cast<double*>(d);
  double* df = dpx + 2 * nc;
  int32_t* dl = reinterpret_cast<int32_t*>(df + 3 * nc);
  float* ds = reinterpret_cast<float*>(dl + nc);
  int32_t* dn = reinterpret_cast<int32_t*>(ds + nc);
  uint8_t* docc = reinterpret_cast<uint8_t*>(dn + 1);
  hipError_t e = hipSuccess;
  if (occupancy) e = hipMemcpyAsync(docc, occupancy, nc, hipMemcpyHostToDevice, ctx->stream);
  int rc = SVO_HIP_OK;
  if (e == hipSuccess) {
    rc = svo_hip_detect_features_dev(ctx, pyr, slot, cam, n_pyr_levels, cell_size, occupancy ? docc : nullptr,
                                     detection_threshold, dn, dpx, f ? df : nullptr, dl, ds);
    if (rc == SVO_HIP_OK) {
      // the whole result block (one slot per grid cell, 48 B each, then the count) in one transfer to page-locked
      // memory; the first n entries of every array go to the caller
      const size_t out_bytes = (size_t)(reinterpret_cast<char*>(dn + 1) - d);
      e = hipMemcpyAsync(hs, d, out_bytes, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e == hipSuccess) {
        *n_out = *reinterpret_cast<const int32_t*>(hs + (reinterpret_cast<char*>(dn) - d));
        const size_t n = (size_t)*n_out;
        if (n) {
          memcpy(px, hs + (reinterpret_cast<char*>(dpx) - d), 2 * n * sizeof(double));
          if (f) memcpy(f, hs + (reinterpret_cast<char*>(df) - d), 3 * n * sizeof(double));
          memcpy(level, hs + (reinterpret_cast<char*>(dl) - d), n * sizeof(int32_t));
          if (score) memcpy(score, hs + (reinterpret_cast<char*>(ds) - d), n * sizeof(float));
        }
      }
    }
  }
  if (e != hipSuccess) return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_detect_features", hipGetErrorString(e));
  return rc;
}

int svo_hip_seed_init_batch_dev(svo_hip_ctx* ctx, int n, double depth_mean, double depth_min, float* a_dev, float* b_dev,
                                float* mu_dev, float* z_range_dev, float* sigma2_dev) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, n >= 0);
  if (n == 0) return SVO_HIP_OK;
  SVO_REQUIRE(ctx, a_dev && b_dev && mu_dev && z_range_dev && sigma2_dev);
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(seed_init_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, (float)depth_mean, (float)depth_min,
                     a_dev, b_dev, mu_dev, z_range_dev, sigma2_dev);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  return SVO_HIP_OK;
}

}  // extern "C"
