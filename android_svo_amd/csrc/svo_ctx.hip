// svo_ctx.hip -- contexts, device memory helpers and HBM-resident image pyramids.
//
// Pyramid layout in HBM: one allocation per batch; slot s occupies
// [s*pyr_bytes, (s+1)*pyr_bytes), its levels are packed back to back (each level
// start rounded up to 16 bytes so dword/dwordx4 loads of the half-sample kernel are
// aligned).  640x480x5 levels = 409 200 B per pyramid.
#include "svo_internal.h"

namespace {

// 2x2 truncating mean, one thread per 4 output pixels (reads 2 x 8 B, writes 4 B):
// vk::halfSample scalar / NEON form, vision.cpp:49-67,89-110.  HBM-bound: 1.25 B moved
// per input byte.
// in / out point at the level inside pyramid slot 0 of the launch; blockIdx.z selects the slot (stride pyr_bytes)
__global__ void half_sample_kernel(const uint8_t* __restrict__ in, int w, int h, uint8_t* __restrict__ out, size_t slot_stride = 0) {
  in += (size_t)blockIdx.z * slot_stride;
  out += (size_t)blockIdx.z * slot_stride;
  const int ow = w >> 1, oh = h >> 1;
  const int quads = (ow + 3) >> 2;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (q >= quads || y >= oh) return;
  const uint8_t* top = in + (size_t)(2 * y) * w + 8 * q;
  const uint8_t* bot = top + w;
  uint8_t* o = out + (size_t)y * ow + 4 * q;
  const int x0 = 4 * q;
  if (x0 + 4 <= ow && (w & 7) == 0 && (ow & 3) == 0) {
    const uint2 t = *reinterpret_cast<const uint2*>(top);
    const uint2 b = *reinterpret_cast<const uint2*>(bot);
    auto px = [](unsigned tw, unsigned bw, int s) -> unsigned {
      return (((tw >> s) & 0xff) + ((tw >> (s + 8)) & 0xff) + ((bw >> s) & 0xff) + ((bw >> (s + 8)) & 0xff)) >> 2;
    };
    const unsigned r = px(t.x, b.x, 0) | (px(t.x, b.x, 16) << 8) | (px(t.y, b.y, 0) << 16) | (px(t.y, b.y, 16) << 24);
    *reinterpret_cast<unsigned*>(o) = r;
  } else {
    for (int i = 0; i < 4 && x0 + i < ow; ++i)
      o[i] = (uint8_t)(((unsigned)top[2 * i] + top[2 * i + 1] + bot[2 * i] + bot[2 * i + 1]) >> 2);
  }
}

// All coarser levels of one pyramid in ONE launch (the per-level kernels above cost a dependent launch each, ~4.5 us, which is
// what a single tracked frame pays: svo_track.hip).  A workgroup takes a full-width strip of 16 rows of level 0 and halves
// it in LDS -- w/2 x 8 -> w/4 x 4 -> w/8 x 2 -> w/16 x 1 -- every level with the same truncating 2x2 mean of the level
// before (vk::halfSample scalar / NEON form), so the bytes are those of the level-by-level build.  Needs width % 16 == 0,
// width <= PYR_STRIP_MAX_W, height % 16 == 0 and at most five levels; other shapes take the per-level kernels.
// src0 != null: level 0 is read from there (page-locked host memory mapped into the device: the image crosses the link
// inside this kernel and is written to the pyramid's level 0 on the way -- no separate copy, no dependent launch behind
// it).  The strip is read row by row, 16 bytes per lane, consecutive lanes consecutive addresses: whole rows per request
// (64 x 16 tiles, i.e. 64-byte requests, read the image at a third of the link rate).
constexpr int PYR_STRIP_MAX_W = 2048;
__global__ __launch_bounds__(256) void pyramid_strip_kernel(uint8_t* __restrict__ base, int w, int h, int n_levels, size_t o1, size_t o2,
                                                            size_t o3, size_t o4, size_t slot_stride, const uint8_t* __restrict__ src0,
                                                            size_t src_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* l0 = lds;                      // [16][w]
  uint8_t* l1 = l0 + 16 * w;              // [8][w/2]
  uint8_t* l2 = l1 + 4 * w;               // [4][w/4]
  uint8_t* l3 = l2 + w;                   // [2][w/8]
  base += (size_t)blockIdx.z * slot_stride;
  if (src0) src0 += (size_t)blockIdx.z * src_stride;       // (a tracker group: the cameras' page-locked images lie src_stride apart)
  const int t = threadIdx.x, nt = blockDim.x;
  const int ty = blockIdx.x;
  {
    const size_t off = (size_t)(16 * ty) * w;                // the strip is contiguous in level 0
    const uint4* s4 = reinterpret_cast<const uint4*>((src0 ? src0 : base) + off);
    uint4* d4 = reinterpret_cast<uint4*>(base + off);
    uint4* l4 = reinterpret_cast<uint4*>(l0);
    for (int i = t; i < w; i += nt) {                        // 16 * w bytes = w uint4
      const uint4 v = s4[i];
      l4[i] = v;
      if (src0) d4[i] = v;
    }
  }
  __syncthreads();
  const int w1 = w >> 1, w2 = w >> 2, w3 = w >> 3, w4 = w >> 4;
  for (int i = t; i < 8 * w1; i += nt) {
    const int y = i / w1, x = i - y * w1;
    const unsigned a = *reinterpret_cast<const unsigned short*>(l0 + (2 * y) * w + 2 * x), b = *reinterpret_cast<const unsigned short*>(l0 + (2 * y + 1) * w + 2 * x);
    const uint8_t v = (uint8_t)(((a & 0xff) + (a >> 8) + (b & 0xff) + (b >> 8)) >> 2);
    l1[y * w1 + x] = v;
    base[o1 + (size_t)(8 * ty + y) * w1 + x] = v;
  }
  if (n_levels <= 2) return;
  __syncthreads();
  for (int i = t; i < 4 * w2; i += nt) {
    const int y = i / w2, x = i - y * w2;
    const uint8_t v = (uint8_t)(((unsigned)l1[(2 * y) * w1 + 2 * x] + l1[(2 * y) * w1 + 2 * x + 1] + l1[(2 * y + 1) * w1 + 2 * x] + l1[(2 * y + 1) * w1 + 2 * x + 1]) >> 2);
    l2[y * w2 + x] = v;
    base[o2 + (size_t)(4 * ty + y) * w2 + x] = v;
  }
  if (n_levels <= 3) return;
  __syncthreads();
  for (int i = t; i < 2 * w3; i += nt) {
    const int y = i / w3, x = i - y * w3;
    const uint8_t v = (uint8_t)(((unsigned)l2[(2 * y) * w2 + 2 * x] + l2[(2 * y) * w2 + 2 * x + 1] + l2[(2 * y + 1) * w2 + 2 * x] + l2[(2 * y + 1) * w2 + 2 * x + 1]) >> 2);
    l3[y * w3 + x] = v;
    base[o3 + (size_t)(2 * ty + y) * w3 + x] = v;
  }
  if (n_levels <= 4) return;
  __syncthreads();
  for (int x = t; x < w4; x += nt) {
    const uint8_t v = (uint8_t)(((unsigned)l3[2 * x] + l3[2 * x + 1] + l3[w3 + 2 * x] + l3[w3 + 2 * x + 1]) >> 2);
    base[o4 + (size_t)ty * w4 + x] = v;
  }
}

}  // namespace

// levels 1.. of n_slots pyramids starting at first_slot, from their level 0 (already in place), on the context stream
// (level0_mapped: device address of a page-locked host image to take level 0 from -- slot k's image at level0_mapped + k *
// mapped_stride; null: level 0 is in place)
int svo_pyramid_build_levels(svo_hip_pyramid* pyr, int first_slot, int n_slots, const uint8_t* level0_mapped, size_t mapped_stride) {
  svo_hip_ctx* ctx = pyr->ctx;
  uint8_t* base = pyr->base + (size_t)first_slot * pyr->pyr_bytes;
  const bool tiled = pyr->width % 16 == 0 && pyr->width <= PYR_STRIP_MAX_W && pyr->height % 16 == 0 && pyr->n_levels <= 5 && pyr->n_levels >= 2 &&
                     (pyr->pyr_bytes % 16) == 0;
  if (level0_mapped && !tiled) {       // shapes the tile kernel does not take: an ordinary copy first
    for (int k = 0; k < n_slots; ++k)
      SVO_CHECK_HIP(ctx, hipMemcpyAsync(base + (size_t)k * pyr->pyr_bytes, level0_mapped + (size_t)k * mapped_stride, (size_t)pyr->width * pyr->height,
                                        hipMemcpyDefault, ctx->stream));
    level0_mapped = nullptr;
  }
  if (pyr->n_levels < 2) return SVO_HIP_OK;
  if (tiled) {
    const size_t o1 = pyr->level_offset[1], o2 = pyr->n_levels > 2 ? pyr->level_offset[2] : 0, o3 = pyr->n_levels > 3 ? pyr->level_offset[3] : 0,
                 o4 = pyr->n_levels > 4 ? pyr->level_offset[4] : 0;
    const size_t lds = (size_t)16 * pyr->width + 4 * pyr->width + pyr->width + pyr->width / 4 + 64;
    hipLaunchKernelGGL(pyramid_strip_kernel, dim3(pyr->height / 16, 1, n_slots), dim3(256), lds, ctx->stream, base, pyr->width, pyr->height,
                       pyr->n_levels, o1, o2, o3, o4, pyr->pyr_bytes, level0_mapped, mapped_stride);
    SVO_CHECK_HIP(ctx, hipGetLastError());
    return SVO_HIP_OK;
  }
  for (int l = 1; l < pyr->n_levels; ++l) {
    const int w = pyr->width >> (l - 1), h = pyr->height >> (l - 1);
    const int ow = w >> 1, oh = h >> 1;
    dim3 block(64), grid(((ow + 3) / 4 + 63) / 64, oh, n_slots);
    hipLaunchKernelGGL(half_sample_kernel, grid, block, 0, ctx->stream, base + pyr->level_offset[l - 1], w, h,
                       base + pyr->level_offset[l], pyr->pyr_bytes);
    SVO_CHECK_HIP(ctx, hipGetLastError());
  }
  return SVO_HIP_OK;
}

extern "C" {

const char* svo_hip_version(void) { return "svo_hip 0.1 (gfx950)"; }

int svo_hip_device_count(int* count) {
  if (!count) return SVO_HIP_ERR_INVALID;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return SVO_HIP_ERR_DEVICE; }
  *count = n;
  return SVO_HIP_OK;
}

int svo_hip_ctx_create(svo_hip_ctx** out, int device, void* stream) {
  if (!out) return SVO_HIP_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SVO_HIP_ERR_DEVICE;
  if (device < 0 || device >= n) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* c = new (std::nothrow) svo_hip_ctx();
  if (!c) return SVO_HIP_ERR_NOMEM;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete c; return SVO_HIP_ERR_DEVICE; }
  if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->n_cu <= 0) c->n_cu = 256;
  if (stream) {
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SVO_HIP_ERR_DEVICE; }
    c->own_stream = true;
  }
  *out = c;
  return SVO_HIP_OK;
}

int svo_hip_ctx_destroy(svo_hip_ctx* ctx) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->staging) (void)hipFree(ctx->staging);
  if (ctx->host_staging) (void)hipHostFree(ctx->host_staging);
  for (svo_seed_block& blk : ctx->seed_pool) {         // (blocks still held by live seed batches go with those batches)
    if (blk.dev) (void)hipFree(blk.dev);
    if (blk.host) (void)hipHostFree(blk.host);
  }
  ctx->seed_pool.clear();
  for (hipEvent_t e : ctx->df_ev) if (e) (void)hipEventDestroy(e);
  if (ctx->host_staging_ev) (void)hipEventDestroy(ctx->host_staging_ev);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return SVO_HIP_OK;
}

int svo_hip_ctx_info(svo_hip_ctx* ctx, svo_hip_ctx_stats* out) {
  if (!ctx || !out) return SVO_HIP_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  out->allocator_calls = ctx->n_allocs;
  out->free_calls = ctx->n_frees;
  out->seed_blocks_in_use = ctx->seed_blocks_in_use;
  out->seed_blocks_free = (int)ctx->seed_pool.size();
  for (const svo_seed_block& blk : ctx->seed_pool) { out->seed_pool_free_device_bytes += blk.dev_bytes; out->seed_pool_free_host_bytes += blk.host_bytes; }
  out->scratch_bytes = ctx->scratch_bytes;
  out->staging_bytes = ctx->staging_bytes + ctx->host_staging_bytes;
  return SVO_HIP_OK;
}

// Drops the free blocks of the seed-batch pool (an application that has shrunk for good; hipFree synchronises the device).
int svo_hip_ctx_trim(svo_hip_ctx* ctx) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (svo_seed_block& blk : ctx->seed_pool) {
    if (blk.dev) { (void)hipFree(blk.dev); ++ctx->n_frees; }
    if (blk.host) { (void)hipHostFree(blk.host); ++ctx->n_frees; }
  }
  ctx->seed_pool.clear();
  return SVO_HIP_OK;
}

int svo_hip_ctx_sync(svo_hip_ctx* ctx) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_HIP_OK;
}

void* svo_hip_ctx_stream(svo_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

const char* svo_hip_last_error(svo_hip_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int svo_hip_malloc(svo_hip_ctx* ctx, void** dev_ptr, size_t bytes) {
  if (!ctx || !dev_ptr) return SVO_HIP_ERR_INVALID;
  *dev_ptr = nullptr;
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(dev_ptr, bytes ? bytes : 1);
  if (e == hipErrorOutOfMemory) return svo_fail(ctx, SVO_HIP_ERR_NOMEM, "hipMalloc", "out of memory");
  SVO_CHECK_HIP(ctx, e);
  return SVO_HIP_OK;
}

int svo_hip_free(svo_hip_ctx* ctx, void* dev_ptr) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  if (!dev_ptr) return SVO_HIP_OK;
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  SVO_CHECK_HIP(ctx, hipFree(dev_ptr));
  return SVO_HIP_OK;
}

int svo_hip_malloc_host(svo_hip_ctx* ctx, void** host_ptr, size_t bytes) {
  if (!ctx || !host_ptr) return SVO_HIP_ERR_INVALID;
  *host_ptr = nullptr;
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  SVO_CHECK_HIP(ctx, hipHostMalloc(host_ptr, bytes ? bytes : 1, hipHostMallocDefault));
  return SVO_HIP_OK;
}

int svo_hip_free_host(svo_hip_ctx* ctx, void* host_ptr) {
  if (!ctx) return SVO_HIP_ERR_INVALID;
  if (!host_ptr) return SVO_HIP_OK;
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  SVO_CHECK_HIP(ctx, hipHostFree(host_ptr));
  return SVO_HIP_OK;
}

int svo_hip_memcpy_h2d(svo_hip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  if (!ctx || (!dst_dev && bytes) || (!src_host && bytes)) return SVO_HIP_ERR_INVALID;
  if (!bytes) return SVO_HIP_OK;
  // pageable host memory: hipMemcpyAsync stages it before returning, so the caller's
  // buffer is not retained (SURVEY 8b "Ownership")
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return SVO_HIP_OK;
}

int svo_hip_memcpy_d2h(svo_hip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
  if (!ctx || (!dst_host && bytes) || (!src_dev && bytes)) return SVO_HIP_ERR_INVALID;
  if (!bytes) return SVO_HIP_OK;
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_HIP_OK;
}

int svo_hip_copy_d2d(svo_hip_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes) {
  if (!ctx || (!dst_dev && bytes) || (!src_dev && bytes)) return SVO_HIP_ERR_INVALID;
  if (!bytes) return SVO_HIP_OK;
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return SVO_HIP_OK;
}

int svo_hip_memset(svo_hip_ctx* ctx, void* dst_dev, int value, size_t bytes) {
  if (!ctx || (!dst_dev && bytes)) return SVO_HIP_ERR_INVALID;
  if (!bytes) return SVO_HIP_OK;
  SVO_CHECK_HIP(ctx, hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
  return SVO_HIP_OK;
}

int svo_hip_pyramid_create(svo_hip_ctx* ctx, int width, int height, int n_levels, int batch,
                           svo_hip_pyramid** out) {
  if (!ctx || !out) return SVO_HIP_ERR_INVALID;
  *out = nullptr;
  SVO_REQUIRE(ctx, width > 0 && height > 0 && batch > 0);
  SVO_REQUIRE(ctx, n_levels >= 1 && n_levels <= SVO_HIP_MAX_LEVELS);
  SVO_REQUIRE(ctx, (width >> (n_levels - 1)) > 0 && (height >> (n_levels - 1)) > 0);
  svo_hip_pyramid* p = new (std::nothrow) svo_hip_pyramid();
  if (!p) return SVO_HIP_ERR_NOMEM;
  p->ctx = ctx; p->width = width; p->height = height; p->n_levels = n_levels; p->batch = batch;
  size_t off = 0;
  for (int l = 0; l < n_levels; ++l) {
    p->level_offset[l] = off;
    off += (size_t)(width >> l) * (size_t)(height >> l);
    off = (off + 15) & ~(size_t)15;
  }
  p->level_offset[n_levels] = off;
  p->pyr_bytes = off;
  void* d = nullptr;
  int rc = svo_hip_malloc(ctx, &d, p->pyr_bytes * (size_t)batch + 64);   // +64: tail slack for 8-byte row reads
  if (rc != SVO_HIP_OK) { delete p; return rc; }
  p->base = (uint8_t*)d;
  *out = p;
  return SVO_HIP_OK;
}

int svo_hip_pyramid_destroy(svo_hip_pyramid* pyr) {
  if (!pyr) return SVO_HIP_ERR_INVALID;
  int rc = svo_hip_free(pyr->ctx, pyr->base);
  delete pyr;
  return rc;
}

int svo_hip_pyramid_upload(svo_hip_pyramid* pyr, int slot, const uint8_t* const* levels) {
  if (!pyr || !levels) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = pyr->ctx;
  SVO_REQUIRE(ctx, slot >= 0 && slot < pyr->batch);
  for (int l = 0; l < pyr->n_levels; ++l) SVO_REQUIRE(ctx, levels[l] != nullptr);
  // The levels are gathered in the context's page-locked staging area (laid out as the slot is) and go down with ONE
  // transfer: five copies out of pageable memory cost five staged, synchronous transfers of the runtime (a 640x480 pyramid:
  // ~60 us against ~35).  The call still does not wait for the transfer: the staging area is guarded by an event.
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  char* hs = nullptr;
  const int rc = svo_ctx_host_staging(ctx, pyr->pyr_bytes, &hs);
  if (rc != SVO_HIP_OK) return rc;
  for (int l = 0; l < pyr->n_levels; ++l)
    memcpy(hs + pyr->level_offset[l], levels[l], (size_t)(pyr->width >> l) * (size_t)(pyr->height >> l));
  const size_t used = pyr->level_offset[pyr->n_levels - 1] + (size_t)(pyr->width >> (pyr->n_levels - 1)) * (size_t)(pyr->height >> (pyr->n_levels - 1));
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(pyr->base + (size_t)slot * pyr->pyr_bytes, hs, used, hipMemcpyHostToDevice, ctx->stream));
  if (!ctx->host_staging_ev) SVO_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->host_staging_ev, hipEventDisableTiming));
  SVO_CHECK_HIP(ctx, hipEventRecord(ctx->host_staging_ev, ctx->stream));
  ctx->host_staging_in_flight = true;
  return SVO_HIP_OK;
}

int svo_hip_pyramid_level_offset(const svo_hip_pyramid* pyr, int level, size_t* offset) {
  if (!pyr || !offset || level < 0 || level >= pyr->n_levels) return SVO_HIP_ERR_INVALID;
  *offset = pyr->level_offset[level];
  return SVO_HIP_OK;
}

int svo_hip_pyramid_upload_packed(svo_hip_pyramid* pyr, int first_slot, int n_slots, const uint8_t* packed) {
  if (!pyr || !packed) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = pyr->ctx;
  SVO_REQUIRE(ctx, first_slot >= 0 && n_slots > 0 && first_slot + n_slots <= pyr->batch);
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(pyr->base + (size_t)first_slot * pyr->pyr_bytes, packed, (size_t)n_slots * pyr->pyr_bytes,
                                    hipMemcpyHostToDevice, ctx->stream));
  return SVO_HIP_OK;
}

int svo_hip_pyramid_upload_level0_and_build(svo_hip_pyramid* pyr, int slot, const uint8_t* level0) {
  if (!pyr || !level0) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = pyr->ctx;
  SVO_REQUIRE(ctx, slot >= 0 && slot < pyr->batch);
  uint8_t* base = pyr->base + (size_t)slot * pyr->pyr_bytes;
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(base, level0, (size_t)pyr->width * pyr->height, hipMemcpyHostToDevice, ctx->stream));
  return svo_pyramid_build_levels(pyr, slot, 1, nullptr, 0);
}

int svo_hip_pyramid_upload_level0_batch_and_build(svo_hip_pyramid* pyr, int first_slot, int n_slots, const uint8_t* level0_packed) {
  if (!pyr || !level0_packed) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = pyr->ctx;
  SVO_REQUIRE(ctx, first_slot >= 0 && n_slots > 0 && first_slot + n_slots <= pyr->batch);
  uint8_t* base = pyr->base + (size_t)first_slot * pyr->pyr_bytes;
  const size_t l0 = (size_t)pyr->width * pyr->height;
  // n_slots level-0 images, back to back on the host, into their slots (one strided transfer)
  SVO_CHECK_HIP(ctx, hipMemcpy2DAsync(base, pyr->pyr_bytes, level0_packed, l0, l0, (size_t)n_slots, hipMemcpyHostToDevice, ctx->stream));
  return svo_pyramid_build_levels(pyr, first_slot, n_slots, nullptr, 0);
}

int svo_hip_pyramid_download_level(svo_hip_pyramid* pyr, int slot, int level, uint8_t* out_host) {
  if (!pyr || !out_host) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = pyr->ctx;
  SVO_REQUIRE(ctx, slot >= 0 && slot < pyr->batch && level >= 0 && level < pyr->n_levels);
  size_t bytes = (size_t)(pyr->width >> level) * (size_t)(pyr->height >> level);
  return svo_hip_memcpy_d2h(ctx, out_host, pyr->base + (size_t)slot * pyr->pyr_bytes + pyr->level_offset[level], bytes);
}

int svo_hip_pyramid_info(const svo_hip_pyramid* pyr, int* width, int* height, int* n_levels, int* batch,
                         size_t* pyr_bytes, void** base_dev) {
  if (!pyr) return SVO_HIP_ERR_INVALID;
  if (width) *width = pyr->width;
  if (height) *height = pyr->height;
  if (n_levels) *n_levels = pyr->n_levels;
  if (batch) *batch = pyr->batch;
  if (pyr_bytes) *pyr_bytes = pyr->pyr_bytes;
  if (base_dev) *base_dev = pyr->base;
  return SVO_HIP_OK;
}

}  // extern "C"
