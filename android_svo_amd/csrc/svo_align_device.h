// svo_align_device.h -- feature_alignment::align2D / align1D as device routines, ONE LANE PER 8x8 PATCH.
//
// Follows S/feature_alignment.cpp:167-281 (the scalar path, canonical on arm64/x86) and :35-152:
// inverse-compositional LK with a mean-offset parameter, f32 throughout, min_update^2 = 0.5^2 for align2D (this
// port's value, :203) and 0.03^2 for align1D, border test before the NaN test (:211-215), `break` on the border
// leaves converged = false but still writes u,v; align1D has the chi2-increase rollback (:117-125, which subtracts
// update[0] from u and update[1] from v as the reference does) and h_inv = 1/H(0,0) * 64 (:63).
//
// Result-identical to the CPU path by construction: a lane walks the 64 pixels of its patch y-then-x exactly as the
// reference's two loops do, so Jres (and align1D's H(0,0) and chi2) are the same 64-term serial f32 sums, rounded at
// the same places (the file is built with -ffp-contract=off: no fused multiply-add anywhere).  64 patches per
// wavefront; nothing crosses lanes.  This is also the instruction-cheapest layout on gfx950: ~15 VALU instructions
// per pixel and iteration with no reduction at all (the earlier 16-lanes-per-patch form paid a butterfly per sum and
// summed in another order, which flipped ~1 % of the `converged` flags).
//
// Register plan per lane: the template gradients dx, dy as 2 x 64 f32 (align1D: 64), the 8x8 template as 16 packed
// words (converted with v_cvt_f32_ubyteN at use), the 9 x 12 bytes of the current image's footprint as 27 words
// (nine unaligned global_load_dwordx3, all in flight together), two rows of nine converted pixels.
#pragma once
#include "svo_device_math.h"

namespace svo_dev {

// Eigen Matrix3f::inverse(): cofactors / det, 3-term sums as a0 + (a1 + a2)
SVO_DEV void inverse3f(const float* m, float* inv) {
#define SVO_M(r, c) m[(r) * 3 + (c)]
#define SVO_COF(i, j) (SVO_M(((i) + 1) % 3, ((j) + 1) % 3) * SVO_M(((i) + 2) % 3, ((j) + 2) % 3) - \
                       SVO_M(((i) + 1) % 3, ((j) + 2) % 3) * SVO_M(((i) + 2) % 3, ((j) + 1) % 3))
  const float c00 = SVO_COF(0, 0), c10 = SVO_COF(1, 0), c20 = SVO_COF(2, 0);
  const float det = c00 * SVO_M(0, 0) + (c10 * SVO_M(1, 0) + c20 * SVO_M(2, 0));
  const float invdet = 1.0f / det;
  inv[0] = c00 * invdet; inv[1] = c10 * invdet; inv[2] = c20 * invdet;
  inv[3] = SVO_COF(0, 1) * invdet; inv[4] = SVO_COF(1, 1) * invdet; inv[5] = SVO_COF(2, 1) * invdet;
  inv[6] = SVO_COF(0, 2) * invdet; inv[7] = SVO_COF(1, 2) * invdet; inv[8] = SVO_COF(2, 2) * invdet;
#undef SVO_COF
#undef SVO_M
}

SVO_DEV unsigned long long load8u(const uint8_t* p) {
  unsigned long long w;
  __builtin_memcpy(&w, p, 8);       // unaligned 8-byte load
  return w;
}

// A 10x10 ref_patch_with_border as 25 little-endian words and an 8x8 ref_patch as 16: what a lane keeps of its patch.
struct PatchWords {
  uint32_t b[25];   // ref_patch_with_border, byte c = row*10 + col
  uint32_t p[16];   // ref_patch, byte r = y*8 + x
};

// byte k (compile-time constant after unrolling) of a packed word array
#define SVO_BYTE(arr, k) (((arr)[(k) >> 2] >> (8 * ((k) & 3))) & 0xffu)

// createPatchFromPatchWithBorder (S/matcher.cpp:138-147): the interior of the bordered patch
SVO_DEV void patch_from_border(PatchWords& pw) {
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = 4 * w + k, y = r >> 3, x = r & 7, c = (y + 1) * 10 + (x + 1);
      v |= SVO_BYTE(pw.b, c) << (8 * k);
    }
    pw.p[w] = v;
  }
}

// 12 bytes of an image row from an arbitrary address (global_load_dwordx3; the pyramid allocation has tail slack)
struct __attribute__((packed, aligned(1))) Row12 { uint32_t w[3]; };
SVO_DEV void load_row12(const uint8_t* p, uint32_t* out) {
  Row12 t;
  __builtin_memcpy(&t, p, 12);
  out[0] = t.w[0]; out[1] = t.w[1]; out[2] = t.w[2];
}
SVO_DEV void row9_f(const uint32_t* w, float* f) {
#pragma unroll
  for (int k = 0; k < 9; ++k) f[k] = (float)SVO_BYTE(w, k);
}

// feature_alignment::align2D for the patch of this lane.  `active` lanes run; the loop is left when no lane of the
// wave is running any more.  Returns `converged`; *px_u, *px_v are rewritten as the reference rewrites
// cur_px_estimate (always, also on failure); *iters = iterations that reached the pixel loop.
SVO_DEV bool align2d_lane(const uint8_t* __restrict__ cur_img, int cols, int rows, int cur_step, const PatchWords& pw,
                          int n_iter, bool active, double* px_u, double* px_v, int* iters) {
  float jx[64], jy[64];
  float H[9];
  {
    float h0 = 0, h1 = 0, h2 = 0, h4 = 0, h5 = 0;     // exact in f32: multiples of 1/4 below 2^22
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      const int y = r >> 3, x = r & 7, c = (y + 1) * 10 + (x + 1);
      jx[r] = (float)(0.5 * ((int)SVO_BYTE(pw.b, c + 1) - (int)SVO_BYTE(pw.b, c - 1)));
      jy[r] = (float)(0.5 * ((int)SVO_BYTE(pw.b, c + 10) - (int)SVO_BYTE(pw.b, c - 10)));
      h0 += jx[r] * jx[r]; h1 += jx[r] * jy[r]; h2 += jx[r]; h4 += jy[r] * jy[r]; h5 += jy[r];
    }
    H[0] = h0; H[1] = h1; H[2] = h2; H[3] = h1; H[4] = h4; H[5] = h5; H[6] = h2; H[7] = h5; H[8] = 64.0f;
  }
  float Hinv[9];
  inverse3f(H, Hinv);

  float mean_diff = 0;
  float u = (float)*px_u;
  float v = (float)*px_v;
  const float min_update_squared = 0.25f;
  bool converged = false;
  bool running = active;
  int it_count = 0;
  for (int iter = 0; iter < n_iter; ++iter) {
    const int u_r = (int)floorf(u);
    const int v_r = (int)floorf(v);
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) running = false;   // `break`
    if (u != u || v != v) running = false;                                             // `return false`
    if (__builtin_amdgcn_ballot_w64(running) == 0ull) break;                           // wave-uniform
    if (running) {
      ++it_count;
      const float subpix_x = u - u_r;
      const float subpix_y = v - v_r;
      const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
      const float wTR = (float)(subpix_x * (1.0 - subpix_y));
      const float wBL = (float)((1.0 - subpix_x) * subpix_y);
      const float wBR = subpix_x * subpix_y;
      const uint8_t* base = cur_img + (v_r - 4) * cur_step + (u_r - 4);
      uint32_t rw[9][3];
#pragma unroll
      for (int y = 0; y < 9; ++y) load_row12(base + y * cur_step, rw[y]);
      float J0 = 0, J1 = 0, J2 = 0;
      float f0[9], f1[9];
      row9_f(rw[0], f0);
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        row9_f(rw[y + 1], f1);
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const int r = y * 8 + x;
          const float search_pixel = wTL * f0[x] + wTR * f0[x + 1] + wBL * f1[x] + wBR * f1[x + 1];
          const float res = search_pixel - (float)SVO_BYTE(pw.p, r) + mean_diff;
          J0 -= res * jx[r];
          J1 -= res * jy[r];
          J2 -= res;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) f0[k] = f1[k];
      }
      const float up0 = Hinv[0] * J0 + (Hinv[1] * J1 + Hinv[2] * J2);
      const float up1 = Hinv[3] * J0 + (Hinv[4] * J1 + Hinv[5] * J2);
      const float up2 = Hinv[6] * J0 + (Hinv[7] * J1 + Hinv[8] * J2);
      u += up0;
      v += up1;
      mean_diff += up2;
      if (up0 * up0 + up1 * up1 < min_update_squared) { converged = true; running = false; }
    }
  }
  *px_u = (double)u;
  *px_v = (double)v;
  *iters = it_count;
  return converged;
}

// feature_alignment::align1D for the patch of this lane (S/feature_alignment.cpp:35-152)
SVO_DEV bool align1d_lane(const uint8_t* __restrict__ cur_img, int cols, int rows, int cur_step, float dir0, float dir1,
                          const PatchWords& pw, int n_iter, bool active, double* px_u, double* px_v, double* h_inv,
                          int* iters) {
  float dv[64];
  float H00 = 0, H01 = 0, H10 = 0, H11 = 0;
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int y = r >> 3, x = r & 7, c = (y + 1) * 10 + (x + 1);
    const float j0 = (float)(0.5 * (dir0 * ((int)SVO_BYTE(pw.b, c + 1) - (int)SVO_BYTE(pw.b, c - 1)) +
                                    dir1 * ((int)SVO_BYTE(pw.b, c + 10) - (int)SVO_BYTE(pw.b, c - 10))));
    dv[r] = j0;
    H00 += j0 * j0; H01 += j0 * 1.0f; H10 += 1.0f * j0; H11 += 1.0f;     // H += J*J^T, serial (:66)
  }
  *h_inv = 1.0 / H00 * 8 * 8;
  const float det = H00 * H11 - H10 * H01;
  const float invdet = 1.0f / det;
  const float Hi00 = H11 * invdet, Hi01 = -H01 * invdet, Hi10 = -H10 * invdet, Hi11 = H00 * invdet;
  float mean_diff = 0;
  float u = (float)*px_u;
  float v = (float)*px_v;
  const float min_update_squared = (float)(0.03 * 0.03);
  float chi2 = 0;
  float up0 = 0, up1 = 0;
  bool converged = false;
  bool running = active;
  int it_count = 0;
  for (int iter = 0; iter < n_iter; ++iter) {
    const int u_r = (int)floorf(u);
    const int v_r = (int)floorf(v);
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) running = false;
    if (u != u || v != v) running = false;
    if (__builtin_amdgcn_ballot_w64(running) == 0ull) break;
    if (running) {
      ++it_count;
      const float subpix_x = u - u_r;
      const float subpix_y = v - v_r;
      const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
      const float wTR = (float)(subpix_x * (1.0 - subpix_y));
      const float wBL = (float)((1.0 - subpix_x) * subpix_y);
      const float wBR = subpix_x * subpix_y;
      const uint8_t* base = cur_img + (v_r - 4) * cur_step + (u_r - 4);
      uint32_t rw[9][3];
#pragma unroll
      for (int y = 0; y < 9; ++y) load_row12(base + y * cur_step, rw[y]);
      float J0 = 0, J1 = 0, new_chi2 = 0;
      float f0[9], f1[9];
      row9_f(rw[0], f0);
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        row9_f(rw[y + 1], f1);
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const int r = y * 8 + x;
          const float search_pixel = wTL * f0[x] + wTR * f0[x + 1] + wBL * f1[x] + wBR * f1[x + 1];
          const float res = search_pixel - (float)SVO_BYTE(pw.p, r) + mean_diff;
          J0 -= res * dv[r];
          J1 -= res;
          new_chi2 += res * res;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) f0[k] = f1[k];
      }
      if (iter > 0 && new_chi2 > chi2) {
        u -= up0;
        v -= up1;
        running = false;
      } else {
        chi2 = new_chi2;
        up0 = Hi00 * J0 + Hi01 * J1;
        up1 = Hi10 * J0 + Hi11 * J1;
        u += up0 * dir0;
        v += up0 * dir1;
        mean_diff += up1;
        if (up0 * up0 + up1 * up1 < min_update_squared) { converged = true; running = false; }
      }
    }
  }
  *px_u = (double)u;
  *px_v = (double)v;
  *iters = it_count;
  return converged;
}

}  // namespace svo_dev
