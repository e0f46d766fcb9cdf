// svo_align_device.h -- feature_alignment::align2D / align1D as device routines, 16 lanes per 8x8 patch.
//
// Follows S/feature_alignment.cpp:167-281 (the scalar path, canonical on arm64/x86) and :35-152:
// inverse-compositional LK with a mean-offset parameter, f32 throughout, min_update^2 = 0.5^2 for align2D (this
// port's value, :203) and 0.03^2 for align1D, border test before the NaN test (:211-215), `break` on the border
// leaves converged = false but still writes u,v; align1D has the chi2-increase rollback (:117-125, which subtracts
// update[0] from u and update[1] from v as the reference does) and h_inv = 1/H(0,0) * 64 (:63).
//
// Numerics vs the CPU path: H = sum J J^T is exact in f32 in any summation order (entries are multiples of 1/4
// below 2^24), so Hinv is bit-identical; Jres is summed four pixels per lane in pixel order and then by a 16-lane
// butterfly instead of a 64-term serial sum, so updates differ by f32 rounding (parity tolerance is stated in tests/).
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

// ---- 16 lanes per patch: four patches per wavefront ---------------------------------------------------------
// A quarter wave per 8x8 patch: lane cl of the group owns pixels (row cl/2, columns 4*(cl%2) .. +3).  A wave then
// refines four patches at once and issues a quarter of the instructions per patch
// (the per-iteration arithmetic on u, v and the weights is uniform within a patch and costs the same whether 16 or 64
// lanes carry it).  Sums: four pixels in pixel order per lane, then a 16-lane butterfly; H is exact in any order.
// All 64 lanes must call these; `active` = this lane's group has a patch to refine.  Every group-level value
// (u, v, converged, iterations) is identical on the 16 lanes of a group.
SVO_DEV unsigned long long load8u(const uint8_t* p) {
  unsigned long long w;
  __builtin_memcpy(&w, p, 8);       // unaligned 8-byte load
  return w;
}

// the low five bytes of a 64-bit word as floats (v_cvt_f32_ubyteN: one instruction per byte)
SVO_DEV void bytes5_f(unsigned long long w, float* f) {
  const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
  f[0] = (float)(lo & 0xffu);
  f[1] = (float)((lo >> 8) & 0xffu);
  f[2] = (float)((lo >> 16) & 0xffu);
  f[3] = (float)(lo >> 24);
  f[4] = (float)(hi & 0xffu);
}

template <typename PatchPtr>
SVO_DEV bool align2d_group16(const uint8_t* __restrict__ cur_img, int cols, int rows, int cur_step, PatchPtr pwb,
                             int n_iter, bool active, double* px_u, double* px_v, int* iters) {
  const int cl = threadIdx.x & 15;
  const int py = cl >> 1, x0 = (cl & 1) * 4;
  float ref_px[4], jx[4], jy[4];
  float h0 = 0, h1 = 0, h2 = 0, h4 = 0, h5 = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = (py + 1) * 10 + (x0 + k + 1);
    ref_px[k] = (float)pwb[c];
    jx[k] = (float)(0.5 * ((int)pwb[c + 1] - (int)pwb[c - 1]));
    jy[k] = (float)(0.5 * ((int)pwb[c + 10] - (int)pwb[c - 10]));
    h0 += jx[k] * jx[k]; h1 += jx[k] * jy[k]; h2 += jx[k]; h4 += jy[k] * jy[k]; h5 += jy[k];
  }
  float H[9];
  H[0] = row16_sum(h0);
  H[1] = row16_sum(h1);
  H[2] = row16_sum(h2);
  H[4] = row16_sum(h4);
  H[5] = row16_sum(h5);
  H[8] = 64.0f;
  H[3] = H[1]; H[6] = H[2]; H[7] = H[5];
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
    if (__ballot(running) == 0ull) break;                  // wave-uniform
    const int u_r = (int)floorf(u);
    const int v_r = (int)floorf(v);
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) running = false;
    if (u != u || v != v) running = false;
    float J0 = 0, J1 = 0, J2 = 0;
    if (running) {
      ++it_count;
      const float subpix_x = u - u_r;
      const float subpix_y = v - v_r;
      const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
      const float wTR = (float)(subpix_x * (1.0 - subpix_y));
      const float wBL = (float)((1.0 - subpix_x) * subpix_y);
      const float wBR = subpix_x * subpix_y;
      // 8-byte loads that stay inside the 9-pixel footprint row: the left half reads columns 0..7, the right half
      // columns 1..8 (its pixels start at byte 3)
      const uint8_t* it = cur_img + (v_r + py - 4) * cur_step + (u_r - 4) + (x0 ? 1 : 0);
      const int sh = x0 ? 24 : 0;
      const unsigned long long a = load8u(it) >> sh, b = load8u(it + cur_step) >> sh;
      float fa[5], fb[5];
      bytes5_f(a, fa);
      bytes5_f(b, fb);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a0 = fa[k], a1 = fa[k + 1];
        const float b0 = fb[k], b1 = fb[k + 1];
        const float search_pixel = wTL * a0 + wTR * a1 + wBL * b0 + wBR * b1;
        const float res = search_pixel - ref_px[k] + mean_diff;
        J0 += res * jx[k]; J1 += res * jy[k]; J2 += res;
      }
    }
    J0 = -row16_sum(J0);
    J1 = -row16_sum(J1);
    J2 = -row16_sum(J2);
    if (running) {
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

template <typename PatchPtr>
SVO_DEV bool align1d_group16(const uint8_t* __restrict__ cur_img, int cols, int rows, int cur_step, float dir0, float dir1,
                             PatchPtr pwb, int n_iter, bool active, double* px_u, double* px_v, double* h_inv,
                             int* iters) {
  const int cl = threadIdx.x & 15;
  const int py = cl >> 1, x0 = (cl & 1) * 4;
  float ref_px[4], j0[4];
  float h00 = 0, h01 = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = (py + 1) * 10 + (x0 + k + 1);
    ref_px[k] = (float)pwb[c];
    j0[k] = (float)(0.5 * (dir0 * ((int)pwb[c + 1] - (int)pwb[c - 1]) + dir1 * ((int)pwb[c + 10] - (int)pwb[c - 10])));
    h00 += j0[k] * j0[k]; h01 += j0[k];
  }
  const float H00 = row16_sum(h00);
  const float H01 = row16_sum(h01);
  const float H11 = 64.0f;
  *h_inv = 1.0 / H00 * 8 * 8;
  const float det = H00 * H11 - H01 * H01;
  const float invdet = 1.0f / det;
  const float Hi00 = H11 * invdet, Hi01 = -H01 * invdet, Hi10 = -H01 * invdet, Hi11 = H00 * invdet;
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
    if (__ballot(running) == 0ull) break;                  // wave-uniform
    const int u_r = (int)floorf(u);
    const int v_r = (int)floorf(v);
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) running = false;
    if (u != u || v != v) running = false;
    float J0 = 0, J1 = 0, c2 = 0;
    if (running) {
      ++it_count;
      const float subpix_x = u - u_r;
      const float subpix_y = v - v_r;
      const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
      const float wTR = (float)(subpix_x * (1.0 - subpix_y));
      const float wBL = (float)((1.0 - subpix_x) * subpix_y);
      const float wBR = subpix_x * subpix_y;
      const uint8_t* it = cur_img + (v_r + py - 4) * cur_step + (u_r - 4) + (x0 ? 1 : 0);
      const int sh = x0 ? 24 : 0;
      const unsigned long long a = load8u(it) >> sh, b = load8u(it + cur_step) >> sh;
      float fa[5], fb[5];
      bytes5_f(a, fa);
      bytes5_f(b, fb);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a0 = fa[k], a1 = fa[k + 1];
        const float b0 = fb[k], b1 = fb[k + 1];
        const float search_pixel = wTL * a0 + wTR * a1 + wBL * b0 + wBR * b1;
        const float res = search_pixel - ref_px[k] + mean_diff;
        J0 += res * j0[k]; J1 += res; c2 += res * res;
      }
    }
    J0 = -row16_sum(J0);
    J1 = -row16_sum(J1);
    const float new_chi2 = row16_sum(c2);
    if (running) {
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
