// svo_align_device.h -- feature_alignment::align2D / align1D as device routines, FOUR LANES (one DPP quad) PER PATCH.
//
// Follows S/feature_alignment.cpp:167-281 (the scalar path, canonical on arm64/x86) and :35-152:
// inverse-compositional LK with a mean-offset parameter, f32 throughout, min_update^2 = 0.5^2 for align2D (this
// port's value, :203) and 0.03^2 for align1D, border test before the NaN test (:211-215), `break` on the border
// leaves converged = false but still writes u,v; align1D has the chi2-increase rollback (:117-125, which subtracts
// update[0] from u and update[1] from v as the reference does) and h_inv = 1/H(0,0) * 64 (:63).
//
// Result-identical to the CPU path by construction.  The reference accumulates Jres (and align1D's H(0,0) and chi2)
// as ONE 64-term serial f32 sum over the pixels, y-then-x; any tree reduction rounds differently and flips
// `converged` on ~1 % of the patches.  Here lane q of a quad owns pixel rows 2q and 2q+1 (pixels 16q .. 16q+15 of the
// serial order): it interpolates its 16 pixels and forms their residual products, then the running sums travel
// through the quad in pixel order -- phase t: every lane continues from the value its left neighbour held after phase
// t-1 (one DPP quad_perm move per sum) and folds its own 16 terms in; after phase t lane t holds the exact prefix sum
// of pixels 0 .. 16t+15, after phase 3 lane 3 holds the reference's Jres, which a quad broadcast hands to all four.
// Every f32 operation is the reference's, in the reference's order (built with -ffp-contract=off: no fused
// multiply-add), so `converged`, the refined pixel and the iteration count are equal bit for bit.
//
// Why four lanes: the chain costs 4 x 16 x 3 additions per iteration whatever the split (64 x 3 for one lane per
// patch), the per-pixel work 11 instructions per pixel whatever the split; a quad keeps 16 patches per wavefront
// (the slowest patch of a wave sets its trip count: 16-wide divergence instead of 64-wide), ~110 VGPRs (4+ waves per
// SIMD instead of 2) and 4x the waves for a small batch (5000 patches: 313 waves instead of 79).  One lane per patch
// measured 31 us / 70 us for 5 000 / 200 000 patches (profiles/r02_align_lane_pmc.txt), two rounds of waves bounded
// by their slowest lane.
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

constexpr int ALIGN_LANES_PER_PATCH = 4;

// What lane q of a quad keeps of its patch: rows 2q .. 2q+3 of the 10x10 ref_patch_with_border (bytes 20q .. 20q+39 =
// ten little-endian words) and rows 2q, 2q+1 of the 8x8 ref_patch (bytes 16q .. 16q+15 = four words).
struct QuadPatch {
  uint32_t b[10];   // byte k = ref_patch_with_border[20q + k]
  uint32_t p[4];    // byte k = ref_patch[16q + k]
};

// byte k (compile-time constant after unrolling) of a packed word array
#define SVO_BYTE(arr, k) (((arr)[(k) >> 2] >> (8 * ((k) & 3))) & 0xffu)

// createPatchFromPatchWithBorder (S/matcher.cpp:138-147): the lane's two rows of the interior of the bordered patch
SVO_DEV void patch_from_border(QuadPatch& qp) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = 4 * w + k, y = r >> 3, x = r & 7;          // local pixel (row y of the lane's two, column x)
      v |= SVO_BYTE(qp.b, (y + 1) * 10 + (x + 1)) << (8 * k);
    }
    qp.p[w] = v;
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

// value of the left neighbour in the quad (lane 0 gets its own: never used)
SVO_DEV float quad_from_left(float v) { return dpp_quad<0x90>(v); }

// The serial 64-term sums of the reference: s[j] -= t[j][k] (SUB) or s[j] += t[j][k] over the quad's pixels in pixel
// order.  On return every lane holds the totals.
template <int NSUM, bool SUB>
SVO_DEV void quad_serial_sums(const float (&t)[NSUM][16], float (&s)[NSUM]) {
#pragma unroll
  for (int phase = 0; phase < 4; ++phase) {
    float a[NSUM];
#pragma unroll
    for (int j = 0; j < NSUM; ++j) a[j] = phase == 0 ? 0.0f : quad_from_left(s[j]);
    // the NSUM chains are independent: interleaved term by term so that consecutive instructions never depend on each
    // other (a dependent f32 add has several cycles of latency; one chain after the other exposed all of it)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
      for (int j = 0; j < NSUM; ++j) a[j] = SUB ? a[j] - t[j][k] : a[j] + t[j][k];
      // (a scheduling barrier per term pins the interleaving: the scheduler otherwise emits one chain after the other
      // to save registers; an empty asm would do too but costs an s_nop per use)
      if (NSUM > 1) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NSUM; ++j) s[j] = a[j];
  }
#pragma unroll
  for (int j = 0; j < NSUM; ++j) s[j] = quad_bcast<3>(s[j]);
}

// The lane's 16 interpolated residuals (rows 2q, 2q+1 of the patch) at integer position (u_r, v_r) with the bilinear
// weights of the reference; `base` = cur_img + (v_r - 4) * cur_step + (u_r - 4).
SVO_DEV void quad_residuals(const uint8_t* __restrict__ base, int cur_step, int q, const float (&pf)[16], float wTL, float wTR,
                            float wBL, float wBR, float mean_diff, float (&res)[16]) {
  // image rows 2q, 2q+1, 2q+2 of the 9-row footprint: the third one is the right neighbour's first, so only lane 3
  // loads it (row 8); the others take it through the quad -- 2 1/4 scattered loads per lane instead of 3
  uint32_t rw[3][3];
  const uint8_t* row = base + (2 * q) * cur_step;
  load_row12(row, rw[0]);
  load_row12(row + cur_step, rw[1]);
  uint32_t last[3] = {0u, 0u, 0u};
  if (q == 3) load_row12(row + 2 * cur_step, last);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint32_t from_right = (uint32_t)dpp_quad<0xF9>((int)rw[0][k]);      // quad_perm [1,2,3,3]
    rw[2][k] = q == 3 ? last[k] : from_right;
  }
  float f[3][9];
#pragma unroll
  for (int y = 0; y < 3; ++y) row9_f(rw[y], f[y]);
#pragma unroll
  for (int y = 0; y < 2; ++y) {
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int r = y * 8 + x;
      const float search_pixel = wTL * f[y][x] + wTR * f[y][x + 1] + wBL * f[y + 1][x] + wBR * f[y + 1][x + 1];
      res[r] = search_pixel - pf[r] + mean_diff;
    }
  }
}

// the lane's 16 reference-patch pixels as floats: the same in every iteration (converting them inside the loop cost 16
// v_cvt_f32_ubyte per iteration, which issue at the f64 rate)
SVO_DEV void patch_floats(const QuadPatch& qp, float (&pf)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) pf[r] = (float)SVO_BYTE(qp.p, r);
}

// feature_alignment::align2D for the patch of this quad.  All four lanes of a quad must call it with the same
// arguments except qp (their own rows); `active` quads run; the loop is left when no lane of the wave runs any more.
// Returns `converged` (the same on the four lanes); *px_u, *px_v are rewritten as the reference rewrites
// cur_px_estimate (always, also on failure); *iters = iterations that reached the pixel loop.
SVO_DEV bool align2d_quad(const uint8_t* __restrict__ cur_img, int cols, int rows, int cur_step, const QuadPatch& qp,
                          int n_iter, bool active, double* px_u, double* px_v, int* iters) {
  const int q = threadIdx.x & 3;
  float pf[16];
  patch_floats(qp, pf);
  float jx[16], jy[16];
  float H[9];
  {
    float h0 = 0, h1 = 0, h2 = 0, h4 = 0, h5 = 0;     // exact in f32 in any order: multiples of 1/4 below 2^22
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int y = r >> 3, x = r & 7, c = (y + 1) * 10 + (x + 1);
      jx[r] = (float)(0.5 * ((int)SVO_BYTE(qp.b, c + 1) - (int)SVO_BYTE(qp.b, c - 1)));
      jy[r] = (float)(0.5 * ((int)SVO_BYTE(qp.b, c + 10) - (int)SVO_BYTE(qp.b, c - 10)));
      h0 += jx[r] * jx[r]; h1 += jx[r] * jy[r]; h2 += jx[r]; h4 += jy[r] * jy[r]; h5 += jy[r];
    }
    h0 = quad_sum(h0); h1 = quad_sum(h1); h2 = quad_sum(h2); h4 = quad_sum(h4); h5 = quad_sum(h5);
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
    if (running) {                                                                     // uniform within a quad
      ++it_count;
      const float subpix_x = u - u_r;
      const float subpix_y = v - v_r;
      const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
      const float wTR = (float)(subpix_x * (1.0 - subpix_y));
      const float wBL = (float)((1.0 - subpix_x) * subpix_y);
      const float wBR = subpix_x * subpix_y;
      float t[3][16];
      quad_residuals(cur_img + (v_r - 4) * cur_step + (u_r - 4), cur_step, q, pf, wTL, wTR, wBL, wBR, mean_diff, t[2]);
#pragma unroll
      for (int r = 0; r < 16; ++r) { t[0][r] = t[2][r] * jx[r]; t[1][r] = t[2][r] * jy[r]; }
      float J[3];
      quad_serial_sums<3, true>(t, J);                       // Jres[k] -= ... over the 64 pixels in order (:233-242)
      const float up0 = Hinv[0] * J[0] + (Hinv[1] * J[1] + Hinv[2] * J[2]);
      const float up1 = Hinv[3] * J[0] + (Hinv[4] * J[1] + Hinv[5] * J[2]);
      const float up2 = Hinv[6] * J[0] + (Hinv[7] * J[1] + Hinv[8] * J[2]);
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

// feature_alignment::align1D for the patch of this quad (S/feature_alignment.cpp:35-152)
SVO_DEV bool align1d_quad(const uint8_t* __restrict__ cur_img, int cols, int rows, int cur_step, float dir0, float dir1,
                          const QuadPatch& qp, int n_iter, bool active, double* px_u, double* px_v, double* h_inv,
                          int* iters) {
  const int q = threadIdx.x & 3;
  float pf[16];
  patch_floats(qp, pf);
  float dv[16];
  float H00, H01;
  {
    float t[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int y = r >> 3, x = r & 7, c = (y + 1) * 10 + (x + 1);
      dv[r] = (float)(0.5 * (dir0 * ((int)SVO_BYTE(qp.b, c + 1) - (int)SVO_BYTE(qp.b, c - 1)) +
                             dir1 * ((int)SVO_BYTE(qp.b, c + 10) - (int)SVO_BYTE(qp.b, c - 10))));
      t[0][r] = dv[r] * dv[r];
      t[1][r] = dv[r];                                      // J[0] * J[1], J[1] = 1
    }
    float s[2];
    quad_serial_sums<2, false>(t, s);                       // H += J*J^T, serial (:66): not exact, the order matters
    H00 = s[0]; H01 = s[1];
  }
  const float H10 = H01, H11 = 64.0f;
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
      float res[16];
      quad_residuals(cur_img + (v_r - 4) * cur_step + (u_r - 4), cur_step, q, pf, wTL, wTR, wBL, wBR, mean_diff, res);
      float t[2][16], c[1][16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { t[0][r] = res[r] * dv[r]; t[1][r] = res[r]; c[0][r] = res[r] * res[r]; }
      float J[2], nc[1];
      quad_serial_sums<2, true>(t, J);                      // Jres -= ...
      quad_serial_sums<1, false>(c, nc);                    // new_chi2 += res*res
      const float new_chi2 = nc[0];
      if (iter > 0 && new_chi2 > chi2) {
        u -= up0;
        v -= up1;
        running = false;
      } else {
        chi2 = new_chi2;
        up0 = Hi00 * J[0] + Hi01 * J[1];
        up1 = Hi10 * J[0] + Hi11 * J[1];
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
