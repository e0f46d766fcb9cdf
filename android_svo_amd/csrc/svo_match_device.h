// svo_match_device.h -- what the stages of the matcher pipeline hand to each other (svo_depth.hip) and the per-item
// geometry of Matcher::findMatchDirect as a device function, so that the tracking chain (svo_track.hip) can form the
// records inside its own planning kernel instead of paying a launch for them.
#pragma once
#include "svo_device_math.h"

namespace svo_dev {

struct SeedRec {                 // 96 B, one per seed of the batch (device scratch owned by the context)
  double uv0[2];                 // path 1: B - step (epipolar abscissa of step 0); path 0: px midpoint (level 0)
  double step[2];                // path 1: epi_dir / n_steps; after stage S: px_cur (level 0) of the match
  float a00, a01, a10, a11;      // inverse affine warp (A_cur_ref^-1 cast to f32)
  float prx, pry;                // ref px on its pyramid level
  float z_inv_min;               // for the NaN test of depth_filter.cpp:333
  int n_steps;                   // epi_length / 0.7 (before the ++ of matcher.cpp:297)
  int search_level;
  int path;                      // 0 direct align, 1 epipolar search, 2 no search (too long), 3 align1D, -1 seed not live
  int status;                    // pre-status for non-live seeds
  int warp_nan;
  int matched;                   // stage S: align2D converged
  int n_zmssd, n_align;
  int pad;                       // findMatchDirect: pyramid slot of the reference keyframe
};
static_assert(sizeof(SeedRec) == 96, "six 16-byte words per record");

// ---- Matcher::findMatchDirect over n (map point, reference feature) pairs (S/matcher.cpp:156-202) ----
struct MdFrame {
  Cam cam;
  double T_cur_w[7];
  int n_pyr_levels;
  int n_kf, n_ref_levels;      // valid ranges of the caller's kf_slot / level values
  int slot_base;               // added to an item's slot in its record: where the caller's keyframe slots start in the pyramid batch
                               // the warp stage reads (a tracker group keeps all cameras' keyframes in one batch); 0 otherwise
};

// a record no stage touches (items past the count, candidates the matcher rejects at once)
SVO_DEV SeedRec md_dead_record() {
  SeedRec rc;
  rc.uv0[0] = rc.uv0[1] = 0.0;
  rc.step[0] = rc.step[1] = 0.0;
  rc.a00 = rc.a01 = rc.a10 = rc.a11 = rc.prx = rc.pry = 0.0f; rc.z_inv_min = 0.0f;
  rc.n_steps = 0; rc.search_level = 0; rc.path = -1; rc.status = 0; rc.warp_nan = 0;
  rc.matched = 0; rc.n_zmssd = 0; rc.n_align = 0; rc.pad = 0;
  return rc;
}

// one item: frame test, depth, affine warp, search level -> SeedRec (path 0 = align2D, 3 = align1D, -1 = rejected);
// T_ref_w: keyframe poses indexed by pyramid slot
SVO_DEV SeedRec md_geometry_item(const MdFrame& fr, const double* __restrict__ T_ref_w, int slot, int level_ref, const double* pr,
                                 const double* fi, const double* pt_pos, bool edgelet, const double* grad, const double* px_cur) {
  const Cam& cam = fr.cam;
  SeedRec rc = md_dead_record();
  rc.uv0[0] = px_cur[0]; rc.uv0[1] = px_cur[1];
  rc.pad = slot + fr.slot_base;
  // isInFrame(px.cast<int>()/(1<<level), halfpatch_size_+2, level) (:164-166)
  // a slot or level outside the pyramids the caller handed over is rejected like a failed frame test (never indexed)
  const bool in_range = slot >= 0 && slot < fr.n_kf && level_ref >= 0 && level_ref < fr.n_ref_levels;
  const int ox = in_range ? (int)pr[0] / (1 << level_ref) : -1, oy = in_range ? (int)pr[1] / (1 << level_ref) : -1;
  if (in_range && is_in_frame_level(cam, ox, oy, 6, level_ref)) {
    const double* Tr = T_ref_w + 7 * (size_t)slot;
    double T_ref_inv[7], T_cur_ref[7];
    se3_inverse(Tr, T_ref_inv);
    se3_mul(fr.T_cur_w, T_ref_inv, T_cur_ref);
    const double dx = T_ref_inv[0] - pt_pos[0], dy = T_ref_inv[1] - pt_pos[1], dz = T_ref_inv[2] - pt_pos[2];
    const double depth = sqrt(dx * dx + dy * dy + dz * dz);
    double Acr[4];
    get_warp_matrix_affine(cam, pr, fi, depth, T_cur_ref, level_ref, Acr);
    int search_level = 0;
    {
      double D = Acr[0] * Acr[3] - Acr[2] * Acr[1];
      while (D > 3.0 && search_level < fr.n_pyr_levels - 1) { search_level += 1; D *= 0.25; }
    }
    rc.search_level = search_level;
    const double det = Acr[0] * Acr[3] - Acr[2] * Acr[1];
    const double invdet = 1.0 / det;
    rc.a00 = (float)(Acr[3] * invdet); rc.a01 = (float)(-Acr[1] * invdet);
    rc.a10 = (float)(-Acr[2] * invdet); rc.a11 = (float)(Acr[0] * invdet);
    rc.warp_nan = rc.a00 != rc.a00;
    rc.prx = (float)pr[0] / (1 << level_ref);
    rc.pry = (float)pr[1] / (1 << level_ref);
    rc.path = 0;
    if (edgelet) {
      double d0 = Acr[0] * grad[0] + Acr[1] * grad[1];
      double d1 = Acr[2] * grad[0] + Acr[3] * grad[1];
      const double n2 = d0 * d0 + d1 * d1;
      if (n2 > 0.0) { const double nn = sqrt(n2); d0 = d0 / nn; d1 = d1 / nn; }
      rc.step[0] = (double)(float)d0; rc.step[1] = (double)(float)d1;
      rc.path = 3;
    }
  }
  return rc;
}

}  // namespace svo_dev
