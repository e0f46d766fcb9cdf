// feature_alignment_hip.h -- batched form of svo::feature_alignment::align2D
// (I/feature_alignment.h:40-47) for callers that hold many patches at once (Reprojector cells,
// SURVEY 8f-2).  The single-patch signature of the reference stays on the CPU: a GPU round trip
// per 8x8 patch would be slower than the 0.3 us scalar call it replaces.
#ifndef SVO_FEATURE_ALIGNMENT_HIP_H_
#define SVO_FEATURE_ALIGNMENT_HIP_H_

#include <vector>

#include <svo/global.h>

#include "svo_hip_bridge.h"

namespace svo {
namespace feature_alignment {

/// n patches on pyramid level `level` of `cur_frame`: ref_patch_with_border [n][100], in/out
/// cur_px_estimate[n] (level coordinates); returns per-patch convergence flags.
inline std::vector<bool> align2D_batch(hip_bridge::Context& ctx, hip_bridge::PyramidCache& pyr, const Frame& cur_frame,
                                       int level, const std::vector<uint8_t>& ref_patch_with_border, const int n_iter,
                                       std::vector<Vector2d>& cur_px_estimate) {
  const int n = (int)cur_px_estimate.size();
  std::vector<bool> out((size_t)n, false);
  if (n == 0) return out;
  if (!ctx.ok()) { hip_bridge::reportDeviceFailure(NULL, "feature_alignment::align2D_batch"); return out; }
  const int slot = pyr.slotOf(cur_frame);
  if (slot < 0) return out;
  std::vector<double> px(2 * (size_t)n);
  std::vector<uint8_t> conv((size_t)n, 0);
  for (int i = 0; i < n; ++i) { px[2 * i] = cur_px_estimate[i][0]; px[2 * i + 1] = cur_px_estimate[i][1]; }
  if (svo_hip_align2d_batch(ctx.get(), pyr.pyramid(), slot, level, n, ref_patch_with_border.data(), NULL, n_iter,
                            px.data(), conv.data(), NULL) != SVO_HIP_OK)
    return out;
  for (int i = 0; i < n; ++i) {
    cur_px_estimate[i] = Vector2d(px[2 * i], px[2 * i + 1]);
    out[i] = conv[i] != 0;
  }
  return out;
}

}  // namespace feature_alignment
}  // namespace svo

#endif  // SVO_FEATURE_ALIGNMENT_HIP_H_
