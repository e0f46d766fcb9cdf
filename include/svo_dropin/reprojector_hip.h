// reprojector_hip.h -- the cell loop of Reprojector::reprojectMap (S/reprojector.cpp:149-166 with reprojectCell
// :180-241) on the device, as ONE function the reference's own reprojectMap calls in place of that loop
// (INTEGRATION.md shows the edit; include/svo_dropin/reprojector.patch is the diff, `make -C oracle dropin-check`
// applies it to a scratch copy and compiles it).  Nothing of the Reprojector class is re-typed here: constructor,
// grid set-up, keyframe selection, reprojectPoint and the candidate bookkeeping stay the reference's code.
//
// The reference tries the candidates of a cell one by one (Matcher::findMatchDirect per trial) until one matches.
// Here every candidate of every cell is matched in one device batch (svo_hip_reproject_cells), then the serial policy
// is replayed over the results with the same side effects on the points, the map and the frame, in the same order.
//
// Grid, Cell and Candidate are private types of Reprojector: the function is a template so that the member function
// that calls it can hand over grid_ without this header naming them.
#ifndef SVO_REPROJECTOR_HIP_H_
#define SVO_REPROJECTOR_HIP_H_

#include <map>
#include <vector>

#include <svo/config.h>
#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/map.h>
#include <svo/matcher.h>
#include <svo/point.h>

#include "svo_hip_bridge.h"

namespace svo {
namespace hip_bridge {

struct ReprojectDevice {
  Context ctx;
  PyramidCache kf_pyr, cur_pyr;
  ReprojectDevice() : ctx(0), kf_pyr(ctx.get(), 16), cur_pyr(ctx.get(), 1) {}
};
inline ReprojectDevice& reprojectDevice() {
  static thread_local ReprojectDevice d;       // the tracking thread's own context (stream)
  return d;
}

/// Returns false (after logging; n_matches / n_trials untouched, no candidate consumed) only when the device call
/// fails: the caller decides what a frame without reprojected features means (the reference: tracking failure).
template <class Grid, class Compare>
bool reprojectCellsHip(Grid& grid, const FramePtr& frame, Map& map, Compare point_quality_comparator,
                       size_t& n_matches_out, size_t& n_trials_out) {
  typedef typename std::remove_pointer<typename std::remove_reference<decltype(grid.cells[0])>::type>::type Cell;
  typedef typename Cell::iterator CellIt;
  ReprojectDevice& dev = reprojectDevice();
  const size_t n_cells = grid.cells.size();
  std::vector<int32_t> cell_offset(n_cells + 1, 0), kf_index, level_ref;
  std::vector<double> px_ref, f_ref, pt_pos, px_cur, grad;
  std::vector<uint8_t> edgelet, deleted;
  std::vector<CellIt> cand_it;
  std::vector<Feature*> cand_ref;
  std::map<int, int> index_of_frame;                          // frame id -> index into kf_frames
  std::vector<const Frame*> kf_frames;
  for (size_t i = 0; i < n_cells; ++i) {
    Cell& cell = *grid.cells.at(grid.cell_order[i]);
    cell.sort(point_quality_comparator);                      // reprojectCell :183
    for (CellIt it = cell.begin(); it != cell.end(); ++it) {
      Feature* ref_ftr = NULL;
      // Matcher::findMatchDirect starts with getCloseViewObs (matcher.cpp:161-162): a point without a usable
      // observation fails there; it is sent along flagged as deleted and counted as a failure below
      const bool usable = it->pt->type_ != Point::TYPE_DELETED && it->pt->getCloseViewObs(frame->pos(), ref_ftr);
      const Feature* r = usable ? ref_ftr : NULL;
      cand_it.push_back(it);
      cand_ref.push_back(usable ? ref_ftr : NULL);
      deleted.push_back(usable ? 0 : 1);
      int k = 0;
      if (r) {
        std::map<int, int>::iterator f = index_of_frame.find(r->frame->id_);
        if (f == index_of_frame.end()) {
          k = (int)kf_frames.size();
          index_of_frame[r->frame->id_] = k;
          kf_frames.push_back(r->frame);
        } else {
          k = f->second;
        }
      }
      kf_index.push_back(k);
      level_ref.push_back(r ? r->level : 0);
      px_ref.push_back(r ? r->px[0] : 0.0); px_ref.push_back(r ? r->px[1] : 0.0);
      for (int c = 0; c < 3; ++c) { f_ref.push_back(r ? r->f[c] : 0.0); pt_pos.push_back(it->pt->pos_[c]); }
      px_cur.push_back(it->px[0]); px_cur.push_back(it->px[1]);
      edgelet.push_back(r && r->type == Feature::EDGELET ? 1 : 0);
      grad.push_back(r ? r->grad[0] : 1.0); grad.push_back(r ? r->grad[1] : 0.0);
    }
    cell_offset[i + 1] = (int32_t)cand_it.size();
  }
  const size_t n_cand = cand_it.size();
  std::vector<uint8_t> tried(n_cand, 0), matched(n_cand, 0);
  std::vector<int32_t> search_level(n_cand, 0), cell_winner(n_cells, -1);
  uint64_t n_matches = 0, n_trials = 0;
  if (n_cand > 0) {
    // every keyframe a candidate refers to must be resident at once, however many there are (the cache grows);
    // candidates address them by cache slot, poses are laid out by slot
    std::vector<int> slot_of_index;
    bool ok = dev.ctx.ok() && dev.kf_pyr.acquire(kf_frames, slot_of_index);
    const int cur_slot = ok ? dev.cur_pyr.slotOf(*frame) : -1;
    ok = ok && cur_slot >= 0;
    if (ok) {
      const int n_slots = dev.kf_pyr.capacity();
      std::vector<double> T_kf_w(7 * (size_t)n_slots, 0.0);
      for (int s = 0; s < n_slots; ++s) T_kf_w[7 * (size_t)s + 6] = 1.0;
      for (size_t k = 0; k < kf_frames.size(); ++k) toPose7(kf_frames[k]->T_f_w_, &T_kf_w[7 * (size_t)slot_of_index[k]]);
      std::vector<int32_t> kf_slot(n_cand);
      for (size_t c = 0; c < n_cand; ++c) kf_slot[c] = kf_frames.empty() ? 0 : slot_of_index[(size_t)kf_index[c]];
      const svo_hip_camera cam = toCamera(frame->cam_);
      double T_cur[7];
      toPose7(frame->T_f_w_, T_cur);
      ok = svo_hip_reproject_cells(dev.ctx.get(), dev.kf_pyr.pyramid(), dev.cur_pyr.pyramid(), cur_slot, &cam, n_slots,
                                   T_kf_w.data(), T_cur, (int)n_cells, cell_offset.data(), kf_slot.data(), px_ref.data(),
                                   f_ref.data(), level_ref.data(), pt_pos.data(), edgelet.data(), grad.data(), deleted.data(),
                                   px_cur.data(), (int)Config::maxFts(), (int)Config::nPyrLevels(), 10, tried.data(),
                                   matched.data(), search_level.data(), cell_winner.data(), &n_matches, &n_trials) == SVO_HIP_OK;
    }
    if (!ok) {
      SVO_ERROR_STREAM("Reprojector: svo_hip_reproject_cells failed (svo_hip_last_error has the reason)");   // one-argument form: valid for both definitions of the macro
      return false;
    }
  }
  n_matches_out = (size_t)n_matches;
  n_trials_out = (size_t)n_trials;

  // ---- side effects of reprojectCell (:188-236) for every visited candidate, in the reference's order
  for (size_t i = 0; i < n_cells; ++i) {
    Cell& cell = *grid.cells.at(grid.cell_order[i]);
    for (int32_t c = cell_offset[i]; c < cell_offset[i + 1]; ++c) {
      if (!tried[c]) continue;
      CellIt it = cand_it[c];
      Point* pt = it->pt;
      if (pt->type_ == Point::TYPE_DELETED) { cell.erase(it); continue; }                      // :190-194
      if (!matched[c]) {                                                                        // :202-211
        pt->n_failed_reproj_++;
        if (pt->type_ == Point::TYPE_UNKNOWN && pt->n_failed_reproj_ > 15) map.safeDeletePoint(pt);
        if (pt->type_ == Point::TYPE_CANDIDATE && pt->n_failed_reproj_ > 30) map.point_candidates_.deleteCandidatePoint(pt);
        cell.erase(it);
        continue;
      }
      pt->n_succeeded_reproj_++;                                                                // :214-216
      if (pt->type_ == Point::TYPE_UNKNOWN && pt->n_succeeded_reproj_ > 10) pt->type_ = Point::TYPE_GOOD;
      Feature* new_feature = new Feature(frame.get(), Vector2d(px_cur[2 * c], px_cur[2 * c + 1]), search_level[c]);
      frame->addFeature(new_feature);
      new_feature->point = pt;
      const Feature* ref_ftr = cand_ref[c];
      if (ref_ftr->type == Feature::EDGELET) {                                                  // :226-231
        Matrix2d A_cur_ref;
        warp::getWarpMatrixAffine(*ref_ftr->frame->cam_, *frame->cam_, ref_ftr->px, ref_ftr->f,
                                  (ref_ftr->frame->pos() - pt->pos_).norm(), frame->T_f_w_ * ref_ftr->frame->T_f_w_.inverse(),
                                  ref_ftr->level, A_cur_ref);
        new_feature->type = Feature::EDGELET;
        new_feature->grad = A_cur_ref * ref_ftr->grad;
        new_feature->grad.normalize();
      }
      cell.erase(it);
    }
  }
  return true;
}

}  // namespace hip_bridge
}  // namespace svo

#endif  // SVO_REPROJECTOR_HIP_H_
