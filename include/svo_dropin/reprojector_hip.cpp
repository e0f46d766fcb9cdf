// reprojector_hip.cpp -- drop-in replacement of svo/reprojector.cpp.
//
// Same class (I/reprojector.h:38-108), same reprojectMap() flow.  The first half -- choosing the close keyframes and
// dropping their map points and the point candidates into the grid cells (reprojector.cpp:75-145) -- is host
// bookkeeping and stays as it is.  The second half -- for every cell, try the candidates in quality order until one
// matches (reprojector.cpp:149-166, reprojectCell :180-241) -- called Matcher::findMatchDirect once per trial; here
// all candidates of all cells are matched in ONE device batch (svo_hip_reproject_cells) and the serial policy is
// replayed over the results, with the same side effects on the points, the map and the frame.
// Compile this file INSTEAD of reprojector.cpp.
#include <algorithm>
#include <map>
#include <vector>

#include <svo/config.h>
#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/map.h>
#include <svo/matcher.h>
#include <svo/point.h>
#include <svo/reprojector.h>

#include "svo_hip_bridge.h"

namespace svo {

namespace {
struct ReprojectDevice {
  hip_bridge::Context ctx;
  hip_bridge::PyramidCache kf_pyr, cur_pyr;
  ReprojectDevice() : ctx(0), kf_pyr(ctx.get(), 16), cur_pyr(ctx.get(), 1) {}
};
ReprojectDevice& device() {
  static thread_local ReprojectDevice d;
  return d;
}
}  // namespace

Reprojector::Reprojector(vk::AbstractCamera* cam, Map& map) : map_(map) { initializeGrid(cam); }

Reprojector::~Reprojector() {
  std::for_each(grid_.cells.begin(), grid_.cells.end(), [&](Cell* c) { delete c; });
}

void Reprojector::initializeGrid(vk::AbstractCamera* cam) {
  grid_.cell_size = Config::gridSize();
  grid_.grid_n_cols = ceil(static_cast<double>(cam->width()) / grid_.cell_size);
  grid_.grid_n_rows = ceil(static_cast<double>(cam->height()) / grid_.cell_size);
  grid_.cells.resize(grid_.grid_n_cols * grid_.grid_n_rows);
  std::for_each(grid_.cells.begin(), grid_.cells.end(), [&](Cell*& c) { c = new Cell; });
  grid_.cell_order.resize(grid_.cells.size());
  for (size_t i = 0; i < grid_.cells.size(); ++i) grid_.cell_order[i] = i;
}

void Reprojector::resetGrid() {
  n_matches_ = 0;
  n_trials_ = 0;
  std::for_each(grid_.cells.begin(), grid_.cells.end(), [&](Cell* c) { c->clear(); });
}

bool Reprojector::pointQualityComparator(Candidate& lhs, Candidate& rhs) {
  if (lhs.pt->type_ > rhs.pt->type_) return true;
  return false;
}

bool Reprojector::reprojectPoint(FramePtr frame, Point* point) {
  Vector2d px(frame->w2c(point->pos_));
  if (frame->cam_->isInFrame(px.cast<int>(), 8)) {           // 8px is the patch size in the matcher
    const int k = static_cast<int>(px[1] / grid_.cell_size) * grid_.grid_n_cols + static_cast<int>(px[0] / grid_.cell_size);
    grid_.cells.at(k)->push_back(Candidate(point, px));
    return true;
  }
  return false;
}

// kept for completeness of the class; reprojectMap below does not call it
bool Reprojector::reprojectCell(Cell& cell, FramePtr frame) {
  (void)cell; (void)frame;
  return false;
}

void Reprojector::reprojectMap(FramePtr frame, std::vector<std::pair<FramePtr, std::size_t> >& overlap_kfs) {
  resetGrid();

  // ---- first half, unchanged (reprojector.cpp:75-145)
  std::list<std::pair<FramePtr, double> > close_kfs;
  map_.getCloseKeyframes(frame, close_kfs);
  close_kfs.sort([](const std::pair<FramePtr, double>& l, const std::pair<FramePtr, double>& r) { return l.second < r.second; });
  size_t n = 0;
  overlap_kfs.reserve(options_.max_n_kfs);
  for (auto it_frame = close_kfs.begin(), ite_frame = close_kfs.end(); it_frame != ite_frame && n < options_.max_n_kfs; ++it_frame, ++n) {
    FramePtr ref_frame = it_frame->first;
    overlap_kfs.push_back(std::pair<FramePtr, size_t>(ref_frame, 0));
    for (auto it_ftr = ref_frame->fts_.begin(), ite_ftr = ref_frame->fts_.end(); it_ftr != ite_ftr; ++it_ftr) {
      if ((*it_ftr)->point == NULL) continue;
      if ((*it_ftr)->point->last_projected_kf_id_ == frame->id_) continue;
      (*it_ftr)->point->last_projected_kf_id_ = frame->id_;
      if (reprojectPoint(frame, (*it_ftr)->point)) overlap_kfs.back().second++;
    }
  }
  {
    std::unique_lock<std::mutex> lock(map_.point_candidates_.mut_);
    auto it = map_.point_candidates_.candidates_.begin();
    while (it != map_.point_candidates_.candidates_.end()) {
      if (!reprojectPoint(frame, it->first)) {
        it->first->n_failed_reproj_ += 3;
        if (it->first->n_failed_reproj_ > 30) {
          map_.point_candidates_.deleteCandidate(*it);
          it = map_.point_candidates_.candidates_.erase(it);
          continue;
        }
      }
      ++it;
    }
  }

  // ---- second half: one batch instead of one findMatchDirect per trial
  ReprojectDevice& dev = device();
  const size_t n_cells = grid_.cells.size();
  std::vector<int32_t> cell_offset(n_cells + 1, 0), kf_slot, level_ref;
  std::vector<double> px_ref, f_ref, pt_pos, px_cur, grad, T_kf_w;
  std::vector<uint8_t> edgelet, deleted;
  std::vector<Cell::iterator> cand_it;
  std::vector<Feature*> cand_ref;
  std::map<int, int> slot_of_frame;                           // frame id -> row of T_kf_w / pyramid slot
  std::vector<const Frame*> kf_frames;
  for (size_t i = 0; i < n_cells; ++i) {
    Cell& cell = *grid_.cells.at(grid_.cell_order[i]);
    cell.sort(&Reprojector::pointQualityComparator);          // reprojectCell :183
    for (Cell::iterator it = cell.begin(); it != cell.end(); ++it) {
      Feature* ref_ftr = NULL;
      // Matcher::findMatchDirect starts with getCloseViewObs (matcher.cpp:161-162): a point without a usable
      // observation fails there; it is sent to the device flagged as deleted and counted as a failure below
      const bool usable = it->pt->type_ != Point::TYPE_DELETED && it->pt->getCloseViewObs(frame->pos(), ref_ftr);
      cand_it.push_back(it);
      cand_ref.push_back(usable ? ref_ftr : NULL);
      deleted.push_back(usable ? 0 : 1);
      const Feature* r = usable ? ref_ftr : NULL;
      int slot = 0;
      if (r) {
        auto f = slot_of_frame.find(r->frame->id_);
        if (f == slot_of_frame.end()) {
          slot = (int)kf_frames.size();
          slot_of_frame[r->frame->id_] = slot;
          kf_frames.push_back(r->frame);
          double T[7];
          hip_bridge::toPose7(r->frame->T_f_w_, T);
          T_kf_w.insert(T_kf_w.end(), T, T + 7);
        } else {
          slot = f->second;
        }
      }
      kf_slot.push_back(slot);
      level_ref.push_back(r ? r->level : 0);
      px_ref.push_back(r ? r->px[0] : 0.0); px_ref.push_back(r ? r->px[1] : 0.0);
      for (int k = 0; k < 3; ++k) { f_ref.push_back(r ? r->f[k] : 0.0); pt_pos.push_back(it->pt->pos_[k]); }
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
  bool done = n_cand == 0;
  if (!done && dev.ctx.ok() && options_.find_match_direct && (int)kf_frames.size() <= 16) {
    // keyframe pyramids: slot k of the cache must hold kf_frames[k]
    bool slots_ok = true;
    for (size_t k = 0; k < kf_frames.size() && slots_ok; ++k) slots_ok = dev.kf_pyr.slotOfAt(*kf_frames[k], (int)k) == (int)k;
    const int cur_slot = dev.cur_pyr.slotOf(*frame);
    const svo_hip_camera cam = hip_bridge::toCamera(frame->cam_);
    double T_cur[7];
    hip_bridge::toPose7(frame->T_f_w_, T_cur);
    if (slots_ok && cur_slot >= 0 &&
        svo_hip_reproject_cells(dev.ctx.get(), dev.kf_pyr.pyramid(), dev.cur_pyr.pyramid(), cur_slot, &cam, (int)kf_frames.size(),
                                T_kf_w.data(), T_cur, (int)n_cells, cell_offset.data(), kf_slot.data(), px_ref.data(),
                                f_ref.data(), level_ref.data(), pt_pos.data(), edgelet.data(), grad.data(), deleted.data(),
                                px_cur.data(), (int)Config::maxFts(), (int)Config::nPyrLevels(), 10, tried.data(),
                                matched.data(), search_level.data(), cell_winner.data(), &n_matches, &n_trials) == SVO_HIP_OK)
      done = true;
  }
  if (!done) {
    SVO_WARN_STREAM("Reprojector: device unavailable, no features reprojected");
    return;
  }
  n_matches_ = (size_t)n_matches;
  n_trials_ = (size_t)n_trials;

  // ---- side effects of reprojectCell (:188-236) for every visited candidate, in the reference's order
  for (size_t i = 0; i < n_cells; ++i) {
    Cell& cell = *grid_.cells.at(grid_.cell_order[i]);
    for (int32_t c = cell_offset[i]; c < cell_offset[i + 1]; ++c) {
      if (!tried[c]) continue;
      Cell::iterator it = cand_it[c];
      Point* pt = it->pt;
      if (pt->type_ == Point::TYPE_DELETED) { cell.erase(it); continue; }                      // :190-194
      if (!matched[c]) {                                                                        // :202-211
        pt->n_failed_reproj_++;
        if (pt->type_ == Point::TYPE_UNKNOWN && pt->n_failed_reproj_ > 15) map_.safeDeletePoint(pt);
        if (pt->type_ == Point::TYPE_CANDIDATE && pt->n_failed_reproj_ > 30) map_.point_candidates_.deleteCandidatePoint(pt);
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
}

}  // namespace svo
