// frame_tracker_batch.h -- the host side of svo_hip_tracker_track, written once against a small Host policy (as
// depth_filter_batch.h is for DepthFilter::updateSeeds): frame_tracker_hip.h instantiates it on the reference's own types
// (svo::Frame / Feature / Point / Map, compile-checked against the reference's headers by `make -C oracle dropin-check`),
// android_svo_amd/host/svo_host.h on its self-contained twins of them -- and that instance RUNS on the GPU
// (svo_host_demo, tests/test_gpu_host_cpp.py), so the flattening of the map's pointer graph into index tables and the
// write-back of a frame's outcome are exercised code, not only compiled code.
//
// What the class does is the host side the reference keeps: it flattens the pointer graph of the map into the index tables
// the tracker walks (when the map has changed), hands the new image over, and applies the outcome to the caller's own
// objects with their own functions -- new_frame->T_f_w_, frame->addFeature(new Feature(...)), the points' reprojection
// counters and types, Map::safeDeletePoint / MapPointCandidates::deleteCandidatePoint for the points the reprojector gave
// up on, overlap_kfs -- so that everything behind the call (structure optimisation, keyframe selection, depth filter, map
// maintenance) runs unchanged on the same data it would have had.
//
// Host policy (all static):
//   types      Frame, FramePtr, Feature, Point, Map, FeatureList (type of Frame::fts_), CandidateList (type of
//              Map::point_candidates_.candidates_); members are used by the reference's names (fts_, key_pts_, id_, point, px, f,
//              level, grad, pos_, type_, n_failed_reproj_, n_succeeded_reproj_, obs_, keyframes_, point_candidates_.mut_ ...)
//   pose7(frame, T[7]) / setPose(frame, T[7])      Frame::T_f_w_ as {t, q(xyzw)}
//   level0(frame, &stride, &cols, &rows)           the frame's full-resolution image
//   makeFeature(frame*, px[2], f[3], level)        new Feature(frame, px, f, level)
//   setEdgelet(feature, grad[2]) / isEdgelet(feature)
//   setCov(frame, cov[36])                         Frame::Cov_
#ifndef SVO_FRAME_TRACKER_BATCH_H_
#define SVO_FRAME_TRACKER_BATCH_H_

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <deque>
#include <list>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "svo_hip.h"

namespace svo {
namespace hip_bridge {

/// svo_hip_ctx with scope (the bindings' Context of svo_hip_bridge.h is the same thing on the reference's side)
class TrackerContext {
 public:
  /// device < 0: no context of its own (a camera of a FrameTrackerGroupT uses the group's)
  explicit TrackerContext(int device) : ctx_(NULL) { if (device >= 0 && svo_hip_ctx_create(&ctx_, device, NULL) != SVO_HIP_OK) ctx_ = NULL; }
  ~TrackerContext() { if (ctx_) svo_hip_ctx_destroy(ctx_); }
  bool ok() const { return ctx_ != NULL; }
  svo_hip_ctx* get() const { return ctx_; }
 private:
  TrackerContext(const TrackerContext&);
  TrackerContext& operator=(const TrackerContext&);
  svo_hip_ctx* ctx_;
};

template <class Host>
class FrameTrackerT {
 public:
  typedef typename Host::Frame Frame;
  typedef typename Host::FramePtr FramePtr;
  typedef typename Host::Feature Feature;
  typedef typename Host::Point Point;
  typedef typename Host::Map Map;
  struct Outcome {
    size_t img_align_n_tracked;      // SparseImgAlign::run (:188)
    size_t repr_n_matches, repr_n_trials;       // reprojector_.n_matches_ / n_trials_ (:206-207)
    bool pose_optimised;             // false: fewer than Config::qualityMinFts() matches, processFrame returns RESULT_FAILURE (:208-215)
    size_t sfba_n_edges_final;       // pose_optimizer's num_obs (:226-229)
    double sfba_thresh, sfba_error_init, sfba_error_final;
  };

  /// cfg: svo_hip_tracker_default_config with the caller's Config values filled in (the bindings do that)
  FrameTrackerT(const svo_hip_camera& cam, const svo_hip_tracker_config& cfg)
      : ctx_(0), err_ctx_(ctx_.get()), trk_(NULL), owns_trk_(true), cfg_(cfg), map_dirty_(true), have_last_(false) {
    if (ctx_.ok() && svo_hip_tracker_create(ctx_.get(), &cam, &cfg, &trk_) != SVO_HIP_OK) trk_ = NULL;
  }
  /// a camera of a FrameTrackerGroupT: handle and context belong to the group (svo_hip_tracker_group_camera)
  FrameTrackerT(svo_hip_ctx* group_ctx, svo_hip_tracker* camera, const svo_hip_tracker_config& cfg)
      : ctx_(-1), err_ctx_(group_ctx), trk_(camera), owns_trk_(false), cfg_(cfg), map_dirty_(true), have_last_(false) {}
  ~FrameTrackerT() { if (trk_ && owns_trk_) svo_hip_tracker_destroy(trk_); }
  bool ok() const { return trk_ != NULL; }
  /// how often the map has been flattened and uploaded (diagnostic)
  size_t mapUploads() const { return n_uploads_; }

  /// page-locked buffer of one full-resolution image (svo_hip_tracker_image_buffer): a new frame whose level 0 lives there
  /// (cv::Mat(rows, cols, CV_8UC1, tracker.imageBuffer()) handed to the Frame constructor) is tracked without the copy of its
  /// image into the buffer; NULL if the tracker could not be created
  uint8_t* imageBuffer() const {
    uint8_t* b = NULL;
    return trk_ && svo_hip_tracker_image_buffer(trk_, &b) == SVO_HIP_OK ? b : NULL;
  }
  /// the map changed behind the tracker's back (keyframe added / removed, points optimised or deleted, candidates added):
  /// flatten it again before the next frame.  processFrame calls this after map_.addKeyframe, optimizeStructure etc.
  void mapChanged() { map_dirty_ = true; }
  /// FrameHandlerBase::optimizeStructure moved points of `frame` (frame_handler_base.cpp:190-210): push their positions
  /// (the tables keep their indices, nothing else of the map changed)
  bool pointsOptimised(const Frame& frame) {
    if (!trk_ || map_dirty_) return trk_ != NULL;              // a full upload is pending anyway
    std::vector<int32_t> idx;
    std::vector<double> pos;
    for (typename Host::FeatureList::const_iterator it = frame.fts_.begin(); it != frame.fts_.end(); ++it) {
      if ((*it)->point == NULL) continue;
      typename std::map<const Point*, int>::const_iterator pi = index_of_point_.find((*it)->point);
      if (pi == index_of_point_.end()) continue;
      idx.push_back(pi->second);
      pos.push_back((*it)->point->pos_[0]); pos.push_back((*it)->point->pos_[1]); pos.push_back((*it)->point->pos_[2]);
    }
    return svo_hip_tracker_update_point_positions(trk_, (int)idx.size(), idx.data(), pos.data()) == SVO_HIP_OK;
  }
  /// last_frame_ was set by somebody else (initialisation, relocalisation)
  void lastFrameChanged() { have_last_ = false; }

  /// new_frame->T_f_w_ = last_frame->T_f_w_; SparseImgAlign::run; Reprojector::reprojectMap; pose_optimizer::optimizeGaussNewton.
  /// Returns false on a device error (the caller treats the frame as a tracking failure).
  bool track(const FramePtr& last_frame, const FramePtr& new_frame, Map& map,
             std::vector<std::pair<FramePtr, size_t> >& overlap_kfs, Outcome& out) {
    const uint8_t* level0 = NULL;
    if (!prepare(last_frame, new_frame, map, &level0)) return false;
    svo_hip_track_result r;
    if (svo_hip_tracker_track(trk_, level0, &r, f_px_.data(), f_f_.data(), f_level_.data(), f_point_.data(), f_edge_.data(), f_grad_.data(),
                              p_type_.data(), p_failed_.data(), p_succ_.data()) != SVO_HIP_OK) {
      fprintf(stderr, "[svo_hip] FrameTracker::track FAILED: %s\n", svo_hip_last_error(err_ctx_));
      return false;
    }
    apply(r, new_frame, map, overlap_kfs, out);
    return true;
  }

  /// The host side BEFORE the device call: the map flattened again if it changed, the last frame handed over if somebody else
  /// set it, the new frame's full-resolution image (*level0: valid until the next prepare).  track() = prepare + the device
  /// call + apply; a FrameTrackerGroupT calls prepare on every camera, svo_hip_tracker_group_track once, then fetch + apply.
  bool prepare(const FramePtr& last_frame, const FramePtr& new_frame, Map& map, const uint8_t** level0_out) {
    if (!trk_) return false;
    if (!map_dirty_) {
      // the depth filter's thread adds candidates behind the tracker's back (its convergence callback is
      // MapPointCandidates::newCandidatePoint, frame_handler_mono.cpp:46-48): a list that has grown is flattened again
      std::unique_lock<std::mutex> lock(map.point_candidates_.mut_);
      if (map.point_candidates_.candidates_.size() != n_candidates_) map_dirty_ = true;
    }
    if (map_dirty_ && !uploadMap(map)) return false;
    if (!have_last_ && !uploadLastFrame(*last_frame)) return false;
    int stride = 0, cols = 0, rows = 0;
    const uint8_t* level0 = Host::level0(*new_frame, &stride, &cols, &rows);
    if (stride != cols) {                                    // the kernels assume stride == cols
      packed_.resize((size_t)rows * cols);
      for (int y = 0; y < rows; ++y) memcpy(&packed_[(size_t)y * cols], level0 + (size_t)y * stride, cols);
      level0 = packed_.data();
    }
    const size_t cap = (size_t)cfg_.max_frame_features, np = points_.size();
    f_px_.resize(cap * 2); f_f_.resize(cap * 3); f_level_.resize(cap); f_point_.resize(cap); f_edge_.resize(cap); f_grad_.resize(cap * 2);
    p_type_.resize(np + 1); p_failed_.resize(np + 1); p_succ_.resize(np + 1);
    *level0_out = level0;
    return true;
  }

  /// a group's camera after svo_hip_tracker_group_track: the frame's features and point counters out of the camera's result block
  bool fetch(svo_hip_track_result* r) {
    return trk_ && svo_hip_tracker_last_result(trk_, r, f_px_.data(), f_f_.data(), f_level_.data(), f_point_.data(), f_edge_.data(), f_grad_.data(),
                                               p_type_.data(), p_failed_.data(), p_succ_.data()) == SVO_HIP_OK;
  }

  /// The host side AFTER the device call: what processFrame would have found on its objects after the three stages
  void apply(const svo_hip_track_result& r, const FramePtr& new_frame, Map& map, std::vector<std::pair<FramePtr, size_t> >& overlap_kfs,
             Outcome& out) {
    const size_t np = points_.size();
    Host::setPose(*new_frame, r.T_f_w);
    for (int i = 0; i < r.n_features; ++i) {                 // Reprojector::reprojectCell :217-231
      Feature* ftr = Host::makeFeature(&*new_frame, &f_px_[2 * i], &f_f_[3 * i], f_level_[i]);      // new Feature(frame, px, f, level)
      ftr->point = f_point_[i] >= 0 ? points_[f_point_[i]] : NULL;       // NULL: dropped by the pose refinement (pose_optimizer.cpp:154-157)
      if (f_edge_[i]) Host::setEdgelet(*ftr, &f_grad_[2 * i]);
      new_frame->addFeature(ftr);
    }
    overlap_kfs.clear();
    for (int i = 0; i < r.n_overlap; ++i) overlap_kfs.push_back(std::make_pair(keyframes_[r.overlap_kf[i]], (size_t)r.overlap_count[i]));
    // point bookkeeping (:126-133, :202-215): counters and promotions as numbers, deletions through the map's own functions
    for (size_t p = 0; p < np; ++p) {
      Point* pt = points_[p];
      if (pt == NULL) continue;                              // deleted in an earlier frame: the object is the map's trash, or gone
      const bool deleted_now = p_type_[p] == (int)Point::TYPE_DELETED && pt->type_ != Point::TYPE_DELETED;
      pt->n_failed_reproj_ = p_failed_[p];
      pt->n_succeeded_reproj_ = p_succ_[p];
      if (!deleted_now) { pt->type_ = (typename Point::PointType)p_type_[p]; continue; }
      if (pt->type_ == Point::TYPE_CANDIDATE) { if (map.point_candidates_.deleteCandidatePoint(pt) && n_candidates_ > 0) --n_candidates_; }
      else map.safeDeletePoint(pt);
      // the point now belongs to the map's trash (freed by Map::emptyTrash): this table forgets the object, the index stays
      // taken (the device tables keep the dead entry until the map is flattened again)
      index_of_point_.erase(pt);
      points_[p] = NULL;
    }
    // (r.map_changed: the device tables followed the deletions themselves -- points unlinked, key points chosen again with
    // Frame::removeKeyPoint's rule -- so the map is not flattened again for them; index_of_point_ stays valid)
    out.img_align_n_tracked = (size_t)r.sia_n_tracked;
    out.repr_n_matches = (size_t)r.n_matches; out.repr_n_trials = (size_t)r.n_trials;
    out.pose_optimised = r.pose.ran != 0;
    out.sfba_n_edges_final = (size_t)r.pose.num_obs;
    out.sfba_thresh = r.pose.estimated_scale; out.sfba_error_init = r.pose.error_init; out.sfba_error_final = r.pose.error_final;
    if (r.pose.ran) Host::setCov(*new_frame, r.pose.Cov);
    have_last_ = true;                                       // the device handed the frame over to itself
  }

  /// FrameHandlerBase::optimizeStructure(frame, max_n_pts, max_iter) (S/frame_handler_base.cpp:190-210) with Point::optimize on
  /// the device over the observations the map tables hold: the selection is the reference's own (std::nth_element by
  /// Point::last_structure_optim_ over the frame's points in fts_ order -- whatever the host's standard library makes of
  /// ties, as on the CPU path); the new positions are written to Point::pos_ and stay in the device tables.
  /// Points the tables do not know (created after the last upload) make the map flatten again first.
  bool optimiseStructure(const FramePtr& frame, Map& map, size_t max_n_pts, int max_iter) {
    if (!trk_) return false;
    std::deque<Point*> pts;
    bool unknown = false;
    for (typename Host::FeatureList::iterator it = frame->fts_.begin(); it != frame->fts_.end(); ++it)
      if ((*it)->point != NULL) {
        pts.push_back((*it)->point);
        if (index_of_point_.find((*it)->point) == index_of_point_.end()) unknown = true;
      }
    if ((map_dirty_ || unknown) && !uploadMap(map)) return false;
    max_n_pts = std::min(max_n_pts, pts.size());
    if (max_n_pts == 0) return true;
    std::nth_element(pts.begin(), pts.begin() + max_n_pts, pts.end(), LastOptimLess());
    std::vector<int32_t> idx;
    std::vector<Point*> chosen;
    for (size_t i = 0; i < max_n_pts; ++i) {
      typename std::map<const Point*, int>::const_iterator pi = index_of_point_.find(pts[i]);
      if (pi == index_of_point_.end()) continue;             // (a point no keyframe or candidate list holds: nothing to optimise it with)
      bool twice = false;
      for (size_t j = 0; j < chosen.size(); ++j) twice = twice || chosen[j] == pts[i];
      if (twice) continue;
      idx.push_back(pi->second); chosen.push_back(pts[i]);
    }
    // one device call takes at most 64 points (Config::structureOptimMaxPts() is 20): larger selections go in chunks --
    // a point's refinement reads keyframe poses and bearings only, never another point, so the split changes nothing
    std::vector<double> pos(idx.size() * 3 + 3);
    for (size_t first = 0; first < idx.size(); first += 64) {
      const size_t n_here = std::min<size_t>(64, idx.size() - first);
      if (svo_hip_tracker_optimize_structure(trk_, (int)n_here, idx.data() + first, max_iter, pos.data() + 3 * first, NULL) != SVO_HIP_OK) return false;
    }
    for (size_t i = 0; i < chosen.size(); ++i) {
      chosen[i]->pos_[0] = pos[3 * i]; chosen[i]->pos_[1] = pos[3 * i + 1]; chosen[i]->pos_[2] = pos[3 * i + 2];
    }
    for (size_t i = 0; i < max_n_pts; ++i) pts[i]->last_structure_optim_ = frame->id_;          // :208
    return true;
  }

  /// new_frame_->setKeyframe(); map_.addKeyframe(new_frame_) (:284-330): keep the frame's pyramid on the device as a keyframe
  bool lastFrameBecameKeyframe(const Frame& frame) {
    const int slot = freeSlot();
    if (slot < 0) return false;                               // max_keyframes pyramids are all in use: raise the capacity
    slot_of_frame_[frame.id_] = slot;
    map_dirty_ = true;
    return svo_hip_tracker_keyframe_from_last_frame(trk_, slot) == SVO_HIP_OK;
  }

 private:
  /// lowest keyframe pyramid slot no known keyframe occupies, or -1
  int freeSlot() const {
    std::vector<char> used((size_t)cfg_.max_keyframes, 0);
    for (std::map<int, int>::const_iterator it = slot_of_frame_.begin(); it != slot_of_frame_.end(); ++it) used[(size_t)it->second] = 1;
    for (int s_ = 0; s_ < cfg_.max_keyframes; ++s_) if (!used[(size_t)s_]) return s_;
    return -1;
  }
  struct LastOptimLess {                                   // ptLastOptimComparator (frame_handler_base.cpp:181-184)
    bool operator()(const Point* a, const Point* b) const { return a->last_structure_optim_ < b->last_structure_optim_; }
  };
  bool uploadLastFrame(const Frame& last) {
    std::vector<double> px, f;
    std::vector<int32_t> pt;
    for (typename Host::FeatureList::const_iterator it = last.fts_.begin(); it != last.fts_.end(); ++it) {
      px.push_back((*it)->px[0]); px.push_back((*it)->px[1]);
      f.push_back((*it)->f[0]); f.push_back((*it)->f[1]); f.push_back((*it)->f[2]);
      typename std::map<const Point*, int>::const_iterator pi = index_of_point_.find((*it)->point);
      pt.push_back((*it)->point && pi != index_of_point_.end() ? pi->second : -1);
    }
    double T[7];
    Host::pose7(last, T);
    std::map<int, int>::const_iterator si = slot_of_frame_.find(last.id_);
    int stride = 0, cols = 0, rows = 0;
    const uint8_t* img = Host::level0(last, &stride, &cols, &rows);
    const bool from_slot = si != slot_of_frame_.end();
    if (!from_slot && stride != cols) return false;
    if (svo_hip_tracker_set_last_frame(trk_, from_slot ? NULL : img, from_slot ? si->second : -1, T, (int)pt.size(), px.data(), f.data(),
                                       pt.data()) != SVO_HIP_OK)
      return false;
    have_last_ = true;
    return true;
  }

  /// svo::Map -> index tables (Map::keyframes_ order, fts_ order, Point::obs_ order, candidates_ order)
  bool uploadMap(Map& map) {
    keyframes_.assign(map.keyframes_.begin(), map.keyframes_.end());
    {
      // pyramid slots of keyframes the map no longer holds (Map::safeDeleteFrame) are free again
      std::map<int, int> live;
      for (size_t k = 0; k < keyframes_.size(); ++k) {
        std::map<int, int>::const_iterator si = slot_of_frame_.find(keyframes_[k]->id_);
        if (si != slot_of_frame_.end()) live[si->first] = si->second;
      }
      slot_of_frame_.swap(live);
    }
    points_.clear(); index_of_point_.clear();
    std::map<int, int> index_of_frame;
    std::vector<int32_t> kf_slot, key, ftr_off(1, 0), ftr_pt, ty, nf, ns, obs_off(1, 0), obs_kf, obs_level, cand;
    std::vector<double> T, pos, obs_px, obs_f, obs_grad;
    std::vector<uint8_t> obs_edge;
    struct Local {
      static int pointIndex(Point* p, std::vector<Point*>& pts, std::map<const Point*, int>& idx) {
        typename std::map<const Point*, int>::iterator it = idx.find(p);
        if (it != idx.end()) return it->second;
        const int i = (int)pts.size();
        pts.push_back(p); idx[p] = i;
        return i;
      }
    };
    for (size_t k = 0; k < keyframes_.size(); ++k) {
      const Frame& kf = *keyframes_[k];
      index_of_frame[kf.id_] = (int)k;
      std::map<int, int>::const_iterator si = slot_of_frame_.find(kf.id_);
      if (si == slot_of_frame_.end()) {                      // a keyframe the device has not seen (initialisation): upload its image
        int stride = 0, cols = 0, rows = 0;
        const uint8_t* img = Host::level0(kf, &stride, &cols, &rows);
        if (stride != cols) return false;
        const int slot = freeSlot();
        if (slot < 0 || svo_hip_tracker_upload_keyframe(trk_, slot, img) != SVO_HIP_OK) return false;
        slot_of_frame_[kf.id_] = slot;
        si = slot_of_frame_.find(kf.id_);
      }
      kf_slot.push_back(si->second);
      double Tk[7];
      Host::pose7(kf, Tk);
      T.insert(T.end(), Tk, Tk + 7);
      for (typename Host::FeatureList::const_iterator it = kf.fts_.begin(); it != kf.fts_.end(); ++it)
        if ((*it)->point != NULL) ftr_pt.push_back(Local::pointIndex((*it)->point, points_, index_of_point_));
      ftr_off.push_back((int32_t)ftr_pt.size());
      for (size_t j = 0; j < 5; ++j) {
        const Feature* kp = j < kf.key_pts_.size() ? kf.key_pts_[j] : NULL;
        key.push_back(kp && kp->point ? Local::pointIndex(kp->point, points_, index_of_point_) : -1);
      }
    }
    {
      std::unique_lock<std::mutex> lock(map.point_candidates_.mut_);
      for (typename Host::CandidateList::iterator it = map.point_candidates_.candidates_.begin();
           it != map.point_candidates_.candidates_.end(); ++it)
        cand.push_back(Local::pointIndex(it->first, points_, index_of_point_));
    }
    for (size_t p = 0; p < points_.size(); ++p) {
      const Point* pt = points_[p];
      pos.push_back(pt->pos_[0]); pos.push_back(pt->pos_[1]); pos.push_back(pt->pos_[2]);
      ty.push_back((int)pt->type_); nf.push_back(pt->n_failed_reproj_); ns.push_back(pt->n_succeeded_reproj_);
      for (typename std::list<Feature*>::const_iterator it = pt->obs_.begin(); it != pt->obs_.end(); ++it) {
        std::map<int, int>::const_iterator fi = index_of_frame.find((*it)->frame->id_);
        if (fi == index_of_frame.end()) continue;            // an observation in a frame that is not (yet / any more) a keyframe of the map
        obs_kf.push_back(fi->second);
        obs_px.push_back((*it)->px[0]); obs_px.push_back((*it)->px[1]);
        obs_f.push_back((*it)->f[0]); obs_f.push_back((*it)->f[1]); obs_f.push_back((*it)->f[2]);
        obs_level.push_back((*it)->level);
        obs_edge.push_back(Host::isEdgelet(**it) ? 1 : 0);
        obs_grad.push_back((*it)->grad[0]); obs_grad.push_back((*it)->grad[1]);
      }
      obs_off.push_back((int32_t)obs_kf.size());
    }
    svo_hip_tracker_map m;
    m.n_kf = (int)keyframes_.size(); m.kf_slot = kf_slot.data(); m.T_kf_w = T.data(); m.kf_key_point = key.data();
    m.kf_ftr_offset = ftr_off.data(); m.kf_ftr_point = ftr_pt.data();
    m.n_points = (int)points_.size(); m.pt_pos = pos.data(); m.pt_type = ty.data(); m.pt_n_failed = nf.data(); m.pt_n_succeeded = ns.data();
    m.pt_obs_offset = obs_off.data(); m.obs_kf = obs_kf.data(); m.obs_px = obs_px.data(); m.obs_f = obs_f.data(); m.obs_level = obs_level.data();
    m.obs_edgelet = obs_edge.data(); m.obs_grad = obs_grad.data();
    m.n_candidates = (int)cand.size(); m.cand_point = cand.data();
    if (svo_hip_tracker_set_map(trk_, &m) != SVO_HIP_OK) return false;
    n_candidates_ = cand.size();
    ++n_uploads_;
    map_dirty_ = false;
    have_last_ = false;                                      // point indices changed: the last frame's features refer to them
    return true;
  }

  TrackerContext ctx_;
  svo_hip_ctx* err_ctx_;                                     // the context whose last error a failure reports (own or the group's)
  svo_hip_tracker* trk_;
  bool owns_trk_;
  svo_hip_tracker_config cfg_;
  bool map_dirty_, have_last_;
  size_t n_uploads_ = 0;
  size_t n_candidates_ = 0;                                  // MapPointCandidates::candidates_.size() as uploaded, minus our own deletions
  std::vector<FramePtr> keyframes_;
  std::vector<Point*> points_;
  std::map<const Point*, int> index_of_point_;
  std::map<int, int> slot_of_frame_;                         // Frame::id_ -> keyframe pyramid slot
  std::vector<double> f_px_, f_f_, f_grad_;
  std::vector<int32_t> f_level_, f_point_, p_type_, p_failed_, p_succ_;
  std::vector<uint8_t> f_edge_;
  std::vector<uint8_t> packed_;                              // a padded image compacted to stride == cols
};

/// N cameras -- N FrameHandlerMono objects, each with its own svo::Map -- tracked together: svo_hip_tracker_group (one chain of
/// launches per call for all of them).  camera(c) is camera c's FrameTrackerT (mapChanged, pointsOptimised, optimiseStructure,
/// lastFrameBecameKeyframe, imageBuffer as for a lone tracker); trackAll is processFrame's three stages for every camera.
template <class Host>
class FrameTrackerGroupT {
 public:
  typedef FrameTrackerT<Host> Camera;
  typedef typename Host::FramePtr FramePtr;
  typedef typename Host::Map Map;
  typedef typename Camera::Outcome Outcome;

  FrameTrackerGroupT(const svo_hip_camera& cam, const svo_hip_tracker_config& cfg, int n_cameras) : ctx_(0), group_(NULL) {
    if (ctx_.ok() && svo_hip_tracker_group_create(ctx_.get(), &cam, &cfg, n_cameras, &group_) != SVO_HIP_OK) group_ = NULL;
    for (int c = 0; group_ && c < n_cameras; ++c) {
      svo_hip_tracker* t = NULL;
      if (svo_hip_tracker_group_camera(group_, c, &t) != SVO_HIP_OK) { svo_hip_tracker_group_destroy(group_); group_ = NULL; break; }
      cameras_.push_back(new Camera(ctx_.get(), t, cfg));
    }
  }
  ~FrameTrackerGroupT() {
    for (size_t c = 0; c < cameras_.size(); ++c) delete cameras_[c];
    if (group_) svo_hip_tracker_group_destroy(group_);
  }
  bool ok() const { return group_ != NULL; }
  size_t size() const { return cameras_.size(); }
  Camera& camera(size_t c) { return *cameras_[c]; }

  /// one frame of every camera: last[c] -> fresh[c] over maps[c].  Returns false on a device error (every camera's frame is then
  /// a tracking failure for its caller); overlap_kfs / out: one entry per camera.
  bool trackAll(const std::vector<FramePtr>& last, const std::vector<FramePtr>& fresh, const std::vector<Map*>& maps,
                std::vector<std::vector<std::pair<FramePtr, size_t> > >& overlap_kfs, std::vector<Outcome>& out) {
    const size_t n = cameras_.size();
    if (!group_ || last.size() != n || fresh.size() != n || maps.size() != n) return false;
    std::vector<const uint8_t*> img(n, (const uint8_t*)NULL);
    for (size_t c = 0; c < n; ++c)
      if (!cameras_[c]->prepare(last[c], fresh[c], *maps[c], &img[c])) return false;
    std::vector<svo_hip_track_result> res(n);
    if (svo_hip_tracker_group_track(group_, img.data(), res.data()) != SVO_HIP_OK) {
      fprintf(stderr, "[svo_hip] FrameTrackerGroup::trackAll FAILED: %s\n", svo_hip_last_error(ctx_.get()));
      return false;
    }
    overlap_kfs.resize(n); out.resize(n);
    for (size_t c = 0; c < n; ++c) {
      if (!cameras_[c]->fetch(&res[c])) return false;
      cameras_[c]->apply(res[c], fresh[c], *maps[c], overlap_kfs[c], out[c]);
    }
    return true;
  }

 private:
  FrameTrackerGroupT(const FrameTrackerGroupT&);
  FrameTrackerGroupT& operator=(const FrameTrackerGroupT&);
  TrackerContext ctx_;
  svo_hip_tracker_group* group_;
  std::vector<Camera*> cameras_;
};

}  // namespace hip_bridge
}  // namespace svo

#endif  // SVO_FRAME_TRACKER_BATCH_H_
