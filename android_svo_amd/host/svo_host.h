// svo_host.h -- self-contained C++ host layer over the C-ABI (include/svo_hip.h).
//
// The reference's public classes (svo::SparseImgAlign, svo::DepthFilter, svo::Seed, Frame/Feature/Point)
// with the same names, methods and protocol, but on minimal own data types (no Eigen, no OpenCV), so that
// the host side can be built and RUN wherever libsvo_hip.so runs.  The bindings that plug into the
// reference's real headers are include/svo_dropin/ (compile-checked only); this file is
// their executable counterpart.  DepthFilter::updateSeeds is NOT a second copy: both sides instantiate
// hip_bridge::DeviceSeedMirror (include/svo_dropin/depth_filter_batch.h) with a small Host policy, so the
// batching / ordering / halt logic the GPU tests exercise is the code the drop-in ships.  The same holds for the tracking
// chain: svo::FrameTracker here and in include/svo_dropin/frame_tracker_hip.h are both hip_bridge::FrameTrackerT
// (frame_tracker_batch.h) -- the flattening of svo::Map's pointer graph into the tracker's index tables and the write-back
// of a frame's outcome (features, counters, Map::safeDeletePoint) run on the GPU from this file's Map / Frame / Point.
// Host logic here:
//   * SparseImgAlign::run       S/sparse_img_align.cpp:51-92   (flatten fts_, upload, run, read back)
//   * DepthFilter protocol      S/depth_filter.cpp:47-229,237-357 (thread, 3-deep frame queue, keyframe
//                               hand-off with the halt flag, std::list<Seed>, age-out, convergence callback)
//   * Map / Frame bookkeeping   S/map.cpp:78-99,256-304 (safeDeletePoint, deleteCandidatePoint), S/frame.cpp:67-165
//                               (setKeyframe, addFeature, setKeyPoints / checkKeyPoints / removeKeyPoint)
// Host threads: the tracking thread and the depth-filter thread each own a svo_hip_ctx (one stream each).
#ifndef SVO_HOST_H_
#define SVO_HOST_H_

#include <array>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <set>
#include <stdexcept>
#include <thread>
#include <vector>

#include "svo_hip.h"
#include "svo_dropin/slot_table.h"
#include "svo_dropin/depth_filter_batch.h"
#include "svo_dropin/frame_tracker_batch.h"

namespace svo {

struct Vector2d { double v[2]; double& operator[](int i) { return v[i]; } const double& operator[](int i) const { return v[i]; } };
struct Vector3d { double v[3]; double& operator[](int i) { return v[i]; } const double& operator[](int i) const { return v[i]; } };

/// {t, q(xyzw)} with the reference's composition rules (I/SE3.h:35-61, I/SO3.h:468-488,523-526)
struct SE3 {
  double p[7];
  SE3() : p{0, 0, 0, 0, 0, 0, 1} {}
  explicit SE3(const double* s) { std::memcpy(p, s, sizeof(p)); }
  static void rot(const double* q, const double* x, double* o) {
    double uv[3] = {q[1] * x[2] - q[2] * x[1], q[2] * x[0] - q[0] * x[2], q[0] * x[1] - q[1] * x[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    for (int i = 0; i < 3; ++i) o[i] = (x[i] + q[3] * uv[i]) + c[i];
  }
  SE3 inverse() const {
    SE3 r;
    const double qi[4] = {-p[3], -p[4], -p[5], p[6]};
    double t[3];
    rot(qi, p, t);
    r.p[0] = -t[0]; r.p[1] = -t[1]; r.p[2] = -t[2];
    r.p[3] = qi[0]; r.p[4] = qi[1]; r.p[5] = qi[2]; r.p[6] = qi[3];
    return r;
  }
  Vector3d operator*(const Vector3d& x) const {
    double t[3];
    rot(p + 3, x.v, t);
    return Vector3d{{p[0] + t[0], p[1] + t[1], p[2] + t[2]}};
  }
};

struct PinholeCamera {            // vk::PinholeCamera: pinhole + optional 5-coefficient radtan distortion (S/pinhole_camera.cpp:19-38)
  int width, height;
  double fx, fy, cx, cy;
  double d[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  svo_hip_camera toC() const {
    svo_hip_camera c;
    c.width = width; c.height = height; c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy;
    for (int k = 0; k < 5; ++k) c.d[k] = d[k];
    c.distortion = std::fabs(d[0]) > 0.0000001 ? 1 : 0;       // distortion_(fabs(d0) > 0.0000001), S/pinhole_camera.cpp:26
    return c;
  }
};

struct Feature;
struct Point {                                      // I/point.h: position + the features that observe it
  enum PointType { TYPE_DELETED, TYPE_CANDIDATE, TYPE_UNKNOWN, TYPE_GOOD };      // I/point.h:33-38
  Vector3d pos_;
  std::list<Feature*> obs_;
  PointType type_ = TYPE_UNKNOWN;
  int n_failed_reproj_ = 0, n_succeeded_reproj_ = 0;                             // the reprojector's counters (I/point.h:49-50)
  int last_structure_optim_ = 0;                                                 // id of the frame that optimised it last (:51)
  explicit Point(const Vector3d& p) : pos_(p) {}
  Point(const Vector3d& p, Feature* ftr) : pos_(p) { obs_.push_front(ftr); }     // S/point.cpp:39-48
  void addFrameRef(Feature* ftr) { obs_.push_front(ftr); }                       // S/point.cpp:52-55
};
struct Frame;
struct Feature {
  enum FeatureType { CORNER, EDGELET };                                          // I/feature.h:26-29
  FeatureType type = CORNER;
  Frame* frame; Vector2d px; Vector3d f; int level; Point* point;
  Vector2d grad{{1.0, 0.0}};                                                     // edgelets: direction of the gradient, normalised
  Feature(Frame* fr, const Vector2d& px_, const Vector3d& f_, int lvl) : frame(fr), px(px_), f(f_), level(lvl), point(nullptr) {}
};

struct Frame {
  static int frame_counter_;
  int id_;
  const PinholeCamera* cam_;
  SE3 T_f_w_;
  std::vector<std::vector<uint8_t>> img_pyr_;      // level l: (w>>l) x (h>>l), stride == cols
  std::list<Feature*> fts_;
  std::vector<Feature*> key_pts_ = std::vector<Feature*>(5, nullptr);   // five features spread over the image: the keyframe's overlap test
  std::array<double, 36> Cov_{};                   // covariance of the refined pose (pose_optimizer.cpp:141)
  bool is_keyframe_ = false;
  Frame(const PinholeCamera* cam, std::vector<std::vector<uint8_t>> pyr) : id_(frame_counter_++), cam_(cam), img_pyr_(std::move(pyr)) {}
  ~Frame() { for (Feature* f : fts_) delete f; }
  bool isKeyframe() const { return is_keyframe_; }
  void setKeyframe() { is_keyframe_ = true; setKeyPoints(); }                    // S/frame.cpp:67-71
  void addFeature(Feature* ftr) { fts_.push_back(ftr); }                         // :75-78

  /// S/frame.cpp:83-92: key features whose point is gone are dropped, then every feature with a point competes again
  void setKeyPoints() {
    for (Feature*& k : key_pts_) if (k != nullptr && k->point == nullptr) k = nullptr;
    for (Feature* ftr : fts_) if (ftr->point != nullptr) checkKeyPoints(ftr);
  }
  /// :98-149.  Slot 0: the feature closest to the image centre (max norm).  Slots 1..4: per quadrant the feature with the
  /// largest product (x - cu)(y - cv); the two left quadrants test x against cv, not cu, as the reference does (:133,142).
  void checkKeyPoints(Feature* ftr) {
    const int cu = cam_->width / 2, cv = cam_->height / 2;
    const double x = ftr->px[0], y = ftr->px[1];
    auto off_centre = [&](const Feature* g) { return std::max(std::fabs(g->px[0] - cu), std::fabs(g->px[1] - cv)); };
    auto product = [&](const Feature* g) { return (g->px[0] - cu) * (g->px[1] - cv); };
    if (key_pts_[0] == nullptr || off_centre(ftr) < off_centre(key_pts_[0])) key_pts_[0] = ftr;
    const bool in_quadrant[4] = {x >= cu && y >= cv, x >= cu && y < cv, x < cv && y < cv, x < cv && y >= cv};
    for (int q = 0; q < 4; ++q) {
      if (!in_quadrant[q]) continue;
      Feature*& slot = key_pts_[1 + q];
      if (slot == nullptr || product(ftr) > product(slot)) slot = ftr;
    }
  }
  /// :154-165
  void removeKeyPoint(Feature* ftr) {
    bool found = false;
    for (Feature*& k : key_pts_) if (k == ftr) { k = nullptr; found = true; }
    if (found) setKeyPoints();
  }
};
inline int Frame::frame_counter_ = 0;
typedef std::shared_ptr<Frame> FramePtr;

namespace hip_bridge {
inline void check(int rc, svo_hip_ctx* ctx, const char* what) {
  if (rc != SVO_HIP_OK) throw std::runtime_error(std::string(what) + ": " + (ctx ? svo_hip_last_error(ctx) : "no context"));
}
/// device copies of frame pyramids, cached by Frame::id_ (the slot bookkeeping is the drop-in's: include/svo_dropin/slot_table.h)
class PyramidCache {
 public:
  PyramidCache(svo_hip_ctx* ctx, int capacity) : ctx_(ctx), table_(capacity) {}
  ~PyramidCache() { if (pyr_) svo_hip_pyramid_destroy(pyr_); }
  int slotOf(const Frame& f) {
    std::vector<const Frame*> one(1, &f);
    std::vector<int> slots;
    acquire(one, slots);
    return slots[0];
  }
  /// all frames of one device call at once: a slot handed out for one of them is not recycled for another
  bool acquire(const std::vector<const Frame*>& frames, std::vector<int>& slots) {
    slots.assign(frames.size(), -1);
    if (frames.empty()) return true;
    std::vector<int> ids;
    std::set<int> distinct;
    for (const Frame* f : frames) { ids.push_back(f->id_); distinct.insert(f->id_); }
    const int need = table_.capacityFor((int)distinct.size());
    // (every frame of a call has the geometry of the first; a cache that holds another geometry starts again)
    const Frame& f = *frames[0];
    const int n_levels = (int)f.img_pyr_.size();
    for (const Frame* g : frames)
      if (g->cam_->width != f.cam_->width || g->cam_->height != f.cam_->height || (int)g->img_pyr_.size() != n_levels) return false;
    if (need != table_.capacity() || !pyr_ || f.cam_->width != width_ || f.cam_->height != height_ || n_levels != n_levels_) {
      if (pyr_) { svo_hip_pyramid_destroy(pyr_); pyr_ = nullptr; }
      check(svo_hip_pyramid_create(ctx_, f.cam_->width, f.cam_->height, n_levels, need, &pyr_), ctx_, "pyramid_create");
      width_ = f.cam_->width; height_ = f.cam_->height; n_levels_ = n_levels;
      table_.reset(need);
      ++n_created_;
    }
    return table_.acquire(ids, slots, [&](size_t k, int s) {
      const Frame& f = *frames[k];
      const uint8_t* lv[SVO_HIP_MAX_LEVELS] = {nullptr};
      for (size_t l = 0; l < f.img_pyr_.size(); ++l) lv[l] = f.img_pyr_[l].data();
      check(svo_hip_pyramid_upload(pyr_, s, lv), ctx_, "pyramid_upload");
      check(svo_hip_ctx_sync(ctx_), ctx_, "sync");
      ++n_uploads_;
      return true;
    });
  }
  svo_hip_pyramid* pyramid() const { return pyr_; }
  int capacity() const { return table_.capacity(); }
  int uploads() const { return n_uploads_; }
 private:
  svo_hip_ctx* ctx_; svo_hip_pyramid* pyr_ = nullptr; int width_ = 0, height_ = 0, n_levels_ = 0; SlotTable table_; int n_uploads_ = 0, n_created_ = 0;
};
}  // namespace hip_bridge

/// I/map.h:34-66, S/map.cpp:256-304: converged seeds that no keyframe holds yet
struct MapPointCandidates {
  typedef std::pair<Point*, Feature*> PointCandidate;
  typedef std::list<PointCandidate> PointCandidateList;
  std::mutex mut_;
  PointCandidateList candidates_;
  std::list<Point*> trash_points_;
  ~MapPointCandidates() {
    for (PointCandidate& c : candidates_) delete c.second;
  }
  bool deleteCandidatePoint(Point* point) {                                      // S/map.cpp:256-269, :297-304
    std::unique_lock<std::mutex> lock(mut_);
    for (auto it = candidates_.begin(); it != candidates_.end(); ++it) {
      if (it->first != point) continue;
      delete it->second;                           // the candidate's only feature
      it->first->type_ = Point::TYPE_DELETED;
      trash_points_.push_back(it->first);
      candidates_.erase(it);
      return true;
    }
    return false;
  }
};

/// I/map.h:69-130: keyframes + the candidates; points are owned through the features that refer to them
struct Map {
  std::list<FramePtr> keyframes_;
  std::list<Point*> trash_points_;
  MapPointCandidates point_candidates_;
  void addKeyframe(FramePtr kf) { keyframes_.push_back(kf); }                    // S/map.cpp:96-99
  void deletePoint(Point* pt) { pt->type_ = Point::TYPE_DELETED; trash_points_.push_back(pt); }   // :90-94
  /// :78-88: every observation lets go of the point (a keyframe that loses a key feature picks its key features again)
  void safeDeletePoint(Point* pt) {
    for (Feature* ftr : pt->obs_) { ftr->point = nullptr; ftr->frame->removeKeyPoint(ftr); }
    pt->obs_.clear();
    deletePoint(pt);
  }
};

/// Host policy of hip_bridge::FrameTrackerT on this file's data model (the drop-in instantiates the same template on the
/// reference's types: include/svo_dropin/frame_tracker_hip.h)
struct HostTrackerPolicy {
  typedef svo::Frame Frame;
  typedef svo::FramePtr FramePtr;
  typedef svo::Feature Feature;
  typedef svo::Point Point;
  typedef svo::Map Map;
  typedef std::list<svo::Feature*> FeatureList;
  typedef MapPointCandidates::PointCandidateList CandidateList;
  static void pose7(const Frame& fr, double T[7]) { std::memcpy(T, fr.T_f_w_.p, sizeof(double) * 7); }
  static void setPose(Frame& fr, const double T[7]) { fr.T_f_w_ = SE3(T); }
  static const uint8_t* level0(const Frame& fr, int* stride, int* cols, int* rows) {
    *stride = fr.cam_->width; *cols = fr.cam_->width; *rows = fr.cam_->height;
    return fr.img_pyr_[0].data();
  }
  static Feature* makeFeature(Frame* fr, const double px[2], const double f[3], int level) {
    return new Feature(fr, Vector2d{{px[0], px[1]}}, Vector3d{{f[0], f[1], f[2]}}, level);
  }
  static void setEdgelet(Feature& ftr, const double g[2]) { ftr.type = Feature::EDGELET; ftr.grad = Vector2d{{g[0], g[1]}}; }
  static bool isEdgelet(const Feature& ftr) { return ftr.type == Feature::EDGELET; }
  static void setCov(Frame& fr, const double cov[36]) { std::memcpy(fr.Cov_.data(), cov, sizeof(double) * 36); }
};

/// The first half of FrameHandlerMono::processFrame (frame_handler_mono.cpp:171-229) in one device call per frame: the
/// executable twin of include/svo_dropin/frame_tracker_hip.h.  cfg: svo_hip_tracker_default_config + the caller's values.
class FrameTracker : public hip_bridge::FrameTrackerT<HostTrackerPolicy> {
 public:
  FrameTracker(const PinholeCamera& cam, const svo_hip_tracker_config& cfg) : hip_bridge::FrameTrackerT<HostTrackerPolicy>(cam.toC(), cfg) {}
};

/// N cameras with a map each, tracked together (hip_bridge::FrameTrackerGroupT on this file's data model; svo_hip_tracker_group)
class FrameTrackerGroup : public hip_bridge::FrameTrackerGroupT<HostTrackerPolicy> {
 public:
  FrameTrackerGroup(const PinholeCamera& cam, const svo_hip_tracker_config& cfg, int n_cameras)
      : hip_bridge::FrameTrackerGroupT<HostTrackerPolicy>(cam.toC(), cfg, n_cameras) {}
};

/// I/sparse_img_align.h:33-79
class SparseImgAlign {
 public:
  enum Method { GaussNewton, LevenbergMarquardt };                             // I/nlls_solver.h:46-48
  enum ScaleEstimatorType { UnitScale, TDistScale, MADScale, NormalScale };
  enum WeightFunctionType { UnitWeight, TDistWeight, TukeyWeight, HuberWeight };
  size_t n_iter_; double eps_; bool stop_ = false; size_t n_meas_ = 0;
  Method method_;
  double mu_ = 0.01f, nu_ = 2.0;
  float scale_ = 0.0f;
  SparseImgAlign(int n_levels, int min_level, int n_iter, Method method, bool display, bool verbose)
      : n_iter_(n_iter), eps_(0.000001), method_(method), max_level_(n_levels), min_level_(min_level) {
    (void)display; (void)verbose;
    hip_bridge::check(svo_hip_ctx_create(&ctx_, 0, nullptr), nullptr, "ctx_create");
    ref_.reset(new hip_bridge::PyramidCache(ctx_, 1));
    cur_.reset(new hip_bridge::PyramidCache(ctx_, 1));
  }
  ~SparseImgAlign() { if (sia_) svo_hip_sia_destroy(sia_); ref_.reset(); cur_.reset(); svo_hip_ctx_destroy(ctx_); }

  /// NLLSSolver::setRobustCostFunction (I/nlls_solver_impl.hpp:229-281): UnitScale switches the weights off
  void setRobustCostFunction(ScaleEstimatorType scale_estimator, WeightFunctionType weight_function) {
    scale_kind_ = (int)scale_estimator; weight_kind_ = (int)weight_function;
  }

  size_t run(FramePtr ref_frame, FramePtr cur_frame) {
    stop_ = false; n_meas_ = 0; chi2_ = 1e10;
    if (ref_frame->fts_.empty()) return 0;                                   // :55-59
    const int n = (int)ref_frame->fts_.size();
    std::vector<double> px(2 * (size_t)n), f(3 * (size_t)n), pos(3 * (size_t)n, 0.0);
    std::vector<uint8_t> hp((size_t)n, 0);
    size_t i = 0;
    for (Feature* ftr : ref_frame->fts_) {                                    // list order defines the patch order
      px[2 * i] = ftr->px[0]; px[2 * i + 1] = ftr->px[1];
      for (int k = 0; k < 3; ++k) f[3 * i + k] = ftr->f[k];
      if (ftr->point) { hp[i] = 1; for (int k = 0; k < 3; ++k) pos[3 * i + k] = ftr->point->pos_[k]; }
      ++i;
    }
    if (!sia_ || cap_ < n) {
      if (sia_) svo_hip_sia_destroy(sia_);
      cap_ = n > 2048 ? n : 2048;
      hip_bridge::check(svo_hip_sia_create(ctx_, 1, cap_, &sia_), ctx_, "sia_create");
    }
    const int rs = ref_->slotOf(*ref_frame), cs = cur_->slotOf(*cur_frame);
    (void)rs; (void)cs;
    const svo_hip_camera cam = cur_frame->cam_->toC();
    svo_hip_sia_params prm{max_level_, min_level_, (int)n_iter_, eps_, 1};
    svo_hip_sia_result res;
    hip_bridge::check(svo_hip_sia_set_option(sia_, SVO_HIP_SIA_OPT_METHOD, method_ == LevenbergMarquardt ? SVO_HIP_SIA_METHOD_LEVENBERG_MARQUARDT
                                                                                                         : SVO_HIP_SIA_METHOD_GAUSS_NEWTON), ctx_, "set_option");
    hip_bridge::check(svo_hip_sia_set_option(sia_, SVO_HIP_SIA_OPT_SCALE_ESTIMATOR, scale_kind_), ctx_, "set_option");
    hip_bridge::check(svo_hip_sia_set_option(sia_, SVO_HIP_SIA_OPT_WEIGHT_FUNCTION, weight_kind_), ctx_, "set_option");
    hip_bridge::check(svo_hip_sia_set_frames(sia_, ref_->pyramid(), cur_->pyramid()), ctx_, "set_frames");
    hip_bridge::check(svo_hip_sia_upload_features(sia_, 0, n, px.data(), f.data(), pos.data(), hp.data()), ctx_, "upload_features");
    hip_bridge::check(svo_hip_sia_upload_poses(sia_, 0, &cam, ref_frame->T_f_w_.p, cur_frame->T_f_w_.p), ctx_, "upload_poses");
    hip_bridge::check(svo_hip_sia_run(sia_, 1, &prm), ctx_, "run");
    hip_bridge::check(svo_hip_sia_download(sia_, 0, &res), ctx_, "download");
    cur_frame->T_f_w_ = SE3(res.T_cur_w);                                     // :89
    std::memcpy(H_.data(), res.H, sizeof(res.H));
    chi2_ = res.chi2; stop_ = res.stop != 0; n_meas_ = (size_t)res.n_tracked * 16;
    if (method_ == LevenbergMarquardt || scale_kind_ != SVO_HIP_SIA_SCALE_UNIT)
      hip_bridge::check(svo_hip_sia_solver_state(sia_, 0, &scale_, &mu_, &nu_), ctx_, "solver_state");
    return (size_t)res.n_tracked;                                             // :91
  }
  std::array<double, 36> getFisherInformation() const {                       // :94-99
    std::array<double, 36> I = H_;
    const double sigma_i_sq = 5e-4 * 255 * 255;
    for (double& v : I) v /= sigma_i_sq;
    return I;
  }
  double getChi2() const { return chi2_; }

 private:
  int max_level_, min_level_;
  int scale_kind_ = SVO_HIP_SIA_SCALE_UNIT, weight_kind_ = SVO_HIP_SIA_WEIGHT_UNIT;
  svo_hip_ctx* ctx_ = nullptr;
  svo_hip_sia* sia_ = nullptr;
  int cap_ = 0;
  std::unique_ptr<hip_bridge::PyramidCache> ref_, cur_;
  std::array<double, 36> H_{};
  double chi2_ = 1e10;
};

/// The detector grid of feature_detection::AbstractDetector (S/feature_detection.cpp:24-64): cells the depth filter
/// marks on keyframes so that initializeSeeds does not seed again where a live seed was just matched.
struct DetectorGrid {
  int cell_size_, grid_n_cols_, grid_n_rows_;
  std::vector<bool> grid_occupancy_;
  DetectorGrid(int img_width, int img_height, int cell_size)
      : cell_size_(cell_size), grid_n_cols_((int)std::ceil((double)img_width / cell_size)),
        grid_n_rows_((int)std::ceil((double)img_height / cell_size)), grid_occupancy_((size_t)grid_n_cols_ * grid_n_rows_, false) {}
  void resetGrid() { std::fill(grid_occupancy_.begin(), grid_occupancy_.end(), false); }
  void setGridOccpuancy(const Vector2d& px) {                                   // :58-64 (spelling as in the reference)
    grid_occupancy_.at((size_t)((int)(px[1] / cell_size_) * grid_n_cols_ + (int)(px[0] / cell_size_))) = true;
  }
};

/// I/depth_filter.h:36-52
struct Seed {
  static int batch_counter, seed_counter;
  int batch_id, id;
  Feature* ftr;
  float a, b, mu, z_range, sigma2;
  Seed(Feature* ftr_, float depth_mean, float depth_min)
      : batch_id(batch_counter), id(seed_counter++), ftr(ftr_), a(10), b(10), mu((float)(1.0 / depth_mean)),
        z_range((float)(1.0 / depth_min)), sigma2(z_range * z_range / 36) {}
};
inline int Seed::batch_counter = 0;
inline int Seed::seed_counter = 0;

/// I/depth_filter.h:60-166.  The detector is out of scope: the new keyframe's features are handed in.
class DepthFilter {
 public:
  typedef std::unique_lock<std::mutex> lock_t;
  typedef std::function<void(Point*, double)> callback_t;
  struct Options {
    int max_n_kfs = 3;
    double seed_convergence_sigma2_thresh = 100.0;
    bool verbose = false;
  } options_;

  explicit DepthFilter(callback_t seed_converged_cb, DetectorGrid* feature_detector = nullptr)
      : feature_detector_(feature_detector), seed_converged_cb_(std::move(seed_converged_cb)) {
    hip_bridge::check(svo_hip_ctx_create(&ctx_, 0, nullptr), nullptr, "ctx_create");   // the filter thread's own stream
    kf_pyr_.reset(new hip_bridge::PyramidCache(ctx_, 8));
    cur_pyr_.reset(new hip_bridge::PyramidCache(ctx_, 2));
  }
  virtual ~DepthFilter() { stopThread(); mirror_.clear(); kf_pyr_.reset(); cur_pyr_.reset(); svo_hip_ctx_destroy(ctx_); }

  void startThread() { thread_stop_ = false; thread_ = new std::thread(&DepthFilter::updateSeedsLoop, this); }
  void stopThread() {
    if (thread_ != nullptr) {
      seeds_updating_halt_ = true;
      { lock_t lock(frame_queue_mut_); thread_stop_ = true; }
      frame_queue_cond_.notify_one();
      if (thread_->joinable()) thread_->join();
      delete thread_;
      thread_ = nullptr;
    }
  }
  void addFrame(FramePtr frame) {                                              // depth_filter.cpp:87-104
    if (thread_ != nullptr) {
      {
        lock_t lock(frame_queue_mut_);
        if (frame_queue_.size() > 2) frame_queue_.pop();
        frame_queue_.push(frame);
      }
      seeds_updating_halt_ = false;
      frame_queue_cond_.notify_one();
    } else {
      updateSeeds(frame);
    }
  }
  void addKeyframe(FramePtr frame, double depth_mean, double depth_min, std::vector<Feature*> new_features) {   // :109-123
    new_keyframe_min_depth_ = depth_min;
    new_keyframe_mean_depth_ = depth_mean;
    if (thread_ != nullptr) {
      lock_t lock(frame_queue_mut_);
      new_keyframe_ = frame;
      new_keyframe_features_ = std::move(new_features);
      new_keyframe_set_ = true;
      seeds_updating_halt_ = true;
      frame_queue_cond_.notify_one();
    } else {
      initializeSeeds(frame, new_features);
    }
  }
  void removeKeyframe(FramePtr frame) {                                        // :153-170
    seeds_updating_halt_ = true;
    lock_t lock(seeds_mut_);
    for (auto it = seeds_.begin(); it != seeds_.end();) it = (it->ftr->frame == frame.get()) ? seeds_.erase(it) : std::next(it);
    seeds_updating_halt_ = false;
  }
  void reset() {                                                               // :172-186
    seeds_updating_halt_ = true;
    { lock_t lock(seeds_mut_); seeds_.clear(); }
    lock_t lock(frame_queue_mut_);
    while (!frame_queue_.empty()) frame_queue_.pop();
    seeds_updating_halt_ = false;
  }
  /// The seeds' state lives on the device between frames (hip_bridge::DeviceSeedMirror): bring it into the list
  void syncSeeds() {
    lock_t lock(seeds_mut_);
    if (!mirror_.syncToHost()) throw std::runtime_error(std::string("seed_batch_download: ") + svo_hip_last_error(ctx_));
  }
  std::list<Seed>& getSeeds() { syncSeeds(); return seeds_; }
  void getSeedsCopy(const FramePtr& frame, std::list<Seed>& seeds) {           // :349-357
    lock_t lock(seeds_mut_);
    if (!mirror_.syncToHost()) throw std::runtime_error(std::string("seed_batch_download: ") + svo_hip_last_error(ctx_));
    for (const Seed& s : seeds_) if (s.ftr->frame == frame.get()) seeds.push_back(s);
  }
  bool idle() {
    lock_t lock(frame_queue_mut_);
    return frame_queue_.empty() && !new_keyframe_set_ && !busy_;
  }

 protected:
  void initializeSeeds(FramePtr frame, const std::vector<Feature*>& new_features) {   // :129-151
    (void)frame;
    seeds_updating_halt_ = true;
    lock_t lock(seeds_mut_);
    ++Seed::batch_counter;
    for (Feature* ftr : new_features) seeds_.push_back(Seed(ftr, (float)new_keyframe_mean_depth_, (float)new_keyframe_min_depth_));
    seeds_updating_halt_ = false;
  }

  void updateSeedsLoop() {                                                      // :191-229
    while (true) {
      FramePtr frame;
      std::vector<Feature*> feats;
      bool is_kf = false;
      {
        lock_t lock(frame_queue_mut_);
        while (frame_queue_.empty() && !new_keyframe_set_ && !thread_stop_) frame_queue_cond_.wait(lock);
        if (thread_stop_) return;
        if (new_keyframe_set_) {
          new_keyframe_set_ = false;
          seeds_updating_halt_ = false;
          while (!frame_queue_.empty()) frame_queue_.pop();
          frame = new_keyframe_;
          feats = std::move(new_keyframe_features_);
          is_kf = true;
        } else {
          frame = frame_queue_.front();
          frame_queue_.pop();
        }
        busy_ = true;
      }
      updateSeeds(frame);
      if (is_kf) initializeSeeds(frame, feats);
      { lock_t lock(frame_queue_mut_); busy_ = false; }
    }
  }

  /// Host policy of hip_bridge::DeviceSeedMirror on this file's data model
  struct BatchHost {
    DepthFilter* df;
    Frame* keyframeOf(const Seed& s) const { return s.ftr->frame; }
    void feature(const Seed& s, double px[2], double f[3], int* level) const {
      px[0] = s.ftr->px[0]; px[1] = s.ftr->px[1];
      for (int k = 0; k < 3; ++k) f[k] = s.ftr->f[k];
      *level = s.ftr->level;
    }
    void pose7(const Frame& fr, double T[7]) const { std::memcpy(T, fr.T_f_w_.p, sizeof(double) * 7); }
    bool keyframeSlots(const std::vector<Frame*>& kfs, std::vector<int>& slots) {
      std::vector<const Frame*> c(kfs.begin(), kfs.end());
      return df->kf_pyr_->acquire(c, slots);
    }
    int currentSlot(Frame& fr) { return df->cur_pyr_->slotOf(fr); }
    svo_hip_pyramid* keyframePyramids() const { return df->kf_pyr_->pyramid(); }
    svo_hip_pyramid* currentPyramids() const { return df->cur_pyr_->pyramid(); }
    svo_hip_camera camera(const Frame& fr) const { return fr.cam_->toC(); }
    bool isKeyframe(const Frame& fr) const { return fr.isKeyframe(); }
    void setGridOccupancy(const double px_cur[2]) {                              // depth_filter.cpp:302-306
      if (df->feature_detector_) df->feature_detector_->setGridOccpuancy(Vector2d{{px_cur[0], px_cur[1]}});
    }
    void converged(Seed& s, const double xyz[3]) {                               // :310-331
      Point* point = new Point(Vector3d{{xyz[0], xyz[1], xyz[2]}}, s.ftr);
      s.ftr->point = point;
      df->seed_converged_cb_(point, s.sigma2);
    }
  };

  /// depth_filter.cpp:237-341 through the shared batched implementation
  virtual void updateSeeds(FramePtr frame) {
    lock_t lock(seeds_mut_);
    if (seeds_.empty()) return;
    svo_hip_df_params prm{3, 10, 1000, options_.seed_convergence_sigma2_thresh};
    BatchHost host{this};
    last_stats_ = mirror_.update(host, ctx_, seeds_, *frame, prm, Seed::batch_counter, options_.max_n_kfs, seeds_updating_halt_, sub_batch_);
    total_uploaded_ += last_stats_.n_uploaded;
    if (last_stats_.n_device_errors) throw std::runtime_error(std::string("depth_filter_update: ") + svo_hip_last_error(ctx_));
  }

 public:
  int sub_batch_ = 4096;                         ///< seeds per device batch (the halt flag is polled in between)
  hip_bridge::SeedBatchStats last_stats_;
  long total_uploaded_ = 0;                      ///< seeds ever mirrored on the device: every seed exactly once
 protected:
  DetectorGrid* feature_detector_ = nullptr;
  callback_t seed_converged_cb_;
  std::list<Seed> seeds_;
  hip_bridge::DeviceSeedMirror<std::list<Seed> > mirror_;    // the seeds' device-resident state
  std::mutex seeds_mut_;
  volatile bool seeds_updating_halt_ = false;
  bool thread_stop_ = false;
  std::thread* thread_ = nullptr;
  std::queue<FramePtr> frame_queue_;
  std::mutex frame_queue_mut_;
  std::condition_variable frame_queue_cond_;
  FramePtr new_keyframe_;
  std::vector<Feature*> new_keyframe_features_;
  bool new_keyframe_set_ = false, busy_ = false;
  double new_keyframe_min_depth_ = 0.0, new_keyframe_mean_depth_ = 0.0;
  svo_hip_ctx* ctx_ = nullptr;
  std::unique_ptr<hip_bridge::PyramidCache> kf_pyr_, cur_pyr_;
};

}  // namespace svo

#endif  // SVO_HOST_H_
