// host_demo.cpp -- runs the C++ host layer (svo_host.h) end to end on the GPU: one SparseImgAlign::run,
// the DepthFilter protocol synchronously, then again with its worker thread while the main thread keeps
// aligning frames (two host threads, two svo_hip contexts, as in the reference: SURVEY 8b "Threading").
// Usage: svo_host_demo <case_dir> <out_dir>      (inputs written by tests/test_gpu_host_cpp.py)
//        svo_host_demo <case_dir> <out_dir> track   the tracking chain: svo::FrameTracker (hip_bridge::FrameTrackerT on this
//                                                   file's data model) over a map built as an object graph from index tables
//        svo_host_demo <case_dir> <out_dir> trackgroup [n]   the same tracking chain for n (default 2) copies of the world at once through
//                                                   svo::FrameTrackerGroup (hip_bridge::FrameTrackerGroupT, svo_hip_tracker_group): outputs of
//                                                   world w under <out_dir>/g<w>/, each equal to the lone tracker's
//        svo_host_demo <case_dir> <out_dir> churn   SURVEY 8(b) "Threading": latency of SparseImgAlign::run on the tracking
//                                                   thread while a second thread (own context) creates seed batches, runs a
//                                                   pass and drops them at keyframe rate -- and with that thread idle
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <sys/stat.h>

#include "svo_host.h"

using namespace svo;

template <typename T>
static std::vector<T> read_bin(const std::string& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("cannot open " + path);
  const std::streamsize n = f.tellg();
  f.seekg(0);
  std::vector<T> v((size_t)n / sizeof(T));
  f.read(reinterpret_cast<char*>(v.data()), n);
  return v;
}
template <typename T>
static void write_bin(const std::string& path, const std::vector<T>& v) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

struct Case {
  PinholeCamera cam;
  int n_levels, n_frames;
  std::vector<FramePtr> frames;
};

static FramePtr load_frame(const std::string& dir, const PinholeCamera* cam, int k, int n_levels) {
  std::vector<std::vector<uint8_t>> pyr;
  for (int l = 0; l < n_levels; ++l) pyr.push_back(read_bin<uint8_t>(dir + "/frame_" + std::to_string(k) + "_L" + std::to_string(l) + ".bin"));
  FramePtr f = std::make_shared<Frame>(cam, std::move(pyr));
  f->T_f_w_ = SE3(read_bin<double>(dir + "/frame_" + std::to_string(k) + "_pose.bin").data());
  return f;
}

static void dump_filter(DepthFilter& df, const std::map<Feature*, int>& index, const std::vector<double>& conv,
                        const std::string& out, const std::string& tag) {
  std::vector<double> rows;
  for (const Seed& s : df.getSeeds()) {
    rows.push_back((double)index.at(s.ftr)); rows.push_back(s.a); rows.push_back(s.b); rows.push_back(s.mu); rows.push_back(s.sigma2);
  }
  write_bin(out + "/" + tag + "_seeds.bin", rows);
  write_bin(out + "/" + tag + "_conv.bin", conv);
}

// ---- svo::FrameTracker over a svo::Map: the object graph (frames, features, points with their observation lists,
// point candidates) is built from the index tables the test wrote, the frames are tracked one after the other (each
// tracked frame is the next one's last frame), and what the tracker left on the objects is written out as tables again.
// What track_demo tracks through: a lone svo::FrameTracker, or camera w of a svo::FrameTrackerGroup.  In a group the worlds run on
// one thread each but in lockstep: a thread only runs while it holds the rendezvous' mutex (the C-ABI context is one host thread's
// at a time), gives it up inside track() and goes on when the group call -- made by the last world to arrive -- has returned.
typedef hip_bridge::FrameTrackerT<HostTrackerPolicy> TrackerCamera;
struct GroupRendezvous {
  std::mutex mu;
  std::condition_variable cv;
  FrameTrackerGroup* group = nullptr;
  size_t n = 0, arrived = 0, failed = 0;                // failed: worlds that gave up (the others must not wait for them)
  unsigned long generation = 0;
  bool ok = true;
  std::vector<FramePtr> last, cur;
  std::vector<Map*> maps;
  std::vector<std::vector<std::pair<FramePtr, size_t>>> overlap;
  std::vector<TrackerCamera::Outcome> out;
};
struct TrackerPort {
  FrameTracker* lone = nullptr;
  GroupRendezvous* rz = nullptr;
  std::unique_lock<std::mutex>* baton = nullptr;       // the world thread's hold on rz->mu
  size_t w = 0;
  TrackerCamera& camera() { return lone ? static_cast<TrackerCamera&>(*lone) : rz->group->camera(w); }
  bool track(const FramePtr& last, const FramePtr& cur, Map& map, std::vector<std::pair<FramePtr, size_t>>& overlap, TrackerCamera::Outcome& oc) {
    if (lone) return lone->track(last, cur, map, overlap, oc);
    GroupRendezvous& r = *rz;
    if (r.failed) return false;
    r.last[w] = last; r.cur[w] = cur; r.maps[w] = &map;
    const unsigned long my_gen = r.generation;
    if (++r.arrived == r.n) {
      r.ok = r.group->trackAll(r.last, r.cur, r.maps, r.overlap, r.out);
      r.arrived = 0;
      ++r.generation;
      r.cv.notify_all();
    } else {
      r.cv.wait(*baton, [&]() { return r.generation != my_gen; });
      if (r.failed) return false;
    }
    if (r.ok) { overlap = r.overlap[w]; oc = r.out[w]; }
    return r.ok;
  }
};

static svo_hip_tracker_config track_config(const std::vector<double>& m) {
  svo_hip_tracker_config cfg;
  svo_hip_tracker_default_config(&cfg);
  cfg.max_keyframes = (int)m[6] + 2;                                          // room for a frame that becomes a keyframe
  cfg.grid_size = (int)m[12]; cfg.max_fts = (int)m[13]; cfg.quality_min_fts = (int)m[14];
  cfg.klt_min_level = (int)m[15]; cfg.max_frame_features = (int)m[16];
  return cfg;
}

static int track_demo(const std::string& dir, const std::string& out, TrackerPort& port) {
  const std::vector<double> m = read_bin<double>(dir + "/track_manifest.bin");
  PinholeCamera cam{(int)m[0], (int)m[1], m[2], m[3], m[4], m[5]};
  const int n_kf = (int)m[6], n_points = (int)m[7], n_obs = (int)m[8], n_cand = (int)m[10], n_frames = (int)m[11];
  const auto kf_pose = read_bin<double>(dir + "/kf_pose.bin"), pt_pos = read_bin<double>(dir + "/pt_pos.bin");
  const auto pt_type = read_bin<int32_t>(dir + "/pt_type.bin"), pt_failed = read_bin<int32_t>(dir + "/pt_n_failed.bin"),
             pt_succ = read_bin<int32_t>(dir + "/pt_n_succeeded.bin"), pt_obs_offset = read_bin<int32_t>(dir + "/pt_obs_offset.bin"),
             obs_point = read_bin<int32_t>(dir + "/obs_point.bin"), obs_kf = read_bin<int32_t>(dir + "/obs_kf.bin"),
             obs_level = read_bin<int32_t>(dir + "/obs_level.bin"), kf_ftr_offset = read_bin<int32_t>(dir + "/kf_ftr_offset.bin"),
             kf_ftr_obs = read_bin<int32_t>(dir + "/kf_ftr_obs.bin"), cand_obs = read_bin<int32_t>(dir + "/cand_obs.bin");
  const auto obs_px = read_bin<double>(dir + "/obs_px.bin"), obs_f = read_bin<double>(dir + "/obs_f.bin"), obs_grad = read_bin<double>(dir + "/obs_grad.bin");
  const auto obs_edgelet = read_bin<uint8_t>(dir + "/obs_edgelet.bin");
  if ((int)obs_point.size() != n_obs || (int)pt_type.size() != n_points || (int)cand_obs.size() != n_cand) throw std::runtime_error("track case: table sizes");

  Map map;
  std::vector<std::unique_ptr<Point>> points;
  std::map<const Point*, int> index_of_point;
  for (int p = 0; p < n_points; ++p) {
    points.emplace_back(new Point(Vector3d{{pt_pos[3 * p], pt_pos[3 * p + 1], pt_pos[3 * p + 2]}}));
    points.back()->type_ = (Point::PointType)pt_type[p];
    points.back()->n_failed_reproj_ = pt_failed[p];
    points.back()->n_succeeded_reproj_ = pt_succ[p];
    index_of_point[points.back().get()] = p;
  }
  std::vector<FramePtr> kfs;
  for (int k = 0; k < n_kf; ++k) {
    std::vector<std::vector<uint8_t>> pyr;
    pyr.push_back(read_bin<uint8_t>(dir + "/kf_" + std::to_string(k) + "_img.bin"));
    kfs.push_back(std::make_shared<Frame>(&cam, std::move(pyr)));
    kfs.back()->T_f_w_ = SE3(&kf_pose[7 * (size_t)k]);
  }
  std::vector<Feature*> feature_of_obs((size_t)n_obs, nullptr);
  auto make_obs_feature = [&](int o) {
    Feature* ftr = new Feature(kfs[obs_kf[o]].get(), Vector2d{{obs_px[2 * o], obs_px[2 * o + 1]}},
                               Vector3d{{obs_f[3 * o], obs_f[3 * o + 1], obs_f[3 * o + 2]}}, obs_level[o]);
    if (obs_edgelet[o]) { ftr->type = Feature::EDGELET; ftr->grad = Vector2d{{obs_grad[2 * o], obs_grad[2 * o + 1]}}; }
    ftr->point = points[obs_point[o]].get();
    feature_of_obs[o] = ftr;
    return ftr;
  };
  for (int k = 0; k < n_kf; ++k)                                               // Frame::fts_ in the keyframe's own order
    for (int j = kf_ftr_offset[k]; j < kf_ftr_offset[k + 1]; ++j) kfs[k]->addFeature(make_obs_feature(kf_ftr_obs[j]));
  for (int c = 0; c < n_cand; ++c) {                                           // candidates: one feature each, in no fts_
    Feature* ftr = make_obs_feature(cand_obs[c]);
    map.point_candidates_.candidates_.push_back(MapPointCandidates::PointCandidate(ftr->point, ftr));
  }
  for (int p = 0; p < n_points; ++p)                                           // Point::obs_ in table order
    for (int o = pt_obs_offset[p]; o < pt_obs_offset[p + 1]; ++o)
      if (feature_of_obs[o]) points[p]->obs_.push_back(feature_of_obs[o]);
  for (int k = 0; k < n_kf; ++k) { kfs[k]->setKeyframe(); map.addKeyframe(kfs[k]); }
  auto key_table = [&]() {
    std::vector<int32_t> key;
    for (int k = 0; k < n_kf; ++k)
      for (int j = 0; j < 5; ++j) { const Feature* kp = kfs[k]->key_pts_[j]; key.push_back(kp && kp->point ? index_of_point.at(kp->point) : -1); }
    return key;
  };
  write_bin(out + "/track_key_before.bin", key_table());

  TrackerCamera& tracker = port.camera();
  if (!tracker.ok()) throw std::runtime_error("svo::FrameTracker: no device tracker");

  // the last frame: a keyframe of the map, or a frame of its own without features (SparseImgAlign::run then returns at once)
  const int last_kf = (int)m[17];
  FramePtr last;
  if (last_kf >= 0) {
    last = kfs[last_kf];
  } else {
    std::vector<std::vector<uint8_t>> pyr;
    pyr.push_back(read_bin<uint8_t>(dir + "/last_img.bin"));
    last = std::make_shared<Frame>(&cam, std::move(pyr));
    last->T_f_w_ = SE3(read_bin<double>(dir + "/last_pose.bin").data());
  }
  std::vector<double> poses, stats, overlap, uploads;
  for (int k = 0; k < n_frames; ++k) {
    std::vector<std::vector<uint8_t>> pyr;
    pyr.push_back(read_bin<uint8_t>(dir + "/trk_frame_" + std::to_string(k) + ".bin"));
    FramePtr cur = std::make_shared<Frame>(&cam, std::move(pyr));
    std::vector<std::pair<FramePtr, size_t>> overlap_kfs;
    FrameTracker::Outcome oc;
    if (!port.track(last, cur, map, overlap_kfs, oc)) throw std::runtime_error("svo::FrameTracker::track failed at frame " + std::to_string(k));
    poses.insert(poses.end(), cur->T_f_w_.p, cur->T_f_w_.p + 7);
    stats.insert(stats.end(), {(double)cur->fts_.size(), (double)oc.repr_n_matches, (double)oc.repr_n_trials, (double)oc.img_align_n_tracked,
                               oc.pose_optimised ? 1.0 : 0.0, (double)oc.sfba_n_edges_final, oc.sfba_error_init, oc.sfba_error_final, (double)overlap_kfs.size()});
    std::vector<double> fpx, fgrad;
    std::vector<int32_t> flevel, fpoint;
    std::vector<uint8_t> fedge;
    for (const Feature* ftr : cur->fts_) {
      fpx.push_back(ftr->px[0]); fpx.push_back(ftr->px[1]);
      fgrad.push_back(ftr->grad[0]); fgrad.push_back(ftr->grad[1]);
      flevel.push_back(ftr->level);
      fpoint.push_back(ftr->point ? index_of_point.at(ftr->point) : -1);
      fedge.push_back(ftr->type == Feature::EDGELET ? 1 : 0);
    }
    const std::string tag = out + "/track_feat_" + std::to_string(k);
    write_bin(tag + "_px.bin", fpx); write_bin(tag + "_grad.bin", fgrad); write_bin(tag + "_level.bin", flevel);
    write_bin(tag + "_point.bin", fpoint); write_bin(tag + "_edgelet.bin", fedge);
    if (k == 0)
      for (const auto& ov : overlap_kfs) {
        int idx = -1;
        for (int j = 0; j < n_kf; ++j) if (kfs[j] == ov.first) idx = j;
        overlap.push_back((double)idx); overlap.push_back((double)ov.second);
      }
    if (k == 0 && m.size() > 18 && m[18] > 0) {
      // FrameHandlerBase::optimizeStructure(new_frame, 20, 5) as processFrame calls it behind the pose refinement (:242-244)
      std::vector<const Point*> had;
      for (const Feature* ftr : cur->fts_) if (ftr->point) had.push_back(ftr->point);
      if (!tracker.optimiseStructure(cur, map, (size_t)m[18], 5)) throw std::runtime_error("svo::FrameTracker::optimiseStructure failed");
      std::vector<double> so;                                                  // point index, x, y, z of every point that was optimised
      for (int p = 0; p < n_points; ++p)
        if (points[p]->last_structure_optim_ == cur->id_) so.insert(so.end(), {(double)p, points[p]->pos_[0], points[p]->pos_[1], points[p]->pos_[2]});
      write_bin(out + "/track_structure_first.bin", so);
    }
    if (k == 0) {                                                              // the map's points as the first frame left them
      std::vector<int32_t> st;
      for (int p = 0; p < n_points; ++p) { st.push_back((int32_t)points[p]->type_); st.push_back(points[p]->n_failed_reproj_); st.push_back(points[p]->n_succeeded_reproj_); }
      write_bin(out + "/track_points_after_first.bin", st);
      write_bin(out + "/track_key_after_first.bin", key_table());
      std::vector<int32_t> linked;                                             // features that still refer to their point, per observation
      for (int o = 0; o < n_obs; ++o) {
        // (a deleted candidate's feature is gone: its observation counts as unlinked)
        bool cand_alive = true;
        if (points[obs_point[o]]->type_ == Point::TYPE_DELETED && pt_type[obs_point[o]] == (int)Point::TYPE_CANDIDATE) cand_alive = false;
        linked.push_back(cand_alive && feature_of_obs[o] && feature_of_obs[o]->point != nullptr ? 1 : 0);
      }
      write_bin(out + "/track_obs_linked_after_first.bin", linked);
    }
    uploads.push_back((double)tracker.mapUploads());
    if (m.size() > 20 && (int)m[20] == k && !map.point_candidates_.candidates_.empty()) {
      // what the depth filter's thread does when a seed converges (MapPointCandidates::newCandidatePoint): a new candidate
      // appears in the list behind the tracker's back -- here a twin of the list's first one
      const MapPointCandidates::PointCandidate& c0 = map.point_candidates_.candidates_.front();
      points.emplace_back(new Point(c0.first->pos_));
      Point* np_ = points.back().get();
      np_->type_ = Point::TYPE_CANDIDATE;
      Feature* nf = new Feature(c0.second->frame, c0.second->px, c0.second->f, c0.second->level);
      nf->point = np_;
      np_->obs_.push_front(nf);
      index_of_point[np_] = (int)points.size() - 1;
      std::unique_lock<std::mutex> lock(map.point_candidates_.mut_);
      map.point_candidates_.candidates_.push_back(MapPointCandidates::PointCandidate(np_, nf));
    }
    if (m.size() > 19 && (int)m[19] == k) {
      // FrameHandlerMono::processFrame :284-330: the tracked frame becomes a keyframe -- setKeyframe (key points), every feature
      // with a point becomes an observation of it (Point::addFrameRef: front of obs_), the map takes the frame, the device
      // keeps its pyramid; the tracker flattens the grown map before the next frame
      cur->setKeyframe();
      for (Feature* ftr : cur->fts_) if (ftr->point != nullptr) ftr->point->addFrameRef(ftr);
      map.addKeyframe(cur);
      kfs.push_back(cur);
      if (!tracker.lastFrameBecameKeyframe(*cur)) throw std::runtime_error("svo::FrameTracker::lastFrameBecameKeyframe failed");
    }
    last = cur;
  }
  write_bin(out + "/track_poses.bin", poses);
  write_bin(out + "/track_stats.bin", stats);
  write_bin(out + "/track_uploads.bin", uploads);
  write_bin(out + "/track_overlap_first.bin", overlap);
  if (port.lone) std::printf("svo_host_demo track OK\n");
  return 0;
}

static int track_lone(const std::string& dir, const std::string& out) {
  const std::vector<double> m = read_bin<double>(dir + "/track_manifest.bin");
  PinholeCamera cam{(int)m[0], (int)m[1], m[2], m[3], m[4], m[5]};
  FrameTracker tracker(cam, track_config(m));
  TrackerPort port;
  port.lone = &tracker;
  return track_demo(dir, out, port);
}

// n copies of the world, one svo::FrameTrackerGroup: every world's outputs must equal the lone tracker's
static int track_group(const std::string& dir, const std::string& out, int n) {
  const std::vector<double> m = read_bin<double>(dir + "/track_manifest.bin");
  PinholeCamera cam{(int)m[0], (int)m[1], m[2], m[3], m[4], m[5]};
  svo_hip_tracker_config cfg = track_config(m);
  cfg.max_items = (cfg.max_items + 15) / 16 * 16;
  FrameTrackerGroup group(cam, cfg, n);
  if (!group.ok()) throw std::runtime_error("svo::FrameTrackerGroup: no device tracker group");
  GroupRendezvous rz;
  rz.group = &group; rz.n = (size_t)n;
  rz.last.resize((size_t)n); rz.cur.resize((size_t)n); rz.maps.resize((size_t)n); rz.overlap.resize((size_t)n); rz.out.resize((size_t)n);
  std::vector<std::string> errors((size_t)n);
  for (int w = 0; w < n; ++w) (void)::mkdir((out + "/g" + std::to_string(w)).c_str(), 0755);
  std::vector<std::thread> worlds;
  for (int w = 0; w < n; ++w)
    worlds.emplace_back([&, w]() {
      std::unique_lock<std::mutex> baton(rz.mu);                               // this world runs only while it holds the baton
      TrackerPort port;
      port.rz = &rz; port.baton = &baton; port.w = (size_t)w;
      try {
        track_demo(dir, out + "/g" + std::to_string(w), port);
      } catch (const std::exception& e) {
        errors[(size_t)w] = e.what();
        rz.ok = false; ++rz.failed; ++rz.generation; rz.cv.notify_all();      // let the others go (they fail at their next frame)
      }
    });
  for (std::thread& t : worlds) t.join();
  for (const std::string& e : errors) if (!e.empty()) throw std::runtime_error(e);
  std::printf("svo_host_demo trackgroup OK (%d worlds)\n", n);
  return 0;
}

// ---- the tracking thread's SparseImgAlign::run while the depth-filter thread creates and drops device seed batches.
// A keyframe's seeds are a svo_hip_seed_batch created when the keyframe is first seen and destroyed when its seeds age out
// or are used up (S/depth_filter.cpp:129-151,256-261): allocator traffic on the depth-filter thread at keyframe rate.
// hipFree synchronises the whole device; with the per-context pool (svo_hip_ctx_info) nothing is allocated or freed after
// warm-up.  Prints median / 99th percentile / worst latency of run() with the second thread idle and busy.
static int churn_demo(const std::string& dir, const std::string& out) {
  const std::vector<double> m = read_bin<double>(dir + "/manifest.bin");
  PinholeCamera cam{(int)m[0], (int)m[1], m[2], m[3], m[4], m[5]};
  const int n_levels = (int)m[6];
  const auto px = read_bin<double>(dir + "/sia_px.bin"), f = read_bin<double>(dir + "/sia_f.bin"), pos = read_bin<double>(dir + "/sia_pos.bin");
  const auto has = read_bin<uint8_t>(dir + "/sia_has.bin");
  FramePtr ref = load_frame(dir, &cam, 0, n_levels);
  std::vector<std::unique_ptr<Point>> points;
  for (size_t i = 0; i < has.size(); ++i) {
    Feature* ftr = new Feature(ref.get(), Vector2d{{px[2 * i], px[2 * i + 1]}}, Vector3d{{f[3 * i], f[3 * i + 1], f[3 * i + 2]}}, 0);
    if (has[i]) { points.emplace_back(new Point(Vector3d{{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}})); ftr->point = points.back().get(); }
    ref->fts_.push_back(ftr);
  }
  FramePtr cur = load_frame(dir, &cam, 1, n_levels);
  const SE3 T_start = ref->T_f_w_;
  SparseImgAlign align(4, 2, 30, SparseImgAlign::GaussNewton, false, false);      // the shipping range L4..L2
  auto one_run = [&]() {
    cur->T_f_w_ = T_start;
    const auto t0 = std::chrono::steady_clock::now();
    align.run(ref, cur);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  };
  for (int k = 0; k < 20; ++k) one_run();
  const auto spx = read_bin<double>(dir + "/seed_px.bin"), sf = read_bin<double>(dir + "/seed_f.bin");
  const auto slevel = read_bin<int32_t>(dir + "/seed_level.bin");
  const int n_all = (int)slevel.size();
  std::atomic<int> mode{0};                  // 0 idle, 1 churn, 2 stop
  std::atomic<long> n_created{0};
  svo_hip_ctx_stats st_before{}, st_after{};
  std::vector<double> create_us;            // per keyframe on the second thread: svo_hip_seed_batch_create (+ the drop of the oldest batch)
  std::thread filter([&]() {
    svo_hip_ctx* ctx = nullptr;
    if (svo_hip_ctx_create(&ctx, 0, nullptr) != SVO_HIP_OK) return;
    std::vector<float> a((size_t)n_all, 10.f), b((size_t)n_all, 10.f), mu((size_t)n_all, 0.5f), zr((size_t)n_all, 1.f), s2((size_t)n_all, 1.f / 36.f);
    std::vector<svo_hip_seed_batch*> alive;
    unsigned rng = 12345u;
    auto keyframe = [&]() {                  // a new keyframe's seeds come, the oldest keyframe's go
      rng = rng * 1664525u + 1013904223u;
      const int n = 300 + (int)((rng >> 8) % 700u);
      svo_hip_seed_batch* sb = nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      if (svo_hip_seed_batch_create(ctx, n < n_all ? n : n_all, spx.data(), sf.data(), slevel.data(), a.data(), b.data(), mu.data(), zr.data(), s2.data(), &sb) == SVO_HIP_OK) {
        alive.push_back(sb);
        ++n_created;
      }
      if (alive.size() > 3) { svo_hip_seed_batch_destroy(alive.front()); alive.erase(alive.begin()); }
      if (mode.load() == 1) create_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    };
    for (int k = 0; k < 64; ++k) keyframe(); // warm-up: the pool has seen the working set (both capacity classes, every size)
    svo_hip_ctx_info(ctx, &st_before);
    while (mode.load() != 2) {
      if (mode.load() == 1) keyframe();
      else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    svo_hip_ctx_info(ctx, &st_after);
    for (svo_hip_seed_batch* sb : alive) svo_hip_seed_batch_destroy(sb);
    svo_hip_ctx_destroy(ctx);
  });
  auto measure = [&](int n) {
    std::vector<double> us;
    for (int k = 0; k < n; ++k) us.push_back(one_run());
    std::sort(us.begin(), us.end());
    return std::vector<double>{us[us.size() / 2], us[us.size() * 99 / 100], us.back()};
  };
  std::this_thread::sleep_for(std::chrono::milliseconds(300));     // the filter thread's warm-up is over
  const std::vector<double> idle = measure(2000);
  const long c0 = n_created.load();
  mode = 1;
  const std::vector<double> busy = measure(2000);
  const long c1 = n_created.load();
  mode = 2;
  filter.join();
  std::printf("churn: SparseImgAlign::run (L4-L2, %zu features) on the tracking thread, us: median / p99 / worst\n", has.size());
  std::printf("churn:   second thread idle:                          %8.1f %8.1f %8.1f\n", idle[0], idle[1], idle[2]);
  std::printf("churn:   second thread creating + dropping batches:   %8.1f %8.1f %8.1f   (%ld batches created meanwhile)\n", busy[0], busy[1], busy[2], c1 - c0);
  std::printf("churn:   allocator calls of the second context during the measurement: %llu, free calls: %llu\n",
              st_after.allocator_calls - st_before.allocator_calls, st_after.free_calls - st_before.free_calls);
  std::sort(create_us.begin(), create_us.end());
  if (!create_us.empty())
    std::printf("churn:   a keyframe on the second thread (create a batch of 300-1000 seeds incl. its upload, drop the oldest), us: median %.1f  p99 %.1f\n",
                create_us[create_us.size() / 2], create_us[create_us.size() * 99 / 100]);
  write_bin(out + "/churn.bin", std::vector<double>{idle[0], idle[1], idle[2], busy[0], busy[1], busy[2], (double)(c1 - c0),
                                                    (double)(st_after.allocator_calls - st_before.allocator_calls),
                                                    (double)(st_after.free_calls - st_before.free_calls)});
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s case_dir out_dir [track|churn]\n", argv[0]); return 2; }
  const std::string dir = argv[1], out = argv[2];
  if (argc > 3 && std::string(argv[3]) == "trackgroup") {
    try {
      return track_group(dir, out, argc > 4 ? std::atoi(argv[4]) : 2);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "svo_host_demo FAILED: %s\n", e.what());
      return 1;
    }
  }
  if (argc > 3 && std::string(argv[3]) == "churn") {
    try {
      return churn_demo(dir, out);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "svo_host_demo FAILED: %s\n", e.what());
      return 1;
    }
  }
  if (argc > 3 && std::string(argv[3]) == "track") {
    try {
      return track_lone(dir, out);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "svo_host_demo FAILED: %s\n", e.what());
      return 1;
    }
  }
  try {
    const std::vector<double> m = read_bin<double>(dir + "/manifest.bin");   // w h fx fy cx cy n_levels n_frames
    Case c;
    c.cam = PinholeCamera{(int)m[0], (int)m[1], m[2], m[3], m[4], m[5]};
    c.n_levels = (int)m[6]; c.n_frames = (int)m[7];
    if (m.size() >= 13) for (int k = 0; k < 5; ++k) c.cam.d[k] = m[8 + (size_t)k];      // radtan coefficients k1 k2 p1 p2 k3
    for (int k = 0; k < c.n_frames; ++k) c.frames.push_back(load_frame(dir, &c.cam, k, c.n_levels));

    // ---- SparseImgAlign: frame 0 (with features + points) -> frame 1 starting from frame 0's pose
    {
      const auto px = read_bin<double>(dir + "/sia_px.bin"), f = read_bin<double>(dir + "/sia_f.bin"), pos = read_bin<double>(dir + "/sia_pos.bin");
      const auto has = read_bin<uint8_t>(dir + "/sia_has.bin");
      FramePtr ref = c.frames[0];
      std::vector<std::unique_ptr<Point>> points;
      for (size_t i = 0; i < has.size(); ++i) {
        Feature* ftr = new Feature(ref.get(), Vector2d{{px[2 * i], px[2 * i + 1]}}, Vector3d{{f[3 * i], f[3 * i + 1], f[3 * i + 2]}}, 0);
        if (has[i]) { points.emplace_back(new Point(Vector3d{{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}})); ftr->point = points.back().get(); }
        ref->fts_.push_back(ftr);
      }
      FramePtr cur = load_frame(dir, &c.cam, 1, c.n_levels);
      cur->T_f_w_ = ref->T_f_w_;                                    // processFrame: new pose starts at the last one (:175)
      SparseImgAlign align(4, 0, 30, SparseImgAlign::GaussNewton, false, false);
      const size_t n_tracked = align.run(ref, cur);
      std::vector<double> res(cur->T_f_w_.p, cur->T_f_w_.p + 7);
      res.push_back((double)n_tracked);
      const auto I = align.getFisherInformation();
      res.insert(res.end(), I.begin(), I.end());
      write_bin(out + "/sia.bin", res);
      // the same pair through NLLSSolver's other branches: Levenberg-Marquardt with setRobustCostFunction(MADScale, HuberWeight)
      {
        FramePtr cur_lm = load_frame(dir, &c.cam, 1, c.n_levels);
        cur_lm->T_f_w_ = ref->T_f_w_;
        SparseImgAlign lm(4, 0, 30, SparseImgAlign::LevenbergMarquardt, false, false);
        lm.setRobustCostFunction(SparseImgAlign::MADScale, SparseImgAlign::HuberWeight);
        const size_t n_lm = lm.run(ref, cur_lm);
        std::vector<double> r2(cur_lm->T_f_w_.p, cur_lm->T_f_w_.p + 7);
        r2.push_back((double)n_lm);
        r2.push_back(lm.getChi2()); r2.push_back((double)lm.scale_); r2.push_back(lm.mu_); r2.push_back(lm.nu_); r2.push_back(lm.stop_ ? 1.0 : 0.0);
        write_bin(out + "/sia_lm.bin", r2);
      }
      // an empty reference frame returns 0 and leaves the pose alone
      FramePtr empty_ref = load_frame(dir, &c.cam, 0, c.n_levels);
      FramePtr cur2 = load_frame(dir, &c.cam, 1, c.n_levels);
      const SE3 before = cur2->T_f_w_;
      if (align.run(empty_ref, cur2) != 0 || std::memcmp(before.p, cur2->T_f_w_.p, sizeof(before.p)) != 0) throw std::runtime_error("empty-frame contract violated");

      // ---- DepthFilter: keyframe 0 with seed batch A, frames 1..kf2-1, keyframe kf2 with seed batch B (the update on a
      // ---- keyframe marks the detector grid), then the remaining frames update seeds of BOTH keyframes
      const auto dm = read_bin<double>(dir + "/depth_mean_min.bin");
      const int kf2 = (int)read_bin<double>(dir + "/second_keyframe.bin")[0];
      struct SeedSet { std::vector<double> px, f; std::vector<int32_t> level; };
      SeedSet setA{read_bin<double>(dir + "/seed_px.bin"), read_bin<double>(dir + "/seed_f.bin"), read_bin<int32_t>(dir + "/seed_level.bin")};
      SeedSet setB{read_bin<double>(dir + "/seedB_px.bin"), read_bin<double>(dir + "/seedB_f.bin"), read_bin<int32_t>(dir + "/seedB_level.bin")};
      auto make_features = [&](const SeedSet& ss, Frame* kf, int id0, std::map<Feature*, int>& index) {
        std::vector<Feature*> fts;
        for (size_t i = 0; i < ss.level.size(); ++i) {
          fts.push_back(new Feature(kf, Vector2d{{ss.px[2 * i], ss.px[2 * i + 1]}}, Vector3d{{ss.f[3 * i], ss.f[3 * i + 1], ss.f[3 * i + 2]}}, ss.level[i]));
          index[fts.back()] = id0 + (int)i;
        }
        return fts;
      };
      const int nA = (int)setA.level.size();
      c.frames[0]->setKeyframe();
      c.frames[kf2]->setKeyframe();
      size_t align_runs = 0;
      std::vector<double> conv_count;
      auto run_protocol = [&](bool threaded, int sub_batch, const std::string& tag, int remove_a_after = -1) {
        Seed::batch_counter = 0;
        std::vector<double> conv;                 // per callback, in callback order: seed id, x, y, z, sigma2
        std::map<Feature*, int> index;
        DetectorGrid grid(c.cam.width, c.cam.height, 30);
        std::vector<Feature*> fa, fb;
        {
          DepthFilter df([&](Point* p, double s2) {
            conv.insert(conv.end(), {(double)index.at(p->obs_.front()), p->pos_[0], p->pos_[1], p->pos_[2], s2});
            delete p;
          }, &grid);
          df.sub_batch_ = sub_batch;
          fa = make_features(setA, c.frames[0].get(), 0, index);
          fb = make_features(setB, c.frames[kf2].get(), nA, index);
          auto wait_idle = [&]() {
            if (!threaded) return;
            while (!df.idle()) {
              FramePtr cur3 = load_frame(dir, &c.cam, 1, c.n_levels);
              cur3->T_f_w_ = ref->T_f_w_;
              if (align.run(ref, cur3) != n_tracked) throw std::runtime_error("alignment changed under concurrency");
              if (std::memcmp(cur3->T_f_w_.p, cur->T_f_w_.p, sizeof(double) * 7) != 0) throw std::runtime_error("pose changed under concurrency");
              ++align_runs;
            }
          };
          if (threaded) df.startThread();
          df.addKeyframe(c.frames[0], dm[0], dm[1], fa);
          wait_idle();
          std::vector<double> frame_us;           // synchronous protocols: what DepthFilter::addFrame (= updateSeeds) took, per frame
          for (int k = 1; k < c.n_frames; ++k) {
            const auto t0 = std::chrono::steady_clock::now();
            if (k == kf2) df.addKeyframe(c.frames[k], dm[0], dm[1], fb);
            else df.addFrame(c.frames[k]);
            if (!threaded) {                      // rows of: us, seeds on the device, uploaded by this call, converged, NaN, list re-read
              frame_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
              const svo::hip_bridge::SeedBatchStats& ls = df.last_stats_;
              frame_us.insert(frame_us.end(), {(double)ls.n_seeds, (double)ls.n_uploaded, (double)ls.n_converged, (double)ls.n_nan, ls.resynced ? 1.0 : 0.0});
            }
            wait_idle();
            if (k == remove_a_after) {
              // Map::removeKeyframe path (depth_filter.cpp:153-170): every seed of the SECOND keyframe leaves the list BEHIND the
              // device mirror's back (removeKeyframe is not virtual in the reference); a copy taken just before shows the synced state
              std::list<Seed> copy_a;
              df.getSeedsCopy(c.frames[kf2], copy_a);
              std::vector<double> rows;
              for (const Seed& s : copy_a) { rows.push_back((double)index.at(s.ftr)); rows.push_back(s.a); rows.push_back(s.b); rows.push_back(s.mu); rows.push_back(s.sigma2); }
              write_bin(out + "/" + tag + "_copy_b.bin", rows);
              df.removeKeyframe(c.frames[kf2]);
            }
          }
          if (threaded) df.stopThread();
          if (!threaded) write_bin(out + "/" + tag + "_frame_us.bin", frame_us);
          dump_filter(df, index, conv, out, tag);
        }
        std::vector<uint8_t> occ(grid.grid_occupancy_.begin(), grid.grid_occupancy_.end());
        write_bin(out + "/" + tag + "_grid.bin", occ);
        conv_count.push_back((double)conv.size() / 5);
        for (Feature* f2 : fa) delete f2;
        for (Feature* f2 : fb) delete f2;
      };
      run_protocol(false, 4096, "sync");          // (a) synchronous protocol (no thread)
      run_protocol(false, 700, "sync_small");     // (b) the same with small device sub-batches: identical results
      run_protocol(true, 4096, "thread");         // (c) worker thread, while this thread keeps running SparseImgAlign
      run_protocol(false, 1000, "remove", kf2 + 1);   // (d) the second keyframe removed one frame after it came: its seeds vanish from the list
      write_bin(out + "/summary.bin", std::vector<double>{(double)align_runs, conv_count[0], conv_count[1], conv_count[2], conv_count[3]});
    }
    std::printf("svo_host_demo OK\n");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "svo_host_demo FAILED: %s\n", e.what());
    return 1;
  }
}
