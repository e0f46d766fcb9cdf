// host_demo.cpp -- runs the C++ host layer (svo_host.h) end to end on the GPU: one SparseImgAlign::run,
// the DepthFilter protocol synchronously, then again with its worker thread while the main thread keeps
// aligning frames (two host threads, two svo_hip contexts, as in the reference: SURVEY 8b "Threading").
// Usage: svo_host_demo <case_dir> <out_dir>      (inputs written by tests/test_gpu_host_cpp.py)
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <string>

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

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s case_dir out_dir\n", argv[0]); return 2; }
  const std::string dir = argv[1], out = argv[2];
  try {
    const std::vector<double> m = read_bin<double>(dir + "/manifest.bin");   // w h fx fy cx cy n_levels n_frames
    Case c;
    c.cam = PinholeCamera{(int)m[0], (int)m[1], m[2], m[3], m[4], m[5]};
    c.n_levels = (int)m[6]; c.n_frames = (int)m[7];
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
      auto run_protocol = [&](bool threaded, int sub_batch, const std::string& tag) {
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
          for (int k = 1; k < c.n_frames; ++k) {
            if (k == kf2) df.addKeyframe(c.frames[k], dm[0], dm[1], fb);
            else df.addFrame(c.frames[k]);
            wait_idle();
          }
          if (threaded) df.stopThread();
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
      write_bin(out + "/summary.bin", std::vector<double>{(double)align_runs, conv_count[0], conv_count[1], conv_count[2]});
    }
    std::printf("svo_host_demo OK\n");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "svo_host_demo FAILED: %s\n", e.what());
    return 1;
  }
}
