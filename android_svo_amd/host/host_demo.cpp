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

      // ---- DepthFilter, keyframe = frame 0, seeds from the case file
      const auto spx = read_bin<double>(dir + "/seed_px.bin"), sf = read_bin<double>(dir + "/seed_f.bin");
      const auto slevel = read_bin<int32_t>(dir + "/seed_level.bin");
      const auto dm = read_bin<double>(dir + "/depth_mean_min.bin");
      const size_t n_seeds = slevel.size();
      auto make_features = [&](std::map<Feature*, int>& index) {
        std::vector<Feature*> fts;
        for (size_t i = 0; i < n_seeds; ++i) {
          fts.push_back(new Feature(c.frames[0].get(), Vector2d{{spx[2 * i], spx[2 * i + 1]}}, Vector3d{{sf[3 * i], sf[3 * i + 1], sf[3 * i + 2]}}, slevel[i]));
          index[fts.back()] = (int)i;
        }
        return fts;
      };
      // (a) synchronous protocol (no thread): addKeyframe -> initializeSeeds, addFrame -> updateSeeds
      std::vector<double> conv_a, conv_b;
      std::map<Feature*, int> index_a, index_b;
      std::vector<Feature*> fa, fb;
      {
        Seed::batch_counter = 0;
        std::map<Point*, int> dummy;
        DepthFilter df([&](Point* p, double s2) { conv_a.insert(conv_a.end(), {p->pos_[0], p->pos_[1], p->pos_[2], s2}); delete p; });
        fa = make_features(index_a);
        c.frames[0]->setKeyframe();
        df.addKeyframe(c.frames[0], dm[0], dm[1], fa);
        for (int k = 1; k < c.n_frames; ++k) df.addFrame(c.frames[k]);
        dump_filter(df, index_a, conv_a, out, "sync");
      }
      // (b) with the worker thread, while this thread keeps running SparseImgAlign (its own context)
      size_t align_runs = 0;
      {
        Seed::batch_counter = 0;
        DepthFilter df([&](Point* p, double s2) { conv_b.insert(conv_b.end(), {p->pos_[0], p->pos_[1], p->pos_[2], s2}); delete p; });
        fb = make_features(index_b);
        df.startThread();
        df.addKeyframe(c.frames[0], dm[0], dm[1], fb);
        auto wait_idle = [&]() {
          while (!df.idle()) {
            FramePtr cur3 = load_frame(dir, &c.cam, 1, c.n_levels);
            cur3->T_f_w_ = ref->T_f_w_;
            if (align.run(ref, cur3) != n_tracked) throw std::runtime_error("alignment changed under concurrency");
            if (std::memcmp(cur3->T_f_w_.p, cur->T_f_w_.p, sizeof(double) * 7) != 0) throw std::runtime_error("pose changed under concurrency");
            ++align_runs;
          }
        };
        wait_idle();
        for (int k = 1; k < c.n_frames; ++k) { df.addFrame(c.frames[k]); wait_idle(); }
        df.stopThread();
        dump_filter(df, index_b, conv_b, out, "thread");
      }
      write_bin(out + "/summary.bin", std::vector<double>{(double)align_runs, (double)conv_a.size() / 4, (double)conv_b.size() / 4});
      for (Feature* f2 : fa) delete f2;
      for (Feature* f2 : fb) delete f2;
    }
    std::printf("svo_host_demo OK\n");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "svo_host_demo FAILED: %s\n", e.what());
    return 1;
  }
}
