// depth_filter_batch.h -- DepthFilter::updateSeeds (S/depth_filter.cpp:237-341) batched for the device, written ONCE
// against a small Host policy so that the drop-in for the reference's own types (depth_filter_hip.cpp) and the
// executable host layer on minimal types (android_svo_amd/host/svo_host.h, run on the GPU by
// tests/test_gpu_host_cpp.py) execute the same code.  Depends on svo_hip.h and the standard library only.
//
// What the reference does seed by seed, in list order, and how this keeps it:
//   * age-out of old seeds (:256-261), halt flag polled per seed (:253)            -> pass 1, identical
//   * visibility test, findEpipolarMatchDirect, computeTau, updateSeed (:264-299)   -> device, per reference keyframe,
//     in sub-batches of at most `sub_batch` seeds; the halt flag is polled between sub-batches, and what the
//     finished sub-batches computed is applied before returning (a seed is either updated by this frame or not,
//     as in the reference, where the prefix that got updated also depends on when the flag rises)
//   * on keyframes: feature_detector_->setGridOccpuancy(matcher_.px_cur_) for every updated seed (:302-306),
//     convergence -> new Point + seed_converged_cb_ + erase (:310-331), NaN -> erase (:333-337)
//                                                                                   -> pass 2, over the list IN LIST
//     ORDER, so callbacks (candidate-list insertion order) and erasures happen in the reference's order whatever
//     order the device batches ran in.  Keyframe buckets are processed in first-appearance order (deterministic; the
//     results do not depend on it).
//
// Host policy (duck-typed; see the two users):
//   Frame*  keyframeOf(const Seed&)                       it->ftr->frame
//   void    feature(const Seed&, double px[2], double f[3], int* level)
//   void    pose7(const Frame&, double T[7])              {t, q(xyzw)} of T_f_w_
//   int     keyframeSlot(Frame&), currentSlot(Frame&)     device pyramid slots (negative: unavailable)
//   svo_hip_pyramid* keyframePyramids(), currentPyramids()
//   svo_hip_camera camera(const Frame&)
//   bool    isKeyframe(const Frame&)
//   void    setGridOccupancy(const double px_cur[2])      feature_detector_->setGridOccpuancy
//   void    converged(Seed&, const double xyz_world[3])   new Point, ftr->point, seed_converged_cb_(point, sigma2)
#ifndef SVO_DROPIN_DEPTH_FILTER_BATCH_H_
#define SVO_DROPIN_DEPTH_FILTER_BATCH_H_

#include <cstddef>
#include <cstdint>
#include <vector>

#include "svo_hip.h"

namespace svo {
namespace hip_bridge {

struct SeedBatchStats {
  int n_seeds = 0, n_aged_out = 0, n_updated = 0, n_failed_matches = 0, n_converged = 0, n_nan = 0;
  int n_device_calls = 0, n_device_errors = 0;
  bool halted = false;
};

template <class Host, class SeedList, class Frame>
SeedBatchStats updateSeedsBatched(Host& host, svo_hip_ctx* ctx, SeedList& seeds, Frame& frame, const svo_hip_df_params& prm,
                                  int batch_counter, int max_n_kfs, const volatile bool& halt, int sub_batch = 4096) {
  typedef typename SeedList::iterator It;
  SeedBatchStats st;
  if (sub_batch < 1) sub_batch = 1;

  // ---- pass 1: age-out and indexing, in list order
  struct Item {
    It it;
    int32_t status;
    bool done;
    float a, b, mu, sigma2;
    double xyz[3], px_cur[2];
  };
  std::vector<Item> items;
  std::vector<decltype(host.keyframeOf(*seeds.begin()))> kfs;      // first-appearance order
  std::vector<std::vector<int> > buckets;
  for (It it = seeds.begin(); it != seeds.end();) {
    if (halt) { st.halted = true; return st; }
    if ((batch_counter - it->batch_id) > max_n_kfs) { it = seeds.erase(it); ++st.n_aged_out; continue; }
    auto kf = host.keyframeOf(*it);
    size_t k = kfs.size();
    while (k > 0 && kfs[k - 1] != kf) --k;                         // newest first: seeds of a keyframe are contiguous
    if (k == 0) { kfs.push_back(kf); buckets.push_back(std::vector<int>()); k = kfs.size(); }
    buckets[k - 1].push_back((int)items.size());
    Item item;
    item.it = it; item.status = -1; item.done = false;
    items.push_back(item);
    ++it;
  }
  st.n_seeds = (int)items.size();
  if (items.empty()) return st;

  const svo_hip_camera cam = host.camera(frame);
  const int cur_slot = host.currentSlot(frame);
  if (cur_slot < 0) return st;
  double T_cur[7];
  host.pose7(frame, T_cur);

  // ---- device: per reference keyframe, sub-batches of at most sub_batch seeds
  std::vector<double> px, f, z, xyz, px_cur;
  std::vector<int32_t> level, status;
  std::vector<float> a, b, mu, zr, s2;
  for (size_t k = 0; k < kfs.size() && !st.halted; ++k) {
    const std::vector<int>& ids = buckets[k];
    const int ref_slot = host.keyframeSlot(*kfs[k]);
    if (ref_slot < 0) continue;
    double T_ref[7];
    host.pose7(*kfs[k], T_ref);
    for (size_t first = 0; first < ids.size(); first += (size_t)sub_batch) {
      if (halt) { st.halted = true; break; }
      const int n = (int)((ids.size() - first < (size_t)sub_batch) ? ids.size() - first : (size_t)sub_batch);
      px.resize(2 * (size_t)n); f.resize(3 * (size_t)n); z.resize((size_t)n); xyz.resize(3 * (size_t)n);
      px_cur.resize(2 * (size_t)n); level.resize((size_t)n); status.resize((size_t)n);
      a.resize((size_t)n); b.resize((size_t)n); mu.resize((size_t)n); zr.resize((size_t)n); s2.resize((size_t)n);
      for (int i = 0; i < n; ++i) {
        const It it = items[(size_t)ids[first + (size_t)i]].it;
        int lvl = 0;
        host.feature(*it, &px[2 * (size_t)i], &f[3 * (size_t)i], &lvl);
        level[(size_t)i] = lvl;
        a[(size_t)i] = it->a; b[(size_t)i] = it->b; mu[(size_t)i] = it->mu; zr[(size_t)i] = it->z_range; s2[(size_t)i] = it->sigma2;
      }
      ++st.n_device_calls;
      const int rc = svo_hip_depth_filter_update(ctx, host.keyframePyramids(), ref_slot, host.currentPyramids(), cur_slot, &cam,
                                                 T_ref, T_cur, n, px.data(), f.data(), level.data(), a.data(), b.data(),
                                                 mu.data(), zr.data(), s2.data(), &prm, status.data(), z.data(), xyz.data(),
                                                 NULL, NULL, px_cur.data(), NULL);
      if (rc != SVO_HIP_OK) { ++st.n_device_errors; continue; }    // device error: these seeds keep their old state
      for (int i = 0; i < n; ++i) {
        Item& item = items[(size_t)ids[first + (size_t)i]];
        item.done = true;
        item.status = status[(size_t)i];
        item.a = a[(size_t)i]; item.b = b[(size_t)i]; item.mu = mu[(size_t)i]; item.sigma2 = s2[(size_t)i];
        for (int c = 0; c < 3; ++c) item.xyz[c] = xyz[3 * (size_t)i + (size_t)c];
        item.px_cur[0] = px_cur[2 * (size_t)i]; item.px_cur[1] = px_cur[2 * (size_t)i + 1];
      }
    }
  }

  // ---- pass 2: apply in list order
  const bool is_keyframe = host.isKeyframe(frame);
  for (size_t j = 0; j < items.size(); ++j) {
    Item& item = items[j];
    if (!item.done) continue;
    It it = item.it;
    it->a = item.a; it->b = item.b; it->mu = item.mu; it->sigma2 = item.sigma2;
    if (item.status == SVO_HIP_SEED_NO_MATCH) ++st.n_failed_matches;
    if (item.status < SVO_HIP_SEED_UPDATED) continue;
    ++st.n_updated;
    if (is_keyframe) host.setGridOccupancy(item.px_cur);           // :302-306
    if (item.status == SVO_HIP_SEED_CONVERGED) {                   // :310-331
      host.converged(*it, item.xyz);
      seeds.erase(it);
      ++st.n_converged;
    } else if (item.status == SVO_HIP_SEED_NAN) {                  // :333-337
      seeds.erase(it);
      ++st.n_nan;
    }
  }
  return st;
}

}  // namespace hip_bridge
}  // namespace svo

#endif  // SVO_DROPIN_DEPTH_FILTER_BATCH_H_
