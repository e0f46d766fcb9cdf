// depth_filter_batch.h -- DepthFilter::updateSeeds (S/depth_filter.cpp:237-341) with the seeds RESIDENT ON THE DEVICE,
// written ONCE against a small Host policy so that the drop-in for the reference's own types (depth_filter_hip.cpp) and
// the executable host layer on minimal types (android_svo_amd/host/svo_host.h, run on the GPU by
// tests/test_gpu_host_cpp.py) execute the same code.  Depends on svo_hip.h and the standard library only.
//
// The reference keeps its seeds in a std::list<Seed> and walks it once per frame.  Here every keyframe's seeds are
// mirrored on the device when they are first seen (svo_hip_seed_batch_create: Feature px / f / level and the Seed
// constructor's state, uploaded ONCE, in chunks of at most `sub_batch` seeds), and a frame costs two poses down and the
// seeds the host has to act on back (round 3 moved 144 B per seed over the link every frame):
//   * age-out of old seeds (:256-261)                     -> whole device batches dropped, their list entries erased
//   * visibility test, findEpipolarMatchDirect, computeTau, updateSeed (:264-299)
//                                                         -> device, all batches of the frame through one set of launches
//     (svo_hip_seed_batch_update_group_async), awaited once; the halt flag (:253) is polled before the frame's pass is
//     enqueued (a seed is either updated by this frame or not, as in the reference, where the prefix that got updated
//     also depends on when the flag rises: here the prefix is everything or nothing)
//   * on keyframes feature_detector_->setGridOccpuancy(matcher_.px_cur_) for every updated seed (:302-306),
//     convergence -> new Point + seed_converged_cb_ + erase (:310-331), NaN -> erase (:333-337)
//                                                         -> the batch's EVENTS, ascending seed index, batches in list
//     order: the callbacks (candidate-list insertion order), grid marks and erasures happen in the reference's order.
// The list entries of live seeds keep their CONSTRUCTION-time a / b / mu / sigma2 until syncToHost() copies the device
// state back (getSeeds() / getSeedsCopy() callers; the reference app has none) -- a converged seed's mu / sigma2 are
// written before its callback, which is all the reference's callback reads.
// Erasures behind the mirror's back (DepthFilter::removeKeyframe, reset: non-virtual in the reference) are noticed at the
// next call -- the list's size or its last seed id no longer match -- and the mirror is re-built from the list: seeds
// still there keep their device state, the others are erased on the device too.
//
// Host policy (duck-typed; see the two users):
//   Frame*  keyframeOf(const Seed&)                       it->ftr->frame
//   void    feature(const Seed&, double px[2], double f[3], int* level)
//   void    pose7(const Frame&, double T[7])              {t, q(xyzw)} of T_f_w_
//   bool    keyframeSlots(const std::vector<Frame*>&, std::vector<int>&)   device pyramid slots of ALL the frame's keyframes,
//                                                         resolved together (slot_table.h); false: unavailable
//   int     currentSlot(Frame&)                           device pyramid slot of the current frame (negative: unavailable)
//   svo_hip_pyramid* keyframePyramids(), currentPyramids()
//   svo_hip_camera camera(const Frame&)
//   bool    isKeyframe(const Frame&)
//   void    setGridOccupancy(const double px_cur[2])      feature_detector_->setGridOccpuancy
//   void    converged(Seed&, const double xyz_world[3])   new Point, ftr->point, seed_converged_cb_(point, sigma2)
#ifndef SVO_DROPIN_DEPTH_FILTER_BATCH_H_
#define SVO_DROPIN_DEPTH_FILTER_BATCH_H_

#include <cstddef>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "svo_hip.h"

namespace svo {
namespace hip_bridge {

struct SeedBatchStats {
  int n_seeds = 0, n_aged_out = 0, n_updated = 0, n_failed_matches = 0, n_converged = 0, n_nan = 0;
  int n_device_calls = 0, n_device_errors = 0;
  int n_uploaded = 0;                  // seeds mirrored on the device by this call (0 on every frame but a keyframe's next)
  bool resynced = false;               // the list had changed behind the mirror's back
  bool halted = false;
};

/// The device mirror of a std::list<Seed>: owned by the depth filter object, used under its seeds mutex.
template <class SeedList>
class DeviceSeedMirror {
 public:
  typedef typename SeedList::iterator It;
  typedef typename SeedList::value_type SeedT;

  DeviceSeedMirror() {}
  ~DeviceSeedMirror() { clear(); }
  DeviceSeedMirror(const DeviceSeedMirror&) = delete;
  DeviceSeedMirror& operator=(const DeviceSeedMirror&) = delete;

  void clear() {
    for (size_t k = 0; k < batches_.size(); ++k) svo_hip_seed_batch_destroy(batches_[k].dev);
    batches_.clear();
    where_.clear();
    known_size_ = 0; known_back_id_ = -1;
  }
  size_t deviceBatches() const { return batches_.size(); }

  /// DepthFilter::updateSeeds(frame).  `sub_batch`: seeds per device batch (an upload unit; one frame's pass covers all).
  template <class Host, class Frame>
  SeedBatchStats update(Host& host, svo_hip_ctx* ctx, SeedList& seeds, Frame& frame, const svo_hip_df_params& prm, int batch_counter,
                        int max_n_kfs, const volatile bool& halt, int sub_batch = 4096) {
    SeedBatchStats st;
    if (sub_batch < 1) sub_batch = 1;
    if (halt) { st.halted = true; return st; }
    // ---- the list against the mirror: new seeds (initializeSeeds pushed a keyframe's batch) or foreign erasures
    const int back_id = seeds.empty() ? -1 : seeds.back().id;
    if (seeds.size() != known_size_ || back_id != known_back_id_) {
      if (!reconcile(host, ctx, seeds, sub_batch, st)) { ++st.n_device_errors; return st; }
    }
    // ---- age-out (:256-261): batches older than max_n_kfs keyframes leave the list and the device
    for (size_t k = 0; k < batches_.size();) {
      if (halt) { st.halted = true; remember(seeds); return st; }
      Batch& b = batches_[k];
      if ((batch_counter - b.batch_id) > max_n_kfs) {
        for (size_t i = 0; i < b.its.size(); ++i)
          if (b.alive[i]) { where_.erase(&*b.its[i]); seeds.erase(b.its[i]); ++st.n_aged_out; }
        svo_hip_seed_batch_destroy(b.dev);
        batches_.erase(batches_.begin() + (std::ptrdiff_t)k);
      } else {
        ++k;
      }
    }
    for (size_t k = 0; k < batches_.size(); ++k) st.n_seeds += batches_[k].n_alive;
    remember(seeds);
    if (batches_.empty()) return st;

    const svo_hip_camera cam = host.camera(frame);
    const int cur_slot = host.currentSlot(frame);
    if (cur_slot < 0) return st;
    double T_cur[7];
    host.pose7(frame, T_cur);
    const bool is_keyframe = host.isKeyframe(frame);

    // ---- device: the batches of the frame through ONE set of launches (svo_hip_seed_batch_update_group_async), awaited in
    // list order
    std::vector<size_t> enqueued;
    if (halt) { st.halted = true; return st; }
    {
      std::vector<svo_hip_seed_batch*> devs;
      std::vector<int> slots;
      std::vector<double> T_refs;
      // the keyframes of every batch with live seeds, resolved TOGETHER: a slot handed out for one keyframe of this pass must
      // not be recycled for another (with more keyframes alive than the cache had slots, a one-at-a-time look-up did that,
      // and the first keyframe's seeds were matched against the second one's image)
      typedef decltype(((SeedT*)0)->ftr->frame) FramePtrT;
      std::vector<FramePtrT> kfs;
      std::vector<size_t> which;
      for (size_t k = 0; k < batches_.size(); ++k)
        if (batches_[k].n_alive != 0) { kfs.push_back(batches_[k].kf); which.push_back(k); }
      std::vector<int> kf_slots;
      if (!kfs.empty() && !host.keyframeSlots(kfs, kf_slots)) { ++st.n_device_errors; return st; }
      for (size_t j = 0; j < which.size(); ++j) {
        Batch& b = batches_[which[j]];
        if (kf_slots[j] < 0) continue;
        double T_ref[7];
        host.pose7(*b.kf, T_ref);
        devs.push_back(b.dev); slots.push_back(kf_slots[j]); T_refs.insert(T_refs.end(), T_ref, T_ref + 7);
        enqueued.push_back(which[j]);
      }
      if (!devs.empty()) {
        st.n_device_calls += (int)devs.size();
        const int rc = svo_hip_seed_batch_update_group_async((int)devs.size(), devs.data(), host.keyframePyramids(), slots.data(),
                                                             host.currentPyramids(), cur_slot, &cam, T_refs.data(), T_cur, &prm,
                                                             is_keyframe ? 1 : 0);
        // A failure leaves up to 8 batches at a time either enqueued or not (svo_hip.h): the collect loop below takes
        // every batch that has a pass pending -- its seeds WERE updated -- and skips the others, so nothing stays
        // pending into the next frame.
        if (rc != SVO_HIP_OK) ++st.n_device_errors;
      }
    }
    // ---- the events, in list order: grid marks, callbacks, erasures (:302-337)
    for (size_t e = 0; e < enqueued.size(); ++e) {
      Batch& b = batches_[enqueued[e]];
      const svo_hip_seed_event* ev = NULL;
      int n_ev = 0;
      int32_t counts[7];
      if (st.n_device_errors && !svo_hip_seed_batch_pending(b.dev)) continue;      // (not reached by a failed group call)
      if (svo_hip_seed_batch_collect(b.dev, &ev, &n_ev, counts) != SVO_HIP_OK) { ++st.n_device_errors; continue; }
      st.n_failed_matches += counts[SVO_HIP_SEED_NO_MATCH + 1];
      st.n_updated += counts[SVO_HIP_SEED_UPDATED + 1] + counts[SVO_HIP_SEED_CONVERGED + 1] + counts[SVO_HIP_SEED_NAN + 1];
      for (int j = 0; j < n_ev; ++j) {
        const svo_hip_seed_event& x = ev[j];
        if (x.index < 0 || (size_t)x.index >= b.its.size() || !b.alive[(size_t)x.index]) continue;
        It it = b.its[(size_t)x.index];
        if (is_keyframe) host.setGridOccupancy(x.px_cur);           // :302-306
        if (x.status == SVO_HIP_SEED_CONVERGED) {                   // :310-331
          it->mu = x.mu; it->sigma2 = x.sigma2;
          host.converged(*it, x.xyz_world);
          forget(b, (size_t)x.index, seeds);
          ++st.n_converged;
        } else if (x.status == SVO_HIP_SEED_NAN) {                  // :333-337
          forget(b, (size_t)x.index, seeds);
          ++st.n_nan;
        }
      }
    }
    // batches without a live seed are of no further use
    for (size_t k = 0; k < batches_.size();) {
      if (batches_[k].n_alive == 0) { svo_hip_seed_batch_destroy(batches_[k].dev); batches_.erase(batches_.begin() + (std::ptrdiff_t)k); }
      else ++k;
    }
    remember(seeds);
    return st;
  }

  /// Copy the device state (a, b, mu, sigma2) of every live seed into its list entry: before the host reads the list
  /// (getSeeds() / getSeedsCopy()).  Returns false on a device error.
  bool syncToHost() {
    std::vector<float> a, b, mu, s2;
    for (size_t k = 0; k < batches_.size(); ++k) {
      Batch& bt = batches_[k];
      const size_t n = bt.its.size();
      a.resize(n); b.resize(n); mu.resize(n); s2.resize(n);
      if (svo_hip_seed_batch_download(bt.dev, a.data(), b.data(), mu.data(), s2.data(), NULL) != SVO_HIP_OK) return false;
      for (size_t i = 0; i < n; ++i)
        if (bt.alive[i]) { It it = bt.its[i]; it->a = a[i]; it->b = b[i]; it->mu = mu[i]; it->sigma2 = s2[i]; }
    }
    return true;
  }

 private:
  struct Batch {
    size_t serial;                       // never reused: what where_ refers to
    int batch_id;                        // Seed::batch_id of its seeds (age-out)
    decltype(((SeedT*)0)->ftr->frame) kf;   // the reference keyframe (Frame*)
    svo_hip_seed_batch* dev;
    std::vector<It> its;                 // device index -> list entry
    std::vector<uint8_t> alive;          // ... still in the list
    int n_alive;
  };
  struct Where { size_t batch_serial; int index; int id; };   // id: Seed::id (a recycled list node is another seed)

  std::vector<Batch> batches_;                           // list order (= creation order of the keyframes' seed batches)
  std::unordered_map<const SeedT*, Where> where_;        // list node -> (its batch, index): the re-build after foreign erasures
  size_t next_serial_ = 0;
  size_t known_size_ = 0;
  int known_back_id_ = -1;

  void remember(const SeedList& seeds) {
    known_size_ = seeds.size();
    known_back_id_ = seeds.empty() ? -1 : seeds.back().id;
  }
  void forget(Batch& b, size_t index, SeedList& seeds) {
    where_.erase(&*b.its[index]);
    seeds.erase(b.its[index]);
    b.alive[index] = 0;
    --b.n_alive;
  }

  /// Bring the mirror in line with the list: seeds the mirror does not know become new device batches (per keyframe batch
  /// id, chunks of at most sub_batch, list order); mirrored seeds that left the list are erased on the device.
  template <class Host>
  bool reconcile(Host& host, svo_hip_ctx* ctx, SeedList& seeds, int sub_batch, SeedBatchStats& st) {
    // present[k][i]: list entries found for batch k
    std::vector<std::vector<uint8_t> > present(batches_.size());
    for (size_t k = 0; k < batches_.size(); ++k) present[k].assign(batches_[k].its.size(), 0);
    std::unordered_map<size_t, size_t> pos_of_serial;
    for (size_t k = 0; k < batches_.size(); ++k) pos_of_serial[batches_[k].serial] = k;
    std::vector<It> fresh;                                         // list entries without a mirror, list order
    for (It it = seeds.begin(); it != seeds.end(); ++it) {
      typename std::unordered_map<const SeedT*, Where>::const_iterator w = where_.find(&*it);
      if (w == where_.end() || w->second.id != it->id) { fresh.push_back(it); continue; }
      present[pos_of_serial[w->second.batch_serial]][(size_t)w->second.index] = 1;
    }
    // foreign erasures
    std::vector<int32_t> gone;
    for (size_t k = 0; k < batches_.size(); ++k) {
      Batch& b = batches_[k];
      gone.clear();
      for (size_t i = 0; i < b.its.size(); ++i)
        if (b.alive[i] && !present[k][i]) { gone.push_back((int32_t)i); b.alive[i] = 0; --b.n_alive; }
      if (!gone.empty()) {
        st.resynced = true;
        if (svo_hip_seed_batch_erase(b.dev, (int)gone.size(), gone.data()) != SVO_HIP_OK) return false;
      }
    }
    if (st.resynced) {                                             // stale node addresses must not alias new seeds
      for (typename std::unordered_map<const SeedT*, Where>::iterator w = where_.begin(); w != where_.end();) {
        const Batch& b = batches_[pos_of_serial[w->second.batch_serial]];
        if (!b.alive[(size_t)w->second.index]) w = where_.erase(w); else ++w;
      }
    }
    for (size_t k = 0; k < batches_.size();) {
      if (batches_[k].n_alive == 0) { svo_hip_seed_batch_destroy(batches_[k].dev); batches_.erase(batches_.begin() + (std::ptrdiff_t)k); }
      else ++k;
    }
    // new seeds: runs of equal (batch id, keyframe), chunked
    std::vector<double> px, f;
    std::vector<int32_t> level;
    std::vector<float> a, b, mu, zr, s2;
    size_t first = 0;
    while (first < fresh.size()) {
      size_t last = first + 1;
      while (last < fresh.size() && last - first < (size_t)sub_batch && fresh[last]->batch_id == fresh[first]->batch_id &&
             host.keyframeOf(*fresh[last]) == host.keyframeOf(*fresh[first])) ++last;
      const size_t n = last - first;
      px.resize(2 * n); f.resize(3 * n); level.resize(n); a.resize(n); b.resize(n); mu.resize(n); zr.resize(n); s2.resize(n);
      for (size_t i = 0; i < n; ++i) {
        const It it = fresh[first + i];
        int lvl = 0;
        host.feature(*it, &px[2 * i], &f[3 * i], &lvl);
        level[i] = lvl;
        a[i] = it->a; b[i] = it->b; mu[i] = it->mu; zr[i] = it->z_range; s2[i] = it->sigma2;
      }
      Batch nb;
      nb.batch_id = fresh[first]->batch_id;
      nb.kf = host.keyframeOf(*fresh[first]);
      nb.dev = NULL;
      if (svo_hip_seed_batch_create(ctx, (int)n, px.data(), f.data(), level.data(), a.data(), b.data(), mu.data(), zr.data(), s2.data(),
                                    &nb.dev) != SVO_HIP_OK) return false;
      nb.its.assign(fresh.begin() + (std::ptrdiff_t)first, fresh.begin() + (std::ptrdiff_t)last);
      nb.alive.assign(n, 1);
      nb.n_alive = (int)n;
      nb.serial = next_serial_++;
      for (size_t i = 0; i < n; ++i) { Where w; w.batch_serial = nb.serial; w.index = (int)i; w.id = nb.its[i]->id; where_[&*nb.its[i]] = w; }
      // batches_ stays in list order: new seeds are pushed to the back of the list, so a new batch goes to the end
      batches_.push_back(nb);
      st.n_uploaded += (int)n;
      first = last;
    }
    return true;
  }
};

}  // namespace hip_bridge
}  // namespace svo

#endif  // SVO_DROPIN_DEPTH_FILTER_BATCH_H_
