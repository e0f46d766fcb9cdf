// slot_table.h -- which slot of a device pyramid batch holds which frame.  Standard library only: shared by the
// PyramidCache of the drop-in bridge (svo_hip_bridge.h, the reference's cv::Mat pyramids) and of the executable host
// layer (android_svo_amd/host/svo_host.h), and run against a mock in tests/host_mock/mirror_mock_test.cpp.
//
// The reference has no such cache (its frames keep their pyramids in host memory); it exists because a keyframe's
// pyramid is uploaded ONCE and then matched against by many frames (S/depth_filter.cpp:237-341: every seed of every
// keyframe alive, S/reprojector.cpp:72-168: every close keyframe).  A call that needs several frames at once -- the
// grouped depth-filter pass takes every keyframe that still owns seeds through one set of launches -- must resolve all
// of them BEFORE it enqueues anything: a slot handed out for one frame of the call may not be recycled for another
// frame of the same call.  `acquire` does that; a one-frame-at-a-time `slotOf` round robin cannot (with more keyframes
// alive than slots it overwrites a slot the same call already handed out, and that keyframe's seeds are then matched
// against another keyframe's image).
#ifndef SVO_DROPIN_SLOT_TABLE_H_
#define SVO_DROPIN_SLOT_TABLE_H_

#include <cstddef>
#include <vector>

namespace svo {
namespace hip_bridge {

class SlotTable {
 public:
  explicit SlotTable(int capacity = 0) { reset(capacity); }
  /// forget everything; `capacity` slots from now on (the caller has re-created the device batch with that many)
  void reset(int capacity) { ids_.assign((size_t)(capacity > 0 ? capacity : 0), -1); next_ = 0; }
  int capacity() const { return (int)ids_.size(); }
  int find(int frame_id) const {
    for (size_t s = 0; s < ids_.size(); ++s) if (ids_[s] == frame_id) return (int)s;
    return -1;
  }
  /// slots a call with `n_distinct` distinct frames needs: capacity() if they fit, else the next multiple of 16
  int capacityFor(int n_distinct) const { return n_distinct <= capacity() ? capacity() : (n_distinct + 15) / 16 * 16; }

  /// Resolve every frame of one call (ids may repeat).  Frames already resident keep their slot; each of the others is
  /// given a slot that no frame of this call uses and `upload(k, slot)` (k = index of its first occurrence in
  /// frame_ids) is called to fill it -- a false return aborts.  Needs capacity() >= number of distinct ids
  /// (capacityFor): returns false without touching anything otherwise.
  template <class Upload>
  bool acquire(const std::vector<int>& frame_ids, std::vector<int>& slots, Upload upload) {
    slots.assign(frame_ids.size(), -1);
    std::vector<char> used(ids_.size(), 0);
    int n_missing = 0, n_resident = 0;
    for (size_t k = 0; k < frame_ids.size(); ++k) {
      bool seen = false;
      for (size_t m = 0; m < k && !seen; ++m) seen = frame_ids[m] == frame_ids[k];
      if (seen) continue;
      const int s = find(frame_ids[k]);
      if (s >= 0) { used[(size_t)s] = 1; ++n_resident; } else ++n_missing;
    }
    if (n_resident + n_missing > capacity()) return false;
    for (size_t k = 0; k < frame_ids.size(); ++k) {
      int s = find(frame_ids[k]);
      if (s < 0) {
        while (used[(size_t)next_]) next_ = (next_ + 1) % capacity();      // terminates: a free slot exists (checked above)
        s = next_;
        next_ = (next_ + 1) % capacity();
        ids_[(size_t)s] = -1;
        if (!upload(k, s)) return false;
        ids_[(size_t)s] = frame_ids[k];
        used[(size_t)s] = 1;
      }
      slots[k] = s;
    }
    return true;
  }

 private:
  std::vector<int> ids_;
  int next_ = 0;
};

}  // namespace hip_bridge
}  // namespace svo

#endif  // SVO_DROPIN_SLOT_TABLE_H_
