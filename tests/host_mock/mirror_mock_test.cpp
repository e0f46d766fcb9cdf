// CPU-only test of hip_bridge::DeviceSeedMirror (include/svo_dropin/depth_filter_batch.h): the list <-> device bookkeeping of the
// drop-in DepthFilter against a MOCK of the svo_hip_seed_batch_* entry points (no GPU, no oracle: the mock "device" gives every
// seed a countdown -- it is "updated" every pass and "converges" when the countdown reaches zero, every 7th seed turns NaN on
// its third pass).  What is checked is the host logic: upload once, events applied in list order, age-out of whole batches,
// erasures behind the mirror's back (removeKeyframe / reset), recycled list nodes, the halt flag, syncToHost; more keyframes
// alive than the pyramid cache has slots (every batch of a pass must see ITS keyframe's image: slot_table.h); a grouped pass
// that fails half-way (what was enqueued is collected, nothing stays pending).
// Built and run by tests/test_host_mirror_mock.py with g++ -std=c++11.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <map>
#include <stdexcept>
#include <vector>

#include "svo_hip.h"
#include "svo_dropin/slot_table.h"
#include "svo_dropin/depth_filter_batch.h"

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); std::exit(1); } } while (0)

// ---------------------------------------------------------------- mock device
struct svo_hip_seed_batch {
  std::vector<float> a, b, mu, sigma2;
  std::vector<int> countdown, passes;
  std::vector<uint8_t> alive;
  std::vector<svo_hip_seed_event> events;
  int32_t counts[7];
  bool pending;
  int report_updated;
  int kf_id;                  // the test encodes the keyframe's id in px[1] of the batch's first seed
};
static int g_created = 0, g_destroyed = 0, g_uploaded_seeds = 0, g_updates = 0;
static int g_fail_group_after = -1;          // >= 0: the grouped pass enqueues this many batches, then reports a device failure
static int g_wrong_image = 0;                // passes in which a batch was handed a slot that holds another keyframe's pyramid
// the mock "pyramid batch": which frame's image every slot holds (what svo_hip_pyramid* points at in this test)
struct svo_hip_pyramid { std::vector<int> holds; };

extern "C" {
int svo_hip_seed_batch_create(svo_hip_ctx*, int n, const double* px, const double*, const int32_t*, const float* a, const float* b,
                              const float* mu, const float*, const float* sigma2, svo_hip_seed_batch** out) {
  svo_hip_seed_batch* s = new svo_hip_seed_batch;
  s->a.assign(a, a + n); s->b.assign(b, b + n); s->mu.assign(mu, mu + n); s->sigma2.assign(sigma2, sigma2 + n);
  s->countdown.resize((size_t)n); s->passes.assign((size_t)n, 0); s->alive.assign((size_t)n, 1);
  for (int i = 0; i < n; ++i) s->countdown[(size_t)i] = (int)px[2 * i];          // the test encodes the countdown in px[0]
  s->pending = false; s->report_updated = 0;
  s->kf_id = (int)px[1];
  ++g_created; g_uploaded_seeds += n;
  *out = s;
  return SVO_HIP_OK;
}
int svo_hip_seed_batch_destroy(svo_hip_seed_batch* s) { if (s) { ++g_destroyed; delete s; } return SVO_HIP_OK; }
int svo_hip_seed_batch_size(const svo_hip_seed_batch* s, int* n, int* n_alive) {
  if (n) *n = (int)s->alive.size();
  if (n_alive) *n_alive = (int)std::count(s->alive.begin(), s->alive.end(), (uint8_t)1);
  return SVO_HIP_OK;
}
int svo_hip_seed_batch_update_async(svo_hip_seed_batch* s, const svo_hip_pyramid*, int, const svo_hip_pyramid*, int, const svo_hip_camera*,
                                    const double*, const double*, const svo_hip_df_params*, int report_updated) {
  if (s->pending) return SVO_HIP_ERR_STATE;
  ++g_updates;
  s->events.clear();
  for (int k = 0; k < 7; ++k) s->counts[k] = 0;
  for (size_t i = 0; i < s->alive.size(); ++i) {
    if (!s->alive[i]) { ++s->counts[SVO_HIP_SEED_ERASED + 1]; continue; }
    ++s->passes[i]; --s->countdown[i];
    s->mu[i] += 1.0f; s->sigma2[i] *= 0.5f;
    int st = SVO_HIP_SEED_UPDATED;
    if (s->countdown[i] <= 0) st = SVO_HIP_SEED_CONVERGED;
    else if (i % 7 == 3 && s->passes[i] == 3) st = SVO_HIP_SEED_NAN;
    ++s->counts[st + 1];
    if (st != SVO_HIP_SEED_UPDATED || report_updated) {
      svo_hip_seed_event e;
      e.index = (int32_t)i; e.status = st; e.mu = s->mu[i]; e.sigma2 = s->sigma2[i];
      e.xyz_world[0] = (double)i; e.xyz_world[1] = s->mu[i]; e.xyz_world[2] = 0.0;
      e.px_cur[0] = (double)i; e.px_cur[1] = (double)s->passes[i];
      s->events.push_back(e);
    }
    if (st != SVO_HIP_SEED_UPDATED) s->alive[i] = 0;
  }
  s->pending = true;
  return SVO_HIP_OK;
}
int svo_hip_seed_batch_update_group_async(int n, svo_hip_seed_batch* const* sbs, const svo_hip_pyramid* ref, const int* slots,
                                          const svo_hip_pyramid* cur, int cur_slot, const svo_hip_camera* cam, const double* T_ref_w,
                                          const double* T_cur_w, const svo_hip_df_params* prm, int report_updated) {
  for (int k = 0; k < n; ++k) if (sbs[k]->pending) return SVO_HIP_ERR_STATE;           // checked before anything is enqueued
  for (int k = 0; k < n; ++k) {
    if (g_fail_group_after >= 0 && k == g_fail_group_after) return SVO_HIP_ERR_DEVICE;   // a later set of launches failed
    if (ref && ref->holds[(size_t)slots[k]] != sbs[k]->kf_id) ++g_wrong_image;
    svo_hip_seed_batch_update_async(sbs[k], ref, slots[k], cur, cur_slot, cam, T_ref_w + 7 * k, T_cur_w, prm, report_updated);
  }
  return SVO_HIP_OK;
}
int svo_hip_seed_batch_pending(const svo_hip_seed_batch* s) { return s && s->pending ? 1 : 0; }
int svo_hip_seed_batch_collect(svo_hip_seed_batch* s, const svo_hip_seed_event** ev, int* n_ev, int32_t counts[7]) {
  if (!s->pending) return SVO_HIP_ERR_STATE;
  s->pending = false;
  if (ev) *ev = s->events.empty() ? NULL : &s->events[0];
  if (n_ev) *n_ev = (int)s->events.size();
  if (counts) for (int k = 0; k < 7; ++k) counts[k] = s->counts[k];
  return SVO_HIP_OK;
}
int svo_hip_seed_batch_erase(svo_hip_seed_batch* s, int n, const int32_t* idx) {
  for (int k = 0; k < n; ++k) s->alive[(size_t)idx[k]] = 0;
  return SVO_HIP_OK;
}
int svo_hip_seed_batch_download(svo_hip_seed_batch* s, float* a, float* b, float* mu, float* sigma2, uint8_t* alive) {
  const size_t n = s->alive.size();
  if (a) std::copy(s->a.begin(), s->a.end(), a);
  if (b) std::copy(s->b.begin(), s->b.end(), b);
  if (mu) std::copy(s->mu.begin(), s->mu.end(), mu);
  if (sigma2) std::copy(s->sigma2.begin(), s->sigma2.end(), sigma2);
  if (alive) std::copy(s->alive.begin(), s->alive.begin() + (std::ptrdiff_t)n, alive);
  return SVO_HIP_OK;
}
}  // extern "C"

// ---------------------------------------------------------------- minimal host types (the shape of the reference's)
struct Frame { int id; bool keyframe; };
struct Feature { Frame* frame; double px[2]; };
struct Seed {
  static int batch_counter, seed_counter;
  int batch_id, id;
  Feature* ftr;
  float a, b, mu, z_range, sigma2;
  Seed(Feature* f) : batch_id(batch_counter), id(seed_counter++), ftr(f), a(10), b(10), mu(0.5f), z_range(1), sigma2(1) {}
};
int Seed::batch_counter = 0;
int Seed::seed_counter = 0;

struct Host {
  std::vector<std::pair<int, double> > converged_log;     // (seed id, sigma2) in callback order
  std::vector<double> grid_log;                           // px_cur[0] of the grid marks, in order
  Frame* keyframeOf(const Seed& s) const { return s.ftr->frame; }
  void feature(const Seed& s, double px[2], double f[3], int* level) const { px[0] = s.ftr->px[0]; px[1] = s.ftr->px[1]; f[0] = f[1] = 0; f[2] = 1; *level = 0; }
  void pose7(const Frame&, double T[7]) const { for (int k = 0; k < 7; ++k) T[k] = k == 6 ? 1.0 : 0.0; }
  // the keyframe pyramid cache of the drop-in in miniature: TWO slots to start with, the real slot bookkeeping
  svo::hip_bridge::SlotTable table;
  svo_hip_pyramid pyr;
  int n_uploads, n_grown;
  Host() : table(2), n_uploads(0), n_grown(0) { pyr.holds.assign(2, -1); }
  struct Up {
    Host* h; const std::vector<Frame*>* kfs;
    bool operator()(size_t k, int slot) const { h->pyr.holds[(size_t)slot] = (*kfs)[k]->id; ++h->n_uploads; return true; }
  };
  bool keyframeSlots(const std::vector<Frame*>& kfs, std::vector<int>& slots) {
    std::vector<int> ids;
    std::map<int, int> distinct;
    for (size_t k = 0; k < kfs.size(); ++k) { ids.push_back(kfs[k]->id); distinct[kfs[k]->id] = 1; }
    const int need = table.capacityFor((int)distinct.size());
    if (need != table.capacity()) { table.reset(need); pyr.holds.assign((size_t)need, -1); ++n_grown; }
    Up up = {this, &kfs};
    return table.acquire(ids, slots, up);
  }
  int currentSlot(Frame&) { return 0; }
  svo_hip_pyramid* keyframePyramids() { return &pyr; }
  svo_hip_pyramid* currentPyramids() const { return NULL; }
  svo_hip_camera camera(const Frame&) const { svo_hip_camera c = svo_hip_camera(); c.width = 640; c.height = 480; return c; }
  bool isKeyframe(const Frame& f) const { return f.keyframe; }
  void setGridOccupancy(const double px_cur[2]) { grid_log.push_back(px_cur[0]); }
  void converged(Seed& s, const double*) { converged_log.push_back(std::make_pair(s.id, (double)s.sigma2)); }
};

typedef std::list<Seed> SeedList;

static std::vector<Feature*> add_keyframe(SeedList& seeds, Frame* kf, int n, int countdown_base) {
  ++Seed::batch_counter;
  std::vector<Feature*> fts;
  for (int i = 0; i < n; ++i) {
    Feature* f = new Feature; f->frame = kf; f->px[0] = countdown_base + (i % 5); f->px[1] = kf->id;
    fts.push_back(f);
    seeds.push_back(Seed(f));
  }
  return fts;
}

int main() {
  svo_hip_df_params prm = svo_hip_df_params();
  volatile bool halt = false;
  Host host;
  SeedList seeds;
  svo::hip_bridge::DeviceSeedMirror<SeedList> mirror;
  Frame kfA = {0, true}, kfB = {1, true}, kfC = {2, true}, cur = {9, false}, cur_kf = {10, true};

  // ---- one keyframe, sub-batches of 40: 100 seeds -> 3 device batches, uploaded once
  add_keyframe(seeds, &kfA, 100, 2);                     // countdowns 2..6
  svo::hip_bridge::SeedBatchStats st = mirror.update(host, NULL, seeds, cur, prm, Seed::batch_counter, 3, halt, 40);
  CHECK(st.n_uploaded == 100 && g_created == 3 && mirror.deviceBatches() == 3 && st.n_device_calls == 3 && st.n_updated == 100);
  CHECK(st.n_converged == 0 && seeds.size() == 100);
  st = mirror.update(host, NULL, seeds, cur, prm, Seed::batch_counter, 3, halt, 40);
  CHECK(st.n_uploaded == 0 && g_created == 3 && !st.resynced);                    // nothing new: no upload, no list walk consequences
  CHECK(st.n_converged == 20 && seeds.size() == 80);                              // countdown 2: every fifth seed, in list order
  for (size_t k = 0; k < host.converged_log.size(); ++k) CHECK(host.converged_log[k].first == (int)(5 * k));
  CHECK(host.converged_log[0].second == 0.25);                                    // the callback sees the device's sigma2
  CHECK(host.grid_log.empty());                                                   // not a keyframe: no grid marks

  // ---- a keyframe frame: every updated seed marks the grid, in list order across the batches; NaN seeds leave on their 3rd pass
  const size_t before = seeds.size();
  st = mirror.update(host, NULL, seeds, cur_kf, prm, Seed::batch_counter, 3, halt, 40);
  CHECK(host.grid_log.size() == before && st.n_updated == (int)before);
  CHECK(std::is_sorted(host.grid_log.begin(), host.grid_log.begin() + 32));      // (indices inside the first batch ascend)
  int want_nan = 0;                                                               // the mock: in-batch index % 7 == 3, third pass, not converging
  for (int i = 0; i < 100; ++i) want_nan += (i % 5 >= 2) && ((i % 40) % 7 == 3);
  CHECK(st.n_converged == 20 && st.n_nan == want_nan && want_nan > 0);
  CHECK(seeds.size() == before - (size_t)st.n_converged - (size_t)st.n_nan);

  // ---- syncToHost: the list entries get the device state
  CHECK(seeds.front().mu == 0.5f);                                                // construction-time value until synced
  CHECK(mirror.syncToHost());
  CHECK(seeds.front().mu == 3.5f && seeds.front().sigma2 == 0.125f);

  // ---- a second keyframe's seeds: only they are uploaded
  const int created_before = g_created, uploaded_before = g_uploaded_seeds;
  std::vector<Feature*> ftsB = add_keyframe(seeds, &kfB, 30, 50);
  st = mirror.update(host, NULL, seeds, cur, prm, Seed::batch_counter, 3, halt, 40);
  CHECK(st.n_uploaded == 30 && g_created == created_before + 1 && g_uploaded_seeds == uploaded_before + 30);

  // ---- removeKeyframe behind the mirror's back: every seed of keyframe A leaves the list; then a third keyframe is added, whose
  // ---- list nodes may reuse the freed addresses
  for (SeedList::iterator it = seeds.begin(); it != seeds.end();) it = (it->ftr->frame == &kfA) ? seeds.erase(it) : ++it;
  CHECK(seeds.size() == 30);
  add_keyframe(seeds, &kfC, 25, 60);
  st = mirror.update(host, NULL, seeds, cur, prm, Seed::batch_counter, 3, halt, 40);
  CHECK(st.resynced && st.n_uploaded == 25 && st.n_seeds == 55 && st.n_updated == 55);
  CHECK(mirror.deviceBatches() == 2);                                             // A's three device batches are gone
  CHECK(mirror.syncToHost());
  for (SeedList::iterator it = seeds.begin(); it != seeds.end(); ++it)
    CHECK(it->mu == (it->ftr->frame == &kfB ? 2.5f : 1.5f));                      // B: two passes, C: one -- no state mixed up by recycled nodes

  // ---- the halt flag: raised before the call -> nothing happens
  halt = true;
  const int updates_before = g_updates;
  st = mirror.update(host, NULL, seeds, cur, prm, Seed::batch_counter, 3, halt, 40);
  CHECK(st.halted && g_updates == updates_before);
  halt = false;

  // ---- age-out: max_n_kfs keyframes later batch B is too old (batch_counter - batch_id > max_n_kfs): its seeds leave list and device
  Seed::batch_counter += 3;
  st = mirror.update(host, NULL, seeds, cur, prm, Seed::batch_counter, 3, halt, 40);
  CHECK(st.n_aged_out == 30 && seeds.size() == 25 && mirror.deviceBatches() == 1);
  for (SeedList::iterator it = seeds.begin(); it != seeds.end(); ++it) CHECK(it->ftr->frame == &kfC);

  // ---- more keyframes alive than the cache has slots (it started with two): every batch of the grouped pass is handed the
  // ---- slot that holds ITS keyframe, over several frames; the cache grows once and uploads every keyframe once
  {
    std::vector<Frame> kfs(20);
    for (int k = 0; k < 20; ++k) { kfs[(size_t)k].id = 100 + k; kfs[(size_t)k].keyframe = true; add_keyframe(seeds, &kfs[(size_t)k], 6, 40); }
    const int up0 = host.n_uploads, grown0 = host.n_grown;
    for (int frame = 0; frame < 3; ++frame) {
      st = mirror.update(host, NULL, seeds, cur, prm, Seed::batch_counter, 1000, halt, 40);
      CHECK(st.n_device_errors == 0 && st.n_updated == st.n_seeds && st.n_seeds > 120);   // every live seed of all 21 keyframes
    }
    CHECK(g_wrong_image == 0);
    CHECK(host.n_grown == grown0 + 1 && host.table.capacity() >= 21);
    CHECK(host.n_uploads - up0 == 21);                     // kfC again after the growth + the 20 new ones, each once
    // ---- a grouped pass that fails after 5 of its 21 batches: those five are collected (their seeds were updated), the others
    // ---- keep their state, nothing is left pending and the next frame runs normally
    g_fail_group_after = 5;
    const int updates0 = g_updates;
    st = mirror.update(host, NULL, seeds, cur, prm, Seed::batch_counter, 1000, halt, 40);
    CHECK(st.n_device_errors == 1 && g_updates == updates0 + 5 && st.n_updated > 0 && st.n_updated < st.n_seeds);
    g_fail_group_after = -1;
    st = mirror.update(host, NULL, seeds, cur, prm, Seed::batch_counter, 1000, halt, 40);
    CHECK(st.n_device_errors == 0 && st.n_updated == st.n_seeds && g_wrong_image == 0);
    for (SeedList::iterator it = seeds.begin(); it != seeds.end();) it = (it->ftr->frame != &kfC) ? seeds.erase(it) : ++it;
    st = mirror.update(host, NULL, seeds, cur, prm, Seed::batch_counter, 1000, halt, 40);
    CHECK(st.resynced && mirror.deviceBatches() == 1);
    for (SeedList::iterator it = seeds.begin(); it != seeds.end(); ++it) CHECK(it->ftr->frame == &kfC);
  }

  // ---- reset(): the list is cleared behind the mirror's back
  seeds.clear();
  st = mirror.update(host, NULL, seeds, cur, prm, Seed::batch_counter, 3, halt, 40);
  CHECK(mirror.deviceBatches() == 0 && st.n_seeds == 0);
  mirror.clear();
  CHECK(g_created == g_destroyed);                                                // every device batch was released
  std::printf("mirror mock test OK: %d device batches created, %d seeds uploaded, %d passes\n", g_created, g_uploaded_seeds, g_updates);
  return 0;
}
