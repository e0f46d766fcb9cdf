// svo_track.hip -- one tracked frame of FrameHandlerMono::processFrame (S/frame_handler_mono.cpp:171-229) as ONE chain of
// kernels on ONE stream with ONE synchronisation at the end:
//
//   new_frame->T_f_w_ = last_frame->T_f_w_                                   (:175)
//   SparseImgAlign(kltMaxLevel, kltMinLevel, 30, GaussNewton).run(last, new)  (:186-188)   svo_sia.hip, fused kernel
//   Reprojector::reprojectMap(new_frame, overlap_kfs)                         (:203)       trk_plan_kernel -> Matcher::findMatchDirect
//                                                                                          batch (svo_depth.hip) -> trk_replay_kernel
//   pose_optimizer::optimizeGaussNewton(...)                                  (:226-229)   svo_refine.hip
//   last_frame_ = new_frame_                                                  (frame_handler_mono.cpp:91)   trk_finish_kernel
//
// What stays on the device between the stages: the pose SparseImgAlign leaves (read by the reprojection and by the pose
// refinement straight from the solver's record), the candidates of every grid cell, the matches, and -- across frames --
// the new frame's pyramid and features, which are the next call's reference frame.  The map the reprojector walks
// (keyframe poses / pyramids / feature lists / key points, points with their observation lists, point candidates) is a
// set of index tables the host uploads when the map changes (keyframe rate); the counters the reprojector keeps on the
// points (n_failed_reproj_, n_succeeded_reproj_, type promotions) are advanced on the device and returned with every
// frame.  A point the reprojector deletes changes the pointer graph (Map::safeDeletePoint clears feature references and
// re-selects key points): the device marks it unlinked, the keyframes that lost a key feature choose again before the next
// frame (trk_rekey_kernel), and map_changed tells the host to apply the same deletions to its own objects -- no new upload.
//
// The two serial policies of the reference are kept by construction, not approximated:
//   * cell lists are formed in the reference's push_back order (closest keyframe first, each keyframe's fts_ order, a
//     point only once: last_projected_kf_id_, then the point candidates) and stably ordered by point type
//     (cell.sort(pointQualityComparator)) -- every item gets its sequence number and a rank inside its cell;
//   * all candidates are matched in one batch, then "first success per cell wins, stop once n_matches exceeds maxFts" is
//     replayed over the results (Matcher::findMatchDirect is a pure function of its candidate).
#include <climits>
#include <cstring>
#include <new>
#include <vector>

#include "svo_internal.h"
#include "svo_match_device.h"
#include "svo_point_refine.h"

using namespace svo_dev;

namespace {

constexpr int TRK_THREADS = 1024;
constexpr int TRK_MAX_SEL = 16;            // >= Reprojector::Options::max_n_kfs
constexpr int TRK_MAX_STRUCT = 64;         // points per structure-optimisation call (Config::structureOptimMaxPts() = 20)
constexpr int TRK_LDS_KF = 256;            // keyframes of the map (svo_hip_tracker_config::max_keyframes <= this): per-keyframe scratch in LDS
constexpr int TRK_LDS_ITEMS = 2048;        // a frame with at most this many candidates keeps their per-cell sort keys in LDS
constexpr int TRK_LDS_CELLS = 2048;        // a grid with at most this many cells keeps the cell counters in LDS
constexpr int TYPE_DELETED = 0, TYPE_CANDIDATE = 1, TYPE_UNKNOWN = 2, TYPE_GOOD = 3;   // Point::PointType (I/point.h:33-38)

// the map as index tables (device pointers)
struct TrkMap {
  int n_kf, n_points, n_candidates;
  const double* T_kf_w;          // [n_kf][7]
  const int* kf_slot;            // [n_kf] pyramid slot
  const int* kf_key_point;       // [n_kf][5]
  const int* kf_ftr_offset;      // [n_kf + 1]
  const int* kf_ftr_point;
  const double* pt_pos;          // [n_points][3]
  int* pt_type;
  int* pt_n_failed;
  int* pt_n_succeeded;
  uint8_t* pt_unlinked;
  const int* pt_obs_offset;      // [n_points + 1]
  const int* obs_kf;
  const double* obs_px;
  const double* obs_f;
  const int* obs_level;
  const uint8_t* obs_edgelet;
  const double* obs_grad;
  const int* cand_point;
};

// scratch and outputs of the planning kernel; cap = capacity of the candidate arrays
struct TrkPlan {
  int cap, n_cells, grid_cols, grid_size, max_n_kfs;
  int* first_seq;                // [n_points]
  int* item_point;               // [cap] kept items in arrival order
  double* item_px;               // [cap][2]
  int* item_cell;                // [cap]
  unsigned long long* item_key;  // [cap]
  int* seg;                      // [cap]
  unsigned long long* seg_key;   // [cap] item_key in segment order (frames beyond TRK_LDS_ITEMS candidates)
  int* cell_count;               // [n_cells + 1]
  int* cell_fill;                // [n_cells]
  // outputs (the candidates of every cell in trial order)
  int* counters;                 // [8]: 0 n_cand, 1 overflow, 2 map_changed, 3 n_overlap, 4 n_features, 5 n_pose_opt, 6.. n_trials (64 bit)
  int* cell_offset;              // [n_cells + 1]
  int* overlap_kf;               // [TRK_MAX_SEL]
  int* overlap_count;            // [TRK_MAX_SEL]
  int* cand_point;               // [cap]
  int* cand_obs;                 // [cap] (-1: Point::getCloseViewObs failed)
  int* cand_level_ref;           // [cap] pyramid level of the reference feature (read by the warp stage)
  uint8_t* cand_deleted;         // [cap]
};

// the new frame's features (what Reprojector::reprojectCell adds to frame->fts_, in creation order) + pose-refinement inputs
struct TrkFeat {
  int cap;
  double* px;        // [cap][2]
  double* f;         // [cap][3]
  double* pos;       // [cap][3]
  int* level;        // [cap]
  int* point;        // [cap]
  uint8_t* edgelet;  // [cap]
  double* grad;      // [cap][2]
  uint8_t* has_point;   // [cap] in/out of the pose refinement
};

// the last frame (reference of the next SparseImgAlign)
struct TrkLast {
  int* n;            // [1]
  double* T_f_w;     // [7]
  double* px;        // [cap][2]
  double* f;         // [cap][3]
  int* point;        // [cap]
  // slot 0 of the SparseImgAlign solver: the hand-over kernel writes the next solve's inputs there directly
  FrameConst* sia_fc;
  double *sia_px, *sia_f, *sia_pos;
  uint8_t* sia_has_point;
  int sia_max_n;
};

// Frame::isVisible (S/frame.cpp:162-172)
SVO_DEV bool frame_is_visible(const Cam& cam, const double* T_f_w, const double* xyz_w) {
  double xyz_f[3], px[2];
  se3_act(T_f_w, xyz_w, xyz_f);
  if (xyz_f[2] < 0.0) return false;
  world2cam(cam, xyz_f, px);
  return px[0] >= 0.0 && px[1] >= 0.0 && px[0] < cam.width && px[1] < cam.height;
}

// v.normalize() of Eigen 3.4: divide by the norm when the squared norm is positive
SVO_DEV void normalize3(double* v) {
  const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (n2 > 0.0) { const double nn = sqrt(n2); v[0] = v[0] / nn; v[1] = v[1] / nn; v[2] = v[2] / nn; }
}

// Point::getCloseViewObs (S/point.cpp:101-125): the observation whose viewing direction is closest to the frame's (first
// maximum of the cosine above zero, else the first observation); false when the angle exceeds 60 degrees
SVO_DEV bool close_view_obs(const TrkMap& m, int p, const double* framepos, const double (*kf_pos)[3], int* obs_out) {
  const double* pos = m.pt_pos + 3 * (size_t)p;
  double od[3] = {framepos[0] - pos[0], framepos[1] - pos[1], framepos[2] - pos[2]};
  normalize3(od);
  const int o0 = m.pt_obs_offset[p], o1 = m.pt_obs_offset[p + 1];
  int min_it = o0;
  double min_cos_angle = 0;
  for (int ob = o0; ob < o1; ob += 4) {                                      // the keyframe indices of four observations at once
    int kf[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) kf[e] = ob + e < o1 ? m.obs_kf[ob + e] : -1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (kf[e] < 0) continue;
      double d[3] = {kf_pos[kf[e]][0] - pos[0], kf_pos[kf[e]][1] - pos[1], kf_pos[kf[e]][2] - pos[2]};   // (*it)->frame->pos() - pos_
      normalize3(d);
      const double cos_angle = od[0] * d[0] + od[1] * d[1] + od[2] * d[2];
      if (cos_angle > min_cos_angle) { min_cos_angle = cos_angle; min_it = ob + e; }
    }
  }
  *obs_out = min_it;
  return !(min_cos_angle < 0.5);
}

// exclusive scan of v[0..n) in place by one workgroup, total -> v[n]; s_part: blockDim.x ints of LDS (64 used).
// Thread t owns a contiguous chunk; the thread sums are scanned inside each wave with lane shifts and across the waves
// through LDS: two barriers (a log-step scan over all 1024 threads took twenty).
SVO_DEV void block_exclusive_scan(int* v, int n, int* s_part) {
  const int t = threadIdx.x, nt = blockDim.x;
  const int lane = t & 63, wave = t >> 6, n_waves = nt >> 6;
  const int chunk = (n + nt - 1) / nt;
  const int lo = min(n, t * chunk), hi = min(n, lo + chunk);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += v[i];
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += s_part[w];
  int run = before + incl - sum;                          // exclusive prefix of this thread's chunk
  for (int i = lo; i < hi; ++i) { const int c = v[i]; v[i] = run; run += c; }
  if (t == nt - 1) v[n] = before + incl;
  __syncthreads();
  (void)n_waves;
}

// ---- Reprojector::reprojectMap up to the cell loop (S/reprojector.cpp:72-146) + the per-candidate choice of the reference
// feature (Point::getCloseViewObs, the first statement of Matcher::findMatchDirect).  One workgroup.
SVO_DEV void trk_plan_body(const TrkMap& m, const TrkPlan& pl, MdFrame mf, const double* __restrict__ T_slot_w,
                           SeedRec* __restrict__ recs, const FrameState* __restrict__ sia_state) {
  const Cam cam = mf.cam;
  __shared__ int s_part[TRK_THREADS];
  __shared__ int s_sel[TRK_MAX_SEL], s_seq_base[TRK_MAX_SEL + 1], s_ftr_off[TRK_MAX_SEL], s_ftr_cnt[TRK_MAX_SEL];
  __shared__ int s_n_close, s_n_sel, s_n_kept, s_overflow, s_changed;
  __shared__ int s_overlap[TRK_MAX_SEL];
  __shared__ double s_T[7], s_framepos[3];
  // per-keyframe scratch, and -- for frames and grids that fit (block-uniform choice) -- the cell counters and the sort keys
  // in segment order: what one phase writes and the next one reads costs an LDS access instead of a trip to L2
  __shared__ int s_kf_close[TRK_LDS_KF];
  __shared__ double s_kf_dist[TRK_LDS_KF], s_kf_pos[TRK_LDS_KF][3];
  __shared__ int s_cell_count[TRK_LDS_CELLS + 1], s_cell_fill[TRK_LDS_CELLS];
  __shared__ unsigned long long s_seg_key[TRK_LDS_ITEMS];
  const int t = threadIdx.x, nt = blockDim.x;
  int* const cell_count = pl.n_cells <= TRK_LDS_CELLS ? s_cell_count : pl.cell_count;
  int* const cell_fill = pl.n_cells <= TRK_LDS_CELLS ? s_cell_fill : pl.cell_fill;
  if (t < 7) s_T[t] = sia_state->T_cur_w[t];
  if (t == 0) { s_n_close = 0; s_n_kept = 0; s_overflow = 0; s_changed = 0; }
  if (t < TRK_MAX_SEL) { s_sel[t] = -1; s_overlap[t] = 0; }
  __syncthreads();
  double T[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { T[k] = s_T[k]; mf.T_cur_w[k] = s_T[k]; }
  if (t == 0) {
    double Tinv[7];
    se3_inverse(T, Tinv);                                                  // cur_frame.pos()
    s_framepos[0] = Tinv[0]; s_framepos[1] = Tinv[1]; s_framepos[2] = Tinv[2];
  }
  // ---- Map::getCloseKeyframes (S/map.cpp:109-131): keyframes one of whose key points is visible in the frame
  for (int k = t; k < m.n_kf; k += nt) {
    int kp[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) kp[j] = m.kf_key_point[5 * k + j];
    bool close = false;
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (!close && kp[j] >= 0) close = frame_is_visible(cam, T, m.pt_pos + 3 * (size_t)kp[j]);
    const double* tk = m.T_kf_w + 7 * (size_t)k;
    double Tk[7], Tinv[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) Tk[i] = tk[i];
    const double dx = T[0] - Tk[0], dy = T[1] - Tk[1], dz = T[2] - Tk[2];
    s_kf_close[k] = close ? 1 : 0;
    s_kf_dist[k] = sqrt(dx * dx + dy * dy + dz * dz);                      // (translation_vec difference).norm()
    se3_inverse(Tk, Tinv);                                                 // Frame::pos() of the keyframe (Point::getCloseViewObs)
    s_kf_pos[k][0] = Tinv[0]; s_kf_pos[k][1] = Tinv[1]; s_kf_pos[k][2] = Tinv[2];
    if (close) atomicAdd(&s_n_close, 1);
  }
  for (int p = t; p < m.n_points; p += nt) pl.first_seq[p] = INT_MAX;
  for (int c = t; c <= pl.n_cells; c += nt) cell_count[c] = 0;
  for (int c = t; c < pl.n_cells; c += nt) cell_fill[c] = 0;
  __syncthreads();
  // ---- close_kfs.sort by distance (stable: std::list::sort), the first max_n_kfs of them (:82-88)
  for (int k = t; k < m.n_kf; k += nt) {
    if (!s_kf_close[k]) continue;
    const double dk = s_kf_dist[k];
    int rank = 0;
    for (int j = 0; j < m.n_kf; ++j)
      if (s_kf_close[j] && (s_kf_dist[j] < dk || (!(dk < s_kf_dist[j]) && j < k))) ++rank;
    if (rank < pl.max_n_kfs) s_sel[rank] = k;
  }
  __syncthreads();
  if (t < TRK_MAX_SEL) {                                                    // the selected keyframes' feature ranges, all at once
    const int n_sel = s_n_close < pl.max_n_kfs ? s_n_close : pl.max_n_kfs;
    if (t < n_sel) {
      const int off = m.kf_ftr_offset[s_sel[t]];
      s_ftr_off[t] = off;
      s_ftr_cnt[t] = m.kf_ftr_offset[s_sel[t] + 1] - off;
    }
  }
  __syncthreads();
  if (t == 0) {
    const int n_sel = s_n_close < pl.max_n_kfs ? s_n_close : pl.max_n_kfs;
    s_n_sel = n_sel;
    int base = 0;
    for (int r = 0; r < n_sel; ++r) { s_seq_base[r] = base; base += s_ftr_cnt[r]; }
    s_seq_base[n_sel] = base;
  }
  __syncthreads();
  const int n_sel = s_n_sel;
  const int m_kf = s_seq_base[n_sel];
  const int m_all = m_kf + m.n_candidates;
  // ---- a point is projected once: the first keyframe feature that refers to it (last_projected_kf_id_, :103-106)
  for (int q = t; q < m_kf; q += nt) {
    int r = 0;
    while (r + 1 < n_sel && q >= s_seq_base[r + 1]) ++r;
    const int p = m.kf_ftr_point[s_ftr_off[r] + (q - s_seq_base[r])];
    if (p >= 0 && !m.pt_unlinked[p]) atomicMin(&pl.first_seq[p], q);
  }
  __syncthreads();
  // ---- reprojectPoint (:246-259) for the keyframe points and the point candidates (:118-138)
  for (int q = t; q < m_all; q += nt) {
    int p, r = -1;
    if (q < m_kf) {
      r = 0;
      while (r + 1 < n_sel && q >= s_seq_base[r + 1]) ++r;
      p = m.kf_ftr_point[s_ftr_off[r] + (q - s_seq_base[r])];
      if (p < 0) continue;
      const int fs = pl.first_seq[p];                                        // (asked for together with the flag)
      if (m.pt_unlinked[p] || fs != q) continue;
    } else {
      p = m.cand_point[q - m_kf];
      if (p < 0 || m.pt_unlinked[p]) continue;
    }
    const int ptype = m.pt_type[p];                                          // (in flight during the projection)
    double xyz_f[3], px[2];
    se3_act(T, m.pt_pos + 3 * (size_t)p, xyz_f);
    world2cam(cam, xyz_f, px);                                             // frame->w2c(point->pos_)
    const int ix = (int)px[0], iy = (int)px[1];                            // px.cast<int>()
    const bool in = px[0] == px[0] && px[1] == px[1] && ix >= 8 && ix < cam.width - 8 && iy >= 8 && iy < cam.height - 8;   // isInFrame(., 8)
    if (in) {
      const int cell = (int)(px[1] / pl.grid_size) * pl.grid_cols + (int)(px[0] / pl.grid_size);
      const int idx = atomicAdd(&s_n_kept, 1);
      if (idx < pl.cap) {
        pl.item_point[idx] = p;
        pl.item_px[2 * idx] = px[0]; pl.item_px[2 * idx + 1] = px[1];
        pl.item_cell[idx] = cell;
        pl.item_key[idx] = ((unsigned long long)(3 - ptype) << 32) | (unsigned)q;   // cell.sort: higher type first, stable
        atomicAdd(&cell_count[cell], 1);
      } else {
        s_overflow = 1;
      }
      if (r >= 0) atomicAdd(&s_overlap[r], 1);                             // overlap_kfs.back().second++ (:112-113)
    } else if (q >= m_kf) {
      // a candidate that is not in the frame (:126-135); the point sits in no cell, nobody else touches it
      const int nf = m.pt_n_failed[p] + 3;
      m.pt_n_failed[p] = nf;
      if (nf > 30) { m.pt_type[p] = TYPE_DELETED; m.pt_unlinked[p] = 1; s_changed = 1; }    // deleteCandidate + erase
    }
  }
  __syncthreads();
  const int n_kept = s_n_kept < pl.cap ? s_n_kept : pl.cap;
  unsigned long long* const seg_key = n_kept <= TRK_LDS_ITEMS ? s_seg_key : pl.seg_key;
  // ---- cell segments
  block_exclusive_scan(cell_count, pl.n_cells, s_part);                    // cell_count[c] = start of cell c, [n_cells] = total
  for (int c = t; c <= pl.n_cells; c += nt) pl.cell_offset[c] = cell_count[c];
  for (int i = t; i < n_kept; i += nt) {
    const int c = pl.item_cell[i];
    const unsigned long long key = pl.item_key[i];
    const int at = cell_count[c] + atomicAdd(&cell_fill[c], 1);
    pl.seg[at] = i;
    seg_key[at] = key;
  }
  __syncthreads();
  // ---- trial order inside the cell, and the reference feature of every candidate
  const double framepos[3] = {s_framepos[0], s_framepos[1], s_framepos[2]};
  for (int s = t; s < n_kept; s += nt) {
    const int i = pl.seg[s];
    const unsigned long long key = seg_key[s];
    const int c = pl.item_cell[i];
    const int p = pl.item_point[i];
    const double pxc[2] = {pl.item_px[2 * i], pl.item_px[2 * i + 1]};
    const int c0 = cell_count[c], c1 = cell_count[c + 1];
    int rank = 0;
    for (int j = c0; j < c1; ++j) rank += seg_key[j] < key ? 1 : 0;
    const int o = c0 + rank;
    const bool deleted = m.pt_type[p] == TYPE_DELETED;
    int obs = -1;
    bool view_ok = false;
    if (!deleted && m.pt_obs_offset[p + 1] > m.pt_obs_offset[p]) view_ok = close_view_obs(m, p, framepos, s_kf_pos, &obs);
    pl.cand_point[o] = p;
    pl.cand_obs[o] = view_ok ? obs : -1;
    pl.cand_deleted[o] = deleted ? 1 : 0;
    // the matcher's record of the candidate (Matcher::findMatchDirect up to the warp, svo_match_device.h): formed here,
    // read by the warp / alignment stages and by the replay
    if (view_ok) {
      pl.cand_level_ref[o] = m.obs_level[obs];
      recs[o] = md_geometry_item(mf, T_slot_w, m.kf_slot[m.obs_kf[obs]], m.obs_level[obs], m.obs_px + 2 * (size_t)obs, m.obs_f + 3 * (size_t)obs,
                                 m.pt_pos + 3 * (size_t)p, m.obs_edgelet[obs] != 0, m.obs_grad + 2 * (size_t)obs, pxc);
    } else {
      // a deleted point is erased from its cell (:190-194), a point without a close view fails the match at once (matcher.cpp:161-162)
      pl.cand_level_ref[o] = 0;
      SeedRec rc = md_dead_record();
      rc.uv0[0] = pxc[0]; rc.uv0[1] = pxc[1];
      recs[o] = rc;
    }
  }
  if (t == 0) {
    pl.counters[0] = n_kept;
    pl.counters[1] = s_overflow;
    pl.counters[2] = s_changed;
    pl.counters[3] = n_sel;
  }
  if (t < TRK_MAX_SEL) { pl.overlap_kf[t] = t < n_sel ? s_sel[t] : -1; pl.overlap_count[t] = t < n_sel ? s_overlap[t] : 0; }
}

// ---- the cell loop of Reprojector::reprojectMap (S/reprojector.cpp:149-166) with reprojectCell (:180-241) replayed over
// the batch results: per cell the first successful candidate wins, the loop stops after the cell that takes n_matches
// beyond max_fts; point bookkeeping (:202-215), the frame's new features (:217-231) and the inputs of the pose refinement.
SVO_DEV void trk_replay_body(const TrkMap& m, const TrkPlan& pl, const TrkFeat& ft, const Cam& cam, const FrameState* __restrict__ sia_state,
                             const SeedRec* __restrict__ recs, int* __restrict__ cell_winner,
                             int* __restrict__ cell_cum, int max_fts, int quality_min_fts) {
  __shared__ int s_part[TRK_THREADS];
  __shared__ int s_cut, s_changed;
  __shared__ unsigned long long s_trials;
  const int t = threadIdx.x, nt = blockDim.x;
  const int n_cells = pl.n_cells;
  if (t == 0) { s_cut = n_cells - 1; s_changed = pl.counters[2]; s_trials = 0ull; }
  // first success per cell
  for (int c = t; c < n_cells; c += nt) {
    int w = -1;
    for (int i = pl.cell_offset[c]; i < pl.cell_offset[c + 1] && w < 0; ++i)
      if (!pl.cand_deleted[i] && recs[i].path >= 0 && recs[i].matched) w = i;      // findMatchDirect returned true
    cell_winner[c] = w;
    cell_cum[c] = w >= 0 ? 1 : 0;
  }
  __syncthreads();
  block_exclusive_scan(cell_cum, n_cells, s_part);                          // cell_cum[c] = matches before cell c
  for (int c = t; c < n_cells; c += nt)
    if (cell_cum[c] + (cell_winner[c] >= 0 ? 1 : 0) > max_fts) atomicMin(&s_cut, c);     // n_matches_ > maxFts after this cell (:164-165)
  __syncthreads();
  const int cut = s_cut;
  double T[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) T[k] = sia_state->T_cur_w[k];
  unsigned long long my_trials = 0ull;
  for (int c = t; c <= cut && c < n_cells; c += nt) {
    const int w = cell_winner[c];
    const int end = w >= 0 ? w + 1 : pl.cell_offset[c + 1];
    for (int i = pl.cell_offset[c]; i < end; ++i) {
      ++my_trials;                                                          // ++n_trials_ (:188)
      if (pl.cand_deleted[i]) continue;                                     // TYPE_DELETED: erased from the cell (:190-194)
      const int p = pl.cand_point[i];
      if (i != w) {                                                         // :202-209
        const int nf = m.pt_n_failed[p] + 1;
        m.pt_n_failed[p] = nf;
        const int ty = m.pt_type[p];
        if ((ty == TYPE_UNKNOWN && nf > 15) || (ty == TYPE_CANDIDATE && nf > 30)) {   // safeDeletePoint / deleteCandidatePoint
          m.pt_type[p] = TYPE_DELETED; m.pt_unlinked[p] = 1; s_changed = 1;
        }
        continue;
      }
      const int ns = m.pt_n_succeeded[p] + 1;                               // :211-214
      m.pt_n_succeeded[p] = ns;
      if (m.pt_type[p] == TYPE_UNKNOWN && ns > 10) m.pt_type[p] = TYPE_GOOD;
      const int fi = cell_cum[c];                                           // creation order = cell order
      if (fi < ft.cap) {
        const SeedRec& rc = recs[i];
        const double u = rc.step[0], v = rc.step[1];                        // px_cur = px_scaled * (1 << search_level_) (matcher.cpp:200)
        const double* pp = m.pt_pos + 3 * (size_t)p;
        ft.px[2 * fi] = u; ft.px[2 * fi + 1] = v;
        double fv[3];
        cam2world(cam, u, v, fv);                                           // Feature(frame, px, level): f = cam2world(px) (I/feature.h:43-51)
        ft.f[3 * fi] = fv[0]; ft.f[3 * fi + 1] = fv[1]; ft.f[3 * fi + 2] = fv[2];
        ft.pos[3 * fi] = pp[0]; ft.pos[3 * fi + 1] = pp[1]; ft.pos[3 * fi + 2] = pp[2];
        ft.level[fi] = rc.search_level;
        ft.point[fi] = p;
        ft.has_point[fi] = 1;
        double g0 = 1.0, g1 = 0.0;
        const int obs = pl.cand_obs[i];
        const bool edge = m.obs_edgelet[obs] != 0;
        if (edge) {                                                         // grad = (A_cur_ref_ * ref_ftr_->grad).normalized() (:224-229)
          const double* Tr = m.T_kf_w + 7 * (size_t)m.obs_kf[obs];
          double T_ref_inv[7], T_cur_ref[7], A[4];
          se3_inverse(Tr, T_ref_inv);
          se3_mul(T, T_ref_inv, T_cur_ref);
          const double dx = T_ref_inv[0] - pp[0], dy = T_ref_inv[1] - pp[1], dz = T_ref_inv[2] - pp[2];
          get_warp_matrix_affine(cam, m.obs_px + 2 * (size_t)obs, m.obs_f + 3 * (size_t)obs, sqrt(dx * dx + dy * dy + dz * dz), T_cur_ref, m.obs_level[obs], A);
          const double gr0 = m.obs_grad[2 * (size_t)obs], gr1 = m.obs_grad[2 * (size_t)obs + 1];
          g0 = A[0] * gr0 + A[1] * gr1;
          g1 = A[2] * gr0 + A[3] * gr1;
          const double n2 = g0 * g0 + g1 * g1;
          if (n2 > 0.0) { const double nn = sqrt(n2); g0 = g0 / nn; g1 = g1 / nn; }
        }
        ft.edgelet[fi] = edge ? 1 : 0;
        ft.grad[2 * fi] = g0; ft.grad[2 * fi + 1] = g1;
      }
    }
  }
  if (my_trials) atomicAdd(&s_trials, my_trials);
  __syncthreads();
  if (t == 0) {
    const int last = cut < n_cells ? cut : n_cells - 1;
    int n_matches = n_cells > 0 ? cell_cum[last] + (cell_winner[last] >= 0 ? 1 : 0) : 0;
    const int n_feat = n_matches < ft.cap ? n_matches : ft.cap;
    pl.counters[2] = s_changed;
    pl.counters[4] = n_feat;
    // processFrame returns before the pose refinement when the reprojector matched too few points (:208-215)
    pl.counters[5] = n_matches < quality_min_fts ? 0 : n_feat;
    pl.counters[6] = (int)(s_trials & 0xffffffffull);
    pl.counters[7] = n_matches;
  }
}

// ---- Frame::removeKeyPoint / setKeyPoints (S/frame.cpp:83-165) for the keyframes that lost a key feature to a point the
// reprojector deleted in the previous frame (Map::safeDeletePoint, S/map.cpp:78-88): one workgroup per keyframe.
// A keyframe none of whose key features lost its point keeps its key features untouched, as in the reference (they are
// incumbents of the frame's history, not necessarily what a fresh selection would give).  In a keyframe that lost one,
// every slot is contested again by every feature that still has a point, in fts_ order with strict improvement: the
// incumbent stays on a tie, otherwise the first feature that reaches the best value wins.  A feature's pixel is the
// observation of its point in this keyframe.
__global__ __launch_bounds__(256) void trk_rekey_kernel(TrkMap m, int* __restrict__ kf_key_point, Cam cam) {
  const int k = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  __shared__ int s_found, s_inc[5], s_best_idx[5][256];
  __shared__ double s_inc_val[5], s_best_val[5][256];
  const int cu = cam.width / 2, cv = cam.height / 2;
  // pixel of point p's observation in keyframe k (false: none)
  auto px_in_kf = [&](int p, double* x, double* y) {
    for (int o = m.pt_obs_offset[p]; o < m.pt_obs_offset[p + 1]; ++o)
      if (m.obs_kf[o] == k) { *x = m.obs_px[2 * (size_t)o]; *y = m.obs_px[2 * (size_t)o + 1]; return true; }
    return false;
  };
  // value of a feature for slot j (0: distance from the centre, smaller is better; 1..4: the quadrant products, larger is
  // better, -inf outside the quadrant -- the two left quadrants test x against cv as the reference does, S/frame.cpp:133,142)
  auto slot_value = [&](int j, double x, double y) {
    if (j == 0) return fmax(fabs(x - cu), fabs(y - cv));
    const bool in = j == 1 ? (x >= cu && y >= cv) : j == 2 ? (x >= cu && y < cv) : j == 3 ? (x < cv && y < cv) : (x < cv && y >= cv);
    return in ? (x - cu) * (y - cv) : -HUGE_VAL;
  };
  if (t == 0) s_found = 0;
  __syncthreads();
  if (t < 5) {
    int p = kf_key_point[5 * k + t];
    if (p >= 0 && m.pt_unlinked[p]) { p = -1; atomicOr(&s_found, 1); }       // the key feature's point is gone (:85-88, :157-162)
    double x = 0, y = 0;
    const bool has = p >= 0 && px_in_kf(p, &x, &y);
    s_inc[t] = has ? p : -1;
    s_inc_val[t] = has ? slot_value(t, x, y) : 0.0;
  }
  __syncthreads();
  if (!s_found) return;                                                        // block-uniform
  double best[5];
  int best_j[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) { best[j] = j == 0 ? HUGE_VAL : -HUGE_VAL; best_j[j] = INT_MAX; }
  const int f0 = m.kf_ftr_offset[k], f1 = m.kf_ftr_offset[k + 1];
  for (int i = f0 + t; i < f1; i += nt) {                                      // ascending per thread: the first best index is kept
    const int p = m.kf_ftr_point[i];
    if (p < 0 || m.pt_unlinked[p]) continue;                                   // ftr->point == NULL
    double x, y;
    if (!px_in_kf(p, &x, &y)) continue;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const double v = slot_value(j, x, y);
      const bool better = j == 0 ? v < best[j] : v > best[j];
      if (better) { best[j] = v; best_j[j] = i; }
    }
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) { s_best_val[j][t] = best[j]; s_best_idx[j][t] = best_j[j]; }
  __syncthreads();
  if (t < 5) {
    double bv = t == 0 ? HUGE_VAL : -HUGE_VAL;
    int bi = INT_MAX;
    for (int q = 0; q < nt; ++q) {                                             // best value, lowest feature index on ties
      const double v = s_best_val[t][q];
      const int i = s_best_idx[t][q];
      if (i == INT_MAX) continue;
      const bool better = t == 0 ? v < bv : v > bv;
      if (better || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    int winner = s_inc[t];
    const bool challenger = bi != INT_MAX && (t == 0 || bv > -HUGE_VAL);       // (a feature outside the quadrant is no contender)
    if (challenger && (winner < 0 || (t == 0 ? bv < s_inc_val[t] : bv > s_inc_val[t]))) winner = m.kf_ftr_point[bi];
    kf_key_point[5 * k + t] = winner;
  }
}

// ---- FrameHandlerBase::optimizeStructure (S/frame_handler_base.cpp:190-210) on the points the host selected: Point::optimize
// of each of them over its observations as the map tables hold them (keyframe pose + bearing, Point::obs_ order), the new
// positions written into the point table, into the solver's copy of the last frame's points (the next SparseImgAlign::run reads
// point->pos_) and into the page-locked result block.  One workgroup.
struct TrkStructSel { int n; int point[TRK_MAX_STRUCT]; };
__global__ __launch_bounds__(256) void trk_structure_kernel(TrkMap m, double* __restrict__ pt_pos, TrkLast last, TrkStructSel sel, int n_iter,
                                                            double* __restrict__ out_pos, int* __restrict__ out_iters,
                                                            unsigned long long* __restrict__ done_flag, unsigned long long seq) {
  __shared__ double s_new[TRK_MAX_STRUCT][3];
  const int t = threadIdx.x, nt = blockDim.x;
  if (t < sel.n) {
    const int p = sel.point[t];
    double P[3] = {pt_pos[3 * (size_t)p], pt_pos[3 * (size_t)p + 1], pt_pos[3 * (size_t)p + 2]};
    const int done = point_refine_one(P, m.pt_obs_offset[p], m.pt_obs_offset[p + 1], n_iter, [&](int o, double* T, double* fo) {
      const double* tk = m.T_kf_w + 7 * (size_t)m.obs_kf[o];
      for (int i = 0; i < 7; ++i) T[i] = tk[i];
      fo[0] = m.obs_f[3 * (size_t)o]; fo[1] = m.obs_f[3 * (size_t)o + 1]; fo[2] = m.obs_f[3 * (size_t)o + 2];
    });
    for (int i = 0; i < 3; ++i) { pt_pos[3 * (size_t)p + i] = P[i]; s_new[t][i] = P[i]; out_pos[3 * t + i] = P[i]; }
    out_iters[t] = done;
  }
  __syncthreads();
  const int n_last = *last.n < last.sia_max_n ? *last.n : last.sia_max_n;
  for (int j = t; j < n_last; j += nt) {
    const int p = last.point[j];
    if (p < 0) continue;
    for (int i = 0; i < sel.n; ++i)
      if (sel.point[i] == p) { last.sia_pos[3 * j] = s_new[i][0]; last.sia_pos[3 * j + 1] = s_new[i][1]; last.sia_pos[3 * j + 2] = s_new[i][2]; }
  }
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(done_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- last_frame_ = new_frame_ (frame_handler_mono.cpp:91) and the result block.  One workgroup.
SVO_DEV void trk_finish_body(const TrkMap& m, const TrkPlan& pl, const TrkFeat& ft, const TrkLast& last, const Cam& cam,
                             const FrameState* __restrict__ sia_state,
                             const svo_hip_pose_opt_result* __restrict__ po, svo_hip_track_result* __restrict__ res,
                             double* __restrict__ out_px, double* __restrict__ out_f, int* __restrict__ out_level,
                             int* __restrict__ out_point, uint8_t* __restrict__ out_edgelet, double* __restrict__ out_grad,
                             int* __restrict__ out_pt_type, int* __restrict__ out_pt_failed, int* __restrict__ out_pt_succeeded,
                             unsigned long long* __restrict__ done_flag, unsigned long long seq) {
  __shared__ double s_Tnew[7];
  const int t = threadIdx.x, nt = blockDim.x;
  const int n_feat = pl.counters[4];
  const int n_po = pl.counters[5];
  const bool refined = n_po > 0 && po->ran != 0;
  if (t == 0) {
    svo_hip_track_result r;
    for (int i = 0; i < 7; ++i) r.T_f_w_sia[i] = sia_state->T_cur_w[i];
    r.sia_n_tracked = sia_state->n_meas / 16;
    for (int i = 0; i < SVO_HIP_MAX_LEVELS; ++i) r.sia_iters[i] = sia_state->iters[i];
    r.sia_stop = sia_state->stop;
    r.n_features = n_feat;
    r.n_matches = (uint64_t)pl.counters[7];
    r.n_trials = (uint64_t)(unsigned)pl.counters[6];
    r.n_overlap = pl.counters[3];
    r.map_changed = pl.counters[2];
    for (int i = 0; i < 16; ++i) { r.overlap_kf[i] = i < TRK_MAX_SEL ? pl.overlap_kf[i] : -1; r.overlap_count[i] = i < TRK_MAX_SEL ? pl.overlap_count[i] : 0; }
    r.n_candidates = pl.counters[0];
    r.items_overflow = pl.counters[1];
    r.pose = *po;
    if (!refined) {                        // the reference did not reach / did not run the refinement: nothing of it is valid
      memset(&r.pose, 0, sizeof(r.pose));
      for (int i = 0; i < 7; ++i) r.pose.T_f_w[i] = sia_state->T_cur_w[i];
    }
    // new_frame_->T_f_w_ when processFrame hands the frame over: the refined pose, or -- too few matches (:211) -- the last frame's
    for (int i = 0; i < 7; ++i) { r.T_f_w[i] = n_po > 0 ? r.pose.T_f_w[i] : last.T_f_w[i]; s_Tnew[i] = r.T_f_w[i]; }
    *res = r;                              // (res lies in host memory: written, never read back here)
  }
  __syncthreads();
  for (int i = t; i < n_feat; i += nt) {
    const int p = ft.has_point[i] ? ft.point[i] : -1;                       // pose_optimizer.cpp:154-157: (*it)->point = NULL
    out_px[2 * i] = ft.px[2 * i]; out_px[2 * i + 1] = ft.px[2 * i + 1];
    out_f[3 * i] = ft.f[3 * i]; out_f[3 * i + 1] = ft.f[3 * i + 1]; out_f[3 * i + 2] = ft.f[3 * i + 2];
    out_level[i] = ft.level[i];
    out_point[i] = p;
    out_edgelet[i] = ft.edgelet[i];
    out_grad[2 * i] = ft.grad[2 * i]; out_grad[2 * i + 1] = ft.grad[2 * i + 1];
    last.px[2 * i] = ft.px[2 * i]; last.px[2 * i + 1] = ft.px[2 * i + 1];
    last.f[3 * i] = ft.f[3 * i]; last.f[3 * i + 1] = ft.f[3 * i + 1]; last.f[3 * i + 2] = ft.f[3 * i + 2];
    last.point[i] = p;
    if (i < last.sia_max_n) {          // the next SparseImgAlign::run walks this frame's features (sparse_img_align.cpp:116-118)
      last.sia_px[2 * i] = ft.px[2 * i]; last.sia_px[2 * i + 1] = ft.px[2 * i + 1];
      last.sia_f[3 * i] = ft.f[3 * i]; last.sia_f[3 * i + 1] = ft.f[3 * i + 1]; last.sia_f[3 * i + 2] = ft.f[3 * i + 2];
      last.sia_pos[3 * i] = ft.pos[3 * i]; last.sia_pos[3 * i + 1] = ft.pos[3 * i + 1]; last.sia_pos[3 * i + 2] = ft.pos[3 * i + 2];
      last.sia_has_point[i] = p >= 0 ? 1 : 0;
    }
  }
  for (int p = t; p < m.n_points; p += nt) {
    out_pt_type[p] = m.pt_type[p]; out_pt_failed[p] = m.pt_n_failed[p]; out_pt_succeeded[p] = m.pt_n_succeeded[p];
  }
  __syncthreads();
  if (t == 0) {
    *last.n = n_feat;
    FrameConst c;
    c.cam = cam;
    for (int i = 0; i < 7; ++i) { const double v = s_Tnew[i]; last.T_f_w[i] = v; c.T_ref_w[i] = v; c.T_cur_w_init[i] = v; }   // :175
    double Tinv[7];
    se3_inverse(c.T_ref_w, Tinv);                                            // Frame::pos()
    c.ref_pos[0] = Tinv[0]; c.ref_pos[1] = Tinv[1]; c.ref_pos[2] = Tinv[2];
    c.n_feat = n_feat < last.sia_max_n ? n_feat : last.sia_max_n; c.pad = 0;
    last.sia_fc[0] = c;
  }
  // the result block lies in page-locked host memory: once every thread's stores have left for the host, the frame's
  // sequence number goes after them and the waiting host thread reads the block
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(done_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the kernels of one camera's chain, and the same bodies for the cameras of a tracker group: workgroup c of a launch takes
// camera c's arguments from a table in device memory (svo_hip_tracker_group_track rewrites it every call)
struct TrkCamArgs {
  TrkMap m;
  TrkPlan pl;
  TrkFeat ft;
  TrkLast last;
  MdFrame mf;
  const double* T_slot_w;
  SeedRec* recs;
  const FrameState* sia_state;
  int *cell_winner, *cell_cum;
  const svo_hip_pose_opt_result* po;
  char* res;                           // the camera's result block (device address of page-locked memory)
  size_t o_px, o_f, o_level, o_point, o_edge, o_grad, o_pt, o_flag;
  int n_points_cap;                    // (unused by the kernels; keeps the layout self-describing)
  unsigned long long seq;
};

__global__ __launch_bounds__(TRK_THREADS) void trk_plan_kernel(TrkMap m, TrkPlan pl, MdFrame mf, const double* __restrict__ T_slot_w,
                                                               SeedRec* __restrict__ recs, const FrameState* __restrict__ sia_state) {
  trk_plan_body(m, pl, mf, T_slot_w, recs, sia_state);
}
__global__ __launch_bounds__(TRK_THREADS) void trk_plan_cams_kernel(const TrkCamArgs* __restrict__ args) {
  const TrkCamArgs& a = args[blockIdx.x];
  trk_plan_body(a.m, a.pl, a.mf, a.T_slot_w, a.recs, a.sia_state);
}

__global__ __launch_bounds__(TRK_THREADS) void trk_replay_kernel(TrkMap m, TrkPlan pl, TrkFeat ft, Cam cam, const FrameState* __restrict__ sia_state,
                                                                 const SeedRec* __restrict__ recs, int* __restrict__ cell_winner,
                                                                 int* __restrict__ cell_cum, int max_fts, int quality_min_fts) {
  trk_replay_body(m, pl, ft, cam, sia_state, recs, cell_winner, cell_cum, max_fts, quality_min_fts);
}
__global__ __launch_bounds__(TRK_THREADS) void trk_replay_cams_kernel(const TrkCamArgs* __restrict__ args, int max_fts, int quality_min_fts) {
  const TrkCamArgs& a = args[blockIdx.x];
  trk_replay_body(a.m, a.pl, a.ft, a.mf.cam, a.sia_state, a.recs, a.cell_winner, a.cell_cum, max_fts, quality_min_fts);
}

SVO_DEV void trk_finish_from_args(const TrkCamArgs& a) {
  char* rd = a.res;
  int* pt = reinterpret_cast<int*>(rd + a.o_pt);
  trk_finish_body(a.m, a.pl, a.ft, a.last, a.mf.cam, a.sia_state, a.po, reinterpret_cast<svo_hip_track_result*>(rd),
                  reinterpret_cast<double*>(rd + a.o_px), reinterpret_cast<double*>(rd + a.o_f), reinterpret_cast<int*>(rd + a.o_level),
                  reinterpret_cast<int*>(rd + a.o_point), reinterpret_cast<uint8_t*>(rd + a.o_edge), reinterpret_cast<double*>(rd + a.o_grad),
                  pt, pt + a.m.n_points, pt + 2 * (size_t)a.m.n_points, reinterpret_cast<unsigned long long*>(rd + a.o_flag), a.seq);
}
__global__ __launch_bounds__(256) void trk_finish_kernel(TrkMap m, TrkPlan pl, TrkFeat ft, TrkLast last, Cam cam, const FrameState* __restrict__ sia_state,
                                                         const svo_hip_pose_opt_result* __restrict__ po, svo_hip_track_result* __restrict__ res,
                                                         double* __restrict__ out_px, double* __restrict__ out_f, int* __restrict__ out_level,
                                                         int* __restrict__ out_point, uint8_t* __restrict__ out_edgelet, double* __restrict__ out_grad,
                                                         int* __restrict__ out_pt_type, int* __restrict__ out_pt_failed, int* __restrict__ out_pt_succeeded,
                                                         unsigned long long* __restrict__ done_flag, unsigned long long seq) {
  trk_finish_body(m, pl, ft, last, cam, sia_state, po, res, out_px, out_f, out_level, out_point, out_edgelet, out_grad, out_pt_type, out_pt_failed,
                  out_pt_succeeded, done_flag, seq);
}
__global__ __launch_bounds__(256) void trk_finish_cams_kernel(const TrkCamArgs* __restrict__ args) { trk_finish_from_args(args[blockIdx.x]); }

// Point::pos_ of n points after the host optimised them: one staged block in, one launch
__global__ void trk_scatter_positions_kernel(int n, const int* __restrict__ idx, const double* __restrict__ pos, double* __restrict__ pt_pos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t p = (size_t)idx[i];
  pt_pos[3 * p] = pos[3 * i]; pt_pos[3 * p + 1] = pos[3 * i + 1]; pt_pos[3 * p + 2] = pos[3 * i + 2];
}

template <typename T>
int trk_alloc(svo_hip_ctx* ctx, T** p, size_t count) {
  void* d = nullptr;
  const int rc = svo_hip_malloc(ctx, &d, (count ? count : 1) * sizeof(T));
  *p = (T*)d;
  return rc;
}

}  // namespace

// What the cameras of a tracker group share (a lone tracker is a group of one): the pyramid batches, the SparseImgAlign solver
// (camera c = slot c), the arrays the batched stages index by camera with a fixed stride, and the page-locked image block.
struct svo_hip_tracker_shared {
  svo_hip_ctx* ctx = nullptr;
  int n_cams = 1;
  svo_hip_pyramid* kf_pyr = nullptr;        // n_cams x max_keyframes slots: camera c's keyframe slot s is slot c * max_keyframes + s
  svo_hip_pyramid* frame_pyr[2] = {nullptr, nullptr};   // n_cams slots each
  int last_idx = 0;                         // frame_pyr[last_idx] holds the cameras' last frames (they advance together)
  svo_hip_sia* sia = nullptr;               // batch n_cams
  uint8_t* img_host = nullptr;              // page-locked, mapped: n_cams images, img_stride apart
  uint8_t* img_dev = nullptr;
  size_t img_stride = 0;
  // [n_cams][...] arrays of the stages that run over all cameras at once
  int* counters = nullptr;                  // [n_cams][8]
  int* cand_level_ref = nullptr;            // [n_cams][cap]
  double *ft_f = nullptr, *ft_pos = nullptr;   // [n_cams][NF][3]
  int* ft_level = nullptr;                  // [n_cams][NF]
  uint8_t* ft_has_point = nullptr;          // [n_cams][NF]
  svo_hip_pose_opt_result* po = nullptr;    // [n_cams]
  // the group's argument table (TrkCamArgs per camera): page-locked staging + device copy
  char* args_host = nullptr;
  char* args_dev = nullptr;
  std::vector<svo_hip_tracker*> members;
  std::vector<void*> dev_allocs;
};

struct svo_hip_tracker {
  svo_hip_ctx* ctx = nullptr;
  svo_hip_camera cam{};
  svo_hip_tracker_config cfg{};
  int n_cells = 0, grid_cols = 0, grid_rows = 0;
  svo_hip_tracker_shared* sh = nullptr;     // owned by the tracker itself (a lone tracker) or by its group
  bool owns_shared = false;
  int cam_index = 0;                        // this camera's slot in the shared solver / frame pyramids / per-camera arrays
  // map tables
  double *T_kf_w = nullptr, *T_slot_w = nullptr, *pt_pos = nullptr, *obs_px = nullptr, *obs_f = nullptr, *obs_grad = nullptr;
  int *kf_slot = nullptr, *kf_key_point = nullptr, *kf_ftr_offset = nullptr, *kf_ftr_point = nullptr, *pt_type = nullptr, *pt_n_failed = nullptr,
      *pt_n_succeeded = nullptr, *pt_obs_offset = nullptr, *obs_kf = nullptr, *obs_level = nullptr, *cand_point = nullptr;
  uint8_t *pt_unlinked = nullptr, *obs_edgelet = nullptr;
  int n_kf = 0, n_points = 0, n_candidates = 0;
  bool have_map = false, have_last = false;
  bool rekey_pending = false;               // the last frame deleted points: keyframes that lost a key feature choose again before the next frame
  int last_n_host = 0;
  int last_max_point = -1;                  // largest map point index the last frame's features refer to (set_last_frame input)
  bool last_from_track = false;             // ... or: the last frame is the previous call's new frame, its features are in the result block
  // plan / replay scratch
  TrkPlan pl{};
  TrkFeat ft{};
  TrkLast last{};
  int *cell_winner = nullptr, *cell_cum = nullptr;
  bool need_gather = true;                  // the solver's slot 0 does not hold the last frame yet (a host upload came in between)
  bool any_edgelet = false;                 // the map holds EDGELET reference features (align1D stage needed)
  svo_hip_pose_opt_result* po = nullptr;    // (its entry of the shared array)
  // result block: [svo_hip_track_result][px][f][level][point][edgelet][grad][pt_type][pt_failed][pt_succeeded]
  char* res_dev = nullptr;                  // device address of res_host
  char* res_host = nullptr;                 // page-locked, mapped into the device: the hand-over kernel writes it directly
  unsigned long long seq = 0;               // frames tracked: the kernel stores it behind the block (o_flag) when the block is complete
  size_t o_flag = 0;
  uint8_t* img_dev = nullptr;               // device address of img_host (this camera's part of the shared image block)
  char* st_host = nullptr;                  // page-locked result of svo_hip_tracker_optimize_structure: [pos 64 x 3][iters 64][flag]
  char* st_dev = nullptr;
  unsigned long long st_seq = 0;
  size_t o_px = 0, o_f = 0, o_level = 0, o_point = 0, o_edge = 0, o_grad = 0, o_pt = 0, res_bytes = 0;
  // page-locked staging: one frame image (inside the shared block), the map tables
  uint8_t* img_host = nullptr;
  char* map_host = nullptr;
  size_t map_host_bytes = 0;
  std::vector<void*> dev_allocs;
};

namespace {

TrkMap make_map(const svo_hip_tracker* t) {
  TrkMap m;
  m.n_kf = t->n_kf; m.n_points = t->n_points; m.n_candidates = t->n_candidates;
  m.T_kf_w = t->T_kf_w; m.kf_slot = t->kf_slot; m.kf_key_point = t->kf_key_point; m.kf_ftr_offset = t->kf_ftr_offset;
  m.kf_ftr_point = t->kf_ftr_point; m.pt_pos = t->pt_pos; m.pt_type = t->pt_type; m.pt_n_failed = t->pt_n_failed;
  m.pt_n_succeeded = t->pt_n_succeeded; m.pt_unlinked = t->pt_unlinked; m.pt_obs_offset = t->pt_obs_offset; m.obs_kf = t->obs_kf;
  m.obs_px = t->obs_px; m.obs_f = t->obs_f; m.obs_level = t->obs_level; m.obs_edgelet = t->obs_edgelet; m.obs_grad = t->obs_grad;
  m.cand_point = t->cand_point;
  return m;
}

}  // namespace

extern "C" {

int svo_hip_tracker_default_config(svo_hip_tracker_config* c) {
  if (!c) return SVO_HIP_ERR_INVALID;
  memset(c, 0, sizeof(*c));
  c->max_keyframes = 64; c->max_points = 1 << 16; c->max_obs = 1 << 18; c->max_kf_features = 1 << 17; c->max_candidates = 1 << 14;
  c->max_items = 1 << 14; c->max_frame_features = 2048;
  c->n_levels = 5;                 // max(Config::nPyrLevels(), Config::kltMaxLevel() + 1) (frame.cpp:63)
  c->klt_max_level = 4; c->klt_min_level = 2; c->sia_n_iter = 30; c->sia_eps = 1e-6;       // config.cpp:62-63, frame_handler_mono.cpp:186-187
  c->grid_size = 20; c->max_fts = 1200; c->quality_min_fts = 40;                            // config.cpp:61,82,83
  c->reproj_max_n_kfs = 10;        // Reprojector::Options::max_n_kfs (I/reprojector.h:41)
  c->n_pyr_levels = 3; c->align_max_iter = 10;                                              // config.cpp:59, I/matcher.h:86
  c->pose_optim_thresh = 2.0; c->pose_optim_num_iter = 10;                                  // config.cpp:66-67
  return SVO_HIP_OK;
}

static void trk_shared_destroy(svo_hip_tracker_shared* sh) {
  if (!sh) return;
  if (sh->sia) svo_hip_sia_destroy(sh->sia);
  if (sh->kf_pyr) svo_hip_pyramid_destroy(sh->kf_pyr);
  for (int i = 0; i < 2; ++i) if (sh->frame_pyr[i]) svo_hip_pyramid_destroy(sh->frame_pyr[i]);
  for (void* p : sh->dev_allocs) if (p) (void)hipFree(p);
  if (sh->img_host) (void)hipHostFree(sh->img_host);
  if (sh->args_host) (void)hipHostFree(sh->args_host);
  delete sh;
}

// one camera's own objects (everything but what svo_hip_tracker_shared holds)
static void trk_member_destroy(svo_hip_tracker* t) {
  for (void* p : t->dev_allocs) if (p) (void)hipFree(p);
  if (t->res_host) (void)hipHostFree(t->res_host);
  if (t->st_host) (void)hipHostFree(t->st_host);
  if (t->map_host) (void)hipHostFree(t->map_host);
  delete t;
}

int svo_hip_tracker_destroy(svo_hip_tracker* t) {
  if (!t) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = t->ctx;
  if (t->sh && !t->owns_shared) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_tracker_destroy", "a camera of a tracker group goes with its group (svo_hip_tracker_group_destroy)");
  (void)hipStreamSynchronize(ctx->stream);
  svo_hip_tracker_shared* sh = t->sh;
  trk_member_destroy(t);
  trk_shared_destroy(sh);
  return SVO_HIP_OK;
}

static int trk_check_config(svo_hip_ctx* ctx, const svo_hip_camera* cam, const svo_hip_tracker_config* cfg) {
  SVO_REQUIRE(ctx, cfg->max_keyframes <= TRK_LDS_KF);
  SVO_REQUIRE(ctx, cfg->max_keyframes >= 1 && cfg->max_points >= 1 && cfg->max_obs >= 1 && cfg->max_kf_features >= 1 && cfg->max_candidates >= 0);
  SVO_REQUIRE(ctx, cfg->max_items >= 1 && cfg->max_frame_features >= 1 && cfg->max_frame_features <= 2816);
  // the cell loop stops AFTER the match that exceeds max_fts (reprojector.cpp:164-165): a frame can gain max_fts + 1 features
  SVO_REQUIRE(ctx, cfg->max_frame_features >= cfg->max_fts + 1);
  SVO_REQUIRE(ctx, cfg->n_levels >= 1 && cfg->n_levels <= SVO_HIP_MAX_LEVELS && cfg->klt_max_level < cfg->n_levels && cfg->klt_min_level >= 0 &&
                       cfg->klt_min_level <= cfg->klt_max_level && cfg->sia_n_iter >= 0);
  SVO_REQUIRE(ctx, cfg->grid_size >= 1 && cfg->max_fts >= 0 && cfg->reproj_max_n_kfs >= 1 && cfg->reproj_max_n_kfs <= TRK_MAX_SEL);
  SVO_REQUIRE(ctx, cfg->n_pyr_levels >= 1 && cfg->n_pyr_levels <= cfg->n_levels && cfg->align_max_iter >= 0 && cfg->pose_optim_num_iter >= 0);
  SVO_REQUIRE(ctx, cam->width > 16 && cam->height > 16);
  return SVO_HIP_OK;
}

// the shared part of n_cams cameras with one camera model and one configuration
static int trk_shared_create(svo_hip_ctx* ctx, const svo_hip_camera* cam, const svo_hip_tracker_config* cfg, int n_cams, svo_hip_tracker_shared** out) {
  *out = nullptr;
  svo_hip_tracker_shared* sh = new (std::nothrow) svo_hip_tracker_shared();
  if (!sh) return SVO_HIP_ERR_NOMEM;
  sh->ctx = ctx; sh->n_cams = n_cams;
  int rc = SVO_HIP_OK;
  auto A = [&](int r) { if (rc == SVO_HIP_OK) rc = r; };
  auto D = [&](auto** p, size_t count) {
    if (rc != SVO_HIP_OK) return;
    rc = trk_alloc(ctx, p, count);
    if (rc == SVO_HIP_OK) sh->dev_allocs.push_back((void*)*p);
  };
  A(svo_hip_pyramid_create(ctx, cam->width, cam->height, cfg->n_levels, n_cams * cfg->max_keyframes, &sh->kf_pyr));
  for (int i = 0; i < 2; ++i) A(svo_hip_pyramid_create(ctx, cam->width, cam->height, cfg->n_levels, n_cams, &sh->frame_pyr[i]));
  A(svo_hip_sia_create(ctx, n_cams, cfg->max_frame_features, &sh->sia));
  // (the library default, pinned here: the chain's decisions -- matches per cell, frame by frame -- equal the CPU chain's)
  if (rc == SVO_HIP_OK) rc = svo_hip_sia_set_option(sh->sia, SVO_HIP_SIA_OPT_ARITH, SVO_HIP_SIA_ARITH_EXACT);
  const size_t N = (size_t)n_cams, C = cfg->max_items, NF = cfg->max_frame_features;
  D(&sh->counters, N * 8); D(&sh->cand_level_ref, N * C);
  D(&sh->ft_f, N * NF * 3); D(&sh->ft_pos, N * NF * 3); D(&sh->ft_level, N * NF); D(&sh->ft_has_point, N * NF); D(&sh->po, N);
  sh->img_stride = ((size_t)cam->width * cam->height + 64 + 255) & ~(size_t)255;
  if (rc == SVO_HIP_OK && hipHostMalloc((void**)&sh->img_host, N * sh->img_stride, hipHostMallocMapped) != hipSuccess) rc = SVO_HIP_ERR_NOMEM;
  if (rc == SVO_HIP_OK && hipHostGetDevicePointer((void**)&sh->img_dev, sh->img_host, 0) != hipSuccess) rc = SVO_HIP_ERR_DEVICE;
  if (n_cams > 1) {
    if (rc == SVO_HIP_OK && hipHostMalloc((void**)&sh->args_host, N * sizeof(TrkCamArgs), hipHostMallocDefault) != hipSuccess) rc = SVO_HIP_ERR_NOMEM;
    D(&sh->args_dev, N * sizeof(TrkCamArgs));
  }
  if (rc != SVO_HIP_OK) { trk_shared_destroy(sh); return rc; }
  *out = sh;
  return SVO_HIP_OK;
}

// camera `index` of the shared part: its own map tables, plan scratch, result block
static int trk_member_create(svo_hip_tracker_shared* sh, int index, const svo_hip_camera* cam, const svo_hip_tracker_config* cfg, svo_hip_tracker** out) {
  svo_hip_ctx* ctx = sh->ctx;
  *out = nullptr;
  svo_hip_tracker* t = new (std::nothrow) svo_hip_tracker();
  if (!t) return SVO_HIP_ERR_NOMEM;
  t->ctx = ctx; t->cam = *cam; t->cfg = *cfg; t->sh = sh; t->cam_index = index;
  t->grid_cols = (cam->width + cfg->grid_size - 1) / cfg->grid_size;        // ceil(width / cell_size) (reprojector.cpp:46-47)
  t->grid_rows = (cam->height + cfg->grid_size - 1) / cfg->grid_size;
  t->n_cells = t->grid_cols * t->grid_rows;
  int rc = SVO_HIP_OK;
  auto D = [&](auto** p, size_t count) {
    if (rc != SVO_HIP_OK) return;
    rc = trk_alloc(ctx, p, count);
    if (rc == SVO_HIP_OK) t->dev_allocs.push_back((void*)*p);
  };
  const size_t K = cfg->max_keyframes, P = cfg->max_points, O = cfg->max_obs, F = cfg->max_kf_features, CN = cfg->max_candidates > 0 ? cfg->max_candidates : 1;
  D(&t->T_kf_w, K * 7); D(&t->T_slot_w, K * 7); D(&t->kf_slot, K); D(&t->kf_key_point, K * 5); D(&t->kf_ftr_offset, K + 1); D(&t->kf_ftr_point, F);
  D(&t->pt_pos, P * 3); D(&t->pt_type, P); D(&t->pt_n_failed, P); D(&t->pt_n_succeeded, P); D(&t->pt_unlinked, P); D(&t->pt_obs_offset, P + 1);
  D(&t->obs_kf, O); D(&t->obs_px, O * 2); D(&t->obs_f, O * 3); D(&t->obs_level, O); D(&t->obs_edgelet, O); D(&t->obs_grad, O * 2); D(&t->cand_point, CN);
  TrkPlan& pl = t->pl;
  const size_t C = cfg->max_items, NC = t->n_cells;
  pl.cap = cfg->max_items; pl.n_cells = t->n_cells; pl.grid_cols = t->grid_cols; pl.grid_size = cfg->grid_size; pl.max_n_kfs = cfg->reproj_max_n_kfs;
  D(&pl.first_seq, P); D(&pl.item_point, C); D(&pl.item_px, C * 2); D(&pl.item_cell, C); D(&pl.item_key, C);
  D(&pl.seg, C); D(&pl.seg_key, C); D(&pl.cell_count, NC + 1); D(&pl.cell_fill, NC); D(&pl.cell_offset, NC + 1); D(&pl.overlap_kf, TRK_MAX_SEL);
  D(&pl.overlap_count, TRK_MAX_SEL); D(&pl.cand_point, C); D(&pl.cand_obs, C); D(&pl.cand_deleted, C);
  pl.counters = sh->counters + 8 * (size_t)index;                              // (per-camera entries of the shared arrays)
  pl.cand_level_ref = sh->cand_level_ref + C * (size_t)index;
  D(&t->cell_winner, NC + 1); D(&t->cell_cum, NC + 1);
  t->po = sh->po + index;
  const size_t NF = cfg->max_frame_features;
  TrkFeat& ft = t->ft;
  ft.cap = cfg->max_frame_features;
  D(&ft.px, NF * 2); D(&ft.point, NF); D(&ft.edgelet, NF); D(&ft.grad, NF * 2);
  ft.f = sh->ft_f + 3 * NF * (size_t)index; ft.pos = sh->ft_pos + 3 * NF * (size_t)index;
  ft.level = sh->ft_level + NF * (size_t)index; ft.has_point = sh->ft_has_point + NF * (size_t)index;
  D(&t->last.n, 1); D(&t->last.T_f_w, 7); D(&t->last.px, NF * 2); D(&t->last.f, NF * 3); D(&t->last.point, NF);
  // result block
  auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
  t->o_px = al(sizeof(svo_hip_track_result)); t->o_f = al(t->o_px + NF * 16); t->o_level = al(t->o_f + NF * 24); t->o_point = al(t->o_level + NF * 4);
  t->o_edge = al(t->o_point + NF * 4); t->o_grad = al(t->o_edge + NF); t->o_pt = al(t->o_grad + NF * 16); t->o_flag = al(t->o_pt + P * 12);
  t->res_bytes = t->o_flag + 64;
  if (rc == SVO_HIP_OK && hipHostMalloc((void**)&t->res_host, t->res_bytes, hipHostMallocMapped) != hipSuccess) rc = SVO_HIP_ERR_NOMEM;
  if (rc == SVO_HIP_OK && hipHostGetDevicePointer((void**)&t->res_dev, t->res_host, 0) != hipSuccess) rc = SVO_HIP_ERR_DEVICE;
  if (rc == SVO_HIP_OK) memset(t->res_host, 0, t->res_bytes);
  t->img_host = sh->img_host + sh->img_stride * (size_t)index;
  t->img_dev = sh->img_dev + sh->img_stride * (size_t)index;
  const size_t st_bytes = TRK_MAX_STRUCT * (24 + 4) + 64;
  if (rc == SVO_HIP_OK && hipHostMalloc((void**)&t->st_host, st_bytes, hipHostMallocMapped) != hipSuccess) rc = SVO_HIP_ERR_NOMEM;
  if (rc == SVO_HIP_OK) memset(t->st_host, 0, st_bytes);
  if (rc == SVO_HIP_OK && hipHostGetDevicePointer((void**)&t->st_dev, t->st_host, 0) != hipSuccess) rc = SVO_HIP_ERR_DEVICE;
  // staging area of svo_hip_tracker_set_map: every table at its capacity (T_kf_w and T_slot_w: two pose tables), 16 bytes of
  // alignment slack per table
  t->map_host_bytes = K * (56 + 56 + 4 + 20 + 4) + 8 + F * 4 + P * (24 + 12 + 4) + 8 + O * (4 + 16 + 24 + 4 + 1 + 16) + CN * 4 + 32 * 16;
  if (rc == SVO_HIP_OK && hipHostMalloc((void**)&t->map_host, t->map_host_bytes, hipHostMallocDefault) != hipSuccess) rc = SVO_HIP_ERR_NOMEM;
  if (rc != SVO_HIP_OK) { trk_member_destroy(t); return rc; }
  (void)hipMemsetAsync(t->last.n, 0, sizeof(int), ctx->stream);
  svo_sia_slot_arrays(sh->sia, index, &t->last.sia_fc, &t->last.sia_px, &t->last.sia_f, &t->last.sia_pos, &t->last.sia_has_point, &t->last.sia_max_n);
  *out = t;
  return SVO_HIP_OK;
}

int svo_hip_tracker_create(svo_hip_ctx* ctx, const svo_hip_camera* cam, const svo_hip_tracker_config* cfg, svo_hip_tracker** out) {
  if (!ctx || !cam || !cfg || !out) return SVO_HIP_ERR_INVALID;
  *out = nullptr;
  int rc = trk_check_config(ctx, cam, cfg);
  if (rc != SVO_HIP_OK) return rc;
  svo_hip_tracker_shared* sh = nullptr;
  rc = trk_shared_create(ctx, cam, cfg, 1, &sh);
  if (rc != SVO_HIP_OK) return rc;
  svo_hip_tracker* t = nullptr;
  rc = trk_member_create(sh, 0, cam, cfg, &t);
  if (rc != SVO_HIP_OK) { trk_shared_destroy(sh); return rc; }
  t->owns_shared = true;
  sh->members.push_back(t);
  *out = t;
  return SVO_HIP_OK;
}

int svo_hip_tracker_upload_keyframe(svo_hip_tracker* t, int slot, const uint8_t* level0) {
  if (!t || !level0) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(t->ctx, slot >= 0 && slot < t->cfg.max_keyframes);
  return svo_hip_pyramid_upload_level0_and_build(t->sh->kf_pyr, t->cam_index * t->cfg.max_keyframes + slot, level0);
}

int svo_hip_tracker_keyframe_from_last_frame(svo_hip_tracker* t, int slot) {
  if (!t) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = t->ctx;
  SVO_REQUIRE(ctx, slot >= 0 && slot < t->cfg.max_keyframes);
  if (!t->have_last) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_tracker_keyframe_from_last_frame", "no frame has been tracked or set yet");
  const svo_hip_pyramid* src = t->sh->frame_pyr[t->sh->last_idx];
  const svo_hip_pyramid* kf = t->sh->kf_pyr;
  return svo_hip_copy_d2d(ctx, kf->base + ((size_t)t->cam_index * t->cfg.max_keyframes + slot) * kf->pyr_bytes,
                          src->base + (size_t)t->cam_index * src->pyr_bytes, src->pyr_bytes);
}

int svo_hip_tracker_set_map(svo_hip_tracker* t, const svo_hip_tracker_map* mp) {
  if (!t || !mp) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = t->ctx;
  const svo_hip_tracker_config& c = t->cfg;
  SVO_REQUIRE(ctx, mp->n_kf >= 0 && mp->n_kf <= c.max_keyframes && mp->n_points >= 0 && mp->n_points <= c.max_points);
  SVO_REQUIRE(ctx, mp->n_candidates >= 0 && mp->n_candidates <= c.max_candidates);
  SVO_REQUIRE(ctx, mp->n_kf == 0 || (mp->kf_slot && mp->T_kf_w && mp->kf_key_point && mp->kf_ftr_offset));
  SVO_REQUIRE(ctx, mp->n_points == 0 || (mp->pt_pos && mp->pt_type && mp->pt_n_failed && mp->pt_n_succeeded && mp->pt_obs_offset));
  SVO_REQUIRE(ctx, mp->n_candidates == 0 || mp->cand_point);
  const int n_ftr = mp->n_kf > 0 ? mp->kf_ftr_offset[mp->n_kf] : 0;
  const int n_obs = mp->n_points > 0 ? mp->pt_obs_offset[mp->n_points] : 0;
  SVO_REQUIRE(ctx, n_ftr >= 0 && n_ftr <= c.max_kf_features && n_obs >= 0 && n_obs <= c.max_obs);
  SVO_REQUIRE(ctx, n_ftr == 0 || mp->kf_ftr_point);
  SVO_REQUIRE(ctx, n_obs == 0 || (mp->obs_kf && mp->obs_px && mp->obs_f && mp->obs_level));
  // every index the kernels follow is checked here, once, on the host
  for (int k = 0; k < mp->n_kf; ++k) {
    SVO_REQUIRE(ctx, mp->kf_slot[k] >= 0 && mp->kf_slot[k] < c.max_keyframes && mp->kf_ftr_offset[k] <= mp->kf_ftr_offset[k + 1] && mp->kf_ftr_offset[k] >= 0);
    for (int j = 0; j < 5; ++j) SVO_REQUIRE(ctx, mp->kf_key_point[5 * k + j] >= -1 && mp->kf_key_point[5 * k + j] < mp->n_points);
  }
  for (int j = 0; j < n_ftr; ++j) SVO_REQUIRE(ctx, mp->kf_ftr_point[j] >= -1 && mp->kf_ftr_point[j] < mp->n_points);
  for (int p = 0; p < mp->n_points; ++p) SVO_REQUIRE(ctx, mp->pt_obs_offset[p] >= 0 && mp->pt_obs_offset[p] <= mp->pt_obs_offset[p + 1] && mp->pt_type[p] >= 0 && mp->pt_type[p] <= 3);
  for (int o = 0; o < n_obs; ++o) SVO_REQUIRE(ctx, mp->obs_kf[o] >= 0 && mp->obs_kf[o] < mp->n_kf && mp->obs_level[o] >= 0 && mp->obs_level[o] < c.n_levels);
  for (int i = 0; i < mp->n_candidates; ++i) SVO_REQUIRE(ctx, mp->cand_point[i] >= -1 && mp->cand_point[i] < mp->n_points);
  {
    // what the tables below take in the staging area (each starts on a 16-byte boundary): checked before anything is written
    const size_t K_ = (size_t)mp->n_kf, P_ = (size_t)mp->n_points, O_ = (size_t)n_obs;
    const size_t need = K_ * (56 + 4 + 20) + (K_ + 1) * 4 + (size_t)n_ftr * 4 + P_ * (24 + 12) + (P_ + 1) * 4 + O_ * (4 + 16 + 24 + 4 + 1 + 16) +
                        (size_t)mp->n_candidates * 4 + (size_t)c.max_keyframes * 56 + 32 * 16;
    if (need > t->map_host_bytes) return svo_fail(ctx, SVO_HIP_ERR_NOMEM, "svo_hip_tracker_set_map", "staging area too small");
  }
  if (t->have_last) {                       // (after the checks: a refused map changes nothing)
    // the last frame's features keep referring to map points by index: a map with fewer points than the largest of them
    // needs svo_hip_tracker_set_last_frame again (the result block still holds the features of a tracked frame)
    int max_point = t->last_max_point;
    if (t->last_from_track) {
      const int32_t* fp = reinterpret_cast<const int32_t*>(t->res_host + t->o_point);
      max_point = -1;
      for (int i = 0; i < t->last_n_host; ++i) if (fp[i] > max_point) max_point = fp[i];
    }
    if (max_point >= mp->n_points) t->have_last = false;     // svo_hip_tracker_track will ask for svo_hip_tracker_set_last_frame
  }
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  // the staging area may still feed the copies of the previous upload
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  char* hs = t->map_host;
  size_t off = 0;
  hipError_t e = hipSuccess;
  auto put = [&](void* dst, const void* src, size_t bytes) {
    if (!bytes || e != hipSuccess) return;
    off = (off + 15) & ~(size_t)15;
    memcpy(hs + off, src, bytes);
    e = hipMemcpyAsync(dst, hs + off, bytes, hipMemcpyHostToDevice, ctx->stream);
    off += bytes;
  };
  const size_t K = mp->n_kf, P = mp->n_points;
  put(t->T_kf_w, mp->T_kf_w, K * 56); put(t->kf_slot, mp->kf_slot, K * 4); put(t->kf_key_point, mp->kf_key_point, K * 20);
  if (K) put(t->kf_ftr_offset, mp->kf_ftr_offset, (K + 1) * 4);
  put(t->kf_ftr_point, mp->kf_ftr_point, (size_t)n_ftr * 4);
  put(t->pt_pos, mp->pt_pos, P * 24); put(t->pt_type, mp->pt_type, P * 4); put(t->pt_n_failed, mp->pt_n_failed, P * 4);
  put(t->pt_n_succeeded, mp->pt_n_succeeded, P * 4);
  if (P) put(t->pt_obs_offset, mp->pt_obs_offset, (P + 1) * 4);
  put(t->obs_kf, mp->obs_kf, (size_t)n_obs * 4); put(t->obs_px, mp->obs_px, (size_t)n_obs * 16); put(t->obs_f, mp->obs_f, (size_t)n_obs * 24);
  put(t->obs_level, mp->obs_level, (size_t)n_obs * 4);
  put(t->cand_point, mp->cand_point, (size_t)mp->n_candidates * 4);
  // T_slot_w: the keyframe poses indexed by pyramid slot (what the matcher's batch indexes)
  {
    off = (off + 15) & ~(size_t)15;
    double* ts = reinterpret_cast<double*>(hs + off);
    memset(ts, 0, (size_t)c.max_keyframes * 56);
    for (int s = 0; s < c.max_keyframes; ++s) ts[7 * s + 6] = 1.0;
    for (int k = 0; k < mp->n_kf; ++k) memcpy(ts + 7 * (size_t)mp->kf_slot[k], mp->T_kf_w + 7 * (size_t)k, 56);
    if (e == hipSuccess) e = hipMemcpyAsync(t->T_slot_w, ts, (size_t)c.max_keyframes * 56, hipMemcpyHostToDevice, ctx->stream);
    off += (size_t)c.max_keyframes * 56;
  }
  if (n_obs) {
    off = (off + 15) & ~(size_t)15;
    uint8_t* ed = reinterpret_cast<uint8_t*>(hs + off);
    if (mp->obs_edgelet) memcpy(ed, mp->obs_edgelet, (size_t)n_obs); else memset(ed, 0, (size_t)n_obs);
    if (e == hipSuccess) e = hipMemcpyAsync(t->obs_edgelet, ed, (size_t)n_obs, hipMemcpyHostToDevice, ctx->stream);
    off += (size_t)n_obs;
    off = (off + 15) & ~(size_t)15;
    double* gr = reinterpret_cast<double*>(hs + off);
    if (mp->obs_grad) memcpy(gr, mp->obs_grad, (size_t)n_obs * 16);
    else for (int o = 0; o < n_obs; ++o) { gr[2 * o] = 1.0; gr[2 * o + 1] = 0.0; }
    if (e == hipSuccess) e = hipMemcpyAsync(t->obs_grad, gr, (size_t)n_obs * 16, hipMemcpyHostToDevice, ctx->stream);
    off += (size_t)n_obs * 16;
  }
  if (e == hipSuccess && P) e = hipMemsetAsync(t->pt_unlinked, 0, P, ctx->stream);
  if (e != hipSuccess) return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_tracker_set_map", hipGetErrorString(e));
  t->n_kf = mp->n_kf; t->n_points = mp->n_points; t->n_candidates = mp->n_candidates;
  t->any_edgelet = false;
  if (mp->obs_edgelet) for (int o = 0; o < n_obs && !t->any_edgelet; ++o) t->any_edgelet = mp->obs_edgelet[o] != 0;
  t->have_map = true; t->rekey_pending = false;
  t->need_gather = true;                    // point positions may have changed
  return SVO_HIP_OK;
}

int svo_hip_tracker_update_point_positions(svo_hip_tracker* t, int n, const int32_t* point, const double* pos) {
  if (!t) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = t->ctx;
  SVO_REQUIRE(ctx, n >= 0 && (n == 0 || (point && pos)));
  if (!t->have_map) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_tracker_update_point_positions", "no map has been set");
  if (n == 0) return SVO_HIP_OK;
  for (int i = 0; i < n; ++i) SVO_REQUIRE(ctx, point[i] >= 0 && point[i] < t->n_points);
  // [pos n x 3 doubles][index n ints] gathered in page-locked memory: one transfer, one scatter launch
  const size_t o_idx = (size_t)n * 24, bytes = o_idx + (size_t)n * 4;
  char* d = nullptr;
  char* hs = nullptr;
  int rc = svo_ctx_staging(ctx, bytes, &d);
  if (rc == SVO_HIP_OK) rc = svo_ctx_host_staging(ctx, bytes, &hs);
  if (rc != SVO_HIP_OK) return rc;
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));      // the staging area may still feed an earlier transfer
  memcpy(hs, pos, (size_t)n * 24);
  memcpy(hs + o_idx, point, (size_t)n * 4);
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(d, hs, bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(trk_scatter_positions_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, reinterpret_cast<const int*>(d + o_idx),
                     reinterpret_cast<const double*>(d), t->pt_pos);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  t->need_gather = true;                    // the solver's copy of the last frame's point positions is stale
  return SVO_HIP_OK;
}

int svo_hip_tracker_optimize_structure(svo_hip_tracker* t, int n, const int32_t* point, int n_iter, double* pos_out, int32_t* iters_out) {
  if (!t) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = t->ctx;
  SVO_REQUIRE(ctx, n >= 0 && n <= TRK_MAX_STRUCT && n_iter >= 0 && (n == 0 || (point && pos_out)));
  if (!t->have_map) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_tracker_optimize_structure", "no map has been set");
  if (n == 0) return SVO_HIP_OK;
  TrkStructSel sel;
  memset(&sel, 0, sizeof(sel));
  sel.n = n;
  for (int i = 0; i < n; ++i) {
    SVO_REQUIRE(ctx, point[i] >= 0 && point[i] < t->n_points);
    for (int j = 0; j < i; ++j) SVO_REQUIRE(ctx, point[j] != point[i]);         // a point is optimised once per call
    sel.point[i] = point[i];
  }
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  const size_t o_it = TRK_MAX_STRUCT * 24, o_flag = o_it + TRK_MAX_STRUCT * 4;
  const unsigned long long seq = ++t->st_seq;
  hipLaunchKernelGGL(trk_structure_kernel, dim3(1), dim3(256), 0, ctx->stream, make_map(t), t->pt_pos, t->last, sel, n_iter,
                     reinterpret_cast<double*>(t->st_dev), reinterpret_cast<int*>(t->st_dev + o_it),
                     reinterpret_cast<unsigned long long*>(t->st_dev + o_flag), seq);
  SVO_CHECK_HIP(ctx, hipGetLastError());
  {
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(t->st_host + o_flag);
    bool seen = false;
    for (long spins = 0; spins < 4000000L; ++spins) {
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
      __builtin_ia32_pause();
    }
    if (!seen) SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
      return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_tracker_optimize_structure", "the kernel did not complete");
  }
  memcpy(pos_out, t->st_host, (size_t)n * 24);
  if (iters_out) memcpy(iters_out, t->st_host + o_it, (size_t)n * 4);
  // (the point table and the solver's copy of the last frame's points were updated by the kernel: nothing to gather)
  return SVO_HIP_OK;
}

int svo_hip_tracker_set_last_frame(svo_hip_tracker* t, const uint8_t* level0, int kf_slot, const double T_f_w[7], int n, const double* px,
                                   const double* f, const int32_t* point) {
  if (!t || !T_f_w) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = t->ctx;
  SVO_REQUIRE(ctx, n >= 0 && n <= t->cfg.max_frame_features && (n == 0 || (px && f && point)));
  SVO_REQUIRE(ctx, level0 || (kf_slot >= 0 && kf_slot < t->cfg.max_keyframes));
  int max_point = -1;
  for (int i = 0; i < n; ++i) {                               // the alignment reads pt_pos[point]: every index is checked here
    SVO_REQUIRE(ctx, point[i] >= -1 && point[i] < (t->have_map ? t->n_points : 0));
    if (point[i] > max_point) max_point = point[i];
  }
  svo_hip_pyramid* dst = t->sh->frame_pyr[t->sh->last_idx];
  const svo_hip_pyramid* kf = t->sh->kf_pyr;
  int rc;
  if (level0) rc = svo_hip_pyramid_upload_level0_and_build(dst, t->cam_index, level0);
  else rc = svo_hip_copy_d2d(ctx, dst->base + (size_t)t->cam_index * dst->pyr_bytes,
                             kf->base + ((size_t)t->cam_index * t->cfg.max_keyframes + kf_slot) * kf->pyr_bytes, dst->pyr_bytes);
  if (rc != SVO_HIP_OK) return rc;
  const int32_t n32 = n;
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(t->last.n, &n32, 4, hipMemcpyHostToDevice, ctx->stream));
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(t->last.T_f_w, T_f_w, 56, hipMemcpyHostToDevice, ctx->stream));
  if (n > 0) {
    SVO_CHECK_HIP(ctx, hipMemcpyAsync(t->last.px, px, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    SVO_CHECK_HIP(ctx, hipMemcpyAsync(t->last.f, f, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    SVO_CHECK_HIP(ctx, hipMemcpyAsync(t->last.point, point, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  t->last_n_host = n;
  t->last_max_point = max_point;
  t->last_from_track = false;
  t->have_last = true;
  t->need_gather = true;
  return SVO_HIP_OK;
}

// a tracked frame's outcome from the camera's page-locked result block (valid until the camera's next frame)
static void trk_copy_out(svo_hip_tracker* t, svo_hip_track_result* result, double* feat_px, double* feat_f, int32_t* feat_level, int32_t* feat_point,
                         uint8_t* feat_edgelet, double* feat_grad, int32_t* pt_type, int32_t* pt_n_failed, int32_t* pt_n_succeeded) {
  const char* rh = t->res_host;
  svo_hip_track_result r;
  memcpy(&r, rh, sizeof(r));
  if (result) *result = r;
  const size_t nf = (size_t)(r.n_features > 0 ? r.n_features : 0);
  if (feat_px) memcpy(feat_px, rh + t->o_px, nf * 16);
  if (feat_f) memcpy(feat_f, rh + t->o_f, nf * 24);
  if (feat_level) memcpy(feat_level, rh + t->o_level, nf * 4);
  if (feat_point) memcpy(feat_point, rh + t->o_point, nf * 4);
  if (feat_edgelet) memcpy(feat_edgelet, rh + t->o_edge, nf);
  if (feat_grad) memcpy(feat_grad, rh + t->o_grad, nf * 16);
  const size_t np4 = (size_t)t->n_points * 4;
  if (pt_type) memcpy(pt_type, rh + t->o_pt, np4);
  if (pt_n_failed) memcpy(pt_n_failed, rh + t->o_pt + np4, np4);
  if (pt_n_succeeded) memcpy(pt_n_succeeded, rh + t->o_pt + 2 * np4, np4);
}

// One frame for every camera of the shared part: ONE chain of launches whatever the number of cameras.  level0[c]: camera c's
// new image (its own page-locked buffer: no copy).
static int trk_track_all(svo_hip_tracker_shared* sh, const uint8_t* const* level0, const char* who) {
  svo_hip_ctx* ctx = sh->ctx;
  const int N = sh->n_cams;
  svo_hip_tracker* t0 = sh->members[0];
  const svo_hip_tracker_config& c = t0->cfg;
  for (int k = 0; k < N; ++k) {
    svo_hip_tracker* t = sh->members[(size_t)k];
    if (!t->have_map || !t->have_last)
      return svo_fail(ctx, SVO_HIP_ERR_STATE, who, "svo_hip_tracker_set_map and svo_hip_tracker_set_last_frame come first (every camera)");
  }
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  const size_t l0 = (size_t)t0->cam.width * t0->cam.height;
  svo_hip_pyramid* ref = sh->frame_pyr[sh->last_idx];
  svo_hip_pyramid* cur = sh->frame_pyr[1 - sh->last_idx];
  // ---- new Frame(cam, img, t): the images cross the link once, from page-locked memory; the pyramids are built on the device
  for (int k = 0; k < N; ++k) {
    svo_hip_tracker* t = sh->members[(size_t)k];
    if (level0[k] != reinterpret_cast<const uint8_t*>(t->img_host)) memcpy(t->img_host, level0[k], l0);      // (svo_hip_tracker_image_buffer: already there)
  }
  int rc = svo_pyramid_build_levels(cur, 0, N, sh->img_dev, sh->img_stride);
  if (rc != SVO_HIP_OK) return rc;
  // ---- SparseImgAlign(kltMaxLevel, kltMinLevel, 30, GaussNewton, false, false).run(last_frame_, new_frame_)
  rc = svo_hip_sia_set_frames(sh->sia, ref, cur);
  if (rc != SVO_HIP_OK) return rc;
  // the previous call's hand-over kernel has written the solver's inputs already, unless the host changed the last frame,
  // the map or point positions since
  for (int k = 0; k < N; ++k) {
    svo_hip_tracker* t = sh->members[(size_t)k];
    if (t->need_gather) rc = svo_sia_prepare_from_device(sh->sia, k, &t->cam, t->last_n_host, t->last.n, t->last.T_f_w, t->last.px, t->last.f, t->last.point, t->pt_pos);
    else rc = svo_sia_note_device_slot(sh->sia, k, &t->cam, t->last_n_host);
    if (rc != SVO_HIP_OK) return rc;
  }
  svo_hip_sia_params sp;
  sp.max_level = c.klt_max_level; sp.min_level = c.klt_min_level; sp.n_iter = c.sia_n_iter; sp.eps = c.sia_eps; sp.early_stop = 1;
  rc = svo_hip_sia_run(sh->sia, N, &sp);
  if (rc != SVO_HIP_OK) return rc;
  // ---- Reprojector::reprojectMap
  const Cam cam = svo_make_cam(t0->cam);
  bool any_edgelet = false;
  for (int k = 0; k < N; ++k) {
    svo_hip_tracker* t = sh->members[(size_t)k];
    if (t->rekey_pending && t->n_kf > 0) {    // Map::safeDeletePoint of the previous frame: Frame::removeKeyPoint on the keyframes
      hipLaunchKernelGGL(trk_rekey_kernel, dim3(t->n_kf), dim3(256), 0, ctx->stream, make_map(t), t->kf_key_point, cam);
      SVO_CHECK_HIP(ctx, hipGetLastError());
    }
    t->rekey_pending = false;
    any_edgelet = any_edgelet || t->any_edgelet;
  }
  SeedRec* recs = nullptr;
  uint32_t* pwb_t = nullptr;
  int n_pad = 0;
  const int cap = t0->pl.cap;
  rc = svo_match_scratch(ctx, N * cap, &recs, &pwb_t, &n_pad);
  if (rc != SVO_HIP_OK) return rc;
  MdFrame mf;
  memset(&mf, 0, sizeof(mf));
  mf.cam = cam; mf.n_pyr_levels = c.n_pyr_levels; mf.n_kf = c.max_keyframes; mf.n_ref_levels = c.n_levels;
  if (N == 1) {
    svo_hip_tracker* t = t0;
    const TrkMap m = make_map(t);
    const FrameState* st = svo_sia_state_dev(sh->sia, 0);
    hipLaunchKernelGGL(trk_plan_kernel, dim3(1), dim3(TRK_THREADS), 0, ctx->stream, m, t->pl, mf, t->T_slot_w, recs, st);
    SVO_CHECK_HIP(ctx, hipGetLastError());
    // warp + align2D (+ align1D when the map holds edgelets) over the candidates
    rc = svo_match_stages(ctx, sh->kf_pyr, cur, 0, &t->cam, t->pl.cap, t->pl.counters, t->pl.cand_level_ref, recs, pwb_t, n_pad, c.n_pyr_levels,
                          c.align_max_iter, t->any_edgelet);
    if (rc != SVO_HIP_OK) return rc;
    hipLaunchKernelGGL(trk_replay_kernel, dim3(1), dim3(TRK_THREADS), 0, ctx->stream, m, t->pl, t->ft, cam, st, recs, t->cell_winner, t->cell_cum,
                       c.max_fts, c.quality_min_fts);
    SVO_CHECK_HIP(ctx, hipGetLastError());
    // ---- pose_optimizer::optimizeGaussNewton(poseOptimThresh, poseOptimNumIter, ...) on the matched features, from the aligned pose
    rc = svo_hip_pose_optimize_batch_dev(ctx, 1, c.max_frame_features, t->pl.counters + 5, st->T_cur_w, t->ft.f, t->ft.pos, t->ft.level, t->ft.has_point,
                                         fabs(t->cam.fx), c.pose_optim_thresh, c.pose_optim_num_iter, t->po);
    if (rc != SVO_HIP_OK) return rc;
    // ---- hand-over + result: written straight into the page-locked block, the frame's sequence number last
    char* rd = t->res_dev;
    const unsigned long long seq = ++t->seq;
    hipLaunchKernelGGL(trk_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, m, t->pl, t->ft, t->last, cam, st, t->po,
                       reinterpret_cast<svo_hip_track_result*>(rd), reinterpret_cast<double*>(rd + t->o_px), reinterpret_cast<double*>(rd + t->o_f),
                       reinterpret_cast<int*>(rd + t->o_level), reinterpret_cast<int*>(rd + t->o_point), reinterpret_cast<uint8_t*>(rd + t->o_edge),
                       reinterpret_cast<double*>(rd + t->o_grad), reinterpret_cast<int*>(rd + t->o_pt), reinterpret_cast<int*>(rd + t->o_pt) + t->n_points,
                       reinterpret_cast<int*>(rd + t->o_pt) + 2 * (size_t)t->n_points, reinterpret_cast<unsigned long long*>(rd + t->o_flag), seq);
    SVO_CHECK_HIP(ctx, hipGetLastError());
  } else {
    // the cameras' arguments: one table, one transfer (the previous call's kernels are through with it: this thread waited for them)
    TrkCamArgs* ah = reinterpret_cast<TrkCamArgs*>(sh->args_host);
    for (int k = 0; k < N; ++k) {
      svo_hip_tracker* t = sh->members[(size_t)k];
      TrkCamArgs& a = ah[k];
      memset(&a, 0, sizeof(a));
      a.m = make_map(t); a.pl = t->pl; a.ft = t->ft; a.last = t->last;
      a.mf = mf; a.mf.slot_base = k * c.max_keyframes;
      a.T_slot_w = t->T_slot_w;
      a.recs = recs + (size_t)k * cap;
      a.sia_state = svo_sia_state_dev(sh->sia, k);
      a.cell_winner = t->cell_winner; a.cell_cum = t->cell_cum;
      a.po = t->po;
      a.res = t->res_dev;
      a.o_px = t->o_px; a.o_f = t->o_f; a.o_level = t->o_level; a.o_point = t->o_point; a.o_edge = t->o_edge; a.o_grad = t->o_grad; a.o_pt = t->o_pt;
      a.o_flag = t->o_flag;
      a.n_points_cap = c.max_points;
      a.seq = ++t->seq;
    }
    SVO_CHECK_HIP(ctx, hipMemcpyAsync(sh->args_dev, sh->args_host, (size_t)N * sizeof(TrkCamArgs), hipMemcpyHostToDevice, ctx->stream));
    const TrkCamArgs* ad = reinterpret_cast<const TrkCamArgs*>(sh->args_dev);
    hipLaunchKernelGGL(trk_plan_cams_kernel, dim3(N), dim3(TRK_THREADS), 0, ctx->stream, ad);
    SVO_CHECK_HIP(ctx, hipGetLastError());
    rc = svo_match_stages_cams(ctx, sh->kf_pyr, cur, &t0->cam, N, cap, sh->counters, 8, sh->cand_level_ref, recs, pwb_t, n_pad, c.n_pyr_levels,
                               c.align_max_iter, any_edgelet);
    if (rc != SVO_HIP_OK) return rc;
    hipLaunchKernelGGL(trk_replay_cams_kernel, dim3(N), dim3(TRK_THREADS), 0, ctx->stream, ad, c.max_fts, c.quality_min_fts);
    SVO_CHECK_HIP(ctx, hipGetLastError());
    const FrameState* st0 = svo_sia_state_dev(sh->sia, 0);
    static_assert(sizeof(FrameState) % sizeof(double) == 0, "the solver records are a whole number of doubles apart");
    rc = svo_pose_optimize_batch_strided(ctx, N, c.max_frame_features, sh->counters + 5, 8, st0->T_cur_w, (int)(sizeof(FrameState) / sizeof(double)),
                                         sh->ft_f, sh->ft_pos, sh->ft_level, sh->ft_has_point, fabs(t0->cam.fx), c.pose_optim_thresh,
                                         c.pose_optim_num_iter, sh->po);
    if (rc != SVO_HIP_OK) return rc;
    hipLaunchKernelGGL(trk_finish_cams_kernel, dim3(N), dim3(256), 0, ctx->stream, ad);
    SVO_CHECK_HIP(ctx, hipGetLastError());
  }
  // the one synchronisation of the frame: wait for every camera's sequence number (a spin on host memory: no driver call on the
  // way back), with the stream's own synchronisation as the fall-back and the error check
  {
    bool all = false;
    for (long spins = 0; spins < 4000000L && !all; ++spins) {                 // a few hundred milliseconds at most
      all = true;
      for (int k = 0; k < N && all; ++k) {
        svo_hip_tracker* t = sh->members[(size_t)k];
        volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(t->res_host + t->o_flag);
        all = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == t->seq;
      }
      if (!all) __builtin_ia32_pause();
    }
    if (!all) SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < N; ++k) {
      svo_hip_tracker* t = sh->members[(size_t)k];
      volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(t->res_host + t->o_flag);
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != t->seq) return svo_fail(ctx, SVO_HIP_ERR_DEVICE, who, "the frame's kernels did not complete");
    }
  }
  sh->last_idx = 1 - sh->last_idx;          // the new frames' pyramids are the next call's references
  for (int k = 0; k < N; ++k) {
    svo_hip_tracker* t = sh->members[(size_t)k];
    const svo_hip_track_result* r = reinterpret_cast<const svo_hip_track_result*>(t->res_host);
    t->last_n_host = r->n_features;
    t->last_from_track = true;
    t->need_gather = false;
    if (r->map_changed) t->rekey_pending = true;
  }
  return SVO_HIP_OK;
}

int svo_hip_tracker_track(svo_hip_tracker* t, const uint8_t* level0, svo_hip_track_result* result, double* feat_px, double* feat_f,
                          int32_t* feat_level, int32_t* feat_point, uint8_t* feat_edgelet, double* feat_grad, int32_t* pt_type,
                          int32_t* pt_n_failed, int32_t* pt_n_succeeded) {
  if (!t || !level0 || !result) return SVO_HIP_ERR_INVALID;
  if (t->sh->n_cams != 1)
    return svo_fail(t->ctx, SVO_HIP_ERR_STATE, "svo_hip_tracker_track", "a camera of a tracker group is tracked with its group (svo_hip_tracker_group_track)");
  const int rc = trk_track_all(t->sh, &level0, "svo_hip_tracker_track");
  if (rc != SVO_HIP_OK) return rc;
  trk_copy_out(t, result, feat_px, feat_f, feat_level, feat_point, feat_edgelet, feat_grad, pt_type, pt_n_failed, pt_n_succeeded);
  return SVO_HIP_OK;
}

// ---- a group of cameras: N trackers with one camera model and one configuration whose frames are tracked TOGETHER -- one
// chain of launches per call whatever N (every kernel of the chain takes one workgroup, or one slice of its grid, per camera).
// Every camera keeps its own map, last frame and result block and is set up with the svo_hip_tracker_* functions of its
// handle (svo_hip_tracker_group_camera); only the per-frame call is the group's.
struct svo_hip_tracker_group {
  svo_hip_tracker_shared* sh = nullptr;
};

int svo_hip_tracker_group_create(svo_hip_ctx* ctx, const svo_hip_camera* cam, const svo_hip_tracker_config* cfg, int n_cameras,
                                 svo_hip_tracker_group** out) {
  if (!ctx || !cam || !cfg || !out) return SVO_HIP_ERR_INVALID;
  *out = nullptr;
  SVO_REQUIRE(ctx, n_cameras >= 1 && n_cameras <= 1024);
  int rc = trk_check_config(ctx, cam, cfg);
  if (rc != SVO_HIP_OK) return rc;
  // a block of the batched warp / alignment stages takes 16 candidates and must not straddle two cameras
  SVO_REQUIRE(ctx, n_cameras == 1 || cfg->max_items % 16 == 0);
  svo_hip_tracker_group* g = new (std::nothrow) svo_hip_tracker_group();
  if (!g) return SVO_HIP_ERR_NOMEM;
  rc = trk_shared_create(ctx, cam, cfg, n_cameras, &g->sh);
  for (int k = 0; k < n_cameras && rc == SVO_HIP_OK; ++k) {
    svo_hip_tracker* t = nullptr;
    rc = trk_member_create(g->sh, k, cam, cfg, &t);
    if (rc == SVO_HIP_OK) g->sh->members.push_back(t);
  }
  if (rc != SVO_HIP_OK) { svo_hip_tracker_group_destroy(g); return rc; }
  *out = g;
  return SVO_HIP_OK;
}

int svo_hip_tracker_group_destroy(svo_hip_tracker_group* g) {
  if (!g) return SVO_HIP_ERR_INVALID;
  if (g->sh) {
    (void)hipStreamSynchronize(g->sh->ctx->stream);
    for (svo_hip_tracker* t : g->sh->members) trk_member_destroy(t);
    trk_shared_destroy(g->sh);
  }
  delete g;
  return SVO_HIP_OK;
}

int svo_hip_tracker_group_camera(svo_hip_tracker_group* g, int index, svo_hip_tracker** camera) {
  if (!g || !g->sh || !camera) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(g->sh->ctx, index >= 0 && index < g->sh->n_cams);
  *camera = g->sh->members[(size_t)index];
  return SVO_HIP_OK;
}

int svo_hip_tracker_group_track(svo_hip_tracker_group* g, const uint8_t* const* level0, svo_hip_track_result* results) {
  if (!g || !g->sh || !level0) return SVO_HIP_ERR_INVALID;
  for (int k = 0; k < g->sh->n_cams; ++k) if (!level0[k]) return SVO_HIP_ERR_INVALID;
  const int rc = trk_track_all(g->sh, level0, "svo_hip_tracker_group_track");
  if (rc != SVO_HIP_OK) return rc;
  if (results) for (int k = 0; k < g->sh->n_cams; ++k) trk_copy_out(g->sh->members[(size_t)k], results + k, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  return SVO_HIP_OK;
}

int svo_hip_tracker_last_result(svo_hip_tracker* t, svo_hip_track_result* result, double* feat_px, double* feat_f, int32_t* feat_level,
                                int32_t* feat_point, uint8_t* feat_edgelet, double* feat_grad, int32_t* pt_type, int32_t* pt_n_failed,
                                int32_t* pt_n_succeeded) {
  if (!t) return SVO_HIP_ERR_INVALID;
  if (!t->last_from_track) return svo_fail(t->ctx, SVO_HIP_ERR_STATE, "svo_hip_tracker_last_result", "no frame has been tracked since the last frame was set");
  trk_copy_out(t, result, feat_px, feat_f, feat_level, feat_point, feat_edgelet, feat_grad, pt_type, pt_n_failed, pt_n_succeeded);
  return SVO_HIP_OK;
}

int svo_hip_tracker_image_buffer(svo_hip_tracker* t, uint8_t** buffer) {
  if (!t || !buffer) return SVO_HIP_ERR_INVALID;
  *buffer = reinterpret_cast<uint8_t*>(t->img_host);
  return SVO_HIP_OK;
}

int svo_hip_tracker_download_key_points(svo_hip_tracker* t, int32_t* kf_key_point) {
  if (!t || !kf_key_point) return SVO_HIP_ERR_INVALID;
  svo_hip_ctx* ctx = t->ctx;
  if (!t->have_map) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_tracker_download_key_points", "no map has been set");
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  if (t->rekey_pending && t->n_kf > 0) {
    hipLaunchKernelGGL(trk_rekey_kernel, dim3(t->n_kf), dim3(256), 0, ctx->stream, make_map(t), t->kf_key_point, svo_make_cam(t->cam));
    SVO_CHECK_HIP(ctx, hipGetLastError());
  }
  t->rekey_pending = false;
  if (t->n_kf > 0) {
    SVO_CHECK_HIP(ctx, hipMemcpyAsync(kf_key_point, t->kf_key_point, (size_t)t->n_kf * 5 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SVO_HIP_OK;
}

int svo_hip_tracker_info(const svo_hip_tracker* t, int* n_cells, int* grid_cols, int* grid_rows, svo_hip_pyramid** keyframe_pyramids) {
  if (!t) return SVO_HIP_ERR_INVALID;
  if (n_cells) *n_cells = t->n_cells;
  if (grid_cols) *grid_cols = t->grid_cols;
  if (grid_rows) *grid_rows = t->grid_rows;
  if (keyframe_pyramids) *keyframe_pyramids = t->sh->kf_pyr;      // (a group member's keyframe slot s is slot cam_index * max_keyframes + s of it)
  return SVO_HIP_OK;
}

}  // extern "C"
