"""The per-frame data path of FrameHandlerMono::processFrame (S/frame_handler_mono.cpp:171-244) as a chain over a synthetic
sequence, with pluggable stages (HIP or oracle); shared by tests/test_gpu_sequence.py and tests/test_oracle_sequence.py.

    new_frame.T_f_w = last_frame.T_f_w                                   (:175)
    SparseImgAlign(kltMaxLevel=4, kltMinLevel, 30).run(last_frame, new)  (:186-188)   last frame's features = what the
                                                                                      reprojector created in it (:217-223)
    Reprojector::reprojectMap: map points of the keyframe into the grid, one match per cell (reprojector.cpp:149-241)
    pose_optimizer::optimizeGaussNewton on the matched features          (:226-229)
(optimizeStructure is left out: with one keyframe every point has a single observation, nothing to refine.)"""
import numpy as np

from android_svo_amd import synth

CELL = 30           # Config::gridSize()
MAX_FTS = 120       # Config::maxFts()


def make_sequence(n_frames=20, n_map=600):
    rng = np.random.default_rng(2024)
    cam = synth.Camera.default()
    scene = synth.PlaneScene(seed=77, depth=2.2, tilt=(0.06, -0.04))
    T0 = synth.se3_from_twist([0.01, -0.02, 0.0], [0.004, -0.003, 0.002])
    step_t, step_r = np.array([0.012, 0.004, -0.003]), np.array([0.0015, -0.0025, 0.002])
    truth = [T0]
    for _ in range(1, n_frames):
        wob = rng.uniform(-0.002, 0.002, 3)
        truth.append(synth.se3_mul(synth.se3_from_twist(step_t + wob, step_r + 0.2 * wob), truth[-1]))
    pyrs = [synth.build_pyramid(scene.render(cam, T)) for T in truth]
    px0 = synth.grid_features(cam, n_map, rng)           # the map: points seen in keyframe 0
    f0 = synth.cam2world(cam, px0)
    pos = scene.intersect(cam, T0, px0[:, 0], px0[:, 1])
    return dict(cam=cam, truth=truth, pyrs=pyrs, px0=px0, f0=f0, pos=pos, T0=T0)


def project(cam, T, pos):
    Xc = np.stack([synth.se3_act(T, p) for p in pos])
    return np.stack([cam.fx * Xc[:, 0] / Xc[:, 2] + cam.cx, cam.fy * Xc[:, 1] / Xc[:, 2] + cam.cy], axis=1)


def cells(cam, px_pred):
    """Reprojector::reprojectPoint: points whose projection is at least 8 px inside the image, bucketed by grid cell in
    map order (reprojector.cpp:246-259)."""
    gc, gr = -(-cam.width // CELL), -(-cam.height // CELL)
    pxi = px_pred.astype(np.int64)
    inside = (pxi[:, 0] >= 8) & (pxi[:, 0] < cam.width - 8) & (pxi[:, 1] >= 8) & (pxi[:, 1] < cam.height - 8)
    cell = (px_pred[:, 1] / CELL).astype(np.int64) * gc + (px_pred[:, 0] / CELL).astype(np.int64)
    idx = np.where(inside)[0]
    lists = [idx[cell[idx] == c] for c in range(gc * gr)]
    off = np.zeros(gc * gr + 1, dtype=np.int32)
    for c, r in enumerate(lists):
        off[c + 1] = off[c] + len(r)
    ids = np.concatenate(lists) if off[-1] else np.zeros(0, dtype=np.int64)
    return off, ids.astype(np.int64)


def run_chain(seq, stages, min_level, perturb_each_frame=0.0):
    """stages: object with align(fp, k, min_level) -> (T, n_tracked); reproject(k, T_sia, off, ids, px_pred) -> dict;
    refine(T_sia, f, pos, level, hp) -> (T, hp_out).  Returns per-frame poses, match counts and matched point ids."""
    cam, truth, pyrs, px0, f0, pos, T0 = (seq[k] for k in ("cam", "truth", "pyrs", "px0", "f0", "pos", "T0"))
    T = T0.copy()
    px, f, p3 = px0.copy(), f0.copy(), pos.copy()
    poses, n_matches, winners = [], [], []
    for k in range(1, len(truth)):
        n = len(px)
        fp = synth.FramePair(cam, pyrs[k - 1], pyrs[k], px, f, p3, np.ones(n, dtype=np.uint8), T, truth[k], T)   # init = last pose (:175)
        T_sia, n_tracked = stages.align(fp, k, min_level)
        assert n_tracked > 50
        px_pred = project(cam, T_sia, pos)
        off, ids = cells(cam, px_pred)
        res = stages.reproject(k, T_sia, off, ids, px_pred)
        win = res["cell_winner"][res["cell_winner"] >= 0]
        assert len(win) >= 50                                              # Config::qualityMinFts()
        px_m, sl = res["px_cur"][win], res["search_level"][win].astype(np.int32)
        f_m = synth.cam2world(cam, px_m)
        T_new, hp_out = stages.refine(T_sia, f_m, pos[ids[win]], sl, np.ones(len(win), dtype=np.uint8))
        keep = hp_out.astype(bool)
        assert keep.sum() >= 20                                            # sfba_n_edges_final (:231)
        T = np.array(T_new)
        T[0] += perturb_each_frame                                         # sensitivity probe: disturb the handed-over pose
        px, f, p3 = px_m[keep], f_m[keep], pos[ids[win]][keep]
        poses.append(T.copy())
        n_matches.append(int(res["n_matches"]))
        winners.append(ids[win][keep])
    return poses, n_matches, winners


class OracleStages:
    def __init__(self, seq):
        from oracle import orc
        self.orc, self.seq = orc, seq
        self.zeros = np.zeros(len(seq["px0"]), dtype=np.int32)

    def align(self, fp, k, min_level):
        o = self.orc.sparse_img_align(fp, max_level=4, min_level=min_level, n_iter=30, early_stop=True)
        return np.array(o.T_cur_w), o.n_tracked

    def reproject(self, k, T_sia, off, ids, px_pred):
        s = self.seq
        z = self.zeros[ids]
        return self.orc.reproject_cells(s["cam"], [s["pyrs"][0]], s["T0"][None, :], s["pyrs"][k], T_sia, off, z, s["px0"][ids], s["f0"][ids], z,
                                        s["pos"][ids], np.zeros(len(ids), np.uint8), np.tile([1.0, 0.0], (len(ids), 1)),
                                        np.zeros(len(ids), np.uint8), px_pred[ids], max_fts=MAX_FTS)

    def refine(self, T_sia, f, pos, level, hp):
        r, hp_out = self.orc.pose_optimize(abs(self.seq["cam"].fx), T_sia, f, pos, level, hp)
        return np.array(r.T_f_w), hp_out
