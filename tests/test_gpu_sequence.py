"""Sequence-level tracking parity (north_star: "poses within tolerance ... on identical input SEQUENCES").

A 20-frame synthetic trajectory over a textured plane is tracked frame to frame the way FrameHandlerMono::processFrame
does (S/frame_handler_mono.cpp:171-244), once through the HIP path and once through the CPU oracle, each chain feeding on
ITS OWN previous outputs:
    new_frame.T_f_w = last_frame.T_f_w                                   (:175)
    SparseImgAlign(kltMaxLevel=4, kltMinLevel, 30).run(last_frame, new)  (:186-188)   last frame's features = what the
                                                                                      reprojector created in it (:217-223)
    Reprojector::reprojectMap: map points of the keyframe into the grid, one match per cell (reprojector.cpp:149-241)
    pose_optimizer::optimizeGaussNewton on the matched features          (:226-229)
(optimizeStructure is left out: with one keyframe every point has a single observation, nothing to refine.)
Asserted per frame: HIP pose vs oracle pose < 1e-4 rad / 1e-3 m (north_star) -- in fact orders of magnitude tighter, since
every integer decision along the chain is equal -- the same matches in the same cells, no growth of the difference along
the sequence, and both chains stay at the sub-pixel level of the ground truth (no drift)."""
import numpy as np
import pytest

from android_svo_amd import hip, synth
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


CELL = 30           # Config::gridSize()
N_FRAMES = 20


def _project(cam, T, pos):
    Xc = np.stack([synth.se3_act(T, p) for p in pos])
    return np.stack([cam.fx * Xc[:, 0] / Xc[:, 2] + cam.cx, cam.fy * Xc[:, 1] / Xc[:, 2] + cam.cy], axis=1)


def _cells(cam, px_pred):
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


class _Chain:
    """state handed from frame to frame: the last frame's pose and features (pixel, bearing, map point, level)"""

    def __init__(self, T0, px, f, pos):
        self.T = T0.copy()
        self.px, self.f, self.pos = px.copy(), f.copy(), pos.copy()
        self.level = np.zeros(len(px), dtype=np.int32)
        self.poses, self.n_matches, self.winners = [], [], []


@pytest.mark.parametrize("min_level", [2, 0], ids=["L4-L2_shipping_default", "L4-L0"])
def test_twenty_frame_tracking_chain(ctx, min_level):
    rng = np.random.default_rng(2024)
    cam = synth.Camera.default()
    scene = synth.PlaneScene(seed=77, depth=2.2, tilt=(0.06, -0.04))
    T0 = synth.se3_from_twist([0.01, -0.02, 0.0], [0.004, -0.003, 0.002])
    step_t, step_r = np.array([0.012, 0.004, -0.003]), np.array([0.0015, -0.0025, 0.002])
    truth = [T0]
    for k in range(1, N_FRAMES):
        wob = rng.uniform(-0.002, 0.002, 3)
        truth.append(synth.se3_mul(synth.se3_from_twist(step_t + wob, step_r + 0.2 * wob), truth[-1]))
    pyrs = [synth.build_pyramid(scene.render(cam, T)) for T in truth]
    # the map: 600 points seen in keyframe 0
    px0 = synth.grid_features(cam, 600, rng)
    f0 = synth.cam2world(cam, px0)
    pos = scene.intersect(cam, T0, px0[:, 0], px0[:, 1])
    n_map = len(px0)

    kf = hip.Pyramid(ctx, cam.width, cam.height, 5, 1)
    kf.upload(0, pyrs[0])
    ref = hip.Pyramid(ctx, cam.width, cam.height, 5, 1)
    cur = hip.Pyramid(ctx, cam.width, cam.height, 5, 1)
    sia = hip.SparseImgAlign(ctx, 1, 2048)
    sia.set_frames(ref, cur)
    prm = sia.params(max_level=4, min_level=min_level, n_iter=30, eps=1e-6, early_stop=True)
    zeros_i = np.zeros(n_map, dtype=np.int32)

    gpu, cpu = _Chain(T0, px0, f0, pos), _Chain(T0, px0, f0, pos)
    for k in range(1, N_FRAMES):
        for chain, on_gpu in ((gpu, True), (cpu, False)):
            n = len(chain.px)
            fp = synth.FramePair(cam, pyrs[k - 1], pyrs[k], chain.px, chain.f, chain.pos, np.ones(n, dtype=np.uint8), chain.T,
                                 truth[k], chain.T)                       # initial pose = the last frame's (:175)
            # ---- sparse image alignment against the last frame
            if on_gpu:
                ref.upload(0, pyrs[k - 1]); cur.upload(0, pyrs[k])
                sia.upload_pair(0, fp)
                sia.run(1, prm)
                r = sia.download(0)
                T_sia, n_tracked = np.array(r.T_cur_w), r.n_tracked
            else:
                o = orc.sparse_img_align(fp, max_level=4, min_level=min_level, n_iter=30, early_stop=True)
                T_sia, n_tracked = np.array(o.T_cur_w), o.n_tracked
            assert n_tracked > 50
            # ---- reprojection of the map (keyframe 0's points) into the new frame, one match per cell
            px_pred = _project(cam, T_sia, pos)
            off, ids = _cells(cam, px_pred)
            deleted = np.zeros(len(ids), dtype=np.uint8)
            if on_gpu:
                res = hip.reproject_cells(ctx, kf, cur, 0, cam, T0[None, :], T_sia, off, zeros_i[ids], px0[ids], f0[ids], zeros_i[ids],
                                          pos[ids], deleted, px_pred[ids], max_fts=120)
            else:
                res = orc.reproject_cells(cam, [pyrs[0]], T0[None, :], pyrs[k], T_sia, off, zeros_i[ids], px0[ids], f0[ids], zeros_i[ids],
                                          pos[ids], np.zeros(len(ids), np.uint8), np.tile([1.0, 0.0], (len(ids), 1)), deleted, px_pred[ids],
                                          max_fts=120)
            win = res["cell_winner"][res["cell_winner"] >= 0]
            assert len(win) >= 50                                          # Config::qualityMinFts()
            px_m, sl = res["px_cur"][win], res["search_level"][win].astype(np.int32)
            f_m = synth.cam2world(cam, px_m)
            # ---- motion-only refinement on the matched features
            hp = np.ones(len(win), dtype=np.uint8)
            if on_gpu:
                pr, hp_out = hip.pose_optimize(ctx, T_sia, f_m, pos[ids[win]], sl, hp, abs(cam.fx))
            else:
                pr, hp_out = orc.pose_optimize(abs(cam.fx), T_sia, f_m, pos[ids[win]], sl, hp)
            keep = hp_out.astype(bool)
            assert keep.sum() >= 20                                        # sfba_n_edges_final (:231)
            # ---- hand-over: the new frame becomes the last frame; its features are the surviving matches
            chain.T = np.array(pr.T_f_w)
            chain.px, chain.f, chain.pos, chain.level = px_m[keep], f_m[keep], pos[ids[win]][keep], sl[keep]
            chain.poses.append(chain.T.copy())
            chain.n_matches.append(int(res["n_matches"]))
            chain.winners.append(ids[win][keep])

    diff = np.array([synth.pose_error(a, b) for a, b in zip(gpu.poses, cpu.poses)])
    err_gpu = np.array([synth.pose_error(a, t) for a, t in zip(gpu.poses, truth[1:])])
    err_cpu = np.array([synth.pose_error(a, t) for a, t in zip(cpu.poses, truth[1:])])
    # north_star tolerance at every frame of the sequence ...
    assert (diff[:, 0] < 1e-4).all() and (diff[:, 1] < 1e-3).all(), diff
    # ... and what the identical decisions actually give: the same cells matched with the same points in every frame,
    # poses equal to rounding noise, with no growth along the sequence
    for a, b in zip(gpu.winners, cpu.winners):
        np.testing.assert_array_equal(a, b)
    assert gpu.n_matches == cpu.n_matches
    assert diff[:, 0].max() < 1e-9 and diff[:, 1].max() < 1e-9, diff.max(axis=0)
    assert diff[-5:].max() <= 10 * max(diff[:5].max(), 1e-13)
    # both chains track the ground truth at the sub-pixel level and do not drift (they re-anchor on the keyframe's map)
    assert err_gpu[:, 0].max() < 2e-3 and err_gpu[:, 1].max() < 5e-3, err_gpu.max(axis=0)      # ~0.5 px at 2.2 m depth
    assert err_gpu[-5:, 1].mean() < 3 * err_gpu[:5, 1].mean() + 1e-4
    np.testing.assert_allclose(err_gpu, err_cpu, atol=1e-9)
    for d in (sia, ref, cur, kf):
        d.destroy()
