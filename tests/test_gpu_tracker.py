"""svo_hip_tracker_track: one frame of FrameHandlerMono::processFrame (S/frame_handler_mono.cpp:171-229) as one chain of
kernels on one stream -- SparseImgAlign against the last frame, Reprojector::reprojectMap with Map::getCloseKeyframes /
Point::getCloseViewObs / Matcher::findMatchDirect, pose_optimizer::optimizeGaussNewton, hand-over of the frame.

  * the reprojection stage against the reference's own compiled Reprojector::reprojectMap on a real svo::Map
    (tests/golden/reproject_map_ref.npz): every integer equal, pixels and gradients bitwise;
  * the 20-frame tracking sequence of tests/test_gpu_sequence.py through the new entry: equal matches in every cell and
    frame as the oracle composition AND as the stage-by-stage chain, poses within 1e-6 (north_star: 1e-4 rad / 1e-3 m)."""
import os

import numpy as np
import pytest

import tracking_chain as tc
from test_oracle_reproject_map import CASES, GOLD, check_map_result, rekey_expected
from android_svo_amd import hip, synth
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def _tracker_for(ctx, cs, key, **cfg):
    trk = hip.Tracker(ctx, cs["cam"], max_keyframes=max(cs["n_kf"], 1), grid_size=cs["cell_size"], **cfg)
    for k in range(cs["n_kf"]):
        trk.upload_keyframe(k, cs["kf_pyr"][k][0])
    trk.set_map(dict(cs, kf_slot=np.arange(cs["n_kf"], dtype=np.int32), kf_key_point=key))
    return trk


@pytest.mark.parametrize("tag,kw,max_fts", CASES, ids=[c[0] for c in CASES])
def test_reprojection_stage_against_reference_fixture(ctx, tag, kw, max_fts):
    g = np.load(GOLD)
    cs = synth.make_map_case(**kw)
    trk = _tracker_for(ctx, cs, g[tag + "_kf_key_point"], max_fts=max_fts, quality_min_fts=20)
    # a last frame without features: SparseImgAlign::run returns at once (:55-59) and the new frame keeps the pose it was
    # given -- exactly the pose the fixture was recorded at
    trk.set_last_frame(cs["T_cur_w"], np.zeros((0, 2)), np.zeros((0, 3)), np.zeros(0, np.int32), img=cs["cur_pyr"][0])
    r = trk.track(cs["cur_pyr"][0])
    np.testing.assert_array_equal(r["T_f_w_sia"], cs["T_cur_w"])
    assert r["result"].items_overflow == 0
    # the features as the reprojector made them: before the pose refinement drops observations
    po, hp = orc.pose_optimize(abs(cs["cam"].fx), cs["T_cur_w"], orc.cam2world(cs["cam"], g[tag + "_feat_px"]), cs["pt_pos"][g[tag + "_feat_point"]],
                               g[tag + "_feat_level"], np.ones(len(g[tag + "_feat_point"]), np.uint8))
    dropped = r["feat_point"] < 0
    np.testing.assert_array_equal(dropped, ~hp.astype(bool))
    res = dict(r, feat_point=np.where(dropped, g[tag + "_feat_point"], r["feat_point"]),
               unlinked=((r["type"] == synth.TYPE_DELETED) & (cs["pt_type"] != synth.TYPE_DELETED)).astype(np.uint8))
    check_map_result(g, tag, res)
    assert r["map_changed"] == 1                                            # every case deletes points
    np.testing.assert_array_equal(r["feat_f"], orc.cam2world(cs["cam"], g[tag + "_feat_px"]))      # Feature(frame, px, level): f = cam2world(px)
    # the pose refinement on those features, from the aligned pose
    rot, trans = synth.pose_error(r["T_f_w"], np.array(po.T_f_w))
    assert rot < 1e-9 and trans < 1e-9, (rot, trans)
    assert r["result"].pose.num_obs == po.num_obs
    # (what follows a frame with deletions: test_tracker_follows_deletions_on_the_device)
    trk.destroy()


@pytest.mark.parametrize("min_level", [2, 0], ids=["L4-L2_shipping_default", "L4-L0"])
def test_twenty_frame_sequence_through_the_tracker(ctx, min_level):
    seq = tc.make_sequence(n_frames=20)
    mp = tc.sequence_map(seq)
    cam = seq["cam"]
    # ---- HIP: the whole-frame entry
    trk = hip.Tracker(ctx, cam, max_keyframes=2, max_points=1024, max_obs=1024, max_kf_features=1024, max_candidates=16, max_items=1024,
                      max_frame_features=1024, grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=min_level)
    trk.upload_keyframe(0, seq["pyrs"][0][0])
    trk.set_map(mp)
    n = len(seq["px0"])
    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)

    def hip_track(k, last):
        r = trk.track(seq["pyrs"][k][0])
        assert r["map_changed"] == 0
        return r
    g_poses, g_n, g_win, g_px = tc.run_tracker_chain(seq, hip_track, min_level)
    trk.destroy()
    # ---- oracle composition of the same frame function
    state = {"pt_type": mp["pt_type"].copy(), "pt_n_failed": mp["pt_n_failed"].copy(), "pt_n_succeeded": mp["pt_n_succeeded"].copy(),
             "unlinked": np.zeros(n, np.uint8)}

    def orc_track(k, last):
        return tc.oracle_track_frame(orc, mp, state, last, seq["pyrs"][k - 1], seq["pyrs"][k], min_level)
    c_poses, c_n, c_win, c_px = tc.run_tracker_chain(seq, orc_track, min_level)
    # ---- the stage-by-stage oracle chain of tests/test_gpu_sequence.py (host-side projection and bucketing)
    s_poses, s_n, s_win = tc.run_chain(seq, tc.OracleStages(seq), min_level)
    truth = seq["truth"][1:]
    diff = np.array([synth.pose_error(a, b) for a, b in zip(g_poses, c_poses)])
    assert (diff[:, 0] < 1e-4).all() and (diff[:, 1] < 1e-3).all(), diff        # north_star tolerance at every frame
    for a, b in zip(g_win, c_win):                                              # every integer decision equal
        np.testing.assert_array_equal(a, b)
    assert g_n == c_n
    # the stage-by-stage chain has no point types: it agrees until the first points are promoted to TYPE_GOOD (more than ten
    # successful reprojections, reprojector.cpp:212-214) and move to the front of their cells
    for a, s in list(zip(c_win, s_win))[:10]:
        np.testing.assert_array_equal(np.sort(a), np.sort(s))
    assert c_n[:10] == s_n[:10]
    assert diff[:5].max() < 1e-12, diff[:5]
    assert diff.max() < 1e-6, diff.max(axis=0)
    for a, b in list(zip(c_poses, s_poses))[:10]:                               # ... and is the same computation until then
        rot, trans = synth.pose_error(a, b)
        assert rot < 1e-12 and trans < 1e-12
    err_gpu = np.array([synth.pose_error(a, t) for a, t in zip(g_poses, truth)])
    assert err_gpu[:, 0].max() < 2e-3 and err_gpu[:, 1].max() < 5e-3, err_gpu.max(axis=0)


def test_image_in_the_tracker_buffer_tracks_the_same(ctx):
    """svo_hip_tracker_image_buffer: frames written into the tracker's own page-locked buffer and passed as that very pointer
    (no copy inside svo_hip_tracker_track) give bit for bit the poses, matches and features of frames passed from caller memory"""
    seq = tc.make_sequence(n_frames=8)
    mp = tc.sequence_map(seq)
    n = len(seq["px0"])
    outs = []
    for through_buffer in (False, True):
        trk = hip.Tracker(ctx, seq["cam"], max_keyframes=2, max_points=1024, max_obs=1024, max_kf_features=1024, max_candidates=16,
                          max_items=1024, max_frame_features=1024, grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=2)
        trk.upload_keyframe(0, seq["pyrs"][0][0])
        trk.set_map(mp)
        trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
        buf = trk.image_buffer()
        assert buf.shape == (seq["cam"].height, seq["cam"].width) and buf.dtype == np.uint8
        frames = []
        for k in range(1, 8):
            if through_buffer:
                buf[:] = seq["pyrs"][k][0]
                r = trk.track(buf)
            else:
                buf[:] = 0                                   # whatever the buffer held is overwritten by the call's own copy
                r = trk.track(seq["pyrs"][k][0])
            frames.append((r["T_f_w"].tobytes(), r["n_matches"], r["feat_px"].tobytes(), r["feat_point"].tobytes()))
        outs.append(frames)
        trk.destroy()
    assert outs[0] == outs[1]
    assert all(f[1] >= 50 for f in outs[0])


def test_update_point_positions_equals_a_fresh_map(ctx):
    """FrameHandlerBase::optimizeStructure moves a few points between two frames: pushing the new positions with
    svo_hip_tracker_update_point_positions must give bit for bit what a fresh upload of the whole map gives (the last
    frame's reference features read the positions through the point table)."""
    seq = tc.make_sequence(n_frames=4)
    mp = tc.sequence_map(seq)
    n = len(seq["px0"])
    rng = np.random.default_rng(5)
    moved = np.sort(rng.choice(n, 20, replace=False)).astype(np.int32)
    pos2 = mp["pt_pos"].copy()
    pos2[moved] += rng.normal(0, 0.01, (len(moved), 3))
    outs = []
    for mode in ("fresh", "update"):
        trk = hip.Tracker(ctx, seq["cam"], max_keyframes=2, max_points=1024, max_obs=1024, max_kf_features=1024, max_candidates=16,
                          max_items=1024, max_frame_features=1024, grid_size=tc.CELL, max_fts=tc.MAX_FTS)
        trk.upload_keyframe(0, seq["pyrs"][0][0])
        trk.set_map(mp)
        trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
        r1 = trk.track(seq["pyrs"][1][0])
        if mode == "fresh":
            # the host's view after frame 1: counters advanced, positions moved -> a whole new upload, then the last frame again
            trk.set_map(dict(mp, pt_pos=pos2, pt_type=r1["type"], pt_n_failed=r1["n_failed"], pt_n_succeeded=r1["n_succeeded"]))
            trk.set_last_frame(r1["T_f_w"], r1["feat_px"], r1["feat_f"], r1["feat_point"], img=seq["pyrs"][1][0])
        else:
            trk.update_point_positions(moved, pos2[moved])
        r2 = trk.track(seq["pyrs"][2][0])
        outs.append(r2)
        trk.destroy()
    a, b = outs
    np.testing.assert_array_equal(a["T_f_w_sia"], b["T_f_w_sia"])
    np.testing.assert_array_equal(a["T_f_w"], b["T_f_w"])
    np.testing.assert_array_equal(a["feat_point"], b["feat_point"])
    assert a["feat_px"].tobytes() == b["feat_px"].tobytes()
    assert a["n_matches"] == b["n_matches"] and a["n_trials"] == b["n_trials"]


def _key_points_of(cs):
    """Frame::setKeyPoints of every keyframe of a map case, as point indices (synth.key_points == the reference's on all
    keyframes of the fixture cases)"""
    key = np.full((cs["n_kf"], 5), -1, np.int32)
    for k in range(cs["n_kf"]):
        o = cs["kf_ftr_obs"][cs["kf_ftr_offset"][k]:cs["kf_ftr_offset"][k + 1]]
        e = synth.key_points(cs["cam"], cs["obs_px"][o], np.ones(len(o), bool))
        key[k] = np.where(e >= 0, cs["obs_point"][o][np.maximum(e, 0)], -1)
    return key


@pytest.mark.parametrize("tag,kw,max_fts", [
    ("many_cells", dict(seed=41, cell_size=4), 1200),                            # 80 x 60 = 4800 cells: cell counters in global memory
    ("many_candidates", dict(seed=42, n_points=6000, n_candidates=400, cell_size=10), 1200),   # > 2048 candidates in the frame: sort keys in global memory
    ("both", dict(seed=43, n_points=6000, n_candidates=300, cell_size=5, edgelet_frac=0.1), 300),
    ("few", dict(seed=44, n_kf=3, n_points=120, n_candidates=10, cell_size=40), 1200),
    ("exact_capacity", dict(seed=45, n_kf=5, n_points=300, n_candidates=25, cell_size=30), 1200)])        # every table of the tracker exactly full
def test_reprojection_stage_beyond_and_below_the_lds_limits(ctx, tag, kw, max_fts):
    """The planning kernel keeps its cell counters and per-cell sort keys in LDS for grids of <= 2048 cells and frames of
    <= 2048 candidates, in global memory beyond: maps on either side of both limits against the oracle's
    Reprojector::reprojectMap (itself pinned by the reference's own compiled code, tests/test_oracle_reproject_map.py) --
    every integer equal, pixels and gradients bitwise."""
    cs = synth.make_map_case(**kw)
    key = _key_points_of(cs)
    ro = orc.reproject_map(cs, key, max_fts=max_fts)
    caps = {}
    if tag == "exact_capacity":
        caps = dict(max_points=cs["n_points"], max_obs=len(cs["obs_kf"]), max_kf_features=len(cs["kf_ftr_point"]), max_candidates=len(cs["cand_point"]))
    trk = _tracker_for(ctx, cs, key, max_fts=max_fts, quality_min_fts=20, max_frame_features=2816, **caps)
    n_cells = trk.n_cells
    trk.set_last_frame(cs["T_cur_w"], np.zeros((0, 2)), np.zeros((0, 3)), np.zeros(0, np.int32), img=cs["cur_pyr"][0])
    r = trk.track(cs["cur_pyr"][0])
    n_cand = int(r["result"].n_candidates)
    if tag in ("many_cells", "both"):
        assert n_cells > 2048
    if tag in ("many_candidates", "both"):
        assert n_cand > 2048
    if tag in ("few", "exact_capacity"):
        assert n_cells <= 2048 and n_cand <= 2048
    assert r["result"].items_overflow == 0
    assert [int(r["n_matches"]), int(r["n_trials"])] == [int(ro["n_matches"]), int(ro["n_trials"])]
    np.testing.assert_array_equal(r["overlap_kf"], ro["overlap_kf"])
    np.testing.assert_array_equal(r["overlap_count"], ro["overlap_count"])
    dropped = r["feat_point"] < 0                                                # by the pose refinement, after the reprojector
    np.testing.assert_array_equal(np.where(dropped, ro["feat_point"], r["feat_point"]), ro["feat_point"])
    np.testing.assert_array_equal(r["feat_level"], ro["feat_level"])
    np.testing.assert_array_equal(r["feat_type"], ro["feat_type"])
    assert np.asarray(r["feat_px"], dtype=np.float64).tobytes() == np.asarray(ro["feat_px"], dtype=np.float64).tobytes()
    assert np.asarray(r["feat_grad"], dtype=np.float64).tobytes() == np.asarray(ro["feat_grad"], dtype=np.float64).tobytes()
    for k in ("type", "n_failed", "n_succeeded"):
        np.testing.assert_array_equal(r[k], ro[k], err_msg=k)
    trk.destroy()


def test_last_frame_indices_are_checked_against_the_map(ctx):
    """The alignment reads pt_pos[point] of the last frame's features: svo_hip_tracker_set_last_frame refuses an index beyond the
    map, and a smaller map set afterwards makes the tracker ask for the last frame again instead of reading past its tables."""
    seq = tc.make_sequence(n_frames=3)
    mp = tc.sequence_map(seq)
    n = len(seq["px0"])
    trk = hip.Tracker(ctx, seq["cam"], max_keyframes=2, grid_size=tc.CELL, max_fts=tc.MAX_FTS, max_frame_features=1024)
    trk.upload_keyframe(0, seq["pyrs"][0][0])
    trk.set_map(mp)
    bad = np.arange(n, dtype=np.int32); bad[5] = n
    with pytest.raises(hip.SvoHipError):
        trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], bad, kf_slot=0)
    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
    r = trk.track(seq["pyrs"][1][0])
    assert r["n_matches"] > 50
    # a map with half the points: the tracked frame's features refer to points that no longer exist
    h = n // 12                                                                   # (the first frame matches points of the top rows: low indices)
    small = dict(mp, n_points=h, kf_ftr_offset=np.array([0, h], np.int32), kf_ftr_point=np.arange(h, dtype=np.int32), pt_pos=mp["pt_pos"][:h],
                 pt_type=mp["pt_type"][:h], pt_n_failed=mp["pt_n_failed"][:h], pt_n_succeeded=mp["pt_n_succeeded"][:h],
                 pt_obs_offset=np.arange(h + 1, dtype=np.int32), obs_kf=mp["obs_kf"][:h], obs_px=mp["obs_px"][:h], obs_f=mp["obs_f"][:h],
                 obs_level=mp["obs_level"][:h], obs_edgelet=mp["obs_edgelet"][:h], obs_grad=mp["obs_grad"][:h],
                 kf_key_point=synth.key_points(seq["cam"], mp["obs_px"][:h], np.ones(h, bool))[None, :])
    trk.set_map(small)
    with pytest.raises(hip.SvoHipError):
        trk.track(seq["pyrs"][2][0])                                              # "set_last_frame comes first"
    keep = r["feat_point"] < h
    trk.set_last_frame(r["T_f_w"], r["feat_px"][keep], r["feat_f"][keep], np.where(r["feat_point"][keep] >= 0, r["feat_point"][keep], -1), img=seq["pyrs"][1][0])
    r2 = trk.track(seq["pyrs"][2][0])
    assert r2["n_matches"] > 5
    trk.destroy()


def test_tracker_follows_deletions_on_the_device(ctx):
    """A frame that deletes map points (Map::safeDeletePoint / deleteCandidatePoint in the reprojector) does not need the map
    uploaded again: the points are unlinked in the device tables and the keyframes that lost a key feature choose their key
    features again with Frame::removeKeyPoint's rule (incumbents of untouched keyframes and slots stay, even non-optimal
    ones).  The next frame tracked straight on equals the next frame tracked after applying the deletions to the host
    tables and uploading them -- every integer, pixels bitwise."""
    tag, kw, max_fts = [c for c in CASES if c[0] == "rekey"][0]
    g = np.load(GOLD)
    cs = synth.make_map_case(**kw)
    # the "rekey" case of the fixture: key features with incumbents put in by hand (oracle/gen_golden.py: rekey_override) -- in some
    # keyframes a point the frame deletes, in others a live feature a fresh selection would not pick -- and what the reference's
    # own Map::safeDeletePoint -> Frame::removeKeyPoint made of them
    key = g[tag + "_kf_key_point"]
    scene = synth.PlaneScene(seed=kw.get("seed", 31), depth=2.0, tilt=(0.08, -0.05))
    T2 = synth.se3_mul(synth.se3_from_twist([0.012, -0.006, 0.004], [0.002, -0.003, 0.001]), cs["T_cur_w"])
    img2 = scene.render(cs["cam"], T2)

    def first_frame():
        trk = _tracker_for(ctx, cs, key, max_fts=max_fts, quality_min_fts=20)
        trk.set_last_frame(cs["T_cur_w"], np.zeros((0, 2)), np.zeros((0, 3)), np.zeros(0, np.int32), img=cs["cur_pyr"][0])
        return trk, trk.track(cs["cur_pyr"][0])
    # ---- A: straight on
    trk, r1 = first_frame()
    assert r1["map_changed"] == 1
    deleted = (r1["type"] == synth.TYPE_DELETED) & (cs["pt_type"] != synth.TYPE_DELETED)
    np.testing.assert_array_equal(deleted.astype(np.uint8), g[tag + "_unlinked"].astype(np.uint8))
    key_dev = trk.download_key_points(cs["n_kf"])
    np.testing.assert_array_equal(key_dev, g[tag + "_kf_key_point_after"])        # the reference's own result
    expect = rekey_expected(cs, key, deleted)
    np.testing.assert_array_equal(key_dev, expect)
    lost = np.array([deleted[key[k][key[k] >= 0]].any() for k in range(cs["n_kf"])])
    assert lost.sum() >= 2 and (~lost).sum() >= 2 and (expect[lost] != key[lost]).any()
    ra = trk.track(img2)
    trk.destroy()
    # ---- B: the deletions applied to the host tables, the map uploaded again, the last frame handed over by the host
    trk, r1b = first_frame()
    np.testing.assert_array_equal(r1b["T_f_w"], r1["T_f_w"])
    kfp = cs["kf_ftr_point"].copy()
    kfp[deleted[kfp]] = -1
    cs2 = dict(cs, pt_type=r1b["type"], pt_n_failed=r1b["n_failed"], pt_n_succeeded=r1b["n_succeeded"], kf_ftr_point=kfp,
               cand_point=np.array([p for p in cs["cand_point"] if not deleted[p]], np.int32))
    trk.set_map(dict(cs2, kf_slot=np.arange(cs["n_kf"], dtype=np.int32), kf_key_point=expect))
    trk.set_last_frame(r1b["T_f_w"], r1b["feat_px"], r1b["feat_f"], r1b["feat_point"], img=cs["cur_pyr"][0])
    rb = trk.track(img2)
    trk.destroy()
    assert ra["n_matches"] > 100
    for k in ("n_matches", "n_trials"):
        assert int(ra[k]) == int(rb[k]), k
    for k in ("overlap_kf", "overlap_count", "feat_point", "feat_level", "feat_type", "type", "n_failed", "n_succeeded"):
        np.testing.assert_array_equal(ra[k], rb[k], err_msg=k)
    assert ra["feat_px"].tobytes() == rb["feat_px"].tobytes() and ra["feat_grad"].tobytes() == rb["feat_grad"].tobytes()
    np.testing.assert_array_equal(ra["T_f_w"], rb["T_f_w"])
    rot, trans = synth.pose_error(ra["T_f_w"], T2)
    assert rot < 3e-3 and trans < 1e-2


def test_structure_optimisation_on_the_device_tables(ctx):
    """svo_hip_tracker_optimize_structure (FrameHandlerBase::optimizeStructure without the observations leaving the device):
    the positions equal Point::optimize over the same observations through svo_hip_point_optimize_batch (itself bit-identical
    to the reference's compiled point.cpp) bit for bit, and the next frame tracked after it equals the next frame tracked
    after svo_hip_tracker_update_point_positions with those positions -- poses and features bitwise."""
    tag, kw, max_fts = CASES[0]
    g = np.load(GOLD)
    cs = synth.make_map_case(**kw)
    key = g[tag + "_kf_key_point"]
    scene = synth.PlaneScene(seed=kw.get("seed", 31), depth=2.0, tilt=(0.08, -0.05))
    T2 = synth.se3_mul(synth.se3_from_twist([0.012, -0.006, 0.004], [0.002, -0.003, 0.001]), cs["T_cur_w"])
    img2 = scene.render(cs["cam"], T2)

    def first_frame():
        trk = _tracker_for(ctx, cs, key, max_fts=max_fts, quality_min_fts=20)
        trk.set_last_frame(cs["T_cur_w"], np.zeros((0, 2)), np.zeros((0, 3)), np.zeros(0, np.int32), img=cs["cur_pyr"][0])
        return trk, trk.track(cs["cur_pyr"][0])
    trk, r1 = first_frame()
    # the points of the new frame's features with at least two observations, the first twenty (Config::structureOptimMaxPts())
    n_obs = np.diff(cs["pt_obs_offset"])
    sel = [int(p) for p in r1["feat_point"] if p >= 0 and n_obs[p] >= 2][:20]
    assert len(sel) == 20
    pos_dev, it_dev = trk.optimize_structure(sel, n_iter=5)
    # the same through the batch entry point: observations in CSR form from the tables
    off, oT, of = [0], [], []
    for p in sel:
        for o in range(cs["pt_obs_offset"][p], cs["pt_obs_offset"][p + 1]):
            oT.append(cs["T_kf_w"][cs["obs_kf"][o]]); of.append(cs["obs_f"][o])
        off.append(len(oT))
    pos_ref, it_ref = hip.point_optimize_batch(ctx, cs["pt_pos"][sel], np.array(off, np.int32), np.array(oT), np.array(of), n_iter=5)
    assert pos_dev.tobytes() == pos_ref.tobytes()
    np.testing.assert_array_equal(it_dev, it_ref)
    assert np.abs(pos_dev - cs["pt_pos"][sel]).max() > 1e-6                        # the points moved (the map's points carry noise)
    with pytest.raises(hip.SvoHipError):
        trk.optimize_structure([sel[0], sel[0]])
    ra = trk.track(img2)
    trk.destroy()
    # ---- the reference path: positions pushed from the host
    trk, r1b = first_frame()
    trk.update_point_positions(sel, pos_ref)
    rb = trk.track(img2)
    trk.destroy()
    np.testing.assert_array_equal(ra["T_f_w_sia"], rb["T_f_w_sia"])
    np.testing.assert_array_equal(ra["T_f_w"], rb["T_f_w"])
    np.testing.assert_array_equal(ra["feat_point"], rb["feat_point"])
    assert ra["feat_px"].tobytes() == rb["feat_px"].tobytes()
    assert int(ra["n_matches"]) == int(rb["n_matches"]) > 100


def test_sequence_over_a_map_that_loses_points(ctx):
    """Six frames along a path over the 14-keyframe map (multi-observation points, point candidates, points close to their
    deletion thresholds): the reprojector deletes points in several of the frames.  The tracker is never given the map
    again -- it unlinks the points and re-selects key points on the device -- and is compared frame by frame with the oracle
    composition of processFrame, whose tables get Map::safeDeletePoint's key-point rule (rekey_expected) applied between the
    frames: every integer equal, poses within 1e-6 (north_star: 1e-4 rad / 1e-3 m)."""
    tag, kw, max_fts = [c for c in CASES if c[0] == "rekey"][0]
    g = np.load(GOLD)
    cs = synth.make_map_case(**kw)
    key = g[tag + "_kf_key_point"].copy()
    scene = synth.PlaneScene(seed=kw.get("seed", 31), depth=2.0, tilt=(0.08, -0.05))
    step = synth.se3_from_twist([0.011, -0.005, 0.003], [0.0015, -0.0025, 0.001])
    poses = [cs["T_cur_w"]]
    for _ in range(5):
        poses.append(synth.se3_mul(step, poses[-1]))
    pyrs = [cs["cur_pyr"]] + [synth.build_pyramid(scene.render(cs["cam"], T)) for T in poses[1:]]
    # ---- HIP: one set_map, then frames only
    trk = _tracker_for(ctx, cs, key, max_fts=max_fts, quality_min_fts=20)
    trk.set_last_frame(cs["T_cur_w"], np.zeros((0, 2)), np.zeros((0, 3)), np.zeros(0, np.int32), img=pyrs[0][0])
    hip_res = [trk.track(p[0]) for p in pyrs]
    trk.destroy()
    # ---- oracle composition with the key points advanced between the frames
    mp = dict(cs, kf_key_point=key.copy())
    state = {"pt_type": cs["pt_type"].astype(np.int32).copy(), "pt_n_failed": cs["pt_n_failed"].astype(np.int32).copy(),
             "pt_n_succeeded": cs["pt_n_succeeded"].astype(np.int32).copy(), "unlinked": np.zeros(cs["n_points"], np.uint8)}
    last = dict(T=cs["T_cur_w"].copy(), px=np.zeros((0, 2)), f=np.zeros((0, 3)), point=np.zeros(0, np.int32))
    frames_with_deletions = 0
    for k, (pyr, rh) in enumerate(zip(pyrs, hip_res)):
        before = state["unlinked"].copy()
        ro = tc.oracle_track_frame(orc, mp, state, last, pyrs[k - 1] if k else pyrs[0], pyr, 2, max_fts=max_fts, quality_min_fts=20)
        assert int(rh["n_matches"]) == int(ro["n_matches"]) and int(rh["n_trials"]) == int(ro["n_trials"]), k
        np.testing.assert_array_equal(rh["feat_point"], ro["feat_point"], err_msg="frame %d" % k)
        np.testing.assert_array_equal(rh["feat_level"], ro["feat_level"])
        for name in ("type", "n_failed", "n_succeeded"):
            np.testing.assert_array_equal(rh[name], ro[name], err_msg="frame %d %s" % (k, name))
        np.testing.assert_array_equal(rh["overlap_kf"], ro["overlap_kf"])
        rot, trans = synth.pose_error(rh["T_f_w"], ro["T_f_w"])
        assert rot < 1e-6 and trans < 1e-6, (k, rot, trans)
        assert int(rh["n_matches"]) > 100
        if (state["unlinked"] != before).any():
            frames_with_deletions += 1
            assert rh["map_changed"] == 1
            mp["kf_key_point"] = rekey_expected(cs, mp["kf_key_point"], state["unlinked"].astype(bool))
        last = dict(T=ro["T_f_w"].copy(), px=ro["feat_px"], f=ro["feat_f"], point=ro["feat_point"])
    assert frames_with_deletions >= 2


def test_last_frame_becomes_a_keyframe(ctx):
    """FrameHandlerMono::processFrame :284-330 -- new_frame_->setKeyframe(); point->addFrameRef(feature); map_.addKeyframe(new_frame_):
    svo_hip_tracker_keyframe_from_last_frame keeps the tracked frame's pyramid on the device as a keyframe, the host uploads the
    grown map (same point numbering: the last frame stays where it is), tracking goes on.  Equal, bit for bit, to a tracker
    that gets the new keyframe's image and the last frame from the host."""
    seq = tc.make_sequence(n_frames=9)
    mp = tc.sequence_map(seq)
    cam = seq["cam"]
    n = len(seq["px0"])
    cfg = dict(max_keyframes=4, grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=2, max_frame_features=1024)

    def start():
        trk = hip.Tracker(ctx, cam, **cfg)
        trk.upload_keyframe(0, seq["pyrs"][0][0])
        trk.set_map(mp)
        trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
        rs = [trk.track(seq["pyrs"][k][0]) for k in range(1, 6)]
        return trk, rs
    trk, rs = start()
    r5 = rs[-1]
    mp2 = tc.map_with_tracked_frame_as_keyframe(seq, mp, r5)
    # ---- A: the device keeps the frame
    trk.keyframe_from_last_frame(1)
    trk.set_map(mp2)
    ra = [trk.track(seq["pyrs"][k][0]) for k in range(6, 9)]
    trk.destroy()
    # ---- B: everything from the host
    trk, rs_b = start()
    np.testing.assert_array_equal(rs_b[-1]["T_f_w"], r5["T_f_w"])
    trk.upload_keyframe(1, seq["pyrs"][5][0])
    trk.set_map(mp2)
    trk.set_last_frame(r5["T_f_w"], r5["feat_px"], r5["feat_f"], r5["feat_point"], img=seq["pyrs"][5][0])
    rb = [trk.track(seq["pyrs"][k][0]) for k in range(6, 9)]
    trk.destroy()
    used_new_kf = False
    for a, b in zip(ra, rb):
        np.testing.assert_array_equal(a["T_f_w"], b["T_f_w"])
        np.testing.assert_array_equal(a["feat_point"], b["feat_point"])
        assert a["feat_px"].tobytes() == b["feat_px"].tobytes()
        assert int(a["n_matches"]) == int(b["n_matches"]) > 50
        used_new_kf = used_new_kf or 1 in list(a["overlap_kf"])
    assert used_new_kf                                                           # the new keyframe took part in the reprojection
    err = np.array([synth.pose_error(a["T_f_w"], t) for a, t in zip(ra, seq["truth"][6:9])])
    assert err[:, 0].max() < 3e-3 and err[:, 1].max() < 1e-2


@pytest.mark.parametrize("seed", list(range(60, 72)))
def test_reprojection_stage_random_maps(ctx, seed):
    """A dozen random maps (number of keyframes, points, candidates, cell size, keyframe spacing, share of edgelets, maxFts cap
    all drawn from the seed) through the tracker's reprojection stage against the oracle's Reprojector::reprojectMap: every
    integer equal, pixels and gradients bitwise, key points after the frame by the reference's rule."""
    rng = np.random.default_rng(seed)
    kw = dict(seed=seed, n_kf=int(rng.integers(3, 16)), n_points=int(rng.integers(150, 2500)), n_candidates=int(rng.integers(0, 250)),
              cell_size=int(rng.choice([8, 12, 20, 25, 30, 40])), edgelet_frac=float(rng.choice([0.0, 0.04, 0.3])),
              kf_step=float(rng.choice([0.05, 0.16, 0.4])))
    max_fts = int(rng.choice([25, 120, 1200]))
    cs = synth.make_map_case(**kw)
    key = _key_points_of(cs)
    ro = orc.reproject_map(cs, key, max_fts=max_fts)
    trk = _tracker_for(ctx, cs, key, max_fts=max_fts, quality_min_fts=10, max_frame_features=2816)
    trk.set_last_frame(cs["T_cur_w"], np.zeros((0, 2)), np.zeros((0, 3)), np.zeros(0, np.int32), img=cs["cur_pyr"][0])
    r = trk.track(cs["cur_pyr"][0])
    assert r["result"].items_overflow == 0
    assert [int(r["n_matches"]), int(r["n_trials"])] == [int(ro["n_matches"]), int(ro["n_trials"])], kw
    np.testing.assert_array_equal(r["overlap_kf"], ro["overlap_kf"])
    np.testing.assert_array_equal(r["overlap_count"], ro["overlap_count"])
    dropped = r["feat_point"] < 0
    np.testing.assert_array_equal(np.where(dropped, ro["feat_point"], r["feat_point"]), ro["feat_point"])
    np.testing.assert_array_equal(r["feat_level"], ro["feat_level"])
    np.testing.assert_array_equal(r["feat_type"], ro["feat_type"])
    assert np.asarray(r["feat_px"], dtype=np.float64).tobytes() == np.asarray(ro["feat_px"], dtype=np.float64).tobytes()
    assert np.asarray(r["feat_grad"], dtype=np.float64).tobytes() == np.asarray(ro["feat_grad"], dtype=np.float64).tobytes()
    for k in ("type", "n_failed", "n_succeeded"):
        np.testing.assert_array_equal(r[k], ro[k], err_msg=k)
    deleted = (r["type"] == synth.TYPE_DELETED) & (cs["pt_type"] != synth.TYPE_DELETED)
    np.testing.assert_array_equal(trk.download_key_points(cs["n_kf"]), rekey_expected(cs, key, deleted))
    trk.destroy()


def test_tracker_edge_cases(ctx):
    """What a tracked frame does at the edges: a map without keyframes, too few matches for the pose refinement
    (frame_handler_mono.cpp:208-215: the new frame gets the last frame's pose, tracking goes on from it), and more candidates
    than max_items (reported, nothing written out of bounds)."""
    seq = tc.make_sequence(n_frames=4)
    mp = tc.sequence_map(seq)
    cam = seq["cam"]
    n = len(seq["px0"])
    # ---- an empty map: nothing to reproject, the pose stays the last frame's
    trk = hip.Tracker(ctx, cam, max_keyframes=2, grid_size=tc.CELL, max_fts=tc.MAX_FTS, max_frame_features=1024)
    empty = dict(mp, n_kf=0, n_points=0, kf_slot=np.zeros(0, np.int32), T_kf_w=np.zeros((0, 7)), kf_key_point=np.zeros((0, 5), np.int32),
                 kf_ftr_offset=np.zeros(1, np.int32), kf_ftr_point=np.zeros(0, np.int32), pt_pos=np.zeros((0, 3)), pt_type=np.zeros(0, np.int32),
                 pt_n_failed=np.zeros(0, np.int32), pt_n_succeeded=np.zeros(0, np.int32), pt_obs_offset=np.zeros(1, np.int32),
                 obs_kf=np.zeros(0, np.int32), obs_px=np.zeros((0, 2)), obs_f=np.zeros((0, 3)), obs_level=np.zeros(0, np.int32),
                 obs_edgelet=np.zeros(0, np.uint8), obs_grad=np.zeros((0, 2)), cand_point=np.zeros(0, np.int32))
    trk.set_map(empty)
    trk.set_last_frame(seq["T0"], np.zeros((0, 2)), np.zeros((0, 3)), np.zeros(0, np.int32), img=seq["pyrs"][0][0])
    r = trk.track(seq["pyrs"][1][0])
    assert int(r["n_matches"]) == 0 and len(r["feat_px"]) == 0 and r["result"].pose.ran == 0 and r["map_changed"] == 0
    np.testing.assert_array_equal(r["T_f_w"], seq["T0"])
    trk.destroy()
    # ---- too few matches: the reference's failure branch, frame by frame equal to the oracle composition
    trk = hip.Tracker(ctx, cam, max_keyframes=2, grid_size=tc.CELL, max_fts=tc.MAX_FTS, max_frame_features=1024, quality_min_fts=100000, klt_min_level=2)
    trk.upload_keyframe(0, seq["pyrs"][0][0])
    trk.set_map(mp)
    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
    state = {"pt_type": mp["pt_type"].copy(), "pt_n_failed": mp["pt_n_failed"].copy(), "pt_n_succeeded": mp["pt_n_succeeded"].copy(),
             "unlinked": np.zeros(n, np.uint8)}
    last = dict(T=seq["T0"].copy(), px=seq["px0"].copy(), f=seq["f0"].copy(), point=np.arange(n, dtype=np.int32))
    for k in (1, 2, 3):
        r = trk.track(seq["pyrs"][k][0])
        ro = tc.oracle_track_frame(orc, mp, state, last, seq["pyrs"][k - 1], seq["pyrs"][k], 2, quality_min_fts=100000)
        assert r["result"].pose.ran == 0 and int(r["n_matches"]) == int(ro["n_matches"]) > 50
        np.testing.assert_array_equal(r["T_f_w"], last["T"])                      # new_frame_->T_f_w_ = last_frame_->T_f_w_ (:211)
        np.testing.assert_array_equal(r["feat_point"], ro["feat_point"])          # the reprojector's features stay on the frame
        rot, trans = synth.pose_error(r["T_f_w_sia"], ro["T_f_w_sia"])
        assert rot < 1e-9 and trans < 1e-9
        last = dict(T=ro["T_f_w"].copy(), px=ro["feat_px"], f=ro["feat_f"], point=ro["feat_point"])
    trk.destroy()
    # ---- more candidates than max_items: flagged; what is returned is still a consistent frame
    trk = hip.Tracker(ctx, cam, max_keyframes=2, grid_size=tc.CELL, max_fts=tc.MAX_FTS, max_frame_features=1024, max_items=64)
    trk.upload_keyframe(0, seq["pyrs"][0][0])
    trk.set_map(mp)
    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
    r = trk.track(seq["pyrs"][1][0])
    assert r["result"].items_overflow == 1 and int(r["result"].n_candidates) == 64
    assert 0 < int(r["n_matches"]) <= 64 and (r["feat_point"][r["feat_point"] >= 0] < n).all()
    trk.destroy()


def test_tracker_refuses_bad_arguments(ctx):
    """Every index the kernels follow is checked on the host: bad arguments come back as errors, nothing is launched."""
    seq = tc.make_sequence(n_frames=3)
    mp = tc.sequence_map(seq)
    cam = seq["cam"]
    n = len(seq["px0"])
    for bad in (dict(max_keyframes=0), dict(max_keyframes=257), dict(max_frame_features=5000), dict(reproj_max_n_kfs=17), dict(klt_max_level=7),
                dict(max_fts=1200, max_frame_features=1200)):             # a frame can gain max_fts + 1 features (reprojector.cpp:164-165)
        with pytest.raises(hip.SvoHipError):
            hip.Tracker(ctx, cam, **bad)
    trk = hip.Tracker(ctx, cam, max_keyframes=2, max_points=n, grid_size=tc.CELL, max_fts=tc.MAX_FTS, max_frame_features=256)
    with pytest.raises(hip.SvoHipError):
        trk.track(seq["pyrs"][1][0])                                              # no map, no last frame
    with pytest.raises(hip.SvoHipError):
        trk.optimize_structure([0])                                               # no map
    for key, val in (("kf_ftr_point", np.full(n, n, np.int32)), ("kf_key_point", np.full((1, 5), n, np.int32)), ("obs_kf", np.ones(n, np.int32)),
                     ("obs_level", np.full(n, 9, np.int32)), ("pt_type", np.full(n, 4, np.int32)), ("kf_slot", np.array([2], np.int32))):
        with pytest.raises(hip.SvoHipError):
            trk.set_map(dict(mp, **{key: val}))
    with pytest.raises(hip.SvoHipError):                                          # more points than the tracker was created for
        trk.set_map(dict(mp, n_points=n + 1, pt_pos=np.zeros((n + 1, 3)), pt_type=np.full(n + 1, 2, np.int32), pt_n_failed=np.zeros(n + 1, np.int32),
                         pt_n_succeeded=np.zeros(n + 1, np.int32), pt_obs_offset=np.concatenate([mp["pt_obs_offset"], [n]]).astype(np.int32)))
    trk.upload_keyframe(0, seq["pyrs"][0][0])
    trk.set_map(mp)
    with pytest.raises(hip.SvoHipError):
        trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)      # 600 features > max_frame_features = 256
    with pytest.raises(hip.SvoHipError):
        trk.set_last_frame(seq["T0"], seq["px0"][:10], seq["f0"][:10], np.arange(10, dtype=np.int32), kf_slot=5)
    with pytest.raises(hip.SvoHipError):
        trk.optimize_structure(list(range(65)))
    with pytest.raises(hip.SvoHipError):
        trk.optimize_structure([n])
    with pytest.raises(hip.SvoHipError):
        trk.update_point_positions([n], np.zeros((1, 3)))
    # ... and the object still works
    trk.set_last_frame(seq["T0"], seq["px0"][:200], seq["f0"][:200], np.arange(200, dtype=np.int32), kf_slot=0)
    r = trk.track(seq["pyrs"][1][0])
    assert int(r["n_matches"]) > 50
    # a refused map changes nothing: this one has fewer points than the tracked frame refers to (accepted, it would make the
    # tracker forget its last frame) and one bad point type
    m = 50
    o_end = int(mp["pt_obs_offset"][m])
    keep = mp["kf_ftr_point"][mp["kf_ftr_point"] < m]
    small = dict(mp, n_points=m, pt_pos=mp["pt_pos"][:m], pt_type=np.concatenate([[4], mp["pt_type"][1:m]]).astype(np.int32),
                 pt_n_failed=mp["pt_n_failed"][:m], pt_n_succeeded=mp["pt_n_succeeded"][:m], pt_obs_offset=mp["pt_obs_offset"][:m + 1],
                 obs_kf=mp["obs_kf"][:o_end], obs_px=mp["obs_px"][:o_end], obs_f=mp["obs_f"][:o_end], obs_level=mp["obs_level"][:o_end],
                 kf_ftr_point=keep, kf_ftr_offset=np.array([0, len(keep)], np.int32), kf_key_point=np.full((1, 5), -1, np.int32))
    for k in ("obs_edgelet", "obs_grad"):
        if small.get(k) is not None:
            small[k] = small[k][:o_end]
    with pytest.raises(hip.SvoHipError):
        trk.set_map(small)
    r2 = trk.track(seq["pyrs"][2][0])                                             # no svo_hip_tracker_set_last_frame needed
    assert int(r2["n_matches"]) > 50
    trk.destroy()
