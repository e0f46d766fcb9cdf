"""The C++ host layer (android_svo_amd/host/svo_host.h: svo::SparseImgAlign, svo::DepthFilter with the
reference's names and protocol) driven end to end on the GPU by android_svo_amd/host/svo_host_demo, and
checked against an oracle replay of the same sequence: a frame-to-frame alignment, the depth-filter protocol
over several frames (keyframe hand-off, per-frame updates, convergence callback, seed removal), run once
synchronously and once with the worker thread while the main thread keeps aligning (two contexts)."""
import os
import subprocess

import numpy as np
import pytest

from android_svo_amd import seedsynth, synth
from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "android_svo_amd", "host", "svo_host_demo")


def _write(path, arr, dtype):
    np.ascontiguousarray(arr, dtype=dtype).tofile(path)


def _bearings(cam, px):
    """Feature::f as the reference forms it: cam->cam2world(px) -- for a radtan camera the cv::undistortPoints route
    (five float iterations; the oracle's restatement of it, parity unpinned) instead of the closed form"""
    if getattr(cam, "dist", None) is None:
        return synth.cam2world(cam, px)
    import ctypes
    c = orc.camera(cam)
    f = np.zeros((len(px), 3))
    for i in range(len(px)):
        orc.lib().svo_orc_cam2world(ctypes.byref(c), ctypes.c_double(px[i, 0]), ctypes.c_double(px[i, 1]), orc._p(f[i], ctypes.c_double))
    return f


@pytest.mark.parametrize("dist", [None, (-0.12, 0.03, 2e-4, -1e-4, 0.0)], ids=["pinhole", "radtan"])
def test_cpp_host_layer_end_to_end(tmp_path, dist):
    """SparseImgAlign + the DepthFilter protocol through the C++ host layer.  `radtan`: a distorted camera end to end --
    images rendered through the distortion, Feature::f by the reference's cv::undistortPoints route, world2cam by the
    pinned forward model -- the same oracle replay must hold (the depth filter's cam2world branch that no reference
    binary pins, run through the C++ layer and not only through the kernel)."""
    assert os.path.exists(DEMO), "build() must have produced android_svo_amd/host/svo_host_demo"
    case, out = tmp_path / "case", tmp_path / "out"
    case.mkdir(); out.mkdir()
    rng = np.random.default_rng(4)
    cam = synth.Camera.default()
    if dist is not None:
        cam.dist = dist
    scene = synth.PlaneScene(seed=4, depth=2.0, tilt=(0.08, 0.05))
    n_frames = 7
    T0 = synth.se3_from_twist([0.02, -0.01, 0.0], [0.01, 0.005, -0.01])
    direction = np.array([1.0, 0.3, 0.1]) / np.linalg.norm([1.0, 0.3, 0.1])
    poses = [T0] + [synth.se3_mul(synth.se3_from_twist(direction * 0.035 * k, rng.uniform(-0.004, 0.004, 3)), T0)
                    for k in range(1, n_frames)]
    pyrs = [synth.build_pyramid(scene.render(cam, T)) for T in poses]
    _write(case / "manifest.bin", [cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, 5, n_frames] + (list(dist) if dist is not None else []),
           np.float64)
    for k in range(n_frames):
        _write(case / ("frame_%d_pose.bin" % k), poses[k], np.float64)
        for l in range(5):
            _write(case / ("frame_%d_L%d.bin" % (k, l)), pyrs[k][l], np.uint8)
    # alignment features on frame 0
    px = synth.grid_features(cam, 800, rng)
    f = _bearings(cam, px)
    pos = scene.intersect(cam, T0, px[:, 0], px[:, 1])
    has = np.ones(len(px), dtype=np.uint8); has[::9] = 0
    for name, arr, dt in (("sia_px", px, np.float64), ("sia_f", f, np.float64), ("sia_pos", pos, np.float64), ("sia_has", has, np.uint8)):
        _write(case / (name + ".bin"), arr, dt)
    # seed batch A on keyframe 0, seed batch B on the second keyframe (frame kf2)
    kf2 = 4
    zbar = None
    sets = []
    for (kf, n_seeds, tag) in ((0, 3000, "seed"), (kf2, 1500, "seedB")):
        spx = np.floor(np.stack([rng.uniform(40, cam.width - 40, n_seeds), rng.uniform(40, cam.height - 40, n_seeds)], axis=1))
        slevel = rng.choice([0, 0, 1, 2], n_seeds).astype(np.int32)
        spx -= spx % (1 << slevel)[:, None]
        sf = _bearings(cam, spx)
        X = scene.intersect(cam, poses[kf], spx[:, 0], spx[:, 1])
        true_depth = np.linalg.norm(X - synth.se3_inv(poses[kf])[:3], axis=1)
        if zbar is None:
            zbar = float(np.median(true_depth))
        _write(case / (tag + "_px.bin"), spx, np.float64); _write(case / (tag + "_f.bin"), sf, np.float64)
        _write(case / (tag + "_level.bin"), slevel, np.int32)
        sets.append((kf, spx, sf, slevel))
    _write(case / "depth_mean_min.bin", [1.1 * zbar, 0.5 * zbar], np.float64)
    _write(case / "second_keyframe.bin", [kf2], np.float64)

    r = subprocess.run([DEMO, str(case), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr

    # ---- SparseImgAlign::run against the oracle
    fp = synth.FramePair(cam, pyrs[0], pyrs[1], px, f, pos, has, poses[0], poses[1], poses[0])
    o = orc.sparse_img_align(fp)
    sia = np.fromfile(out / "sia.bin")
    rot, trans = synth.pose_error(sia[:7], np.array(o.T_cur_w))
    assert rot < 1e-4 and trans < 1e-3
    assert int(sia[7]) == o.n_tracked
    np.testing.assert_allclose(sia[8:44], np.array(o.H) / (5e-4 * 255 * 255), rtol=1e-6, atol=1e-6)
    # ... and with method_ = LevenbergMarquardt and setRobustCostFunction(MADScale, HuberWeight) on the C++ object
    olm = orc.sparse_img_align(fp, method=1, scale_estimator=2, weight_function=3)
    lm = np.fromfile(out / "sia_lm.bin")
    rot, trans = synth.pose_error(lm[:7], np.array(olm.T_cur_w))
    assert rot < 1e-8 and trans < 1e-8, (rot, trans)
    assert int(lm[7]) == olm.n_tracked and abs(lm[8] - olm.chi2) <= 2.5e-7 * olm.chi2 and np.float32(lm[9]) == np.float32(olm.scale)
    assert int(lm[12]) == olm.stop

    # ---- DepthFilter protocol: oracle replay of the reference's loop.  The seed list is [batch A..., batch B...]; every
    # frame walks it in list order: converged seeds fire the callback and leave, NaN seeds leave; on a keyframe every
    # updated seed marks its detector-grid cell (depth_filter.cpp:302-306).  Without the worker thread addKeyframe only
    # initialises seeds (depth_filter.cpp:109-123), with it the keyframe itself is also an update frame (:191-229).
    cell, gcols, grows = 30, int(np.ceil(cam.width / 30)), int(np.ceil(cam.height / 30))

    def replay(update_on_keyframe, remove_a_after=None):
        st = []
        copy_a = None
        for (kf, spx, sf, slevel) in sets:
            a, b, mu, zr, s2 = seedsynth.seed_ctor(1.1 * zbar, 0.5 * zbar, len(spx))
            st.append(dict(kf=kf, px=spx, f=sf, level=slevel, a=a, b=b, mu=mu, zr=zr, s2=s2, alive=np.ones(len(spx), dtype=bool),
                           active=(kf == 0)))
        conv, grid = [], np.zeros(gcols * grows, dtype=np.uint8)
        for k in range(1, n_frames):
            if k != kf2 or update_on_keyframe:
                id0 = 0
                for S in st:
                    if S["active"] and S["alive"].any():
                        idx = np.where(S["alive"])[0]
                        aa, bb, mm, ss = (S[v][idx].copy() for v in ("a", "b", "mu", "s2"))
                        res = orc.update_seeds(cam, pyrs[S["kf"]], pyrs[k], poses[S["kf"]], poses[k], S["px"][idx], S["f"][idx],
                                               S["level"][idx], aa, bb, mm, S["zr"][idx].copy(), ss)
                        S["a"][idx], S["b"][idx], S["mu"][idx], S["s2"][idx] = aa, bb, mm, ss
                        for j in np.where(res["status"] == 4)[0]:
                            conv.append((id0 + idx[j], res["xyz_world"][j], ss[j]))
                        if k == kf2:
                            for j in np.where(res["status"] >= 3)[0]:
                                pc = res["px_cur"][j]
                                grid[int(pc[1] / cell) * gcols + int(pc[0] / cell)] = 1
                        S["alive"][idx[(res["status"] == 4) | (res["status"] == 5)]] = False
                    id0 += len(S["px"])
            if k == kf2:
                st[1]["active"] = True
            if k == remove_a_after:                     # DepthFilter::removeKeyframe(second keyframe): its seeds leave the list
                S = st[1]
                copy_a = np.stack([np.where(S["alive"])[0].astype(np.float64)] + [S[v][S["alive"]].astype(np.float64) for v in ("a", "b", "mu", "s2")], axis=1)
                S["alive"][:] = False
        return (st, conv, grid, copy_a) if remove_a_after is not None else (st, conv, grid)

    summary = np.fromfile(out / "summary.bin")
    expected = {"sync": replay(False), "sync_small": None, "thread": replay(True)}
    expected["sync_small"] = expected["sync"]
    for t, tag in enumerate(("sync", "sync_small", "thread")):
        st, conv_expected, grid_expected = expected[tag]
        assert 0.2 * 3000 < len(conv_expected) < 4500          # the sequence really converges a good part of the seeds
        rows = np.fromfile(out / (tag + "_seeds.bin")).reshape(-1, 5)
        conv = np.fromfile(out / (tag + "_conv.bin")).reshape(-1, 5)
        assert int(summary[1 + t]) == len(conv)
        # the callbacks fire for the same seeds IN THE SAME ORDER as the reference's list walk
        np.testing.assert_array_equal(conv[:, 0].astype(int), np.array([c[0] for c in conv_expected]))
        np.testing.assert_allclose(conv[:, 1:4], np.stack([c[1] for c in conv_expected]), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(conv[:, 4], np.array([c[2] for c in conv_expected]), rtol=1e-5)
        # the seeds still in the list: same ones, same order, same state
        alive_ids = np.concatenate([np.where(st[0]["alive"])[0], 3000 + np.where(st[1]["alive"])[0]])
        np.testing.assert_array_equal(rows[:, 0].astype(int), alive_ids)
        mu_exp = np.concatenate([st[0]["mu"][st[0]["alive"]], st[1]["mu"][st[1]["alive"]]])
        b_exp = np.concatenate([st[0]["b"][st[0]["alive"]], st[1]["b"][st[1]["alive"]]])
        np.testing.assert_allclose(rows[:, 3], mu_exp, rtol=1e-5)
        np.testing.assert_allclose(rows[:, 2], b_exp, rtol=1e-5)
        # detector grid marked by the keyframe update: equal cell for cell
        grid = np.fromfile(out / (tag + "_grid.bin"), dtype=np.uint8)
        np.testing.assert_array_equal(grid, grid_expected)
        if tag == "thread":
            assert grid.sum() > 50
        else:
            assert grid.sum() == 0                              # no update runs on a keyframe without the worker thread
        # converged points are where the plane is
        d = np.abs((conv[:, 1:4] - scene.d * scene.n / (scene.n @ scene.n)) @ scene.n)
        assert np.median(d) < 0.05
    # (d) the second keyframe removed behind the mirror's back one frame after it came (DepthFilter::removeKeyframe is not
    # virtual): the copy taken just before holds the device state of its seeds after their one update, afterwards the first
    # keyframe's seeds go on as if nothing had happened -- same callbacks in the same order as the plain run
    st, conv_expected, _, copy_expected = replay(False, remove_a_after=kf2 + 1)
    rows = np.fromfile(out / "remove_seeds.bin").reshape(-1, 5)
    conv = np.fromfile(out / "remove_conv.bin").reshape(-1, 5)
    copy_b = np.fromfile(out / "remove_copy_b.bin").reshape(-1, 5)
    assert int(summary[4]) == len(conv) == len(conv_expected) and len(conv) > 0.2 * 3000
    np.testing.assert_array_equal(conv[:, 0].astype(int), np.array([c[0] for c in conv_expected]))
    assert (conv[:, 0] < 3000).all()
    np.testing.assert_array_equal(copy_b[:, 0].astype(int), 3000 + copy_expected[:, 0].astype(int))
    np.testing.assert_allclose(copy_b[:, 1:], copy_expected[:, 1:], rtol=1e-5)          # getSeedsCopy sees the device's state
    assert len(copy_b) > 1000 and not st[1]["alive"].any()
    assert not np.allclose(copy_b[:, 3], seedsynth.seed_ctor(1.1 * zbar, 0.5 * zbar, 1)[2][0])      # ... not the constructor's mu
    np.testing.assert_array_equal(rows[:, 0].astype(int), np.where(st[0]["alive"])[0])
    np.testing.assert_allclose(rows[:, 3], st[0]["mu"][st[0]["alive"]], rtol=1e-5)
    # device sub-batch size does not change anything
    np.testing.assert_array_equal(np.fromfile(out / "sync_seeds.bin"), np.fromfile(out / "sync_small_seeds.bin"))
    np.testing.assert_array_equal(np.fromfile(out / "sync_conv.bin"), np.fromfile(out / "sync_small_conv.bin"))
    assert summary[0] >= 1                                     # the main thread really aligned frames meanwhile
    # what DepthFilter::addFrame (= updateSeeds through the device mirror, host side included) took per frame in the synchronous
    # protocols: a record for profiles/ (3000 seeds of one keyframe, then + 1500 of a second), no timing assertion
    frame_us = np.fromfile(out / "sync_frame_us.bin").reshape(-1, 6)
    assert len(frame_us) == n_frames - 1 and (frame_us[:, 0] > 0).all()
    log_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(log_dir):
        with open(os.path.join(log_dir, "host_cpp_frame_us_%s.txt" % ("radtan" if dist is not None else "pinhole")), "w") as fh:
            for tag in ("sync", "sync_small", "remove"):
                fh.write("%s: per addFrame / addKeyframe call: us | seeds on the device, uploaded by the call, converged, NaN, list re-read\n" % tag)
                for row in np.fromfile(out / (tag + "_frame_us.bin")).reshape(-1, 6):
                    fh.write("   %8.1f | %5d %5d %5d %5d %d\n" % (row[0], row[1], row[2], row[3], row[4], row[5]))


# ---- svo::FrameTracker (hip_bridge::FrameTrackerT, the template the drop-in instantiates on the reference's types) run on
# ---- the GPU over a svo::Map built as an object graph
def _write_track_case(case, cs, frames, cfg, last_kf=-1, last_img=None, last_pose=None):
    cam = cs["cam"]
    n_kf, n_points = int(cs["n_kf"]), int(cs["n_points"])
    _write(case / "track_manifest.bin",
           [cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, n_kf, n_points, len(cs["obs_point"]), len(cs["kf_ftr_obs"]), len(cs["cand_obs"]),
            len(frames), cfg["grid_size"], cfg["max_fts"], cfg["quality_min_fts"], cfg["klt_min_level"], cfg["max_frame_features"], last_kf,
            cfg.get("structure_optim_max_pts", 0), cfg.get("keyframe_at", -1), cfg.get("new_candidate_at", -1)], np.float64)
    for k in range(n_kf):
        _write(case / ("kf_%d_img.bin" % k), cs["kf_pyr"][k][0], np.uint8)
    _write(case / "kf_pose.bin", cs["T_kf_w"], np.float64)
    for name, dt in (("pt_pos", np.float64), ("pt_type", np.int32), ("pt_n_failed", np.int32), ("pt_n_succeeded", np.int32), ("pt_obs_offset", np.int32),
                     ("obs_point", np.int32), ("obs_kf", np.int32), ("obs_px", np.float64), ("obs_f", np.float64), ("obs_level", np.int32),
                     ("obs_edgelet", np.uint8), ("obs_grad", np.float64), ("kf_ftr_offset", np.int32), ("kf_ftr_obs", np.int32), ("cand_obs", np.int32)):
        _write(case / (name + ".bin"), cs[name], dt)
    for k, img in enumerate(frames):
        _write(case / ("trk_frame_%d.bin" % k), img, np.uint8)
    if last_kf < 0:
        _write(case / "last_img.bin", last_img, np.uint8)
        _write(case / "last_pose.bin", last_pose, np.float64)


def _run_track_demo(case, out):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([DEMO, str(case), str(out), "track"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    return lambda name, dt: np.fromfile(out / name, dtype=dt)


def test_cpp_frame_tracker_twenty_frame_sequence(tmp_path):
    """The tracking sequence of tests/test_gpu_tracker.py through the C++ host twin: a svo::Map with one keyframe whose
    features hold the map points, svo::FrameTracker::track frame after frame (each tracked svo::Frame, with the svo::Feature
    objects the tracker added to it, is the next call's last frame).  Poses, features and match counts of every frame are
    those of the Python-driven svo_hip_tracker chain -- bit for bit: both sides hand the C-ABI the same tables."""
    import tracking_chain as tc
    from android_svo_amd import hip
    assert os.path.exists(DEMO)
    case, out = tmp_path / "case", tmp_path / "out"
    case.mkdir(); out.mkdir()
    seq = tc.make_sequence(n_frames=20)
    mp = tc.sequence_map(seq)
    n = len(seq["px0"])
    cs = dict(mp, obs_point=np.arange(n, dtype=np.int32), kf_ftr_obs=np.arange(n, dtype=np.int32), cand_obs=np.zeros(0, np.int32))
    cfg = dict(grid_size=tc.CELL, max_fts=tc.MAX_FTS, quality_min_fts=40, klt_min_level=2, max_frame_features=1024)
    _write_track_case(case, cs, [seq["pyrs"][k][0] for k in range(1, 20)], cfg, last_kf=0)
    rd = _run_track_demo(case, out)
    np.testing.assert_array_equal(rd("track_key_before.bin", np.int32).reshape(1, 5), mp["kf_key_point"])       # Frame::setKeyPoints of the twin
    ctx = hip.Context(0)
    trk = hip.Tracker(ctx, seq["cam"], max_keyframes=2, grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=2, max_frame_features=1024)
    trk.upload_keyframe(0, seq["pyrs"][0][0])
    trk.set_map(mp)
    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
    poses = rd("track_poses.bin", np.float64).reshape(-1, 7)
    stats = rd("track_stats.bin", np.float64).reshape(-1, 9)
    assert len(poses) == 19
    for k in range(1, 20):
        r = trk.track(seq["pyrs"][k][0])
        np.testing.assert_array_equal(poses[k - 1], r["T_f_w"])
        assert stats[k - 1, 0] == len(r["feat_px"]) and stats[k - 1, 1] == r["n_matches"] and stats[k - 1, 2] == r["n_trials"]
        assert stats[k - 1, 3] == int(r["result"].sia_n_tracked) and stats[k - 1, 4] == 1 and stats[k - 1, 5] == int(r["result"].pose.num_obs)
        np.testing.assert_array_equal(rd("track_feat_%d_px.bin" % (k - 1), np.float64).reshape(-1, 2), r["feat_px"])
        np.testing.assert_array_equal(rd("track_feat_%d_point.bin" % (k - 1), np.int32), r["feat_point"])
        np.testing.assert_array_equal(rd("track_feat_%d_level.bin" % (k - 1), np.int32), r["feat_level"])
    err = np.array([synth.pose_error(a, t) for a, t in zip(poses, seq["truth"][1:])])
    assert err[:, 0].max() < 2e-3 and err[:, 1].max() < 5e-3
    trk.destroy(); ctx.close()


def test_cpp_frame_tracker_on_a_map_with_deletions(tmp_path):
    """svo::FrameTracker on the 14-keyframe map of the reprojection fixture (tests/golden/reproject_map_ref.npz, recorded from
    the reference's own compiled Reprojector::reprojectMap on a real svo::Map): the twin flattens its object graph (keyframe
    feature lists, observation lists, point candidates, key points) into the tracker's tables, and applies the outcome --
    matches, counters, type promotions, and the deletions through Map::safeDeletePoint / deleteCandidatePoint -- to its
    objects.  Everything the fixture holds must come out equal; then the map is flattened again (it changed) and a second
    frame is tracked against it."""
    from test_oracle_reproject_map import CASES, GOLD
    assert os.path.exists(DEMO)
    case, out = tmp_path / "case", tmp_path / "out"
    case.mkdir(); out.mkdir()
    tag, kw, max_fts = [c for c in CASES if c[0] == "wide"][0]      # (a keyframe loses a key feature to a deleted point in this case)
    g = np.load(GOLD)
    cs = synth.make_map_case(**kw)
    cfg = dict(grid_size=cs["cell_size"], max_fts=max_fts, quality_min_fts=20, klt_min_level=2, max_frame_features=2048, structure_optim_max_pts=20,
               new_candidate_at=1)
    # the second frame: the same scene a small step further (make_map_case's own scene; the same image twice would make the
    # alignment's update exactly zero, for which SE3::exp returns a NaN translation -- in the reference too)
    scene = synth.PlaneScene(seed=kw.get("seed", 31), depth=2.0, tilt=(0.08, -0.05))
    T2 = synth.se3_mul(synth.se3_from_twist([0.012, -0.006, 0.004], [0.002, -0.003, 0.001]), cs["T_cur_w"])
    img2 = scene.render(cs["cam"], T2)
    img3 = scene.render(cs["cam"], synth.se3_mul(synth.se3_from_twist([0.012, -0.006, 0.004], [0.002, -0.003, 0.001]), T2))
    _write_track_case(case, cs, [cs["cur_pyr"][0], img2, img3], cfg, last_kf=-1, last_img=cs["cur_pyr"][0], last_pose=cs["T_cur_w"])
    rd = _run_track_demo(case, out)
    n_kf, n_points = cs["n_kf"], cs["n_points"]
    # Frame::setKeyPoints of the twin == the reference's (the fixture's key points)
    np.testing.assert_array_equal(rd("track_key_before.bin", np.int32).reshape(n_kf, 5), g[tag + "_kf_key_point"])
    stats = rd("track_stats.bin", np.float64).reshape(-1, 9)
    assert [int(stats[0, 1]), int(stats[0, 2])] == [int(v) for v in g[tag + "_n"]]
    # the features Reprojector::reprojectCell created, in creation order (the pose refinement may have dropped some points)
    np.testing.assert_array_equal(rd("track_feat_0_px.bin", np.float64).reshape(-1, 2).view(np.uint64), g[tag + "_feat_px"].view(np.uint64))
    np.testing.assert_array_equal(rd("track_feat_0_level.bin", np.int32), g[tag + "_feat_level"])
    fp = rd("track_feat_0_point.bin", np.int32)
    assert ((fp == g[tag + "_feat_point"]) | (fp == -1)).all() and (fp >= 0).sum() >= 0.8 * len(fp)
    np.testing.assert_array_equal(rd("track_feat_0_edgelet.bin", np.uint8), g[tag + "_feat_type"].astype(np.uint8))
    assert rd("track_feat_0_grad.bin", np.float64).tobytes() == g[tag + "_feat_grad"].tobytes()
    # the map's points as the first frame left them: counters, promotions, deletions
    st = rd("track_points_after_first.bin", np.int32).reshape(n_points, 3)
    np.testing.assert_array_equal(st[:, 0], g[tag + "_type"])
    np.testing.assert_array_equal(st[:, 1], g[tag + "_n_failed"])
    np.testing.assert_array_equal(st[:, 2], g[tag + "_n_succeeded"])
    deleted_now = (st[:, 0] == synth.TYPE_DELETED) & (cs["pt_type"] != synth.TYPE_DELETED)
    assert deleted_now.sum() > 0
    np.testing.assert_array_equal(deleted_now.astype(np.uint8), g[tag + "_unlinked"].astype(np.uint8))
    # safeDeletePoint: every observation of a deleted point let go of it, all others still refer to theirs
    linked = rd("track_obs_linked_after_first.bin", np.int32)
    np.testing.assert_array_equal(linked == 0, deleted_now[cs["obs_point"]])
    # overlap keyframes of the first frame (:110-113)
    ov = rd("track_overlap_first.bin", np.float64).reshape(-1, 2).astype(np.int64)
    np.testing.assert_array_equal(ov[:, 0], g[tag + "_overlap_kf"][:len(ov)])
    np.testing.assert_array_equal(ov[:, 1], g[tag + "_overlap_count"][:len(ov)])
    # removeKeyPoint / setKeyPoints after the deletions: no key feature refers to a deleted point, and a keyframe that lost one
    # chose again among its remaining features
    key_after = rd("track_key_after_first.bin", np.int32).reshape(n_kf, 5)
    np.testing.assert_array_equal(key_after, g[tag + "_kf_key_point_after"])     # what the reference's own removeKeyPoint / setKeyPoints left
    assert (key_after != g[tag + "_kf_key_point"]).any()
    assert not deleted_now[key_after[key_after >= 0]].any()
    for k in range(n_kf):
        o = cs["kf_ftr_obs"][cs["kf_ftr_offset"][k]:cs["kf_ftr_offset"][k + 1]]
        alive = ~deleted_now[cs["obs_point"][o]]
        expect = synth.key_points(cs["cam"], cs["obs_px"][o], alive)
        if (key_after[k] != g[tag + "_kf_key_point"][k]).any():                 # the keyframe lost a key feature: all five were chosen again
            np.testing.assert_array_equal(key_after[k], np.where(expect >= 0, cs["obs_point"][o][np.maximum(expect, 0)], -1))
    # FrameHandlerBase::optimizeStructure(new_frame, 20, 5) behind the first frame: twenty of the frame's points (the reference's own
    # std::nth_element picks them), each equal to Point::optimize over its observations in the tables, bit for bit
    from android_svo_amd import hip
    so = rd("track_structure_first.bin", np.float64).reshape(-1, 4)
    sel = so[:, 0].astype(np.int64)
    assert len(sel) == 20 and len(set(sel.tolist())) == 20 and set(sel.tolist()) <= set(fp[fp >= 0].tolist())
    off, oT, of = [0], [], []
    for p in sel:
        for o in range(cs["pt_obs_offset"][p], cs["pt_obs_offset"][p + 1]):
            oT.append(cs["T_kf_w"][cs["obs_kf"][o]]); of.append(cs["obs_f"][o])
        off.append(len(oT))
    ctx = hip.Context(0)
    pos_ref, _ = hip.point_optimize_batch(ctx, cs["pt_pos"][sel], np.array(off, np.int32), np.array(oT), np.array(of), n_iter=5)
    ctx.close()
    assert np.ascontiguousarray(so[:, 1:]).tobytes() == pos_ref.tobytes()
    # the second frame: tracked straight on (the device tables followed the deletions and the new positions)
    assert stats[1, 4] == 1 and stats[1, 1] >= 0.8 * stats[0, 1] and stats[1, 3] > 0.5 * stats[0, 0], stats
    # map uploads: one before the first frame; none for the second (the deletions of the first are followed on the device); one
    # for the third (a candidate was added to the list after the second frame, as the depth filter's thread does)
    np.testing.assert_array_equal(rd("track_uploads.bin", np.float64), [1, 1, 2])
    assert stats[2, 4] == 1 and stats[2, 1] >= 0.8 * stats[0, 1]
    poses = rd("track_poses.bin", np.float64).reshape(-1, 7)
    rot, trans = synth.pose_error(poses[1], T2)
    assert rot < 3e-3 and trans < 1e-2, (rot, trans)                            # the second frame's true pose


def test_cpp_frame_tracker_promotes_a_frame_to_keyframe(tmp_path):
    """The keyframe flow of FrameHandlerMono::processFrame (:284-330) through the C++ host twin: after the fifth tracked frame
    the svo::Frame becomes a keyframe (Frame::setKeyframe -> key points, Point::addFrameRef for its features, Map::addKeyframe),
    svo::FrameTracker::lastFrameBecameKeyframe keeps its pyramid on the device, and the next call flattens the grown object
    graph (two keyframes, two observations per re-observed point, newest first).  All eight frames equal the Python-driven
    tracker fed with the same grown map as tables (tests/test_gpu_tracker.py::test_last_frame_becomes_a_keyframe) bit for bit."""
    import tracking_chain as tc
    from android_svo_amd import hip
    assert os.path.exists(DEMO)
    case, out = tmp_path / "case", tmp_path / "out"
    case.mkdir(); out.mkdir()
    seq = tc.make_sequence(n_frames=9)
    mp = tc.sequence_map(seq)
    n = len(seq["px0"])
    cs = dict(mp, obs_point=np.arange(n, dtype=np.int32), kf_ftr_obs=np.arange(n, dtype=np.int32), cand_obs=np.zeros(0, np.int32))
    cfg = dict(grid_size=tc.CELL, max_fts=tc.MAX_FTS, quality_min_fts=40, klt_min_level=2, max_frame_features=1024, keyframe_at=4)
    _write_track_case(case, cs, [seq["pyrs"][k][0] for k in range(1, 9)], cfg, last_kf=0)
    rd = _run_track_demo(case, out)
    poses = rd("track_poses.bin", np.float64).reshape(-1, 7)
    ctx = hip.Context(0)
    trk = hip.Tracker(ctx, seq["cam"], max_keyframes=4, grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=2, max_frame_features=1024)
    trk.upload_keyframe(0, seq["pyrs"][0][0])
    trk.set_map(mp)
    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
    used_new_kf = False
    for k in range(1, 9):
        r = trk.track(seq["pyrs"][k][0])
        np.testing.assert_array_equal(poses[k - 1], r["T_f_w"], err_msg="frame %d" % k)
        np.testing.assert_array_equal(rd("track_feat_%d_point.bin" % (k - 1), np.int32), r["feat_point"])
        np.testing.assert_array_equal(rd("track_feat_%d_px.bin" % (k - 1), np.float64).reshape(-1, 2), r["feat_px"])
        used_new_kf = used_new_kf or 1 in list(r["overlap_kf"])
        if k == 5:
            trk.keyframe_from_last_frame(1)
            trk.set_map(tc.map_with_tracked_frame_as_keyframe(seq, mp, r))
    assert used_new_kf
    trk.destroy(); ctx.close()



def _files_equal(a_dir, b_dir):
    names = sorted(f for f in os.listdir(a_dir) if f.startswith("track_") and f.endswith(".bin"))
    assert len(names) > 10
    for f in names:
        if f == "track_uploads.bin":
            continue
        assert (a_dir / f).read_bytes() == (b_dir / f).read_bytes(), f
    return names


@pytest.mark.parametrize("which", ["sequence", "deletions"])
def test_cpp_frame_tracker_group_equals_the_lone_tracker(tmp_path, which):
    """svo::FrameTrackerGroup (hip_bridge::FrameTrackerGroupT, the template a drop-in with several FrameHandlerMono objects would
    instantiate on the reference's types; svo_hip_tracker_group underneath): three copies of the world -- map object graph,
    frames, candidates -- tracked together, one group call per frame-set, every camera's outcome applied to its own objects
    (features added, counters, Map::safeDeletePoint, a tracked frame turned into a keyframe, a candidate added behind the tracker's
    back, structure optimisation).  Every file a world writes must equal, byte for byte, what the lone svo::FrameTracker wrote
    for the same case."""
    import tracking_chain as tc
    from test_oracle_reproject_map import CASES
    assert os.path.exists(DEMO)
    case, out, outg = tmp_path / "case", tmp_path / "out", tmp_path / "outg"
    case.mkdir(); out.mkdir(); outg.mkdir()
    if which == "sequence":
        seq = tc.make_sequence(n_frames=12)
        mp = tc.sequence_map(seq)
        n = len(seq["px0"])
        cs = dict(mp, obs_point=np.arange(n, dtype=np.int32), kf_ftr_obs=np.arange(n, dtype=np.int32), cand_obs=np.zeros(0, np.int32))
        cfg = dict(grid_size=tc.CELL, max_fts=tc.MAX_FTS, quality_min_fts=40, klt_min_level=2, max_frame_features=1024, keyframe_at=5)
        _write_track_case(case, cs, [seq["pyrs"][k][0] for k in range(1, 12)], cfg, last_kf=0)
    else:
        tag, kw, max_fts = [c for c in CASES if c[0] == "wide"][0]
        cs = synth.make_map_case(**kw)
        cfg = dict(grid_size=cs["cell_size"], max_fts=max_fts, quality_min_fts=20, klt_min_level=2, max_frame_features=2048,
                   structure_optim_max_pts=20, new_candidate_at=1)
        scene = synth.PlaneScene(seed=kw.get("seed", 31), depth=2.0, tilt=(0.08, -0.05))
        step = synth.se3_from_twist([0.012, -0.006, 0.004], [0.002, -0.003, 0.001])
        T2 = synth.se3_mul(step, cs["T_cur_w"])
        _write_track_case(case, cs, [cs["cur_pyr"][0], scene.render(cs["cam"], T2), scene.render(cs["cam"], synth.se3_mul(step, T2))], cfg,
                          last_kf=-1, last_img=cs["cur_pyr"][0], last_pose=cs["T_cur_w"])
    _run_track_demo(case, out)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([DEMO, str(case), str(outg), "trackgroup", "3"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "trackgroup OK (3 worlds)" in p.stdout
    for w in range(3):
        _files_equal(out, outg / ("g%d" % w))
