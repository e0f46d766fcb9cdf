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


def test_cpp_host_layer_end_to_end(tmp_path):
    assert os.path.exists(DEMO), "build() must have produced android_svo_amd/host/svo_host_demo"
    case, out = tmp_path / "case", tmp_path / "out"
    case.mkdir(); out.mkdir()
    rng = np.random.default_rng(4)
    cam = synth.Camera.default()
    scene = synth.PlaneScene(seed=4, depth=2.0, tilt=(0.08, 0.05))
    n_frames = 7
    T0 = synth.se3_from_twist([0.02, -0.01, 0.0], [0.01, 0.005, -0.01])
    direction = np.array([1.0, 0.3, 0.1]) / np.linalg.norm([1.0, 0.3, 0.1])
    poses = [T0] + [synth.se3_mul(synth.se3_from_twist(direction * 0.035 * k, rng.uniform(-0.004, 0.004, 3)), T0)
                    for k in range(1, n_frames)]
    pyrs = [synth.build_pyramid(scene.render(cam, T)) for T in poses]
    _write(case / "manifest.bin", [cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, 5, n_frames], np.float64)
    for k in range(n_frames):
        _write(case / ("frame_%d_pose.bin" % k), poses[k], np.float64)
        for l in range(5):
            _write(case / ("frame_%d_L%d.bin" % (k, l)), pyrs[k][l], np.uint8)
    # alignment features on frame 0
    px = synth.grid_features(cam, 800, rng)
    f = synth.cam2world(cam, px)
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
        sf = synth.cam2world(cam, spx)
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

    # ---- DepthFilter protocol: oracle replay of the reference's loop.  The seed list is [batch A..., batch B...]; every
    # frame walks it in list order: converged seeds fire the callback and leave, NaN seeds leave; on a keyframe every
    # updated seed marks its detector-grid cell (depth_filter.cpp:302-306).  Without the worker thread addKeyframe only
    # initialises seeds (depth_filter.cpp:109-123), with it the keyframe itself is also an update frame (:191-229).
    cell, gcols, grows = 30, int(np.ceil(cam.width / 30)), int(np.ceil(cam.height / 30))

    def replay(update_on_keyframe):
        st = []
        for (kf, spx, sf, slevel) in sets:
            a, b, mu, zr, s2 = seedsynth.seed_ctor(1.1 * zbar, 0.5 * zbar, len(spx))
            st.append(dict(kf=kf, px=spx, f=sf, level=slevel, a=a, b=b, mu=mu, zr=zr, s2=s2, alive=np.ones(len(spx), dtype=bool),
                           active=(kf == 0)))
        conv, grid = [], np.zeros(gcols * grows, dtype=np.uint8)
        for k in range(1, n_frames):
            if k != kf2 or update_on_keyframe:
                id0 = 0
                for S in st:
                    if S["active"]:
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
        return st, conv, grid

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
    # device sub-batch size does not change anything
    np.testing.assert_array_equal(np.fromfile(out / "sync_seeds.bin"), np.fromfile(out / "sync_small_seeds.bin"))
    np.testing.assert_array_equal(np.fromfile(out / "sync_conv.bin"), np.fromfile(out / "sync_small_conv.bin"))
    assert summary[0] >= 1                                     # the main thread really aligned frames meanwhile
