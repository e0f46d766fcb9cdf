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
    # seeds on frame 0
    n_seeds = 3000
    spx = np.floor(np.stack([rng.uniform(40, cam.width - 40, n_seeds), rng.uniform(40, cam.height - 40, n_seeds)], axis=1))
    slevel = rng.choice([0, 0, 1, 2], n_seeds).astype(np.int32)
    spx -= spx % (1 << slevel)[:, None]
    sf = synth.cam2world(cam, spx)
    X = scene.intersect(cam, T0, spx[:, 0], spx[:, 1])
    true_depth = np.linalg.norm(X - synth.se3_inv(T0)[:3], axis=1)
    zbar = float(np.median(true_depth))
    _write(case / "seed_px.bin", spx, np.float64); _write(case / "seed_f.bin", sf, np.float64)
    _write(case / "seed_level.bin", slevel, np.int32)
    _write(case / "depth_mean_min.bin", [1.1 * zbar, 0.5 * zbar], np.float64)

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

    # ---- DepthFilter protocol: oracle replay (seeds leave the list when they converge or go NaN)
    a, b, mu, zr, s2 = seedsynth.seed_ctor(1.1 * zbar, 0.5 * zbar, n_seeds)
    alive = np.ones(n_seeds, dtype=bool)
    conv_expected = []
    for k in range(1, n_frames):
        idx = np.where(alive)[0]
        aa, bb, mm, ss = (v[idx].copy() for v in (a, b, mu, s2))
        res = orc.update_seeds(cam, pyrs[0], pyrs[k], poses[0], poses[k], spx[idx], sf[idx], slevel[idx], aa, bb, mm,
                               zr[idx].copy(), ss)
        a[idx], b[idx], mu[idx], s2[idx] = aa, bb, mm, ss
        done = (res["status"] == 4) | (res["status"] == 5)
        for j in np.where(res["status"] == 4)[0]:
            conv_expected.append((idx[j], res["xyz_world"][j], ss[j]))
        alive[idx[done]] = False
    n_conv_expected = len(conv_expected)
    assert 0.2 * n_seeds < n_conv_expected < n_seeds          # the sequence really converges a good part of the seeds

    summary = np.fromfile(out / "summary.bin")
    for tag, n_conv in (("sync", int(summary[1])), ("thread", int(summary[2]))):
        rows = np.fromfile(out / (tag + "_seeds.bin")).reshape(-1, 5)
        conv = np.fromfile(out / (tag + "_conv.bin")).reshape(-1, 4)
        assert n_conv == len(conv)
        assert abs(n_conv - n_conv_expected) <= 0.01 * n_conv_expected + 2
        ids = rows[:, 0].astype(int)
        exp_alive = np.where(alive)[0]
        common = np.intersect1d(ids, exp_alive)
        assert len(common) >= 0.98 * max(len(ids), len(exp_alive))
        sel = np.isin(ids, common)
        order = np.argsort(ids[sel])
        got = rows[sel][order]
        np.testing.assert_allclose(got[:, 3], mu[np.sort(common)], rtol=5e-3)           # mu of the seeds still alive
        np.testing.assert_allclose(np.median(np.abs(got[:, 3] - mu[np.sort(common)]) / mu[np.sort(common)]), 0, atol=1e-5)
        # converged points are where the plane is
        d = np.abs((conv[:, :3] - scene.d * scene.n / (scene.n @ scene.n)) @ scene.n)
        assert np.median(d) < 0.05
    # the threaded run is the synchronous run (every frame was processed because the feeder waited for idle)
    np.testing.assert_array_equal(np.fromfile(out / "sync_seeds.bin"), np.fromfile(out / "thread_seeds.bin"))
    np.testing.assert_array_equal(np.fromfile(out / "sync_conv.bin"), np.fromfile(out / "thread_conv.bin"))
    assert summary[0] >= 1                                     # the main thread really aligned frames meanwhile
