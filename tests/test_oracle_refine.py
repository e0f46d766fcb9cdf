"""Oracle restatements of the two small refinements against the reference.

Point::optimize is pinned end to end by the reference's own compiled point.cpp (tests/golden/refine_ref.npz made
by oracle/gen_golden.py).  pose_optimizer.cpp cannot be built here (its logging macros need the Android NDK
header), so pose_optimizer::optimizeGaussNewton is PARITY UNPINNED as a whole: every piece it is made of is pinned
(projection and SE3 algebra, Frame::jacobian_xyz2uv, Tukey weight, MAD scale, vk::getMedian, Eigen LDLT 6x6 and
6x6 inverse) and the glue is checked against ground truth."""
import zlib

import numpy as np

from android_svo_amd import synth
from oracle import orc


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def test_point_optimize_bit_identical_to_reference(golden):
    g = golden("refine_ref.npz")
    pos0, off, Ts, fs, pos_true, iters = synth.make_point_opt_cases()
    assert [crc(pos0), crc(Ts), crc(fs)] == [int(v) for v in g["point_crc"]]
    moved = 0
    for i in range(len(pos0)):
        out, it = orc.point_optimize(pos0[i], Ts[off[i]:off[i + 1]], fs[off[i]:off[i + 1]], n_iter=int(iters[i]))
        np.testing.assert_array_equal(out, g["point_out"][i])
        moved += np.linalg.norm(out - pos_true[i]) < np.linalg.norm(pos0[i] - pos_true[i])
    assert moved > 0.9 * len(pos0)            # and it does what it is for


def test_robust_cost_pieces_and_small_algebra(golden):
    g = golden("refine_ref.npz")
    assert all(orc.tukey_weight(float(x)) == float(w) for x, w in zip(g["tukey_x"], g["tukey"]))
    for k in range(5):
        e = g["err%d" % k]
        assert np.float32(1.48) * np.float32(orc.median_f(e)) == g["mad"][k]            # MADScaleEstimator::compute
    for k in range(4):
        d = g["dd%d" % k]
        assert np.sort(d)[len(d) // 2] == g["med"][k]                                   # vk::getMedian: element n/2
    for A, inv in zip(g["A6"], g["inv6"]):
        assert np.abs(orc.inverse6(A) - inv).max() <= 1e-12 * np.abs(inv).max()         # Eigen's blocked kernel: tolerance
    for A, b, x in zip(g["A3"], g["b3"], g["x3"]):
        np.testing.assert_array_equal(orc.ldlt3_solve(A, b), x)


def test_pose_optimize_recovers_pose_and_rejects_outliers():
    for seed in (5, 6, 7):
        pc = synth.make_pose_opt_case(seed=seed, n=400)
        em = abs(pc.cam.fx)
        r, hp = orc.pose_optimize(em, pc.T_f_w_init, pc.f, pc.pos, pc.level, pc.has_point)
        rot0, tr0 = synth.pose_error(pc.T_f_w_init, pc.T_f_w_true)
        rot1, tr1 = synth.pose_error(np.array(r.T_f_w), pc.T_f_w_true)
        assert r.ran == 1 and rot1 < 0.2 * rot0 and tr1 < 0.2 * tr0, (rot0, rot1, tr0, tr1)
        had = pc.has_point.astype(bool)
        removed = had & ~hp.astype(bool)
        assert removed[pc.outlier & had].mean() > 0.95          # gross outliers go
        assert removed[~pc.outlier & had].mean() < 0.15         # inliers stay (2 px threshold, 0.3 px noise x 2^level)
        assert r.num_obs == int(had.sum()) - r.n_deleted and r.n_deleted == int(removed.sum())
        assert 0 < r.error_final < r.error_init
        C = np.array(r.Cov).reshape(6, 6)
        assert np.allclose(C, C.T, rtol=1e-6, atol=1e-18) and (np.diag(C) > 0).all()


def test_pose_optimize_without_observations_is_a_no_op():
    pc = synth.make_pose_opt_case(seed=8, n=50)
    r, hp = orc.pose_optimize(abs(pc.cam.fx), pc.T_f_w_init, pc.f, pc.pos, pc.level, np.zeros(50, dtype=np.uint8))
    assert r.ran == 0 and list(r.T_f_w) == list(pc.T_f_w_init) and not hp.any()
