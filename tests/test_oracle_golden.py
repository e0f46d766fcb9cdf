"""The CPU oracle (oracle/svo_oracle.c) against golden vectors produced by the
reference's own code (oracle/gen_golden.py -> tests/golden/*.npz) and against the
known-answer vector recorded in SURVEY.md 8(a-9) / BASELINE.md 2.

Bars: bit-exact for bytes/integers/indices; floating point within the stated
tolerance (most cases are bit-identical because the restatement follows the
reference statement by statement with FMA contraction off).
"""
import ctypes as C
import zlib

import numpy as np
import pytest

from android_svo_amd import synth
from oracle import orc

D = C.c_double
p = orc._p


def _vec(fn, n_out, *ins):
    out = np.zeros(n_out)
    keep = [orc.f64(a) for a in ins]
    getattr(orc.lib(), fn)(*[p(a, D) for a in keep], p(out, D))
    return out


def test_se3_algebra(golden):
    g = golden("se3.npz")
    for i in range(len(g["A"])):
        np.testing.assert_array_equal(_vec("svo_orc_se3_mul", 7, g["A"][i], g["B"][i]), g["mul"][i])
        np.testing.assert_array_equal(_vec("svo_orc_se3_inverse", 7, g["A"][i]), g["inv"][i])
        np.testing.assert_array_equal(_vec("svo_orc_se3_act", 3, g["A"][i], g["p"][i]), g["act"][i])
        np.testing.assert_array_equal(_vec("svo_orc_se3_exp", 7, g["tw"][i]), g["exp"][i])   # NaNs compare equal
        np.testing.assert_array_equal(_vec("svo_orc_so3_log", 3, g["q"][i]), g["log"][i])
        np.testing.assert_array_equal(_vec("svo_orc_se3_rotation_matrix", 9, g["A"][i]), g["rot"][i])
    # the theta == 0 quirk really is NaN in the reference
    assert np.isnan(g["exp"][1][:3]).all() and not np.isnan(g["exp"][1][3:]).any()


def test_jacobian_and_ldlt(golden):
    g = golden("algebra.npz")
    for i in range(len(g["xyz"])):
        np.testing.assert_array_equal(_vec("svo_orc_jacobian_xyz2uv", 12, g["xyz"][i]), g["J"][i])
    for i in range(len(g["H"])):
        x = _vec("svo_orc_ldlt6_solve", 6, g["H"][i], g["b"][i])
        # bit-identical, rank-deficient and zero matrices included: the restatement follows Eigen's summation
        # order inside the unrolled triangular solves (binary-split redux, SSE2 packet form for L^T)
        np.testing.assert_array_equal(x, g["x"][i])


def _fp_from_small(g):
    cam = synth.Camera(int(g["width"]), int(g["height"]), *[float(v) for v in g["cam"]])
    return synth.FramePair(cam, [g["ref%d" % l] for l in range(5)], [g["cur%d" % l] for l in range(5)],
                           g["px"], g["f"], g["pos"], g["has_point"], g["T_ref_w"], g["T_cur_w_true"],
                           g["T_cur_w_init"])


def test_gn_driver_small(golden):
    """Reference NLLSSolver + Eigen LDLT + SE3 drove the restated residual body when the
    fixture was made; the fully restated loop must land on the same pose."""
    g = golden("gn_small.npz")
    fp = _fp_from_small(g)
    r = orc.sparse_img_align(fp)
    assert list(r.iters)[:5] == list(g["iters"][:5])
    assert r.n_tracked == int(g["n_tracked"])
    np.testing.assert_allclose(np.array(r.T_cur_w), g["T_out"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r.chi2, float(g["chi2"]), rtol=1e-12)
    np.testing.assert_allclose(np.array(r.H), g["H"], rtol=1e-12, atol=1e-9)
    r2 = orc.sparse_img_align(fp, max_level=4, min_level=2)      # the shipping L4->L2 configuration
    assert list(r2.iters)[:5] == list(g["iters_l2"][:5])
    np.testing.assert_allclose(np.array(r2.T_cur_w), g["T_out_l2"], rtol=0, atol=1e-12)
    # some features carry no 3D point: they must never be tracked
    assert r.n_tracked <= int((g["has_point"] != 0).sum())


def test_gn_driver_full_size(golden):
    g = golden("gn_full.npz")
    for i in range(len(g["seed"])):
        fp = synth.make_frame_pair(seed=int(g["seed"][i]), n_features=int(g["n"][i]))
        crc = zlib.crc32(fp.ref_pyr[0].tobytes()) & 0xFFFFFFFF
        if crc != int(g["crc_ref"][i]):
            pytest.skip("synthetic generator drifted from the one that made the fixture")
        r = orc.sparse_img_align(fp)
        assert list(r.iters)[:5] == list(g["iters"][i][:5])
        np.testing.assert_allclose(np.array(r.T_cur_w), g["T_out"][i], rtol=0, atol=1e-12)
        rot, trans = synth.pose_error(np.array(r.T_cur_w), g["T_true"][i])
        assert rot < 1e-3 and trans < 2e-3      # also close to ground truth


def test_align2d_and_align1d(golden):
    g = golden("align.npz")
    cur = g["cur"]
    n_conv = 0
    for i in range(len(g["px_in"])):
        ok, px, _ = orc.align2d(cur, g["pwb"][i], g["patch"][i], int(g["n_iter"][i]), g["px_in"][i])
        assert ok == bool(g["ok"][i]), i
        np.testing.assert_array_equal(px, g["px_out"][i], err_msg=str(i))       # bit-identical, NaN == NaN
        n_conv += ok
        ok1, px1, hinv, _ = orc.align1d(cur, g["dirs"][i], g["pwb"][i], g["patch"][i], int(g["n_iter"][i]),
                                        g["px_in"][i])
        assert ok1 == bool(g["ok1"][i]), i
        np.testing.assert_array_equal(px1, g["px_out1"][i], err_msg=str(i))
        np.testing.assert_array_equal(hinv, g["hinv"][i])
    assert n_conv > 150          # the fixture is dominated by ordinary converging cases
    assert not g["ok"][0] and not g["ok"][1] and not g["ok"][2] and not g["ok"][3]   # border / flat cases


def test_matcher_pieces(golden):
    g = golden("matcher.npz")
    L = orc.lib()
    cur, ref = g["cur"], g["ref"]
    w, h, fx, fy, cx, cy = g["cam"]
    cam = orc.camera(synth.Camera(int(w), int(h), fx, fy, cx, cy))
    for i in range(len(g["zs"])):
        x0, y0 = int(g["zxy"][i, 0]), int(g["zxy"][i, 1])
        ptr = C.cast(cur.ctypes.data + y0 * cur.shape[1] + x0, C.POINTER(C.c_uint8))
        zp = np.ascontiguousarray(g["zp"][i])
        assert L.svo_orc_zmssd(p(zp, C.c_uint8), ptr, cur.shape[1]) == int(g["zs"][i])
    ref_pyr = synth.build_pyramid(ref, 3)
    n_nonzero_level = 0
    for i in range(len(g["A"])):
        A = np.zeros(4)
        px_ref, f_ref, T = orc.f64(g["px_ref"][i]), orc.f64(g["f_ref"][i]), orc.f64(g["T_cur_ref"][i])
        lvl = int(g["level_ref"][i])
        if i != 17:      # A[17] was overwritten by hand in the fixture
            L.svo_orc_get_warp_matrix_affine(C.byref(cam), C.byref(cam), p(px_ref, D), p(f_ref, D),
                                             D(float(g["depth"][i])), p(T, D), C.c_int(lvl), p(A, D))
            np.testing.assert_array_equal(A, g["A"][i], err_msg=str(i))
        A = orc.f64(g["A"][i])
        best = L.svo_orc_get_best_search_level(p(A, D), C.c_int(2))
        assert best == int(g["best"][i])
        n_nonzero_level += best > 0
        patch = np.full(100, 7, dtype=np.uint8)
        img = ref_pyr[lvl]
        L.svo_orc_warp_affine(p(A, D), p(img, C.c_uint8), C.c_int(img.shape[1]), C.c_int(img.shape[0]),
                              p(px_ref, D), C.c_int(lvl), C.c_int(best), C.c_int(5), p(patch, C.c_uint8))
        np.testing.assert_array_equal(patch, g["patches"][i], err_msg=str(i))    # bytes: bit-exact
        pb = np.zeros(64, dtype=np.uint8)
        L.svo_orc_patch_from_border(p(patch, C.c_uint8), p(pb, C.c_uint8))
        np.testing.assert_array_equal(pb, patch.reshape(10, 10)[1:9, 1:9].reshape(64))
        f_cur = orc.f64(g["f_cur"][i])
        d = D(0)
        ok = L.svo_orc_depth_from_triangulation(p(T, D), p(f_ref, D), p(f_cur, D), C.byref(d))
        assert bool(ok) == bool(g["tri_ok"][i]), i
        if ok:
            np.testing.assert_allclose(d.value, g["tri_depth"][i], rtol=1e-13)
        fo = np.zeros(3)
        L.svo_orc_cam2world(C.byref(cam), D(px_ref[0]), D(px_ref[1]), p(fo, D))
        np.testing.assert_array_equal(fo, g["f_ref"][i])
    assert n_nonzero_level >= 8 and (g["patches"][17] == 7).all() and not g["tri_ok"][20]


def test_vision(golden):
    g = golden("vision.npz")
    L = orc.lib()
    cur = g["cur"]
    for (u, v), val in zip(g["uv"], g["val"]):
        got = L.svo_orc_interpolate_8u(p(cur, C.c_uint8), C.c_int(cur.shape[1]), C.c_float(u), C.c_float(v))
        assert np.float32(got) == val
    img = np.ascontiguousarray(g["img"])
    out = np.zeros((24, 32), dtype=np.uint8)
    L.svo_orc_half_sample(p(img, C.c_uint8), 64, 48, p(out, C.c_uint8))
    np.testing.assert_array_equal(out, g["half_scalar"])
    np.testing.assert_array_equal(out, synth.half_sample(img))       # generator uses the same rule
    L.svo_orc_half_sample_sse2form(p(img, C.c_uint8), 64, 48, p(out, C.c_uint8))
    np.testing.assert_array_equal(out, g["half_sse2"])
    assert (g["half_sse2"] != g["half_scalar"]).any()                # the two ISAs really do differ
    odd = np.ascontiguousarray(img[:, :50])
    out2 = np.zeros((24, 25), dtype=np.uint8)
    L.svo_orc_half_sample(p(odd, C.c_uint8), 50, 48, p(out2, C.c_uint8))
    np.testing.assert_array_equal(out2, g["half_odd"])


def test_update_seed_known_answer():
    """SURVEY.md 8(a-9) / BASELINE.md 2: values printed by the reference's own
    DepthFilter::updateSeed (includes the sqrt(2)-for-sqrt(2 pi) constant)."""
    s = orc.seed_init(2.0, 1.0)
    np.testing.assert_allclose(s, [10, 10, 0.5, 1.0, 0.0277778], rtol=2e-6)
    s2 = orc.update_seed(0.52, 0.01, s)
    np.testing.assert_allclose(s2[[0, 1, 2, 4]], [10.4296455, 9.88126183, 0.511521995, 0.0118117034], rtol=3e-7)
    assert s2[3] == 1.0
    # NaN guard (depth_filter.cpp:371): negative variance sum leaves the seed untouched
    s3 = orc.update_seed(0.5, -1.0, s)
    np.testing.assert_array_equal(s3, s)
