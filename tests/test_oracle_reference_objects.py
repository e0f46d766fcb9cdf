"""The CPU oracle against the reference's OWN compiled SparseImgAlign / Matcher member functions.

tests/golden/sia_ref.npz, epi_ref.npz and match_direct_ref.npz were produced by oracle/gen_golden.py from
oracle/ref/ref_objects.cpp, which runs the reference's precomputeReferencePatches, computeResiduals, solve,
update, the NLLSSolver driver, Matcher::findEpipolarMatchDirect and Matcher::findMatchDirect unmodified on
real svo::Frame / Feature / Point objects.  These pin the parts of the oracle that earlier fixtures could only
reach piecewise: the residual / Jacobian bodies and the glue of the epipolar search.

Bars: bytes, integers, flags, iteration counts and the f32 patch cache bit-exact; f64 Jacobian cache and H
bit-identical (same statement order, FMA contraction off); poses <= 1e-12; the triangulated depth <= 1e-12
relative (Eigen vectorises the 3-term sums of depthFromTriangulation).
"""
import zlib

import numpy as np
import pytest

from android_svo_amd import synth
from oracle import gen_golden, orc


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@pytest.mark.parametrize("case", gen_golden.SIA_REF_CASES, ids=[c[0] for c in gen_golden.SIA_REF_CASES])
def test_sparse_img_align_against_reference_run(golden, case):
    name, kw, max_level, min_level, n_iter = case
    g = golden("sia_ref.npz")
    fp = gen_golden.make_sia_case(kw)
    n = len(fp.px)
    assert [crc(fp.ref_pyr[0]), crc(fp.cur_pyr[0]), crc(fp.px), crc(fp.pos)] == [int(v) for v in g[name + "_crc"]], \
        "synthetic generator drifted from the fixture"
    o = orc.sparse_img_align(fp, max_level=max_level, min_level=min_level, n_iter=n_iter, early_stop=True)
    assert o.n_tracked == int(g[name + "_n_tracked"])
    assert int(o.stop) == int(g[name + "_stop"])
    rot, trans = synth.pose_error(np.array(o.T_cur_w), g[name + "_T"])
    assert rot <= 1e-12 and trans <= 1e-12, (rot, trans)
    if n == 0:
        np.testing.assert_array_equal(np.array(o.T_cur_w), fp.T_cur_w_init)      # run() returns before touching the pose
        return
    np.testing.assert_array_equal(np.array(o.H), g[name + "_H"])                 # H_ of the last evaluation, bit for bit
    assert o.chi2 == float(g[name + "_chi2"])
    # the solver's iter_ after a level = residual evaluations of that level - 1 (the loop index at the break)
    for level in range(min_level, max_level + 1):
        assert o.iters[level] == int(g[name + "_iter"][level]) + 1, level
    # caches as they stand after the last level: a fresh precompute at min_level reproduces them
    T = np.array([0, 0, 0, 0, 0, 0, 1.0])
    _, _, cache, jac, visible = orc.sia_single_eval(fp, min_level, T, want_caches=True)
    np.testing.assert_array_equal(visible, g[name + "_visible"])
    jac = jac.reshape(-1, 6)
    vis = g[name + "_visible"].astype(bool)
    k = min(n, 64)
    np.testing.assert_array_equal(cache[:k][vis[:k]], g[name + "_cache64"][vis[:k]])
    np.testing.assert_array_equal(jac[:k * 16].reshape(k, 16, 6)[vis[:k]], g[name + "_jac64"].reshape(k, 16, 6)[vis[:k]])
    if vis.all():
        assert [crc(cache), crc(jac)] == [int(v) for v in g[name + "_cache_crc"]]


@pytest.mark.parametrize("case", [c for c in gen_golden.SIA_REF_CASES if c[1].get("n_features", 1)], ids=[c[0] for c in gen_golden.SIA_REF_CASES if c[1].get("n_features", 1)])
def test_fixed_work_mode_against_the_reference_members(golden, case):
    """The oracle's fixed-work mode (early_stop off: exactly n_iter evaluations per level -- the throughput workload of the
    bench) against the reference's compiled computeResiduals / solve / update driven for exactly n_iter evaluations per level
    (ref_sparse_img_align_run_fixed_work: the reference's own loop has an unconditional error-increase exit): pose, H_ and
    chi2_ bit for bit."""
    name, kw, max_level, min_level, n_iter = case
    g = golden("sia_ref.npz")
    fp = gen_golden.make_sia_case(kw)
    o = orc.sparse_img_align(fp, max_level=max_level, min_level=min_level, n_iter=n_iter, early_stop=False)
    np.testing.assert_array_equal(np.array(o.T_cur_w), g[name + "_fw_T"])
    np.testing.assert_array_equal(np.array(o.H), g[name + "_fw_H"])
    assert o.chi2 == float(g[name + "_fw_chi2"]) and o.n_tracked == int(g[name + "_fw_n_tracked"])
    assert all(o.iters[level] == n_iter for level in range(min_level, max_level + 1))


def test_find_epipolar_match_direct_against_reference(golden):
    g = golden("epi_ref.npz")
    sc, d_est, d_min, d_max = gen_golden.epi_case_inputs()
    assert [crc(sc.ref_pyr[0]), crc(sc.cur_pyr[0]), crc(sc.px)] == [int(v) for v in g["crc"]]
    np.testing.assert_array_equal(d_min, g["d_min"])
    paths = {0: 0, 1: 0, 2: 0}
    for i in range(len(d_est)):
        o = orc.find_epipolar_match(sc.cam, sc.ref_pyr, sc.cur_pyr, g["T_cur_ref"], sc.px[i], sc.f[i], int(sc.level[i]),
                                    d_est[i], d_min[i], d_max[i])
        paths[o.path] += 1
        assert bool(o.ok) == bool(g["ok"][i]), i
        assert o.search_level == int(g["search_level"][i]), i
        assert o.epi_length == g["epi_length"][i], i
        assert bytes(o.patch_with_border) == g["pwb"][i].tobytes(), i            # warped reference patch, bytes
        if o.ok:
            assert tuple(o.px_cur) == tuple(g["px_cur"][i]), i                   # sub-pixel match, bit-identical
            assert abs(o.depth - g["depth"][i]) <= 1e-12 * abs(g["depth"][i]), i
    assert paths[0] > 100 and paths[1] > 300                                     # both branches exercised
    assert int(g["ok"].sum()) < len(d_est)                                       # and failures too


def test_find_match_direct_against_reference(golden):
    g = golden("match_direct_ref.npz")
    fp, px_in, lvl, edge, grad = gen_golden.match_direct_inputs()
    assert [crc(fp.ref_pyr[0]), crc(fp.cur_pyr[0]), crc(fp.px)] == [int(v) for v in g["crc"]]
    np.testing.assert_array_equal(px_in, g["px_in"])
    n_ok = 0
    for i in range(len(px_in)):
        ok, px_out, sl = orc.find_match_direct(fp.cam, fp.ref_pyr, fp.cur_pyr, fp.T_ref_w, fp.T_cur_w_true, fp.px[i],
                                               fp.f[i], int(lvl[i]), fp.pos[i], px_in[i], edgelet=bool(edge[i]),
                                               grad=grad[i])
        assert ok == bool(g["ok"][i]), i
        np.testing.assert_array_equal(px_out, g["px_out"][i])
        if g["search_level"][i] >= 0:        # -1: the reference returned before choosing a level
            assert sl == int(g["search_level"][i]), i
        n_ok += ok
    assert 0 < n_ok < len(px_in)
