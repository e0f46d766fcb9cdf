"""The drop-in binding include/svo_dropin/sparse_img_align_hip.cpp EXECUTED on the reference's own types, on the GPU.

oracle/_ref/libsvo_dropin_run.so (`make -C oracle dropin-run`, built where /root/reference is mounted, travels to the GPU box)
holds that translation unit compiled against the reference's unmodified headers, the reference's compiled point.o / config.o /
robust_cost.o / pinhole_camera.o, and a harness that lays two svo::Frame objects, a vk::PinholeCamera and the SparseImgAlign
object out by hand (their constructors need the OpenCV library; oracle/ref/dropin_run.cpp says exactly what is by hand).
What runs is what FrameHandlerMono would run after the one-file swap of INTEGRATION.md: svo::SparseImgAlign::run(ref_frame,
cur_frame) -> std::list walk over ref_frame->fts_, hip_bridge pyramid upload from cv::Mat headers, camera from the
vk::PinholeCamera, libsvo_hip.so, results written back into cur_frame->T_f_w_ / H_ / chi2_ / stop_ / n_meas_.

Compared with SparseImgAlign::run of the reference's OWN compiled code on the same frames (tests/golden/sia_ref.npz,
sia_nlls_ref.npz)."""
import ctypes as C
import os

import numpy as np
import pytest

from android_svo_amd import synth
from oracle import gen_golden, orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libsvo_dropin_run.so")
D = C.c_double


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.skip("oracle/_ref/libsvo_dropin_run.so is built where /root/reference is mounted (make -C oracle dropin-run)")
    # RTLD_LAZY: the constructor and destructor of the drop-in class (never called here) reference cv::Mat members of the OpenCV
    # library, which this image does not have
    return C.CDLL(LIB, mode=os.RTLD_LAZY)


def run_dropin(lib, fp, max_level, min_level, n_iter, method=0, scale_estimator=0, weight_function=0, row_pad=0):
    rp, cp = orc.pyr_ptrs(fp.ref_pyr), orc.pyr_ptrs(fp.cur_pyr)
    px, f, pos = (np.ascontiguousarray(a, dtype=np.float64) for a in (fp.px, fp.f, fp.pos))
    hp = np.ascontiguousarray(fp.has_point, dtype=np.uint8)
    T_ref, T_init = np.ascontiguousarray(fp.T_ref_w, dtype=np.float64), np.ascontiguousarray(fp.T_cur_w_init, dtype=np.float64)
    d5 = np.ascontiguousarray(getattr(fp, "dist", None) if getattr(fp, "dist", None) is not None else np.zeros(5), dtype=np.float64)   # a radtan vk::PinholeCamera
    T_out, fisher, smn = np.zeros(7), np.zeros(36), np.zeros(3)
    nt, chi2, stop = C.c_size_t(0), D(0), C.c_int(0)
    cam = fp.cam
    rc = lib.dropin_sparse_img_align_run(C.c_int(cam.width), C.c_int(cam.height), D(cam.fx), D(cam.fy), D(cam.cx), D(cam.cy), _p(d5, D),
                                         C.c_int(len(fp.ref_pyr)), rp, cp, C.c_int(len(px)), _p(px, D), _p(f, D), _p(pos, D), _p(hp, C.c_uint8),
                                         _p(T_ref, D), _p(T_init, D), C.c_int(max_level), C.c_int(min_level), C.c_int(n_iter), C.c_int(method),
                                         C.c_int(scale_estimator), C.c_int(weight_function), _p(T_out, D), C.byref(nt), _p(fisher, D),
                                         C.byref(chi2), C.byref(stop), _p(smn, D), C.c_int(row_pad))
    assert rc == 0
    return {"T": T_out, "n_tracked": nt.value, "fisher": fisher, "chi2": chi2.value, "stop": stop.value, "scale": np.float32(smn[0]),
            "mu": smn[1], "nu": smn[2]}


REF_CASES = gen_golden.SIA_REF_CASES


@pytest.mark.parametrize("case", REF_CASES, ids=[c[0] for c in REF_CASES])
def test_dropin_run_against_the_reference_run(lib, golden, case):
    """Gauss-Newton as the app constructs it: the pose cur_frame->T_f_w_ ends with, the return value, stop_ and
    getFisherInformation() = H_ / (5e-4 * 255^2) against the reference's own run."""
    name, kw, max_level, min_level, n_iter = case
    g = golden("sia_ref.npz")
    fp = gen_golden.make_sia_case(kw)
    r = run_dropin(lib, fp, max_level, min_level, n_iter)
    rot, trans = synth.pose_error(r["T"], g[name + "_T"])
    assert rot < 1e-4 and trans < 1e-3, (rot, trans)            # north_star tolerance
    assert rot < 2e-5 and trans < 5e-5, (rot, trans)            # what a chi2-order exit flip can cost at most here
    assert r["n_tracked"] == int(g[name + "_n_tracked"]) and r["stop"] == int(g[name + "_stop"])
    if len(fp.px) == 0:
        assert np.array_equal(r["T"], np.asarray(fp.T_cur_w_init, dtype=np.float64))   # run() returns 0 and leaves the pose alone (:55-59)
        return
    if rot < 1e-9:                                               # the same evaluation sequence
        H = g[name + "_H"] / (5e-4 * 255 * 255)
        assert np.abs(r["fisher"] - H).max() <= 1e-6 * np.abs(H).max()
        assert abs(r["chi2"] - float(g[name + "_chi2"])) <= 1e-4 * float(g[name + "_chi2"])


def test_dropin_run_with_padded_image_rows(lib):
    """cv::Mat levels that are not continuous (step = cols + 13, noise in the padding): the bridge compacts them before the
    upload; the run is bit for bit the run on continuous images."""
    fp = gen_golden.make_sia_case(dict(seed=12345, n_features=200))
    a = run_dropin(lib, fp, 4, 0, 30)
    b = run_dropin(lib, fp, 4, 0, 30, row_pad=13)
    assert np.array_equal(a["T"], b["T"]) and a["chi2"] == b["chi2"] and a["n_tracked"] == b["n_tracked"] == 200


def test_dropin_align2d_batch_against_the_reference_fixture(lib, golden):
    """feature_alignment::align2D_batch (feature_alignment_hip.h) on a real svo::Frame against feature_alignment::align2D of
    the reference's own compiled code, patch by patch (tests/golden/align.npz): the flags equal, the pixels bit for bit."""
    g = golden("align.npz")
    cur = np.ascontiguousarray(g["cur"], dtype=np.uint8)
    h, w = cur.shape
    for budget in np.unique(g["n_iter"]):
        sel = np.where(g["n_iter"] == budget)[0]
        same_interior = np.array([np.array_equal(g["pwb"][i].reshape(10, 10)[1:9, 1:9].reshape(64), g["patch"][i]) for i in sel])
        sel = sel[same_interior]                      # (the wrapper takes the bordered patch only: its interior is the 8x8 patch)
        pwb = np.ascontiguousarray(g["pwb"][sel], dtype=np.uint8)
        px = np.ascontiguousarray(g["px_in"][sel], dtype=np.float64).copy()
        conv = np.zeros(len(sel), np.uint8)
        assert lib.dropin_align2d_batch(C.c_int(w), C.c_int(h), _p(cur, C.c_uint8), C.c_int(len(sel)), _p(pwb, C.c_uint8), C.c_int(int(budget)),
                                        _p(px, D), _p(conv, C.c_uint8)) == 0
        np.testing.assert_array_equal(conv.astype(bool), g["ok"][sel].astype(bool))
        want = g["px_out"][sel]
        both_nan = np.isnan(px) & np.isnan(want)
        assert ((px.view(np.uint64) == want.view(np.uint64)) | both_nan).all()


NLLS = [("c0_200", (1, 0, 0)), ("c0_200", (0, 2, 2)), ("nulls_320", (1, 2, 3)), ("c1_2000", (1, 1, 1)), ("border_320", (0, 3, 1)), ("iters5", (1, 1, 2))]


@pytest.mark.parametrize("name,combo", NLLS, ids=[gen_golden.nlls_key(n, c) for n, c in NLLS])
def test_dropin_method_and_robust_cost_go_down_with_the_call(lib, golden, name, combo):
    """method_ = LevenbergMarquardt and setRobustCostFunction on the C++ object (the reference's own base-class member, its
    own estimator / weight classes from robust_cost.o): the binding reads them off the object (dynamic types of
    scale_estimator_ / weight_function_), and mu_, nu_, scale_ come back into it."""
    g = golden("sia_nlls_ref.npz")
    case = {c[0]: c for c in REF_CASES}[name]
    _, kw, max_level, min_level, n_iter = case
    fp = gen_golden.make_sia_case(kw)
    r = run_dropin(lib, fp, max_level, min_level, n_iter, method=combo[0], scale_estimator=combo[1], weight_function=combo[2])
    k = gen_golden.nlls_key(name, combo)
    rot, trans = synth.pose_error(r["T"], g[k + "_T"])
    assert rot < 5e-8 and trans < 5e-8, (rot, trans)
    assert r["n_tracked"] == int(g[k + "_n_tracked"]) and r["stop"] == int(g[k + "_stop"])
    assert abs(r["chi2"] - float(g[k + "_chi2"])) <= 2.5e-7 * float(g[k + "_chi2"])
    if combo[1]:
        assert r["scale"] == np.float32(g[k + "_scale_mu_nu"][0])
    if combo[0]:
        assert abs(r["mu"] - g[k + "_scale_mu_nu"][1]) <= 1e-9 * g[k + "_scale_mu_nu"][1] and r["nu"] == g[k + "_scale_mu_nu"][2]


@pytest.mark.parametrize("seed,n", [(5, 400), (6, 1200), (7, 37)])
def test_dropin_pose_optimizer_on_a_real_frame(lib, seed, n):
    """pose_optimizer::optimizeGaussNewton of include/svo_dropin/pose_optimizer_hip.cpp (compiled instead of the reference's
    pose_optimizer.cpp) called on a svo::Frame with real Feature / Point objects: frame->T_f_w_, frame->Cov_, the
    observations whose Feature::point it nulls and the four outputs, against the oracle restatement of the reference's
    function (the reference's own translation unit does not compile here: SURVEY 8c)."""
    pc = synth.make_pose_opt_case(seed=seed, n=n)
    cam = pc.cam
    em = abs(cam.fx)
    o, hp_o = orc.pose_optimize(em, pc.T_f_w_init, pc.f, pc.pos, pc.level, pc.has_point)
    f, pos = np.ascontiguousarray(pc.f, dtype=np.float64), np.ascontiguousarray(pc.pos, dtype=np.float64)
    level = np.ascontiguousarray(pc.level, dtype=np.int32)
    hp = np.ascontiguousarray(pc.has_point, dtype=np.uint8).copy()
    T_in = np.ascontiguousarray(pc.T_f_w_init, dtype=np.float64)
    T_out, cov, outs, d5 = np.zeros(7), np.zeros(36), np.zeros(4), np.zeros(5)
    rc = lib.dropin_pose_optimize(C.c_int(cam.width), C.c_int(cam.height), D(cam.fx), D(cam.fy), D(cam.cx), D(cam.cy), _p(d5, D), D(2.0),
                                  C.c_int(10), _p(T_in, D), C.c_int(len(f)), _p(f, D), _p(pos, D), _p(level, C.c_int32), _p(hp, C.c_uint8),
                                  _p(T_out, D), _p(cov, D), _p(outs, D))
    assert rc == 0
    rot, trans = synth.pose_error(T_out, np.array(o.T_f_w))
    assert rot < 1e-10 and trans < 1e-10, (rot, trans)
    assert outs[0] == o.estimated_scale                                   # the k-th element of the f32 errors: exact
    assert abs(outs[1] - o.error_init) <= 1e-12 * o.error_init and abs(outs[2] - o.error_final) <= 1e-8 * o.error_final
    assert (hp != hp_o).sum() <= 1 and abs(int(outs[3]) - int(o.num_obs)) <= 1   # only an observation on the threshold may flip
    Co = np.array(o.Cov)
    assert np.abs(cov - Co).max() <= 1e-6 * np.abs(Co).max()


# ---- the reprojector drop-in: the reference's own reprojector.cpp with reprojector.patch applied, under the harness that runs
# ---- the reference's Reprojector::reprojectMap on a real svo::Map (oracle/_ref/libsvo_dropin_reproject.so)
REPROJ_LIB = os.path.join(ROOT, "oracle", "_ref", "libsvo_dropin_reproject.so")


@pytest.fixture(scope="module")
def reproj_lib():
    if not os.path.exists(REPROJ_LIB):
        pytest.skip("oracle/_ref/libsvo_dropin_reproject.so is built where /root/reference is mounted (make -C oracle dropin-run)")
    return C.CDLL(REPROJ_LIB, mode=os.RTLD_LAZY)


def _map_cases():
    from test_oracle_reproject_map import CASES
    return CASES


@pytest.mark.parametrize("tag,kw,max_fts", _map_cases(), ids=[c[0] for c in _map_cases()])
def test_patched_reprojector_on_a_real_map(reproj_lib, golden, tag, kw, max_fts):
    """Reprojector::reprojectMap of the PATCHED reference file -- close-keyframe selection, the projection into cells, the
    candidate loop: the reference's own statements; the cell loop: hip_bridge::reprojectCellsHip -> libsvo_hip.so on the GPU,
    then the reference's serial policy -- on a real svo::Map with keyframes, multi-observation points and candidates,
    against the fixture the UNPATCHED reference left (tests/golden/reproject_map_ref.npz): every integer equal, pixels and
    gradients bitwise, the map's bookkeeping (point types, counters, deletions, key features chosen again) included."""
    from oracle.ref import refpy
    from test_oracle_reproject_map import check_map_result
    g = golden("reproject_map_ref.npz")
    cs = synth.make_map_case(**kw)
    ko = None
    if tag == "rekey":                      # the key features the fixture's run started from
        ko = np.where(g[tag + "_kf_key_point"] == refpy_keys_of_near(g), -2, g[tag + "_kf_key_point"]).astype(np.int32)
    r = refpy.reproject_map(cs, max_fts=max_fts, key_override=ko, library=reproj_lib)
    np.testing.assert_array_equal(r["kf_key_point"], g[tag + "_kf_key_point"])
    check_map_result(g, tag, r)
    np.testing.assert_array_equal(r["kf_key_point_after"], g[tag + "_kf_key_point_after"])


def refpy_keys_of_near(g):
    """the reference's own key-point choice for the map of the "near" / "rekey" cases (the rekey fixture differs from it only
    in the slots that were put in by hand)"""
    return g["near_kf_key_point"]


# ---- the frame-tracker binding (frame_tracker_hip.h): the same map harness compiled with -DREF_RUN_TRACKER_BINDING
TRACKER_LIB = os.path.join(ROOT, "oracle", "_ref", "libsvo_dropin_tracker.so")


@pytest.fixture(scope="module")
def tracker_lib():
    if not os.path.exists(TRACKER_LIB):
        pytest.skip("oracle/_ref/libsvo_dropin_tracker.so is built where /root/reference is mounted (make -C oracle dropin-run)")
    return C.CDLL(TRACKER_LIB, mode=os.RTLD_LAZY)


@pytest.mark.parametrize("tag,kw,max_fts", _map_cases(), ids=[c[0] for c in _map_cases()])
def test_frame_tracker_binding_on_a_real_map(tracker_lib, golden, tag, kw, max_fts):
    """svo::hip_bridge::FrameTracker::track(last, fresh, map, overlap, outcome) -- what replaces lines 175-229 of
    FrameHandlerMono::processFrame -- on a real svo::Map: the binding flattens the reference's keyframes,
    points, observations and candidates, the chain runs on the GPU, and the outcome is applied to the reference's objects with
    the reference's own functions (Frame::addFeature, Map::safeDeletePoint -> Frame::removeKeyPoint, deleteCandidatePoint).
    The reprojection stage against what the reference's Reprojector::reprojectMap leaves on the same scene (every integer,
    pixels and gradients bit for bit; the observations the pose refinement then drops are the ones the oracle drops), the
    refined pose against the oracle's refinement of the fixture's features."""
    from oracle.ref import refpy
    from test_oracle_reproject_map import check_map_result
    g = golden("reproject_map_ref.npz")
    cs = synth.make_map_case(**kw)
    ko = None
    if tag == "rekey":
        ko = np.where(g[tag + "_kf_key_point"] == refpy_keys_of_near(g), -2, g[tag + "_kf_key_point"]).astype(np.int32)
    r = refpy.reproject_map(cs, max_fts=max_fts, key_override=ko, library=tracker_lib)
    pose, sfba, opt = np.zeros(7), np.zeros(4), C.c_int(0)
    assert tracker_lib.dropin_tracker_last(_p(pose, D), _p(sfba, D), C.byref(opt)) == 1, "the binding reported a failure"
    np.testing.assert_array_equal(r["kf_key_point"], g[tag + "_kf_key_point"])
    n_feat = len(g[tag + "_feat_point"])
    enough = n_feat >= 40                                           # Config::qualityMinFts() (config.cpp:83): below it processFrame stops before the refinement
    assert bool(opt.value) == enough
    if enough:
        po, hp = orc.pose_optimize(abs(cs["cam"].fx), cs["T_cur_w"], orc.cam2world(cs["cam"], g[tag + "_feat_px"]),
                                   cs["pt_pos"][g[tag + "_feat_point"]], g[tag + "_feat_level"], np.ones(n_feat, np.uint8))
        dropped = r["feat_point"] < 0                               # Feature::point nulled by the refinement's outlier test
        np.testing.assert_array_equal(dropped, ~hp.astype(bool))
        r = dict(r, feat_point=np.where(dropped, g[tag + "_feat_point"], r["feat_point"]))
        rot, trans = synth.pose_error(pose, np.array(po.T_f_w))
        assert rot < 1e-9 and trans < 1e-9, (rot, trans)
        assert int(sfba[0]) == int(po.num_obs)
    else:
        np.testing.assert_array_equal(pose, np.asarray(cs["T_cur_w"], dtype=np.float64))
    check_map_result(g, tag, r)
    np.testing.assert_array_equal(r["kf_key_point_after"], g[tag + "_kf_key_point_after"])
    # FrameHandlerBase::optimizeStructure through the binding (FrameTracker::optimiseStructure: the frame's points refined on
    # the device) against the reference's compiled Point::optimize on the same objects from the same start: bit for bit
    diff, moved = D(0), D(0)
    n_pts = tracker_lib.dropin_tracker_structure(C.byref(diff), C.byref(moved))
    assert n_pts >= 30 and diff.value == 0.0 and moved.value > 1e-6, (n_pts, diff.value, moved.value)
    # a second frame through the same tracker object, the frame just tracked being the last frame now (its features and points
    # are the alignment's reference: the hand-over inside the binding); the same view (lowest bit of every 13th pixel flipped),
    # the pose starting where the last ended
    sec = np.zeros(8)
    tracker_lib.dropin_tracker_second(_p(sec, D))
    assert sec[0] == 1.0 and sec[3] >= 0.8 * n_feat and sec[6] >= sec[3] and sec[7] >= 1, sec
    assert 0 < sec[1] <= sec[2] and sec[1] >= 0.9 * sec[2], sec      # SparseImgAlign used (nearly) every last-frame feature that has a point
    assert sec[4] < 5e-2, sec                                        # the same view: the pose stays within what the refinement of the new matches moves it (1.6e-2 with 41 matches)


def test_depth_filter_mirror_on_real_seeds(lib):
    """hip_bridge::DeviceSeedMirror<std::list<svo::Seed>> driven with hip_bridge::DepthFilterRefHost -- the work of
    DepthFilterHip::updateSeeds -- on the reference's own Seed / Feature / Frame / vk::PinholeCamera objects: 2500 seeds of one
    keyframe over five frames.  Against the oracle's replay of the reference's loop (depth_filter.cpp:237-341): the convergence
    callbacks fire for the same seeds in the same order with the same points, the seeds left in the list are the same ones in
    the same order with the same state."""
    from android_svo_amd import seedsynth
    rng = np.random.default_rng(11)
    cam = synth.Camera.default()
    scene = synth.PlaneScene(seed=4, depth=2.0, tilt=(0.08, 0.05))
    n_frames = 6
    T0 = synth.se3_from_twist([0.02, -0.01, 0.0], [0.01, 0.005, -0.01])
    direction = np.array([1.0, 0.3, 0.1]) / np.linalg.norm([1.0, 0.3, 0.1])
    poses = [T0] + [synth.se3_mul(synth.se3_from_twist(direction * 0.035 * k, rng.uniform(-0.004, 0.004, 3)), T0) for k in range(1, n_frames)]
    pyrs = [synth.build_pyramid(scene.render(cam, T)) for T in poses]
    n_seeds = 2500
    spx = np.floor(np.stack([rng.uniform(40, cam.width - 40, n_seeds), rng.uniform(40, cam.height - 40, n_seeds)], axis=1))
    slevel = rng.choice([0, 0, 1, 2], n_seeds).astype(np.int32)
    spx -= spx % (1 << slevel)[:, None]
    sf = np.ascontiguousarray(synth.cam2world(cam, spx), dtype=np.float64)
    X = scene.intersect(cam, poses[0], spx[:, 0], spx[:, 1])
    zbar = float(np.median(np.linalg.norm(X - synth.se3_inv(poses[0])[:3], axis=1)))
    depth_mean, depth_min = 1.1 * zbar, 0.5 * zbar

    # ---- the oracle's replay of the list walk
    a, b, mu, zr, s2 = seedsynth.seed_ctor(depth_mean, depth_min, n_seeds)
    alive = np.ones(n_seeds, dtype=bool)
    conv_expected = []
    for k in range(1, n_frames):
        idx = np.where(alive)[0]
        aa, bb, mm, ss = (v[idx].copy() for v in (a, b, mu, s2))
        res = orc.update_seeds(cam, pyrs[0], pyrs[k], poses[0], poses[k], spx[idx], sf[idx], slevel[idx], aa, bb, mm, zr[idx].copy(), ss)
        a[idx], b[idx], mu[idx], s2[idx] = aa, bb, mm, ss
        for j in np.where(res["status"] == 4)[0]:
            conv_expected.append((int(idx[j]), res["xyz_world"][j], float(ss[j])))
        alive[idx[(res["status"] == 4) | (res["status"] == 5)]] = False

    # ---- the mirror on the reference's objects
    kp = orc.pyr_ptrs(pyrs[0])
    cps = (C.POINTER(C.POINTER(C.c_uint8)) * (n_frames - 1))()
    keep = []
    for k in range(1, n_frames):
        pp = orc.pyr_ptrs(pyrs[k])
        keep.append(pp)
        cps[k - 1] = C.cast(pp, C.POINTER(C.POINTER(C.c_uint8)))
    T_kf = np.ascontiguousarray(poses[0], dtype=np.float64)
    T_cur = np.ascontiguousarray(np.stack(poses[1:]), dtype=np.float64)
    px = np.ascontiguousarray(spx, dtype=np.float64)
    rows, conv = np.zeros((n_seeds, 5)), np.zeros((n_seeds, 5))
    n_rows, n_conv = C.c_int(0), C.c_int(0)
    d5 = np.zeros(5)
    rc = lib.dropin_depth_filter_frames(C.c_int(cam.width), C.c_int(cam.height), D(cam.fx), D(cam.fy), D(cam.cx), D(cam.cy), _p(d5, D), C.c_int(5),
                                        kp, _p(T_kf, D), C.c_int(n_frames - 1), cps, _p(T_cur, D), C.c_int(n_seeds), _p(px, D), _p(sf, D),
                                        _p(slevel, C.c_int32), D(depth_mean), D(depth_min), C.c_int(3), _p(rows, D), C.byref(n_rows),
                                        _p(conv, D), C.byref(n_conv))
    assert rc == 0
    rows, conv = rows[:n_rows.value], conv[:n_conv.value]
    assert 0.2 * n_seeds < len(conv_expected)                      # the sequence really converges a good part of the seeds
    np.testing.assert_array_equal(conv[:, 0].astype(int), np.array([c[0] for c in conv_expected]))      # same seeds, same order
    np.testing.assert_allclose(conv[:, 1:4], np.stack([c[1] for c in conv_expected]), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(conv[:, 4], np.array([c[2] for c in conv_expected]), rtol=1e-6)
    np.testing.assert_array_equal(rows[:, 0].astype(int), np.where(alive)[0])
    # the state of the seeds left: the oracle's f32 values bit for bit in (nearly) every seed
    exact = (rows[:, 3] == mu[alive].astype(np.float64)) & (rows[:, 2] == b[alive].astype(np.float64)) & (rows[:, 4] == s2[alive].astype(np.float64)) & \
        (rows[:, 1] == a[alive].astype(np.float64))
    assert exact.mean() >= 0.999, exact.mean()
    np.testing.assert_allclose(rows[:, 3], mu[alive], rtol=1e-5)
    np.testing.assert_allclose(rows[:, 4], s2[alive], rtol=1e-5)
