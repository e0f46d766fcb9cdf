"""GPU parity of the NLLSSolver branches no caller of the reference enables (android_svo_amd/csrc/svo_nlls.hip):
Levenberg-Marquardt and the robust cost, through svo_hip_sia_set_option + svo_hip_sia_run, against SparseImgAlign::run
executed by the reference's own compiled code (tests/golden/sia_nlls_ref.npz) and against the oracle that fixture pins
(tests/test_oracle_nlls.py)."""
import numpy as np
import pytest

from android_svo_amd import hip, synth
from oracle import gen_golden, orc

pytestmark = pytest.mark.gpu

CASES = {c[0]: c for c in gen_golden.SIA_REF_CASES}
PAIRS = [(n, c) for n in gen_golden.SIA_NLLS_CASES for c in gen_golden.SIA_NLLS_COMBOS]


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def _run(ctx, fps, max_level, min_level, n_iter, combo, early_stop=True):
    cam = fps[0].cam
    B = len(fps)
    ref = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
    cur = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
    sia = hip.SparseImgAlign(ctx, B, max(max(len(fp.px) for fp in fps), 1))
    sia.set_frames(ref, cur)
    for i, fp in enumerate(fps):
        ref.upload(i, fp.ref_pyr)
        cur.upload(i, fp.cur_pyr)
        sia.upload_pair(i, fp)
    sia.set_method(combo[0])
    sia.set_robust_cost_function(combo[1], combo[2])
    sia.run(B, sia.params(max_level=max_level, min_level=min_level, n_iter=n_iter, eps=1e-6, early_stop=early_stop))
    out = [(sia.download(i), sia.solver_state(i)) for i in range(B)]
    mode = sia.last_run_mode()
    for o in (sia, ref, cur):
        o.destroy()
    return out, mode


@pytest.mark.parametrize("name,combo", PAIRS, ids=[gen_golden.nlls_key(n, c) for n, c in PAIRS])
def test_against_the_reference_run(ctx, golden, name, combo):
    g = golden("sia_nlls_ref.npz")
    _, kw, max_level, min_level, n_iter = CASES[name]
    fp = gen_golden.make_sia_case(kw)
    out, mode = _run(ctx, [fp], max_level, min_level, n_iter, combo)
    r, (scale, mu, nu) = out[0]
    assert mode == 0                                            # the streaming kernels
    k = gen_golden.nlls_key(name, combo)
    rot, trans = synth.pose_error(np.array(r.T_cur_w), g[k + "_T"])
    print(k, "rot", rot, "trans", trans, "scale", scale, g[k + "_scale_mu_nu"][0], "mu", mu, g[k + "_scale_mu_nu"][1], "iters", list(r.iters[:5]),
          "chi2", r.chi2, float(g[k + "_chi2"]), "stop", r.stop, int(g[k + "_stop"]))
    assert rot < 1e-4 and trans < 1e-3, (rot, trans)            # north_star tolerance
    # chi2 of this path is the reference's one-f32 sum in the reference's order, so Levenberg-Marquardt's accept / reject
    # decisions and the exits are the reference's: the poses agree to rounding level (observed: <= 1e-11 rad, 1.3e-9 in one
    # large-motion case; the bound leaves room for that case's conditioning) and chi2_ is the reference's f32 bit for bit in 53 of the 54 runs, one unit in the last place off
    # in that one
    assert rot < 5e-8 and trans < 5e-8, (rot, trans)
    assert abs(r.chi2 - float(g[k + "_chi2"])) <= 2.5e-7 * float(g[k + "_chi2"])
    if rot < 1e-12:
        assert r.chi2 == float(g[k + "_chi2"])                  # bit for bit
    assert r.n_tracked == int(g[k + "_n_tracked"])
    assert int(r.stop) == int(g[k + "_stop"])
    H = np.array(r.H)
    assert np.abs(H - g[k + "_H"]).max() <= 1e-9 * np.abs(g[k + "_H"]).max()     # for LM: the damped matrix of the last trial
    if combo[1]:
        assert np.float32(scale) == np.float32(g[k + "_scale_mu_nu"][0])      # scale_ bit for bit
    if combo[0]:
        assert abs(mu - g[k + "_scale_mu_nu"][1]) <= 1e-9 * g[k + "_scale_mu_nu"][1] and nu == g[k + "_scale_mu_nu"][2]


def _cases(names):
    return [gen_golden.make_sia_case(CASES[n][1]) for n in names]


@pytest.mark.parametrize("combo", [(1, 0, 0), (1, 2, 3), (0, 1, 1)], ids=["lm", "lm_mad_huber", "gn_tdist"])
def test_a_ragged_batch_equals_the_single_runs_and_the_oracle(ctx, combo):
    """Five frame pairs of different sizes (one of them without features) in ONE call: every slot finishes on its own
    trial count, and is bit for bit what it is when run alone."""
    fps = _cases(["c0_200", "iters5", "empty", "c0_l2", "c0_200"])
    fps[4] = synth.make_frame_pair(seed=4711, n_features=333)
    fps[3] = synth.make_frame_pair(seed=4712, n_features=64)
    batch, _ = _run(ctx, fps, 4, 0, 12, combo)
    for i, fp in enumerate(fps):
        alone, _ = _run(ctx, [fp], 4, 0, 12, combo)
        r, st = batch[i]
        a, st_a = alone[0]
        assert np.array_equal(np.array(r.T_cur_w), np.array(a.T_cur_w)) and r.chi2 == a.chi2 and r.n_tracked == a.n_tracked
        assert list(r.iters) == list(a.iters) and st == st_a
        o = orc.sparse_img_align(fp, 4, 0, 12, method=combo[0], scale_estimator=combo[1], weight_function=combo[2])
        rot, trans = synth.pose_error(np.array(r.T_cur_w), np.array(o.T_cur_w))
        assert rot < 1e-9 and trans < 1e-9, (i, rot, trans)
        assert r.n_tracked == o.n_tracked and int(r.stop) == o.stop
        if len(fp.px):
            assert r.chi2 == o.chi2
            if combo[1]:
                assert np.float32(st[0]) == np.float32(o.scale)
        else:
            assert r.n_tracked == 0 and np.array_equal(np.array(r.T_cur_w), np.asarray(fp.T_cur_w_init, dtype=np.float64))


def test_fixed_work_gauss_newton_with_a_robust_cost(ctx):
    """early_stop = 0 (exactly n_iter evaluations per level) with weights, against the oracle in the same mode"""
    fp = synth.make_frame_pair(seed=99, n_features=500)
    out, _ = _run(ctx, [fp], 4, 1, 6, (0, 2, 2), early_stop=False)
    r, st = out[0]
    o = orc.sparse_img_align(fp, 4, 1, 6, early_stop=False, scale_estimator=2, weight_function=2)
    rot, trans = synth.pose_error(np.array(r.T_cur_w), np.array(o.T_cur_w))
    assert rot < 1e-9 and trans < 1e-9, (rot, trans)
    assert list(r.iters[1:5]) == [6, 6, 6, 6] and r.chi2 == o.chi2 and np.float32(st[0]) == np.float32(o.scale)


def test_zero_iterations(ctx):
    """n_iter = 0: Gauss-Newton evaluates nothing; Levenberg-Marquardt still makes the evaluation that sets chi2_ (:109)"""
    fp = synth.make_frame_pair(seed=5, n_features=150)
    for combo in [(1, 0, 0), (1, 1, 1), (0, 1, 1)]:
        out, _ = _run(ctx, [fp], 3, 1, 0, combo)
        r, _st = out[0]
        o = orc.sparse_img_align(fp, 3, 1, 0, method=combo[0], scale_estimator=combo[1], weight_function=combo[2])
        assert np.allclose(np.array(r.T_cur_w), np.array(o.T_cur_w), rtol=0, atol=1e-15)
        assert r.chi2 == o.chi2 and r.n_tracked == o.n_tracked, (combo, r.chi2, o.chi2, r.n_tracked, o.n_tracked)


def test_unit_scale_is_the_plain_solver_and_other_entry_points_refuse(ctx):
    fp = synth.make_frame_pair(seed=31, n_features=300)
    cam = fp.cam
    ref = hip.Pyramid(ctx, cam.width, cam.height, 5, 1)
    cur = hip.Pyramid(ctx, cam.width, cam.height, 5, 1)
    sia = hip.SparseImgAlign(ctx, 1, 300)
    sia.set_frames(ref, cur)
    ref.upload(0, fp.ref_pyr); cur.upload(0, fp.cur_pyr); sia.upload_pair(0, fp)
    prm = sia.params(n_iter=10)
    sia.run(1, prm)
    plain = sia.download(0)
    assert sia.last_run_mode() == 1
    sia.set_robust_cost_function(hip.SIA_SCALE_UNIT, hip.SIA_WEIGHT_TUKEY)       # UnitScale: use_weights_ stays false
    sia.run(1, prm)
    same = sia.download(0)
    assert sia.last_run_mode() == 1 and np.array_equal(np.array(plain.T_cur_w), np.array(same.T_cur_w))
    with pytest.raises(hip.SvoHipError):
        sia.solver_state(0)                                                      # no such run yet
    sia.set_method(hip.SIA_METHOD_LEVENBERG_MARQUARDT)
    sia.begin(1, prm)
    sia.level_begin(4)
    with pytest.raises(hip.SvoHipError):
        sia.accumulate()                                                         # the step-wise form is Gauss-Newton only
    sia.finish()
    with pytest.raises(hip.SvoHipError):
        sia.set_option(hip.SIA_OPT_METHOD, 2)
    sia.run(1, prm)
    assert sia.last_run_mode() == 0 and sia.solver_state(0)[2] >= 2.0
    sia.set_method(hip.SIA_METHOD_GAUSS_NEWTON)
    sia.run(1, prm)
    again = sia.download(0)
    assert sia.last_run_mode() == 1 and np.array_equal(np.array(plain.T_cur_w), np.array(again.T_cur_w))
    for o in (sia, ref, cur):
        o.destroy()


REF_CASES = gen_golden.SIA_REF_CASES


@pytest.mark.parametrize("case", REF_CASES, ids=[c[0] for c in REF_CASES])
def test_gauss_newton_with_chi2_in_the_reference_order(ctx, golden, case):
    """SVO_HIP_SIA_CHI2_REFERENCE_ORDER: plain Gauss-Newton whose chi2 is the reference's one-f32 sum -- every exit falls
    where SparseImgAlign::run of the reference's own code takes it (iteration counts per level equal, no exception), chi2_
    bit for bit, the pose to rounding level.  (The default sum, per patch and then in f64, can move an error-increase exit
    by one iteration: tests/test_gpu_parity.py::test_sparse_img_align_against_reference_run.)"""
    name, kw, max_level, min_level, n_iter = case
    g = golden("sia_ref.npz")
    fp = gen_golden.make_sia_case(kw)
    cam = fp.cam
    ref = hip.Pyramid(ctx, cam.width, cam.height, 5, 1)
    cur = hip.Pyramid(ctx, cam.width, cam.height, 5, 1)
    sia = hip.SparseImgAlign(ctx, 1, max(len(fp.px), 1))
    sia.set_frames(ref, cur)
    ref.upload(0, fp.ref_pyr); cur.upload(0, fp.cur_pyr); sia.upload_pair(0, fp)
    sia.set_option(hip.SIA_OPT_CHI2, hip.SIA_CHI2_REFERENCE_ORDER)
    sia.run(1, sia.params(max_level=max_level, min_level=min_level, n_iter=n_iter, eps=1e-6, early_stop=True))
    r = sia.download(0)
    assert sia.last_run_mode() == 0
    rot, trans = synth.pose_error(np.array(r.T_cur_w), g[name + "_T"])
    assert rot < 5e-8 and trans < 5e-8, (rot, trans)                 # (1.2e-8 m in the large-motion case, <= 1e-11 otherwise)
    assert r.n_tracked == int(g[name + "_n_tracked"]) and int(r.stop) == int(g[name + "_stop"])
    if len(fp.px):
        # the reference's iter_ after a level is the index its loop broke at: evaluations = iter_ + 1 (n_iter when it ran out)
        want = [min(int(g[name + "_iter"][l]) + 1, n_iter) for l in range(min_level, max_level + 1)]
        assert [r.iters[l] for l in range(min_level, max_level + 1)] == want
        assert abs(r.chi2 - float(g[name + "_chi2"])) <= 2.5e-7 * float(g[name + "_chi2"])
        if rot < 1e-12:
            assert r.chi2 == float(g[name + "_chi2"])
        H = np.array(r.H)
        assert np.abs(H - g[name + "_H"]).max() <= 1e-9 * np.abs(g[name + "_H"]).max()
    for o in (sia, ref, cur):
        o.destroy()


def test_random_small_configs_fuzz(ctx):
    """Many small random problems (sizes around the 64-patch tile boundaries, several image sizes and level ranges,
    features without points, features near / outside the border, large motions) through every branch: each slot against
    its own oracle run -- scale_ bit for bit wherever the first pose sees a patch, the tracked count and the pose for the
    well-posed ones; the degenerate ones (nothing visible: a NaN scale; a handful of patches: a rank-deficient system)
    must simply not fault."""
    rng = np.random.default_rng(777)
    combos = [(1, 0, 0), (0, 1, 1), (1, 2, 2), (0, 3, 3), (1, 1, 3), (0, 2, 1)]
    n_checked = n_degenerate = 0
    for gi, (w, h) in enumerate(((320, 240), (640, 480), (336, 208))):
        group = []
        for _ in range(10):
            n = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 200, 511]))
            fp = synth.make_frame_pair(seed=int(rng.integers(1, 10**6)), width=w, height=h, n_features=n, border=int(rng.choice([4, 8, 24, 48])),
                                       null_point_every=int(rng.choice([0, 0, 2, 5])), t_mag=float(rng.choice([0.01, 0.05, 0.2])),
                                       r_mag=float(rng.choice([0.005, 0.03])))
            group.append(fp)
        for ci, combo in enumerate(combos):
            mx, mn, it = ((4, 0, 30), (4, 2, 12), (3, 3, 4), (2, 0, 3))[(gi + ci) % 4]
            out, _ = _run(ctx, group, mx, mn, it, combo)
            for i, fp in enumerate(group):
                o = orc.sparse_img_align(fp, mx, mn, it, method=combo[0], scale_estimator=combo[1], weight_function=combo[2])
                r, (scale, mu, nu) = out[i]
                want, got = np.array(o.T_cur_w), np.array(r.T_cur_w)
                well_posed = o.n_tracked >= 24 and not np.isnan(want).any() and synth.pose_error(want, fp.T_cur_w_true)[0] < 0.02
                if not well_posed:
                    n_degenerate += 1
                    continue
                n_checked += 1
                rot, trans = synth.pose_error(got, want)
                assert rot < 1e-6 and trans < 1e-6, (gi, combo, i, rot, trans, len(fp.px))
                assert r.n_tracked == o.n_tracked and int(r.stop) == o.stop, (gi, combo, i)
                if combo[1] and rot < 1e-12:
                    assert np.float32(scale) == np.float32(o.scale), (gi, combo, i, scale, o.scale)
    assert n_checked > 60 and n_degenerate > 5, (n_checked, n_degenerate)


def test_full_size_batch_of_c1_frames(ctx):
    """64 frame pairs of BASELINE config C1's size (2000 patches, 640x480, L4..L0) in one call through Levenberg-Marquardt
    with MAD / Huber: four distinct scenes, sixteen replicas each.  Replicas must agree bit for bit (a slot's result does
    not depend on its position or its neighbours), every scene with its oracle run, and the in-order chi2 sum takes the
    workgroup path here (32 000 values per frame)."""
    scenes = [synth.make_frame_pair(seed=500 + i, n_features=2000) for i in range(4)]
    fps = [scenes[i % 4] for i in range(64)]
    combo = (1, 2, 3)
    out, mode = _run(ctx, fps, 4, 0, 30, combo)
    assert mode == 0
    for s in range(4):
        o = orc.sparse_img_align(scenes[s], 4, 0, 30, method=1, scale_estimator=2, weight_function=3)
        r0, st0 = out[s]
        rot, trans = synth.pose_error(np.array(r0.T_cur_w), np.array(o.T_cur_w))
        assert rot < 5e-8 and trans < 5e-8, (s, rot, trans)
        assert r0.n_tracked == o.n_tracked and int(r0.stop) == o.stop and np.float32(st0[0]) == np.float32(o.scale)
        assert abs(r0.chi2 - o.chi2) <= 2.5e-7 * o.chi2
        for k in range(s + 4, 64, 4):
            r, st = out[k]
            assert np.array_equal(np.array(r.T_cur_w), np.array(r0.T_cur_w)) and r.chi2 == r0.chi2 and st == st0 and list(r.iters) == list(r0.iters), (s, k)


def test_one_solver_with_a_growing_number_of_slots(ctx):
    """The branch state (per-slot records, the error / term buffers) is sized by the slots a call uses and grows with them:
    a solver run with 2, then 6, then 3 slots gives every slot the same bits each time, and goes back to the fused kernel
    when the options are taken off again."""
    fps = [synth.make_frame_pair(seed=900 + i, n_features=n) for i, n in enumerate((300, 64, 555, 200, 17, 410))]
    cam = fps[0].cam
    ref = hip.Pyramid(ctx, cam.width, cam.height, 5, 6)
    cur = hip.Pyramid(ctx, cam.width, cam.height, 5, 6)
    sia = hip.SparseImgAlign(ctx, 6, 600)
    sia.set_frames(ref, cur)
    for i, fp in enumerate(fps):
        ref.upload(i, fp.ref_pyr); cur.upload(i, fp.cur_pyr); sia.upload_pair(i, fp)
    prm = sia.params(n_iter=10)
    sia.run(6, prm)
    plain = [np.array(sia.download(i).T_cur_w) for i in range(6)]
    assert sia.last_run_mode() == 1
    sia.set_method(hip.SIA_METHOD_LEVENBERG_MARQUARDT)
    sia.set_robust_cost_function(hip.SIA_SCALE_TDIST, hip.SIA_WEIGHT_TUKEY)
    seen = {}
    for n_slots in (2, 6, 3):
        sia.run(n_slots, prm)
        assert sia.last_run_mode() == 0
        for i in range(n_slots):
            r, st = sia.download(i), sia.solver_state(i)
            key = ([float(v).hex() for v in r.T_cur_w], r.chi2, list(r.iters), st)
            assert seen.setdefault(i, key) == key, (n_slots, i)
        with pytest.raises(hip.SvoHipError):                 # not a frame of that run (its record may be an older run's)
            sia.solver_state(n_slots)
    sia.set_method(hip.SIA_METHOD_GAUSS_NEWTON)
    sia.set_robust_cost_function(hip.SIA_SCALE_UNIT, hip.SIA_WEIGHT_UNIT)
    sia.run(6, prm)
    assert sia.last_run_mode() == 1
    for i in range(6):
        assert np.array_equal(np.array(sia.download(i).T_cur_w), plain[i])
    for o in (sia, ref, cur):
        o.destroy()


@pytest.mark.parametrize("combo", [(0, 2, 1), (1, 2, 3), (0, 1, 2), (1, 3, 1)], ids=["gn_mad_tdist", "lm_mad_huber", "gn_tdist_tukey", "lm_normal_tdist"])
def test_identical_frames_a_zero_scale_and_nan_weights(ctx, combo):
    """Reference and current image identical, the pose already exact: every residual is exactly 0, the MAD / Normal scale is
    0, res / scale_ is 0 / 0 and the TDist / Huber weights NaN (Tukey's comparison turns a NaN into weight 0) -- in the
    reference too.  The run must end like the oracle's (same stop flag and tracked count, a pose that is the oracle's or NaN
    where the oracle's is NaN), not hang or fault: NaN terms take the in-order sum's one-lane path."""
    fp = synth.make_frame_pair(seed=77, n_features=180)
    fp.cur_pyr = [l.copy() for l in fp.ref_pyr]
    fp.T_cur_w_init = np.array(fp.T_ref_w, dtype=np.float64)
    out, _ = _run(ctx, [fp], 4, 1, 8, combo)
    r, (scale, mu, nu) = out[0]
    o = orc.sparse_img_align(fp, 4, 1, 8, method=combo[0], scale_estimator=combo[1], weight_function=combo[2])
    assert int(r.stop) == o.stop and r.n_tracked == o.n_tracked, (r.stop, o.stop, r.n_tracked, o.n_tracked)
    assert (np.float32(scale) == np.float32(o.scale)) or (np.isnan(scale) and np.isnan(o.scale))
    got, want = np.array(r.T_cur_w), np.array(o.T_cur_w)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (got, want)
    if not np.isnan(want).any():
        rot, trans = synth.pose_error(got, want)
        assert rot < 1e-9 and trans < 1e-9, (rot, trans)
