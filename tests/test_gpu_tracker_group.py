"""svo_hip_tracker_group: FrameHandlerMono::processFrame (S/frame_handler_mono.cpp:171-229) of N independent cameras per call --
one chain of launches for all of them (north_star's "concurrent frame pairs" for the whole per-frame chain).  What is checked:
every camera's outcome is, bit for bit and frame after frame, what a lone svo_hip_tracker gives on the same inputs -- poses,
the frame's features (pixels, bearings, levels, points, gradients) and the map's point counters -- with cameras that differ in
map size, in the frames they see and in what happens to them in between (one of them turns its last frame into a keyframe and
goes on with a two-keyframe map); a group's camera refuses the lone tracker's per-frame call and its destructor."""
import ctypes as C

import numpy as np
import pytest

import tracking_chain as tc
from android_svo_amd import hip

pytestmark = pytest.mark.gpu

CFG = dict(max_keyframes=4, max_points=1024, max_obs=4096, max_kf_features=2048, max_candidates=16, max_items=1024,
           max_frame_features=1024, grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=2)
KEYS = ("T_f_w", "T_f_w_sia", "feat_px", "feat_f", "feat_level", "feat_point", "feat_type", "feat_grad", "type", "n_failed", "n_succeeded")


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def _start(trk, seq, mp):
    n = len(seq["px0"])
    trk.upload_keyframe(0, seq["pyrs"][0][0])
    trk.set_map(mp)
    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)


def _same(a, b, what):
    for k in KEYS:
        assert a[k].tobytes() == b[k].tobytes(), (what, k)
    assert a["n_matches"] == b["n_matches"] and a["n_trials"] == b["n_trials"] and a["map_changed"] == b["map_changed"], what
    assert list(a["overlap_kf"]) == list(b["overlap_kf"]) and list(a["overlap_count"]) == list(b["overlap_count"]), what


def test_cameras_of_a_group_track_as_lone_trackers_do(ctx):
    seq_a = tc.make_sequence(n_frames=10, n_map=600)
    seq_b = tc.make_sequence(n_frames=10, n_map=530)             # another map, other feature counts (same kernel-shape class, see below)
    # camera: (sequence, the frames it sees in order)
    plan = [(seq_a, list(range(1, 9))), (seq_b, list(range(1, 9))), (seq_a, [2, 3, 4, 5, 6, 7, 8, 9]), (seq_b, [1, 1, 2, 2, 3, 4, 5, 6])]
    maps = [tc.sequence_map(s) for s, _ in plan]
    # ---- lone trackers, one camera after the other
    lone = []
    for (seq, frames), mp in zip(plan, maps):
        trk = hip.Tracker(ctx, seq["cam"], **CFG)
        _start(trk, seq, mp)
        lone.append([trk.track(seq["pyrs"][k][0]) for k in frames])
        trk.destroy()
    # ---- the same cameras as a group: one call per frame for all four
    grp = hip.TrackerGroup(ctx, seq_a["cam"], len(plan), **CFG)
    for cam_trk, (seq, _), mp in zip(grp.cameras, plan, maps):
        _start(cam_trk, seq, mp)
    for step in range(8):
        res = grp.track([seq["pyrs"][frames[step]][0] for seq, frames in plan])
        for c, cam_trk in enumerate(grp.cameras):
            got = cam_trk.last_result()
            _same(got, lone[c][step], (c, step))
            assert list(res[c].T_f_w) == list(got["T_f_w"])
    assert len({lone[c][-1]["T_f_w"].tobytes() for c in range(len(plan))}) == len(plan)      # (the cameras really did different things)
    # a camera of a group is tracked and destroyed with its group
    r = hip.CTrackResult()
    img = np.ascontiguousarray(seq_a["pyrs"][1][0])
    assert ctx.lib.svo_hip_tracker_track(grp.cameras[0].h, img.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(r), None, None, None, None, None, None,
                                         None, None, None) == -4
    assert ctx.lib.svo_hip_tracker_destroy(grp.cameras[0].h) == -4
    grp.destroy()


def test_a_camera_whose_first_frame_falls_into_another_shape_class(ctx):
    """The fused SparseImgAlign kernel's shape (tiles per wave) is chosen per LAUNCH by the largest feature count among the
    cameras' last frames, and a frame's sums are grouped by the wave that owns its tiles: a camera whose last frame is in another
    shape class than the group's largest (here 420 features beside 600: 7 against 10 tiles) gets a pose that differs from its
    lone run in the last bits -- as svo_hip_sia_run documents for any batch (include/svo_hip.h).  Tracked frames carry about
    max_fts features, so this concerns the first frame after svo_hip_tracker_set_last_frame with a large feature set; checked:
    rounding level on that frame, same matches, and within the north-star tolerance on the frames after."""
    from android_svo_amd import synth
    seq_a = tc.make_sequence(n_frames=5, n_map=600)
    seq_b = tc.make_sequence(n_frames=5, n_map=420)
    mp_a, mp_b = tc.sequence_map(seq_a), tc.sequence_map(seq_b)
    trk = hip.Tracker(ctx, seq_b["cam"], **CFG)
    _start(trk, seq_b, mp_b)
    want = [trk.track(seq_b["pyrs"][k][0]) for k in range(1, 5)]
    trk.destroy()
    grp = hip.TrackerGroup(ctx, seq_a["cam"], 2, **CFG)
    _start(grp.cameras[0], seq_a, mp_a)
    _start(grp.cameras[1], seq_b, mp_b)
    for k in range(1, 5):
        grp.track([seq_a["pyrs"][k][0], seq_b["pyrs"][k][0]])
        got = grp.cameras[1].last_result()
        rot, trans = synth.pose_error(got["T_f_w"], want[k - 1]["T_f_w"])
        assert (rot < 1e-9 and trans < 1e-9) if k == 1 else (rot < 1e-4 and trans < 1e-3), (k, rot, trans)
        if k == 1:
            assert np.array_equal(got["feat_point"], want[0]["feat_point"])
    grp.destroy()


def test_a_camera_of_a_group_turns_its_last_frame_into_a_keyframe(ctx):
    """svo_hip_tracker_keyframe_from_last_frame + a grown map on ONE camera of a group (its keyframe slots are its own: slot 1 of
    camera 1 is not slot 1 of camera 0), the other camera going on untouched: both equal to lone trackers doing the same."""
    seq = tc.make_sequence(n_frames=9)
    mp = tc.sequence_map(seq)

    def lone(promote):
        trk = hip.Tracker(ctx, seq["cam"], **CFG)
        _start(trk, seq, mp)
        rs = [trk.track(seq["pyrs"][k][0]) for k in range(1, 6)]
        if promote:
            trk.keyframe_from_last_frame(1)
            trk.set_map(tc.map_with_tracked_frame_as_keyframe(seq, mp, rs[-1]))
        rs += [trk.track(seq["pyrs"][k][0]) for k in range(6, 9)]
        trk.destroy()
        return rs
    want = [lone(False), lone(True)]
    grp = hip.TrackerGroup(ctx, seq["cam"], 2, **CFG)
    for t in grp.cameras:
        _start(t, seq, mp)
    got = [[], []]
    for k in range(1, 6):
        grp.track([seq["pyrs"][k][0]] * 2)
        for c in range(2):
            got[c].append(grp.cameras[c].last_result())
    grp.cameras[1].keyframe_from_last_frame(1)
    grp.cameras[1].set_map(tc.map_with_tracked_frame_as_keyframe(seq, mp, got[1][-1]))
    for k in range(6, 9):
        grp.track([seq["pyrs"][k][0]] * 2)
        for c in range(2):
            got[c].append(grp.cameras[c].last_result())
    for c in range(2):
        for i, (a, b) in enumerate(zip(got[c], want[c])):
            _same(a, b, (c, i))
    assert any(1 in list(r["overlap_kf"]) for r in got[1][5:])                  # camera 1's new keyframe took part
    assert not any(1 in list(r["overlap_kf"]) for r in got[0])
    grp.destroy()


def test_group_of_one_and_bad_arguments(ctx):
    seq = tc.make_sequence(n_frames=4)
    mp = tc.sequence_map(seq)
    trk = hip.Tracker(ctx, seq["cam"], **CFG)
    _start(trk, seq, mp)
    want = [trk.track(seq["pyrs"][k][0]) for k in range(1, 4)]
    trk.destroy()
    grp = hip.TrackerGroup(ctx, seq["cam"], 1, **CFG)
    _start(grp.cameras[0], seq, mp)
    for k in range(1, 4):
        grp.track([seq["pyrs"][k][0]])
        _same(grp.cameras[0].last_result(), want[k - 1], k)
    grp.destroy()
    # a camera without a map / last frame: the whole call is refused, nothing is enqueued
    grp = hip.TrackerGroup(ctx, seq["cam"], 2, **CFG)
    _start(grp.cameras[0], seq, mp)
    with pytest.raises(hip.SvoHipError):
        grp.track([seq["pyrs"][1][0]] * 2)
    _start(grp.cameras[1], seq, mp)
    grp.track([seq["pyrs"][1][0]] * 2)
    _same(grp.cameras[1].last_result(), want[0], "after the refused call")
    grp.destroy()
    with pytest.raises(hip.SvoHipError):
        hip.TrackerGroup(ctx, seq["cam"], 2, **dict(CFG, max_items=1000))          # not a multiple of 16


def test_group_cameras_over_maps_that_lose_points(ctx):
    """Two cameras of a group over the 14-keyframe map with multi-observation points, point candidates and points close to their
    deletion thresholds (tests/test_gpu_tracker.py::test_sequence_over_a_map_that_loses_points): the reprojector deletes points
    in several frames, the device unlinks them and re-selects key points per camera (trk_rekey_kernel inside the group call);
    camera 1 sees the frames one step behind camera 0, so the two maps diverge.  Every camera equal, bit for bit, to a lone
    tracker on the same frames; deletions really happen."""
    from test_oracle_reproject_map import CASES, GOLD
    from android_svo_amd import synth
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
    cfg = dict(max_keyframes=max(cs["n_kf"], 1), grid_size=cs["cell_size"], max_fts=max_fts, quality_min_fts=20, max_items=16384)
    frames = [[0, 1, 2, 3, 4, 5], [0, 0, 1, 2, 3, 4]]

    def setup(trk):
        for k in range(cs["n_kf"]):
            trk.upload_keyframe(k, cs["kf_pyr"][k][0])
        trk.set_map(dict(cs, kf_slot=np.arange(cs["n_kf"], dtype=np.int32), kf_key_point=key))
        trk.set_last_frame(cs["T_cur_w"], np.zeros((0, 2)), np.zeros((0, 3)), np.zeros(0, np.int32), img=pyrs[0][0])
    want = []
    for fr in frames:
        trk = hip.Tracker(ctx, cs["cam"], **cfg)
        setup(trk)
        want.append([trk.track(pyrs[k][0]) for k in fr])
        trk.destroy()
    grp = hip.TrackerGroup(ctx, cs["cam"], 2, **cfg)
    for t in grp.cameras:
        setup(t)
    changed = 0
    for s in range(6):
        grp.track([pyrs[frames[c][s]][0] for c in range(2)])
        for c in range(2):
            got = grp.cameras[c].last_result()
            _same(got, want[c][s], (c, s))
            changed += got["map_changed"]
    assert changed >= 3
    grp.destroy()
