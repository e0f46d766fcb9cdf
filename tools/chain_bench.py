#!/usr/bin/env python3
"""Per-frame time of the tracking front end -- SparseImgAlign against the last frame, reprojection of the keyframe's map
(one match per grid cell), motion-only pose refinement: the chain of tests/tracking_chain.py, i.e. what
FrameHandlerMono::processFrame runs per image -- through the HIP library's host-buffer entry points (uploads of the two
pyramids and of the features included, as a drop-in caller pays them) and through the CPU oracle on one host core.
Diagnostic; not run by the driver.  Prints one JSON line.

    python tools/chain_bench.py [--frames 20] [--min-level 2]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tracking_chain as tc  # noqa: E402
from android_svo_amd import hip  # noqa: E402


class Timed:
    """wraps a stages object and accumulates wall time per stage"""

    def __init__(self, inner, sync=None):
        self.inner, self.sync = inner, sync
        self.t = {"align": 0.0, "reproject": 0.0, "refine": 0.0}
        self.calls = 0

    def _run(self, name, *a):
        t0 = time.perf_counter()
        r = getattr(self.inner, name)(*a)
        if self.sync:
            self.sync()
        self.t[name] += time.perf_counter() - t0
        return r

    def align(self, *a):
        self.calls += 1
        return self._run("align", *a)

    def reproject(self, *a):
        return self._run("reproject", *a)

    def refine(self, *a):
        return self._run("refine", *a)


class ResidentStages:
    """The HIP stages the way the drop-in runs them (include/svo_dropin/svo_hip_bridge.h): the previous frame's pyramid
    stays on the device and becomes the reference, only level 0 of the new image crosses PCIe (the pyramid is built on
    the device)."""

    def __init__(self, ctx, seq):
        from test_gpu_sequence import HipStages
        self.h = HipStages(ctx, seq)
        self.seq = seq
        self.pyr = [self.h.ref, self.h.cur]
        self.pyr[0].upload_level0_and_build(0, seq["pyrs"][0][0])
        self.last = 0                                            # index of the pyramid that holds frame k - 1

    def align(self, fp, k, min_level):
        h = self.h
        ref, cur = self.pyr[self.last], self.pyr[1 - self.last]
        if k == 1:                                               # a new run of the chain starts from the keyframe again
            ref.upload_level0_and_build(0, self.seq["pyrs"][0][0])
        cur.upload_level0_and_build(0, self.seq["pyrs"][k][0])
        h.sia.set_frames(ref, cur)
        h.cur = cur                                              # the reprojection matches against it
        h.sia.upload_pair(0, fp)
        h.sia.run(1, h.sia.params(max_level=4, min_level=min_level, n_iter=30, eps=1e-6, early_stop=True))
        r = h.sia.download(0)
        self.last = 1 - self.last
        return np.array(r.T_cur_w), r.n_tracked

    def reproject(self, *a):
        return self.h.reproject(*a)

    def refine(self, *a):
        return self.h.refine(*a)

    def destroy(self):
        self.h.destroy()


def oracle_tracker_leg(seq, min_level, max_fts=None, stop_when_lost=False):
    """The same frame function composed from the CPU oracle's pieces (tests/tracking_chain.py: oracle_track_frame) on one
    host thread: per-frame time and the poses, for the parity figure of the tracker leg."""
    from oracle import orc
    mp = tc.sequence_map(seq)
    n = len(seq["px0"])
    state = {"pt_type": mp["pt_type"].copy(), "pt_n_failed": mp["pt_n_failed"].copy(), "pt_n_succeeded": mp["pt_n_succeeded"].copy(),
             "unlinked": np.zeros(n, np.uint8)}
    times = []

    def track(k, last):
        t0 = time.perf_counter()
        r = tc.oracle_track_frame(orc, mp, state, last, seq["pyrs"][k - 1], seq["pyrs"][k], min_level, max_fts=max_fts or tc.MAX_FTS)
        times.append(time.perf_counter() - t0)
        return r
    poses, n_matches, winners, _ = tc.run_tracker_chain(seq, track, min_level, stop_when_lost=stop_when_lost)
    return {"frames": len(times), "ms_per_frame_total": float(np.mean(times) * 1e3)}, poses, winners


def tracker_poses(ctx, seq, min_level, max_fts=None, stop_when_lost=False):
    """poses and matched points of the sequence through svo_hip_tracker_track (for the parity figure)"""
    mp = tc.sequence_map(seq)
    n = len(seq["px0"])
    trk = hip.Tracker(ctx, seq["cam"], max_keyframes=2, max_points=1024, max_obs=1024, max_kf_features=1024, max_candidates=16, max_items=1024,
                      max_frame_features=1024, grid_size=tc.CELL, max_fts=max_fts or tc.MAX_FTS, klt_min_level=min_level)
    trk.upload_keyframe(0, seq["pyrs"][0][0])
    trk.set_map(mp)
    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
    poses, _, winners, _ = tc.run_tracker_chain(seq, lambda k, last: trk.track(seq["pyrs"][k][0]), min_level, stop_when_lost=stop_when_lost)
    trk.destroy()
    return poses, winners


def single_stream_chain(ctx, n_frames=20):
    """What bench.py reports as the single-stream figure: per-frame wall time of svo_hip_tracker_track over the synthetic
    tracking sequence at the shipping pyramid range (L4-L2) and at L4-L0, next to the CPU oracle's composition of the same
    frame function on one host thread, with the largest pose difference between the two chains."""
    seq = tc.make_sequence(n_frames=n_frames)
    out = {"what": "one camera, one frame at a time: image in -> SparseImgAlign vs the last frame -> Reprojector::reprojectMap (one match per grid "
                   "cell of %d map points) -> pose refinement -> pose + features out (svo_hip_tracker_track, one stream, one synchronisation); "
                   "640x480, %d frames" % (len(seq["px0"]), n_frames - 1)}
    for tag, min_level in (("L4_L2_shipping_default", 2), ("L4_L0", 0)):
        leg = tracker_leg(ctx, seq, min_level)
        cpu, c_poses, c_win = oracle_tracker_leg(seq, min_level)
        g_poses, g_win = tracker_poses(ctx, seq, min_level)
        from android_svo_amd import synth
        diff = np.array([synth.pose_error(a, b) for a, b in zip(g_poses, c_poses)])
        same = all(np.array_equal(a, b) for a, b in zip(g_win, c_win))
        assert diff[:, 0].max() < 1e-4 and diff[:, 1].max() < 1e-3 and same, "tracking chain parity violated"
        out[tag] = {"ms_per_frame": leg["ms_per_frame_total"], "ms_per_frame_median": leg["ms_per_frame_median"], "ms_per_frame_min": leg["ms_per_frame_min"],
                    **({"ms_per_frame_image_in_tracker_buffer": leg["ms_per_frame_image_in_tracker_buffer"]} if min_level == 2 else {}),
                    "ms_per_frame_image_in_tracker_buffer_median": leg["ms_per_frame_image_in_tracker_buffer_median"],
                    "cpu_oracle_1_thread_ms_per_frame": cpu["ms_per_frame_total"], "speedup_vs_1_thread": cpu["ms_per_frame_total"] / leg["ms_per_frame_total"],
                    "max_pose_diff_vs_cpu_chain": {"rot_rad": float(diff[:, 0].max()), "trans_m": float(diff[:, 1].max())},
                    "matched_points_equal_in_every_frame": bool(same)}
    # N cameras = N trackers on N contexts driven by N host threads (no grouped entry point): whole-job frames/s, L4-L2
    out["cameras_frames_per_s"] = multi_camera(seq, 2, cams=(1, 4, 8), repeats=4)
    # ... and through svo_hip_tracker_group_track: one host thread, one chain of launches per call for all cameras
    out["group_frames_per_s"] = group_cameras(ctx, seq, 2, cams=(1, 8, 64), repeats=3)
    return out


def tracker_leg(ctx, seq, min_level, repeats=3):
    """The same frames through svo_hip_tracker_track: the whole chain of a frame enqueued on one stream with the aligned
    pose, the candidates and the matches staying on the device, level 0 of the new image from page-locked staging, one
    synchronisation per frame.  Timed around the C call (the wall time a C++ caller sees)."""
    import ctypes as C
    mp = tc.sequence_map(seq)
    n = len(seq["px0"])
    trk = hip.Tracker(ctx, seq["cam"], max_keyframes=2, max_points=1024, max_obs=1024, max_kf_features=1024, max_candidates=16, max_items=1024,
                      max_frame_features=1024, grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=min_level)
    trk.upload_keyframe(0, seq["pyrs"][0][0])
    imgs = [np.ascontiguousarray(p[0]) for p in seq["pyrs"]]
    res = hip.CTrackResult()
    times = []
    # untimed passes first, for at least 50 ms: the sequence was rendered on the host with the GPU idle, and an idle chip
    # runs its first tens of milliseconds far below its clocks (bench_c2.timed)
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.05:
        trk.set_map(mp)
        trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
        for k in range(1, len(imgs)):
            ctx.check(ctx.lib.svo_hip_tracker_track(trk.h, imgs[k].ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(res), None, None, None, None, None,
                                                    None, None, None, None), "tracker_track")
    for rep in range(repeats + 1):
        trk.set_map(mp)                                             # fresh point counters: every pass tracks the same sequence
        trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
        ctx.sync()
        for k in range(1, len(imgs)):
            t0 = time.perf_counter()
            rc = ctx.lib.svo_hip_tracker_track(trk.h, imgs[k].ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(res), None, None, None, None, None,
                                               None, None, None, None)
            dt = time.perf_counter() - t0
            ctx.check(rc, "tracker_track")
            assert res.n_matches >= 50 and res.map_changed == 0
            if rep > 0:
                times.append(dt)
    # the same frames with the image already in the tracker's page-locked buffer (svo_hip_tracker_image_buffer: a camera
    # pipeline that lets its frames land there): the copy into the buffer is outside the timed call
    buf = trk.image_buffer()
    buf_p = buf.ctypes.data_as(C.POINTER(C.c_uint8))
    times_buf = []
    for rep in range(repeats + 1):
        trk.set_map(mp)
        trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
        ctx.sync()
        for k in range(1, len(imgs)):
            buf[:] = imgs[k]
            t0 = time.perf_counter()
            rc = ctx.lib.svo_hip_tracker_track(trk.h, buf_p, C.byref(res), None, None, None, None, None, None, None, None, None)
            dt = time.perf_counter() - t0
            ctx.check(rc, "tracker_track")
            assert res.n_matches >= 50 and res.map_changed == 0
            if rep > 0:
                times_buf.append(dt)
    trk.destroy()
    t = np.array(times) * 1e3
    tb = np.array(times_buf) * 1e3
    return {"frames": len(t), "ms_per_frame_total": float(t.mean()), "ms_per_frame_median": float(np.median(t)), "ms_per_frame_min": float(t.min()),
            "ms_per_frame_image_in_tracker_buffer": float(tb.mean()), "ms_per_frame_image_in_tracker_buffer_median": float(np.median(tb)),
            "what": "svo_hip_tracker_track: image upload + pyramid + SparseImgAlign + reprojectMap + pose refinement + hand-over + result download, one sync"}


def multi_camera(seq, min_level, cams=(1, 2, 4, 8), repeats=6):
    """north_star's "concurrent frame pairs" for the WHOLE per-frame chain: N independent cameras, each its own svo_hip_tracker
    on its own context (= its own stream) driven by its own host thread, one frame at a time each -- what an application with N
    cameras does with the C-ABI as it is (no grouped entry point: every kernel of a tracker's chain is a single workgroup or a
    few, so N chains occupy N sets of CUs side by side).  Returns frames/s of the whole job per camera count and checks that
    every camera's last pose of every pass is bit-equal to the one-camera run's."""
    import ctypes as C
    import threading
    mp = tc.sequence_map(seq)
    n = len(seq["px0"])
    imgs = [np.ascontiguousarray(p[0]) for p in seq["pyrs"]]
    ptrs = [im.ctypes.data_as(C.POINTER(C.c_uint8)) for im in imgs]
    idx = np.arange(n, dtype=np.int32)
    out = {}
    ref_pose = None
    for n_cam in cams:
        ctxs = [hip.Context(0) for _ in range(n_cam)]
        trks = [hip.Tracker(c, seq["cam"], max_keyframes=2, max_points=1024, max_obs=1024, max_kf_features=1024, max_candidates=16, max_items=1024,
                            max_frame_features=1024, grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=min_level) for c in ctxs]
        for t in trks:
            t.upload_keyframe(0, seq["pyrs"][0][0])
        start = threading.Barrier(n_cam + 1)
        done = threading.Barrier(n_cam + 1)
        last_pose = [None] * n_cam
        errs = []

        def work(ci):
            ctx, trk = ctxs[ci], trks[ci]
            res = hip.CTrackResult()
            try:
                for rep in range(repeats + 1):
                    trk.set_map(mp)
                    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], idx, kf_slot=0)
                    ctx.sync()
                    if rep == 1:
                        start.wait()                       # pass 0 is the warm-up; the clock runs over passes 1..repeats
                    for k in range(1, len(imgs)):
                        rc = ctx.lib.svo_hip_tracker_track(trk.h, ptrs[k], C.byref(res), None, None, None, None, None, None, None, None, None)
                        if rc != 0:
                            raise RuntimeError("tracker_track failed: %d" % rc)
                    pose = tuple(float(v) for v in res.T_f_w)
                    if last_pose[ci] is not None and pose != last_pose[ci]:
                        raise RuntimeError("camera %d: a pass ended on another pose than the one before" % ci)
                    last_pose[ci] = pose
            except Exception as e:      # noqa: BLE001
                errs.append(e)
                try:
                    start.abort(); done.abort()
                except Exception:
                    pass
                return
            done.wait()
        ths = [threading.Thread(target=work, args=(i,)) for i in range(n_cam)]
        [t.start() for t in ths]
        start.wait()
        t0 = time.perf_counter()
        done.wait()
        dt = time.perf_counter() - t0
        [t.join() for t in ths]
        if errs:
            raise errs[0]
        if ref_pose is None:
            ref_pose = last_pose[0]
        assert all(p == ref_pose for p in last_pose), "a camera's result depends on its company"
        frames = n_cam * repeats * (len(imgs) - 1)
        out[str(n_cam)] = frames / dt
        for t in trks:
            t.destroy()
        for c in ctxs:
            c.close()
    return out


def group_cameras(ctx, seq, min_level, cams=(1, 8, 64), repeats=4):
    """The same N cameras through svo_hip_tracker_group_track: ONE host thread, one chain of launches per call for all cameras
    (every kernel takes a workgroup, or a slice of its grid, per camera).  Whole-job frames/s per camera count; every camera's last
    pose of every pass bit-equal to a lone tracker's."""
    mp = tc.sequence_map(seq)
    n = len(seq["px0"])
    idx = np.arange(n, dtype=np.int32)
    cfg = dict(max_keyframes=2, max_points=1024, max_obs=1024, max_kf_features=1024, max_candidates=16, max_items=1024,
               max_frame_features=1024, grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=min_level)
    lone = hip.Tracker(ctx, seq["cam"], **cfg)
    lone.upload_keyframe(0, seq["pyrs"][0][0])
    lone.set_map(mp)
    lone.set_last_frame(seq["T0"], seq["px0"], seq["f0"], idx, kf_slot=0)
    for k in range(1, len(seq["pyrs"])):
        want = lone.track(seq["pyrs"][k][0])["T_f_w"]
    lone.destroy()
    out = {}
    for n_cam in cams:
        grp = hip.TrackerGroup(ctx, seq["cam"], n_cam, **cfg)
        bufs = [t.image_buffer() for t in grp.cameras]                 # the cameras' frames land in the trackers' own page-locked buffers
        for t in grp.cameras:
            t.upload_keyframe(0, seq["pyrs"][0][0])
        dt = 0.0
        for rep in range(repeats + 1):
            for t in grp.cameras:
                t.set_map(mp)
                t.set_last_frame(seq["T0"], seq["px0"], seq["f0"], idx, kf_slot=0)
            ctx.sync()
            t0 = time.perf_counter()
            t_copy = 0.0
            for k in range(1, len(seq["pyrs"])):
                tc0 = time.perf_counter()
                for b in bufs:
                    b[:] = seq["pyrs"][k][0]
                t_copy += time.perf_counter() - tc0                    # (the camera pipeline's write into the buffer: not the tracker's time)
                res = grp.track(bufs)
            if rep > 0:
                dt += time.perf_counter() - t0 - t_copy
            for c in range(n_cam):
                assert list(res[c].T_f_w) == list(want), "camera %d of %d differs from the lone tracker" % (c, n_cam)
        out[str(n_cam)] = n_cam * repeats * (len(seq["pyrs"]) - 1) / dt
        grp.destroy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--min-level", type=int, default=2)     # the shipping default: L4 -> L2
    ap.add_argument("--tracker-only", action="store_true")
    ap.add_argument("--cameras", action="store_true", help="only the multi-camera figure (1 / 2 / 4 / 8 trackers on as many host threads)")
    ap.add_argument("--camera-counts", default="1,2,4,8")
    ap.add_argument("--group-only", action="store_true", help="only svo_hip_tracker_group_track for --camera-counts (tools/trace_group.sh)")
    args = ap.parse_args()
    if args.group_only:
        seq = tc.make_sequence(n_frames=args.frames)
        print(json.dumps({"group_frames_per_s_by_cameras": group_cameras(hip.Context(0), seq, args.min_level, cams=tuple(int(c) for c in args.camera_counts.split(",")))}))
        return
    if args.cameras:
        seq = tc.make_sequence(n_frames=args.frames)
        print(json.dumps({"group_frames_per_s_by_cameras": group_cameras(hip.Context(0), seq, args.min_level, cams=tuple(int(c) for c in args.camera_counts.split(","))),
                          "min_level": args.min_level}))
        print(json.dumps({"chain_frames_per_s_by_cameras": multi_camera(seq, args.min_level, cams=tuple(int(c) for c in args.camera_counts.split(","))),
                          "min_level": args.min_level, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}))
        return
    if args.tracker_only:
        seq = tc.make_sequence(n_frames=args.frames)
        print(json.dumps({"hip_tracker": tracker_leg(hip.Context(0), seq, args.min_level)}))
        return
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_sequence import HipStages

    seq = tc.make_sequence(n_frames=args.frames)
    ctx = hip.Context(0)
    out = {"what": "per-frame time of SparseImgAlign -> reprojection (one match per cell) -> pose refinement over a %d-frame "
                   "synthetic sequence (640x480, %d map points, L4-L%d)" % (args.frames, len(seq["px0"]), args.min_level)}
    for name, make in (("hip", lambda: HipStages(ctx, seq)), ("hip_resident", lambda: ResidentStages(ctx, seq)),
                       ("cpu_oracle_1_thread", lambda: tc.OracleStages(seq))):
        inner = make()
        if name.startswith("hip"):
            tc.run_chain(seq, inner, args.min_level)          # warm-up (first launches, clocks)
        st = Timed(inner, ctx.sync if name.startswith("hip") else None)
        tc.run_chain(seq, st, args.min_level)
        n = st.calls
        out[name] = {"frames": n, "ms_per_frame": {k: 1e3 * v / n for k, v in st.t.items()},
                     "ms_per_frame_total": 1e3 * sum(st.t.values()) / n}
        if hasattr(inner, "destroy"):
            inner.destroy()
    out["hip_tracker"] = tracker_leg(ctx, seq, args.min_level)
    out["speedup"] = out["cpu_oracle_1_thread"]["ms_per_frame_total"] / out["hip_resident"]["ms_per_frame_total"]
    out["speedup_tracker"] = out["cpu_oracle_1_thread"]["ms_per_frame_total"] / out["hip_tracker"]["ms_per_frame_total"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
