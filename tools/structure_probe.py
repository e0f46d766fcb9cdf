#!/usr/bin/env python3
"""Diagnostic: wall time of svo_hip_tracker_track and of svo_hip_tracker_optimize_structure (20 points, 5 iterations, as
FrameHandlerMono::processFrame calls optimizeStructure behind the pose refinement) per frame of the 20-frame test sequence."""
import ctypes as C
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

ctx = hip.Context(0)
seq = tc.make_sequence(n_frames=20)
mp = tc.sequence_map(seq)
n = len(seq["px0"])
trk = hip.Tracker(ctx, seq["cam"], max_keyframes=2, grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=2, max_frame_features=1024)
trk.upload_keyframe(0, seq["pyrs"][0][0])
imgs = [np.ascontiguousarray(p[0]) for p in seq["pyrs"]]
res = hip.CTrackResult()
fp = np.zeros(1024, np.int32)
pos = np.zeros((64, 3))
it = np.zeros(64, np.int32)
tt, ts = [], []
for rep in range(4):
    trk.set_map(mp)
    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
    ctx.sync()
    for k in range(1, len(imgs)):
        t0 = time.perf_counter()
        rc = ctx.lib.svo_hip_tracker_track(trk.h, imgs[k].ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(res), None, None, None,
                                           fp.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None, None)
        t1 = time.perf_counter()
        ctx.check(rc, "track")
        sel = np.ascontiguousarray([p for p in fp[:res.n_features] if p >= 0][:20], dtype=np.int32)
        t2 = time.perf_counter()
        rc = ctx.lib.svo_hip_tracker_optimize_structure(trk.h, len(sel), sel.ctypes.data_as(C.POINTER(C.c_int32)), 5,
                                                        pos.ctypes.data_as(C.POINTER(C.c_double)), it.ctypes.data_as(C.POINTER(C.c_int32)))
        t3 = time.perf_counter()
        ctx.check(rc, "structure")
        if rep > 0:
            tt.append(t1 - t0)
            ts.append(t3 - t2)
print(json.dumps({"track_ms": float(np.mean(tt) * 1e3), "optimize_structure_ms": float(np.mean(ts) * 1e3), "iterations_of_the_first_points": it[:5].tolist()}))
