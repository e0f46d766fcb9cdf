#!/usr/bin/env python3
"""Diagnostic: a few thousand tracked frames (the 20-frame test sequence over and over, with a structure step per frame and a map
upload per pass) -- resident memory of the process and free device memory before and after: the per-frame path allocates
nothing."""
import ctypes as C
import json
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tracking_chain as tc  # noqa: E402
from android_svo_amd import hip  # noqa: E402


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


_hiprt = C.CDLL("libamdhip64.so")


def free_dev_mb():
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert _hiprt.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value / 1e6


ctx = hip.Context(0)
seq = tc.make_sequence(n_frames=20)
mp = tc.sequence_map(seq)
n = len(seq["px0"])
trk = hip.Tracker(ctx, seq["cam"], max_keyframes=2, grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=2, max_frame_features=1024)
trk.upload_keyframe(0, seq["pyrs"][0][0])
imgs = [np.ascontiguousarray(p[0]) for p in seq["pyrs"]]
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 150
out = {}
t0 = time.perf_counter()
for rep in range(passes):
    trk.set_map(mp)
    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], np.arange(n, dtype=np.int32), kf_slot=0)
    for k in range(1, len(imgs)):
        r = trk.track(imgs[k])
        sel = [int(p) for p in r["feat_point"] if p >= 0][:20]
        trk.optimize_structure(sel, 5)
    if rep == 4:
        out["rss_mb_after_warmup"], out["free_device_mb_after_warmup"] = rss_mb(), free_dev_mb()
out["frames"] = passes * (len(imgs) - 1)
out["rss_mb_at_end"], out["free_device_mb_at_end"] = rss_mb(), free_dev_mb()
out["seconds"] = time.perf_counter() - t0
print(json.dumps(out))
