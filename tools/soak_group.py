#!/usr/bin/env python3
"""Diagnostic: a tracker group of N cameras over a few thousand frame-sets -- the 20-frame test sequence over and over, camera c
starting `c % 4` frames into it, a structure step on every camera per frame-set and a map upload per pass -- checked against lone
trackers doing the same (every camera's pose of every frame bit-equal), with the process's resident memory, the free device
memory and the context's allocator calls before and after: the per-frame-set path allocates nothing.
    python tools/soak_group.py [n_cameras=16] [passes=40]"""
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

_hiprt = C.CDLL("libamdhip64.so")


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


def free_dev_mb():
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert _hiprt.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value / 1e6


n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 16
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = hip.Context(0)
seq = tc.make_sequence(n_frames=24)
mp = tc.sequence_map(seq)
n = len(seq["px0"])
idx = np.arange(n, dtype=np.int32)
imgs = [np.ascontiguousarray(p[0]) for p in seq["pyrs"]]
cfg = dict(max_keyframes=2, max_points=1024, max_obs=1024, max_kf_features=1024, max_candidates=16, max_items=1024, max_frame_features=1024,
           grid_size=tc.CELL, max_fts=tc.MAX_FTS, klt_min_level=2)
STEPS = 20


def start(trk):
    trk.set_map(mp)
    trk.set_last_frame(seq["T0"], seq["px0"], seq["f0"], idx, kf_slot=0)


# ---- what a lone tracker gives for each of the four starting offsets (one pass: every pass repeats it)
want = []
lone = hip.Tracker(ctx, seq["cam"], **cfg)
lone.upload_keyframe(0, imgs[0])
for off in range(4):
    start(lone)
    poses = []
    for s in range(STEPS):
        r = lone.track(imgs[1 + off + s])
        lone.optimize_structure([int(p) for p in r["feat_point"] if p >= 0][:20], 5)
        poses.append(r["T_f_w"].tobytes())
    want.append(poses)
lone.destroy()

grp = hip.TrackerGroup(ctx, seq["cam"], n_cam, **cfg)
for t in grp.cameras:
    t.upload_keyframe(0, imgs[0])
out = {"cameras": n_cam, "passes": passes}
t0 = time.perf_counter()
for rep in range(passes):
    for t in grp.cameras:
        start(t)
    for s in range(STEPS):
        res = grp.track([imgs[1 + (c % 4) + s] for c in range(n_cam)])
        for c, t in enumerate(grp.cameras):
            assert bytes(np.array(res[c].T_f_w).tobytes()) == want[c % 4][s], "pass %d step %d camera %d differs from the lone tracker" % (rep, s, c)
            r = t.last_result()
            t.optimize_structure([int(p) for p in r["feat_point"] if p >= 0][:20], 5)
    if rep == 2:
        out["rss_mb_after_warmup"], out["free_device_mb_after_warmup"], out["allocator_calls_after_warmup"] = rss_mb(), free_dev_mb(), ctx.info()["allocator_calls"]
out["frame_sets"] = passes * STEPS
out["frames"] = passes * STEPS * n_cam
out["rss_mb_at_end"], out["free_device_mb_at_end"], out["allocator_calls_at_end"] = rss_mb(), free_dev_mb(), ctx.info()["allocator_calls"]
out["seconds"] = time.perf_counter() - t0
grp.destroy()
print(json.dumps(out))
