#!/usr/bin/env python3
"""Per-level and per-evaluation cost of the fused SparseImgAlign kernel: the default bench workload (640x480, 2000 patches,
L4-L0, fixed work) with n_iter = 1, 2, 5, 15, 30 evaluations per level; a line fit of the launch time gives the cost of one
evaluation and of one level's set-up (reference patches, per-tile Hessian rows, level change).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from android_svo_amd import hip, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = hip.Context(0)
fps = [synth.make_frame_pair(seed=12345 + i, n_features=2000) for i in range(16)]
cam = fps[0].cam
ref = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
cur = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
sia = hip.SparseImgAlign(ctx, B, 2000)
sia.set_frames(ref, cur)
for s in range(B):
    fp = fps[s % len(fps)]
    ref.upload(s, fp.ref_pyr); cur.upload(s, fp.cur_pyr); sia.upload_pair(s, fp)
out = {"frame_pairs_per_launch": B, "ms_per_launch": {}}
for arith, tag in ((hip.SIA_ARITH_EXACT, "exact"), (hip.SIA_ARITH_FAST, "fast")):
    sia.set_option(hip.SIA_OPT_ARITH, arith)
    ks, ts = [], []
    for k in (1, 2, 5, 15, 30):
        prm = sia.params(max_level=4, min_level=0, n_iter=k, eps=1e-6, early_stop=False)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:
            sia.run(B, prm); ctx.sync()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                sia.run(B, prm)
            ctx.sync()
            dt = (time.perf_counter() - t0) / 10
            best = dt if best is None or dt < best else best
        ks.append(5 * k); ts.append(best * 1e3)
    a, c = np.polyfit(ks, ts, 1)
    per_cu = 256.0 / B                       # workgroups a CU runs one after the other per launch
    out["ms_per_launch"][tag] = dict(zip(map(str, ks), ts))
    out[tag] = {"us_per_evaluation_per_workgroup": a * 1e3 * per_cu, "us_per_level_per_workgroup": c * 1e3 * per_cu / 5,
                "level_setup_in_evaluations": (c / 5) / a, "share_of_fixed_work_launch": c / ts[-1]}
print(json.dumps(out))
