#!/usr/bin/env python3
"""Per-call time of the host-buffer entry points the C++ drop-in layer uses once per frame (pageable arguments in,
results out, synchronised): pose refinement and the reprojection cell loop."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from android_svo_amd import hip, synth  # noqa: E402

ctx = hip.Context(0)
pc = synth.make_pose_opt_case(seed=40, n=1200)
em = abs(pc.cam.fx)
for _ in range(5):
    hip.pose_optimize(ctx, pc.T_f_w_init, pc.f, pc.pos, pc.level, pc.has_point, em)
t0 = time.perf_counter()
for _ in range(100):
    hip.pose_optimize(ctx, pc.T_f_w_init, pc.f, pc.pos, pc.level, pc.has_point, em)
print("svo_hip_pose_optimize (1200 observations, host buffers): %.1f us per call" % ((time.perf_counter() - t0) / 100 * 1e6))

# one tracked frame as the drop-in SparseImgAlign::run does it: features + poses in, solve, result out (pyramids resident)
fp = synth.make_frame_pair(seed=3, n_features=200)
ref = hip.Pyramid(ctx, fp.cam.width, fp.cam.height, 5, 1); cur = hip.Pyramid(ctx, fp.cam.width, fp.cam.height, 5, 1)
ref.upload(0, fp.ref_pyr); cur.upload(0, fp.cur_pyr)
sia = hip.SparseImgAlign(ctx, 1, 200); sia.set_frames(ref, cur)
prm = sia.params(early_stop=True)
def frame():
    sia.upload_pair(0, fp); sia.run(1, prm); return sia.download(0)
for _ in range(5): frame()
t0 = time.perf_counter()
for _ in range(100): frame()
t_all = (time.perf_counter() - t0) / 100 * 1e6
sia.upload_pair(0, fp)
t0 = time.perf_counter()
for _ in range(100): sia.run(1, prm); ctx.sync()
t_run = (time.perf_counter() - t0) / 100 * 1e6
t0 = time.perf_counter()
for _ in range(100): sia.upload_pair(0, fp); ctx.sync()
t_up = (time.perf_counter() - t0) / 100 * 1e6
print("SparseImgAlign, 200 features, early exits: upload + run + download %.1f us per frame (run alone %.1f, upload alone %.1f)" % (t_all, t_run, t_up))
