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
