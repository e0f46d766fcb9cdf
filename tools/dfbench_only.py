#!/usr/bin/env python3
"""Diagnostic: only the depth-filter pass of bench_c2.py / bench_c4.py, for rocprofv3 passes over its kernels.
  tools/dfbench_only.py [seeds] [width height] [sigma_scale] [passes]
defaults = config C2 (100 000 seeds, 640x480, fresh seeds); `1000000 1280 720 0.0045` = config C4 on one GPU."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from android_svo_amd import hip, seedsynth
ctx = hip.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
ss = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
passes = int(sys.argv[5]) if len(sys.argv) > 5 else 4
sc = seedsynth.make_seed_case(n_seeds=n, seed=9, width=w, height=h)
kf = hip.Pyramid(ctx, w, h, 5, 1); cf = hip.Pyramid(ctx, w, h, 5, 1)
kf.upload(0, sc.ref_pyr); cf.upload(0, sc.cur_pyr)
sigma2 = sc.sigma2 if ss <= 0.0 else (sc.sigma2 * np.float32(ss)).astype(np.float32)
sb = hip.SeedBatch(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, sigma2)
state, state0 = hip.pack_seed_state(sb)
for _ in range(passes):
    ctx.check(ctx.lib.svo_hip_copy_d2d(ctx.h, C.c_void_p(state.ptr), C.c_void_p(state0.ptr), C.c_size_t(state.nbytes)), "d2d")
    hip.depth_filter_update(ctx, kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w, sb)
ctx.sync()
