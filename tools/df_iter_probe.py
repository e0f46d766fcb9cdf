#!/usr/bin/env python3
"""Diagnostic: stage times of the depth-filter pass against Matcher::Options::align_max_iter (0, 1, 2, 3, 5, 10): what the
alignment stage costs before its first iteration (records, patch words, gradients, H^-1) and per iteration.
  tools/df_iter_probe.py [seeds width height sigma_scale]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_c2  # noqa: E402
from android_svo_amd import hip, seedsynth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
ss = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
ctx = hip.Context(0)
sc = seedsynth.make_seed_case(n_seeds=n, seed=9, width=w, height=h)
kf = hip.Pyramid(ctx, w, h, 5, 1); cf = hip.Pyramid(ctx, w, h, 5, 1)
kf.upload(0, sc.ref_pyr); cf.upload(0, sc.cur_pyr)
sigma2 = sc.sigma2 if ss <= 0 else (sc.sigma2 * np.float32(ss)).astype(np.float32)
sb = hip.SeedBatch(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, sigma2)
state, state0 = hip.pack_seed_state(sb)
for it in (0, 1, 2, 3, 5, 10):
    prm = hip.depth_filter_params(align_max_iter=it)

    def run():
        ctx.check(ctx.lib.svo_hip_copy_d2d(ctx.h, C.c_void_p(state.ptr), C.c_void_p(state0.ptr), C.c_size_t(state.nbytes)), "d2d")
        hip.depth_filter_update(ctx, kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w, sb, prm)
    for _ in range(20):
        run()
    ctx.sync()
    st = bench_c2.stage_times(ctx, run, repeats=8)
    na = sb.n_align.download()
    print("align_max_iter %2d: align %.1f us (geometry %.1f search %.1f finalize %.1f)  mean iterations run %.2f, per 16-seed wave max %.2f" % (
        it, st["align"], st["geometry"], st["search"], st["finalize"], na.mean(), na[:len(na) // 16 * 16].reshape(-1, 16).max(1).mean()))
