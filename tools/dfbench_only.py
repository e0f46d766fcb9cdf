#!/usr/bin/env python3
"""Diagnostic: only the depth-filter pass of bench_c2.py (100 k seeds), for PMC passes over its kernels."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from android_svo_amd import hip, seedsynth
ctx = hip.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sc = seedsynth.make_seed_case(n_seeds=n, seed=9)
kf = hip.Pyramid(ctx, 640, 480, 5, 1); cf = hip.Pyramid(ctx, 640, 480, 5, 1)
kf.upload(0, sc.ref_pyr); cf.upload(0, sc.cur_pyr)
sb = hip.SeedBatch(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, sc.sigma2)
state, state0 = hip.pack_seed_state(sb)
for _ in range(4):
    ctx.check(ctx.lib.svo_hip_copy_d2d(ctx.h, C.c_void_p(state.ptr), C.c_void_p(state0.ptr), C.c_size_t(state.nbytes)), "d2d")
    hip.depth_filter_update(ctx, kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w, sb)
ctx.sync()
