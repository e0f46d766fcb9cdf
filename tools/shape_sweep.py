#!/usr/bin/env python3
"""Diagnostic: the timed C1 workload (1024 pairs, fixed work) of the fused kernel at the default arithmetic level over the shape
options (SVO_HIP_SIA_OPT_EXTRA_LDS: waves with a third tile in LDS; SVO_HIP_SIA_OPT_OLD_TILES: tiles of the older wave of a
SIMD): were the automatic choices, tuned on the EXACT instance (20 spilled VGPRs), still the best for MOMENTS_F32 (10)?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from android_svo_amd import hip, synth  # noqa: E402

ctx = hip.Context(0)
B, n_scenes = 1024, 16
fps = [synth.make_frame_pair(seed=12345 + i, n_features=2000) for i in range(n_scenes)]
cam = fps[0].cam
ref = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
cur = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
sia = hip.SparseImgAlign(ctx, B, 2000)
sia.set_frames(ref, cur)
for s in range(B):
    fp = fps[s % n_scenes]
    ref.upload(s, fp.ref_pyr); cur.upload(s, fp.cur_pyr); sia.upload_pair(s, fp)
prm = sia.params(max_level=4, min_level=0, n_iter=30, eps=1e-6, early_stop=False)


def rate(steps=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        sia.run(B, prm); ctx.sync()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            sia.run(B, prm)
        ctx.sync()
        best = max(best, B * steps / (time.perf_counter() - t0))
    return best


base = rate()
print("automatic: %.1f k frames/s" % (base / 1e3))
for old in (0, 3, 4, 5):
    for extra in (-1, 0, 1, 2, 3):
        if old == 0 and extra == -1:
            continue
        try:
            sia.set_option(hip.SIA_OPT_OLD_TILES, old)
            sia.set_option(hip.SIA_OPT_EXTRA_LDS, extra)
            r = rate()
            print("old_tiles %d extra_lds %2d: %.1f k (%+.1f %%)" % (old, extra, r / 1e3, 100 * (r / base - 1)))
        except hip.SvoHipError as e:
            print("old_tiles %d extra_lds %2d: refused (%s)" % (old, extra, str(e)[:60]))
sia.set_option(hip.SIA_OPT_OLD_TILES, 0); sia.set_option(hip.SIA_OPT_EXTRA_LDS, -1)
print("automatic again: %.1f k" % (rate() / 1e3))
