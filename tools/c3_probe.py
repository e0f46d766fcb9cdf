#!/usr/bin/env python3
"""Diagnostic: the patch-sharded solve on ONE rank without RCCL in the way (shared-memory communicator of world size 1: the
exchange is a no-op), BASELINE config C3's shape (8 pairs of 1280x720, 2000 patches, 5 x 30 evaluations):
  run_sharded : svo_hip_sia_run_sharded (one launch per Gauss-Newton step: control step at the head, reduce rows at the tail)
  stepwise    : begin / level_begin / (accumulate, solve_update) x 30 / finish through the C-ABI (two launches per step)
  fused       : svo_hip_sia_run (frame-parallel fused kernel, no exchange) for reference"""
import json
import os
import sys
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from android_svo_amd import hip, synth  # noqa: E402

ctx = hip.Context(0)
B, N = 8, 2000
fps = [synth.make_frame_pair(seed=3300 + i, n_features=N, width=1280, height=720) for i in range(B)]
cam = fps[0].cam
ref = hip.Pyramid(ctx, cam.width, cam.height, 5, B); cur = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
sia = hip.SparseImgAlign(ctx, B, N)
sia.set_frames(ref, cur)
for s, fp in enumerate(fps):
    ref.upload(s, fp.ref_pyr); cur.upload(s, fp.cur_pyr); sia.upload_pair(s, fp)
prm = sia.params(early_stop=False)
comm = hip.Comm(ctx, 0, 1, kind="shm", name="/svo_probe_" + uuid.uuid4().hex[:10], slot_bytes=1 << 16)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


def stepwise():
    sia.begin(B, prm)
    for level in range(4, -1, -1):
        sia.level_begin(level)
        for _ in range(prm.n_iter):
            sia.accumulate()
            sia.solve_update()
    sia.finish()


out = {"run_sharded_ms": timed(lambda: hip.sia_run_sharded(sia, comm, B, prm))}
a = [list(r.T_cur_w) for r in sia.download_all(B)]
out["stepwise_ms"] = timed(stepwise)
b = [list(r.T_cur_w) for r in sia.download_all(B)]
out["bitwise_equal"] = a == b
out["fused_frame_parallel_ms"] = timed(lambda: sia.run(B, prm))
print(json.dumps(out))
