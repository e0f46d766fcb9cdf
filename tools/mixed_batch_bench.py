#!/usr/bin/env python3
"""A 256-pair C1 batch (2000 patches each) against the same batch with ONE frame replaced by a frame of 12 or 5 patches:
svo_hip_sia_run launches the instance whose workgroups pick the entry-by-entry Hessian rows by their own patch count
(run_fused in svo_sia.hip), so the mixed batch must run within a few per cent of the pure one.  Diagnostic; prints one
JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from android_svo_amd import hip, synth  # noqa: E402


def main():
    B, N, steps = 256, 2000, 20
    ctx = hip.Context(0)
    fps = [synth.make_frame_pair(seed=12345 + i, n_features=N) for i in range(16)]
    tiny = synth.make_frame_pair(seed=999, n_features=5)
    cam = fps[0].cam
    ref = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
    cur = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
    sia = hip.SparseImgAlign(ctx, B, N)
    sia.set_frames(ref, cur)
    for s in range(B):
        fp = fps[s % len(fps)]
        ref.upload(s, fp.ref_pyr); cur.upload(s, fp.cur_pyr); sia.upload_pair(s, fp)
    out = {"what": "256 x C1 frame pairs (2000 patches); slot 100 replaced by a frame with few patches: svo_hip_sia_run launches the "
                   "instance whose workgroups choose the entry-by-entry Hessian rows by their own patch count"}

    def timed(prm):
        for _ in range(3):
            sia.run(B, prm)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            sia.run(B, prm)
        ctx.sync()
        return (time.perf_counter() - t0) / steps * 1e3
    prms = {"fixed_work": sia.params(early_stop=False), "reference_exits": sia.params(early_stop=True)}
    pure = {k: timed(p) for k, p in prms.items()}
    for n_tiny in (12, 5):
        tiny = synth.make_frame_pair(seed=999, n_features=n_tiny)
        ref.upload(100, tiny.ref_pyr); cur.upload(100, tiny.cur_pyr); sia.upload_pair(100, tiny)
        for k, p in prms.items():
            mixed = timed(p)
            out["%s_tiny%d" % (k, n_tiny)] = {"pure_ms": pure[k], "mixed_ms": mixed, "ratio": mixed / pure[k]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
