#!/usr/bin/env python3
"""Diagnostic: time svo_hip_sia_run on the C1 batch for one or more builds of the library, no parity check
(for experimental builds whose results are deliberately wrong).  Usage: tools/ktime.py lib1.so [lib2.so ...]
Each library is timed in its own child process."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] != "--child"):
    for lib in sys.argv[1:]:
        env = dict(os.environ, SVO_HIP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(lib, out.stdout.strip() or out.stderr.strip()[-300:])
    sys.exit(0)

sys.path.insert(0, ROOT)
from android_svo_amd import hip, synth  # noqa: E402

ctx = hip.Context(0)
B = int(os.environ.get("KT_B", "256"))
N = int(os.environ.get("KT_N", "2000"))
fps = [synth.make_frame_pair(seed=12345 + i, n_features=N) for i in range(4)]
cam = fps[0].cam
ref = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
cur = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
sia = hip.SparseImgAlign(ctx, B, N)
sia.set_frames(ref, cur)
for s in range(B):
    ref.upload(s, fps[s % 4].ref_pyr); cur.upload(s, fps[s % 4].cur_pyr); sia.upload_pair(s, fps[s % 4])
prm = sia.params(early_stop=False)
for _ in range(3):
    sia.run(B, prm)
ctx.sync()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        sia.run(B, prm)
    ctx.sync()
    best = min(best, (time.perf_counter() - t0) / 10)
print("%.4f ms/launch  %.0f frames/s" % (best * 1e3, B / best))
