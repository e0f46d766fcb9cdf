#!/usr/bin/env python3
"""Diagnostic (not part of the product or of bench.py): run the fused SparseImgAlign kernel from a build with
-DSVO_STAMPS (`make -C android_svo_amd/csrc stamps` -> build/libsvo_hip_stamps.so) and print where wave 1 of each workgroup spends its cycles per
Gauss-Newton evaluation: evaluation+wave reduction / first barrier wait / sum over the waves + one-lane solve + barrier.

Environment: SVO_STAMPS_SCENES (distinct synthetic scenes tiled over the batch, default 4; bench.py uses 64),
SVO_STAMPS_FEATURES (2000), SVO_STAMPS_RUNS (launches before the one that is read: the clock settles after ~100),
SVO_HIP_STAMPS_LIB (another stamps build), and for more output SVO_STAMPS_BLOCKS=1 (workgroup cycles per scene: the launch
is as slow as its slowest scene), SVO_STAMPS_COUNTS=1 (per scene: re-factorisations of H, tile rows corrected, patches
outside the image at those corrections), SVO_STAMPS_XCD=1 (workgroup cycles by XCD), SVO_STAMPS_LEVELS=1 (cycles per
evaluation by pyramid level)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from android_svo_amd import hip, synth  # noqa: E402

hip.LIB_PATH = os.environ.get("SVO_HIP_STAMPS_LIB") or os.path.join(ROOT, "build", "libsvo_hip_stamps.so")
ctx = hip.Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NS = int(os.environ.get("SVO_STAMPS_SCENES", "4"))
fps = [synth.make_frame_pair(seed=12345 + i, n_features=int(os.environ.get("SVO_STAMPS_FEATURES", "2000"))) for i in range(NS)]
cam = fps[0].cam
ref = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
cur = hip.Pyramid(ctx, cam.width, cam.height, 5, B)
sia = hip.SparseImgAlign(ctx, B, int(os.environ.get("SVO_STAMPS_FEATURES", "2000")))
sia.set_frames(ref, cur)
for s in range(B):
    ref.upload(s, fps[s % NS].ref_pyr); cur.upload(s, fps[s % NS].cur_pyr); sia.upload_pair(s, fps[s % NS])
prm = sia.params(early_stop=False)
for _ in range(int(os.environ.get("SVO_STAMPS_RUNS", "2"))):
    sia.run(B, prm)
ctx.sync()
L = ctx.lib
import ctypes as C
# FrameState.x lives inside the download struct? not exposed: read through a dedicated debug accessor is overkill,
# the stamps are returned in H[...]? -> the stamped build stores them in x[0..2]; fetch via svo_hip_sia_download_x
buf = (C.c_double * 27)()
rows = []
for s in range(min(B, 8)):
    L.svo_hip_sia_debug_x(sia.h, s, buf)
    rows.append([buf[0], buf[1], buf[2], buf[3], buf[4]])
    clk = buf[5]
rows = np.array(rows) / 150.0
print("core clock over the kernel (s_memtime / s_memrealtime): %.3f GHz" % clk)
print("cycles per evaluation: eval+wave-reduce, barrier wait, sum+solve+barrier, of which LDLT, exp+mul")
print(rows)
print("per wave: evaluation cycles (waves 0..7), then wait at the first barrier (waves 0..7); frame 0")
L.svo_hip_sia_debug_x(sia.h, 0, buf)
print(np.array(list(buf[6:14])) / 150.0)
print(np.array(list(buf[14:22])) / 150.0)
print("wave 0 between the barriers: sum+compare+readlane, (solve), solve->exp, (exp+mul), after exp, whole")
print(np.array(list(buf[22:26])) / 150.0)
print("precompute + its barrier, cycles per level (wave 1):", buf[26] / 5.0, " of which interpolation+W store+gradient sums:", buf[23] / 5.0, " H rows:", buf[24] / 5.0)
if os.environ.get("SVO_STAMPS_COUNT"):
    vals = []
    for s in range(B):
        L.svo_hip_sia_debug_x(sia.h, s, buf)
        vals.append(buf[24] / 1e6)
    print("refactorisations per frame pair (stamp[7] / 1e6): mean %.1f min %.1f max %.1f" % (np.mean(vals), np.min(vals), np.max(vals)))
if os.environ.get("SVO_STAMPS_LEVELS"):
    print("whole evaluation (eval + barriers + solve) cycles per evaluation by level 0..4 (wave 1, frame 0):",
          np.array(list(sia.download(0).H[21:26])) / 30.0)
if os.environ.get("SVO_STAMPS_BLOCKS"):
    cyc = np.array([sia.download(s).chi2 for s in range(B)])
    per_scene = cyc[:NS]
    order = np.argsort(per_scene)
    print("workgroup cycles over the %d scenes: min %.0f  median %.0f  max %.0f   slowest scenes %s" %
          (NS, per_scene.min(), np.median(per_scene), per_scene.max(), [(int(i), int(per_scene[i])) for i in order[-4:]]))
if os.environ.get("SVO_STAMPS_COUNTS"):
    rows = []
    for s_ in range(NS):
        r_ = sia.download(s_)
        rows.append((s_, int(r_.chi2), int(r_.H[30]), int(r_.H[31]), int(r_.H[32])))
    rows.sort(key=lambda t: t[1])
    print("scene, workgroup cycles, re-factorisations of H, tile rows corrected, gone patches walked (per frame pair):")
    for t in rows[:3] + rows[len(rows) // 2:len(rows) // 2 + 1] + rows[-6:]:
        print("  ", t)
if os.environ.get("SVO_STAMPS_XCD"):
    cyc = np.array([sia.download(s).chi2 for s in range(B)])
    print("mean workgroup cycles by slot %% 8 (workgroups go round the 8 XCDs): %s" % np.round(cyc.reshape(-1, 8).mean(axis=0) / 1e6, 3))
    print("mean workgroup cycles by scene group of 8 (slot %% 64 // 8): %s" % np.round(cyc.reshape(-1, 64).mean(axis=0).reshape(8, 8).mean(axis=1) / 1e6, 3))
