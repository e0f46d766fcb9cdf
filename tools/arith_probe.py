#!/usr/bin/env python3
"""The arithmetic levels of the fused SparseImgAlign kernel (svo_hip_sia_set_option(SVO_HIP_SIA_OPT_ARITH, ...)) beside each
other on 64 scenes of config C1: the reference's Gauss-Newton exits and fixed work; per level the largest pose distance to the
CPU oracle, and against the EXACT level: H_ bitwise, iteration counts, n_tracked.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from android_svo_amd import hip, synth  # noqa: E402
from oracle import orc  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = hip.Context(0)
fps = [synth.make_frame_pair(seed=12345 + i, n_features=2000) for i in range(n_scenes)]
cam = fps[0].cam
ref = hip.Pyramid(ctx, cam.width, cam.height, 5, n_scenes)
cur = hip.Pyramid(ctx, cam.width, cam.height, 5, n_scenes)
sia = hip.SparseImgAlign(ctx, n_scenes, 2000)
sia.set_frames(ref, cur)
for s, fp in enumerate(fps):
    ref.upload(s, fp.ref_pyr); cur.upload(s, fp.cur_pyr); sia.upload_pair(s, fp)
out = {}
for es in (True, False):
    prm = sia.params(max_level=4, min_level=0, n_iter=30, eps=1e-6, early_stop=es)
    oracle = [orc.sparse_img_align(fp, n_iter=30, early_stop=es) for fp in fps]
    res = {}
    for name, level in (("exact", hip.SIA_ARITH_EXACT), ("moments_f32", hip.SIA_ARITH_MOMENTS_F32), ("fast", hip.SIA_ARITH_FAST)):
        sia.set_option(hip.SIA_OPT_ARITH, level)
        sia.run(n_scenes, prm)
        res[name] = sia.download_all(n_scenes)
    sia.set_option(hip.SIA_OPT_ARITH, hip.SIA_ARITH_EXACT)
    mode = "reference_exits" if es else "fixed_work"
    out[mode] = {}
    for name, r in res.items():
        err = np.array([synth.pose_error(np.array(r[i].T_cur_w), np.array(oracle[i].T_cur_w)) for i in range(n_scenes)])
        ex = res["exact"]
        out[mode][name] = {
            "max_rot_rad_vs_cpu": float(err[:, 0].max()), "max_trans_m_vs_cpu": float(err[:, 1].max()),
            "n_tracked_equal_to_cpu": bool(all(int(r[i].n_tracked) == int(oracle[i].n_tracked) for i in range(n_scenes))),
            "scenes_with_H_bitwise_equal_to_exact": int(sum(list(r[i].H) == list(ex[i].H) for i in range(n_scenes))),
            "scenes_with_iteration_counts_equal_to_exact": int(sum(list(r[i].iters) == list(ex[i].iters) for i in range(n_scenes))),
            "scenes_with_iteration_counts_equal_to_cpu": int(sum(list(r[i].iters)[:5] == list(oracle[i].iters)[:5] for i in range(n_scenes)))}
out["scenes"] = n_scenes
print(json.dumps(out))
