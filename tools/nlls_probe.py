"""Latency of svo_hip_sia_run through NLLSSolver's other branches (android_svo_amd/csrc/svo_nlls.hip) beside the default path,
one frame pair per call (C0: 200 patches, C1: 2000), L4..L0, n_iter 30, the reference's exits.  Median of 30 calls after 5
warm-ups, wall clock around run() + download().  Usage: python tools/nlls_probe.py > profiles/rNN_nlls_probe.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from android_svo_amd import hip, synth  # noqa: E402

CONFIGS = [
    ("gauss-newton (fused kernel)", {}),
    ("gauss-newton, streaming kernels", {hip.SIA_OPT_MODE: hip.SIA_MODE_STREAM}),
    ("gauss-newton, chi2 in the reference's order", {hip.SIA_OPT_CHI2: hip.SIA_CHI2_REFERENCE_ORDER}),
    ("gauss-newton + MAD / Tukey", {hip.SIA_OPT_SCALE_ESTIMATOR: hip.SIA_SCALE_MAD, hip.SIA_OPT_WEIGHT_FUNCTION: hip.SIA_WEIGHT_TUKEY}),
    ("gauss-newton + TDist / TDist", {hip.SIA_OPT_SCALE_ESTIMATOR: hip.SIA_SCALE_TDIST, hip.SIA_OPT_WEIGHT_FUNCTION: hip.SIA_WEIGHT_TDIST}),
    ("gauss-newton + Normal / Huber", {hip.SIA_OPT_SCALE_ESTIMATOR: hip.SIA_SCALE_NORMAL, hip.SIA_OPT_WEIGHT_FUNCTION: hip.SIA_WEIGHT_HUBER}),
    ("levenberg-marquardt", {hip.SIA_OPT_METHOD: hip.SIA_METHOD_LEVENBERG_MARQUARDT}),
    ("levenberg-marquardt + MAD / Huber", {hip.SIA_OPT_METHOD: hip.SIA_METHOD_LEVENBERG_MARQUARDT, hip.SIA_OPT_SCALE_ESTIMATOR: hip.SIA_SCALE_MAD,
                                           hip.SIA_OPT_WEIGHT_FUNCTION: hip.SIA_WEIGHT_HUBER}),
]


def main():
    ctx = hip.Context(0)
    for n, seed in ((200, 12345), (2000, 12346)):
        fp = synth.make_frame_pair(seed=seed, n_features=n)
        cam = fp.cam
        ref = hip.Pyramid(ctx, cam.width, cam.height, 5, 1)
        cur = hip.Pyramid(ctx, cam.width, cam.height, 5, 1)
        ref.upload(0, fp.ref_pyr); cur.upload(0, fp.cur_pyr)
        print("%d patches, %dx%d, L4..L0, n_iter 30" % (n, cam.width, cam.height))
        for name, opts in CONFIGS:
            sia = hip.SparseImgAlign(ctx, 1, n)
            sia.set_frames(ref, cur)
            sia.upload_pair(0, fp)
            for k, v in opts.items():
                sia.set_option(k, v)
            prm = sia.params()
            ts = []
            for it in range(35):
                t0 = time.perf_counter()
                sia.run(1, prm)
                r = sia.download(0)
                ts.append(time.perf_counter() - t0)
            ts = np.array(ts[5:]) * 1e6
            rot, trans = synth.pose_error(np.array(r.T_cur_w), fp.T_cur_w_true) if hasattr(fp, "T_cur_w_true") else (float("nan"), float("nan"))
            print("  %-46s %8.0f us median (%6.0f .. %6.0f)  evaluations/level %s  tracked %d" %
                  (name, np.median(ts), ts.min(), ts.max(), list(r.iters[:5]), r.n_tracked))
            sia.destroy()
        ref.destroy(); cur.destroy()
    ctx.close()


if __name__ == "__main__":
    main()
