#!/usr/bin/env python3
"""How far the HIP tracking chain and the CPU oracle's chain drift apart over a LONG sequence (Next 8c): both
solvers of a frame leave on the first error increase with a rollback, so a frame's final pose depends on the pose it
started from, and any two implementations that differ in the last bit of one sum separate along the chain.  200 frames
of a camera that swings around the keyframe's viewpoint, each chain feeding on its own outputs (svo_hip_tracker_track on
the GPU; tests/tracking_chain.py: oracle_track_frame on one host thread).  Prints one JSON line with the curve.

    python tools/long_chain.py [--frames 200] [--min-level 2]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tracking_chain as tc  # noqa: E402
import chain_bench  # noqa: E402
from android_svo_amd import hip, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--min-level", type=int, default=2)
    ap.add_argument("--max-fts", type=int, default=1200,
                    help="Config::maxFts().  The 20-frame test chain uses 120 with 352 grid cells: the cell loop then stops after the top third of "
                         "the image (grid_.cell_order is the identity in this port, reprojector.cpp:54), the pose is weakly constrained and BOTH "
                         "chains lose track after ~30 frames; the reference's own default, 1200, covers every cell")
    args = ap.parse_args()
    seq = tc.make_sequence(n_frames=args.frames, orbit=True)
    ctx = hip.Context(0)
    g_poses, g_win = chain_bench.tracker_poses(ctx, seq, args.min_level, max_fts=args.max_fts, stop_when_lost=True)
    _, c_poses, c_win = chain_bench.oracle_tracker_leg(seq, args.min_level, max_fts=args.max_fts, stop_when_lost=True)
    lost = {"hip_tracked_frames": len(g_poses), "cpu_tracked_frames": len(c_poses), "frames_in_sequence": args.frames - 1}
    m = min(len(g_poses), len(c_poses))
    g_poses, c_poses, g_win, c_win = g_poses[:m], c_poses[:m], g_win[:m], c_win[:m]
    truth = seq["truth"][1:m + 1]
    diff = np.array([synth.pose_error(a, b) for a, b in zip(g_poses, c_poses)])
    err_g = np.array([synth.pose_error(a, t) for a, t in zip(g_poses, truth)])
    err_c = np.array([synth.pose_error(a, t) for a, t in zip(c_poses, truth)])
    same = [bool(np.array_equal(a, b)) for a, b in zip(g_win, c_win)]
    first_diff = same.index(False) + 1 if False in same else None
    every = list(range(0, len(diff), 10)) + [len(diff) - 1]
    print(json.dumps({
        "what": "HIP tracking chain vs CPU oracle chain over %d frames (L4-L%d, maxFts %d), each feeding on its own outputs" % (len(diff), args.min_level, args.max_fts),
        "tolerance": "1e-4 rad / 1e-3 m (north_star)",
        "tracking": dict(lost, note="a chain ends at the frame for which processFrame would return RESULT_FAILURE (fewer than 50 matches, or fewer "
                                    "than 20 observations left by the pose refinement): the reference's own behaviour on this synthetic walk, CPU "
                                    "oracle and HIP alike"),
        "max_diff": {"rot_rad": float(diff[:, 0].max()), "trans_m": float(diff[:, 1].max())},
        "diff_curve_every_10_frames": [{"frame": k + 1, "rot_rad": float(diff[k, 0]), "trans_m": float(diff[k, 1])} for k in every],
        "matched_points_equal_until_frame": first_diff or len(diff),
        "frames_with_equal_matches": int(sum(same)),
        "max_err_vs_ground_truth": {"hip": {"rot_rad": float(err_g[:, 0].max()), "trans_m": float(err_g[:, 1].max())},
                                    "cpu": {"rot_rad": float(err_c[:, 0].max()), "trans_m": float(err_c[:, 1].max())}}}))


if __name__ == "__main__":
    main()
