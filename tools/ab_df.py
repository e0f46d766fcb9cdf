#!/usr/bin/env python3
"""One line per call: the depth-filter pass at config C2's size and at config C4's on one GPU, as bench.py measures them
(bench_c2.measure_depth_filter), + align2D x 5000 / x 200000 -- for same-box A/B runs of library builds (tools/ab_df.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_c2  # noqa: E402
from android_svo_amd import hip  # noqa: E402

import ctypes as C  # noqa: E402

ctx = hip.Context(0)


def stages(sc, sb, kf, cf):
    """stage durations (geometry, search, align, finalize) of one more pass, where the library has the hook"""
    try:
        ctx.lib.svo_hip_df_set_profiling
    except AttributeError:
        return ""
    ctx.check(ctx.lib.svo_hip_df_set_profiling(ctx.h, 1), "df_set_profiling")
    best = None
    for _ in range(5):
        hip.depth_filter_update(ctx, kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w, sb)
        us = (C.c_double * 4)()
        ctx.check(ctx.lib.svo_hip_df_get_profile(ctx.h, us), "df_get_profile")
        best = list(us) if best is None or sum(us) < sum(best) else best
    ctx.check(ctx.lib.svo_hip_df_set_profiling(ctx.h, 0), "df_set_profiling")
    return " [geo %.1f search %.1f align %.1f fin %.1f]" % tuple(best)


a5, _ = bench_c2.measure_align2d(ctx, 5000, steps=20, warmup=3)
a200, _ = bench_c2.measure_align2d(ctx, 200000, steps=20, warmup=3)
c2, sc, sb, pyr = bench_c2.measure_depth_filter(ctx, 100000, steps=20, warmup=3)
st2 = stages(sc, sb, *pyr)
sb.free(); [p.destroy() for p in pyr]
c4, sc, sb, pyr = bench_c2.measure_depth_filter(ctx, 1000000, steps=10, warmup=2, width=1280, height=720, sigma_scale=0.0045, compact=True)
st4 = stages(sc, sb, *pyr)
print("%-34s align2D 5k %.1f us  200k %.1f us | C2 %.1f us%s | C4 %.1f us%s (%d packed)" % (
    os.environ.get("SVO_HIP_LIB", "default")[-34:], a5["us_per_batch"], a200["us_per_batch"], c2["us_per_frame"], st2, c4["us_per_frame"], st4,
    c4["converged_records_packed"]))
