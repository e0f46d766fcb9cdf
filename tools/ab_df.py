#!/usr/bin/env python3
"""One line per call: the depth-filter pass at config C2's size and at config C4's on one GPU, as bench.py measures them
(bench_c2.measure_depth_filter), + align2D x 5000 / x 200000 -- for same-box A/B runs of library builds (tools/ab_df.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_c2  # noqa: E402
from android_svo_amd import hip  # noqa: E402

ctx = hip.Context(0)


def stages(res):
    st = res.get("stages_us")          # HIP events between the stages (bench_c2.stage_times), where the library has the hook
    return " [geo %.1f search %.1f align %.1f fin %.1f]" % (st["geometry"], st["search"], st["align"], st["finalize"]) if st else ""


a5, _ = bench_c2.measure_align2d(ctx, 5000, steps=20, warmup=3)
a200, _ = bench_c2.measure_align2d(ctx, 200000, steps=20, warmup=3)
import hashlib


def digest(sb):
    """the seeds' state and outcome after the timed passes: equal between two builds that compute the same thing"""
    h = hashlib.sha256()
    for arr in (sb.a, sb.b, sb.mu, sb.sigma2, sb.status, sb.px_cur):
        h.update(arr.download().tobytes())
    return h.hexdigest()[:12]


c2, sc, sb, pyr = bench_c2.measure_depth_filter(ctx, 100000, steps=20, warmup=3)
st2 = stages(c2)
d2 = digest(sb)
sb.free(); [p.destroy() for p in pyr]
c4, sc, sb, pyr = bench_c2.measure_depth_filter(ctx, 1000000, steps=10, warmup=2, width=1280, height=720, sigma_scale=0.0045, compact=True)
st4 = stages(c4)
d4 = digest(sb)
print("%-34s align2D 5k %.1f us  200k %.1f us | C2 %.1f us%s | C4 %.1f us%s (%d packed) | state sha %s %s" % (
    os.environ.get("SVO_HIP_LIB", "default")[-34:], a5["us_per_batch"], a200["us_per_batch"], c2["us_per_frame"], st2, c4["us_per_frame"], st4,
    c4["converged_records_packed"], d2, d4))
