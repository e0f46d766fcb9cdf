#!/usr/bin/env python3
"""CPU SparseImgAlign timings the way SURVEY.md 8(d) asks for them: one thread, median of 20 runs after 3 warm-ups,
C0 (~200 patches) and the C1 shape (2000 patches), reference early-stop semantics and fixed work (30 evaluations per
level), for the C restatement (port) and -- where oracle/_ref travelled -- the reference's own compiled code.
Prints one JSON object.  Host-only: needs no GPU."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from android_svo_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402
from oracle.ref import refpy  # noqa: E402


def median_ms(fn, runs=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(runs):
        t = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t) * 1e3)
    return statistics.median(ts)


def main():
    out = {"unit": "ms per frame pair, 1 thread, median of 20", "cpu": os.uname().machine, "threads": 1}
    have_ref = refpy.available()
    for name, seed, n in (("C0_200_patches", 12345, 200), ("C1_2000_patches", 12346, 2000)):
        fp = synth.make_frame_pair(seed=seed, n_features=n)
        row = {"port_early_stop": median_ms(lambda: orc.sparse_img_align(fp, n_iter=30, early_stop=True)),
               "port_fixed_work_150_evaluations": median_ms(lambda: orc.sparse_img_align(fp, n_iter=30, early_stop=False), runs=5, warm=1)}
        if have_ref:
            row["reference_early_stop"] = median_ms(lambda: refpy.sparse_img_align_run(fp, n_iter=30))
        row["evaluations_early_stop"] = int(sum(orc.sparse_img_align(fp, n_iter=30, early_stop=True).iters[:5]))
        out[name] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
