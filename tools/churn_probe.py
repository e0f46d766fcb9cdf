#!/usr/bin/env python3
"""SURVEY 8(b) "Threading": does the depth-filter thread's seed-batch traffic disturb the tracking thread?
Writes a small case (two frames, 800 alignment features, 3000 seed features) and runs `svo_host_demo <case> <out> churn` --
SparseImgAlign::run on the main thread while a second thread with its own context creates seed batches, runs a pass over
them and drops them at keyframe rate -- once against the library as built (per-context seed-batch pool: no allocator call
after warm-up) and, if build/nopool/libsvo_hip.so exists (the same sources with -DSVO_NO_SEED_POOL: every batch hipMalloc'ed
and hipFree'd, rounds 3-4), once against that.  Prints both; run on the GPU box from the repo root."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from android_svo_amd import synth  # noqa: E402

DEMO = os.path.join(ROOT, "android_svo_amd", "host", "svo_host_demo")


def write_case(case):
    rng = np.random.default_rng(4)
    cam = synth.Camera.default()
    scene = synth.PlaneScene(seed=4, depth=2.0, tilt=(0.08, 0.05))
    T0 = synth.se3_from_twist([0.02, -0.01, 0.0], [0.01, 0.005, -0.01])
    T1 = synth.se3_mul(synth.se3_from_twist(np.array([0.033, 0.01, 0.003]), rng.uniform(-0.004, 0.004, 3)), T0)
    w = lambda name, arr, dt: np.ascontiguousarray(arr, dtype=dt).tofile(os.path.join(case, name))
    w("manifest.bin", [cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, 5, 2], np.float64)
    for k, T in enumerate((T0, T1)):
        w("frame_%d_pose.bin" % k, T, np.float64)
        for l, img in enumerate(synth.build_pyramid(scene.render(cam, T))):
            w("frame_%d_L%d.bin" % (k, l), img, np.uint8)
    px = synth.grid_features(cam, 800, rng)
    w("sia_px.bin", px, np.float64); w("sia_f.bin", synth.cam2world(cam, px), np.float64)
    w("sia_pos.bin", scene.intersect(cam, T0, px[:, 0], px[:, 1]), np.float64); w("sia_has.bin", np.ones(len(px)), np.uint8)
    n = 3000
    spx = np.floor(np.stack([rng.uniform(40, cam.width - 40, n), rng.uniform(40, cam.height - 40, n)], axis=1))
    w("seed_px.bin", spx, np.float64); w("seed_f.bin", synth.cam2world(cam, spx), np.float64); w("seed_level.bin", np.zeros(n), np.int32)


def main():
    with tempfile.TemporaryDirectory() as tmp:
        case, out = os.path.join(tmp, "case"), os.path.join(tmp, "out")
        os.mkdir(case); os.mkdir(out)
        write_case(case)
        runs = [("pooled seed batches (the library as built)", None)]
        nopool = os.path.join(ROOT, "build", "nopool")
        if os.path.exists(os.path.join(nopool, "libsvo_hip.so")):
            runs.append(("every batch hipMalloc'ed / hipFree'd (build/nopool: -DSVO_NO_SEED_POOL)", nopool))
        for title, libdir in runs:
            env = dict(os.environ)
            if libdir:
                env["LD_LIBRARY_PATH"] = libdir + ":" + env.get("LD_LIBRARY_PATH", "")
            r = subprocess.run([DEMO, case, out, "churn"], capture_output=True, text=True, timeout=300, env=env)
            print("# " + title)
            print(r.stdout.strip() if r.returncode == 0 else "FAILED: " + r.stdout + r.stderr)


if __name__ == "__main__":
    main()
