#!/usr/bin/env python3
"""Diagnostic: device time of pose_refine_kernel for one frame of n observations as a function of the iteration budget
(the slope is one robust Gauss-Newton step, the intercept the scale estimate + covariance + outlier test + medians)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from android_svo_amd import hip, synth  # noqa: E402

ctx = hip.Context(0)
out = {}
for n in (121, 1200):
    pc = synth.make_pose_opt_case(seed=40, n=n)
    em = abs(pc.cam.fx)
    d = [ctx.to_device(hip._f64(pc.T_f_w_init[None, :])), ctx.to_device(hip._f64(pc.f)), ctx.to_device(hip._f64(pc.pos)),
         ctx.to_device(np.ascontiguousarray(pc.level, dtype=np.int32)), ctx.to_device(np.ascontiguousarray(pc.has_point, dtype=np.uint8)),
         ctx.to_device(np.full(1, n, dtype=np.int32)), ctx.to_device(np.ascontiguousarray(pc.has_point, dtype=np.uint8))]
    dres = ctx.empty((C.sizeof(hip.CPoseOptResult),), np.uint8)
    for n_iter in (0, 1, 2, 5, 10):
        def run():
            ctx.check(ctx.lib.svo_hip_copy_d2d(ctx.h, C.c_void_p(d[4].ptr), C.c_void_p(d[6].ptr), C.c_size_t(n)), "restore")
            ctx.check(ctx.lib.svo_hip_pose_optimize_batch_dev(ctx.h, 1, n, C.c_void_p(d[5].ptr), C.c_void_p(d[0].ptr), C.c_void_p(d[1].ptr),
                                                              C.c_void_p(d[2].ptr), C.c_void_p(d[3].ptr), C.c_void_p(d[4].ptr), C.c_double(em),
                                                              C.c_double(2.0), n_iter, C.c_void_p(dres.ptr)), "pose_optimize")
        for _ in range(5):
            run()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(200):
            run()
        ctx.sync()
        dt = (time.perf_counter() - t0) / 200 * 1e6
        res = hip.CPoseOptResult.from_buffer_copy(dres.download().tobytes())
        out["n%d_iter%d" % (n, n_iter)] = {"us_per_call_incl_restore_copy": round(dt, 2), "n_iter_done": res.n_iter_done}
print(json.dumps(out, indent=0))
