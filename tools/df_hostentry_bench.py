#!/usr/bin/env python3
"""DepthFilter::updateSeeds through the HOST-BUFFER entry (svo_hip_depth_filter_update: what the drop-in DepthFilter calls per
keyframe sub-batch, include/svo_dropin/depth_filter_batch.h): seed arrays in from pageable memory, the pass, state and
per-seed outputs back, synchronised -- the PCIe-inclusive figure beside bench.py's resident one.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from android_svo_amd import hip, seedsynth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = hip.Context(0)
sc = seedsynth.make_seed_case(n_seeds=n, seed=9)
kf = hip.Pyramid(ctx, sc.cam.width, sc.cam.height, 5, 1)
cf = hip.Pyramid(ctx, sc.cam.width, sc.cam.height, 5, 1)
kf.upload(0, sc.ref_pyr); cf.upload(0, sc.cur_pyr)
cam = hip.make_camera(sc.cam)
prm = hip.depth_filter_params()
f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
px, f, lvl = f64(sc.px), f64(sc.f), np.ascontiguousarray(sc.level, dtype=np.int32)
Tr, Tc = f64(sc.T_ref_w), f64(sc.T_cur_w)
st = np.zeros(n, np.int32); z = np.zeros(n); xyz = np.zeros((n, 3)); nz = np.zeros(n, np.int32); na = np.zeros(n, np.int32)
pc = np.zeros((n, 2)); sl = np.zeros(n, np.int32)
P = lambda a, t: a.ctypes.data_as(C.POINTER(t))


def call():
    a, b, mu, s2 = sc.a.copy(), sc.b.copy(), sc.mu.copy(), sc.sigma2.copy()
    t0 = time.perf_counter()
    rc = ctx.lib.svo_hip_depth_filter_update(ctx.h, kf.h, 0, cf.h, 0, C.byref(cam), P(Tr, C.c_double), P(Tc, C.c_double), n, P(px, C.c_double),
                                             P(f, C.c_double), P(lvl, C.c_int32), P(a, C.c_float), P(b, C.c_float), P(mu, C.c_float),
                                             P(sc.z_range, C.c_float), P(s2, C.c_float), C.byref(prm), P(st, C.c_int32), P(z, C.c_double),
                                             P(xyz, C.c_double), P(nz, C.c_int32), P(na, C.c_int32), P(pc, C.c_double), P(sl, C.c_int32))
    dt = time.perf_counter() - t0
    ctx.check(rc, "depth_filter_update")
    return dt


t_w = time.perf_counter()
while time.perf_counter() - t_w < 0.1:
    call()
ts = sorted(call() for _ in range(30))
bytes_in, bytes_out = n * (16 + 24 + 4 + 20), n * (16 + 4 + 8 + 24 + 4 + 4 + 16 + 4)
print(json.dumps({"seeds": n, "what": "svo_hip_depth_filter_update: pageable host arrays in, DepthFilter::updateSeeds, state and outputs back, synchronised",
                  "ms_per_call_median": ts[len(ts) // 2] * 1e3, "ms_per_call_min": ts[0] * 1e3, "seeds_per_s": n / ts[len(ts) // 2],
                  "bytes_in": bytes_in, "bytes_out": bytes_out, "status_counts": np.bincount(st, minlength=6).tolist()}))
