"""Stage times (HIP events between the launches) of the grouped depth-filter pass at the reference's sizes."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from android_svo_amd import hip, seedsynth  # noqa: E402

ctx = hip.Context(0)
for n_b, n_s in ((4, 500), (1, 500), (1, 2000), (8, 2000)):
    mk = seedsynth.make_multi_keyframe_case((n_s,) * n_b, seed=9)
    kf = hip.Pyramid(ctx, 640, 480, 5, n_b)
    cf = hip.Pyramid(ctx, 640, 480, 5, 1)
    cf.upload(0, mk.cur_pyr)
    for k, sc in enumerate(mk.keyframes):
        kf.upload(k, sc.ref_pyr)
    rs = [hip.ResidentSeeds(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, sc.sigma2) for sc in mk.keyframes]
    T_refs = np.stack([sc.T_ref_w for sc in mk.keyframes])
    slots = list(range(n_b))

    def grouped():
        hip.ResidentSeeds.update_group_async(rs, kf, slots, cf, 0, mk.cam, T_refs, mk.T_cur_w)
        for r in rs:
            r.collect_raw()

    for _ in range(20):
        grouped()
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); grouped(); ts.append(time.perf_counter() - t0)
    ctx.check(ctx.lib.svo_hip_df_set_profiling(ctx.h, 1), "prof")
    acc = []
    for _ in range(30):
        grouped()
        us = (C.c_double * 4)()
        ctx.check(ctx.lib.svo_hip_df_get_profile(ctx.h, us), "get")
        acc.append(list(us))
    ctx.check(ctx.lib.svo_hip_df_set_profiling(ctx.h, 0), "prof")
    med = np.median(np.array(acc), axis=0)
    print("%d x %d: frame %.1f us | geometry %.1f search %.1f align %.1f finalize %.1f (sum %.1f; the rest: two event launches + wait + host)"
          % (n_b, n_s, np.median(ts) * 1e6, med[0], med[1], med[2], med[3], med.sum()), flush=True)
    for r in rs:
        r.destroy()
    kf.destroy(); cf.destroy()
