"""What a frame of the drop-in DepthFilter costs at the sizes the reference produces (a few keyframes, a few hundred seeds
each): one pass per keyframe (svo_hip_seed_batch_update_async per batch, six launches each) against ONE set of launches for
all of them (svo_hip_seed_batch_update_group_async) in its six-launch and its two-launch form
(svo_hip_df_set_small_pass_limit), events collected, host side included.  -> profiles/r*_df_realistic_sizes.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from android_svo_amd import hip, seedsynth  # noqa: E402

ctx = hip.Context(0)
for n_b, n_s in ((4, 500), (4, 2000), (8, 300), (1, 500), (1, 2000), (1, 4000), (1, 8000), (2, 8000), (1, 16000)):
    mk = seedsynth.make_multi_keyframe_case((n_s,) * n_b, seed=9)
    kf = hip.Pyramid(ctx, 640, 480, 5, n_b)
    cf = hip.Pyramid(ctx, 640, 480, 5, 1)
    cf.upload(0, mk.cur_pyr)
    for k, sc in enumerate(mk.keyframes):
        kf.upload(k, sc.ref_pyr)
    rs = []
    T_refs = np.stack([sc.T_ref_w for sc in mk.keyframes])
    slots = list(range(n_b))

    def per_keyframe():
        for k, r in enumerate(rs):
            r.update_async(kf, k, cf, 0, mk.cam, T_refs[k], mk.T_cur_w)
        for r in rs:
            r.collect_raw()

    def grouped():
        hip.ResidentSeeds.update_group_async(rs, kf, slots, cf, 0, mk.cam, T_refs, mk.T_cur_w)
        for r in rs:
            r.collect_raw()

    out = []
    for name, fn, limit in (("one pass per keyframe", per_keyframe, 0), ("one launch set (6 launches)", grouped, 0),
                            ("one launch set, small-pass form (2 launches)", grouped, 16384)):
        ctx.set_small_pass_limit(limit)
        for r in rs:
            r.destroy()
        rs[:] = [hip.ResidentSeeds(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, sc.sigma2) for sc in mk.keyframes]   # same seed state for both forms
        for _ in range(20):
            fn()
        ts = []
        for _ in range(60):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out.append("%s %.1f us (median; best %.1f)" % (name, ts[30] * 1e6, ts[0] * 1e6))
    print("%d keyframes x %d seeds: %s" % (n_b, n_s, " | ".join(out)), flush=True)
    for r in rs:
        r.destroy()
    kf.destroy()
    cf.destroy()
