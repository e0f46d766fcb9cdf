import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from android_svo_amd import hip, seedsynth
ctx = hip.Context(0)
for n_b, n_s in ((4, 500), (4, 2000), (1, 2000), (1, 500)):
    sc = seedsynth.make_seed_case(n_seeds=n_s, seed=9)
    kf = hip.Pyramid(ctx, 640, 480, 5, 1); cf = hip.Pyramid(ctx, 640, 480, 5, 1)
    kf.upload(0, sc.ref_pyr); cf.upload(0, sc.cur_pyr)
    rs = [hip.ResidentSeeds(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, sc.sigma2) for _ in range(n_b)]
    def frame():
        for r in rs: r.update_async(kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w)
        for r in rs: r.collect_raw()
    for _ in range(20): frame()
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); frame(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("%d batches x %d seeds: %.1f us per frame (median), %.1f min" % (n_b, n_s, ts[25]*1e6, ts[0]*1e6))
    for r in rs: r.destroy()
    kf.destroy(); cf.destroy()
