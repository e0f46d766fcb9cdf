import sys, numpy as np, ctypes as C
sys.path.insert(0,'/root/repo')
from android_svo_amd import hip, seedsynth
ctx = hip.Context(0)
sc = seedsynth.make_seed_case(n_seeds=100000, seed=9)
kf = hip.Pyramid(ctx, 640, 480, 5, 1); cf = hip.Pyramid(ctx, 640, 480, 5, 1)
kf.upload(0, sc.ref_pyr); cf.upload(0, sc.cur_pyr)
sb = hip.SeedBatch(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, sc.sigma2)
for _ in range(3):
    hip.depth_filter_update(ctx, kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w, sb)
ctx.sync()
a = sb.n_zmssd.download(); b = sb.n_align.download()
print(sys.argv[1], "mean", a[a>0].mean(), b[b>0].mean(), "median", np.median(a[a>0]), np.median(b[b>0]))
