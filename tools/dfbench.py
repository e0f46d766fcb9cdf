import sys, time, numpy as np, ctypes as C
sys.path.insert(0,'/root/repo')
from android_svo_amd import hip, seedsynth
ctx = hip.Context(0)
sc = seedsynth.make_seed_case(n_seeds=100000, seed=9)
kf = hip.Pyramid(ctx, 640, 480, 5, 1); cf = hip.Pyramid(ctx, 640, 480, 5, 1)
kf.upload(0, sc.ref_pyr); cf.upload(0, sc.cur_pyr)
sb = hip.SeedBatch(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, sc.sigma2)
state, state0 = hip.pack_seed_state(sb)
def run():
    ctx.check(ctx.lib.svo_hip_copy_d2d(ctx.h, C.c_void_p(state.ptr), C.c_void_p(state0.ptr), C.c_size_t(state.nbytes)), "d2d")
    hip.depth_filter_update(ctx, kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w, sb)
for _ in range(3): run()
ctx.sync(); t0=time.perf_counter()
for _ in range(20): run()
ctx.sync(); print(sys.argv[1] if len(sys.argv)>1 else "", (time.perf_counter()-t0)/20*1e6, "us")
