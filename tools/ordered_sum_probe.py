"""Time svo_hip_ordered_sum_f32_dev (android_svo_amd/csrc/svo_ordered_sum.h: the in-order f32 sum by a workgroup) for a range of
lengths, 50 launches each, and check the result against the scalar loop.  GPU box: python tools/ordered_sum_probe.py"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from android_svo_amd import hip
ctx = hip.Context(0)
rng = np.random.default_rng(1)
for n in (1024, 1600, 2048, 3200, 4096, 8192, 32000, 131072, 1048576):
    x = (rng.normal(0, 6, n).astype(np.float32) ** 2).astype(np.float32)
    d_v = hip.DeviceArray(ctx, x); d_o = hip.DeviceArray(ctx, shape=(1,), dtype=np.float32)
    def run(k):
        for _ in range(k):
            ctx.check(ctx.lib.svo_hip_ordered_sum_f32_dev(ctx.h, C.c_void_p(d_v.ptr), C.c_size_t(n), C.c_void_p(d_o.ptr)), "os")
        ctx.sync()
    run(5)
    t0 = time.perf_counter(); run(50); dt = (time.perf_counter() - t0) / 50
    want = np.add.accumulate(x, dtype=np.float32)[-1]
    print("n %8d  %8.1f us per sum  (%.2f ns per element)  exact %s" % (n, dt * 1e6, dt * 1e9 / n, d_o.download()[0] == want))
