"""Soak of the drop-in SparseImgAlign binding on the reference types (oracle/_ref/libsvo_dropin_run.so, GPU box): 240 runs that
switch frame size, camera model, padded / continuous images and solver branch from call to call; every repeat of a configuration
must be bit-equal and the process must not grow.  python tools/dropin_soak.py"""
import sys, os, ctypes as C, numpy as np, psutil, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_dropin_binding as T
from oracle import gen_golden
lib = C.CDLL(T.LIB, mode=os.RTLD_LAZY)
cases = {c[0]: c for c in gen_golden.SIA_REF_CASES}
names = ["c0_200", "nulls_320", "hd_1280", "border_320", "iters5", "radtan_600"]
fps = {n: gen_golden.make_sia_case(cases[n][1]) for n in names}
proc = psutil.Process()
ref = {}
t0 = time.time()
for it in range(240):
    n = names[it % len(names)]
    _, kw, mx, mn, ni = cases[n]
    combo = [(0, 0, 0), (1, 0, 0), (0, 2, 3), (1, 1, 1)][(it // len(names)) % 4]
    r = T.run_dropin(lib, fps[n], mx, mn, ni, method=combo[0], scale_estimator=combo[1], weight_function=combo[2], row_pad=(it % 5 == 0) * 7)
    key = (n, combo)
    sig = (r["T"].tobytes(), r["chi2"], r["n_tracked"])
    assert ref.setdefault(key, sig) == sig, ("run-to-run difference", key)
    if it in (30, 239):
        print("iteration", it, "rss MB", proc.memory_info().rss / 2**20)
print("240 runs over 6 frame sizes / cameras and 4 solver branches: every repeat bit-equal; %.1f s" % (time.time() - t0))
