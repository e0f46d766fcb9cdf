import sys, time, json
sys.path.insert(0, '/root/repo')
from android_svo_amd import hip, synth
ctx = hip.Context(0)
out = {}
for n in (5, 12, 16, 64, 121):
    fp = synth.make_frame_pair(seed=999, n_features=n)
    cam = fp.cam
    ref = hip.Pyramid(ctx, cam.width, cam.height, 5, 1); cur = hip.Pyramid(ctx, cam.width, cam.height, 5, 1)
    sia = hip.SparseImgAlign(ctx, 1, 128)
    sia.set_frames(ref, cur); ref.upload(0, fp.ref_pyr); cur.upload(0, fp.cur_pyr); sia.upload_pair(0, fp)
    for es in (False, True):
        prm = sia.params(early_stop=es)
        for _ in range(3): sia.run(1, prm)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(20): sia.run(1, prm)
        ctx.sync()
        r = sia.download(0)
        out["n%d_%s" % (n, "early" if es else "fixed")] = {"ms": (time.perf_counter() - t0) / 20 * 1e3, "iters": list(r.iters)[:5], "stop": r.stop}
print(json.dumps(out, indent=0))
