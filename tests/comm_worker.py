"""One rank of the native multi-GPU entry points (svo_hip_sia_run_sharded, svo_hip_seed_gather_converged_dev), started by
tests/test_gpu_comm.py as a subprocess.  usage: comm_worker.py <what> <transport> <rank> <world> <token> <out.npz>
  what      = sia | sia_c3 | sia_c3_full | sia_timing | seeds
  transport = shm (token = segment name; the ranks may share one GPU) | rccl (token = file holding the 128-byte unique id)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from android_svo_amd import hip, seedsynth, synth  # noqa: E402


def main():
    what, transport, rank, world, token, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    ctx = hip.Context(0 if transport == "shm" else rank)
    if transport == "shm":
        comm = hip.Comm(ctx, rank, world, kind="shm", name=token, slot_bytes=8 << 20)
    else:
        comm = hip.Comm(ctx, rank, world, kind="rccl", unique_id=open(token, "rb").read())
    if what in ("sia_c3", "sia_c3_full"):
        # BASELINE config C3's shape: 1280x720 pairs with 2000 patches, patch-sharded; "sia_c3": two pairs, short fixed
        # iteration budget; "sia_c3_full": the configuration's 8 concurrent pairs x 5 levels x 30 evaluations
        n_pairs, n_it = (8, 30) if what == "sia_c3_full" else (2, 4)
        fps = [synth.make_frame_pair(seed=3300 + i, n_features=2000, width=1280, height=720) for i in range(n_pairs)]
        cam = fps[0].cam
        ref = hip.Pyramid(ctx, cam.width, cam.height, 5, len(fps))
        cur = hip.Pyramid(ctx, cam.width, cam.height, 5, len(fps))
        sia = hip.SparseImgAlign(ctx, len(fps), 2000)
        sia.set_frames(ref, cur)
        for s, fp in enumerate(fps):
            ref.upload(s, fp.ref_pyr); cur.upload(s, fp.cur_pyr); sia.upload_pair(s, fp)
        prm = sia.params(max_level=4, min_level=0, n_iter=n_it, eps=1e-6, early_stop=False)
        hip.sia_run_sharded(sia, comm, len(fps), prm)
        r = sia.download_all(len(fps))
        np.savez(out, T=np.array([list(x.T_cur_w) for x in r]), n=np.array([x.n_tracked for x in r]), H=np.array([list(x.H) for x in r]))
    elif what == "sia_timing":
        # the configuration that exposed a race inside one launch of the sharded solve (a block writing its new partial row
        # while another block of the frame still read the previous evaluation's exchanged rows from the same buffer): small
        # shards (short launches), event records around every launch (other launch timing), several solves in a row
        fps = [synth.make_frame_pair(seed=12345 + i, n_features=2000) for i in range(6)]
        cam = fps[0].cam
        ref = hip.Pyramid(ctx, cam.width, cam.height, 5, len(fps))
        cur = hip.Pyramid(ctx, cam.width, cam.height, 5, len(fps))
        sia = hip.SparseImgAlign(ctx, len(fps), 2000)
        sia.set_frames(ref, cur)
        for s, fp in enumerate(fps):
            ref.upload(s, fp.ref_pyr); cur.upload(s, fp.cur_pyr); sia.upload_pair(s, fp)
        prm = sia.params(max_level=4, min_level=0, n_iter=30, eps=1e-6, early_stop=False)
        sia.set_profiling(True)
        Ts = []
        for rep in range(4):
            hip.sia_run_sharded(sia, comm, len(fps), prm)
            Ts.append(np.array([list(x.T_cur_w) for x in sia.download_all(len(fps))]))
        sia.get_profile()
        np.savez(out, T=np.stack(Ts))
    elif what == "sia":
        # three frame pairs of different sizes (one with point-less features), identical on every rank
        fps = [synth.make_frame_pair(seed=900 + i, n_features=n, null_point_every=k) for i, (n, k) in enumerate(((700, 0), (333, 7), (1500, 0)))]
        cam = fps[0].cam
        ref = hip.Pyramid(ctx, cam.width, cam.height, 5, len(fps))
        cur = hip.Pyramid(ctx, cam.width, cam.height, 5, len(fps))
        sia = hip.SparseImgAlign(ctx, len(fps), max(len(f.px) for f in fps))
        sia.set_frames(ref, cur)
        for s, fp in enumerate(fps):
            ref.upload(s, fp.ref_pyr); cur.upload(s, fp.cur_pyr); sia.upload_pair(s, fp)
        res = {}
        for tag, es in (("early", True), ("fixed", False)):
            prm = sia.params(max_level=4, min_level=0, n_iter=30 if es else 6, eps=1e-6, early_stop=es)
            hip.sia_run_sharded(sia, comm, len(fps), prm, graph=(transport == "rccl" and os.environ.get("SVO_TEST_GRAPH") == "1"))
            if transport == "rccl" and os.environ.get("SVO_TEST_GRAPH") == "1":
                hip.sia_run_sharded(sia, comm, len(fps), prm)       # a second solve replays the instantiated graphs
            r = sia.download_all(len(fps))
            res[tag + "_T"] = np.array([list(x.T_cur_w) for x in r])
            res[tag + "_n"] = np.array([x.n_tracked for x in r])
            res[tag + "_H"] = np.array([list(x.H) for x in r])
            res[tag + "_iters"] = np.array([list(x.iters)[:5] for x in r])
            # the shard was in force for the sharded call only: a plain run on the same object is the whole problem again
            sia.run(len(fps), prm)
            res[tag + "_T_plain"] = np.array([list(x.T_cur_w) for x in sia.download_all(len(fps))])
        np.savez(out, **res)
    else:
        sc = seedsynth.make_seed_case(n_seeds=6000, seed=13)
        lo, hi = (len(sc.px) * rank) // world, (len(sc.px) * (rank + 1)) // world
        kf = hip.Pyramid(ctx, sc.cam.width, sc.cam.height, 5, 1)
        cf = hip.Pyramid(ctx, sc.cam.width, sc.cam.height, 5, 1)
        kf.upload(0, sc.ref_pyr); cf.upload(0, sc.cur_pyr)
        s2 = (sc.sigma2 * 0.0012).astype(np.float32)        # tight seeds: a good part converges in one pass
        sb = hip.SeedBatch(ctx, sc.px[lo:hi], sc.f[lo:hi], sc.level[lo:hi], sc.a[lo:hi], sc.b[lo:hi], sc.mu[lo:hi], sc.z_range[lo:hi], s2[lo:hi])
        hip.depth_filter_update(ctx, kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w, sb)
        rec, counts = hip.seed_gather_converged(ctx, comm, sb, lo, cap=hi - lo)
        rec_small, counts_small = hip.seed_gather_converged(ctx, comm, sb, lo, cap=10)      # overflow is reported, not hidden
        st = sb.status.download()
        np.savez(out, rec=rec, counts=counts, rec_small=rec_small, counts_small=counts_small, n_conv_local=int((st == 4).sum()),
                 local_ids=np.where(st == 4)[0] + lo, local_mu=sb.mu.download()[st == 4], local_xyz=sb.xyz.download()[st == 4])
    ctx.sync()
    comm.destroy()


if __name__ == "__main__":
    main()
