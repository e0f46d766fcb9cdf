#!/usr/bin/env python3
"""bench_c4.py -- BASELINE config C4: seed-sharded DepthFilter, 1M seeds split across the GPUs of one node,
1280x720 keyframe, RCCL gather of the converged depths.  (bench.py is the headline metric; the driver does not run
this file.)

  python bench_c4.py                       one GPU, all seeds
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_c4.py
                                           one rank per GPU: every rank holds the keyframe and the current frame
                                           (two 1 227 600-byte pyramids) and a contiguous slice of the seeds; the
                                           only exchange is the gather of converged records after the update
                                           (counts, then one padded all-gather: SURVEY 8e).
One step = one DepthFilter::updateSeeds pass over all seeds (inputs resident in HBM) + the gather.
Prints one JSON line on rank 0."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from android_svo_amd import dist as svodist, hip, launcher, seedsynth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="started plainly with N > 1: this process starts the N ranks itself")
    ap.add_argument("--seeds", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gather", choices=["native", "torch"], default="native",
                    help="native: svo_hip_seed_gather_converged_dev (libsvo_hip.so calls RCCL); torch: android_svo_amd/dist.py driver")
    ap.add_argument("--sigma-scale", type=float, default=0.0045,
                    help="seed variance relative to a fresh seed: small enough that part of the seeds converge in this pass")
    args = ap.parse_args()
    if args.gpus > 1 and not launcher.launched_by_torchrun():
        sys.exit(launcher.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)                      # RCCL prints a banner on stdout
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    multi = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    # rehearsal of the multi-rank path on a box with fewer GPUs than ranks (bench.py has the same): shared device(s), gloo
    # process group, the library's shared-memory transport instead of RCCL; marked in the JSON line, measures nothing
    n_dev = torch.cuda.device_count()
    rehearsal = multi and n_dev < world
    if rehearsal:
        local_rank = local_rank % max(n_dev, 1)
    torch.cuda.set_device(local_rank)
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # the same synthetic keyframe / frame / seed set on every rank (same generator seed); each rank keeps its slice
    sc = seedsynth.make_seed_case(n_seeds=args.seeds, seed=9, width=args.width, height=args.height)
    lo, hi = svodist.shard_range(args.seeds, rank, world)
    stream = torch.cuda.Stream(device=local_rank)
    ctx = hip.Context(local_rank, stream=stream.cuda_stream)
    kf = hip.Pyramid(ctx, args.width, args.height, 5, 1)
    cf = hip.Pyramid(ctx, args.width, args.height, 5, 1)
    kf.upload(0, sc.ref_pyr)
    cf.upload(0, sc.cur_pyr)
    sl = slice(lo, hi)
    # seeds that have already been observed a few times (small variance) so that a visible share converges this pass
    sigma2 = (sc.sigma2[sl] * np.float32(args.sigma_scale)).astype(np.float32)
    sb = hip.SeedBatch(ctx, sc.px[sl], sc.f[sl], sc.level[sl], sc.a[sl], sc.b[sl], sc.mu[sl], sc.z_range[sl], sigma2)
    state, state0 = hip.pack_seed_state(sb)          # a | b | mu | sigma2 in one block: one copy restores all seeds
    n_conv_total = 0
    comm = None
    rec_all = cnt_all = None
    cap = hi - lo if world == 1 else max(1, (args.seeds + world - 1) // world)
    if multi and args.gather == "native":
        # the C-ABI's own communicator: rank 0's ncclUniqueId travels through torch.distributed.run's process group
        if rehearsal:
            import uuid
            nm = ["/svo_c4_" + uuid.uuid4().hex[:10] if rank == 0 else None]
            dist.broadcast_object_list(nm, src=0)
            comm = hip.Comm(ctx, rank, world, kind="shm", name=nm[0], slot_bytes=max(1 << 20, cap * 48 + 64))
        else:
            uid = [hip.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = hip.Comm(ctx, rank, world, kind="rccl", unique_id=uid[0])
        rec_all = ctx.empty((world * cap, 6), np.float64)
        cnt_all = ctx.empty((world,), np.int32)

    def step():
        nonlocal n_conv_total
        # same seed state every step
        ctx.check(ctx.lib.svo_hip_copy_d2d(ctx.h, C.c_void_p(state.ptr), C.c_void_p(state0.ptr), C.c_size_t(state.nbytes)), "d2d")
        hip.depth_filter_update(ctx, kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w, sb)
        if comm is not None:
            # packed on the GPU, two fixed-size RCCL all-gathers enqueued by libsvo_hip.so, no host round trip
            ctx.check(ctx.lib.svo_hip_seed_gather_converged_dev(
                ctx.h, comm.h, sb.n, C.c_longlong(lo), C.c_void_p(sb.status.ptr), C.c_void_p(sb.mu.ptr), C.c_void_p(sb.sigma2.ptr),
                C.c_void_p(sb.xyz.ptr), cap, C.c_void_p(rec_all.ptr), C.c_void_p(cnt_all.ptr)), "seed_gather_converged")
        else:
            rec = svodist.gather_converged_device(ctx, sb, lo, stream)   # torch.distributed driver
            n_conv_total = int(rec.shape[0])

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()

    # set-up: 40 passes (~30 ms) bring a chip that idled through the case generation to its running clocks (a fixed count:
    # a step holds collectives, every rank must run the same number; see bench_c2.timed)
    for _ in range(40):
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if comm is not None:
        n_conv_total = int(np.minimum(cnt_all.download(), cap).sum())
    if rank == 0:
        out = {"metric": "DepthFilter seed updates/s (BASELINE config C4)", "value": args.seeds * args.steps / dt, "unit": "seeds/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64 geometry / f32 image math / int32 ZMSSD",
               "data": "synthetic",
               **({"rehearsal": "%d ranks share %d GPU(s): gloo process group, shared-memory exchange -- a code-path rehearsal, not a scaling measurement" % (world, n_dev)} if rehearsal else {}),
               "config": {"workload": "C4: DepthFilter::updateSeeds, %d seeds on a %dx%d keyframe, seeds sharded over %d GPU(s), gather of converged records"
                                      % (args.seeds, args.width, args.height, world),
                          **({"comm_ranks": comm.count()} if comm is not None else {}),
                          "seeds_per_gpu": hi - lo, "converged_records_gathered": int(n_conv_total),
                          "gather": "svo_hip_seed_gather_converged_dev (RCCL called by libsvo_hip.so)" if comm is not None else "torch.distributed driver",
                          "note": "a step = the update pass + the on-device packing of the converged records + the RCCL all-gather"}}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.barrier()
        if comm is not None:
            ctx.sync()
            comm.destroy()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
