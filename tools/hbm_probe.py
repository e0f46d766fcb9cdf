#!/usr/bin/env python3
"""Measured HBM ceiling of the box ("confirm on the box ... a D2D triad and report the
measured ceiling too").  Uses the library's own C-ABI (svo_hip_copy_d2d / svo_hip_memset on the context's
stream) plus a torch triad a = b + s*c, HIP-event timed.  Prints one JSON line.

    python tools/hbm_probe.py [--mib 2048] [--reps 20]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    from android_svo_amd import hip

    n = args.mib << 20
    stream = torch.cuda.Stream()
    ctx = hip.Context(0, stream=stream.cuda_stream)
    import ctypes as C
    lib = ctx.lib
    a, b = ctx.malloc(n), ctx.malloc(n)

    def memset(p, v):
        ctx.check(lib.svo_hip_memset(ctx.h, C.c_void_p(p), C.c_int(v), C.c_size_t(n)), "memset")

    def copy(dst, src):
        ctx.check(lib.svo_hip_copy_d2d(ctx.h, C.c_void_p(dst), C.c_void_p(src), C.c_size_t(n)), "d2d")

    memset(a, 1)
    memset(b, 2)
    ctx.sync()

    def timed(fn, bytes_moved):
        for _ in range(3):
            fn()
        ctx.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        ctx.sync()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        return bytes_moved / dt / 1e9

    out = {"buffer_MiB": args.mib, "reps": args.reps, "unit": "GB/s"}
    out["copy_d2d"] = timed(lambda: copy(a, b), 2 * n)          # read n + write n
    out["memset"] = timed(lambda: memset(a, 3), n)                               # write n
    with torch.cuda.stream(stream):
        m = n // 4
        x = torch.ones(m, dtype=torch.float32, device="cuda")
        y = torch.ones(m, dtype=torch.float32, device="cuda")
        z = torch.empty(m, dtype=torch.float32, device="cuda")

        def triad():
            with torch.cuda.stream(stream):
                torch.add(x, y, alpha=0.5, out=z)

        out["triad"] = timed(triad, 3 * n)                                     # read 2n + write n

        def rsum():
            with torch.cuda.stream(stream):
                x.sum()

        out["read_sum"] = timed(rsum, n)
    # host -> device upload of image pyramids through the C-ABI (pageable numpy buffers, as a caller would hand them over)
    import numpy as np
    from android_svo_amd import synth
    B = 64
    pyr = hip.Pyramid(ctx, 640, 480, 5, B)
    levels = synth.build_pyramid(np.random.default_rng(0).integers(0, 256, (480, 640)).astype(np.uint8))
    for s in range(4):
        pyr.upload(s, levels)
    ctx.sync()
    t0 = time.perf_counter()
    for s in range(B):
        pyr.upload(s, levels)
    ctx.sync()
    dt = time.perf_counter() - t0
    out["pyramid_upload_h2d"] = {"GBps": B * 409200 / dt / 1e9, "pyramids_per_s": B / dt, "bytes_per_pyramid": 409200,
                                 "note": "svo_hip_pyramid_upload from pageable host memory, one call per pyramid"}
    # the same from page-locked memory (svo_hip_malloc_host), asynchronous, one sync at the end; and level 0 only
    # with the pyramid built on the device
    hp = C.c_void_p()
    ctx.check(lib.svo_hip_malloc_host(ctx.h, C.byref(hp), C.c_size_t(409200)), "malloc_host")
    off = 0
    arr = (C.POINTER(C.c_uint8) * hip.MAX_LEVELS)()
    for l, im in enumerate(levels):
        C.memmove(hp.value + off, im.ctypes.data, im.nbytes)
        arr[l] = C.cast(hp.value + off, C.POINTER(C.c_uint8))
        off += im.nbytes
    for s in range(4):
        ctx.check(lib.svo_hip_pyramid_upload(pyr.h, s, arr), "upload")
    ctx.sync()
    t0 = time.perf_counter()
    for s in range(B):
        ctx.check(lib.svo_hip_pyramid_upload(pyr.h, s, arr), "upload")
    ctx.sync()
    dt = time.perf_counter() - t0
    out["pyramid_upload_h2d_pinned"] = {"GBps": B * 409200 / dt / 1e9, "pyramids_per_s": B / dt}
    t0 = time.perf_counter()
    for s in range(B):
        ctx.check(lib.svo_hip_pyramid_upload_level0_and_build(pyr.h, s, arr[0]), "build")
    ctx.sync()
    dt = time.perf_counter() - t0
    out["level0_upload_and_device_pyramid_pinned"] = {"GBps": B * 307200 / dt / 1e9, "pyramids_per_s": B / dt}
    # a whole batch of pyramids in the device layout, one transfer
    pb, w_, h_, nl_, b_ = C.c_size_t(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    base = C.c_void_p()
    ctx.check(lib.svo_hip_pyramid_info(pyr.h, C.byref(w_), C.byref(h_), C.byref(nl_), C.byref(b_), C.byref(pb), C.byref(base)), "info")
    hpk = C.c_void_p()
    ctx.check(lib.svo_hip_malloc_host(ctx.h, C.byref(hpk), C.c_size_t(B * pb.value)), "malloc_host")
    for s in range(B):
        for l, im in enumerate(levels):
            o = C.c_size_t(0)
            lib.svo_hip_pyramid_level_offset(pyr.h, l, C.byref(o))
            C.memmove(hpk.value + s * pb.value + o.value, im.ctypes.data, im.nbytes)
    for _ in range(2):
        ctx.check(lib.svo_hip_pyramid_upload_packed(pyr.h, 0, B, C.cast(hpk, C.POINTER(C.c_uint8))), "packed")
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.check(lib.svo_hip_pyramid_upload_packed(pyr.h, 0, B, C.cast(hpk, C.POINTER(C.c_uint8))), "packed")
    ctx.sync()
    dt = (time.perf_counter() - t0) / 10
    out["pyramid_upload_packed_pinned"] = {"GBps": B * pb.value / dt / 1e9, "pyramids_per_s": B / dt, "pyramids_per_transfer": B}
    chk = pyr.download_level(B - 1, 2)
    assert (chk == levels[2]).all()
    # level 0 of the whole batch in one strided transfer, coarser levels built on the device (4 launches per batch)
    hl0 = C.c_void_p()
    ctx.check(lib.svo_hip_malloc_host(ctx.h, C.byref(hl0), C.c_size_t(B * 307200)), "malloc_host")
    for s in range(B):
        C.memmove(hl0.value + s * 307200, levels[0].ctypes.data, 307200)
    for _ in range(2):
        ctx.check(lib.svo_hip_pyramid_upload_level0_batch_and_build(pyr.h, 0, B, C.cast(hl0, C.POINTER(C.c_uint8))), "l0batch")
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.check(lib.svo_hip_pyramid_upload_level0_batch_and_build(pyr.h, 0, B, C.cast(hl0, C.POINTER(C.c_uint8))), "l0batch")
    ctx.sync()
    dt = (time.perf_counter() - t0) / 10
    out["level0_batch_upload_and_device_pyramids_pinned"] = {"GBps": B * 307200 / dt / 1e9, "pyramids_per_s": B / dt, "pyramids_per_transfer": B}
    for l in range(5):
        assert (pyr.download_level(B - 1, l) == levels[l]).all()
    ctx.check(lib.svo_hip_free_host(ctx.h, hl0), "free_host")
    ctx.check(lib.svo_hip_free_host(ctx.h, hpk), "free_host")
    ctx.check(lib.svo_hip_free_host(ctx.h, hp), "free_host")
    out["vendor_peak"] = 8000.0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
