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
    out["vendor_peak"] = 8000.0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
