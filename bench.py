#!/usr/bin/env python3
"""bench.py -- SparseImgAlign frames/s at 640x480, L4-L0 (BASELINE.json metric) on MI355X.

One "step" = one pass of the hot path over one batch of frame pairs: the whole coarse-to-fine
solve (5 pyramid levels x 30 Gauss-Newton evaluations, fixed work = config C1) of `batch`
independent 640x480 pairs with 2000 patches each, inputs already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU -- launched by torch.distributed.run, or started plainly, in which case this process starts
  the N ranks itself (android_svo_amd/launcher.py) and relays rank 0's line.  Default sharding is by frame
  pair (independent objects, no data-path collective, weak scaling).  --mode allreduce runs
  BASELINE config C3's variant instead: every frame's patches are split over the ranks and the
  per-frame 6x6 H / 6x1 b sums are all-reduced (RCCL over xGMI) at every Gauss-Newton step.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel and
`cpu_baseline` (host cores).  What the roofline object says and where every number in it comes from:
  * the dominant kernel (sia_fused_kernel) is bound by VALU issue, not by HBM: `bound` = "valu", `achieved` = issue
    cycles it needs per second (per-type VALU instruction counts from rocprofv3 PMC passes of this very command, stored
    in profiles/r*_pmc_fused.json: 2 cycles per wave64 f32/int instruction, 4 per f64, 8 per transcendental), `peak` =
    1024 SIMDs x 2.4 GHz, `frac` <= 1; the launch time is measured live with HIP events on the context stream;
  * `traffic` / `hbm`: physical bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (lower bound as counted,
    upper bound with the gfx950 x2 correction of wide reads) against the 8 TB/s peak;
  * `algorithmic`: SURVEY 8(d)'s reference-layout bytes (945 B / patch / level + 881 B / patch / evaluation) over the
    same launch time -- above the HBM peak because the kernel never moves the 768 B/patch fp64 Jacobian stream;
  * `roofline_jacobian_pass`: the streaming implementation of the same pass (sia_residual_kernel), which IS bound by
    HBM: physical GB/s against the peak (north_star: >= 50 %).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from android_svo_amd import hip, launcher, synth  # noqa: E402   (none of them touches the GPU at import)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_PRECOMPUTE = 945           # per patch per level
BYTES_RESIDUAL = 881             # per patch per Gauss-Newton evaluation
# frame pairs per GPU per launch.  One 8-wave workgroup solves one pair and fills one CU: with 256 pairs every CU runs
# exactly one workgroup and the launch lasts as long as its slowest scene; more workgroups per CU balance that
# (same box, default arithmetic, profiles/r04_batch_sweep.txt: 1024 / 2048 / 4096 pairs -> 225.1 / 227.1 / 229.0 k frames/s;
# 4096 pairs are 3.9 GB of resident pyramids of the 288 GB) and the streaming form of the pass gains more
DEFAULT_BATCH = 4096


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # 100 steps = 0.23 s of timed GPU work at N = 1
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH, help="frame pairs per GPU per step")
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--width", type=int, default=640, help="image width (1280 for BASELINE config C3)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic scenes tiled over the batch")
    ap.add_argument("--mode", choices=["frames", "allreduce", "allreduce-torch"], default="frames",
                    help="allreduce: patch-sharded solve through the C-ABI (svo_hip_sia_run_sharded, RCCL called by libsvo_hip.so); "
                         "allreduce-torch: the same exchange driven from Python through torch.distributed (android_svo_amd/dist.py)")
    ap.add_argument("--early-stop", action="store_true", help="reference GN exits instead of fixed work")
    ap.add_argument("--graph", action="store_true",
                    help="--mode allreduce: svo_hip_sia_set_sharded_graph (one HIP graph per level inside the library); --mode allreduce-torch: torch graphs")
    ap.add_argument("--stream-mode", action="store_true",
                    help="time the streaming implementation (one launch per Gauss-Newton evaluation) instead of the fused kernel: the run the PMC "
                         "passes of the Jacobian pass profile (tools/pmc_stream.sh)")
    ap.add_argument("--arith", choices=["default", "exact", "moments_f32", "fast"], default="default",
                    help="arithmetic level of the fused kernel in the TIMED workload: default = the library's = exact "
                         "(SVO_HIP_SIA_ARITH_EXACT: the reference's arithmetic statement by statement, f64 Jres moments -- the headline); "
                         "moments_f32 / fast = the two opt-in levels, for their PMC passes (tools/pmc_fused.sh <tag> --arith moments_f32), "
                         "reported as secondary sections")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose-json", action="store_true", help="keep the explanatory strings in the JSON line (default: numbers and sources only; "
                                                                "what every key means is in profiles/BENCH_KEYS.md)")
    ap.add_argument("--cpu-frames-per-thread", type=int, default=16)
    ap.add_argument("--latency-probe", action="store_true",
                    help="(kept for old command lines: the single-pair latency is measured by default, outside the timed region, unless --no-secondary)")
    ap.add_argument("--profile-events", type=int, default=1, help="record HIP events around the heavy kernels in the timed region")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the extra measurements outside the timed region (single-pair latency, single-stream tracking chain, reference early-stop "
                         "semantics, streaming Jacobian pass, image uploads overlapped with the solve, the compact C2 / C4 blocks): use it when the run "
                         "is profiled, the kernel average in the rocprofv3 summary must be that of the timed launches")
    ap.add_argument("--allow-stale-profile", action="store_true",
                    help="use a PMC profile under profiles/ although the kernel sources have changed since it was taken (A/B work only)")
    return ap.parse_args()


def host_cpus():
    """(logical CPUs of the host, CPUs this process may actually use): the second is the smaller of the affinity mask
    and the cgroup CPU quota (a one-GPU box of the pool shows 256 logical CPUs and grants 16)."""
    nproc = os.cpu_count() or 1
    usable = nproc
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            usable = min(usable, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return nproc, max(1, usable)


def cpu_baseline(fps, n_iter=30, early_stop=False, frames_per_thread=16):
    """CPU baseline on the host cores, bounded sample.

    Both workloads run the reference's OWN compiled code (oracle/_ref, `kind: reference`) when the prebuilt library
    travelled to this box, else the C oracle (`kind: port`; the line then also carries `reference_check` = none).
    Fixed work (the default, BASELINE config C1's "30 GN iters"): the reference's compiled computeResiduals / solve /
    update, exactly 30 evaluations per level -- the loop around them is oracle/ref/ref_objects.cpp's, because the
    reference's own loop cannot be made to do fixed work (its error-increase exit is unconditional); results equal
    the port's bit for bit (tests/test_oracle_reference_objects.py).  Early stop (--early-stop): the reference's
    whole optimize().

    Threads: one independent frame pair at a time per thread (the reference's run() is serial), once with as many
    threads as the process is granted CPUs and -- when the host shows more logical CPUs than that -- once with one thread
    per logical CPU ("all host cores, core count stated"); `value` is the better of the two, `cores` the thread
    count it was measured with, `legs` holds both."""
    from oracle import orc
    orc.lib()
    try:
        from oracle.ref import refpy
        have_ref = refpy.available()
        if have_ref:
            refpy.lib()
    except Exception:
        have_ref = False
    nproc, usable = host_cpus()
    use_ref = bool(have_ref)

    def one(fp):
        if use_ref and early_stop:
            refpy.sparse_img_align_run(fp, n_iter=n_iter)
        elif use_ref:
            refpy.sparse_img_align_fixed_work(fp, n_iter=n_iter)
        else:
            orc.sparse_img_align(fp, n_iter=n_iter, early_stop=early_stop)

    def leg(n_threads, per_thread):
        done = [0] * n_threads

        def work(t):
            for k in range(per_thread):
                one(fps[(t + k) % len(fps)])
                done[t] += 1
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        return {"threads": n_threads, "frames": sum(done), "seconds": dt, "frames_per_s": sum(done) / dt}
    per_thread = frames_per_thread * (8 if early_stop else 1)
    legs = [leg(usable, per_thread)]
    if nproc > usable:       # one thread per logical CPU as well: bounded to about the same total work
        legs.append(leg(nproc, max(1, per_thread * usable // nproc)))
    best = max(legs, key=lambda l: l["frames_per_s"])
    # single-thread figure too (the reference's run() is serial)
    t1 = time.perf_counter()
    one(fps[0])
    single = time.perf_counter() - t1
    out = {"value": best["frames_per_s"], "unit": "frames/s", "cores": best["threads"], "nproc": nproc, "cpus_granted_to_this_process": usable,
           "legs": legs, "kind": "reference" if use_ref else "port", "single_thread_frames_per_s": 1.0 / single,
           "sample": "%d frame pairs (640x480, %d patches, L4-L0, %s) on %d threads in %.1f s; 1 thread: %.2f frames/s" %
                     (best["frames"], len(fps[0].px), "early stop" if early_stop else "30 GN evaluations/level fixed work",
                      best["threads"], best["seconds"], 1.0 / single)}
    if use_ref and not early_stop:
        out["driver"] = "harness loop around the reference's compiled computeResiduals/solve/update"   # (30 per level: profiles/BENCH_KEYS.md)
    return out


def c0_leg(ctx, n_scenes=8, batch=1024):
    """BASELINE config C0 -- the reference's own CPU-runnable case: one 640x480 pair, ~200 patches (a 40-px grid: 165), L4-L0 --
    timed as SURVEY 8(d) 'CPU baseline timing' asks: both Gauss-Newton modes; ONE PINNED CORE, median of 20 runs after 3
    warm-ups (the reference's run() is serial); all granted cores, one independent pair per thread.  Early stop is the
    reference's OWN compiled SparseImgAlign (oracle/_ref, kind "reference") when the prebuilt library is on this box, else the
    port; fixed work is the reference's compiled computeResiduals / solve / update in a 30-evaluation loop (see cpu_baseline).  Beside it the GPU's C0 figures:
    one pair at a time (latency) and `batch` pairs per launch (the 4-wave shape: two pairs per CU)."""
    from oracle import orc
    orc.lib()
    try:
        from oracle.ref import refpy
        have_ref = refpy.available()
        if have_ref:
            refpy.lib()
    except Exception:
        have_ref = False
    fps = [synth.make_frame_pair(seed=777 + i, n_features=200) for i in range(n_scenes)]
    n_patches = len(fps[0].px)
    es = (lambda fp: refpy.sparse_img_align_run(fp, n_iter=30)) if have_ref else (lambda fp: orc.sparse_img_align(fp, n_iter=30, early_stop=True))
    fw = (lambda fp: refpy.sparse_img_align_fixed_work(fp, n_iter=30)) if have_ref else (lambda fp: orc.sparse_img_align(fp, n_iter=30, early_stop=False))

    def pinned_median(fn):
        old = None
        try:
            old = os.sched_getaffinity(0)
            os.sched_setaffinity(0, {sorted(old)[-1]})               # this thread on one of the CPUs the process was granted
        except Exception:
            old = None
        try:
            for _ in range(3):
                fn(fps[0])
            ts = []
            for k in range(20):
                t = time.perf_counter()
                fn(fps[k % len(fps)])
                ts.append(time.perf_counter() - t)
        finally:
            if old is not None:
                os.sched_setaffinity(0, old)
        return float(np.median(ts)) * 1e3

    def all_cores(fn, per_thread):
        n_threads = host_cpus()[1]
        def work(t):
            for k in range(per_thread):
                fn(fps[(t + k) % len(fps)])
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
        [t.start() for t in th]
        [t.join() for t in th]
        return n_threads * per_thread / (time.perf_counter() - t0), n_threads

    es_ms, fw_ms = pinned_median(es), pinned_median(fw)
    es_fps, cores = all_cores(es, 40)
    fw_fps, _ = all_cores(fw, 8)
    cpu = {"es_ms_1core": es_ms, "fw_ms_1core": fw_ms, "es_fps": es_fps, "fw_fps": fw_fps, "cores": cores,
           "kind_es": "reference" if have_ref else "port", "kind_fw": "reference" if have_ref else "port"}
    # ---- the GPU on the same pairs
    cam = fps[0].cam
    ref = hip.Pyramid(ctx, cam.width, cam.height, 5, batch)
    cur = hip.Pyramid(ctx, cam.width, cam.height, 5, batch)
    sia = hip.SparseImgAlign(ctx, batch, n_patches)
    sia.set_frames(ref, cur)
    for s_ in range(batch):
        fp = fps[s_ % len(fps)]
        ref.upload(s_, fp.ref_pyr); cur.upload(s_, fp.cur_pyr); sia.upload_pair(s_, fp)
    gpu = {"pairs_per_launch": batch}
    worst = 0.0
    for tag, early in (("es", True), ("fw", False)):
        prm = sia.params(max_level=4, min_level=0, n_iter=30, eps=1e-6, early_stop=early)
        for n_l, key in ((1, "%s_ms_1pair" % tag), (batch, "%s_fps" % tag)):
            prewarm(ctx, lambda: sia.run(n_l, prm), 0.05)
            reps = 20
            t = time.perf_counter()
            for _ in range(reps):
                sia.run(n_l, prm)
                if n_l == 1:
                    ctx.sync()
            ctx.sync()
            dt = (time.perf_counter() - t) / reps
            gpu[key] = dt * 1e3 if n_l == 1 else n_l / dt
        res = sia.download_all(len(fps))
        for i, fp in enumerate(fps):                       # every C0 scene against the CPU path, both modes
            o = orc.sparse_img_align(fp, n_iter=30, early_stop=early)
            worst = max(worst, *synth.pose_error(np.array(res[i].T_cur_w), np.array(o.T_cur_w)))
            # (with the reference's exits an "error increased" decision can fall one evaluation earlier or later -- f32 chi2 summed
            # in another order -- and with it the last evaluation's patch count: only the fixed-work run must agree exactly)
            assert early or int(res[i].n_tracked) == int(o.n_tracked)
    assert worst < 1e-4, "C0 pose parity violated: %g" % worst
    sia.destroy(); ref.destroy(); cur.destroy()
    return {"patches": n_patches, "cpu": cpu, "gpu": gpu, "max_pose_err_vs_cpu": worst}


N_SIMD = 1024                    # 256 CUs x 4 SIMD-32
PEAK_CLOCK_GHZ = 2.4             # MI355X_MICROARCH.md, chip-level parameters


def latest_profile(pattern):
    """newest profiles/<pattern> (files are named per round: r02_..., r03_...)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def pmc_of(path, kernel_prefix, allow_stale=False, pairs=None):
    """counters per dispatch of the first kernel whose name starts with kernel_prefix (tools/pmc_json.py output).  A profile
    taken from other kernel sources than the tree's (tools/pmc_json.py stores their sha256) is refused: returns the string
    that says so."""
    try:
        d = json.load(open(path))
    except Exception:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_json
    have, want = d.get("source_sha256"), pmc_json.source_sha256()
    if have != want and not allow_stale:
        changed = sorted(k for k in want if not have or have.get(k) != want[k])
        return "%s was taken from other kernel sources than this tree's (%s differ%s): re-run tools/pmc_fused.sh" % (
            os.path.relpath(path, ROOT), ", ".join(changed), "" if len(changed) > 1 else "s")
    # every counter of these kernels is proportional to the frame pairs in the launch (256 -> 1024 pairs: x 4.000 on all of
    # them): a profile taken at another launch size is scaled to the one being timed
    prof_pairs = d.get("frame_pairs_per_launch")
    scale = (float(pairs) / prof_pairs) if (pairs and prof_pairs) else 1.0
    for name, ctr in d.get("kernels", {}).items():
        if name.startswith(kernel_prefix):
            out = {k: (v * scale if isinstance(v, (int, float)) else v) for k, v in ctr.items()}
            return dict(out, _kernel=name, _file=os.path.relpath(path, ROOT), _command=d.get("command", ""),
                        _scaled_from_pairs=prof_pairs if scale != 1.0 else None)
    return None


def valu_issue_cycles(c):
    """VALU issue cycles of one launch from the per-type instruction counters: a wave64 f32 / int instruction holds its
    SIMD-32 for 2 cycles, an f64 one for 4, a transcendental for 8 (MI355X_MICROARCH.md, 'Per-instruction cycle
    constants'; fp64 vector peak is half the fp32 one).  Conversions (v_cvt_f64_f32, v_cvt_f32_ubyte*, ...) issue at the
    fp64 rate on this chip -- measured, tools/probes/valu_rate_probe.hip: 1.88 ns against 1.90 ns for v_fma_f64 and
    1.11 ns for v_add_f32 with two waves per SIMD -- and are counted at 4.  Compares, shifts, min/max and lane reads are
    in that slower class too but have no counter of their own: they stay at 2, so the sum is still a lower bound."""
    f64 = c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_FMA_F64", 0.0)
    trans = c.get("SQ_INSTS_VALU_TRANS_F64", 0.0) + c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    cvt = c.get("SQ_INSTS_VALU_CVT", 0.0)
    total = c["SQ_INSTS_VALU"]
    return 2.0 * (total - f64 - trans - cvt) + 4.0 * (f64 + cvt) + 8.0 * trans, f64, trans


PREWARM_S = 0.1

# keys that hold explanations, not measurements: left out of the JSON line unless --verbose-json (profiles/BENCH_KEYS.md
# says what every key means; round 3's 14 kB line lost its first half in the driver's tail)
PROSE_KEYS = {"through_round3_host_buffer_entry_us", "what", "note", "rule", "traffic_rule", "sample", "legs", "measured_ceiling", "ms_per_frame_image_in_tracker_buffer_median",
              "ms_per_frame_min", "algorithmic_bytes", "algorithmic_GBps", "nproc", "cpus_granted_to_this_process", "patches_per_s",
              "seeds_per_s", "ms_per_frame_median", "converged", "steps_secondary", "transcendental_insts", "bytes_per_launch_as_counted",
              "upload_only_ms_per_step", "bytes_uploaded_per_step", "status_counts", "effective_clock_GHz_under_profiler", "frac_at_that_clock",
              "speedup_vs_1_thread", "mean_zmssd", "launches"}


def slim(o, verbose=False):
    """the JSON line without explanatory strings, floats to 6 significant digits"""
    if isinstance(o, dict):
        return {k: slim(v, verbose) for k, v in o.items() if verbose or k not in PROSE_KEYS}
    if isinstance(o, (list, tuple)):
        return [slim(v, verbose) for v in o]
    if isinstance(o, float):
        return float("%.6g" % o) if o == o and abs(o) != float("inf") else None
    if isinstance(o, (np.floating,)):
        return float("%.6g" % float(o))
    if isinstance(o, (np.integer,)):
        return int(o)
    return o


def prewarm(ctx, fn, seconds=PREWARM_S):
    """Part of the set-up, never of a timed region: call fn (a launch of the workload about to be measured) for `seconds` so
    that the chip is at its running clocks.  Every measurement here follows seconds of host work (scene generation, uploads,
    oracle checks) with the GPU idle, and an idle MI355X was seen to run its first tens of milliseconds many times slower
    (bench_c2.timed has the figures)."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()
        ctx.sync()


def with_uploads(torch, ctx, stream, local_rank, sia, ref, fps, n_slots, prm, steps):
    """Upload-inclusive throughput: every step a NEW current image per frame pair (level 0, n_slots x 307 200 B from one
    page-locked buffer) crosses PCIe on a second stream, its pyramid is built on the device
    (svo_hip_pyramid_upload_level0_batch_and_build), and the transfer of step k+1 overlaps the solve of step k
    (two current-pyramid batches, events both ways).  Reference pyramids and features stay resident: one new image per
    pair per step is what a tracked sequence costs (the current frame is the next reference)."""
    import ctypes as C
    cam = fps[0].cam
    up_stream = torch.cuda.Stream(device=local_rank)
    ctx_up = hip.Context(local_rank, stream=up_stream.cuda_stream)
    bufs = [hip.Pyramid(ctx_up, cam.width, cam.height, 5, n_slots) for _ in range(2)]
    l0 = cam.width * cam.height
    hp = C.c_void_p()
    ctx_up.check(ctx_up.lib.svo_hip_malloc_host(ctx_up.h, C.byref(hp), C.c_size_t(n_slots * l0)), "malloc_host")
    host = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(n_slots, l0))
    for s in range(n_slots):
        host[s] = fps[s % len(fps)].cur_pyr[0].reshape(-1)
    ev_up = [torch.cuda.Event() for _ in range(2)]
    ev_done = [torch.cuda.Event() for _ in range(2)]

    def upload(b):
        with torch.cuda.stream(up_stream):
            up_stream.wait_event(ev_done[b])                       # the solve that read this buffer has finished
            ctx_up.check(ctx_up.lib.svo_hip_pyramid_upload_level0_batch_and_build(bufs[b].h, 0, n_slots, C.cast(hp, C.POINTER(C.c_uint8))),
                         "upload_level0_batch_and_build")
            ev_up[b].record(up_stream)

    def solve(b):
        with torch.cuda.stream(stream):
            stream.wait_event(ev_up[b])
            sia.set_frames(ref, bufs[b])
            sia.run(n_slots, prm)
            ev_done[b].record(stream)

    for b in range(2):
        ev_done[b].record(stream)
    upload(0)
    for k in range(2):                                              # warm-up
        upload((k + 1) % 2)
        solve(k % 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        upload((k + 1) % 2)                                         # prefetch the next step's images
        solve(k % 2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the uploads alone, for reference
    t1 = time.perf_counter()
    for k in range(steps):
        upload(k % 2)
    torch.cuda.synchronize()
    dt_up = time.perf_counter() - t1
    r = sia.download(0)
    ctx_up.lib.svo_hip_free_host(ctx_up.h, hp)
    for b in bufs:
        b.destroy()
    return {"what": "one new 640x480 image per frame pair per step uploaded from page-locked memory on a second stream "
                    "(level 0 only, pyramid built on the device), double-buffered and overlapped with the solve",
            "value": n_slots * steps / dt, "unit": "frames/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
            "upload_only_ms_per_step": dt_up / steps * 1e3, "upload_GBps": n_slots * l0 / (dt_up / steps) / 1e9,
            "bytes_uploaded_per_step": n_slots * l0}, r


def main():
    args = parse_args()
    # `python bench.py --gpus N` started plainly: this process starts the N ranks itself (as a child process; it has not
    # touched the GPU and never does), relays rank 0's JSON line and leaves with the children's status
    if args.gpus > 1 and not launcher.launched_by_torchrun():
        sys.exit(launcher.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    # RCCL prints a version banner on stdout when a communicator is created; the contract is ONE JSON
    # line on stdout, so everything else this process (or a library) prints is sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    import torch
    import torch.distributed as dist
    multi = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)   # launched by torch.distributed.run
    # Rehearsal of the multi-rank code path on a box with fewer GPUs than ranks (the one-GPU development box): the ranks share
    # the devices there are, the process group is gloo (RCCL refuses two ranks on one GPU), the sharded solve exchanges over
    # the library's shared-memory transport.  The JSON line says so ("rehearsal"): such a run measures nothing about scaling.
    n_dev = torch.cuda.device_count()
    rehearsal = multi and n_dev < world
    dev_index = local_rank % max(n_dev, 1) if rehearsal else local_rank
    local_rank = dev_index
    torch.cuda.set_device(dev_index)
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    B = args.batch
    n_feat = args.features
    allreduce = args.mode in ("allreduce", "allreduce-torch") and multi
    native = args.mode == "allreduce"
    # ---- synthetic inputs (host), then resident in HBM before anything is timed
    def make_scenes(first_seed):
        # (rendering is numpy-bound and releases the GIL for most of its time: a few threads cut the set-up time)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(8, host_cpus()[1]))) as pool:
            return list(pool.map(lambda i: synth.make_frame_pair(seed=first_seed + i, n_features=n_feat, width=args.width, height=args.height),
                                 range(args.distinct)))
    fps = make_scenes(12345 + 17 * rank)
    cam = fps[0].cam
    stream = torch.cuda.Stream(device=local_rank)
    ctx = hip.Context(local_rank, stream=stream.cuda_stream)
    n_slots = B * world if allreduce else B
    ref = hip.Pyramid(ctx, cam.width, cam.height, 5, n_slots)
    cur = hip.Pyramid(ctx, cam.width, cam.height, 5, n_slots)
    sia = hip.SparseImgAlign(ctx, n_slots, n_feat)
    sia.set_frames(ref, cur)
    if args.stream_mode:
        sia.set_mode(stream=True)
    if args.arith == "exact":
        args.arith = "default"         # the library default IS the reference's arithmetic (nothing is set on the solver object)
    ARITH_OF = {"default": hip.SIA_ARITH_EXACT, "moments_f32": hip.SIA_ARITH_MOMENTS_F32, "fast": hip.SIA_ARITH_FAST}
    if args.arith != "default":
        sia.set_option(hip.SIA_OPT_ARITH, ARITH_OF[args.arith])
    if allreduce:
        # every rank holds every frame of the global batch (same seeds on all ranks), evaluates its patch shard
        fps = make_scenes(12345)
        sia.set_shard(rank, world)
    for s in range(n_slots):
        fp = fps[s % len(fps)]
        ref.upload(s, fp.ref_pyr)
        cur.upload(s, fp.cur_pyr)
        sia.upload_pair(s, fp)
    prm = sia.params(max_level=4, min_level=0, n_iter=30, eps=1e-6, early_stop=args.early_stop)
    aligner = None
    graphed = None
    comm = None
    if allreduce and native:
        # the communicator of the C-ABI: rank 0's ncclUniqueId reaches the other ranks through the process group that
        # torch.distributed.run set up (any side channel would do); from here on the exchange is libsvo_hip.so -> RCCL
        if rehearsal:
            import uuid
            nm = ["/svo_bench_" + uuid.uuid4().hex[:10] if rank == 0 else None]
            dist.broadcast_object_list(nm, src=0)
            comm = hip.Comm(ctx, rank, world, kind="shm", name=nm[0], slot_bytes=1 << 20)
        else:
            uid = [hip.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = hip.Comm(ctx, rank, world, kind="rccl", unique_id=uid[0])
    elif allreduce:
        from android_svo_amd import dist as svodist
        aligner = svodist.HipShardedAligner(sia, n_slots, prm, rank, world, stream)
        graphed = svodist.GraphedAllreduceSolver(aligner, prm.max_level, prm.min_level, prm.n_iter, stream) if args.graph else None

    comm_ranks = comm.count() if comm is not None else None     # what the transport itself reports (ncclCommCount)

    def step():
        if not allreduce:
            sia.run(n_slots, prm)
            return
        if comm is not None:
            hip.sia_run_sharded(sia, comm, n_slots, prm, graph=bool(args.graph))
            return
        if graphed is not None:
            graphed.run()
            return
        with torch.cuda.stream(stream):      # kernels and the RCCL all-reduce share this stream
            svodist.run_allreduce(aligner, prm.max_level, prm.min_level, prm.n_iter)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()

    # set-up: the chip at its running clocks before the W warm-up steps (a fixed count where a step holds a collective:
    # every rank must run the same number of them)
    if allreduce:
        for _ in range(5):
            step()
    else:
        prewarm(ctx, step)
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    sia.set_profiling(bool(args.profile_events))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = sia.get_profile() if args.profile_events else None
    sia.set_profiling(False)
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- single-pair latency with the reference's early-stop semantics (informative, outside the timed region)
    latency_ms = None
    if rank == 0 and not allreduce and world == 1 and not args.no_secondary:
        sia1 = hip.SparseImgAlign(ctx, 1, n_feat)
        sia1.set_frames(ref, cur)
        sia1.upload_pair(0, fps[0])
        prm1 = sia1.params(max_level=4, min_level=0, n_iter=30, eps=1e-6, early_stop=True)
        for _ in range(3):
            sia1.run(1, prm1)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(20):
            sia1.run(1, prm1)
            ctx.sync()
        latency_ms = (time.perf_counter() - t1) / 20 * 1e3
        sia1.destroy()

    # ---- one camera, one frame at a time: the whole per-frame chain through svo_hip_tracker_track (outside the timed region)
    chain = None
    if rank == 0 and not allreduce and world == 1 and not args.no_secondary and not args.early_stop and args.width == 640:
        for d_ in ("tools", "tests"):
            if os.path.join(ROOT, d_) not in sys.path:
                sys.path.insert(0, os.path.join(ROOT, d_))
        import chain_bench
        chain = chain_bench.single_stream_chain(ctx)

    # ---- parity check (outside the timed region): EVERY distinct scene against the CPU oracle, every replica bitwise
    res = sia.download(0)
    results = sia.download_all(n_slots)
    frames_global = B * world
    value = frames_global * args.steps / dt

    out = None
    if rank == 0:
        from oracle import orc          # checker + cpu_baseline leg only
        n_scenes = min(len(fps), n_slots)
        oracle_res = [None] * n_scenes

        def oracle_worker(t, nt):
            for i in range(t, n_scenes, nt):
                oracle_res[i] = orc.sparse_img_align(fps[i], n_iter=30, early_stop=args.early_stop)
        nt = max(1, min(host_cpus()[1], n_scenes))
        ths = [threading.Thread(target=oracle_worker, args=(t, nt)) for t in range(nt)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        o = oracle_res[0]
        rot, trans = synth.pose_error(np.array(res.T_cur_w), np.array(o.T_cur_w))
        scene_err = np.array([synth.pose_error(np.array(results[i].T_cur_w), np.array(oracle_res[i].T_cur_w)) for i in range(n_scenes)])
        tracked_equal = all(int(results[i].n_tracked) == int(oracle_res[i].n_tracked) for i in range(n_scenes))
        # all replicated slots of one scene must agree bit for bit (deterministic reductions)
        replicas_checked = 0
        for i in range(n_scenes, n_slots):
            assert list(results[i].T_cur_w) == list(results[i % n_scenes].T_cur_w), "slot %d differs from its scene's first slot" % i
            replicas_checked += 1
        n_res_per_frame = res.n_residual_patches       # patches accumulated over all evaluations of one frame
        n_pre_per_frame = res.n_precompute_patches
        if allreduce:
            # counters are per rank shard; scale to the whole frame for the algorithmic-bytes figure
            n_res_per_frame = o.n_residual_patches
            n_pre_per_frame = o.n_precompute_patches
        evals = sum(res.iters[:5])
        bytes_frame = n_pre_per_frame * BYTES_PRECOMPUTE + n_res_per_frame * BYTES_RESIDUAL
        roofline = None
        mode = sia.last_run_mode() if not allreduce else 0
        default_c1 = (not allreduce and not args.early_stop and n_feat == 2000 and args.width == 640 and
                      args.height == 480 and args.distinct == 64)       # the configuration the committed PMC passes profiled
        if prof and prof["residual_launches"]:
            launches = prof["residual_launches"]
            avg_ms = prof["residual_ms"] / launches
            avg_s = avg_ms * 1e-3
            if mode == 1:
                # fused: ONE launch runs the whole coarse-to-fine solve of every frame pair of this rank
                kernel = "sia_fused_kernel"
                alg_bytes = float(bytes_frame) * n_slots
            else:
                # streaming: one launch evaluates every live patch of every frame of this rank once
                kernel = "sia_residual_kernel"
                units = (n_res_per_frame / max(evals, 1)) * n_slots / (world if allreduce else 1)
                alg_bytes = units * BYTES_RESIDUAL
            alg = {"bytes_per_launch": alg_bytes, "GBps": alg_bytes / avg_s / 1e9, "ratio_to_hbm_peak": alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS,
                   "note": "SURVEY 8(d): 945 B/patch/level + 881 B/patch/evaluation in the REFERENCE's data layout (768 B of fp64 Jacobian "
                           "cache per patch); the kernels form H and Jres from {sum dx^2, sum dx dy, sum dy^2} and two moments per patch and "
                           "never move that stream, so this ratio is not a roofline fraction"}
            pmc_file = latest_profile({"default": "r*_pmc_fused.json", "moments_f32": "r*_pmc_fused_m32.json", "fast": "r*_pmc_fused_fast.json"}[args.arith]
                                      if mode == 1 else "r*_pmc_stream.json")
            ctr = pmc_of(pmc_file, kernel, args.allow_stale_profile, pairs=n_slots) if (pmc_file and default_c1) else None
            profile_refused = ctr if isinstance(ctr, str) else None
            if profile_refused:
                ctr = None
            if mode == 1:
                roofline = {"bound": "valu", "kernel": kernel, "achieved": None, "peak": N_SIMD * PEAK_CLOCK_GHZ,
                            "unit": "G VALU issue-cycles/s", "frac": None, "traffic": None,
                            "avg_launch_us": avg_ms * 1e3, "launches": int(launches), "algorithmic": alg,
                            "note": "one launch = whole coarse-to-fine solve of %d frame pairs (150 Gauss-Newton evaluations each); the kernel is "
                                    "bound by VALU issue (f32 image math + fp64 projection / normal equations) plus a serial solve phase between two "
                                    "barriers per evaluation, not by HBM" % n_slots}
                if profile_refused:
                    roofline["profile_refused"] = profile_refused
                if ctr:
                    cyc, n_f64, n_trans = valu_issue_cycles(ctr)
                    roofline["achieved"] = cyc / avg_s / 1e9
                    roofline["frac"] = roofline["achieved"] / roofline["peak"]
                    clk = None
                    if ctr.get("GRBM_GUI_ACTIVE") and ctr.get("_kernel_avg_us"):
                        clk = ctr["GRBM_GUI_ACTIVE"] / 8.0 / (ctr["_kernel_avg_us"] * 1e-6) / 1e9
                    roofline["valu"] = {"insts_per_launch": ctr["SQ_INSTS_VALU"], "f64_insts": n_f64, "transcendental_insts": n_trans,
                                        "issue_cycles_per_launch": cyc,
                                        "rule": "2 cycles per wave64 f32/int instruction, 4 per f64 add/mul/fma and per conversion (v_cvt_* issue "
                                                "at the f64 rate: measured, tools/probes/valu_rate_probe.hip), 8 per transcendental; compares, shifts, "
                                                "min/max and lane reads are in the slower class too but are counted at 2: a lower bound",
                                        "cvt_insts": ctr.get("SQ_INSTS_VALU_CVT"),
                                        "effective_clock_GHz_under_profiler": clk,
                                        "frac_at_that_clock": (cyc / avg_s / 1e9) / (N_SIMD * clk) if clk else None}
                    lo = (ctr.get("FETCH_SIZE", 0.0) + ctr.get("WRITE_SIZE", 0.0)) * 1024.0
                    hi = (2.0 * ctr.get("FETCH_SIZE", 0.0) + ctr.get("WRITE_SIZE", 0.0)) * 1024.0
                    compulsory = n_slots * (2 * 409200 + n_feat * 65)       # both pyramids once + px/f/pos/has_point per feature
                    roofline["traffic"] = hi
                    roofline["hbm"] = {"bytes_per_launch_as_counted": lo, "bytes_per_launch_wide_read_corrected": hi,
                                       "GBps": [lo / avg_s / 1e9, hi / avg_s / 1e9], "peak": HBM_PEAK_GBS,
                                       "frac": [lo / avg_s / 1e9 / HBM_PEAK_GBS, hi / avg_s / 1e9 / HBM_PEAK_GBS],
                                       "compulsory_bytes_per_launch": compulsory,
                                       "note": "(FETCH_SIZE + WRITE_SIZE) KB as counted (Infinity-Cache hits included) and with the gfx950 x2 "
                                               "correction of FETCH_SIZE, which holds for 16 B/lane streams (the stored interpolated patches) "
                                               "but not for the 8 B/lane image rows: lower and upper bound"}
                    roofline["sources"] = [ctr["_file"], "profiles/%s" % os.path.basename(pmc_file).replace("_pmc_fused.json", "_kernel_stats.md")]
                    if ctr.get("_scaled_from_pairs"):
                        roofline["counters_scaled_from_frame_pairs_per_launch"] = ctr["_scaled_from_pairs"]
            else:
                ach = alg_bytes / avg_s / 1e9
                roofline = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": avg_ms * 1e3, "launches": int(launches),
                            "algorithmic": alg, "note": "one launch = one Gauss-Newton evaluation of %d frame pairs" % n_slots}
                if ctr:
                    phys = (2.0 * ctr.get("FETCH_SIZE", 0.0) + ctr.get("WRITE_SIZE", 0.0)) * 1024.0
                    roofline.update({"achieved": phys / avg_s / 1e9, "frac": phys / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": phys,
                                     "sources": [ctr["_file"]]})
                    if ctr.get("_scaled_from_pairs"):
                        roofline["counters_scaled_from_frame_pairs_per_launch"] = ctr["_scaled_from_pairs"]
            if prof["precompute_launches"]:
                roofline["precompute_avg_launch_us"] = prof["precompute_ms"] / prof["precompute_launches"] * 1e3
            hp = os.path.join(ROOT, "profiles", "r01_hbm_probe.json")
            if os.path.exists(hp):
                try:
                    m = json.load(open(hp))
                    roofline["measured_ceiling"] = {"copy_d2d": m["copy_d2d"], "triad": m["triad"], "fill": m["memset"], "unit": "GB/s",
                                                    "source": "profiles/r01_hbm_probe.json (tools/hbm_probe.py on this pool)"}
                except Exception:
                    pass

        # ---- secondary measurements, outside the timed region (rank 0, one GPU, default workload)
        early = None
        fast = None
        m32_sec = None
        jac = None
        upl = None
        if not allreduce and world == 1 and not args.no_secondary and not args.early_stop:
            # (1) the same batch with the REFERENCE's Gauss-Newton exits (error increase, |x| <= eps)
            prm_es = sia.params(max_level=4, min_level=0, n_iter=30, eps=1e-6, early_stop=True)
            prewarm(ctx, lambda: sia.run(n_slots, prm_es))
            es_steps = max(5, min(args.steps, 20))
            t1 = time.perf_counter()
            for _ in range(es_steps):
                sia.run(n_slots, prm_es)
            ctx.sync()
            dt_es = time.perf_counter() - t1
            r_es = sia.download(0)
            o_es = orc.sparse_img_align(fps[0], n_iter=30, early_stop=True)
            rot_es, trans_es = synth.pose_error(np.array(r_es.T_cur_w), np.array(o_es.T_cur_w))
            early = {"what": "the same %d frame pairs with the reference's own Gauss-Newton exits (svo_hip_sia_params.early_stop = 1)" % n_slots,
                     "value": n_slots * es_steps / dt_es, "unit": "frames/s", "steps": es_steps, "ms_per_step": dt_es / es_steps * 1e3,
                     "gn_evaluations_frame0": int(sum(r_es.iters[:5])),
                     "pose_err_vs_cpu_ref": {"rot_rad": rot_es, "trans_m": trans_es},
                     "pose_err_vs_ground_truth": dict(zip(("rot_rad", "trans_m"), synth.pose_error(np.array(r_es.T_cur_w), fps[0].T_cur_w_true)))}
            assert rot_es < 1e-4 and trans_es < 1e-3, "early-stop pose parity violated: %g rad %g m" % (rot_es, trans_es)
            # (1b) the same batch, fixed work, at the two OPT-IN arithmetic levels of the fused kernel -- MOMENTS_F32 (a patch's two
            # gradient moments summed in f32) and FAST (contracted interpolation, f32 sums) --, every distinct scene against the CPU
            # oracle.  Narrower than the reference's arithmetic: secondary figures, never the headline
            other = {}
            for lvl_name, lvl, prof_pat in (("moments_f32", hip.SIA_ARITH_MOMENTS_F32, "r*_pmc_fused_m32.json"), ("fast", hip.SIA_ARITH_FAST, "r*_pmc_fused_fast.json")):
                sia.set_option(hip.SIA_OPT_ARITH, lvl)
                try:
                    prewarm(ctx, lambda: sia.run(n_slots, prm))
                    fa_steps = max(5, min(args.steps, 20))
                    t1 = time.perf_counter()
                    for _ in range(fa_steps):
                        sia.run(n_slots, prm)
                    ctx.sync()
                    dt_fa = time.perf_counter() - t1
                    r_fa = sia.download_all(n_slots)
                finally:
                    sia.set_option(hip.SIA_OPT_ARITH, ARITH_OF[args.arith])
                fa_err = np.array([synth.pose_error(np.array(r_fa[i].T_cur_w), np.array(oracle_res[i].T_cur_w)) for i in range(n_scenes)])
                # the same VALU-issue accounting as `roofline`, from the PMC passes of that instance (tools/pmc_fused.sh <tag> --arith <level>)
                fa_roof = None
                fa_file = latest_profile(prof_pat)
                fa_ctr = pmc_of(fa_file, "sia_fused_kernel", args.allow_stale_profile, pairs=n_slots) if (fa_file and default_c1) else None
                if isinstance(fa_ctr, str):
                    fa_roof = {"profile_refused": fa_ctr}
                elif fa_ctr:
                    fa_cyc, _, _ = valu_issue_cycles(fa_ctr)
                    fa_s = dt_fa / fa_steps
                    fa_roof = {"bound": "valu", "frac": fa_cyc / fa_s / 1e9 / (N_SIMD * PEAK_CLOCK_GHZ), "sources": [fa_ctr["_file"]]}
                other[lvl_name] = {"what": "the timed workload with svo_hip_sia_set_option(SVO_HIP_SIA_OPT_ARITH, SVO_HIP_SIA_ARITH_%s)" % lvl_name.upper(),
                                   "value": n_slots * fa_steps / dt_fa, "unit": "frames/s", "ms_per_step": dt_fa / fa_steps * 1e3,
                                   "roofline": fa_roof,
                                   "pose_err_vs_cpu_ref": {"max_rot_rad_over_scenes": float(fa_err[:, 0].max()),
                                                           "max_trans_m_over_scenes": float(fa_err[:, 1].max()),
                                                           "n_tracked_equal_in_every_scene": bool(all(int(r_fa[i].n_tracked) == int(oracle_res[i].n_tracked) for i in range(n_scenes)))}}
                assert fa_err[:, 0].max() < 1e-4 and fa_err[:, 1].max() < 1e-3, "%s-arithmetic pose parity violated: %s" % (lvl_name, fa_err.max(axis=0))
            fast, m32_sec = other["fast"], other["moments_f32"]
            # (2) the streaming implementation of the Jacobian / residual pass: the HBM-bound form (north_star: >= 50 % of the HBM roofline)
            sia.set_mode(stream=True)           # an option of this solver object (svo_hip_sia_set_option), not a process-wide switch
            try:
                prewarm(ctx, lambda: sia.run(n_slots, prm))
                sia.set_profiling(True)
                sia.run(n_slots, prm)
                sp = sia.get_profile()
                sia.set_profiling(False)
            finally:
                sia.set_mode(stream=False)
            # (3) upload-inclusive rate: a new current image per pair per step over PCIe, overlapped with the solve
            upl, r_up = with_uploads(torch, ctx, stream, local_rank, sia, ref, fps, n_slots, prm, max(5, min(args.steps, 20)))
            rot_u, trans_u = synth.pose_error(np.array(r_up.T_cur_w), np.array(o.T_cur_w))
            upl["pose_err_vs_cpu_ref"] = {"rot_rad": rot_u, "trans_m": trans_u}
            assert rot_u < 1e-4 and trans_u < 1e-3
            sia.set_frames(ref, cur)
            if sp["residual_launches"]:
                s_avg = sp["residual_ms"] / sp["residual_launches"] * 1e-3
                units = (n_res_per_frame / max(evals, 1)) * n_slots
                jac = {"kernel": "sia_residual_kernel", "bound": "hbm", "avg_launch_us": s_avg * 1e6, "launches": int(sp["residual_launches"]),
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": None, "frac": None, "traffic": None,
                       "algorithmic": {"bytes_per_launch": units * BYTES_RESIDUAL, "GBps": units * BYTES_RESIDUAL / s_avg / 1e9},
                       "note": "one launch = one Gauss-Newton evaluation of %d frame pairs: per-pixel f32 caches + {x,y,z,1/z} + "
                               "{sum dx^2, sum dx dy, sum dy^2} streamed from HBM (227 B/patch instead of the reference layout's 881 B)" % n_slots}
                sfile = latest_profile("r*_pmc_stream.json")
                sc = pmc_of(sfile, "sia_residual_kernel", args.allow_stale_profile, pairs=n_slots) if (sfile and default_c1) else None
                if isinstance(sc, str):
                    jac["profile_refused"] = sc
                    sc = None
                if sc:
                    phys = (2.0 * sc.get("FETCH_SIZE", 0.0) + sc.get("WRITE_SIZE", 0.0)) * 1024.0
                    jac.update({"achieved": phys / s_avg / 1e9, "frac": phys / s_avg / 1e9 / HBM_PEAK_GBS, "traffic": phys,
                                "traffic_rule": "2 x FETCH_SIZE + WRITE_SIZE: every stream of this kernel is a 16 B/lane coalesced read, the case the "
                                                "gfx950 FETCH_SIZE correction is calibrated for (MI355X_MICROARCH.md, HBM)",
                                "sources": [sc["_file"]]})
        # ---- the other single-GPU configurations of BASELINE.json in compact form (outside the timed region): C2 =
        # align2D x 5000 patches + DepthFilter update x 100 k seeds per frame; C4 on ONE GPU = 1 M seeds on a 1280x720
        # keyframe incl. the on-device packing of the converged records (everything but the exchange).  `frac_hbm` is
        # SURVEY 8(d)'s algorithmic bytes over the HBM peak: at these sizes the kernels are launch- / latency-bound.
        c2 = None
        c4 = None
        if not allreduce and world == 1 and not args.no_secondary and not args.early_stop:
            import bench_c2
            c2 = {"what": "BASELINE config C2 on this GPU, inputs resident in HBM; per-call time = best of 3 rounds of back-to-back calls after 30 ms of warm-up (python bench_c2.py gives the long form)"}
            c2["align2d"], _ = bench_c2.measure_align2d(ctx, 5000, steps=20, warmup=3)
            # 5000 patches are 313 waves on 1024 SIMDs -- a launch-latency figure; 200 000 patches say what the kernel sustains
            a200, _ = bench_c2.measure_align2d(ctx, 200000, steps=20, warmup=3)
            c2["align2d_200k"] = {k_: a200[k_] for k_ in ("patches", "patches_per_s", "us_per_batch", "mean_iters", "frac_hbm")}
            c2["depth_filter"], _, sb2, pyr2 = bench_c2.measure_depth_filter(ctx, 100000, steps=20, warmup=3)
            sb2.free()
            [p_.destroy() for p_ in pyr2]
            pf2 = latest_profile("r*_pmc_df_c2.json")
            if pf2 and c2["depth_filter"].get("stages_us"):
                c2["depth_filter"]["roofline"] = bench_c2.stage_rooflines(c2["depth_filter"]["stages_us"], pf2)
            # the SAME pass through the entry the drop-in DepthFilter calls per frame (device-resident seed batch: poses down,
            # events back, one wait), host side included -- what ships, beside the resident-array figure above
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import df_hostentry_bench
            he = df_hostentry_bench.measure(ctx, 100000, reps=20)
            c2["depth_filter"]["through_dropin_entry_us"] = he["resident"]["ms_per_call_median"] * 1e3
            c2["depth_filter"]["through_dropin_entry_keyframe_us"] = he["resident_keyframe"]["ms_per_call_median"] * 1e3
            c2["depth_filter"]["through_round3_host_buffer_entry_us"] = he["host"]["ms_per_call_median"] * 1e3      # (verbose line only)
            c4, _, sb4, pyr4 = bench_c2.measure_depth_filter(ctx, 1000000, steps=10, warmup=2, width=1280, height=720, sigma_scale=0.0045,
                                                             compact=True)
            pf4 = latest_profile("r*_pmc_df_c4.json")
            if pf4 and c4.get("stages_us"):
                c4["roofline"] = bench_c2.stage_rooflines(c4["stages_us"], pf4)
            c4["what"] = ("BASELINE config C4 on ONE GPU: DepthFilter::updateSeeds over 1 M seeds of a 1280x720 keyframe + packing of the "
                          "converged records on the device; the multi-GPU form (seeds sharded, RCCL gather) is bench_c4.py")
            sb4.free()
            [p_.destroy() for p_ in pyr4]
        c0 = None
        if not args.no_cpu_baseline and world == 1 and not allreduce and not args.no_secondary and not args.early_stop and args.width == 640:
            c0 = c0_leg(ctx)
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # reported at N=1 only (rank 0)
            cpu = cpu_baseline(fps[:16], n_iter=30, early_stop=args.early_stop, frames_per_thread=args.cpu_frames_per_thread)
            if early is not None:
                # the reference's OWN compiled SparseImgAlign (oracle/_ref, kind "reference") on the early-stop workload
                early["cpu_baseline"] = cpu_baseline(fps[:16], n_iter=30, early_stop=True, frames_per_thread=5 * args.cpu_frames_per_thread)
        out = {
            "metric": "SparseImgAlign frames/s at %dx%d L4-L0; pose err vs CPU ref" % (args.width, args.height),
            **({"rehearsal": "%d ranks share %d GPU(s): gloo process group%s -- a code-path rehearsal, not a scaling measurement" %
                             (world, n_dev, ", shared-memory exchange instead of RCCL" if allreduce else "")} if rehearsal else {}),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 normal equations / f32 image math" + {"default": " (as the reference)", "moments_f32": " (f32 moment sums per patch)",
                                                                 "fast": " (contracted, f32 sums per patch)"}[args.arith], "data": "synthetic",
            "config": {"workload": "%s: SparseImgAlign %dx%d, %d patches, 5 pyramid levels (L4-L0), %s" %
                                   ("C1" if args.width == 640 else "C3 shape", args.width, args.height, n_feat, "reference early-stop GN" if args.early_stop else "30 GN evaluations per level (fixed work)"),
                       "frame_pairs_per_gpu_per_step": B, "global_frame_pairs_per_step": frames_global,
                       "parallelism": ("patch-sharded + per-GN-step all-reduce of H/b (C3 variant), " +
                                       ("RCCL called by libsvo_hip.so (svo_hip_sia_run_sharded)" if native else "torch.distributed driver") +
                                       (", HIP-graph replay" if args.graph else "") if allreduce
                                       else "frame-parallel (no collective)") + ", %d GPU(s)" % world,
                       "distinct_scenes": args.distinct, "arithmetic": {"default": "EXACT (library default, the reference's)", "moments_f32": "MOMENTS_F32 (opt-in)", "fast": "FAST (opt-in)"}[args.arith],
                       **({"comm_ranks": comm_ranks} if comm_ranks is not None else {}),
                       "implementation": "fused" if (not allreduce and mode == 1) else "streaming"},     # one workgroup per pair, one launch per solve / one launch per GN evaluation
            "pose_err_vs_cpu_ref": {"rot_rad": rot, "trans_m": trans, "tolerance": "1e-4 rad / 1e-3 m",
                                    "scenes_checked": int(n_scenes), "max_rot_rad_over_scenes": float(scene_err[:, 0].max()),
                                    "max_trans_m_over_scenes": float(scene_err[:, 1].max()), "n_tracked_equal_in_every_scene": bool(tracked_equal),
                                    "replica_slots_bitwise_equal": int(replicas_checked)},
            "pose_err_vs_ground_truth": dict(zip(("rot_rad", "trans_m"), synth.pose_error(np.array(res.T_cur_w), fps[0].T_cur_w_true))),
            "gn_evaluations_per_frame": int(evals),
            "single_pair_latency_ms_early_stop": latency_ms,
            "single_stream_chain_ms": chain["L4_L2_shipping_default"]["ms_per_frame"] if chain else None,
            "single_stream_chain": chain,
            "algorithmic_bytes_per_frame": int(bytes_frame),
            "whole_solve_algorithmic_GBps": bytes_frame * value / 1e9,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "roofline_jacobian_pass": jac,
            "reference_semantics": early,
            "moments_f32_arithmetic": m32_sec,
            "fast_arithmetic": fast,
            "with_image_uploads": upl,
            "c0": c0,
            "c2": c2,
            "c4_one_gpu": c4,
        }
        assert scene_err[:, 0].max() < 1e-4 and scene_err[:, 1].max() < 1e-3, "pose parity violated: %s (rotation error per scene: %s)" % (scene_err.max(axis=0), np.array2string(scene_err[:, 0], precision=2))
        assert tracked_equal or args.early_stop, "n_tracked differs from the oracle's in some scene"
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(slim(out, args.verbose_json), separators=(",", ":")) + "\n").encode())
    if multi:
        dist.barrier()
        graphed = None          # captured graphs hold RCCL kernels: release them before the communicator goes away
        aligner = None
        if comm is not None:
            ctx.sync()
            comm.destroy()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
