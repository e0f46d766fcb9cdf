#!/usr/bin/env python3
"""bench_c2.py -- secondary metrics of SURVEY.md 8(d), BASELINE config C2 (and C4 shape with --seeds 1000000):
feature_alignment::align2D over 5000 8x8 patches and one DepthFilter::updateSeeds pass over 100k seeds
per frame on one MI355X, inputs resident in HBM, next to the CPU oracle on the host cores.
Prints one JSON line.  (bench.py is the headline metric; this file is not run by the driver.)"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from android_svo_amd import hip, seedsynth  # noqa: E402

HBM_PEAK_GBS = 8000.0
N_SIMD = 1024                     # 256 CUs x 4 SIMD-32
PEAK_CLOCK_GHZ = 2.4
DF_STAGES = (("geometry", "df_geometry_kernel"), ("search", "df_search_kernel"), ("align", "df_align_kernel"), ("finalize", "df_finalize_kernel"))


def stage_times(ctx, run_pass, repeats=5):
    """{stage: microseconds} of the depth-filter pass, measured live with HIP events on the context stream between the
    stages (svo_hip_df_set_profiling; the alignment stage's three launches count as one stage): the pass with the smallest
    total of `repeats`."""
    ctx.check(ctx.lib.svo_hip_df_set_profiling(ctx.h, 1), "df_set_profiling")
    best = None
    try:
        for _ in range(repeats):
            run_pass()
            us = (C.c_double * 4)()
            ctx.check(ctx.lib.svo_hip_df_get_profile(ctx.h, us), "df_get_profile")
            best = list(us) if best is None or sum(us) < sum(best) else best
    finally:
        ctx.check(ctx.lib.svo_hip_df_set_profiling(ctx.h, 0), "df_set_profiling")
    return {name: best[k] for k, (name, _) in enumerate(DF_STAGES)}


def stage_rooflines(stages_us, pmc_path):
    """Per stage of the depth-filter pass (kernel df_<stage>_kernel; its live time is stages_us[stage]): {bound, frac, valu_frac,
    hbm_frac, lane_util}.
    lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) -- the share of a VALU instruction's 64 lanes that were
    switched on (a quad whose seed has converged, a thread past its last ZMSSD step ride along masked) -- when the profile holds
    the two counters; valu_frac is then the USEFUL fraction, issue fraction x lane_util (the issue fraction alone counts a
    masked lane as busy; it is valu_frac / lane_util).
    issue fraction = VALU issue cycles of the stage's launches (per-type instruction counters of the rocprofv3 PMC passes in
    `pmc_path`, made by tools/pmc_c2.sh: 2 cycles per wave64 f32/int instruction, 4 per f64 and per conversion, 8 per
    transcendental -- bench.py's rule for the fused SparseImgAlign kernel) / (1024 SIMDs x 2.4 GHz x the stage's live time);
    hbm_frac = (2 x FETCH_SIZE + WRITE_SIZE) KB of the same passes / live time / 8 TB/s.  `bound` names the larger one.
    A profile taken from other kernel sources than the tree's is refused (returns the reason as a string)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_json
    try:
        d = json.load(open(pmc_path))
    except Exception:
        return None
    have, want = d.get("source_sha256"), pmc_json.source_sha256("df")
    if have != want:
        changed = sorted(k for k in want if not have or have.get(k) != want[k])
        return "%s was taken from other kernel sources (%s): re-run tools/pmc_c2.sh" % (os.path.relpath(pmc_path, ROOT), ", ".join(changed))
    passes = float(d.get("passes", 4))
    out = {}
    for name, prefix in DF_STAGES:
        ctr = None
        for kname, c in d.get("kernels", {}).items():
            if kname.startswith(prefix):
                ctr = c
                break
        if ctr is None or name not in stages_us:
            continue
        per_pass = {k: v * ctr.get("_dispatches", {}).get(k, passes) / passes for k, v in ctr.items() if isinstance(v, (int, float))}
        f64 = per_pass.get("SQ_INSTS_VALU_ADD_F64", 0.0) + per_pass.get("SQ_INSTS_VALU_MUL_F64", 0.0) + per_pass.get("SQ_INSTS_VALU_FMA_F64", 0.0)
        trans = per_pass.get("SQ_INSTS_VALU_TRANS_F64", 0.0) + per_pass.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        cvt = per_pass.get("SQ_INSTS_VALU_CVT", 0.0)
        cyc = 2.0 * (per_pass["SQ_INSTS_VALU"] - f64 - trans - cvt) + 4.0 * (f64 + cvt) + 8.0 * trans
        t = stages_us[name] * 1e-6
        valu = cyc / (N_SIMD * PEAK_CLOCK_GHZ * 1e9 * t)
        lane_util = None
        if per_pass.get("SQ_THREAD_CYCLES_VALU") and per_pass.get("SQ_ACTIVE_INST_VALU"):
            lane_util = min(1.0, per_pass["SQ_THREAD_CYCLES_VALU"] / (64.0 * per_pass["SQ_ACTIVE_INST_VALU"]))
            valu *= lane_util
        phys = (2.0 * per_pass.get("FETCH_SIZE", 0.0) + per_pass.get("WRITE_SIZE", 0.0)) * 1024.0
        hbm = phys / t / 1e9 / HBM_PEAK_GBS
        r4 = lambda v: float("%.4g" % v)
        out[name] = {"bound": "valu" if valu >= hbm else "hbm", "frac": r4(max(valu, hbm)), "valu_frac": r4(valu), "hbm_frac": r4(hbm)}
        if lane_util is not None:
            out[name]["lane_util"] = r4(lane_util)
    out["source"] = os.path.relpath(pmc_path, ROOT)
    return out


TIMED_REPEATS = 3
TIMED_MIN_WARM_S = 0.03


def timed(ctx, fn, steps, warmup, repeats=TIMED_REPEATS):
    """seconds per call: `steps` calls enqueued back to back, one synchronisation; the best of `repeats` such rounds, after
    at least `warmup` calls and TIMED_MIN_WARM_S of device work.  These passes last 20 us - 1 ms and follow seconds of
    host-side case generation with the GPU idle: a 5-step round of the 0.7 ms C4 pass was once measured at 15 ms per step
    (22 x) right after such a pause, with two warm-up calls -- the figure of an idle chip, not of the kernels."""
    for _ in range(warmup):
        fn()
    ctx.sync()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < TIMED_MIN_WARM_S:
        fn()
        ctx.sync()
    best = None
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        ctx.sync()
        dt = (time.perf_counter() - t0) / steps
        best = dt if best is None or dt < best else best
    return best


def measure_align2d(ctx, patches=5000, steps=20, warmup=3):
    """feature_alignment::align2D over `patches` 8x8 patches resident in HBM (BASELINE config C2, first half).  Returns the
    result dict and the case (for the CPU leg)."""
    ac = seedsynth.make_align_case(n=patches)
    pyr = hip.Pyramid(ctx, ac.cam.width, ac.cam.height, 5, 1)
    pyr.upload(0, ac.cur_pyr)
    d_pwb = ctx.to_device(ac.pwb)
    d_px0 = ctx.to_device(ac.px_init)
    d_px = ctx.empty(ac.px_init.shape, np.float64)
    d_conv = ctx.empty((patches,), np.uint8)
    d_it = ctx.empty((patches,), np.int32)

    def run_align():
        # restore the initial estimates (device to device) then refine
        ctx.check(ctx.lib.svo_hip_copy_d2d(ctx.h, C.c_void_p(d_px.ptr), C.c_void_p(d_px0.ptr), C.c_size_t(d_px.nbytes)), "d2d")
        ctx.check(ctx.lib.svo_hip_align2d_batch_dev(ctx.h, pyr.h, 0, 0, patches, C.c_void_p(d_pwb.ptr), None, 10,
                                                    C.c_void_p(d_px.ptr), C.c_void_p(d_conv.ptr), C.c_void_p(d_it.ptr)), "align2d")
    t_align = timed(ctx, run_align, steps, warmup)
    iters = d_it.download()
    alg_align = float(np.sum(197 + 81 * iters))
    res = {"patches": patches, "patches_per_s": patches / t_align, "us_per_batch": t_align * 1e6,
           "converged": int(d_conv.download().sum()), "mean_iters": float(iters.mean()),
           "algorithmic_bytes": alg_align, "algorithmic_GBps": alg_align / t_align / 1e9,
           "frac_hbm": alg_align / t_align / 1e9 / HBM_PEAK_GBS}
    for d in (d_pwb, d_px0, d_px, d_conv, d_it):
        d.free()
    pyr.destroy()
    return res, ac


def measure_depth_filter(ctx, seeds=100000, steps=20, warmup=3, width=640, height=480, sigma_scale=None, compact=False):
    """One DepthFilter::updateSeeds pass over `seeds` seeds resident in HBM (BASELINE config C2, second half; with
    width=1280, height=720, seeds=1000000, compact=True the one-GPU form of config C4: the pass plus the on-device packing
    of the converged records, i.e. everything but the exchange).  Returns (result dict, case, SeedBatch, pyramids)."""
    sc = seedsynth.make_seed_case(n_seeds=seeds, seed=9, width=width, height=height)
    kf = hip.Pyramid(ctx, sc.cam.width, sc.cam.height, 5, 1)
    cf = hip.Pyramid(ctx, sc.cam.width, sc.cam.height, 5, 1)
    kf.upload(0, sc.ref_pyr)
    cf.upload(0, sc.cur_pyr)
    sigma2 = sc.sigma2 if sigma_scale is None else (sc.sigma2 * np.float32(sigma_scale)).astype(np.float32)
    sb = hip.SeedBatch(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, sigma2)
    state, state0 = hip.pack_seed_state(sb)          # a | b | mu | sigma2 in one block: one copy restores all seeds
    rec = cnt = None
    if compact:
        rec, cnt = ctx.empty((seeds, 6), np.float64), ctx.empty((1,), np.int32)

    def run_df():
        # same seed state every step (one 16 B/seed device-to-device copy), then the pass
        ctx.check(ctx.lib.svo_hip_copy_d2d(ctx.h, C.c_void_p(state.ptr), C.c_void_p(state0.ptr), C.c_size_t(state.nbytes)), "d2d")
        hip.depth_filter_update(ctx, kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w, sb)
        if compact:
            ctx.check(ctx.lib.svo_hip_seed_compact_converged_dev(ctx.h, seeds, C.c_longlong(0), C.c_void_p(sb.status.ptr), C.c_void_p(sb.mu.ptr),
                                                                 C.c_void_p(sb.sigma2.ptr), C.c_void_p(sb.xyz.ptr), C.c_void_p(rec.ptr),
                                                                 C.c_void_p(cnt.ptr)), "seed_compact_converged")
    t_df = timed(ctx, run_df, steps, warmup)
    stages = None
    try:
        stages = stage_times(ctx, run_df)
    except (AttributeError, hip.SvoHipError):
        pass                                  # a library build without the hook (A/B runs against an older build)
    nz, na, st = sb.n_zmssd.download(), sb.n_align.download(), sb.status.download()
    alg_df = float(np.sum(44 + 56 + 100 + 64 * nz.astype(np.int64) + 81 * na.astype(np.int64)))
    res = {"seeds": seeds, "image": "%dx%d" % (width, height), "seeds_per_s": seeds / t_df, "us_per_frame": t_df * 1e6,
           "status_counts": np.bincount(st, minlength=6).tolist(), "mean_zmssd": float(nz.mean()),
           "mean_align_iters": float(na.mean()), "algorithmic_bytes": alg_df,
           "algorithmic_GBps": alg_df / t_df / 1e9, "frac_hbm": alg_df / t_df / 1e9 / HBM_PEAK_GBS}
    if stages:
        res["stages_us"] = stages
    if compact:
        res["converged_records_packed"] = int(cnt.download()[0])
        rec.free(); cnt.free()
    return res, sc, sb, (kf, cf)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patches", type=int, default=5000)
    ap.add_argument("--seeds", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    from oracle import orc

    ctx = hip.Context(0)
    out = {"config": {"workload": "C2: align2D x %d patches + DepthFilter update x %d seeds, 640x480" % (args.patches, args.seeds)}}

    # ---------------- align2D ----------------
    out["align2d"], ac = measure_align2d(ctx, args.patches, args.steps, args.warmup)

    # ---------------- depth filter ----------------
    out["depth_filter"], sc, sb, (kf, cf) = measure_depth_filter(ctx, args.seeds, args.steps, args.warmup)

    # pure Bayes update (44 B/seed)
    n = args.seeds
    x = ctx.to_device((sc.mu + 0.01).astype(np.float32))
    tau2 = ctx.to_device(np.full(n, 1e-2, dtype=np.float32))

    def run_us():
        ctx.check(ctx.lib.svo_hip_update_seed_batch_dev(ctx.h, n, C.c_void_p(x.ptr), C.c_void_p(tau2.ptr), C.c_void_p(sb.a.ptr),
                                                        C.c_void_p(sb.b.ptr), C.c_void_p(sb.mu.ptr), C.c_void_p(sb.z_range.ptr),
                                                        C.c_void_p(sb.sigma2.ptr)), "update_seed")
    t_us = timed(ctx, run_us, args.steps, args.warmup)
    out["update_seed"] = {"seeds_per_s": n / t_us, "us": t_us * 1e6, "algorithmic_GBps": 44.0 * n / t_us / 1e9,
                          "frac_hbm": 44.0 * n / t_us / 1e9 / HBM_PEAK_GBS}

    # ---------------- next rows f-4: pose refinement (1200 observations/frame) and structure refinement ----------------
    from android_svo_amd import synth
    pcs = [synth.make_pose_opt_case(seed=40 + k, n=1200) for k in range(4)]
    em = abs(pcs[0].cam.fx)
    for B in (1, 256):
        T = np.stack([pcs[k % 4].T_f_w_init for k in range(B)])
        f = np.stack([pcs[k % 4].f for k in range(B)]); pos = np.stack([pcs[k % 4].pos for k in range(B)])
        lvl = np.stack([pcs[k % 4].level for k in range(B)]); hp0 = np.stack([pcs[k % 4].has_point for k in range(B)])
        d = [ctx.to_device(hip._f64(T)), ctx.to_device(hip._f64(f)), ctx.to_device(hip._f64(pos)),
             ctx.to_device(np.ascontiguousarray(lvl, dtype=np.int32)), ctx.to_device(np.ascontiguousarray(hp0, dtype=np.uint8)),
             ctx.to_device(np.full(B, 1200, dtype=np.int32)), ctx.to_device(np.ascontiguousarray(hp0, dtype=np.uint8))]
        dres = ctx.empty((B * C.sizeof(hip.CPoseOptResult),), np.uint8)

        def run_pose():
            ctx.check(ctx.lib.svo_hip_copy_d2d(ctx.h, C.c_void_p(d[4].ptr), C.c_void_p(d[6].ptr), C.c_size_t(B * 1200)), "restore")
            ctx.check(ctx.lib.svo_hip_pose_optimize_batch_dev(ctx.h, B, 1200, C.c_void_p(d[5].ptr), C.c_void_p(d[0].ptr), C.c_void_p(d[1].ptr),
                                                              C.c_void_p(d[2].ptr), C.c_void_p(d[3].ptr), C.c_void_p(d[4].ptr), C.c_double(em),
                                                              C.c_double(2.0), 10, C.c_void_p(dres.ptr)), "pose_optimize")
        t_p = timed(ctx, run_pose, args.steps, args.warmup)
        out["pose_refine_B%d" % B] = {"frames_per_s": B / t_p, "us_per_launch": t_p * 1e6, "observations_per_frame": 1200}
        for v in d + [dres]:
            v.free()
    pos0, off, Ts, fs, _, _ = synth.make_point_opt_cases(n_points=20000)
    dp0, dp, do, dT, dF = ctx.to_device(hip._f64(pos0)), ctx.to_device(hip._f64(pos0)), ctx.to_device(off), ctx.to_device(hip._f64(Ts)), ctx.to_device(hip._f64(fs))

    def run_points():
        ctx.check(ctx.lib.svo_hip_copy_d2d(ctx.h, C.c_void_p(dp.ptr), C.c_void_p(dp0.ptr), C.c_size_t(pos0.nbytes)), "restore")
        ctx.check(ctx.lib.svo_hip_point_optimize_batch_dev(ctx.h, len(pos0), 5, C.c_void_p(dp.ptr), C.c_void_p(do.ptr), C.c_void_p(dT.ptr),
                                                           C.c_void_p(dF.ptr), None), "point_optimize")
    t_pt = timed(ctx, run_points, args.steps, args.warmup)
    out["point_refine"] = {"points_per_s": len(pos0) / t_pt, "us_per_launch": t_pt * 1e6, "points": len(pos0), "observations": int(off[-1])}
    if not args.no_cpu_baseline:
        t0 = time.perf_counter()
        for k in range(20):
            orc.pose_optimize(em, pcs[k % 4].T_f_w_init, pcs[k % 4].f, pcs[k % 4].pos, pcs[k % 4].level, pcs[k % 4].has_point)
        t_cp = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for i in range(2000):
            orc.point_optimize(pos0[i], Ts[off[i]:off[i + 1]], fs[off[i]:off[i + 1]], n_iter=5)
        t_cq = (time.perf_counter() - t0) / 2000
        out["refine_cpu_port_1thread"] = {"pose_refine_frames_per_s": 1.0 / t_cp, "point_refine_points_per_s_incl_ctypes": 1.0 / t_cq}

    # ---------------- next row f-2: the cell loop of Reprojector::reprojectMap (3000 candidates, 768 cells) ----------------
    rc = synth.make_reproject_case(seed=21, width=640, height=480, n_points=3000)
    rref = hip.Pyramid(ctx, 640, 480, 5, 3)
    rcur = hip.Pyramid(ctx, 640, 480, 5, 1)
    for k in range(3):
        rref.upload(k, rc["kf_pyr"][k])
    rcur.upload(0, rc["cur_pyr"])
    off, ids = synth.flatten_cells(rc, rc["trial"])
    rdel = (rc["ptype"][ids] == synth.TYPE_DELETED).astype(np.uint8)
    rargs = (rc["T_kf_w"], rc["T_cur_w"], off, rc["slot"][ids], rc["px_ref"][ids], rc["f_ref"][ids], rc["level"][ids], rc["pos"][ids], rdel, rc["px_cur"][ids])

    def run_reproject():
        return hip.reproject_cells(ctx, rref, rcur, 0, rc["cam"], *rargs)
    t_rp = timed(ctx, run_reproject, args.steps, args.warmup)
    rres = run_reproject()
    out["reproject_cells"] = {"us_per_frame_incl_host_copies": t_rp * 1e6, "candidates": int(len(ids)), "cells": int(rc["n_cells"]),
                              "n_matches": int(rres["n_matches"]), "n_trials": int(rres["n_trials"])}
    if not args.no_cpu_baseline:
        t0 = time.perf_counter()
        for _ in range(3):
            orc.reproject_cells(rc["cam"], rc["kf_pyr"], rc["T_kf_w"], rc["cur_pyr"], rc["T_cur_w"], off, rc["slot"][ids], rc["px_ref"][ids],
                                rc["f_ref"][ids], rc["level"][ids], rc["pos"][ids], np.zeros(len(ids), np.uint8), np.tile([1.0, 0.0], (len(ids), 1)),
                                rdel, rc["px_cur"][ids])
        out["reproject_cells"]["cpu_port_us_per_frame_1thread"] = (time.perf_counter() - t0) / 3 * 1e6

    # ---------------- next row f-3: FastDetector::detect on a 640x480 keyframe (3 levels, 20 px cells) ----------------
    dn, dpx, df_, dl, dsc = ctx.empty((1,), np.int32), ctx.empty((768, 2), np.float64), ctx.empty((768, 3), np.float64), ctx.empty((768,), np.int32), ctx.empty((768,), np.float32)
    ccam = hip.make_camera(sc.cam)

    def run_detect():
        ctx.check(ctx.lib.svo_hip_detect_features_dev(ctx.h, kf.h, 0, C.byref(ccam), 3, 20, None, C.c_double(10.0), C.c_void_p(dn.ptr),
                                                      C.c_void_p(dpx.ptr), C.c_void_p(df_.ptr), C.c_void_p(dl.ptr), C.c_void_p(dsc.ptr)), "detect")
    t_det = timed(ctx, run_detect, args.steps, args.warmup)
    px_bytes = sum((640 >> l) * (480 >> l) for l in range(3))
    out["detect_features"] = {"us_per_keyframe": t_det * 1e6, "features": int(dn.download()[0]), "launches": 8,
                              "algorithmic_bytes": 3 * px_bytes, "note": "reads each level twice (decision, suppression) and writes its score image once"}
    if not args.no_cpu_baseline:
        t0 = time.perf_counter()
        for _ in range(5):
            orc.detect_features(sc.ref_pyr)
        out["detect_features"]["cpu_port_us_per_keyframe_1thread"] = (time.perf_counter() - t0) / 5 * 1e6

    # ---------------- CPU oracle (bounded sample) ----------------
    if not args.no_cpu_baseline:
        n_thr = max(1, min(os.cpu_count() or 1, 16))
        m = min(args.patches, 5000)
        t0 = time.perf_counter()
        for i in range(m):
            orc.align2d(ac.cur_pyr[0], ac.pwb[i], ac.patch[i], 10, ac.px_init[i])
        t_cpu_a = (time.perf_counter() - t0) / m
        ns = min(args.seeds, 20000)
        sl = [slice(t * ns // n_thr, (t + 1) * ns // n_thr) for t in range(n_thr)]

        def work(s):
            a, b, mu, s2 = (v[s].copy() for v in (sc.a, sc.b, sc.mu, sc.sigma2))
            orc.update_seeds(sc.cam, sc.ref_pyr, sc.cur_pyr, sc.T_ref_w, sc.T_cur_w, sc.px[s], sc.f[s], sc.level[s], a, b, mu,
                             sc.z_range[s].copy(), s2)
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(s,)) for s in sl]
        [t.start() for t in th]
        [t.join() for t in th]
        t_cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {"kind": "port", "align2d_patches_per_s_1thread_incl_ctypes": 1.0 / t_cpu_a,
                               "depth_filter_seeds_per_s": ns / t_cpu_s, "cores": n_thr,
                               "sample": "%d patches on 1 thread (python loop around the C call), %d seeds on %d threads" % (m, ns, n_thr)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
