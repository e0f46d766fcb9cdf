#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own code (oracle/_ref/libsvo_ref.so).

Run only where /root/reference is mounted:   make -C oracle ref && python oracle/gen_golden.py
The fixtures hold inputs and the reference's outputs (data only, no reference text).
"""
from __future__ import annotations

import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from android_svo_amd import synth  # noqa: E402
from oracle.ref import refpy  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def rand_pose(rng, t=0.5, r=0.5):
    return synth.se3_from_twist(rng.uniform(-t, t, 3), rng.uniform(-r, r, 3))


def gen_se3(rng):
    n = 64
    A = np.stack([rand_pose(rng) for _ in range(n)])
    B = np.stack([rand_pose(rng, 2.0, 3.0) for _ in range(n)])
    p = rng.uniform(-3, 3, (n, 3))
    tw = rng.uniform(-0.3, 0.3, (n, 6))
    tw[0, 3:] = [1e-12, -2e-12, 1e-12]      # small-angle branch of the rotation part
    tw[1, 3:] = 0.0                          # theta == 0: NaN translation quirk
    tw[2] *= 1e-7
    q = B[:, 3:].copy()
    q[3] = [0, 0, 0, 1.0]                    # n < NEAR_ZERO branch of SO3::log
    q[4] = [0.6, 0.0, 0.8, 1e-12]            # |w| < NEAR_ZERO branch
    np.savez_compressed(
        os.path.join(OUT, "se3.npz"), A=A, B=B, p=p, tw=tw, q=q,
        mul=np.stack([refpy.se3_mul(a, b) for a, b in zip(A, B)]),
        inv=np.stack([refpy.se3_inverse(a) for a in A]),
        act=np.stack([refpy.se3_act(a, x) for a, x in zip(A, p)]),
        exp=np.stack([refpy.se3_exp(x) for x in tw]),
        log=np.stack([refpy.so3_log(x) for x in q]),
        rot=np.stack([refpy.rotation_matrix(a) for a in A]))


def gen_algebra(rng):
    n = 64
    xyz = rng.uniform(-1, 1, (n, 3)) + [0, 0, 2.5]
    J = np.stack([refpy.jacobian_xyz2uv(x) for x in xyz])
    Hs, bs = [], []
    for i in range(48):
        M = rng.normal(size=(40, 6)) * rng.uniform(0.1, 100, 6)
        H = M.T @ M
        if i % 8 == 7:
            H[:, 5] = 0; H[5, :] = 0          # rank deficient (pseudo-inverse branch)
        if i == 40:
            H[:] = 0                          # all-zero matrix
        Hs.append(H.reshape(36)); bs.append(rng.normal(size=6) * 100)
    Hs, bs = np.stack(Hs), np.stack(bs)
    x = np.stack([refpy.ldlt6_solve(h, b) for h, b in zip(Hs, bs)])
    np.savez_compressed(os.path.join(OUT, "algebra.npz"), xyz=xyz, J=J, H=Hs, b=bs, x=x)


def gen_gn(rng):
    """Reference NLLSSolver/Eigen/SE3 driving the restated residual body."""
    fp = synth.make_frame_pair(seed=777, width=320, height=240, n_features=120, null_point_every=7,
                               t_mag=0.015, r_mag=0.006)
    d = refpy.driven_sparse_align(fp)
    d2 = refpy.driven_sparse_align(fp, max_level=4, min_level=2)
    save = dict(width=320, height=240, px=fp.px, f=fp.f, pos=fp.pos, has_point=fp.has_point,
                T_ref_w=fp.T_ref_w, T_cur_w_init=fp.T_cur_w_init, T_cur_w_true=fp.T_cur_w_true,
                cam=np.array([fp.cam.fx, fp.cam.fy, fp.cam.cx, fp.cam.cy]),
                T_out=d["T_cur_w"], n_tracked=d["n_tracked"], chi2=d["chi2"], iters=d["iters"], H=d["H"],
                T_out_l2=d2["T_cur_w"], n_tracked_l2=d2["n_tracked"], chi2_l2=d2["chi2"], iters_l2=d2["iters"])
    for l in range(5):
        save["ref%d" % l] = fp.ref_pyr[l]
        save["cur%d" % l] = fp.cur_pyr[l]
    np.savez_compressed(os.path.join(OUT, "gn_small.npz"), **save)
    # full-size cases by generator seed; image checksums guard against generator drift
    cases = []
    for seed, n in ((12345, 200), (12346, 2000), (12347, 1200)):
        fp = synth.make_frame_pair(seed=seed, n_features=n)
        d = refpy.driven_sparse_align(fp)
        d2 = refpy.driven_sparse_align(fp, max_level=4, min_level=2)
        cases.append(dict(seed=seed, n=n, crc_ref=crc(fp.ref_pyr[0]), crc_cur=crc(fp.cur_pyr[0]),
                          crc_px=crc(fp.px), T_out=d["T_cur_w"], n_tracked=d["n_tracked"], chi2=d["chi2"],
                          iters=d["iters"], T_out_l2=d2["T_cur_w"], iters_l2=d2["iters"],
                          T_true=fp.T_cur_w_true))
    np.savez_compressed(os.path.join(OUT, "gn_full.npz"),
                        **{k: np.array([c[k] for c in cases]) for k in cases[0]})


def small_scene(seed=4242, w=160, h=120):
    cam = synth.Camera(w, h, 125.0, 125.0, w / 2 - 0.5, h / 2 - 0.5)
    scene = synth.PlaneScene(seed=seed, depth=2.0)
    T0 = synth.se3_from_twist([0, 0, 0], [0, 0, 0])
    T1 = synth.se3_from_twist([0.06, -0.02, 0.01], [0.004, -0.006, 0.01])
    return cam, scene, T0, T1, scene.render(cam, T0), scene.render(cam, T1)


def gen_align(rng):
    cam, scene, T0, T1, ref, cur = small_scene()
    n = 256
    pwb = np.zeros((n, 100), dtype=np.uint8)
    patch = np.zeros((n, 64), dtype=np.uint8)
    px_in = np.zeros((n, 2))
    n_iter = np.full(n, 10, dtype=np.int32)
    for i in range(n):
        cx, cy = rng.integers(8, cam.width - 8), rng.integers(8, cam.height - 8)
        pwb[i] = ref[cy - 5:cy + 5, cx - 5:cx + 5].reshape(100)
        patch[i] = refpy.patch_from_border(pwb[i])
        # the same world point seen in `cur`, perturbed
        X = scene.intersect(cam, T0, np.array([float(cx)]), np.array([float(cy)]))[0]
        Xc = synth.se3_act(T1, X)
        px_in[i] = [cam.fx * Xc[0] / Xc[2] + cam.cx, cam.fy * Xc[1] / Xc[2] + cam.cy]
        px_in[i] += rng.uniform(-2, 2, 2)
    px_in[0] = [3.2, 40.0]                       # left border -> break before first iteration
    px_in[1] = [cam.width - 4.5, 60.0]           # right border
    px_in[2] = [80.3, cam.height - 3.9]          # bottom border
    pwb[3] = 128                                  # flat template: singular H -> inf/NaN path
    patch[3] = 128
    n_iter[4] = 1; n_iter[5] = 0; n_iter[6] = 3
    pwb[7] = rng.integers(0, 256, 100)            # noise template (diverges / leaves image)
    patch[7] = refpy.patch_from_border(pwb[7])
    ok = np.zeros(n, dtype=np.uint8)
    px_out = np.zeros((n, 2))
    for i in range(n):
        o, p = refpy.align2d(cur, pwb[i], patch[i], int(n_iter[i]), px_in[i])
        ok[i] = o; px_out[i] = p
    # align1D
    dirs = rng.normal(size=(n, 2)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    ok1 = np.zeros(n, dtype=np.uint8)
    px_out1 = np.zeros((n, 2))
    hinv = np.zeros(n)
    for i in range(n):
        o, p, h = refpy.align1d(cur, dirs[i], pwb[i], patch[i], int(n_iter[i]), px_in[i])
        ok1[i] = o; px_out1[i] = p; hinv[i] = h
    np.savez_compressed(os.path.join(OUT, "align.npz"), cur=cur, pwb=pwb, patch=patch, px_in=px_in,
                        n_iter=n_iter, ok=ok, px_out=px_out, dirs=dirs, ok1=ok1, px_out1=px_out1, hinv=hinv)


def gen_matcher(rng):
    cam, scene, T0, T1, ref, cur = small_scene()
    n = 256
    # ZMSSD
    zp = rng.integers(0, 256, (n, 64)).astype(np.uint8)
    zp[:64] = np.stack([cur[y:y + 8, x:x + 8].reshape(64) for x, y in
                        zip(rng.integers(0, 150, 64), rng.integers(0, 110, 64))])
    zxy = np.stack([rng.integers(0, cam.width - 8, n), rng.integers(0, cam.height - 8, n)], axis=1)
    zxy[0] = [0, 0]
    zp[1] = 255; zp[2] = 0
    zs = np.array([refpy.zmssd(zp[i], cur, int(zxy[i, 0]), int(zxy[i, 1])) for i in range(n)], dtype=np.int64)
    # warp matrix / search level / warpAffine / triangulation / cam2world
    px_ref = np.stack([rng.uniform(12, cam.width - 12, n), rng.uniform(12, cam.height - 12, n)], axis=1)
    level_ref = rng.integers(0, 3, n).astype(np.int32)
    f_ref = np.stack([refpy.cam2world(cam, u, v) for u, v in px_ref])
    depth = rng.uniform(0.8, 4.0, n)
    T_cur_ref = np.stack([synth.se3_from_twist(rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.15, 0.15, 3))
                          for _ in range(n)])
    T_cur_ref[:16, 2] += rng.uniform(0.5, 1.5, 16)   # forward motion: det(A) > 3 -> search level > 0
    T_cur_ref[16] = synth.se3_from_twist([0, 0, 0], [0, 0, 0])
    A = np.stack([refpy.get_warp_matrix_affine(cam, px_ref[i], f_ref[i], depth[i], T_cur_ref[i], int(level_ref[i]))
                  for i in range(n)])
    A[17] = 0.0                                      # zero matrix -> NaN inverse -> patch untouched (sentinel 7)
    # (a singular non-zero A gives an inf inverse, NaN sample coordinates and an out-of-bounds
    #  read in the reference: undefined behaviour, not a fixture)
    best = np.array([refpy.get_best_search_level(A[i], 2) for i in range(n)], dtype=np.int32)
    ref_pyr = synth.build_pyramid(ref, 3)
    patches = np.stack([refpy.warp_affine(A[i], ref_pyr[level_ref[i]], px_ref[i], int(level_ref[i]), int(best[i]), 5)
                        for i in range(n)])
    f_cur = np.stack([refpy.cam2world(cam, u, v) for u, v in
                      zip(rng.uniform(0, cam.width, n), rng.uniform(0, cam.height, n))])
    f_cur[20] = synth.quat_rot(T_cur_ref[20, 3:], f_ref[20])   # parallel rays: det < 1e-6 -> false
    tri = [refpy.depth_from_triangulation(T_cur_ref[i], f_ref[i], f_cur[i]) for i in range(n)]
    np.savez_compressed(
        os.path.join(OUT, "matcher.npz"), cur=cur, ref=ref, zp=zp, zxy=zxy, zs=zs,
        cam=np.array([cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy]), px_ref=px_ref,
        level_ref=level_ref, f_ref=f_ref, depth=depth, T_cur_ref=T_cur_ref, A=A, best=best,
        patches=patches, f_cur=f_cur, tri_ok=np.array([t[0] for t in tri], dtype=np.uint8),
        tri_depth=np.array([t[1] for t in tri]))


def gen_vision(rng):
    cam, scene, T0, T1, ref, cur = small_scene()
    n = 512
    uv = np.stack([rng.uniform(0, cam.width - 1.001, n), rng.uniform(0, cam.height - 1.001, n)], axis=1).astype(np.float32)
    uv[0] = [0, 0]; uv[1] = [10, 20]; uv[2] = [10.5, 20.5]
    val = np.array([refpy.interpolate_8u(cur, float(u), float(v)) for u, v in uv], dtype=np.float32)
    img = rng.integers(0, 256, (48, 64)).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "vision.npz"), cur=cur, uv=uv, val=val, img=img,
                        half_sse2=refpy.half_sample(img), half_scalar=refpy.half_sample(img, force_scalar=True),
                        half_odd=refpy.half_sample(img[:, :50].copy()))



# ---- fixtures from the reference's own SparseImgAlign / Matcher member functions (oracle/ref/ref_objects.cpp) ----
SIA_REF_CASES = [
    # name, make_frame_pair kwargs, max_level, min_level, n_iter
    ("c0_200", dict(seed=12345, n_features=200), 4, 0, 30),
    ("c1_2000", dict(seed=12346, n_features=2000), 4, 0, 30),
    ("c0_l2", dict(seed=12345, n_features=200), 4, 2, 30),
    ("nulls_320", dict(seed=777, width=320, height=240, n_features=120, null_point_every=7, t_mag=0.015, r_mag=0.006), 4, 0, 30),
    ("border_320", dict(seed=901, width=320, height=240, n_features=400, border=4, t_mag=0.02, r_mag=0.01), 4, 0, 30),
    ("bigmotion_320", dict(seed=902, width=320, height=240, n_features=300, border=6, t_mag=0.12, r_mag=0.04), 4, 0, 30),
    ("iters5", dict(seed=12347, n_features=600), 3, 1, 5),
    ("empty", dict(seed=12345, n_features=0), 4, 0, 30),
    # BASELINE config C3's frame size, the shipping level range (Config::kltMaxLevel / kltMinLevel: L4..L2) at C1's size,
    # a frame above the fused kernel's 2816 features (the streaming kernels)
    ("hd_1280", dict(seed=12350, width=1280, height=720, n_features=2000), 4, 0, 30),
    ("c1_l4_l2", dict(seed=12351, n_features=2000), 4, 2, 30),
    ("big_3500", dict(seed=12352, n_features=3500), 4, 0, 30),
    # a vk::PinholeCamera with radtan distortion: the reference's compiled world2cam (pinhole_camera.cpp:79-106) inside computeResiduals
    ("radtan_600", dict(seed=77, n_features=600, dist=(-0.05, 0.01, 1e-3, -5e-4, 2e-3)), 4, 0, 30),
]


def make_sia_case(kw):
    kw = dict(kw)
    dist = kw.pop("dist", None)
    if dist is not None:
        fp = synth.make_frame_pair(**kw)
        fp.dist = list(dist)
        return fp
    if kw.get("n_features", 1) == 0:
        kw["n_features"] = 8
        fp = synth.make_frame_pair(**kw)
        fp.px, fp.f, fp.pos, fp.has_point = fp.px[:0], fp.f[:0], fp.pos[:0], fp.has_point[:0]
        return fp
    return synth.make_frame_pair(**kw)


def gen_sia_ref():
    out = {}
    for name, kw, max_level, min_level, n_iter in SIA_REF_CASES:
        fp = make_sia_case(kw)
        r = refpy.sparse_img_align_run(fp, max_level=max_level, min_level=min_level, n_iter=n_iter)
        n = len(fp.px)
        out[name + "_crc"] = np.array([crc(fp.ref_pyr[0]), crc(fp.cur_pyr[0]), crc(fp.px), crc(fp.pos)], dtype=np.uint64)
        out[name + "_T"] = r["T_cur_w"]
        out[name + "_n_tracked"] = np.array(r["n_tracked"])
        out[name + "_H"] = r["H"]
        out[name + "_chi2"] = np.array(r["chi2"])
        out[name + "_stop"] = np.array(r["stop"])
        out[name + "_iter"] = r["iter"]
        out[name + "_n_meas"] = r["n_meas"]
        out[name + "_visible"] = r["visible"]
        k = min(n, 64)                      # the caches of the first 64 patches in full, checksums of everything
        out[name + "_cache64"] = r["ref_patch_cache"][:k]
        out[name + "_jac64"] = r["jacobian_cache"][:k * 16]
        out[name + "_cache_crc"] = np.array([crc(r["ref_patch_cache"]), crc(r["jacobian_cache"])], dtype=np.uint64)
        # fixed work: exactly n_iter evaluations per level through the reference's compiled computeResiduals / solve / update
        fw = refpy.sparse_img_align_fixed_work(fp, max_level=max_level, min_level=min_level, n_iter=n_iter)
        out[name + "_fw_T"] = fw["T_cur_w"]
        out[name + "_fw_H"] = fw["H"]
        out[name + "_fw_chi2"] = np.array(fw["chi2"])
        out[name + "_fw_n_tracked"] = np.array(fw["n_tracked"])
        print("sia_ref", name, "n", n, "tracked", r["n_tracked"], "iter", r["iter"][:5], "stop", r["stop"])
    np.savez_compressed(os.path.join(OUT, "sia_ref.npz"), **out)


# the other branches of NLLSSolver on the same frame pairs: (method, scale estimator, weight function), the reference's enums
SIA_NLLS_COMBOS = [(1, 0, 0), (0, 1, 1), (0, 2, 2), (0, 2, 3), (0, 3, 1), (1, 1, 1), (1, 2, 3), (1, 3, 2), (1, 1, 2)]
SIA_NLLS_CASES = ["c0_200", "c1_2000", "nulls_320", "border_320", "bigmotion_320", "iters5"]


def nlls_key(name, combo):
    return "%s_m%d_s%d_w%d" % ((name,) + tuple(combo))


def gen_sia_nlls_ref():
    """SparseImgAlign::run of the reference's compiled code with method_ = LevenbergMarquardt and / or a robust cost set
    through its own setRobustCostFunction (oracle/ref/ref_objects.cpp: ref_sparse_img_align_run_ex)."""
    out = {}
    cases = {c[0]: c for c in SIA_REF_CASES}
    for name in SIA_NLLS_CASES:
        _, kw, max_level, min_level, n_iter = cases[name]
        fp = make_sia_case(kw)
        for combo in SIA_NLLS_COMBOS:
            r = refpy.sparse_img_align_run(fp, max_level=max_level, min_level=min_level, n_iter=n_iter, method=combo[0],
                                           scale_estimator=combo[1], weight_function=combo[2])
            k = nlls_key(name, combo)
            out[k + "_T"] = r["T_cur_w"]
            out[k + "_n_tracked"] = np.array(r["n_tracked"])
            out[k + "_H"] = r["H"]
            out[k + "_chi2"] = np.array(r["chi2"])
            out[k + "_stop"] = np.array(r["stop"])
            out[k + "_iter"] = r["iter"]
            out[k + "_n_meas"] = r["n_meas"]
            out[k + "_scale_mu_nu"] = np.array([np.float64(r["scale"]), r["mu"], r["nu"]])
            print("sia_nlls_ref", k, "tracked", r["n_tracked"], "iter", r["iter"][:5], "stop", r["stop"], "scale", r["scale"])
    np.savez_compressed(os.path.join(OUT, "sia_nlls_ref.npz"), **out)


def epi_case_inputs():
    """600 seeds on a 320x240 keyframe with four kinds of depth interval (see tests)."""
    from android_svo_amd import seedsynth
    sc = seedsynth.make_seed_case(n_seeds=600, seed=7, width=320, height=240, border=24)
    rng = np.random.default_rng(1)
    d_est, d_min, d_max = np.zeros(600), np.zeros(600), np.zeros(600)
    for i in range(600):
        d = sc.true_depth[i]
        mode = i % 4
        if mode == 0:
            lo, hi = d * 0.98, d * 1.02                 # short epipolar segment: direct align2D
        elif mode == 1:
            lo, hi = d * 0.5, d * 3.0                   # ZMSSD search
        elif mode == 2:
            lo, hi = d * 0.15, d * 50.0                 # long segment, part of it outside the image
        else:
            lo, hi = d * rng.uniform(0.3, 0.9), d * rng.uniform(1.1, 4)
        d_est[i], d_min[i], d_max[i] = d * rng.uniform(0.9, 1.1), lo, hi
    return sc, d_est, d_min, d_max


def gen_epi_ref():
    sc, d_est, d_min, d_max = epi_case_inputs()
    n = len(d_est)
    ok = np.zeros(n, dtype=np.uint8)
    depth, px_cur, lvl, epi = np.zeros(n), np.zeros((n, 2)), np.zeros(n, dtype=np.int32), np.zeros(n)
    pwb = np.zeros((n, 100), dtype=np.uint8)
    for i in range(n):
        r = refpy.find_epipolar_match_direct(sc.cam, sc.ref_pyr, sc.cur_pyr, sc.T_ref_w, sc.T_cur_w, sc.px[i], sc.f[i],
                                             int(sc.level[i]), d_est[i], d_min[i], d_max[i])
        ok[i], depth[i], px_cur[i], lvl[i], epi[i], pwb[i] = r["ok"], r["depth"], r["px_cur"], r["search_level"], r["epi_length"], r["patch_with_border"]
    T_cur_ref = refpy.se3_mul(sc.T_cur_w, refpy.se3_inverse(sc.T_ref_w))
    np.savez_compressed(os.path.join(OUT, "epi_ref.npz"), crc=np.array([crc(sc.ref_pyr[0]), crc(sc.cur_pyr[0]), crc(sc.px)], dtype=np.uint64),
                        d_est=d_est, d_min=d_min, d_max=d_max, T_cur_ref=T_cur_ref, ok=ok, depth=depth, px_cur=px_cur,
                        search_level=lvl, epi_length=epi, pwb=pwb)
    print("epi_ref ok", int(ok.sum()), "of", n)


def match_direct_inputs():
    fp = synth.make_frame_pair(seed=4711, width=320, height=240, n_features=300, border=12)
    rng = np.random.default_rng(3)
    n = len(fp.px)
    px_in, lvl, edge, grad = np.zeros((n, 2)), np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.uint8), np.zeros((n, 2))
    for i in range(n):
        Xc = synth.se3_act(fp.T_cur_w_true, fp.pos[i])
        px_in[i] = np.array([fp.cam.fx * Xc[0] / Xc[2] + fp.cam.cx, fp.cam.fy * Xc[1] / Xc[2] + fp.cam.cy]) + rng.uniform(-2.5, 2.5, 2)
        lvl[i] = i % 3
        edge[i] = 1 if i % 5 == 0 else 0
        g = rng.normal(size=2)
        grad[i] = g / np.linalg.norm(g)
    return fp, px_in, lvl, edge, grad


def gen_match_direct_ref():
    fp, px_in, lvl, edge, grad = match_direct_inputs()
    n = len(px_in)
    ok, px_out, sl = np.zeros(n, dtype=np.uint8), np.zeros((n, 2)), np.zeros(n, dtype=np.int32)
    for i in range(n):
        r = refpy.find_match_direct(fp.cam, fp.ref_pyr, fp.cur_pyr, fp.T_ref_w, fp.T_cur_w_true, fp.px[i], fp.f[i],
                                    int(lvl[i]), fp.pos[i], px_in[i], edgelet=bool(edge[i]), grad=grad[i])
        ok[i], px_out[i], sl[i] = r["ok"], r["px_cur"], r["search_level"]
    np.savez_compressed(os.path.join(OUT, "match_direct_ref.npz"),
                        crc=np.array([crc(fp.ref_pyr[0]), crc(fp.cur_pyr[0]), crc(fp.px)], dtype=np.uint64),
                        px_in=px_in, level=lvl, edgelet=edge, grad=grad, ok=ok, px_out=px_out, search_level=sl)
    print("match_direct_ref ok", int(ok.sum()), "of", n)


def gen_refine_ref():
    """Pieces of the two small refinements that the reference build holds: Point::optimize end to end,
    Tukey weight, MAD scale, vk::getMedian, Eigen 6x6 inverse and 3x3 LDLT."""
    rng = np.random.default_rng(99)
    pos0, off, Ts, fs, _, iters = synth.make_point_opt_cases()
    out = np.stack([refpy.point_optimize(pos0[i], Ts[off[i]:off[i + 1]], fs[off[i]:off[i + 1]], n_iter=int(iters[i]))
                    for i in range(len(pos0))])
    xs = np.concatenate([np.linspace(0, 12, 400), rng.uniform(0, 9, 200)]).astype(np.float32)
    tuk = np.array([refpy.tukey_weight(float(x)) for x in xs], dtype=np.float32)
    errs = [rng.uniform(0, 3, m).astype(np.float32) for m in (1, 2, 7, 1000, 1001)]
    mad = np.array([refpy.mad_scale(e) for e in errs], dtype=np.float32)
    dd = [rng.uniform(0, 9, m) for m in (1, 4, 999, 1000)]
    med = np.array([refpy.median_d(d) for d in dd])
    As = []
    for i in range(32):
        M = rng.normal(size=(30, 6)) * rng.uniform(0.1, 50, 6)
        As.append(M.T @ M)
    As = np.stack(As)
    inv = np.stack([refpy.inverse6(A) for A in As])
    A3, b3 = [], []
    for i in range(64):
        M = rng.normal(size=(rng.integers(2, 12), 3))
        A3.append(M.T @ M); b3.append(rng.normal(size=3))
    A3[5][:] = 0
    A3, b3 = np.stack(A3), np.stack(b3)
    x3 = np.stack([refpy.ldlt3_solve(a, b) for a, b in zip(A3, b3)])
    save = dict(point_crc=np.array([crc(pos0), crc(Ts), crc(fs)], dtype=np.uint64), point_out=out, tukey_x=xs, tukey=tuk, mad=mad,
                med=med, A6=As, inv6=inv, A3=A3, b3=b3, x3=x3)
    for k, e in enumerate(errs):
        save["err%d" % k] = e
    for k, d in enumerate(dd):
        save["dd%d" % k] = d
    np.savez_compressed(os.path.join(OUT, "refine_ref.npz"), **save)
    print("refine_ref: points", len(out))


def gen_shitomasi_ref():
    """vk::shiTomasiScore (vision.cpp) on a rendered image and on noise, borders included."""
    rng = np.random.default_rng(123)
    cam, scene, T0, T1, ref, cur = small_scene()
    noise = rng.integers(0, 256, (96, 128)).astype(np.uint8)
    out = {}
    for name, img in (("scene", ref), ("noise", noise)):
        h, w = img.shape
        uv = np.stack([rng.integers(0, w, 1500), rng.integers(0, h, 1500)], axis=1).astype(np.int32)
        uv[:8] = [[0, 0], [4, 4], [5, 5], [w - 6, h - 6], [w - 5, h - 5], [w - 1, h - 1], [5, h - 6], [w - 6, 5]]
        out[name + "_img"] = img
        out[name + "_uv"] = uv
        out[name + "_score"] = np.array([refpy.shi_tomasi_score(img, int(u), int(v)) for u, v in uv], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "shitomasi_ref.npz"), **out)
    print("shitomasi_ref written")


def run_reproject_reference(cs, max_fts):
    off_raw, ids_raw = synth.flatten_cells(cs, cs["raw"])
    n = len(ids_raw)
    g = lambda a: a[ids_raw]
    return refpy.reproject_cells(cs["cam"], cs["kf_pyr"], cs["T_kf_w"], cs["cur_pyr"], cs["T_cur_w"], off_raw, g(cs["slot"]),
                                 g(cs["px_ref"]), g(cs["f_ref"]), g(cs["level"]), g(cs["pos"]), np.zeros(n, np.uint8),
                                 np.tile([1.0, 0.0], (n, 1)), g(cs["ptype"]), g(cs["n_failed"]), g(cs["n_succeeded"]),
                                 g(cs["px_cur"]), max_fts=max_fts), ids_raw


def gen_reproject_ref():
    """The cell loop of Reprojector::reprojectMap executed by the reference's own compiled Reprojector::reprojectCell
    (with Matcher::findMatchDirect, the point bookkeeping and Frame::addFeature) on two settings of Config::maxFts()."""
    cs = synth.make_reproject_case()
    out = dict(crc=np.array([crc(cs["cur_pyr"][0]), crc(cs["px_cur"]), crc(cs["pos"])], dtype=np.uint64))
    for tag, max_fts in (("full", 1200), ("cap", 40)):
        r, ids_raw = run_reproject_reference(cs, max_fts)
        out[tag + "_n"] = np.array([r["n_matches"], r["n_trials"]], dtype=np.int64)
        out[tag + "_feat_point"] = ids_raw[r["feat_cand"]]                 # map point of every new feature, creation order
        out[tag + "_feat_px"] = r["feat_px"]
        out[tag + "_feat_level"] = r["feat_level"]
        pt_nf, pt_ns, pt_ty, left = (np.full(len(cs["ptype"]), -1, np.int32) for _ in range(4))
        pt_nf[ids_raw], pt_ns[ids_raw], pt_ty[ids_raw], left[ids_raw] = r["n_failed"], r["n_succeeded"], r["type"], r["left_in_cell"]
        out[tag + "_n_failed"], out[tag + "_n_succeeded"], out[tag + "_type"], out[tag + "_left"] = pt_nf, pt_ns, pt_ty, left
        print("reproject_ref", tag, r["n_matches"], r["n_trials"])
    np.savez_compressed(os.path.join(OUT, "reproject_ref.npz"), **out)


MAP_REF_CASES = (("near", dict(seed=31), 1200), ("cap", dict(seed=31), 40), ("wide", dict(seed=32, n_kf=9, n_points=900, n_candidates=60, cell_size=25, kf_step=0.55), 1200),
                 ("rekey", dict(seed=31), 1200))


def rekey_override(cs, unlinked0):
    """Key features for the "rekey" case: the reference's own choice except for the centre slot of every keyframe and the
    lower-right slot of every third -- in the odd keyframes the centre slot holds a point the frame is going to delete
    (unlinked0: what it deletes with the reference's own key points), so that Map::safeDeletePoint -> Frame::removeKeyPoint
    makes the keyframe choose again; elsewhere some other live feature's point, an incumbent a fresh selection would not pick
    and that must survive."""
    rng = np.random.default_rng(5)
    cu, cv = cs["cam"].width // 2, cs["cam"].height // 2
    ko = np.full((cs["n_kf"], 5), -2, np.int32)
    dead0 = unlinked0.astype(bool)
    for k in range(cs["n_kf"]):
        o = cs["kf_ftr_obs"][cs["kf_ftr_offset"][k]:cs["kf_ftr_offset"][k + 1]]
        pts = cs["obs_point"][o]
        pool = [int(p) for p in pts if (dead0[p] if k % 2 else (not dead0[p] and cs["pt_type"][p] != synth.TYPE_DELETED))]
        if pool:
            ko[k, 0] = pool[int(rng.integers(len(pool)))]
        if k % 3 == 0:
            quad = [int(cs["obs_point"][oo]) for oo in o if cs["obs_px"][oo][0] >= cu and cs["obs_px"][oo][1] >= cv and not dead0[cs["obs_point"][oo]]
                    and cs["pt_type"][cs["obs_point"][oo]] != synth.TYPE_DELETED]
            if quad:
                ko[k, 1] = quad[int(rng.integers(len(quad)))]
    return ko


def gen_reproject_map_ref():
    """The WHOLE Reprojector::reprojectMap executed by the reference's own compiled code on a real svo::Map with keyframes,
    multi-observation points and point candidates (oracle/ref/ref_objects.cpp: ref_reproject_map): close-keyframe
    selection and ordering, the projection into grid cells, the candidate loop, the cell loop with its bookkeeping.
    Stored: the key points the call started from (Frame::setKeyPoints' choice; in the "rekey" case with some incumbents put
    in by hand, rekey_override), every output, and the key points Map::safeDeletePoint -> Frame::removeKeyPoint left."""
    out = {}
    for tag, kw, max_fts in MAP_REF_CASES:
        cs = synth.make_map_case(**kw)
        ko = None
        if tag == "rekey":
            ko = rekey_override(cs, refpy.reproject_map(cs, max_fts=max_fts)["unlinked"])
        r = refpy.reproject_map(cs, max_fts=max_fts, key_override=ko)
        out[tag + "_crc"] = np.array([crc(cs["cur_pyr"][0]), crc(cs["obs_px"]), crc(cs["pt_pos"]), crc(cs["kf_ftr_obs"])], dtype=np.uint64)
        out[tag + "_n"] = np.array([r["n_matches"], r["n_trials"]], dtype=np.int64)
        for k in ("kf_key_point", "kf_key_point_after", "type", "n_failed", "n_succeeded", "unlinked", "overlap_kf", "overlap_count", "feat_point", "feat_px",
                  "feat_level", "feat_type", "feat_grad"):
            out[tag + "_" + k] = r[k]
        lost = np.array([bool(r["unlinked"][kk[kk >= 0]].any()) for kk in r["kf_key_point"]])
        if tag == "rekey":
            assert lost.sum() >= 2 and (~lost).sum() >= 2 and (r["kf_key_point_after"][lost] != r["kf_key_point"][lost]).any()
            assert (r["kf_key_point_after"][~lost] == r["kf_key_point"][~lost]).all()
        print("reproject_map_ref", tag, "keyframes that lost a key feature:", int(lost.sum()), "key slots changed:", int((r["kf_key_point_after"] != r["kf_key_point"]).sum()))
        n_del_loop = int(((r["unlinked"] == 1) & (cs["pt_type"] != synth.TYPE_CANDIDATE)).sum())
        print("reproject_map_ref", tag, r["n_matches"], r["n_trials"], "overlap", list(r["overlap_kf"]), "deleted in the cell loop", n_del_loop,
              "candidates deleted", int(((r["unlinked"] == 1) & (cs["pt_type"] == synth.TYPE_CANDIDATE)).sum()), "edgelet features", int((r["feat_type"] == 1).sum()))
    np.savez_compressed(os.path.join(OUT, "reproject_map_ref.npz"), **out)


CAMERA_REF_CASES = (
    # name, width, height, fx, fy, cx, cy, (k1, k2, p1, p2, k3)
    ("pinhole_vga", 640, 480, 500.0, 500.0, 319.5, 239.5, (0.0, 0.0, 0.0, 0.0, 0.0)),
    ("radtan_strong", 640, 480, 458.654, 457.296, 367.215, 248.375, (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)),
    ("radtan_k3", 752, 480, 420.0, 418.5, 371.3, 236.9, (-0.12, 0.03, 2e-4, -1e-4, 0.004)),
    ("radtan_hd", 1280, 720, 1000.0, 1002.0, 639.5, 359.5, (0.05, -0.02, -3e-4, 5e-4, 0.001)),
    ("tiny_k1", 640, 480, 500.0, 500.0, 319.5, 239.5, (5e-8, 0.3, 0.1, 0.1, 0.1)),       # |d0| <= 1e-7: distortion_ stays false
)


def camera_ref_inputs(case):
    """Deterministic inputs of one camera case: camera-frame points (some behind the camera, some far off axis), unit-plane
    points, pixels, integer observations around the image border."""
    name, w, h, fx, fy, cx, cy, dist = case
    rng = np.random.default_rng(sum(ord(ch) for ch in name))     # a fixed seed per case name
    cam = synth.Camera(w, h, fx, fy, cx, cy)
    cam.dist = dist
    n = 400
    xyz = np.stack([rng.uniform(-2.5, 2.5, n), rng.uniform(-2.0, 2.0, n), rng.uniform(0.3, 6.0, n)], axis=1)
    xyz[::37, 2] *= -1.0                       # behind the camera: still plain arithmetic
    uv = np.stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-0.9, 0.9, n)], axis=1)
    px = np.stack([rng.uniform(-20, w + 20, n), rng.uniform(-20, h + 20, n)], axis=1)
    obs = np.stack([rng.integers(-12, w + 12, n), rng.integers(-12, h + 12, n)], axis=1).astype(np.int32)
    obs[:40, 0] = rng.choice([0, 1, 7, 8, 9, w - 10, w - 9, w - 8, w - 1, w], 40)
    obs[40:80, 1] = rng.choice([0, 1, 7, 8, 9, h - 10, h - 9, h - 8, h - 1, h], 40)
    return cam, xyz, uv, px, obs


def gen_camera_ref():
    """vk::PinholeCamera::world2cam (both overloads, pinhole and radtan), the distortion-free cam2world and
    vk::AbstractCamera::isInFrame, executed by the reference's own compiled members (ref_camera.cpp)."""
    out = {}
    for case in CAMERA_REF_CASES:
        name = case[0]
        cam, xyz, uv, px, obs = camera_ref_inputs(case)
        out[name + "_crc"] = np.array([crc(xyz), crc(uv), crc(px), crc(obs)], dtype=np.uint64)
        out[name + "_px_of_xyz"] = refpy.pinhole_world2cam(cam, xyz)
        out[name + "_px_of_uv"] = refpy.pinhole_world2cam_uv(cam, uv)
        if abs(cam.dist[0]) <= 1e-7:
            out[name + "_f_of_px"] = refpy.pinhole_cam2world(cam, px)
        for boundary, level in ((0, 0), (8, 0), (8, 1), (6, 2), (9, 3)):
            plain, lev = refpy.camera_is_in_frame(cam.width, cam.height, obs, boundary, level)
            out["%s_in_b%d" % (name, boundary)] = plain
            out["%s_in_b%d_l%d" % (name, boundary, level)] = lev
    np.savez_compressed(os.path.join(OUT, "camera_ref.npz"), **out)
    print("camera_ref", len(out), "arrays")


def main():
    if "--camera-only" in sys.argv:
        assert refpy.available(), "build oracle/_ref first: make -C oracle ref"
        gen_camera_ref()
        return
    if "--nlls-only" in sys.argv:
        assert refpy.available(), "build oracle/_ref first: make -C oracle ref"
        gen_sia_nlls_ref()
        return
    if "--map-only" in sys.argv:
        assert refpy.available(), "build oracle/_ref first: make -C oracle ref"
        gen_reproject_map_ref()
        return
    assert refpy.available(), "build oracle/_ref first: make -C oracle ref"
    os.makedirs(OUT, exist_ok=True)
    if not any(a in sys.argv for a in ("--objects-only", "--refine-only", "--shitomasi-only", "--reproject-only")):
        rng = np.random.default_rng(20240607)
        gen_se3(rng); gen_algebra(rng); gen_gn(rng); gen_align(rng); gen_matcher(rng); gen_vision(rng)
    if not any(a in sys.argv for a in ("--refine-only", "--shitomasi-only", "--reproject-only")):
        gen_sia_ref(); gen_sia_nlls_ref(); gen_epi_ref(); gen_match_direct_ref()
    if "--shitomasi-only" not in sys.argv and "--reproject-only" not in sys.argv:
        gen_refine_ref()
    if "--reproject-only" not in sys.argv:
        gen_shitomasi_ref()
    gen_reproject_ref()
    gen_reproject_map_ref()
    gen_camera_ref()
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
