"""ctypes binding of the CPU oracle (oracle/libsvo_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsvo_oracle.so")
MAX_LEVELS = 8


class Camera(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("d", C.c_double * 5), ("distortion", C.c_int)]


class SiaParams(C.Structure):
    _fields_ = [("max_level", C.c_int), ("min_level", C.c_int), ("n_iter", C.c_int),
                ("eps", C.c_double), ("early_stop", C.c_int)]


class SiaResult(C.Structure):
    _fields_ = [("T_cur_w", C.c_double * 7), ("n_tracked", C.c_size_t), ("H", C.c_double * 36),
                ("Jres", C.c_double * 6), ("chi2", C.c_double), ("stop", C.c_int),
                ("iters", C.c_int * MAX_LEVELS), ("n_precompute_patches", C.c_long),
                ("n_residual_patches", C.c_long)]


class Seed(C.Structure):
    _fields_ = [("a", C.c_float), ("b", C.c_float), ("mu", C.c_float), ("z_range", C.c_float),
                ("sigma2", C.c_float)]


class EpiResult(C.Structure):
    _fields_ = [("ok", C.c_int), ("depth", C.c_double), ("px_cur", C.c_double * 2),
                ("search_level", C.c_int), ("epi_length", C.c_double), ("n_zmssd", C.c_int),
                ("n_align_iters", C.c_int), ("path", C.c_int), ("patch_with_border", C.c_uint8 * 100)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "svo_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libsvo_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        override = os.environ.get("SVO_ORACLE_LIB")          # e.g. the sanitizer build of `make -C oracle asan-check`
        if not override:
            build()
        _lib = C.CDLL(override or _LIB_PATH)
        _lib.svo_orc_interpolate_8u.restype = C.c_float
        _lib.svo_orc_compute_tau.restype = C.c_double
        _lib.svo_orc_sia_eval.restype = C.c_double
        _lib.svo_orc_sia_open.restype = C.c_void_p
    return _lib


def camera(cam, dist: Sequence[float] | None = None) -> Camera:
    if dist is None:
        dist = getattr(cam, "dist", None)        # a camera object may carry its radtan coefficients
    c = Camera()
    c.width, c.height = int(cam.width), int(cam.height)
    c.fx, c.fy, c.cx, c.cy = cam.fx, cam.fy, cam.cx, cam.cy
    d = list(dist) if dist is not None else [0.0] * 5
    for i in range(5):
        c.d[i] = d[i]
    c.distortion = 1 if abs(d[0]) > 1e-7 else 0   # pinhole_camera.cpp:27
    return c


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def pyr_ptrs(pyr):
    arr = (C.POINTER(C.c_uint8) * MAX_LEVELS)()
    for i, im in enumerate(pyr):
        assert im.flags["C_CONTIGUOUS"] and im.dtype == np.uint8
        arr[i] = _p(im, C.c_uint8)
    return arr


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def sparse_img_align(fp, max_level=4, min_level=0, n_iter=30, eps=1e-6, early_stop=True,
                     T_cur_w_init=None, method=0, scale_estimator=0, weight_function=0) -> SiaResult:
    """method 0 GaussNewton / 1 LevenbergMarquardt; scale_estimator 0 Unit (no weights) / 1 TDist / 2 MAD / 3 Normal;
    weight_function 0 Unit / 1 TDist / 2 Tukey / 3 Huber.  The result carries `.scale` (scale_ after the run)."""
    L = lib()
    cam = camera(fp.cam, getattr(fp, "dist", None))
    prm = SiaParams(max_level, min_level, n_iter, eps, 1 if early_stop else 0)
    out = SiaResult()
    rp, cp = pyr_ptrs(fp.ref_pyr), pyr_ptrs(fp.cur_pyr)
    px, f, pos = f64(fp.px), f64(fp.f), f64(fp.pos)
    hp = np.ascontiguousarray(fp.has_point, dtype=np.uint8)
    T_ref = f64(fp.T_ref_w)
    T_init = f64(fp.T_cur_w_init if T_cur_w_init is None else T_cur_w_init)
    scale = C.c_float(0)
    L.svo_orc_sparse_img_align_ex(C.byref(cam), rp, cp, C.c_int(len(px)), _p(px, C.c_double),
                                  _p(f, C.c_double), _p(pos, C.c_double), _p(hp, C.c_uint8),
                                  _p(T_ref, C.c_double), _p(T_init, C.c_double), C.byref(prm), C.c_int(method),
                                  C.c_int(scale_estimator), C.c_int(weight_function), C.byref(out), C.byref(scale))
    out.scale = np.float32(scale.value)
    return out


def sia_single_eval(fp, level, T_cur_from_ref, want_caches=False):
    L = lib()
    cam = camera(fp.cam)
    n = len(fp.px)
    px, f, pos = f64(fp.px), f64(fp.f), f64(fp.pos)
    hp = np.ascontiguousarray(fp.has_point, dtype=np.uint8)
    out28 = np.zeros(28)
    nm = C.c_long(0)
    cache = np.zeros((n, 16), dtype=np.float32)
    jac = np.zeros((n, 16, 6), dtype=np.float64)
    vis = np.zeros(n, dtype=np.uint8)
    T_ref = f64(fp.T_ref_w)
    T = f64(T_cur_from_ref)
    L.svo_orc_sia_single_eval(C.byref(cam), _p(fp.ref_pyr[level], C.c_uint8), _p(fp.cur_pyr[level], C.c_uint8),
                              C.c_int(level), C.c_int(n), _p(px, C.c_double), _p(f, C.c_double),
                              _p(pos, C.c_double), _p(hp, C.c_uint8), _p(T_ref, C.c_double),
                              _p(T, C.c_double), _p(out28, C.c_double), C.byref(nm),
                              _p(cache, C.c_float), _p(jac, C.c_double), _p(vis, C.c_uint8))
    if want_caches:
        return out28, nm.value, cache, jac, vis
    return out28, nm.value


def align2d(img, pwb, patch, n_iter, px):
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    pwb = np.ascontiguousarray(pwb, dtype=np.uint8)
    patch = np.ascontiguousarray(patch, dtype=np.uint8)
    p = f64(px).copy()
    it = C.c_int(0)
    ok = L.svo_orc_align2d(_p(img, C.c_uint8), img.shape[1], img.shape[0], img.shape[1],
                           _p(pwb, C.c_uint8), _p(patch, C.c_uint8), n_iter, _p(p, C.c_double), C.byref(it))
    return bool(ok), p, it.value


def align1d(img, direction, pwb, patch, n_iter, px):
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    pwb = np.ascontiguousarray(pwb, dtype=np.uint8)
    patch = np.ascontiguousarray(patch, dtype=np.uint8)
    d = np.ascontiguousarray(direction, dtype=np.float32)
    p = f64(px).copy()
    it = C.c_int(0)
    hinv = C.c_double(0)
    ok = L.svo_orc_align1d(_p(img, C.c_uint8), img.shape[1], img.shape[0], img.shape[1], _p(d, C.c_float),
                           _p(pwb, C.c_uint8), _p(patch, C.c_uint8), n_iter, _p(p, C.c_double),
                           C.byref(hinv), C.byref(it))
    return bool(ok), p, hinv.value, it.value


def update_seed(x, tau2, seed5):
    s = Seed(*[float(v) for v in seed5])
    lib().svo_orc_update_seed(C.c_float(x), C.c_float(tau2), C.byref(s))
    return np.array([s.a, s.b, s.mu, s.z_range, s.sigma2], dtype=np.float32)


def seed_init(depth_mean, depth_min):
    s = Seed()
    lib().svo_orc_seed_init(C.byref(s), C.c_float(depth_mean), C.c_float(depth_min))
    return np.array([s.a, s.b, s.mu, s.z_range, s.sigma2], dtype=np.float32)


def compute_tau(T_ref_cur, f, z, px_error_angle):
    T = f64(T_ref_cur)
    ff = f64(f)
    return lib().svo_orc_compute_tau(_p(T, C.c_double), _p(ff, C.c_double), C.c_double(z),
                                     C.c_double(px_error_angle))


def find_epipolar_match(cam, ref_pyr, cur_pyr, T_cur_ref, px_ref, f_ref, level_ref, d_est, d_min, d_max,
                        n_pyr_levels=3, align_max_iter=10, max_steps=1000) -> EpiResult:
    c = camera(cam)
    out = EpiResult()
    T = f64(T_cur_ref)
    p = f64(px_ref)
    ff = f64(f_ref)
    lib().svo_orc_find_epipolar_match_direct(
        C.byref(c), pyr_ptrs(ref_pyr), pyr_ptrs(cur_pyr), _p(T, C.c_double), _p(p, C.c_double),
        _p(ff, C.c_double), C.c_int(level_ref), C.c_double(d_est), C.c_double(d_min), C.c_double(d_max),
        C.c_int(n_pyr_levels), C.c_int(align_max_iter), C.c_int(max_steps), C.byref(out))
    return out


def update_seeds(cam, ref_pyr, cur_pyr, T_ref_w, T_cur_w, px, f, level, a, b, mu, z_range, sigma2,
                 n_pyr_levels=3, align_max_iter=10, max_steps=1000, conv_thresh=100.0):
    """In-place update of (a, b, mu, sigma2); returns dict of per-seed outputs."""
    c = camera(cam)
    n = len(px)
    px, f = f64(px), f64(f)
    level = np.ascontiguousarray(level, dtype=np.int32)
    for arr in (a, b, mu, z_range, sigma2):
        assert arr.dtype == np.float32 and arr.flags["C_CONTIGUOUS"]
    status = np.zeros(n, dtype=np.int32)
    z = np.zeros(n)
    xyz = np.zeros((n, 3))
    nz = np.zeros(n, dtype=np.int32)
    na = np.zeros(n, dtype=np.int32)
    Tr, Tc = f64(T_ref_w), f64(T_cur_w)
    px_cur = np.zeros((n, 2))
    sl = np.zeros(n, dtype=np.int32)
    lib().svo_orc_update_seeds_ex(
        C.byref(c), pyr_ptrs(ref_pyr), pyr_ptrs(cur_pyr), _p(Tr, C.c_double), _p(Tc, C.c_double),
        C.c_int(n), _p(px, C.c_double), _p(f, C.c_double), _p(level, C.c_int), _p(a, C.c_float),
        _p(b, C.c_float), _p(mu, C.c_float), _p(z_range, C.c_float), _p(sigma2, C.c_float),
        C.c_int(n_pyr_levels), C.c_int(align_max_iter), C.c_int(max_steps), C.c_double(conv_thresh),
        _p(status, C.c_int), _p(z, C.c_double), _p(xyz, C.c_double), _p(nz, C.c_int), _p(na, C.c_int),
        _p(px_cur, C.c_double), _p(sl, C.c_int))
    return {"status": status, "z": z, "xyz_world": xyz, "n_zmssd": nz, "n_align_iters": na, "px_cur": px_cur,
            "search_level": sl}


def find_match_direct(cam, ref_pyr, cur_pyr, T_ref_w, T_cur_w, px_ref, f_ref, level_ref, pt_pos, px_cur, edgelet=False,
                      grad=(1.0, 0.0), n_pyr_levels=3, align_max_iter=10):
    c = camera(cam)
    Tr, Tc = f64(T_ref_w), f64(T_cur_w)
    pr, fr, pp, g = f64(px_ref), f64(f_ref), f64(pt_pos), f64(grad)
    pc = f64(px_cur).copy()
    sl = C.c_int(0)
    ok = lib().svo_orc_find_match_direct(C.byref(c), pyr_ptrs(ref_pyr), pyr_ptrs(cur_pyr), _p(Tr, C.c_double),
                                         _p(Tc, C.c_double), _p(pr, C.c_double), _p(fr, C.c_double), C.c_int(level_ref),
                                         _p(pp, C.c_double), C.c_int(1 if edgelet else 0), _p(g, C.c_double),
                                         C.c_int(n_pyr_levels), C.c_int(align_max_iter), _p(pc, C.c_double), C.byref(sl))
    return bool(ok), pc, sl.value


# ---- next rows f-4: pose_optimizer::optimizeGaussNewton, Point::optimize ----
class PoseOptResult(C.Structure):
    _fields_ = [("ran", C.c_int), ("T_f_w", C.c_double * 7), ("estimated_scale", C.c_double),
                ("error_init", C.c_double), ("error_final", C.c_double), ("num_obs", C.c_size_t),
                ("Cov", C.c_double * 36), ("n_iter_done", C.c_int), ("n_deleted", C.c_int)]


def pose_optimize(em, T_f_w, f, pos, level, has_point, reproj_thresh=2.0, n_iter=10):
    """Returns (PoseOptResult, has_point after the outlier test)."""
    T, ff, pp = f64(T_f_w), f64(f), f64(pos)
    lv = np.ascontiguousarray(level, dtype=np.int32)
    hp = np.ascontiguousarray(has_point, dtype=np.uint8).copy()
    out = PoseOptResult()
    lib().svo_orc_pose_optimize(C.c_double(em), C.c_double(reproj_thresh), C.c_int(n_iter), _p(T, C.c_double),
                                C.c_int(len(lv)), _p(ff, C.c_double), _p(pp, C.c_double), _p(lv, C.c_int),
                                _p(hp, C.c_uint8), C.byref(out))
    return out, hp


def point_optimize(pos, obs_T, obs_f, n_iter=5):
    p = f64(pos).copy()
    T, ff = f64(obs_T), f64(obs_f)
    it = C.c_int(0)
    lib().svo_orc_point_optimize(C.c_int(n_iter), _p(p, C.c_double), C.c_int(len(T)), _p(T, C.c_double),
                                 _p(ff, C.c_double), C.byref(it))
    return p, it.value


def tukey_weight(x):
    lib().svo_orc_tukey_weight.restype = C.c_float
    return float(lib().svo_orc_tukey_weight(C.c_float(x)))


def median_f(v):
    a = np.ascontiguousarray(v, dtype=np.float32)
    lib().svo_orc_median_f.restype = C.c_float
    return float(lib().svo_orc_median_f(_p(a, C.c_float), C.c_int(len(a))))


def inverse6(A):
    a = f64(np.asarray(A).reshape(36))
    out = np.zeros(36)
    lib().svo_orc_inverse6(_p(a, C.c_double), _p(out, C.c_double))
    return out.reshape(6, 6)


def ldlt3_solve(A, b):
    a, bb = f64(np.asarray(A).reshape(9)), f64(b)
    x = np.zeros(3)
    lib().svo_orc_ldlt3_solve(_p(a, C.c_double), _p(bb, C.c_double), _p(x, C.c_double))
    return x


# ---- next row f-3: FastDetector::detect ----
def fast(img, threshold=10):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cap = w * h
    xs, ys, ss = (np.zeros(cap, dtype=np.int32) for _ in range(3))
    n = lib().svo_orc_fast(_p(img, C.c_uint8), w, h, C.c_int(threshold), C.c_int(cap), _p(xs, C.c_int), _p(ys, C.c_int),
                           _p(ss, C.c_int))
    return xs[:n].copy(), ys[:n].copy(), ss[:n].copy()


def shi_tomasi_score(img, u, v):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    lib().svo_orc_shi_tomasi_score.restype = C.c_float
    return float(lib().svo_orc_shi_tomasi_score(_p(img, C.c_uint8), img.shape[1], img.shape[0], C.c_int(u), C.c_int(v)))


def detect_features(pyr, n_pyr_levels=3, cell_size=20, occupancy=None, detection_threshold=10.0):
    h, w = pyr[0].shape
    gc, gr = -(-w // cell_size), -(-h // cell_size)
    px = np.zeros((gc * gr, 2), dtype=np.int32)
    lvl = np.zeros(gc * gr, dtype=np.int32)
    sc = np.zeros(gc * gr, dtype=np.float32)
    occ = None if occupancy is None else np.ascontiguousarray(occupancy, dtype=np.uint8)
    n = lib().svo_orc_detect_features(pyr_ptrs(pyr), w, h, C.c_int(n_pyr_levels), C.c_int(cell_size),
                                      None if occ is None else _p(occ, C.c_uint8), C.c_double(detection_threshold),
                                      _p(px, C.c_int), _p(lvl, C.c_int), _p(sc, C.c_float))
    return px[:n].copy(), lvl[:n].copy(), sc[:n].copy()


# ---- next row f-2: the cell loop of Reprojector::reprojectMap ----
def reproject_cells(cam, kf_pyrs, T_kf_w, cur_pyr, T_cur_w, cell_offset, kf_slot, px_ref, f_ref, level_ref, pt_pos, edgelet,
                    grad, deleted, px_cur, max_fts=1200, n_pyr_levels=3, align_max_iter=10):
    c = camera(cam)
    n_kf, n_cells, n = len(kf_pyrs), len(cell_offset) - 1, int(cell_offset[-1])
    kp = (C.POINTER(C.POINTER(C.c_uint8)) * n_kf)()
    keep = []
    for k in range(n_kf):
        pp = pyr_ptrs(kf_pyrs[k])
        keep.append(pp)
        kp[k] = C.cast(pp, C.POINTER(C.POINTER(C.c_uint8)))
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    co, ks, lr = i32(cell_offset), i32(kf_slot), i32(level_ref)
    Tk, Tc, pr, fr, pp3, gr = f64(T_kf_w), f64(T_cur_w), f64(px_ref), f64(f_ref), f64(pt_pos), f64(grad)
    pc = f64(px_cur).copy()
    ed, de = np.ascontiguousarray(edgelet, dtype=np.uint8), np.ascontiguousarray(deleted, dtype=np.uint8)
    tried, matched = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
    sl, win = np.zeros(max(n, 1), np.int32), np.zeros(max(n_cells, 1), np.int32)
    nm, nt = C.c_size_t(0), C.c_size_t(0)
    I, DD = C.c_int, C.c_double
    lib().svo_orc_reproject_cells(C.byref(c), C.c_int(n_kf), kp, _p(Tk, DD), pyr_ptrs(cur_pyr), _p(Tc, DD), C.c_int(n_cells),
                                  _p(co, I), _p(ks, I), _p(pr, DD), _p(fr, DD), _p(lr, I), _p(pp3, DD), _p(ed, C.c_uint8),
                                  _p(gr, DD), _p(de, C.c_uint8), _p(pc, DD), C.c_int(max_fts), C.c_int(n_pyr_levels),
                                  C.c_int(align_max_iter), _p(tried, C.c_uint8), _p(matched, C.c_uint8), _p(sl, I), _p(win, I),
                                  C.byref(nm), C.byref(nt))
    return {"tried": tried[:n], "matched": matched[:n], "search_level": sl[:n], "cell_winner": win[:n_cells], "px_cur": pc,
            "n_matches": nm.value, "n_trials": nt.value}


# ---- camera model (a-13): vk::PinholeCamera::world2cam / cam2world, vk::AbstractCamera::isInFrame ----
def world2cam(cam, xyz):
    c = camera(cam)
    xyz = f64(xyz).reshape(-1, 3)
    out = np.zeros((len(xyz), 2))
    for i in range(len(xyz)):
        lib().svo_orc_world2cam(C.byref(c), _p(xyz[i], C.c_double), _p(out[i], C.c_double))
    return out


def world2cam_uv(cam, uv):
    c = camera(cam)
    uv = f64(uv).reshape(-1, 2)
    out = np.zeros((len(uv), 2))
    for i in range(len(uv)):
        lib().svo_orc_world2cam_uv(C.byref(c), _p(uv[i], C.c_double), _p(out[i], C.c_double))
    return out


def cam2world(cam, px):
    c = camera(cam)
    px = f64(px).reshape(-1, 2)
    out = np.zeros((len(px), 3))
    for i in range(len(px)):
        lib().svo_orc_cam2world(C.byref(c), C.c_double(px[i, 0]), C.c_double(px[i, 1]), _p(out[i], C.c_double))
    return out


def is_in_frame(cam, obs, boundary, level=-1):
    c = camera(cam)
    obs = np.ascontiguousarray(obs, dtype=np.int32).reshape(-1, 2)
    return np.array([lib().svo_orc_is_in_frame(C.byref(c), int(o[0]), int(o[1]), int(boundary), int(level)) for o in obs],
                    dtype=np.uint8)



# ---- Reprojector::reprojectMap on a flattened svo::Map (svo_orc_reproject_map) ----
class OrcMap(C.Structure):
    _fields_ = [("n_kf", C.c_int), ("T_kf_w", C.POINTER(C.c_double)), ("kf_key_point", C.POINTER(C.c_int)),
                ("kf_ftr_offset", C.POINTER(C.c_int)), ("kf_ftr_point", C.POINTER(C.c_int)), ("n_points", C.c_int),
                ("pt_pos", C.POINTER(C.c_double)), ("pt_type", C.POINTER(C.c_int)), ("pt_n_failed", C.POINTER(C.c_int)),
                ("pt_n_succeeded", C.POINTER(C.c_int)), ("pt_obs_offset", C.POINTER(C.c_int)), ("obs_kf", C.POINTER(C.c_int)),
                ("obs_px", C.POINTER(C.c_double)), ("obs_f", C.POINTER(C.c_double)), ("obs_level", C.POINTER(C.c_int)),
                ("obs_edgelet", C.POINTER(C.c_uint8)), ("obs_grad", C.POINTER(C.c_double)), ("n_candidates", C.c_int),
                ("cand_point", C.POINTER(C.c_int))]


def reproject_map(cs, kf_key_point, T_cur_w=None, max_fts=1200, max_n_kfs=10, n_pyr_levels=3, align_max_iter=10, state=None):
    """cs: a map case (android_svo_amd.synth.make_map_case layout); kf_key_point [n_kf][5] = the keyframes' key points
    (point indices, -1 for none).  state (optional): dict with pt_type / pt_n_failed / pt_n_succeeded / unlinked arrays
    that are updated in place (a sequence of frames over one map); default: copies of the case's arrays."""
    cam = camera(cs["cam"])
    n_kf, n_pts = cs["n_kf"], cs["n_points"]
    kp = (C.POINTER(C.POINTER(C.c_uint8)) * n_kf)()
    keep = []
    for k in range(n_kf):
        pp = pyr_ptrs(cs["kf_pyr"][k])
        keep.append(pp)
        kp[k] = C.cast(pp, C.POINTER(C.POINTER(C.c_uint8)))
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    if state is None:
        state = {"pt_type": i32(cs["pt_type"]).copy(), "pt_n_failed": i32(cs["pt_n_failed"]).copy(),
                 "pt_n_succeeded": i32(cs["pt_n_succeeded"]).copy(), "unlinked": np.zeros(n_pts, np.uint8)}
    a = dict(Tk=f64(cs["T_kf_w"]), key=i32(kf_key_point), ko=i32(cs["kf_ftr_offset"]), kfp=i32(cs["kf_ftr_point"]), pos=f64(cs["pt_pos"]),
             oo=i32(cs["pt_obs_offset"]), ok=i32(cs["obs_kf"]), opx=f64(cs["obs_px"]), of=f64(cs["obs_f"]), ol=i32(cs["obs_level"]),
             oe=np.ascontiguousarray(cs["obs_edgelet"], dtype=np.uint8), og=f64(cs["obs_grad"]), cp=i32(cs["cand_point"]))
    I, DD, U = C.c_int, C.c_double, C.c_uint8
    m = OrcMap(n_kf, _p(a["Tk"], DD), _p(a["key"], I), _p(a["ko"], I), _p(a["kfp"], I), n_pts, _p(a["pos"], DD), _p(state["pt_type"], I),
               _p(state["pt_n_failed"], I), _p(state["pt_n_succeeded"], I), _p(a["oo"], I), _p(a["ok"], I), _p(a["opx"], DD), _p(a["of"], DD),
               _p(a["ol"], I), _p(a["oe"], U), _p(a["og"], DD), len(a["cp"]), _p(a["cp"], I))
    Tc = f64(cs["T_cur_w"] if T_cur_w is None else T_cur_w)
    cw, ch, g = cs["cam"].width, cs["cam"].height, cs["cell_size"]
    n_cells = (-(-cw // g)) * (-(-ch // g))
    n_ov, ov_kf, ov_cnt = C.c_int(0), np.zeros(max(max_n_kfs, 1), np.int32), np.zeros(max(max_n_kfs, 1), np.int32)
    nf = C.c_int(0)
    fpx, fl, fp, fe, fg = np.zeros((n_cells, 2)), np.zeros(n_cells, np.int32), np.zeros(n_cells, np.int32), np.zeros(n_cells, np.uint8), np.zeros((n_cells, 2))
    nm, nt = C.c_size_t(0), C.c_size_t(0)
    cur = pyr_ptrs(cs["cur_pyr"] if "cur_pyr_override" not in cs else cs["cur_pyr_override"])
    lib().svo_orc_reproject_map(C.byref(cam), C.byref(m), kp, cur, _p(Tc, DD), I(g), I(max_fts), I(max_n_kfs), I(n_pyr_levels),
                                I(align_max_iter), _p(state["unlinked"], U), C.byref(n_ov), _p(ov_kf, I), _p(ov_cnt, I), C.byref(nf),
                                _p(fpx, DD), _p(fl, I), _p(fp, I), _p(fe, U), _p(fg, DD), C.byref(nm), C.byref(nt))
    k = nf.value
    return {"type": state["pt_type"], "n_failed": state["pt_n_failed"], "n_succeeded": state["pt_n_succeeded"], "unlinked": state["unlinked"],
            "overlap_kf": ov_kf[:n_ov.value], "overlap_count": ov_cnt[:n_ov.value], "feat_point": fp[:k], "feat_px": fpx[:k],
            "feat_level": fl[:k], "feat_type": fe[:k].astype(np.int32), "feat_grad": fg[:k], "n_matches": nm.value, "n_trials": nt.value}
