"""ctypes binding of csrc/libsvo_hip.so (the C-ABI of include/svo_hip.h).

This is the host-side mirror used by tests and bench.py; it adds nothing to the
computation.  There is NO CPU fallback: if the library is missing or no GPU is
present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVO_HIP_LIB") or os.path.join(_HERE, "csrc", "libsvo_hip.so")   # override: A/B builds of the kernels
MAX_LEVELS = 8
REDUCE_DOUBLES = 32

SEED_BEHIND, SEED_NOT_IN_FRAME, SEED_NO_MATCH, SEED_UPDATED, SEED_CONVERGED, SEED_NAN = range(6)

# svo_hip_sia_set_option (per solver object; the library itself reads no environment variable)
SIA_OPT_MODE, SIA_OPT_WAVES, SIA_OPT_CHUNKS, SIA_OPT_EXTRA_LDS, SIA_OPT_OLD_TILES, SIA_OPT_ARITH = range(6)
SIA_OPT_METHOD, SIA_OPT_SCALE_ESTIMATOR, SIA_OPT_WEIGHT_FUNCTION, SIA_OPT_CHI2 = 6, 7, 8, 9
SIA_CHI2_PER_PATCH, SIA_CHI2_REFERENCE_ORDER = 0, 1
# vk::NLLSSolver's enumerators (I/nlls_solver.h:46-48)
SIA_METHOD_GAUSS_NEWTON, SIA_METHOD_LEVENBERG_MARQUARDT = 0, 1
SIA_SCALE_UNIT, SIA_SCALE_TDIST, SIA_SCALE_MAD, SIA_SCALE_NORMAL = range(4)
SIA_WEIGHT_UNIT, SIA_WEIGHT_TDIST, SIA_WEIGHT_TUKEY, SIA_WEIGHT_HUBER = range(4)
SIA_MODE_AUTO, SIA_MODE_STREAM = 0, 1
SIA_ARITH_EXACT, SIA_ARITH_FAST, SIA_ARITH_MOMENTS_F32 = 0, 1, 2      # include/svo_hip.h: the reference's arithmetic / contracted f32 sums in the fused kernel
# options every SparseImgAlign object created from now on starts with: {option: value}.  Test fixtures and A/B scripts
# set this (a Python-side default, applied per object through the C-ABI).
SIA_DEFAULT_OPTIONS: dict = {}


class SvoHipError(RuntimeError):
    pass


class CCamera(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("d", C.c_double * 5), ("distortion", C.c_int)]


class CCtxStats(C.Structure):
    """svo_hip_ctx_stats (include/svo_hip.h)"""
    _fields_ = [("allocator_calls", C.c_ulonglong), ("free_calls", C.c_ulonglong),
                ("seed_pool_free_device_bytes", C.c_ulonglong), ("seed_pool_free_host_bytes", C.c_ulonglong),
                ("scratch_bytes", C.c_ulonglong), ("staging_bytes", C.c_ulonglong),
                ("seed_blocks_in_use", C.c_int), ("seed_blocks_free", C.c_int)]


class CSiaParams(C.Structure):
    _fields_ = [("max_level", C.c_int), ("min_level", C.c_int), ("n_iter", C.c_int), ("eps", C.c_double),
                ("early_stop", C.c_int)]


class CSiaResult(C.Structure):
    _fields_ = [("T_cur_w", C.c_double * 7), ("n_tracked", C.c_uint64), ("H", C.c_double * 36),
                ("chi2", C.c_double), ("stop", C.c_int), ("iters", C.c_int * MAX_LEVELS),
                ("n_precompute_patches", C.c_uint64), ("n_residual_patches", C.c_uint64)]


class CDfParams(C.Structure):
    _fields_ = [("n_pyr_levels", C.c_int), ("align_max_iter", C.c_int), ("max_epi_search_steps", C.c_int),
                ("seed_convergence_sigma2_thresh", C.c_double)]


_lib = None


def load_library() -> C.CDLL:
    """Load libsvo_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SvoHipError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C android_svo_amd/csrc)" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.svo_hip_last_error.restype = C.c_char_p
        _lib.svo_hip_version.restype = C.c_char_p
        _lib.svo_hip_ctx_stream.restype = C.c_void_p
    return _lib


def make_camera(cam, dist: Optional[Sequence[float]] = None) -> CCamera:
    if dist is None:
        dist = getattr(cam, "dist", None)        # a camera object may carry its radtan coefficients
    c = CCamera()
    c.width, c.height = int(cam.width), int(cam.height)
    c.fx, c.fy, c.cx, c.cy = cam.fx, cam.fy, cam.cx, cam.cy
    d = list(dist) if dist is not None else [0.0] * 5
    for i in range(5):
        c.d[i] = d[i]
    c.distortion = 1 if abs(d[0]) > 1e-7 else 0     # pinhole_camera.cpp:27
    return c


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


class Context:
    """svo_hip_ctx: one device + one stream."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.svo_hip_ctx_create(C.byref(self.h), C.c_int(device), C.c_void_p(stream))
        if rc != 0:
            raise SvoHipError("svo_hip_ctx_create failed (%d): no usable MI355X / HIP device %d" % (rc, device))

    def check(self, rc: int, what: str = ""):
        if rc != 0:
            msg = self.lib.svo_hip_last_error(self.h)
            raise SvoHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def sync(self):
        self.check(self.lib.svo_hip_ctx_sync(self.h), "sync")

    @property
    def stream(self) -> int:
        return self.lib.svo_hip_ctx_stream(self.h) or 0

    def info(self) -> dict:
        """svo_hip_ctx_info: allocator / free calls made for this context's work areas and seed-batch pool, pool occupancy"""
        st = CCtxStats()
        self.check(self.lib.svo_hip_ctx_info(self.h, C.byref(st)), "ctx_info")
        return {k: int(getattr(st, k)) for k, _ in CCtxStats._fields_}

    def trim(self):
        """svo_hip_ctx_trim: the free blocks of the seed-batch pool go back to the driver (synchronises the device)"""
        self.check(self.lib.svo_hip_ctx_trim(self.h), "ctx_trim")

    def set_small_pass_limit(self, max_seeds: int):
        """svo_hip_df_set_small_pass_limit: depth-filter passes over resident seed batches of at most this many seed
        records take the two-launch form (0: never)."""
        self.check(self.lib.svo_hip_df_set_small_pass_limit(self.h, int(max_seeds)), "df_set_small_pass_limit")

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self.check(self.lib.svo_hip_malloc(self.h, C.byref(p), C.c_size_t(nbytes)), "malloc")
        return p.value

    def free(self, ptr: int):
        self.check(self.lib.svo_hip_free(self.h, C.c_void_p(ptr)), "free")

    def to_device(self, arr: np.ndarray) -> "DeviceArray":
        return DeviceArray(self, arr=np.ascontiguousarray(arr))

    def empty(self, shape, dtype) -> "DeviceArray":
        return DeviceArray(self, shape=shape, dtype=dtype)

    def close(self):
        if self.h:
            self.lib.svo_hip_ctx_destroy(self.h)
            self.h = C.c_void_p()


class DeviceArray:
    """A typed device allocation owned through the C-ABI (svo_hip_malloc)."""

    def __init__(self, ctx: Context, arr: Optional[np.ndarray] = None, shape=None, dtype=None):
        self.ctx = ctx
        if arr is not None:
            shape, dtype = arr.shape, arr.dtype
        self.shape = tuple(np.atleast_1d(shape)) if not isinstance(shape, tuple) else shape
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = ctx.malloc(max(self.nbytes, 1))
        if arr is not None:
            self.upload(arr)

    def upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        assert a.nbytes == self.nbytes
        self.ctx.check(self.ctx.lib.svo_hip_memcpy_h2d(self.ctx.h, C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p),
                                                       C.c_size_t(self.nbytes)), "h2d")
        self.ctx.sync()        # the source numpy buffer may be a temporary

    def download(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        self.ctx.check(self.ctx.lib.svo_hip_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                                                       C.c_size_t(self.nbytes)), "d2h")
        return out

    def free(self):
        if self.ptr:
            self.ctx.free(self.ptr)
            self.ptr = 0


def ordered_sum_f32(ctx: Context, values: np.ndarray) -> np.float32:
    """svo_hip_ordered_sum_f32_dev: the f32 sum of `values` in index order, rounded as a scalar loop rounds it"""
    v = np.ascontiguousarray(values, dtype=np.float32).ravel()
    d_v = DeviceArray(ctx, v) if len(v) else None
    d_o = DeviceArray(ctx, shape=(1,), dtype=np.float32)
    ctx.check(ctx.lib.svo_hip_ordered_sum_f32_dev(ctx.h, C.c_void_p(d_v.ptr if d_v else 0), C.c_size_t(len(v)), C.c_void_p(d_o.ptr)), "ordered_sum")
    out = d_o.download()[0]
    d_o.free()
    if d_v:
        d_v.free()
    return out


class Pyramid:
    """svo_hip_pyramid: a batch of image pyramids resident in HBM."""

    def __init__(self, ctx: Context, width: int, height: int, n_levels: int = 5, batch: int = 1):
        self.ctx, self.width, self.height, self.n_levels, self.batch = ctx, width, height, n_levels, batch
        self.h = C.c_void_p()
        ctx.check(ctx.lib.svo_hip_pyramid_create(ctx.h, width, height, n_levels, batch, C.byref(self.h)), "pyramid_create")

    def upload(self, slot: int, levels: List[np.ndarray]):
        assert len(levels) >= self.n_levels
        arr = (C.POINTER(C.c_uint8) * MAX_LEVELS)()
        keep = []
        for l in range(self.n_levels):
            im = np.ascontiguousarray(levels[l], dtype=np.uint8)
            assert im.shape == (self.height >> l, self.width >> l), (im.shape, l)
            keep.append(im)
            arr[l] = _ptr(im, C.c_uint8)
        self.ctx.check(self.ctx.lib.svo_hip_pyramid_upload(self.h, slot, arr), "pyramid_upload")
        self.ctx.sync()

    def upload_level0_and_build(self, slot: int, img: np.ndarray):
        im = np.ascontiguousarray(img, dtype=np.uint8)
        assert im.shape == (self.height, self.width)
        self.ctx.check(self.ctx.lib.svo_hip_pyramid_upload_level0_and_build(self.h, slot, _ptr(im, C.c_uint8)), "pyramid_build")
        self.ctx.sync()

    def download_level(self, slot: int, level: int) -> np.ndarray:
        out = np.empty((self.height >> level, self.width >> level), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.svo_hip_pyramid_download_level(self.h, slot, level, _ptr(out, C.c_uint8)), "pyramid_download")
        return out

    def destroy(self):
        if self.h:
            self.ctx.lib.svo_hip_pyramid_destroy(self.h)
            self.h = C.c_void_p()


class SparseImgAlign:
    """Batched svo::SparseImgAlign (I/sparse_img_align.h:33-79) over `batch` frame pairs."""

    def __init__(self, ctx: Context, batch: int, max_features: int):
        self.ctx, self.batch, self.max_features = ctx, batch, max_features
        self.h = C.c_void_p()
        ctx.check(ctx.lib.svo_hip_sia_create(ctx.h, batch, max_features, C.byref(self.h)), "sia_create")
        for opt, val in SIA_DEFAULT_OPTIONS.items():
            self.set_option(opt, val)

    def set_option(self, option: int, value: int):
        self.ctx.check(self.ctx.lib.svo_hip_sia_set_option(self.h, int(option), int(value)), "sia_set_option")

    def set_mode(self, stream: bool):
        """svo_hip_sia_run: True = always the streaming kernels, False = automatic (the fused kernel where it applies)"""
        self.set_option(SIA_OPT_MODE, SIA_MODE_STREAM if stream else SIA_MODE_AUTO)

    def set_method(self, method: int):
        """NLLSSolver::method_: SIA_METHOD_GAUSS_NEWTON (default) or SIA_METHOD_LEVENBERG_MARQUARDT"""
        self.set_option(SIA_OPT_METHOD, method)

    def set_robust_cost_function(self, scale_estimator: int, weight_function: int):
        """NLLSSolver::setRobustCostFunction (I/nlls_solver_impl.hpp:229-281); SIA_SCALE_UNIT switches the weights off"""
        self.set_option(SIA_OPT_SCALE_ESTIMATOR, scale_estimator)
        self.set_option(SIA_OPT_WEIGHT_FUNCTION, weight_function)

    def solver_state(self, slot: int = 0):
        """(scale_, mu_, nu_) of a slot after a run with Levenberg-Marquardt or a robust cost"""
        sc, mu, nu = C.c_float(), C.c_double(), C.c_double()
        self.ctx.check(self.ctx.lib.svo_hip_sia_solver_state(self.h, slot, C.byref(sc), C.byref(mu), C.byref(nu)), "sia_solver_state")
        return np.float32(sc.value), mu.value, nu.value

    def set_frames(self, ref: Pyramid, cur: Pyramid):
        self.ref, self.cur = ref, cur
        self.ctx.check(self.ctx.lib.svo_hip_sia_set_frames(self.h, ref.h, cur.h), "sia_set_frames")

    def upload_pair(self, slot: int, fp, T_cur_w_init=None):
        px, f, pos = _f64(fp.px), _f64(fp.f), _f64(fp.pos)
        hp = np.ascontiguousarray(fp.has_point, dtype=np.uint8)
        L = self.ctx.lib
        self.ctx.check(L.svo_hip_sia_upload_features(self.h, slot, len(px), _ptr(px, C.c_double), _ptr(f, C.c_double),
                                                     _ptr(pos, C.c_double), _ptr(hp, C.c_uint8)), "sia_upload_features")
        cam = make_camera(fp.cam, getattr(fp, "dist", None))
        Tr = _f64(fp.T_ref_w)
        Ti = _f64(fp.T_cur_w_init if T_cur_w_init is None else T_cur_w_init)
        self.ctx.check(L.svo_hip_sia_upload_poses(self.h, slot, C.byref(cam), _ptr(Tr, C.c_double), _ptr(Ti, C.c_double)),
                       "sia_upload_poses")
        self.ctx.sync()

    def set_shard(self, rank: int, world: int):
        self.ctx.check(self.ctx.lib.svo_hip_sia_set_shard(self.h, rank, world), "sia_set_shard")

    @staticmethod
    def params(max_level=4, min_level=0, n_iter=30, eps=1e-6, early_stop=True) -> CSiaParams:
        return CSiaParams(max_level, min_level, n_iter, eps, 1 if early_stop else 0)

    def run(self, n_slots: int, prm: CSiaParams):
        self.ctx.check(self.ctx.lib.svo_hip_sia_run(self.h, n_slots, C.byref(prm)), "sia_run")

    # stepwise form
    def begin(self, n_slots, prm): self.ctx.check(self.ctx.lib.svo_hip_sia_begin(self.h, n_slots, C.byref(prm)), "sia_begin")
    def level_begin(self, level): self.ctx.check(self.ctx.lib.svo_hip_sia_level_begin(self.h, level), "sia_level_begin")
    def accumulate(self): self.ctx.check(self.ctx.lib.svo_hip_sia_accumulate(self.h), "sia_accumulate")
    def solve_update(self): self.ctx.check(self.ctx.lib.svo_hip_sia_solve_update(self.h), "sia_solve_update")
    def finish(self): self.ctx.check(self.ctx.lib.svo_hip_sia_finish(self.h), "sia_finish")

    def reduce_buffer(self):
        p = C.c_void_p()
        n = C.c_size_t()
        self.ctx.check(self.ctx.lib.svo_hip_sia_reduce_buffer(self.h, C.byref(p), C.byref(n)), "sia_reduce_buffer")
        return p.value, n.value

    def set_reduce_buffer(self, dev_ptr: int):
        self.ctx.check(self.ctx.lib.svo_hip_sia_set_reduce_buffer(self.h, C.c_void_p(dev_ptr)), "sia_set_reduce_buffer")

    def last_run_mode(self) -> int:
        m = C.c_int(-1)
        self.ctx.check(self.ctx.lib.svo_hip_sia_last_run_mode(self.h, C.byref(m)), "sia_last_run_mode")
        return m.value

    def set_profiling(self, enable: bool):
        self.ctx.check(self.ctx.lib.svo_hip_sia_set_profiling(self.h, 1 if enable else 0), "sia_set_profiling")

    def get_profile(self):
        rms, pms = C.c_double(), C.c_double()
        rn, pn = C.c_uint64(), C.c_uint64()
        self.ctx.check(self.ctx.lib.svo_hip_sia_get_profile(self.h, C.byref(rms), C.byref(rn), C.byref(pms), C.byref(pn)),
                       "sia_get_profile")
        return {"residual_ms": rms.value, "residual_launches": rn.value, "precompute_ms": pms.value,
                "precompute_launches": pn.value}

    def download(self, slot: int) -> CSiaResult:
        out = CSiaResult()
        self.ctx.check(self.ctx.lib.svo_hip_sia_download(self.h, slot, C.byref(out)), "sia_download")
        return out

    def download_all(self, n_slots: int):
        arr = (CSiaResult * n_slots)()
        self.ctx.check(self.ctx.lib.svo_hip_sia_download_all(self.h, n_slots, arr), "sia_download_all")
        return list(arr)

    def download_caches(self, slot: int, n: int):
        ref = np.zeros((n, 16), dtype=np.float32)
        dx = np.zeros((n, 16), dtype=np.float32)
        dy = np.zeros((n, 16), dtype=np.float32)
        vis = np.zeros(n, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.svo_hip_sia_download_caches(self.h, slot, _ptr(ref, C.c_float), _ptr(dx, C.c_float),
                                                                _ptr(dy, C.c_float), _ptr(vis, C.c_uint8)), "sia_download_caches")
        return ref, dx, dy, vis

    def download_fused_patches(self, slot: int, n: int, level: int):
        """reference patches as the fused kernel forms them at `level`: (ref[n][16], dx[n][16], dy[n][16], valid[n])"""
        ref = np.zeros((n, 16), dtype=np.float32)
        dx = np.zeros((n, 16), dtype=np.float32)
        dy = np.zeros((n, 16), dtype=np.float32)
        valid = np.zeros(n, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.svo_hip_sia_download_fused_patches(self.h, slot, level, _ptr(ref, C.c_float), _ptr(dx, C.c_float),
                                                                       _ptr(dy, C.c_float), _ptr(valid, C.c_uint8)), "sia_download_fused_patches")
        return ref, dx, dy, valid

    def destroy(self):
        if self.h:
            self.ctx.lib.svo_hip_sia_destroy(self.h)
            self.h = C.c_void_p()


def align2d_batch(ctx: Context, cur: Pyramid, slot: int, level: int, pwb: np.ndarray, patch: np.ndarray, n_iter: int,
                  px: np.ndarray):
    """feature_alignment::align2D over n patches (host buffers in, host buffers out)."""
    n = len(px)
    pwb = np.ascontiguousarray(pwb, dtype=np.uint8).reshape(n, 100)
    patch = np.ascontiguousarray(patch, dtype=np.uint8).reshape(n, 64)
    p = _f64(px).copy()
    conv = np.zeros(n, dtype=np.uint8)
    iters = np.zeros(n, dtype=np.int32)
    ctx.check(ctx.lib.svo_hip_align2d_batch(ctx.h, cur.h, slot, level, n, _ptr(pwb, C.c_uint8), _ptr(patch, C.c_uint8),
                                            n_iter, _ptr(p, C.c_double), _ptr(conv, C.c_uint8), _ptr(iters, C.c_int32)),
              "align2d_batch")
    return conv.astype(bool), p, iters


def align1d_batch(ctx: Context, cur: Pyramid, slot: int, level: int, pwb: np.ndarray, dirs: np.ndarray, n_iter: int,
                  px: np.ndarray):
    """feature_alignment::align1D over n patches; returns (converged, px, h_inv, iters)."""
    n = len(px)
    d_pwb = ctx.to_device(np.ascontiguousarray(pwb, dtype=np.uint8).reshape(n, 100))
    d_dir = ctx.to_device(np.ascontiguousarray(dirs, dtype=np.float32).reshape(n, 2))
    d_px = ctx.to_device(_f64(px))
    d_conv, d_h, d_it = ctx.empty((n,), np.uint8), ctx.empty((n,), np.float64), ctx.empty((n,), np.int32)
    ctx.check(ctx.lib.svo_hip_align1d_batch_dev(ctx.h, cur.h, slot, level, n, C.c_void_p(d_pwb.ptr), C.c_void_p(d_dir.ptr),
                                                n_iter, C.c_void_p(d_px.ptr), C.c_void_p(d_conv.ptr), C.c_void_p(d_h.ptr),
                                                C.c_void_p(d_it.ptr)), "align1d_batch")
    out = d_conv.download().astype(bool), d_px.download(), d_h.download(), d_it.download()
    for d in (d_pwb, d_dir, d_px, d_conv, d_h, d_it):
        d.free()
    return out


def match_direct_batch(ctx: Context, ref: Pyramid, cur: Pyramid, cur_slot: int, cam, T_ref_w, T_cur_w, kf_slot, px_ref,
                       f_ref, level_ref, pt_pos, px_cur, edgelet=None, grad=None, n_pyr_levels=3, align_max_iter=10):
    """Matcher::findMatchDirect over n items; returns (success, px_cur, search_level)."""
    n = len(px_cur)
    T_ref_w = _f64(T_ref_w).reshape(-1, 7)
    d = [ctx.to_device(T_ref_w), ctx.to_device(np.ascontiguousarray(kf_slot, dtype=np.int32)), ctx.to_device(_f64(px_ref)),
         ctx.to_device(_f64(f_ref)), ctx.to_device(np.ascontiguousarray(level_ref, dtype=np.int32)),
         ctx.to_device(_f64(pt_pos)), ctx.to_device(_f64(px_cur))]
    d_e = ctx.to_device(np.ascontiguousarray(edgelet, dtype=np.uint8)) if edgelet is not None else None
    d_g = ctx.to_device(_f64(grad)) if grad is not None else None
    d_ok, d_sl = ctx.empty((n,), np.uint8), ctx.empty((n,), np.int32)
    c = make_camera(cam)
    Tc = _f64(T_cur_w)
    ctx.check(ctx.lib.svo_hip_match_direct_batch_dev(
        ctx.h, ref.h, cur.h, cur_slot, C.byref(c), len(T_ref_w), C.c_void_p(d[0].ptr), _ptr(Tc, C.c_double), n,
        C.c_void_p(d[1].ptr), C.c_void_p(d[2].ptr), C.c_void_p(d[3].ptr), C.c_void_p(d[4].ptr), C.c_void_p(d[5].ptr),
        C.c_void_p(d_e.ptr if d_e else None), C.c_void_p(d_g.ptr if d_g else None), n_pyr_levels, align_max_iter,
        C.c_void_p(d[6].ptr), C.c_void_p(d_ok.ptr), C.c_void_p(d_sl.ptr)), "match_direct_batch")
    out = d_ok.download().astype(bool), d[6].download(), d_sl.download()
    for x in d + [d_ok, d_sl] + [v for v in (d_e, d_g) if v is not None]:
        x.free()
    return out


def update_seed_batch(ctx: Context, x, tau2, a, b, mu, z_range, sigma2):
    """static DepthFilter::updateSeed over SoA float32 arrays; returns the new (a, b, mu, sigma2)."""
    n = len(x)
    d = {k: ctx.to_device(np.ascontiguousarray(v, dtype=np.float32)) for k, v in
         dict(x=x, tau2=tau2, a=a, b=b, mu=mu, z_range=z_range, sigma2=sigma2).items()}
    ctx.check(ctx.lib.svo_hip_update_seed_batch_dev(ctx.h, n, *[C.c_void_p(d[k].ptr) for k in
                                                                ("x", "tau2", "a", "b", "mu", "z_range", "sigma2")]),
              "update_seed_batch")
    out = tuple(d[k].download() for k in ("a", "b", "mu", "sigma2"))
    for v in d.values():
        v.free()
    return out


def camera_batch(ctx: Context, cam, xyz=None, uv=None, px=None, obs=None, boundary=0, level=-1):
    """The device camera model over a batch: returns (px_of_xyz, px_of_uv, f_of_px, in_frame), None where not asked."""
    c = make_camera(cam)
    arrs = [a for a in (xyz, uv, px, obs) if a is not None]
    n = len(arrs[0])
    xyz = _f64(xyz).reshape(n, 3) if xyz is not None else None
    uv = _f64(uv).reshape(n, 2) if uv is not None else None
    px = _f64(px).reshape(n, 2) if px is not None else None
    obs = np.ascontiguousarray(obs, dtype=np.int32).reshape(n, 2) if obs is not None else None
    o_xyz = np.zeros((n, 2)) if xyz is not None else None
    o_uv = np.zeros((n, 2)) if uv is not None else None
    o_f = np.zeros((n, 3)) if px is not None else None
    o_in = np.zeros(n, dtype=np.uint8) if obs is not None else None

    def ptr(a, t):
        return _ptr(a, t) if a is not None else None
    ctx.check(ctx.lib.svo_hip_camera_batch(ctx.h, C.byref(c), n, ptr(xyz, C.c_double), ptr(uv, C.c_double), ptr(px, C.c_double),
                                           ptr(obs, C.c_int32), int(boundary), int(level), ptr(o_xyz, C.c_double),
                                           ptr(o_uv, C.c_double), ptr(o_f, C.c_double), ptr(o_in, C.c_uint8)), "camera_batch")
    return o_xyz, o_uv, o_f, o_in


class Comm:
    """svo_hip_comm: RCCL (kind="rccl": `unique_id` = the 128 bytes rank 0 got from Comm.unique_id()) or the host-staged
    shared-memory transport (kind="shm": `name` = "/segment-name")."""

    def __init__(self, ctx: Context, rank: int, world: int, kind: str = "rccl", unique_id: Optional[bytes] = None,
                 name: Optional[str] = None, slot_bytes: int = 1 << 20):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.h = C.c_void_p()
        if kind == "rccl":
            buf = (C.c_char * 128).from_buffer_copy(unique_id)
            ctx.check(ctx.lib.svo_hip_comm_create_rccl(ctx.h, buf, rank, world, C.byref(self.h)), "comm_create_rccl")
        else:
            ctx.check(ctx.lib.svo_hip_comm_create_shm(ctx.h, name.encode(), rank, world, C.c_size_t(slot_bytes), C.byref(self.h)),
                      "comm_create_shm")

    @classmethod
    def from_nccl(cls, ctx: Context, nccl_comm: int, rank: int, world: int) -> "Comm":
        """Adopt an ncclComm_t the application made itself (svo_hip_comm_from_nccl): the library uses it, never destroys it."""
        self = cls.__new__(cls)
        self.ctx, self.rank, self.world = ctx, rank, world
        self.h = C.c_void_p()
        ctx.check(ctx.lib.svo_hip_comm_from_nccl(ctx.h, C.c_void_p(nccl_comm), rank, world, C.byref(self.h)), "comm_from_nccl")
        return self

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * 128)()
        rc = load_library().svo_hip_comm_unique_id(buf)
        if rc != 0:
            raise SvoHipError("svo_hip_comm_unique_id failed (%d): is librccl loadable?" % rc)
        return bytes(buf)

    def count(self) -> int:
        """ranks the transport itself reports (ncclCommCount / the shared-memory header): svo_hip_comm_count"""
        n = C.c_int(0)
        self.ctx.check(self.ctx.lib.svo_hip_comm_count(self.h, C.byref(n)), "comm_count")
        return n.value

    def destroy(self):
        if self.h:
            self.ctx.lib.svo_hip_comm_destroy(self.h)
            self.h = C.c_void_p()


def sia_run_sharded(sia, comm: Comm, n_slots: int, prm, graph: Optional[bool] = None):
    if graph is not None:
        sia.ctx.check(sia.ctx.lib.svo_hip_sia_set_sharded_graph(sia.h, 1 if graph else 0), "sia_set_sharded_graph")
    sia.ctx.check(sia.ctx.lib.svo_hip_sia_run_sharded(sia.h, comm.h, n_slots, C.byref(prm)), "sia_run_sharded")


def seed_gather_converged(ctx: Context, comm: Comm, seeds, id_offset: int, cap: int):
    """Returns (records [total][6] ordered by rank then seed, counts[world]) after the device-side gather."""
    rec = ctx.empty((comm.world * cap, 6), np.float64)
    cnt = ctx.empty((comm.world,), np.int32)
    ctx.check(ctx.lib.svo_hip_seed_gather_converged_dev(
        ctx.h, comm.h, seeds.n, C.c_longlong(id_offset), C.c_void_p(seeds.status.ptr), C.c_void_p(seeds.mu.ptr),
        C.c_void_p(seeds.sigma2.ptr), C.c_void_p(seeds.xyz.ptr), cap, C.c_void_p(rec.ptr), C.c_void_p(cnt.ptr)),
        "seed_gather_converged")
    counts = cnt.download()
    r = rec.download().reshape(comm.world, cap, 6)
    out = np.concatenate([r[k, :min(int(counts[k]), cap)] for k in range(comm.world)], axis=0)
    rec.free(); cnt.free()
    return out, counts


def compute_tau_batch(ctx: Context, T_ref_cur, f, z, px_error_angle):
    n = len(z)
    df, dz, dt = ctx.to_device(_f64(f)), ctx.to_device(_f64(z)), ctx.empty((n,), np.float64)
    T = _f64(T_ref_cur)
    ctx.check(ctx.lib.svo_hip_compute_tau_batch_dev(ctx.h, n, _ptr(T, C.c_double), C.c_void_p(df.ptr), C.c_void_p(dz.ptr),
                                                    C.c_double(px_error_angle), C.c_void_p(dt.ptr)), "compute_tau_batch")
    out = dt.download()
    for v in (df, dz, dt):
        v.free()
    return out


class SeedBatch:
    """SoA seed state resident on the device (svo::Seed a,b,mu,z_range,sigma2 + Feature px,f,level)."""

    def __init__(self, ctx: Context, px, f, level, a, b, mu, z_range, sigma2):
        self.ctx, self.n = ctx, len(px)
        self.px = ctx.to_device(_f64(px))
        self.f = ctx.to_device(_f64(f))
        self.level = ctx.to_device(np.ascontiguousarray(level, dtype=np.int32))
        self.a, self.b, self.mu, self.z_range, self.sigma2 = (
            ctx.to_device(np.ascontiguousarray(v, dtype=np.float32)) for v in (a, b, mu, z_range, sigma2))
        self.status = ctx.empty((self.n,), np.int32)
        self.z = ctx.empty((self.n,), np.float64)
        self.xyz = ctx.empty((self.n, 3), np.float64)
        self.n_zmssd = ctx.empty((self.n,), np.int32)
        self.n_align = ctx.empty((self.n,), np.int32)
        self.px_cur = ctx.empty((self.n, 2), np.float64)          # Matcher::px_cur_ of each seed's match
        self.search_level = ctx.empty((self.n,), np.int32)        # Matcher::search_level_

    def reset_state(self, a, b, mu, sigma2):
        for d, v in ((self.a, a), (self.b, b), (self.mu, mu), (self.sigma2, sigma2)):
            d.upload(np.ascontiguousarray(v, dtype=np.float32))

    def free(self):
        for d in (self.px, self.f, self.level, self.a, self.b, self.mu, self.z_range, self.sigma2, self.status, self.z,
                  self.xyz, self.n_zmssd, self.n_align, self.px_cur, self.search_level) + tuple(getattr(self, "_blocks", ())):
            d.free()


class CSeedEvent(C.Structure):
    _fields_ = [("index", C.c_int32), ("status", C.c_int32), ("mu", C.c_float), ("sigma2", C.c_float),
                ("xyz_world", C.c_double * 3), ("px_cur", C.c_double * 2)]


SEED_EVENT_DTYPE = np.dtype([("index", np.int32), ("status", np.int32), ("mu", np.float32), ("sigma2", np.float32),
                             ("xyz_world", np.float64, (3,)), ("px_cur", np.float64, (2,))])
SEED_ERASED = -1


class ResidentSeeds:
    """svo_hip_seed_batch: the seeds of one keyframe resident on the device from creation to their end (uploaded once;
    per frame two poses go down and only the converged / NaN seeds -- on keyframes also the updated ones -- come back)."""

    def __init__(self, ctx: Context, px, f, level, a, b, mu, z_range, sigma2):
        self.ctx, self.n = ctx, len(px)
        self.h = C.c_void_p()
        px, f = _f64(px), _f64(f)
        lvl = np.ascontiguousarray(level, dtype=np.int32)
        st = [np.ascontiguousarray(v, dtype=np.float32) for v in (a, b, mu, z_range, sigma2)]
        ctx.check(ctx.lib.svo_hip_seed_batch_create(ctx.h, self.n, _ptr(px, C.c_double), _ptr(f, C.c_double), _ptr(lvl, C.c_int32),
                                                    *[_ptr(v, C.c_float) for v in st], C.byref(self.h)), "seed_batch_create")

    def update_async(self, ref: Pyramid, ref_slot: int, cur: Pyramid, cur_slot: int, cam, T_ref_w, T_cur_w, prm=None, report_updated=False):
        prm = prm or depth_filter_params()
        c = make_camera(cam)
        Tr, Tc = _f64(T_ref_w), _f64(T_cur_w)
        self.ctx.check(self.ctx.lib.svo_hip_seed_batch_update_async(self.h, ref.h, ref_slot, cur.h, cur_slot, C.byref(c), _ptr(Tr, C.c_double),
                                                                    _ptr(Tc, C.c_double), C.byref(prm), 1 if report_updated else 0),
                       "seed_batch_update_async")

    @staticmethod
    def update_group_async(batches, ref: Pyramid, ref_slots, cur: Pyramid, cur_slot: int, cam, T_ref_w, T_cur_w, prm=None, report_updated=False):
        """svo_hip_seed_batch_update_group_async: the pass over several batches (one per keyframe: slot ref_slots[k] of
        `ref`, pose T_ref_w[k]) as ONE set of launches; collect every batch afterwards."""
        ctx = batches[0].ctx
        prm = prm or depth_filter_params()
        c = make_camera(cam)
        hs = (C.c_void_p * len(batches))(*[b.h for b in batches])
        slots = np.ascontiguousarray(ref_slots, dtype=np.int32)
        Tr, Tc = _f64(np.asarray(T_ref_w, dtype=np.float64).reshape(len(batches), 7)), _f64(T_cur_w)
        ctx.check(ctx.lib.svo_hip_seed_batch_update_group_async(len(batches), hs, ref.h, _ptr(slots, C.c_int), cur.h, cur_slot, C.byref(c),
                                                                _ptr(Tr, C.c_double), _ptr(Tc, C.c_double), C.byref(prm),
                                                                1 if report_updated else 0), "seed_batch_update_group_async")

    def collect_raw(self):
        """(address of the batch's page-locked event block, number of events, status counts [7]) -- no copy"""
        ev = C.POINTER(CSeedEvent)()
        n = C.c_int(0)
        counts = (C.c_int32 * 7)()
        self.ctx.check(self.ctx.lib.svo_hip_seed_batch_collect(self.h, C.byref(ev), C.byref(n), counts), "seed_batch_collect")
        return (C.addressof(ev.contents) if n.value else 0), n.value, np.array(list(counts), dtype=np.int64)

    def collect(self):
        """(events as a structured array -- a copy --, status counts [7]: slot = status + 1)"""
        ev = C.POINTER(CSeedEvent)()
        n = C.c_int(0)
        counts = (C.c_int32 * 7)()
        self.ctx.check(self.ctx.lib.svo_hip_seed_batch_collect(self.h, C.byref(ev), C.byref(n), counts), "seed_batch_collect")
        if n.value:
            buf = (C.c_char * (n.value * C.sizeof(CSeedEvent))).from_address(C.addressof(ev.contents))
            arr = np.frombuffer(buf, dtype=SEED_EVENT_DTYPE).copy()
        else:
            arr = np.zeros(0, dtype=SEED_EVENT_DTYPE)
        return arr, np.array(list(counts), dtype=np.int64)

    def update(self, *args, **kw):
        self.update_async(*args, **kw)
        return self.collect()

    def erase(self, indices):
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        self.ctx.check(self.ctx.lib.svo_hip_seed_batch_erase(self.h, len(idx), _ptr(idx, C.c_int32)), "seed_batch_erase")

    def n_alive(self) -> int:
        n, na = C.c_int(0), C.c_int(0)
        self.ctx.check(self.ctx.lib.svo_hip_seed_batch_size(self.h, C.byref(n), C.byref(na)), "seed_batch_size")
        return na.value

    def download(self):
        out = {k: np.empty(self.n, np.float32) for k in ("a", "b", "mu", "sigma2")}
        out["alive"] = np.empty(self.n, np.uint8)
        self.ctx.check(self.ctx.lib.svo_hip_seed_batch_download(self.h, _ptr(out["a"], C.c_float), _ptr(out["b"], C.c_float),
                                                                _ptr(out["mu"], C.c_float), _ptr(out["sigma2"], C.c_float),
                                                                _ptr(out["alive"], C.c_uint8)), "seed_batch_download")
        return out

    def status(self) -> np.ndarray:
        p = C.c_void_p()
        self.ctx.check(self.ctx.lib.svo_hip_seed_batch_arrays(self.h, None, None, None, None, C.byref(p)), "seed_batch_arrays")
        return DeviceView(self.ctx, p.value, (self.n,), np.int32).download()

    def destroy(self):
        if self.h:
            self.ctx.lib.svo_hip_seed_batch_destroy(self.h)
            self.h = C.c_void_p()


class DeviceView:
    """A typed window into somebody else's device allocation (never freed through the view)."""

    def __init__(self, ctx: Context, ptr: int, shape, dtype):
        self.ctx, self.ptr, self.shape, self.dtype = ctx, ptr, tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize

    def download(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        self.ctx.check(self.ctx.lib.svo_hip_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                                                       C.c_size_t(self.nbytes)), "d2h")
        return out

    def upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        assert a.nbytes == self.nbytes
        self.ctx.check(self.ctx.lib.svo_hip_memcpy_h2d(self.ctx.h, C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p),
                                                       C.c_size_t(self.nbytes)), "h2d")
        self.ctx.sync()

    def free(self):
        pass


def pack_seed_state(sb: "SeedBatch"):
    """Move a, b, mu, sigma2 of a SeedBatch into ONE device block (so that a benchmark can restore the state of all seeds
    with one device-to-device copy); returns (block, pristine copy of the block)."""
    ctx, n = sb.ctx, sb.n
    block = ctx.empty((4, n), np.float32)
    host = np.stack([sb.a.download(), sb.b.download(), sb.mu.download(), sb.sigma2.download()])
    block.upload(host)
    for k, name in enumerate(("a", "b", "mu", "sigma2")):
        getattr(sb, name).free()
        setattr(sb, name, DeviceView(ctx, block.ptr + 4 * n * k, (n,), np.float32))
    pristine = ctx.to_device(host)
    sb._blocks = (block, pristine)        # freed with the batch
    return block, pristine


def depth_filter_params(n_pyr_levels=3, align_max_iter=10, max_epi_search_steps=1000, conv_thresh=100.0) -> CDfParams:
    return CDfParams(n_pyr_levels, align_max_iter, max_epi_search_steps, conv_thresh)


def depth_filter_update(ctx: Context, ref: Pyramid, ref_slot: int, cur: Pyramid, cur_slot: int, cam, T_ref_w, T_cur_w,
                        seeds: SeedBatch, prm: Optional[CDfParams] = None, lo: int = 0, hi: Optional[int] = None):
    """DepthFilter::updateSeeds for seeds [lo, hi) of a device-resident batch (asynchronous)."""
    prm = prm or depth_filter_params()
    hi = seeds.n if hi is None else hi
    c = make_camera(cam)
    Tr, Tc = _f64(T_ref_w), _f64(T_cur_w)

    def off(d, per):
        return C.c_void_p(d.ptr + lo * per)
    ctx.check(ctx.lib.svo_hip_depth_filter_update_dev(
        ctx.h, ref.h, ref_slot, cur.h, cur_slot, C.byref(c), _ptr(Tr, C.c_double), _ptr(Tc, C.c_double), hi - lo,
        off(seeds.px, 16), off(seeds.f, 24), off(seeds.level, 4), off(seeds.a, 4), off(seeds.b, 4), off(seeds.mu, 4),
        off(seeds.z_range, 4), off(seeds.sigma2, 4), C.byref(prm), off(seeds.status, 4), off(seeds.z, 8),
        off(seeds.xyz, 24), off(seeds.n_zmssd, 4), off(seeds.n_align, 4), off(seeds.px_cur, 16),
        off(seeds.search_level, 4)), "depth_filter_update")


def epipolar_match_batch(ctx: Context, ref: Pyramid, ref_slot: int, cur: Pyramid, cur_slot: int, cam, T_ref_w, T_cur_w,
                         px, f, level, d_est, d_min, d_max, prm: Optional[CDfParams] = None):
    """Matcher::findEpipolarMatchDirect over n reference features with explicit depth intervals; returns a dict of
    ok, depth, px_cur, search_level, epi_length, n_zmssd, n_align_iters."""
    prm = prm or depth_filter_params()
    n = len(px)
    c = make_camera(cam)
    Tr, Tc = _f64(T_ref_w), _f64(T_cur_w)
    d_px, d_f = ctx.to_device(_f64(px)), ctx.to_device(_f64(f))
    d_lvl = ctx.to_device(np.ascontiguousarray(level, dtype=np.int32))
    d_dep = ctx.to_device(_f64(np.stack([d_est, d_min, d_max])))
    o = {"ok": ctx.empty((n,), np.uint8), "depth": ctx.empty((n,), np.float64), "px_cur": ctx.empty((n, 2), np.float64),
         "search_level": ctx.empty((n,), np.int32), "epi_length": ctx.empty((n,), np.float64),
         "n_zmssd": ctx.empty((n,), np.int32), "n_align_iters": ctx.empty((n,), np.int32)}
    ctx.check(ctx.lib.svo_hip_epipolar_match_batch_dev(
        ctx.h, ref.h, ref_slot, cur.h, cur_slot, C.byref(c), _ptr(Tr, C.c_double), _ptr(Tc, C.c_double), n,
        C.c_void_p(d_px.ptr), C.c_void_p(d_f.ptr), C.c_void_p(d_lvl.ptr), C.c_void_p(d_dep.ptr), C.byref(prm),
        C.c_void_p(o["ok"].ptr), C.c_void_p(o["depth"].ptr), C.c_void_p(o["px_cur"].ptr), C.c_void_p(o["search_level"].ptr),
        C.c_void_p(o["epi_length"].ptr), C.c_void_p(o["n_zmssd"].ptr), C.c_void_p(o["n_align_iters"].ptr)),
        "epipolar_match_batch")
    out = {k: v.download() for k, v in o.items()}
    out["ok"] = out["ok"].astype(bool)
    for v in list(o.values()) + [d_px, d_f, d_lvl, d_dep]:
        v.free()
    return out


# ---- next rows f-4: pose_optimizer::optimizeGaussNewton, Point::optimize -----------------------------------------
class CPoseOptResult(C.Structure):
    _fields_ = [("ran", C.c_int32), ("n_iter_done", C.c_int32), ("n_deleted", C.c_int32), ("pad_", C.c_int32),
                ("num_obs", C.c_uint64), ("T_f_w", C.c_double * 7), ("estimated_scale", C.c_double),
                ("error_init", C.c_double), ("error_final", C.c_double), ("Cov", C.c_double * 36)]


def pose_optimize(ctx: Context, T_f_w, f, pos, level, has_point, error_multiplier2: float, reproj_thresh: float = 2.0,
                  n_iter: int = 10):
    """pose_optimizer::optimizeGaussNewton for one frame (host buffers).  Returns (CPoseOptResult, has_point after
    the outlier test)."""
    T, ff, pp = _f64(T_f_w), _f64(f), _f64(pos)
    lv = np.ascontiguousarray(level, dtype=np.int32)
    hp = np.ascontiguousarray(has_point, dtype=np.uint8).copy()
    res = CPoseOptResult()
    ctx.check(ctx.lib.svo_hip_pose_optimize(ctx.h, len(lv), _ptr(T, C.c_double), _ptr(ff, C.c_double), _ptr(pp, C.c_double),
                                            _ptr(lv, C.c_int32), _ptr(hp, C.c_uint8), C.c_double(error_multiplier2),
                                            C.c_double(reproj_thresh), n_iter, C.byref(res)), "pose_optimize")
    return res, hp


def pose_optimize_batch(ctx: Context, T_f_w, f, pos, level, has_point, n_feat, error_multiplier2: float,
                        reproj_thresh: float = 2.0, n_iter: int = 10):
    """Batch form: T_f_w [B,7], f/pos [B,max_n,3], level/has_point [B,max_n], n_feat [B].  Returns (list of
    CPoseOptResult, has_point [B,max_n])."""
    T, ff, pp = _f64(T_f_w), _f64(f), _f64(pos)
    B, max_n = ff.shape[0], ff.shape[1]
    d = [ctx.to_device(T), ctx.to_device(ff), ctx.to_device(pp), ctx.to_device(np.ascontiguousarray(level, dtype=np.int32)),
         ctx.to_device(np.ascontiguousarray(has_point, dtype=np.uint8)), ctx.to_device(np.ascontiguousarray(n_feat, dtype=np.int32))]
    dres = ctx.empty((B * C.sizeof(CPoseOptResult),), np.uint8)
    ctx.check(ctx.lib.svo_hip_pose_optimize_batch_dev(
        ctx.h, B, max_n, C.c_void_p(d[5].ptr), C.c_void_p(d[0].ptr), C.c_void_p(d[1].ptr), C.c_void_p(d[2].ptr),
        C.c_void_p(d[3].ptr), C.c_void_p(d[4].ptr), C.c_double(error_multiplier2), C.c_double(reproj_thresh), n_iter,
        C.c_void_p(dres.ptr)), "pose_optimize_batch")
    raw = dres.download()
    hp = d[4].download().reshape(B, max_n)
    res = [CPoseOptResult.from_buffer_copy(raw[i * C.sizeof(CPoseOptResult):(i + 1) * C.sizeof(CPoseOptResult)].tobytes())
           for i in range(B)]
    for v in d + [dres]:
        v.free()
    return res, hp


def point_optimize_batch(ctx: Context, pos, obs_offset, obs_T, obs_f, n_iter: int = 5):
    """Point::optimize for a batch of map points; observations in CSR form.  Returns (pos [n,3], iters [n])."""
    n = len(pos)
    dp = ctx.to_device(_f64(pos))
    do = ctx.to_device(np.ascontiguousarray(obs_offset, dtype=np.int32))
    dT, dF = ctx.to_device(_f64(obs_T)), ctx.to_device(_f64(obs_f))
    di = ctx.empty((max(n, 1),), np.int32)
    ctx.check(ctx.lib.svo_hip_point_optimize_batch_dev(ctx.h, n, n_iter, C.c_void_p(dp.ptr), C.c_void_p(do.ptr),
                                                       C.c_void_p(dT.ptr), C.c_void_p(dF.ptr), C.c_void_p(di.ptr)),
              "point_optimize_batch")
    out, it = dp.download().reshape(n, 3), di.download()[:n]
    for v in (dp, do, dT, dF, di):
        v.free()
    return out, it


def ldlt6_solve_batch(ctx: Context, H, b):
    Hm, bv = _f64(np.asarray(H).reshape(-1, 36)), _f64(np.asarray(b).reshape(-1, 6))
    x = np.zeros_like(bv)
    ctx.check(ctx.lib.svo_hip_ldlt6_solve_batch(ctx.h, len(Hm), _ptr(Hm, C.c_double), _ptr(bv, C.c_double),
                                                _ptr(x, C.c_double)), "ldlt6_solve_batch")
    return x


# ---- next row f-3: FastDetector::detect, Seed::Seed ---------------------------------------------------------------
def detect_grid(width: int, height: int, cell_size: int):
    gc, gr = C.c_int(0), C.c_int(0)
    load_library().svo_hip_detect_grid(width, height, cell_size, C.byref(gc), C.byref(gr))
    return gc.value, gr.value


def detect_features(ctx: Context, pyr: Pyramid, slot: int, cam=None, n_pyr_levels: int = 3, cell_size: int = 20,
                    occupancy=None, detection_threshold: float = 10.0):
    """FastDetector::detect on one pyramid slot.  Returns px [n,2] f64 (level-0 pixel), f [n,3] (when cam is given),
    level [n] i32, score [n] f32 in grid-cell order."""
    gc, gr = detect_grid(pyr.width, pyr.height, cell_size)
    nc = gc * gr
    px, f = np.zeros((nc, 2)), np.zeros((nc, 3))
    lvl, sc = np.zeros(nc, dtype=np.int32), np.zeros(nc, dtype=np.float32)
    n = C.c_int32(0)
    occ = None if occupancy is None else np.ascontiguousarray(occupancy, dtype=np.uint8)
    c = make_camera(cam) if cam is not None else None
    ctx.check(ctx.lib.svo_hip_detect_features(
        ctx.h, pyr.h, slot, C.byref(c) if c is not None else None, n_pyr_levels, cell_size,
        None if occ is None else _ptr(occ, C.c_uint8), C.c_double(detection_threshold), C.byref(n), _ptr(px, C.c_double),
        _ptr(f, C.c_double) if c is not None else None, _ptr(lvl, C.c_int32), _ptr(sc, C.c_float)), "detect_features")
    k = n.value
    return px[:k].copy(), (f[:k].copy() if c is not None else None), lvl[:k].copy(), sc[:k].copy()


def seed_init_batch(ctx: Context, n: int, depth_mean: float, depth_min: float):
    """Seed::Seed for n seeds; returns (a, b, mu, z_range, sigma2) f32 arrays."""
    d = [ctx.empty((max(n, 1),), np.float32) for _ in range(5)]
    ctx.check(ctx.lib.svo_hip_seed_init_batch_dev(ctx.h, n, C.c_double(depth_mean), C.c_double(depth_min),
                                                  *[C.c_void_p(v.ptr) for v in d]), "seed_init_batch")
    out = tuple(v.download()[:n] for v in d)
    for v in d:
        v.free()
    return out


# ---- next row f-2: the cell loop of Reprojector::reprojectMap --------------------------------------------------------
def reproject_cells(ctx: Context, ref: Pyramid, cur: Pyramid, cur_slot: int, cam, T_kf_w, T_cur_w, cell_offset, kf_slot,
                    px_ref, f_ref, level_ref, pt_pos, deleted, px_cur, edgelet=None, grad=None, max_fts: int = 1200,
                    n_pyr_levels: int = 3, align_max_iter: int = 10):
    """svo_hip_reproject_cells: candidates bucketed per cell in trial order; returns dict(tried, matched, search_level,
    cell_winner, px_cur, n_matches, n_trials)."""
    c = make_camera(cam)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    co, ks, lr = i32(cell_offset), i32(kf_slot), i32(level_ref)
    n_cells, n = len(co) - 1, int(co[-1])
    Tk, Tc, pr, fr, pp = _f64(T_kf_w), _f64(T_cur_w), _f64(px_ref), _f64(f_ref), _f64(pt_pos)
    pc = _f64(px_cur).copy()
    de = np.ascontiguousarray(deleted, dtype=np.uint8)
    ed = None if edgelet is None else np.ascontiguousarray(edgelet, dtype=np.uint8)
    gr = None if grad is None else _f64(grad)
    tried, matched = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
    sl, win = np.zeros(max(n, 1), np.int32), np.zeros(max(n_cells, 1), np.int32)
    nm, nt = C.c_uint64(0), C.c_uint64(0)
    ctx.check(ctx.lib.svo_hip_reproject_cells(
        ctx.h, ref.h, cur.h, cur_slot, C.byref(c), len(Tk), _ptr(Tk, C.c_double), _ptr(Tc, C.c_double), n_cells,
        _ptr(co, C.c_int32), _ptr(ks, C.c_int32), _ptr(pr, C.c_double), _ptr(fr, C.c_double), _ptr(lr, C.c_int32),
        _ptr(pp, C.c_double), None if ed is None else _ptr(ed, C.c_uint8), None if gr is None else _ptr(gr, C.c_double),
        _ptr(de, C.c_uint8), _ptr(pc, C.c_double), max_fts, n_pyr_levels, align_max_iter, _ptr(tried, C.c_uint8),
        _ptr(matched, C.c_uint8), _ptr(sl, C.c_int32), _ptr(win, C.c_int32), C.byref(nm), C.byref(nt)), "reproject_cells")
    return {"tried": tried[:n], "matched": matched[:n], "search_level": sl[:n], "cell_winner": win[:n_cells], "px_cur": pc,
            "n_matches": nm.value, "n_trials": nt.value}


# ---- one tracked frame on one stream (svo_hip_tracker_*: FrameHandlerMono::processFrame up to the pose refinement) -----
class CTrackerConfig(C.Structure):
    _fields_ = [("max_keyframes", C.c_int), ("max_points", C.c_int), ("max_obs", C.c_int), ("max_kf_features", C.c_int),
                ("max_candidates", C.c_int), ("max_items", C.c_int), ("max_frame_features", C.c_int), ("n_levels", C.c_int),
                ("klt_max_level", C.c_int), ("klt_min_level", C.c_int), ("sia_n_iter", C.c_int), ("sia_eps", C.c_double),
                ("grid_size", C.c_int), ("max_fts", C.c_int), ("quality_min_fts", C.c_int), ("reproj_max_n_kfs", C.c_int),
                ("n_pyr_levels", C.c_int), ("align_max_iter", C.c_int), ("pose_optim_thresh", C.c_double),
                ("pose_optim_num_iter", C.c_int)]


class CTrackerMap(C.Structure):
    _fields_ = [("n_kf", C.c_int), ("kf_slot", C.POINTER(C.c_int32)), ("T_kf_w", C.POINTER(C.c_double)),
                ("kf_key_point", C.POINTER(C.c_int32)), ("kf_ftr_offset", C.POINTER(C.c_int32)), ("kf_ftr_point", C.POINTER(C.c_int32)),
                ("n_points", C.c_int), ("pt_pos", C.POINTER(C.c_double)), ("pt_type", C.POINTER(C.c_int32)),
                ("pt_n_failed", C.POINTER(C.c_int32)), ("pt_n_succeeded", C.POINTER(C.c_int32)), ("pt_obs_offset", C.POINTER(C.c_int32)),
                ("obs_kf", C.POINTER(C.c_int32)), ("obs_px", C.POINTER(C.c_double)), ("obs_f", C.POINTER(C.c_double)),
                ("obs_level", C.POINTER(C.c_int32)), ("obs_edgelet", C.POINTER(C.c_uint8)), ("obs_grad", C.POINTER(C.c_double)),
                ("n_candidates", C.c_int), ("cand_point", C.POINTER(C.c_int32))]


class CTrackResult(C.Structure):
    _fields_ = [("T_f_w", C.c_double * 7), ("T_f_w_sia", C.c_double * 7), ("sia_n_tracked", C.c_uint64),
                ("sia_iters", C.c_int32 * MAX_LEVELS), ("sia_stop", C.c_int32), ("n_features", C.c_int32), ("n_matches", C.c_uint64),
                ("n_trials", C.c_uint64), ("n_overlap", C.c_int32), ("map_changed", C.c_int32), ("overlap_kf", C.c_int32 * 16),
                ("overlap_count", C.c_int32 * 16), ("n_candidates", C.c_int32), ("items_overflow", C.c_int32), ("pose", CPoseOptResult)]


class Tracker:
    """svo_hip_tracker: SparseImgAlign -> Reprojector::reprojectMap -> pose_optimizer on one stream, one synchronisation per
    frame, the frame's matches handed over to the next call on the device."""

    def __init__(self, ctx: Context, cam, _group_handle=None, _cfg=None, **overrides):
        self.ctx, self.cam = ctx, cam
        self.ccam = make_camera(cam)
        if _group_handle is not None:              # a camera of a TrackerGroup: the handle belongs to the group
            self.cfg, self.h, self._in_group = _cfg, _group_handle, True
        else:
            self.cfg = CTrackerConfig()
            ctx.check(ctx.lib.svo_hip_tracker_default_config(C.byref(self.cfg)), "tracker_default_config")
            for k, v in overrides.items():
                assert hasattr(self.cfg, k), k
                setattr(self.cfg, k, v)
            self.h = C.c_void_p()
            self._in_group = False
            ctx.check(ctx.lib.svo_hip_tracker_create(ctx.h, C.byref(self.ccam), C.byref(self.cfg), C.byref(self.h)), "tracker_create")
        nc, gc, gr = C.c_int(), C.c_int(), C.c_int()
        ctx.check(ctx.lib.svo_hip_tracker_info(self.h, C.byref(nc), C.byref(gc), C.byref(gr), None), "tracker_info")
        self.n_cells, self.grid_cols, self.grid_rows = nc.value, gc.value, gr.value
        self.n_points = 0

    def upload_keyframe(self, slot: int, img: np.ndarray):
        im = np.ascontiguousarray(img, dtype=np.uint8)
        assert im.shape == (self.cam.height, self.cam.width)
        self.ctx.check(self.ctx.lib.svo_hip_tracker_upload_keyframe(self.h, slot, _ptr(im, C.c_uint8)), "tracker_upload_keyframe")
        self.ctx.sync()

    def keyframe_from_last_frame(self, slot: int):
        self.ctx.check(self.ctx.lib.svo_hip_tracker_keyframe_from_last_frame(self.h, slot), "tracker_keyframe_from_last_frame")

    def optimize_structure(self, point, n_iter: int = 5):
        """FrameHandlerBase::optimizeStructure on the given points over the observations the map tables hold.
        Returns (pos [n,3], iters [n])."""
        pt = np.ascontiguousarray(point, dtype=np.int32)
        pos, it = np.zeros((max(len(pt), 1), 3)), np.zeros(max(len(pt), 1), np.int32)
        self.ctx.check(self.ctx.lib.svo_hip_tracker_optimize_structure(self.h, len(pt), _ptr(pt, C.c_int32), int(n_iter), _ptr(pos, C.c_double),
                                                                       _ptr(it, C.c_int32)), "tracker_optimize_structure")
        return pos[:len(pt)], it[:len(pt)]

    def image_buffer(self) -> np.ndarray:
        """the tracker's page-locked image buffer as a (height, width) u8 view (svo_hip_tracker_image_buffer): an image written
        there and passed to track() as this very array is not copied again"""
        p = C.POINTER(C.c_uint8)()
        self.ctx.check(self.ctx.lib.svo_hip_tracker_image_buffer(self.h, C.byref(p)), "tracker_image_buffer")
        return np.ctypeslib.as_array(p, shape=(self.cam.height, self.cam.width))

    def download_key_points(self, n_kf: int) -> np.ndarray:
        """[n_kf][5] point indices of the keyframes' key features as the device holds them (after the re-selections that follow
        deletions)"""
        out = np.full((max(n_kf, 1), 5), -1, dtype=np.int32)
        self.ctx.check(self.ctx.lib.svo_hip_tracker_download_key_points(self.h, _ptr(out, C.c_int32)), "tracker_download_key_points")
        return out[:n_kf]

    def set_map(self, mp: dict):
        """mp: the index tables of android_svo_amd.synth.make_map_case (plus kf_slot, kf_key_point)"""
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        a = dict(ks=i32(mp["kf_slot"]), Tk=_f64(mp["T_kf_w"]), key=i32(mp["kf_key_point"]), ko=i32(mp["kf_ftr_offset"]), kp=i32(mp["kf_ftr_point"]),
                 pos=_f64(mp["pt_pos"]), ty=i32(mp["pt_type"]), nf=i32(mp["pt_n_failed"]), ns=i32(mp["pt_n_succeeded"]), oo=i32(mp["pt_obs_offset"]),
                 ok=i32(mp["obs_kf"]), opx=_f64(mp["obs_px"]), of=_f64(mp["obs_f"]), ol=i32(mp["obs_level"]), cp=i32(mp["cand_point"]))
        ed = np.ascontiguousarray(mp["obs_edgelet"], dtype=np.uint8) if mp.get("obs_edgelet") is not None else None
        gr = _f64(mp["obs_grad"]) if mp.get("obs_grad") is not None else None
        I, D = C.c_int32, C.c_double
        m = CTrackerMap(len(a["ks"]), _ptr(a["ks"], I), _ptr(a["Tk"], D), _ptr(a["key"], I), _ptr(a["ko"], I), _ptr(a["kp"], I), len(a["ty"]),
                        _ptr(a["pos"], D), _ptr(a["ty"], I), _ptr(a["nf"], I), _ptr(a["ns"], I), _ptr(a["oo"], I), _ptr(a["ok"], I), _ptr(a["opx"], D),
                        _ptr(a["of"], D), _ptr(a["ol"], I), None if ed is None else _ptr(ed, C.c_uint8), None if gr is None else _ptr(gr, D),
                        len(a["cp"]), _ptr(a["cp"], I))
        self.ctx.check(self.ctx.lib.svo_hip_tracker_set_map(self.h, C.byref(m)), "tracker_set_map")
        self.ctx.sync()
        self.n_points = len(a["ty"])

    def update_point_positions(self, point, pos):
        pt, pp = np.ascontiguousarray(point, dtype=np.int32), _f64(pos)
        self.ctx.check(self.ctx.lib.svo_hip_tracker_update_point_positions(self.h, len(pt), _ptr(pt, C.c_int32), _ptr(pp, C.c_double)),
                       "tracker_update_point_positions")
        self.ctx.sync()

    def set_last_frame(self, T_f_w, px, f, point, img: Optional[np.ndarray] = None, kf_slot: int = -1):
        T, p, ff = _f64(T_f_w), _f64(px), _f64(f)
        pt = np.ascontiguousarray(point, dtype=np.int32)
        im = None if img is None else np.ascontiguousarray(img, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.svo_hip_tracker_set_last_frame(self.h, None if im is None else _ptr(im, C.c_uint8), kf_slot, _ptr(T, C.c_double),
                                                                   len(pt), _ptr(p, C.c_double), _ptr(ff, C.c_double), _ptr(pt, C.c_int32)),
                       "tracker_set_last_frame")
        self.ctx.sync()

    def _outputs(self):
        nf = self.cfg.max_frame_features
        if not hasattr(self, "_out") or len(self._out["pt_type"]) != max(self.n_points, 1):
            self._out = dict(px=np.zeros((nf, 2)), f=np.zeros((nf, 3)), level=np.zeros(nf, np.int32), point=np.zeros(nf, np.int32),
                             edgelet=np.zeros(nf, np.uint8), grad=np.zeros((nf, 2)), pt_type=np.zeros(max(self.n_points, 1), np.int32),
                             pt_n_failed=np.zeros(max(self.n_points, 1), np.int32), pt_n_succeeded=np.zeros(max(self.n_points, 1), np.int32))
        return self._out

    def track(self, img: np.ndarray, want_points: bool = True) -> dict:
        im = np.ascontiguousarray(img, dtype=np.uint8)
        assert im.shape == (self.cam.height, self.cam.width)
        o = self._outputs()
        res = CTrackResult()
        I, D = C.c_int32, C.c_double
        self.ctx.check(self.ctx.lib.svo_hip_tracker_track(
            self.h, _ptr(im, C.c_uint8), C.byref(res), _ptr(o["px"], D), _ptr(o["f"], D), _ptr(o["level"], I), _ptr(o["point"], I),
            _ptr(o["edgelet"], C.c_uint8), _ptr(o["grad"], D), _ptr(o["pt_type"], I) if want_points else None,
            _ptr(o["pt_n_failed"], I) if want_points else None, _ptr(o["pt_n_succeeded"], I) if want_points else None), "tracker_track")
        return self._as_dict(res, o)

    def last_result(self) -> dict:
        """svo_hip_tracker_last_result: the outcome of the camera's last tracked frame (what track() returns), read again from its
        page-locked result block -- how the cameras of a TrackerGroup hand out their features and point counters"""
        o = self._outputs()
        res = CTrackResult()
        I, D = C.c_int32, C.c_double
        self.ctx.check(self.ctx.lib.svo_hip_tracker_last_result(
            self.h, C.byref(res), _ptr(o["px"], D), _ptr(o["f"], D), _ptr(o["level"], I), _ptr(o["point"], I), _ptr(o["edgelet"], C.c_uint8),
            _ptr(o["grad"], D), _ptr(o["pt_type"], I), _ptr(o["pt_n_failed"], I), _ptr(o["pt_n_succeeded"], I)), "tracker_last_result")
        return self._as_dict(res, o)

    def _as_dict(self, res, o) -> dict:
        n = res.n_features
        return {"result": res, "T_f_w": np.array(res.T_f_w), "T_f_w_sia": np.array(res.T_f_w_sia), "n_matches": int(res.n_matches),
                "n_trials": int(res.n_trials), "feat_px": o["px"][:n].copy(), "feat_f": o["f"][:n].copy(), "feat_level": o["level"][:n].copy(),
                "feat_point": o["point"][:n].copy(), "feat_type": o["edgelet"][:n].astype(np.int32), "feat_grad": o["grad"][:n].copy(),
                "overlap_kf": np.array(res.overlap_kf[:res.n_overlap]), "overlap_count": np.array(res.overlap_count[:res.n_overlap]),
                "type": o["pt_type"][:self.n_points].copy(), "n_failed": o["pt_n_failed"][:self.n_points].copy(),
                "n_succeeded": o["pt_n_succeeded"][:self.n_points].copy(), "map_changed": int(res.map_changed)}

    def destroy(self):
        if self.h and not self._in_group:
            self.ctx.lib.svo_hip_tracker_destroy(self.h)
        self.h = C.c_void_p()


class TrackerGroup:
    """svo_hip_tracker_group: n cameras (one camera model, one configuration) tracked together, one chain of launches per call.
    `cameras[c]` is a Tracker bound to camera c's handle: set_map / set_last_frame / upload_keyframe / optimize_structure /
    image_buffer / last_result work on it; track() and destroy() are the group's."""

    def __init__(self, ctx: Context, cam, n_cameras: int, **overrides):
        self.ctx, self.cam, self.n = ctx, cam, int(n_cameras)
        self.cfg = CTrackerConfig()
        ctx.check(ctx.lib.svo_hip_tracker_default_config(C.byref(self.cfg)), "tracker_default_config")
        for k, v in overrides.items():
            assert hasattr(self.cfg, k), k
            setattr(self.cfg, k, v)
        self.ccam = make_camera(cam)
        self.h = C.c_void_p()
        ctx.check(ctx.lib.svo_hip_tracker_group_create(ctx.h, C.byref(self.ccam), C.byref(self.cfg), self.n, C.byref(self.h)), "tracker_group_create")
        self.cameras = []
        for c in range(self.n):
            th = C.c_void_p()
            ctx.check(ctx.lib.svo_hip_tracker_group_camera(self.h, c, C.byref(th)), "tracker_group_camera")
            self.cameras.append(Tracker(ctx, cam, _group_handle=th, _cfg=self.cfg))
        self._ptrs = (C.POINTER(C.c_uint8) * self.n)()
        self._res = (CTrackResult * self.n)()

    def track(self, imgs) -> list:
        """one frame of every camera; imgs[c]: (height, width) u8 (camera c's image_buffer() array is not copied).  Returns the
        cameras' CTrackResult records (features and point counters: cameras[c].last_result())."""
        keep = []
        for c in range(self.n):
            im = np.ascontiguousarray(imgs[c], dtype=np.uint8)
            assert im.shape == (self.cam.height, self.cam.width)
            keep.append(im)
            self._ptrs[c] = im.ctypes.data_as(C.POINTER(C.c_uint8))
        self.ctx.check(self.ctx.lib.svo_hip_tracker_group_track(self.h, self._ptrs, self._res), "tracker_group_track")
        return [self._res[c] for c in range(self.n)]

    def destroy(self):
        if self.h:
            for t in self.cameras:
                t.h = C.c_void_p()
            self.ctx.lib.svo_hip_tracker_group_destroy(self.h)
            self.h = C.c_void_p()
