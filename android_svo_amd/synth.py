"""Deterministic synthetic inputs for the SVO hot path.

A textured plane is rendered analytically for a reference pose and a current
pose; pyramids use the truncating 2x2 mean (the arm64 / scalar form of
vk::halfSample, reference vision.cpp:89-110) and are treated as *inputs* by
every consumer (oracle, HIP path, reference harness), so the SSE2/NEON rounding
split of the reference never enters a comparison.

Everything here is numpy on the host; nothing is timed.
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Tuple

import numpy as np

N_LEVELS = 5


@dataclasses.dataclass
class Camera:
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float

    @staticmethod
    def default(width: int = 640, height: int = 480) -> "Camera":
        f = 500.0 if width <= 640 else 1000.0
        return Camera(width, height, f, f, width / 2 - 0.5, height / 2 - 0.5)


# ---- SE3 helpers in the reference's storage order [t(3), q(xyzw)] ---------------

def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz])


def quat_rot(q, p):
    qv = np.asarray(q[:3])
    uv = 2.0 * np.cross(qv, p)
    return p + q[3] * uv + np.cross(qv, uv)


def se3_from_twist(t, w):
    """Pose with translation t and rotation exp(w) (plain axis-angle; generator only)."""
    w = np.asarray(w, dtype=np.float64)
    th = float(np.linalg.norm(w))
    if th < 1e-12:
        q = np.array([0.0, 0.0, 0.0, 1.0])
    else:
        q = np.concatenate([math.sin(th / 2) * w / th, [math.cos(th / 2)]])
    return np.concatenate([np.asarray(t, dtype=np.float64), q])


def se3_inv(T):
    qi = np.array([-T[3], -T[4], -T[5], T[6]])
    return np.concatenate([-quat_rot(qi, T[:3]), qi])


def se3_mul(A, B):
    return np.concatenate([A[:3] + quat_rot(A[3:], B[:3]), quat_mul(A[3:], B[3:])])


def se3_act(T, p):
    return T[:3] + quat_rot(T[3:], p)


def rot_matrix(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def pose_error(Ta, Tb) -> Tuple[float, float]:
    """(rotation error rad, translation error m) of E = Ta * Tb^-1."""
    E = se3_mul(np.asarray(Ta, dtype=np.float64), se3_inv(np.asarray(Tb, dtype=np.float64)))
    n = float(np.linalg.norm(E[3:6]))
    w = float(E[6])
    ang = 2.0 * math.atan2(n, abs(w))
    return ang, float(np.linalg.norm(E[:3]))


# ---- scene ------------------------------------------------------------------------

class PlaneScene:
    """Plane n.X = d in the world frame with a band-limited texture."""

    def __init__(self, seed: int = 12345, depth: float = 2.0, tilt=(0.08, -0.05)):
        rng = np.random.default_rng(seed)
        n = np.array([tilt[0], tilt[1], 1.0])
        self.n = n / np.linalg.norm(n)
        self.d = depth * self.n[2]
        # in-plane basis
        e1 = np.cross([0.0, 1.0, 0.0], self.n)
        self.e1 = e1 / np.linalg.norm(e1)
        self.e2 = np.cross(self.n, self.e1)
        k = 6
        self.freq = rng.uniform(4.0, 28.0, size=(k, 2)) * rng.choice([-1.0, 1.0], size=(k, 2))
        self.phase = rng.uniform(0, 2 * math.pi, size=k)
        self.amp = rng.uniform(0.5, 1.0, size=k)
        self.grid = rng.uniform(-1.0, 1.0, size=(96, 96))
        self.grid_scale = 18.0  # cells per metre

    def texture(self, s, t):
        val = np.zeros_like(s)
        for (fa, fb), ph, am in zip(self.freq, self.phase, self.amp):
            val += am * np.sin(fa * s + fb * t + ph)
        gs = (s * self.grid_scale) % 95.0
        gt = (t * self.grid_scale) % 95.0
        i0 = np.floor(gs).astype(np.int64)
        j0 = np.floor(gt).astype(np.int64)
        a = gs - i0
        b = gt - j0
        g = self.grid
        noise = ((1 - a) * (1 - b) * g[j0, i0] + a * (1 - b) * g[j0, i0 + 1]
                 + (1 - a) * b * g[j0 + 1, i0] + a * b * g[j0 + 1, i0 + 1])
        val = val / np.sum(self.amp) * 0.55 + noise * 0.45
        return val

    def intersect(self, cam: Camera, T_f_w, u, v):
        """World points hit by pixel rays (u, v) of a camera at pose T_f_w."""
        T_w_f = se3_inv(np.asarray(T_f_w, dtype=np.float64))
        R = rot_matrix(T_w_f[3:])
        o = T_w_f[:3]
        x = (u - cam.cx) / cam.fx
        y = (v - cam.cy) / cam.fy
        if getattr(cam, "dist", None) is not None and any(cam.dist):
            x, y = undistort_normalized(cam.dist, x, y)      # radtan camera: the ray of a DISTORTED pixel
        dirs = np.stack([x, y, np.ones_like(x)], axis=-1) @ R.T
        lam = (self.d - o @ self.n) / (dirs @ self.n)
        return o + dirs * lam[..., None]

    def render(self, cam: Camera, T_f_w) -> np.ndarray:
        v, u = np.mgrid[0:cam.height, 0:cam.width].astype(np.float64)
        X = self.intersect(cam, T_f_w, u, v)
        s = X @ self.e1
        t = X @ self.e2
        val = self.texture(s, t)
        img = np.clip(np.rint(127.5 + 110.0 * val), 0, 255).astype(np.uint8)
        return np.ascontiguousarray(img)


def undistort_normalized(dist, xd, yd, iters: int = 40):
    """Inverse of the 5-coefficient radtan model vk::PinholeCamera::world2cam applies (S/pinhole_camera.cpp:88-104; d =
    k1, k2, p1, p2, k3) on normalised coordinates, by fixed-point iteration in f64 to convergence -- the generator's
    ground truth for images of a distorted camera (the reference itself inverts with cv::undistortPoints: five float
    iterations)."""
    k1, k2, p1, p2, k3 = (float(c) for c in dist)
    xd, yd = np.asarray(xd, dtype=np.float64), np.asarray(yd, dtype=np.float64)
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        cdist = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
        dx = p1 * 2.0 * x * y + p2 * (r2 + 2.0 * x * x)
        dy = p1 * (r2 + 2.0 * y * y) + p2 * 2.0 * x * y
        x, y = (xd - dx) / cdist, (yd - dy) / cdist
    return x, y


def half_sample(img: np.ndarray) -> np.ndarray:
    h, w = img.shape
    a = img[0:h - h % 2:2, 0:w - w % 2:2].astype(np.uint16)
    b = img[0:h - h % 2:2, 1:w:2].astype(np.uint16)
    c = img[1:h:2, 0:w - w % 2:2].astype(np.uint16)
    d = img[1:h:2, 1:w:2].astype(np.uint16)
    return np.ascontiguousarray(((a + b + c + d) // 4).astype(np.uint8))


def build_pyramid(img: np.ndarray, n_levels: int = N_LEVELS) -> List[np.ndarray]:
    pyr = [np.ascontiguousarray(img)]
    for _ in range(1, n_levels):
        pyr.append(half_sample(pyr[-1]))
    return pyr


def cam2world(cam: Camera, px: np.ndarray) -> np.ndarray:
    x = (px[:, 0] - cam.cx) / cam.fx
    y = (px[:, 1] - cam.cy) / cam.fy
    z = np.ones_like(x)
    n = np.sqrt(x * x + y * y + z * z)
    return np.stack([x / n, y / n, z / n], axis=1)


@dataclasses.dataclass
class FramePair:
    cam: Camera
    ref_pyr: List[np.ndarray]
    cur_pyr: List[np.ndarray]
    px: np.ndarray          # [n,2] f64 level-0 pixel
    f: np.ndarray           # [n,3] f64 unit bearing
    pos: np.ndarray         # [n,3] f64 world point
    has_point: np.ndarray   # [n] u8
    T_ref_w: np.ndarray     # [7]
    T_cur_w_true: np.ndarray
    T_cur_w_init: np.ndarray
    dist: object = None     # optional radtan coefficients (k1,k2,p1,p2,k3) of the forward model


def grid_features(cam: Camera, n_target: int, rng, border: int = 48) -> np.ndarray:
    w = cam.width - 2 * border
    h = cam.height - 2 * border
    step = math.sqrt(w * h / float(n_target))
    nx = max(1, int(math.ceil(w / step)))
    ny = max(1, int(math.ceil(n_target / nx)))
    while nx * ny < n_target:
        ny += 1
    xs = border + (np.arange(nx) + 0.0) * (w - 1.0) / max(nx, 1)
    ys = border + (np.arange(ny) + 0.0) * (h - 1.0) / max(ny, 1)
    gx, gy = np.meshgrid(xs, ys)
    px = np.stack([gx.ravel(), gy.ravel()], axis=1)[:n_target]
    px = np.floor(px) + rng.uniform(0.0, 1.0, size=px.shape)
    return np.ascontiguousarray(px)


def make_frame_pair(seed: int = 12345, width: int = 640, height: int = 480, n_features: int = 200,
                    depth: float = 2.0, null_point_every: int = 0,
                    t_mag: float = 0.03, r_mag: float = 0.01, border: int = 48) -> FramePair:
    rng = np.random.default_rng(seed)
    cam = Camera.default(width, height)
    scene = PlaneScene(seed=seed, depth=depth,
                       tilt=(rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1)))
    T_ref_w = se3_from_twist(rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.02, 0.02, 3))
    xi_t = rng.uniform(-t_mag, t_mag, 3)
    xi_r = rng.uniform(-r_mag, r_mag, 3)
    xi_r[np.abs(xi_r) < 1e-4] = 1e-3          # non-zero rotation
    T_cur_ref = se3_from_twist(xi_t, xi_r)
    T_cur_w = se3_mul(T_cur_ref, T_ref_w)
    ref_img = scene.render(cam, T_ref_w)
    cur_img = scene.render(cam, T_cur_w)
    px = grid_features(cam, n_features, rng, border=border)
    f = cam2world(cam, px)
    pos = scene.intersect(cam, T_ref_w, px[:, 0], px[:, 1])
    has_point = np.ones(len(px), dtype=np.uint8)
    if null_point_every:
        has_point[::null_point_every] = 0
    return FramePair(cam, build_pyramid(ref_img), build_pyramid(cur_img), px, np.ascontiguousarray(f),
                     np.ascontiguousarray(pos), has_point, T_ref_w, T_cur_w, T_ref_w.copy())


# ---- inputs for the two small refinements (pose_optimizer::optimizeGaussNewton, Point::optimize) ----
@dataclasses.dataclass
class PoseOptCase:
    cam: Camera
    T_f_w_init: np.ndarray   # [7] pose to refine (SparseImgAlign's output in the pipeline)
    T_f_w_true: np.ndarray
    f: np.ndarray            # [n,3] measured unit bearings in the frame (reprojection matches)
    pos: np.ndarray          # [n,3] map points
    level: np.ndarray        # [n] i32 pyramid level of the matched feature
    has_point: np.ndarray    # [n] u8
    outlier: np.ndarray      # [n] bool: measurement deliberately corrupted


def make_pose_opt_case(seed: int = 5, n: int = 400, px_noise: float = 0.3, outlier_frac: float = 0.08,
                       pose_err=(0.01, 0.004), null_every: int = 11) -> PoseOptCase:
    rng = np.random.default_rng(seed)
    cam = Camera.default()
    T_true = se3_from_twist(rng.uniform(-0.2, 0.2, 3), rng.uniform(-0.05, 0.05, 3))
    px = np.stack([rng.uniform(20, cam.width - 20, n), rng.uniform(20, cam.height - 20, n)], axis=1)
    depth = rng.uniform(1.0, 4.0, n)
    ray = np.stack([(px[:, 0] - cam.cx) / cam.fx, (px[:, 1] - cam.cy) / cam.fy, np.ones(n)], axis=1) * depth[:, None]
    Tinv = se3_inv(T_true)
    pos = np.stack([se3_act(Tinv, r) for r in ray])
    level = rng.choice([0, 0, 1, 2], n).astype(np.int32)
    meas = px + rng.normal(size=(n, 2)) * px_noise * (1 << level)[:, None]
    outlier = rng.uniform(size=n) < outlier_frac
    meas[outlier] += rng.uniform(8, 40, (int(outlier.sum()), 2)) * rng.choice([-1, 1], (int(outlier.sum()), 2))
    f = cam2world(cam, meas)
    has = np.ones(n, dtype=np.uint8)
    if null_every:
        has[::null_every] = 0
    T_init = se3_mul(se3_from_twist(rng.uniform(-1, 1, 3) * pose_err[0], rng.uniform(-1, 1, 3) * pose_err[1]), T_true)
    return PoseOptCase(cam, T_init, T_true, np.ascontiguousarray(f), np.ascontiguousarray(pos), level, has, outlier)


def make_point_opt_cases(seed: int = 5, n_points: int = 300):
    """Map points with 2..8 observations each (pose of the observing keyframe + measured bearing), CSR layout.
    Returns pos0 [n,3], obs_offset [n+1], obs_T [m,7], obs_f [m,3], pos_true [n,3], n_iter [n]."""
    rng = np.random.default_rng(seed)
    pos0, pos_true, off, Ts, fs, iters = [], [], [0], [], [], []
    for t in range(n_points):
        X = rng.uniform(-1, 1, 3) + [0, 0, 3.0]
        n_obs = int(rng.integers(2, 9))
        for _ in range(n_obs):
            T = se3_from_twist(rng.uniform(-0.4, 0.4, 3), rng.uniform(-0.1, 0.1, 3))
            p = se3_act(T, X)
            p = p / np.linalg.norm(p) + rng.normal(size=3) * 1e-3
            Ts.append(T)
            fs.append(p / np.linalg.norm(p))
        off.append(off[-1] + n_obs)
        pos_true.append(X)
        pos0.append(X + rng.normal(size=3) * (0.05 if t % 3 else 0.5))
        iters.append(5 if t % 2 else 20)
    return (np.array(pos0), np.array(off, dtype=np.int32), np.array(Ts), np.array(fs), np.array(pos_true),
            np.array(iters, dtype=np.int32))


# ---- input for the reprojection cell loop (Reprojector::reprojectMap's second half) ----
TYPE_DELETED, TYPE_CANDIDATE, TYPE_UNKNOWN, TYPE_GOOD = 0, 1, 2, 3


def make_reproject_case(seed: int = 21, width: int = 320, height: int = 240, n_points: int = 900, n_kf: int = 3, cell_size: int = 20):
    """Map points seen from n_kf keyframes, projected into a new frame and bucketed into grid cells in the order the
    reference meets them (keyframe by keyframe).  Returns a dict of arrays; `order` is the trial order inside the cells
    after the reference's stable sort by point type (GOOD > UNKNOWN > CANDIDATE > DELETED)."""
    rng = np.random.default_rng(seed)
    cam = Camera(width, height, 250.0 * width / 320, 250.0 * width / 320, width / 2 - 0.5, height / 2 - 0.5)
    scene = PlaneScene(seed=seed, depth=2.0, tilt=(0.1, -0.07))
    kf_poses = [se3_from_twist(rng.uniform(-0.1, 0.1, 3) + [0, 0, 0.2 * k], rng.uniform(-0.03, 0.03, 3)) for k in range(n_kf)]
    T_cur_w = se3_from_twist(rng.uniform(-0.05, 0.05, 3) + [0, 0, 0.5], rng.uniform(-0.02, 0.02, 3))
    kf_pyr = [build_pyramid(scene.render(cam, T)) for T in kf_poses]
    cur_pyr = build_pyramid(scene.render(cam, T_cur_w))
    slot = np.sort(rng.integers(0, n_kf, n_points)).astype(np.int32)          # keyframe by keyframe
    level = rng.choice([0, 0, 1, 2], n_points).astype(np.int32)
    px_ref = np.stack([rng.uniform(10, width - 10, n_points), rng.uniform(10, height - 10, n_points)], axis=1)
    f_ref = cam2world(cam, px_ref)
    pos = np.zeros((n_points, 3))
    for k in range(n_kf):
        m = slot == k
        pos[m] = scene.intersect(cam, kf_poses[k], px_ref[m, 0], px_ref[m, 1])
    R = rot_matrix(T_cur_w[3:])
    Xc = pos @ R.T + T_cur_w[:3]
    px_cur = np.stack([cam.fx * Xc[:, 0] / Xc[:, 2] + cam.cx, cam.fy * Xc[:, 1] / Xc[:, 2] + cam.cy], axis=1)
    px_cur += rng.uniform(-1.5, 1.5, (n_points, 2))
    bad = rng.uniform(size=n_points) < 0.3                                     # poor predictions: these candidates tend to fail
    px_cur[bad] += rng.uniform(5, 9, (int(bad.sum()), 2)) * rng.choice([-1, 1], (int(bad.sum()), 2))
    pxi = px_cur.astype(np.int64)                                              # Vector2d::cast<int>()
    inside = (pxi[:, 0] >= 8) & (pxi[:, 0] < width - 8) & (pxi[:, 1] >= 8) & (pxi[:, 1] < height - 8)   # isInFrame(px, 8)
    gc = -(-width // cell_size)
    gr = -(-height // cell_size)
    cell = (px_cur[:, 1] / cell_size).astype(np.int64) * gc + (px_cur[:, 0] / cell_size).astype(np.int64)
    ptype = rng.choice([TYPE_DELETED, TYPE_CANDIDATE, TYPE_UNKNOWN, TYPE_UNKNOWN, TYPE_GOOD], n_points).astype(np.int32)
    n_failed = np.where(ptype == TYPE_UNKNOWN, rng.integers(13, 17, n_points), rng.integers(28, 32, n_points)).astype(np.int32)
    n_succ = rng.integers(8, 12, n_points).astype(np.int32)
    idx = np.where(inside)[0]
    # raw cell contents in insertion order, then the reference's stable sort by type (descending)
    raw = [idx[cell[idx] == c] for c in range(gc * gr)]
    trial = [r[np.argsort(-ptype[r], kind="stable")] for r in raw]
    return dict(cam=cam, kf_pyr=kf_pyr, cur_pyr=cur_pyr, T_kf_w=np.stack(kf_poses), T_cur_w=T_cur_w, n_cells=gc * gr,
                raw=raw, trial=trial, slot=slot, level=level, px_ref=px_ref, f_ref=np.ascontiguousarray(f_ref), pos=pos,
                px_cur=px_cur, ptype=ptype, n_failed=n_failed, n_succeeded=n_succ)


def flatten_cells(case, cells):
    """CSR view of per-cell candidate index lists: (cell_offset, candidate ids)."""
    off = np.zeros(case["n_cells"] + 1, dtype=np.int32)
    for c, r in enumerate(cells):
        off[c + 1] = off[c] + len(r)
    ids = np.concatenate([r for r in cells]) if off[-1] else np.zeros(0, dtype=np.int64)
    return off, ids.astype(np.int64)


def make_map_case(seed: int = 31, width: int = 320, height: int = 240, n_kf: int = 14, n_points: int = 1400, n_candidates: int = 150,
                  cell_size: int = 20, edgelet_frac: float = 0.04, kf_step: float = 0.16):
    """A small svo::Map as index tables (what Reprojector::reprojectMap walks, S/reprojector.cpp:72-168): n_kf keyframes along
    a trajectory over a textured plane -- more than the ten the reprojector uses, some of them far from the current frame
    -- map points with one to several observations (Point::obs_ order: newest keyframe first, as addFrameRef pushes to
    the front), keyframe feature lists in a shuffled order, point candidates (one observation, in no feature list), a few
    deleted points that are still referenced, counters close to the deletion / promotion thresholds, a few edgelets.
    Observation o lies in keyframe obs_kf[o] and belongs to point obs_point[o]; the observations of a point are stored
    contiguously (pt_obs_offset)."""
    rng = np.random.default_rng(seed)
    cam = Camera(width, height, 250.0 * width / 320, 250.0 * width / 320, width / 2 - 0.5, height / 2 - 0.5)
    scene = PlaneScene(seed=seed, depth=2.0, tilt=(0.08, -0.05))
    kf_poses = []
    for k in range(n_kf):
        # a sideways trajectory: the far keyframes no longer see what the current frame sees
        t = np.array([kf_step * (k - n_kf // 2), 0.03 * np.sin(k), 0.02 * k]) + rng.uniform(-0.02, 0.02, 3)
        kf_poses.append(se3_from_twist(t, rng.uniform(-0.03, 0.03, 3)))
    T_cur_w = se3_mul(se3_from_twist(rng.uniform(-0.03, 0.03, 3), rng.uniform(-0.015, 0.015, 3)), kf_poses[n_kf // 2 + 1])
    kf_pyr = [build_pyramid(scene.render(cam, T)) for T in kf_poses]
    cur_pyr = build_pyramid(scene.render(cam, T_cur_w))
    n_all = n_points + n_candidates
    home = rng.integers(0, n_kf, n_all)
    px_home = np.stack([rng.uniform(8, width - 8, n_all), rng.uniform(8, height - 8, n_all)], axis=1)
    pos = np.zeros((n_all, 3))
    for k in range(n_kf):
        m = home == k
        if m.any():
            pos[m] = scene.intersect(cam, kf_poses[k], px_home[m, 0], px_home[m, 1])
    pos[:n_points] += rng.normal(0, 0.004, (n_points, 3))                       # map points are not exactly on the surface
    ptype = rng.choice([TYPE_UNKNOWN, TYPE_UNKNOWN, TYPE_GOOD, TYPE_GOOD, TYPE_DELETED], n_all, p=[0.3, 0.3, 0.19, 0.19, 0.02]).astype(np.int32)
    ptype[n_points:] = TYPE_CANDIDATE
    n_failed = np.where(ptype == TYPE_UNKNOWN, rng.integers(12, 17, n_all), rng.integers(24, 32, n_all)).astype(np.int32)
    n_succ = rng.integers(8, 12, n_all).astype(np.int32)
    obs_point, obs_kf, obs_px, obs_level = [], [], [], []
    pt_obs_offset = [0]
    for p in range(n_all):
        kfs_p = [int(home[p])]
        if p < n_points:
            for k in range(n_kf):
                if k != home[p] and abs(k - home[p]) <= 3 and rng.uniform() < 0.45:
                    kfs_p.append(k)
        for k in sorted(kfs_p, reverse=True):                                   # obs_.push_front: newest first
            if k == home[p]:
                px = px_home[p]
            else:
                Xc = se3_act(kf_poses[k], pos[p])
                px = np.array([cam.fx * Xc[0] / Xc[2] + cam.cx, cam.fy * Xc[1] / Xc[2] + cam.cy]) + rng.uniform(-0.3, 0.3, 2)
                if not (4 <= px[0] < width - 4 and 4 <= px[1] < height - 4):
                    continue
            obs_point.append(p); obs_kf.append(k); obs_px.append(px); obs_level.append(int(rng.choice([0, 0, 1, 2])))
        pt_obs_offset.append(len(obs_point))
    obs_point, obs_kf = np.array(obs_point, np.int32), np.array(obs_kf, np.int32)
    obs_px, obs_level = np.array(obs_px), np.array(obs_level, np.int32)
    obs_f = np.ascontiguousarray(cam2world(cam, obs_px))
    obs_edgelet = (rng.uniform(size=len(obs_point)) < edgelet_frac).astype(np.uint8)
    ang = rng.uniform(0, 2 * np.pi, len(obs_point))
    obs_grad = np.stack([np.cos(ang), np.sin(ang)], axis=1)
    obs_grad[obs_edgelet == 0] = [1.0, 0.0]
    # keyframe feature lists (fts_ order): the observations that lie in the keyframe, shuffled; candidates' features are in none
    kf_ftr_offset, kf_ftr_obs = [0], []
    for k in range(n_kf):
        o = np.where((obs_kf == k) & (obs_point < n_points))[0]
        rng.shuffle(o)
        kf_ftr_obs.extend(o.tolist())
        kf_ftr_offset.append(len(kf_ftr_obs))
    kf_ftr_obs = np.array(kf_ftr_obs, np.int32)
    cand_point = np.arange(n_points, n_all, dtype=np.int32)
    rng.shuffle(cand_point)
    cand_obs = np.array([pt_obs_offset[p] for p in cand_point], np.int32)
    return dict(cam=cam, cell_size=cell_size, kf_pyr=kf_pyr, cur_pyr=cur_pyr, T_kf_w=np.stack(kf_poses), T_cur_w=T_cur_w, n_kf=n_kf,
                n_points=n_all, pt_pos=pos, pt_type=ptype, pt_n_failed=n_failed, pt_n_succeeded=n_succ,
                pt_obs_offset=np.array(pt_obs_offset, np.int32), obs_point=obs_point, obs_kf=obs_kf, obs_px=obs_px, obs_f=obs_f,
                obs_level=obs_level, obs_edgelet=obs_edgelet, obs_grad=obs_grad, kf_ftr_offset=np.array(kf_ftr_offset, np.int32),
                kf_ftr_obs=kf_ftr_obs, kf_ftr_point=obs_point[kf_ftr_obs].astype(np.int32), cand_point=cand_point, cand_obs=cand_obs)


def key_points(cam: Camera, px: np.ndarray, has_point: np.ndarray) -> np.ndarray:
    """Frame::setKeyPoints on an empty key_pts_ (S/frame.cpp:79-133): the indices of the five key features of a frame among
    its features in fts_ order -- closest to the centre, and the most "cornerward" one of each quadrant with the reference's
    own (quirky) quadrant tests and products -- or -1.  Host bookkeeping: the tracker's map upload wants the POINT of
    each key feature."""
    cu, cv = cam.width // 2, cam.height // 2
    key = [-1] * 5
    for i in range(len(px)):
        if not has_point[i]:
            continue
        x, y = px[i]
        if key[0] < 0 or max(abs(x - cu), abs(y - cv)) < max(abs(px[key[0]][0] - cu), abs(px[key[0]][1] - cv)):
            key[0] = i
        prod = (x - cu) * (y - cv)
        for slot, cond in ((1, x >= cu and y >= cv), (2, x >= cu and y < cv), (3, x < cv and y < cv), (4, x < cv and y >= cv)):
            if cond and (key[slot] < 0 or prod > (px[key[slot]][0] - cu) * (px[key[slot]][1] - cv)):
                key[slot] = i
    return np.array(key, dtype=np.int32)
