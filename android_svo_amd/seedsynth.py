"""Synthetic depth-filter and align2D workloads: a keyframe with seeds on a
grid, one current frame at a known baseline, and 8x8 patches cut from the reference image."""
from __future__ import annotations

import dataclasses
from typing import List

import numpy as np

from . import synth


@dataclasses.dataclass
class SeedCase:
    cam: synth.Camera
    ref_pyr: List[np.ndarray]
    cur_pyr: List[np.ndarray]
    T_ref_w: np.ndarray
    T_cur_w: np.ndarray
    px: np.ndarray        # [n,2] f64 level-0 pixel of the seed's feature
    f: np.ndarray         # [n,3] f64 unit bearing
    level: np.ndarray     # [n] i32 pyramid level of the feature
    a: np.ndarray         # Seed state, f32
    b: np.ndarray
    mu: np.ndarray
    z_range: np.ndarray
    sigma2: np.ndarray
    true_depth: np.ndarray  # [n] range along the bearing (ground truth)


def seed_ctor(depth_mean: float, depth_min: float, n: int):
    """svo::Seed::Seed (depth_filter.cpp:36-45): a=b=10, mu=1/mean, z_range=1/min, sigma2=z_range^2/36."""
    dm, dn = np.float32(depth_mean), np.float32(depth_min)
    mu = np.float32(1.0 / float(dm))
    zr = np.float32(1.0 / float(dn))
    s2 = np.float32(np.float32(zr * zr) / np.float32(36))
    full = lambda v: np.full(n, v, dtype=np.float32)
    return full(10), full(10), full(mu), full(zr), full(s2)


def make_seed_case(n_seeds: int = 4096, seed: int = 7, width: int = 640, height: int = 480, baseline: float = 0.08,
                   depth: float = 2.0, levels=(0, 0, 0, 1, 2), border: int = 40) -> SeedCase:
    rng = np.random.default_rng(seed)
    cam = synth.Camera.default(width, height)
    scene = synth.PlaneScene(seed=seed, depth=depth, tilt=(rng.uniform(-0.15, 0.15), rng.uniform(-0.15, 0.15)))
    T_ref_w = synth.se3_from_twist(rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.02, 0.02, 3))
    direction = rng.normal(size=3) * [1.0, 1.0, 0.25]
    direction /= np.linalg.norm(direction)
    T_cur_ref = synth.se3_from_twist(direction * baseline, rng.uniform(-0.01, 0.01, 3))
    T_cur_w = synth.se3_mul(T_cur_ref, T_ref_w)
    ref_img = scene.render(cam, T_ref_w)
    cur_img = scene.render(cam, T_cur_w)
    w, h = width - 2 * border, height - 2 * border
    stride = np.sqrt(w * h / float(n_seeds))
    nx = max(1, int(np.ceil(w / stride)))
    ny = int(np.ceil(n_seeds / nx))
    xs = border + np.arange(nx) * (w - 1.0) / nx
    ys = border + np.arange(ny) * (h - 1.0) / ny
    gx, gy = np.meshgrid(xs, ys)
    px = np.stack([gx.ravel(), gy.ravel()], axis=1)[:n_seeds]
    px = np.floor(px)                                   # FAST corners sit on integer pixels
    level = rng.choice(np.asarray(levels, dtype=np.int32), size=n_seeds)
    px = px - (px % (1 << level)[:, None])              # a level-l corner is a multiple of 2^l
    f = synth.cam2world(cam, px)
    X = scene.intersect(cam, T_ref_w, px[:, 0], px[:, 1])
    Xc = (X - synth.se3_inv(T_ref_w)[:3])
    true_depth = np.linalg.norm(Xc, axis=1)
    zbar = float(np.median(true_depth))
    a, b, mu, zr, s2 = seed_ctor(1.1 * zbar, 0.5 * zbar, n_seeds)
    return SeedCase(cam, synth.build_pyramid(ref_img), synth.build_pyramid(cur_img), T_ref_w, T_cur_w,
                    np.ascontiguousarray(px), np.ascontiguousarray(f), level.astype(np.int32), a, b, mu, zr, s2,
                    true_depth)


@dataclasses.dataclass
class MultiKeyframeCase:
    """One current frame and the seeds of several keyframes of the same scene (what one DepthFilter::updateSeeds call walks)."""
    cam: synth.Camera
    cur_pyr: List[np.ndarray]
    T_cur_w: np.ndarray
    keyframes: List[SeedCase]      # per keyframe: its pyramid (ref_pyr), pose (T_ref_w) and seeds; cur_pyr / T_cur_w repeated


def make_multi_keyframe_case(sizes, seed: int = 5, width: int = 640, height: int = 480, baseline: float = 0.08, depth: float = 2.0,
                             levels=(0, 0, 0, 1, 2), border: int = 40) -> MultiKeyframeCase:
    rng = np.random.default_rng(seed)
    cam = synth.Camera.default(width, height)
    scene = synth.PlaneScene(seed=seed, depth=depth, tilt=(rng.uniform(-0.15, 0.15), rng.uniform(-0.15, 0.15)))
    T_cur_w = synth.se3_from_twist(rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.02, 0.02, 3))
    cur_pyr = synth.build_pyramid(scene.render(cam, T_cur_w))
    kfs = []
    for n in sizes:
        direction = rng.normal(size=3) * [1.0, 1.0, 0.25]
        direction /= np.linalg.norm(direction)
        T_ref_cur = synth.se3_from_twist(direction * baseline * rng.uniform(0.6, 1.4), rng.uniform(-0.01, 0.01, 3))
        T_ref_w = synth.se3_mul(T_ref_cur, T_cur_w)
        ref_img = scene.render(cam, T_ref_w)
        px = np.stack([rng.integers(border, width - border, n), rng.integers(border, height - border, n)], axis=1).astype(np.float64)
        level = rng.choice(np.asarray(levels, dtype=np.int32), size=n)
        px = px - (px % (1 << level)[:, None])
        f = synth.cam2world(cam, px)
        X = scene.intersect(cam, T_ref_w, px[:, 0], px[:, 1])
        true_depth = np.linalg.norm(X - synth.se3_inv(T_ref_w)[:3], axis=1)
        zbar = float(np.median(true_depth))
        a, b, mu, zr, s2 = seed_ctor(1.1 * zbar, 0.5 * zbar, n)
        kfs.append(SeedCase(cam, synth.build_pyramid(ref_img), cur_pyr, T_ref_w, T_cur_w, np.ascontiguousarray(px),
                            np.ascontiguousarray(f), level.astype(np.int32), a, b, mu, zr, s2, true_depth))
    return MultiKeyframeCase(cam, cur_pyr, T_cur_w, kfs)


@dataclasses.dataclass
class AlignCase:
    cam: synth.Camera
    cur_pyr: List[np.ndarray]
    pwb: np.ndarray       # [n,100] u8
    patch: np.ndarray     # [n,64] u8
    px_init: np.ndarray   # [n,2] f64
    px_true: np.ndarray   # [n,2] f64


def make_align_case(n: int = 5000, seed: int = 11, width: int = 640, height: int = 480, jitter: float = 2.0) -> AlignCase:
    """8x8 patches cut from the reference image at integer centres; the same world points seen in a
    slightly moved current image, initial px = truth + U(-jitter, jitter)^2."""
    rng = np.random.default_rng(seed)
    cam = synth.Camera.default(width, height)
    scene = synth.PlaneScene(seed=seed, depth=2.0)
    T_ref_w = synth.se3_from_twist([0, 0, 0], [0, 0, 0])
    T_cur_w = synth.se3_from_twist(rng.uniform(-0.02, 0.02, 3), rng.uniform(-0.004, 0.004, 3))
    ref_img = scene.render(cam, T_ref_w)
    cur_img = scene.render(cam, T_cur_w)
    cx = rng.integers(16, width - 16, n)
    cy = rng.integers(16, height - 16, n)
    idx_y = cy[:, None, None] + np.arange(-5, 5)[None, :, None]
    idx_x = cx[:, None, None] + np.arange(-5, 5)[None, None, :]
    pwb = ref_img[idx_y, idx_x].reshape(n, 100)
    patch = pwb.reshape(n, 10, 10)[:, 1:9, 1:9].reshape(n, 64)
    X = scene.intersect(cam, T_ref_w, cx.astype(np.float64), cy.astype(np.float64))
    R = synth.rot_matrix(T_cur_w[3:])
    Xc = X @ R.T + T_cur_w[:3]
    px_true = np.stack([cam.fx * Xc[:, 0] / Xc[:, 2] + cam.cx, cam.fy * Xc[:, 1] / Xc[:, 2] + cam.cy], axis=1)
    px_init = px_true + rng.uniform(-jitter, jitter, (n, 2))
    return AlignCase(cam, synth.build_pyramid(cur_img), np.ascontiguousarray(pwb), np.ascontiguousarray(patch),
                     np.ascontiguousarray(px_init), px_true)
