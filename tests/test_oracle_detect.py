"""Oracle restatement of the seed producer: FastDetector::detect.

vk::shiTomasiScore is pinned by the reference's own vision.cpp (tests/golden/shitomasi_ref.npz).  cv::FAST is
OpenCV 4.5.4 code that is not present under /root/reference (neither source nor a host library): PARITY UNPINNED.
The restatement is checked here against an independent brute-force evaluation of the published definition
(9 contiguous circle pixels brighter/darker than the centre by more than t; score = largest threshold for which the
pixel is still a corner; 3x3 non-maximum suppression with strict comparisons)."""
import numpy as np

from android_svo_amd import synth
from oracle import orc

DX = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
DY = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]


def test_shi_tomasi_bit_identical_to_reference(golden):
    g = golden("shitomasi_ref.npz")
    for name in ("scene", "noise"):
        img = g[name + "_img"]
        got = np.array([orc.shi_tomasi_score(img, int(u), int(v)) for u, v in g[name + "_uv"]], dtype=np.float32)
        np.testing.assert_array_equal(got, g[name + "_score"])
        assert (g[name + "_score"][:8] == 0).sum() >= 4          # the border rule returns 0


def _is_corner(img, x, y, t):
    v = int(img[y, x])
    ring = [int(img[y + DY[k], x + DX[k]]) for k in range(16)]
    for sign in (1, -1):
        hit = [sign * (v - r) > t for r in ring]
        for s in range(16):
            if all(hit[(s + j) % 16] for j in range(9)):
                return True
    return False


def _brute_force_fast(img, t):
    h, w = img.shape
    score = np.zeros((h, w), dtype=np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if not _is_corner(img, x, y, t):
                continue
            s = t
            while s < 255 and _is_corner(img, x, y, s + 1):      # the largest threshold that still gives a corner
                s += 1
            score[y, x] = s
    pts = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s and all(s > score[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0)):
                pts.append((x, y, s))
    return pts


def test_fast_matches_brute_force_definition():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (40, 56)).astype(np.uint8)
    img[10:30, 12:40] = (img[10:30, 12:40] // 8) + 40            # a flat region with weak texture
    img[18:24, 20:30] = 230                                      # and a bright block: real corners
    xs, ys, ss = orc.fast(img, 10)
    want = _brute_force_fast(img, 10)
    assert len(want) > 20
    assert [(int(x), int(y)) for x, y in zip(xs, ys)] == [(x, y) for x, y, _ in want]        # same points, row-major
    # cornerScore = (max over the arcs of the smallest |difference|) - 1 = the largest threshold that still gives a corner
    np.testing.assert_array_equal(ss, np.array([s for _, _, s in want]))


def test_detect_features_grid_rules():
    fp = synth.make_frame_pair(seed=12345, n_features=10)
    px, lvl, sc = orc.detect_features(fp.ref_pyr, n_pyr_levels=3, cell_size=20)
    assert len(px) > 300 and (sc > 10.0).all()
    cells = (px[:, 1] // 20) * 32 + px[:, 0] // 20
    assert (np.diff(cells) > 0).all()                            # one feature per cell, in cell order
    assert (px % (1 << lvl)[:, None] == 0).all()                 # level-l corners sit on multiples of 2^l
    occ = np.zeros(32 * 24, dtype=np.uint8)
    occ[cells[::2]] = 1
    px2, lvl2, sc2 = orc.detect_features(fp.ref_pyr, occupancy=occ)
    cells2 = (px2[:, 1] // 20) * 32 + px2[:, 0] // 20
    assert not occ[cells2].any() and set(cells2) == set(cells[1::2])
    # the winner of a cell has the best Shi-Tomasi score among that cell's corners of all three levels
    best = {}
    for L in range(3):
        xs, ys, _ = orc.fast(fp.ref_pyr[L], 10)
        for x, y in zip(xs, ys):
            k = int((y << L) // 20) * 32 + int((x << L) // 20)
            s = orc.shi_tomasi_score(fp.ref_pyr[L], int(x), int(y))
            best[k] = max(best.get(k, 10.0), s)
    for k, s in zip(cells, sc):
        assert best[int(k)] == s
