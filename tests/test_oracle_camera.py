"""The CPU oracle's camera model against the reference's OWN compiled vk::PinholeCamera members.

tests/golden/camera_ref.npz was produced by oracle/gen_golden.py (gen_camera_ref) from oracle/ref/ref_camera.cpp, which runs
PinholeCamera::world2cam (both overloads: pinhole and the 5-coefficient radtan forward model, pinhole_camera.cpp:73-106),
the distortion-free branch of PinholeCamera::cam2world (:44-53,70) and vk::AbstractCamera::isInFrame
(I/abstract_camera.h:58-72) unmodified on a hand-laid camera object.  Bars: pixels and bearings bit-identical, flags equal.
(The distorted branch of cam2world calls cv::undistortPoints, third-party code absent from the image: parity unpinned.)"""
import zlib

import numpy as np
import pytest

from oracle import gen_golden, orc


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@pytest.mark.parametrize("case", gen_golden.CAMERA_REF_CASES, ids=[c[0] for c in gen_golden.CAMERA_REF_CASES])
def test_camera_model_against_reference(golden, case):
    g = golden("camera_ref.npz")
    name = case[0]
    cam, xyz, uv, px, obs = gen_golden.camera_ref_inputs(case)
    assert [crc(xyz), crc(uv), crc(px), crc(obs)] == [int(v) for v in g[name + "_crc"]], "input generator drifted from the fixture"
    np.testing.assert_array_equal(orc.world2cam(cam, xyz), g[name + "_px_of_xyz"])
    np.testing.assert_array_equal(orc.world2cam_uv(cam, uv), g[name + "_px_of_uv"])
    if name + "_f_of_px" in g.files:
        np.testing.assert_array_equal(orc.cam2world(cam, px), g[name + "_f_of_px"])
    for boundary, level in ((0, 0), (8, 0), (8, 1), (6, 2), (9, 3)):
        np.testing.assert_array_equal(orc.is_in_frame(cam, obs, boundary), g["%s_in_b%d" % (name, boundary)])
        np.testing.assert_array_equal(orc.is_in_frame(cam, obs, boundary, level), g["%s_in_b%d_l%d" % (name, boundary, level)])


def test_fixture_covers_both_models(golden):
    g = golden("camera_ref.npz")
    # the strongly distorted camera really bends the projection; the tiny-k1 camera is treated as distortion-free
    cam, xyz, uv, px, obs = gen_golden.camera_ref_inputs(gen_golden.CAMERA_REF_CASES[1])
    ideal = np.stack([cam.fx * uv[:, 0] + cam.cx, cam.fy * uv[:, 1] + cam.cy], axis=1)
    assert np.abs(g["radtan_strong_px_of_uv"] - ideal).max() > 20.0
    cam, xyz, uv, px, obs = gen_golden.camera_ref_inputs(gen_golden.CAMERA_REF_CASES[4])
    ideal = np.stack([cam.fx * uv[:, 0] + cam.cx, cam.fy * uv[:, 1] + cam.cy], axis=1)
    np.testing.assert_array_equal(g["tiny_k1_px_of_uv"], ideal)
    assert "tiny_k1_f_of_px" in g.files and "radtan_strong_f_of_px" not in g.files
