"""hip_bridge::DeviceSeedMirror -- the list <-> device bookkeeping of the drop-in DepthFilter (include/svo_dropin/
depth_filter_batch.h) -- on the CPU against a mock of the svo_hip_seed_batch_* entry points: upload once, events in list
order, age-out, erasures behind the mirror's back, recycled list nodes, the halt flag, syncToHost.  Compiled with the
reference's language level (-std=c++11).  The GPU run of the same template is tests/test_gpu_host_cpp.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_seed_mirror_bookkeeping_against_a_mock_device(tmp_path):
    exe = tmp_path / "mirror_mock_test"
    src = os.path.join(ROOT, "tests", "host_mock", "mirror_mock_test.cpp")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), src, "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mirror mock test OK" in r.stdout
