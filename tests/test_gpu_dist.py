"""The NCCL (= RCCL) code paths on the one GPU this build can reach: a single rank under torch.distributed.run.
The patch-sharded all-reduce solve of bench.py --mode allreduce (RCCL called by libsvo_hip.so) and --mode allreduce-torch
(the torch.distributed driver), and the seed-sharded depth filter with the on-device gather (bench_c4.py).  More ranks:
tests/test_gpu_comm.py (several processes on the one GPU over the host-staged transport) and the gloo tests on the CPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_args, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-500:]                    # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


def test_allreduce_solve_native_and_torch_driver():
    """BASELINE config C3's variant with one rank: the patch-sharded solve with the all-reduce of the normal equations
    (a) through the C-ABI (svo_hip_sia_run_sharded: libsvo_hip.so calls RCCL itself) and (b) driven from Python through
    torch.distributed -- the same kernels and the same sums, so the same bits."""
    common = ["bench.py", "--gpus", "1", "--batch", "4", "--features", "600", "--steps", "2", "--warmup", "1", "--distinct", "4",
              "--no-cpu-baseline", "--profile-events", "0"]
    native = _torchrun(common + ["--mode", "allreduce"], 29541)
    torch_driver = _torchrun(common + ["--mode", "allreduce-torch"], 29542)
    for d in (native, torch_driver):
        assert d["n_gpus"] == 1 and d["pose_err_vs_cpu_ref"]["rot_rad"] < 1e-9 and d["pose_err_vs_cpu_ref"]["trans_m"] < 1e-9
        assert "all-reduce" in d["config"]["parallelism"]
    assert "libsvo_hip.so" in native["config"]["parallelism"]
    assert native["pose_err_vs_cpu_ref"] == torch_driver["pose_err_vs_cpu_ref"]


def test_seed_sharded_depth_filter_with_device_gather():
    d = _torchrun(["bench_c4.py", "--seeds", "60000", "--width", "640", "--height", "480", "--steps", "2", "--warmup", "1",
                   "--sigma-scale", "0.0012"], 29543)
    assert d["n_gpus"] == 1 and d["config"]["seeds_per_gpu"] == 60000
    assert 0 < d["config"]["converged_records_gathered"] <= 60000
