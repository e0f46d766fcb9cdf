"""The NCCL (= RCCL) code paths on the one GPU this build can reach: a single rank under torch.distributed.run.
The patch-sharded all-reduce solve of bench.py --mode allreduce, eager and replayed from HIP graphs, and the
seed-sharded depth filter with the on-device gather (bench_c4.py).  More ranks are covered by the gloo tests."""
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


def test_allreduce_solve_eager_and_graph_replay():
    common = ["bench.py", "--gpus", "1", "--mode", "allreduce", "--batch", "4", "--features", "600", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline", "--profile-events", "0"]
    eager = _torchrun(common, 29541)
    graph = _torchrun(common + ["--graph"], 29542)
    for d in (eager, graph):
        assert d["n_gpus"] == 1 and d["pose_err_vs_cpu_ref"]["rot_rad"] < 1e-9 and d["pose_err_vs_cpu_ref"]["trans_m"] < 1e-9
        assert "all-reduce" in d["config"]["parallelism"]
    assert graph["pose_err_vs_cpu_ref"] == eager["pose_err_vs_cpu_ref"]      # the same kernels, the same bits
    assert "HIP-graph" in graph["config"]["parallelism"]


def test_seed_sharded_depth_filter_with_device_gather():
    d = _torchrun(["bench_c4.py", "--seeds", "60000", "--width", "640", "--height", "480", "--steps", "2", "--warmup", "1",
                   "--sigma-scale", "0.0012"], 29543)
    assert d["n_gpus"] == 1 and d["config"]["seeds_per_gpu"] == 60000
    assert 0 < d["config"]["converged_records_gathered"] <= 60000
