"""bench.py's multi-rank code path, rehearsed with more ranks than GPUs (bench.py then uses a gloo process group and, for the
sharded mode, the library's shared-memory exchange; its JSON line says `rehearsal`).  The driver launches the same command
on a multi-GPU node at round end; nothing here measures scaling -- it keeps the branch that only runs with WORLD_SIZE > 1
executable (rank-0-only sections, the timing all-reduce, the communicator hand-over, the parity check over the global batch)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(n, port, *args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-secondary",
           "--no-cpu-baseline", "--distinct", "4"] + list(args)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                   # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_frame_parallel_two_ranks():
    d = _torchrun(2, 29671, "--batch", "8")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "rehearsal" in d
    assert d["config"]["global_frame_pairs_per_step"] == 16
    assert d["pose_err_vs_cpu_ref"]["max_rot_rad_over_scenes"] < 1e-4 and d["value"] > 0


def test_bench_sharded_three_ranks():
    d = _torchrun(3, 29672, "--mode", "allreduce", "--batch", "1")
    assert d["n_gpus"] == 3 and "rehearsal" in d and "patch-sharded" in d["config"]["parallelism"]
    assert d["pose_err_vs_cpu_ref"]["max_rot_rad_over_scenes"] < 1e-8          # fixed work: the same evaluation sequence as the oracle


def _plain(script, *args):
    """`python <script> --gpus N ...` started plainly: no launcher, no RANK / WORLD_SIZE in the environment -- the script
    starts its own ranks (android_svo_amd/launcher.py) and relays rank 0's one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, script)] + list(args), cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_started_plainly_starts_its_own_ranks():
    d = _plain("bench.py", "--gpus", "2", "--batch", "8", "--steps", "2", "--warmup", "1", "--no-secondary", "--no-cpu-baseline",
               "--distinct", "4")
    assert d["n_gpus"] == 2 and "rehearsal" in d and d["config"]["global_frame_pairs_per_step"] == 16
    assert d["pose_err_vs_cpu_ref"]["max_rot_rad_over_scenes"] < 1e-4


def test_bench_sharded_started_plainly_reports_the_communicators_rank_count():
    d = _plain("bench.py", "--gpus", "2", "--mode", "allreduce", "--batch", "1", "--steps", "2", "--warmup", "1", "--no-secondary",
               "--no-cpu-baseline", "--distinct", "2")
    assert d["n_gpus"] == 2 and d["config"]["comm_ranks"] == 2 and "patch-sharded" in d["config"]["parallelism"]


def test_bench_c4_gpus_2_started_plainly():
    d = _plain("bench_c4.py", "--gpus", "2", "--seeds", "40000", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 2 and "rehearsal" in d and d["config"]["comm_ranks"] == 2
    assert d["config"]["converged_records_gathered"] > 0 and d["config"]["seeds_per_gpu"] == 20000


def test_default_bench_line_fits_the_drivers_tail_and_carries_the_contract():
    """`python bench.py` as the driver runs it (N = 1, every secondary section on): ONE JSON line of at most 6 KiB (round 3's 14 kB
    line lost its first half in the driver's tail), with the contract keys, `roofline` and `cpu_baseline` objects, the per-stage
    bounds of the depth filter and the drop-in entry's timing.  (One of the two places where a process on the GPU box maps reference object code: bench.py's `cpu_baseline` / `c0` legs
    time the reference's own compiled SparseImgAlign where the prebuilt oracle/_ref/libsvo_ref.so is present; the other is
    tests/test_gpu_dropin_binding.py, which RUNS the drop-in bindings against the reference's own types.)"""
    d_line = None
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d_line = lines[0]
    assert len(d_line.encode()) <= 6144, len(d_line)
    d = json.loads(d_line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["vs_baseline"] is None and "workload" in d["config"] and d["config"]["workload"].startswith("C1")
    r = d["roofline"]
    assert r["bound"] in ("valu", "hbm") and 0.0 < r["frac"] <= 1.0 and r["peak"] > 0 and r["traffic"] > 0 and "profile_refused" not in r
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("port", "reference") and (c["kind"] == "port" or "driver" in c)
    assert d["pose_err_vs_cpu_ref"]["max_rot_rad_over_scenes"] < 1e-4 and d["pose_err_vs_cpu_ref"]["n_tracked_equal_in_every_scene"]
    for lvl in ("moments_f32_arithmetic", "fast_arithmetic"):
        assert d[lvl]["value"] > 0 and 0.0 < d[lvl]["roofline"]["frac"] <= 1.0 and d[lvl]["pose_err_vs_cpu_ref"]["max_rot_rad_over_scenes"] < 1e-6
    df = d["c2"]["depth_filter"]
    assert set(df["stages_us"]) == {"geometry", "search", "align", "finalize"} and 0 < df["through_dropin_entry_us"] < 400
    for stage in ("geometry", "search", "align", "finalize"):
        assert df["roofline"][stage]["bound"] in ("valu", "hbm") and 0.0 < df["roofline"][stage]["frac"] <= 1.0
        assert 0.0 < d["c4_one_gpu"]["roofline"][stage]["frac"] <= 1.0
    assert d["single_stream_chain"]["L4_L2_shipping_default"]["matched_points_equal_in_every_frame"] is True
    # the headline is measured at the reference's arithmetic (the library default), the narrower levels are secondary legs
    assert "as the reference" in d["dtype"] and d["config"]["arithmetic"].startswith("EXACT")
    assert d["pose_err_vs_cpu_ref"]["max_rot_rad_over_scenes"] < 1e-7
    # BASELINE config C0 beside C1 (SURVEY 8d "CPU baseline timing"): one pinned core and all cores, both Gauss-Newton modes
    c0 = d["c0"]
    assert 100 <= c0["patches"] <= 260 and c0["max_pose_err_vs_cpu"] < 1e-4
    assert c0["cpu"]["es_ms_1core"] > 0 and c0["cpu"]["fw_ms_1core"] > c0["cpu"]["es_ms_1core"] and c0["cpu"]["kind_es"] in ("reference", "port")
    assert c0["gpu"]["es_fps"] > c0["cpu"]["es_fps"] and c0["gpu"]["fw_fps"] > c0["cpu"]["fw_fps"]
    # N cameras: N trackers on N host threads, and one tracker group (one chain of launches per call)
    ch = d["single_stream_chain"]
    assert ch["group_frames_per_s"]["8"] > 2.0 * ch["group_frames_per_s"]["1"] and ch["cameras_frames_per_s"]["1"] > 0
    for stage in ("search", "align"):
        assert 0.0 < df["roofline"][stage]["lane_util"] <= 1.0
