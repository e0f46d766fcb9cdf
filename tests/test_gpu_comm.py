"""The native multi-GPU entry points of the C-ABI (no torch in the data path): svo_hip_sia_run_sharded (patch-sharded
SparseImgAlign with one all-reduce of the normal equations per Gauss-Newton step, BASELINE config C3's variant) and
svo_hip_seed_gather_converged_dev (seed-sharded depth filter, config C4's exchange).

Only one GPU is reachable from this build, so:
  * world size 2 runs as TWO PROCESSES ON THE ONE GPU over the host-staged shared-memory transport: the sharding, the
    masking of finished frames, the lock-step decisions and the gather layout are the product's; only the wire differs;
  * the RCCL transport itself is driven with one rank (communicator from a ncclUniqueId, ncclAllReduce / ncclAllGather
    on the context stream).  More ranks over xGMI are the driver's 8-GPU run."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

from android_svo_amd import hip, seedsynth, synth
from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "comm_worker.py")


def _run_ranks(what, transport, world, tmp_path):
    if transport == "shm":
        token = "/svo_test_" + uuid.uuid4().hex[:12]
    else:
        token = str(tmp_path / "nccl_id.bin")
        open(token, "wb").write(hip.Comm.unique_id())
    outs = [str(tmp_path / ("%s_%s_%d.npz" % (what, transport, r))) for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, what, transport, str(r), str(world), token, outs[r]], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se.decode()[-3000:]
    return [dict(np.load(o)) for o in outs]          # materialised: a later run may reuse the file names


def _check_sia(results):
    fps = [synth.make_frame_pair(seed=900 + i, n_features=n, null_point_every=k) for i, (n, k) in enumerate(((700, 0), (333, 7), (1500, 0)))]
    for tag, es, n_iter in (("early", True, 30), ("fixed", False, 6)):
        for r in results[1:]:       # every rank ends with the same bits: identical sums -> identical decisions
            np.testing.assert_array_equal(r[tag + "_T"], results[0][tag + "_T"])
            np.testing.assert_array_equal(r[tag + "_H"], results[0][tag + "_H"])
            np.testing.assert_array_equal(r[tag + "_iters"], results[0][tag + "_iters"])
        for s, fp in enumerate(fps):
            # svo_hip_sia_run on the same object afterwards: no shard left behind (whole frames, frame-parallel kernel)
            rot, trans = synth.pose_error(results[0][tag + "_T_plain"][s], results[0][tag + "_T"][s])
            assert rot < 2e-5 and trans < 5e-5, (tag, s, rot, trans)
            o = orc.sparse_img_align(fp, n_iter=n_iter, early_stop=es)
            rot, trans = synth.pose_error(results[0][tag + "_T"][s], np.array(o.T_cur_w))
            assert rot < 1e-4 and trans < 1e-3, (tag, s, rot, trans)             # north_star tolerance
            assert rot < 2e-5 and trans < 5e-5, (tag, s, rot, trans)
            assert int(results[0][tag + "_n"][s]) == o.n_tracked
            if not es:                                                           # same evaluation sequence: tight
                assert rot < 1e-9 and trans < 1e-9, (s, rot, trans)
                assert list(results[0][tag + "_iters"][s]) == list(o.iters)[:5]


def test_sharded_sparse_img_align_two_ranks_on_one_gpu(tmp_path):
    _check_sia(_run_ranks("sia", "shm", 2, tmp_path))


def test_sharded_sparse_img_align_three_ranks_on_one_gpu(tmp_path):
    _check_sia(_run_ranks("sia", "shm", 3, tmp_path))      # shards of unequal size (700/3, 333/3)


def test_sharded_sparse_img_align_rccl_one_rank(tmp_path):
    _check_sia(_run_ranks("sia", "rccl", 1, tmp_path))


def test_sharded_sparse_img_align_rccl_graph_replay(tmp_path, monkeypatch):
    """svo_hip_sia_set_sharded_graph: the per-level launch sequence (kernels + ncclAllReduce) captured once and replayed:
    the same bits as the eager sequence."""
    eager = _run_ranks("sia", "rccl", 1, tmp_path)
    monkeypatch.setenv("SVO_TEST_GRAPH", "1")
    graph = _run_ranks("sia", "rccl", 1, tmp_path)
    _check_sia(graph)
    for tag in ("early", "fixed"):
        np.testing.assert_array_equal(graph[0][tag + "_T"], eager[0][tag + "_T"])
        np.testing.assert_array_equal(graph[0][tag + "_H"], eager[0][tag + "_H"])


def test_config_c3_shape_two_ranks_on_one_gpu(tmp_path):
    """1280x720 frame pairs with 2000 patches each through the patch-sharded all-reduce path at world size 2."""
    res = _run_ranks("sia_c3", "shm", 2, tmp_path)
    np.testing.assert_array_equal(res[0]["T"], res[1]["T"])
    np.testing.assert_array_equal(res[0]["H"], res[1]["H"])
    for s in range(2):
        fp = synth.make_frame_pair(seed=3300 + s, n_features=2000, width=1280, height=720)
        o = orc.sparse_img_align(fp, n_iter=4, early_stop=False)
        rot, trans = synth.pose_error(res[0]["T"][s], np.array(o.T_cur_w))
        assert rot < 1e-9 and trans < 1e-9, (s, rot, trans)
        assert int(res[0]["n"][s]) == o.n_tracked == 2000


@pytest.mark.parametrize("world", [3, 4])
def test_sharded_solve_does_not_depend_on_launch_timing(tmp_path, world):
    """Regression test for a race inside one launch of svo_hip_sia_run_sharded (found by rehearsing `bench.py --mode allreduce`
    with 3 and 4 ranks on one GPU): the evaluation kernel read the previous evaluation's exchanged block partials at its head
    and wrote its own partials at its tail into the SAME buffer, so a block that finished early could overwrite rows another
    block of the frame had not read yet -- the ranks then stepped differently and drifted apart by ~1e-4 rad, in about half
    of the runs with three or more ranks and event records around the launches.  The partials are double-buffered now: four
    solves in a row, event profiling on, every rank and every repetition the same bits, poses at the oracle's."""
    res = _run_ranks("sia_timing", "shm", world, tmp_path)
    for r in res:
        for rep in range(r["T"].shape[0]):
            np.testing.assert_array_equal(r["T"][rep], res[0]["T"][0])
    for s in range(6):
        fp = synth.make_frame_pair(seed=12345 + s, n_features=2000)
        o = orc.sparse_img_align(fp, n_iter=30, early_stop=False)
        rot, trans = synth.pose_error(res[0]["T"][0][s], np.array(o.T_cur_w))
        assert rot < 1e-8 and trans < 1e-8, (s, rot, trans)


@pytest.mark.parametrize("world", [2, 3])
def test_config_c3_full_size_on_one_gpu(tmp_path, world):
    """BASELINE config C3 at its full size -- 8 concurrent 1280x720 pairs, 2000 patches each, 5 levels x 30 evaluations --
    through the patch-sharded path (one launch + one all-reduce of the block partials per Gauss-Newton step) at world
    sizes 2 and 3 (processes on the one GPU, host-staged transport): every rank ends with the same bits, poses within 1e-9
    of the oracle's fixed-work runs (same evaluation sequence), north_star's tolerance with five orders to spare."""
    res = _run_ranks("sia_c3_full", "shm", world, tmp_path)
    for r in res[1:]:
        np.testing.assert_array_equal(r["T"], res[0]["T"])
        np.testing.assert_array_equal(r["H"], res[0]["H"])
    for s in range(8):
        fp = synth.make_frame_pair(seed=3300 + s, n_features=2000, width=1280, height=720)
        o = orc.sparse_img_align(fp, n_iter=30, early_stop=False)
        rot, trans = synth.pose_error(res[0]["T"][s], np.array(o.T_cur_w))
        assert rot < 1e-9 and trans < 1e-9, (s, rot, trans)
        assert int(res[0]["n"][s]) == o.n_tracked == 2000


def _check_seeds(results, world):
    n = 6000
    total = sum(int(r["n_conv_local"]) for r in results)
    assert 200 < total < n
    for k, r in enumerate(results):
        # every rank holds the full gather: ordered by rank, then by seed, ids global
        np.testing.assert_array_equal(r["rec"], results[0]["rec"])
        assert list(r["counts"]) == [int(x["n_conv_local"]) for x in results]
        assert len(r["rec"]) == total
    rec = results[0]["rec"]
    ids = np.concatenate([r["local_ids"] for r in results])
    np.testing.assert_array_equal(rec[:, 0].astype(int), ids)
    np.testing.assert_array_equal(rec[:, 1], np.concatenate([r["local_mu"] for r in results]).astype(np.float64))
    np.testing.assert_array_equal(rec[:, 3:], np.concatenate([r["local_xyz"] for r in results]))
    assert (np.diff(rec[:, 0]) > 0).all()
    # a capacity that is too small is reported through the counts (true count > cap), the records are the first `cap`
    for r in results:
        assert list(r["counts_small"]) == list(r["counts"])
        assert len(r["rec_small"]) == sum(min(int(c), 10) for c in r["counts"])


def test_seed_gather_two_ranks_on_one_gpu(tmp_path):
    _check_seeds(_run_ranks("seeds", "shm", 2, tmp_path), 2)


def test_seed_gather_rccl_one_rank(tmp_path):
    _check_seeds(_run_ranks("seeds", "rccl", 1, tmp_path), 1)


def test_adopted_nccl_communicator(tmp_path):
    """svo_hip_comm_from_nccl: a communicator the application created itself (here: RCCL called directly, one rank) carries the
    sharded solve -- same bits as the library's own communicator -- and is still alive after svo_hip_comm_destroy."""
    import ctypes as C
    rccl = None
    for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    assert rccl is not None, "librccl is part of the image"
    ctx = hip.Context(0)
    uid = (C.c_char * 128)()
    assert rccl.ncclGetUniqueId(uid) == 0

    class NcclId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    nid = NcclId()
    C.memmove(C.byref(nid), uid, 128)
    comm_ptr = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm_ptr), 1, nid, 0) == 0
    fps = [synth.make_frame_pair(seed=900 + i, n_features=n) for i, n in enumerate((700, 1500))]
    cam = fps[0].cam
    ref = hip.Pyramid(ctx, cam.width, cam.height, 5, 2); cur = hip.Pyramid(ctx, cam.width, cam.height, 5, 2)
    sia = hip.SparseImgAlign(ctx, 2, 1500)
    sia.set_frames(ref, cur)
    for s, fp in enumerate(fps):
        ref.upload(s, fp.ref_pyr); cur.upload(s, fp.cur_pyr); sia.upload_pair(s, fp)
    prm = sia.params(max_level=4, min_level=0, n_iter=6, eps=1e-6, early_stop=False)
    adopted = hip.Comm.from_nccl(ctx, comm_ptr.value, 0, 1)
    hip.sia_run_sharded(sia, adopted, 2, prm)
    Ta = np.array([list(x.T_cur_w) for x in sia.download_all(2)])
    adopted.destroy()
    own = hip.Comm(ctx, 0, 1, kind="rccl", unique_id=hip.Comm.unique_id())
    hip.sia_run_sharded(sia, own, 2, prm)
    To = np.array([list(x.T_cur_w) for x in sia.download_all(2)])
    own.destroy()
    np.testing.assert_array_equal(Ta, To)
    for s, fp in enumerate(fps):
        o = orc.sparse_img_align(fp, n_iter=6, early_stop=False)
        rot, trans = synth.pose_error(Ta[s], np.array(o.T_cur_w))
        assert rot < 1e-9 and trans < 1e-9
    # the adopted communicator is the application's: still usable, and the application's to destroy
    cnt = C.c_int(0)
    assert rccl.ncclCommCount(comm_ptr, C.byref(cnt)) == 0 and cnt.value == 1
    assert rccl.ncclCommDestroy(comm_ptr) == 0
    ctx.close()



def test_dead_segment_of_a_killed_rank0_is_not_joined(tmp_path):
    """A rank 0 that was killed while it sat in its first barrier leaves a segment with the magic set and arrived == world - 1
    under the group's name: the barrier of that DEAD segment opens for the next attaching rank's own increment.  The attaching
    rank must not trust a segment until a live rank 0 has set `go` behind the barrier -- it keeps watching the name and moves
    to the segment the new rank 0 creates."""
    import struct
    import time
    token = "/svo_test_" + uuid.uuid4().hex[:12]
    world, slot_bytes = 2, 8 << 20                       # comm_worker.py's geometry
    total = 256 + world * slot_bytes
    path = "/dev/shm" + token
    with open(path, "wb") as fh:                         # header: magic, world, slot_bytes (u64), arrived, generation, go
        fh.write(struct.pack("<IIQIII", 0x53564F43, world, slot_bytes, world - 1, 0, 0))
        fh.truncate(total)
    outs = [str(tmp_path / ("dead_%d.npz" % r)) for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    start = lambda r: subprocess.Popen([sys.executable, WORKER, "seeds", "shm", str(r), str(world), token, outs[r]], cwd=ROOT, env=env,
                                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    p1 = start(1)                                        # the attaching rank first: it finds the dead segment under the name
    time.sleep(4.0)
    p0 = start(0)
    for p in (p0, p1):
        so, se = p.communicate(timeout=200)
        assert p.returncode == 0, se.decode()[-3000:]
    _check_seeds([dict(np.load(o)) for o in outs], world)
    assert not os.path.exists(path)                      # rank 0 removed the name once everybody was in
