"""World-size-2 gloo tests (CPU) of the multi-GPU plumbing in android_svo_amd/dist.py:
patch-sharded alignment with a per-Gauss-Newton-step all-reduce, and seed sharding with the
gather of converged seeds.  The per-rank compute is the CPU oracle; the driver code under test
(shard_range, run_allreduce, gather_converged) is the same one bench.py uses on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, what, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from android_svo_amd import dist as svodist, seedsynth, synth
        from oracle import orc
        if what == "align":
            from oracle_aligner import OracleShardedAligner
            fps = [synth.make_frame_pair(seed=70 + i, n_features=n) for i, n in enumerate((150, 301))]
            al = OracleShardedAligner(fps, rank, world)
            svodist.run_allreduce(al, 4, 0, 30)
            q.put((rank, [p.tolist() for p in al.poses], [st["n_meas"] for st in al.state]))
        else:
            sc = seedsynth.make_seed_case(n_seeds=1500, seed=13)
            lo, hi = svodist.shard_range(len(sc.px), rank, world)
            a, b, mu, s2 = (v[lo:hi].copy() for v in (sc.a, sc.b, sc.mu, sc.sigma2))
            o = None
            for _ in range(14):     # enough frames for part of the seeds to converge
                o = orc.update_seeds(sc.cam, sc.ref_pyr, sc.cur_pyr, sc.T_ref_w, sc.T_cur_w, sc.px[lo:hi], sc.f[lo:hi],
                                     sc.level[lo:hi], a, b, mu, sc.z_range[lo:hi].copy(), s2)
            conv = np.where(o["status"] == 4)[0]
            rec = svodist.gather_converged(conv + lo, mu[conv], s2[conv], o["xyz_world"][conv])
            q.put((rank, rec.tolist(), int(len(conv))))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _run(what, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, what, q)) for r in range(world)]
    [p.start() for p in procs]
    out = [q.get(timeout=300) for _ in range(world)]
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return sorted(out)


def test_shard_range_partitions():
    from android_svo_amd.dist import shard_range
    for n in (0, 1, 7, 64, 2000, 100003):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in edges) - min(h - l for l, h in edges) <= 1


@pytest.mark.timeout(600)
def test_patch_sharded_alignment_with_allreduce():
    from android_svo_amd import synth
    from oracle import orc
    out = _run("align")
    poses0, poses1 = np.array(out[0][1]), np.array(out[1][1])
    np.testing.assert_array_equal(poses0, poses1)              # identical control flow on both ranks
    for i, n in enumerate((150, 301)):
        fp = synth.make_frame_pair(seed=70 + i, n_features=n)
        o = orc.sparse_img_align(fp)
        rot, trans = synth.pose_error(poses0[i], np.array(o.T_cur_w))
        assert rot < 1e-6 and trans < 1e-6, (rot, trans)       # fp64 sums split in two: order noise only
        assert out[0][2][i] == o.n_tracked * 16


@pytest.mark.timeout(600)
def test_seed_sharding_and_gather_of_converged():
    from android_svo_amd import seedsynth
    from oracle import orc
    out = _run("seeds")
    rec0, rec1 = np.array(out[0][1]), np.array(out[1][1])
    np.testing.assert_array_equal(rec0, rec1)                  # all ranks see the same gathered set
    assert len(rec0) == out[0][2] + out[1][2] and len(rec0) > 0
    sc = seedsynth.make_seed_case(n_seeds=1500, seed=13)
    a, b, mu, s2 = sc.a.copy(), sc.b.copy(), sc.mu.copy(), sc.sigma2.copy()
    for _ in range(14):
        o = orc.update_seeds(sc.cam, sc.ref_pyr, sc.cur_pyr, sc.T_ref_w, sc.T_cur_w, sc.px, sc.f, sc.level, a, b, mu,
                             sc.z_range.copy(), s2)
    conv = np.where(o["status"] == 4)[0]
    np.testing.assert_array_equal(rec0[:, 0].astype(int), conv)          # same seeds, rank order = seed order
    np.testing.assert_array_equal(rec0[:, 1], mu[conv].astype(np.float64))
    np.testing.assert_array_equal(rec0[:, 3:6], o["xyz_world"][conv])
