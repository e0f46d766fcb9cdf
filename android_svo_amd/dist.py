"""Multi-GPU plumbing for the SVO hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Two partitionings:
  * frames / seeds are independent objects -> shard them across ranks, no data-path collective
    (`shard_range`); converged seeds are collected with `gather_converged`;
  * one frame pair split by patches (BASELINE config C3): every rank evaluates its contiguous
    slice of every frame's patches, the per-frame sums (32 doubles: 21 H + 6 Jres + chi2 + n_meas)
    are all-reduced once per Gauss-Newton step, then every rank runs the identical solve
    (`run_allreduce`).  The aligner is any object with the step-wise interface of
    include/svo_hip.h (begin / level_begin / accumulate / solve_update / finish) that exposes the
    reduce buffer as a torch tensor.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

REDUCE_DOUBLES = 32


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of n items owned by `rank` (same formula as the device code)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def run_allreduce(aligner, max_level: int, min_level: int, n_iter: int, group=None) -> None:
    """Coarse-to-fine solve with the normal equations exchanged at every Gauss-Newton step.
    All ranks hold identical control state, so the data-dependent exits stay in lock-step."""
    import torch.distributed as dist
    aligner.begin()
    for level in range(max_level, min_level - 1, -1):
        aligner.level_begin(level)
        for _ in range(n_iter):
            aligner.accumulate()
            dist.all_reduce(aligner.reduce_tensor, op=dist.ReduceOp.SUM, group=group)
            aligner.solve_update()
    aligner.finish()


class HipShardedAligner:
    """hip.SparseImgAlign evaluating this rank's patch shard; the reduce buffer is a torch CUDA
    tensor so that RCCL reduces it in place on the solver's stream."""

    def __init__(self, sia, n_slots: int, prm, rank: int, world: int, stream):
        import torch
        self.sia, self.n_slots, self.prm, self.stream = sia, n_slots, prm, stream
        sia.set_shard(rank, world)
        with torch.cuda.stream(stream):
            self.reduce_tensor = torch.zeros(sia.batch * REDUCE_DOUBLES, dtype=torch.float64, device="cuda")
        sia.set_reduce_buffer(self.reduce_tensor.data_ptr())

    def _on_stream(self):
        import torch
        return torch.cuda.stream(self.stream)

    def begin(self):
        self.sia.begin(self.n_slots, self.prm)

    def level_begin(self, level):
        self.sia.level_begin(level)

    def accumulate(self):
        self.sia.accumulate()

    def solve_update(self):
        self.sia.solve_update()

    def finish(self):
        self.sia.finish()


def gather_converged(ids: np.ndarray, mu: np.ndarray, sigma2: np.ndarray, xyz: np.ndarray, group=None,
                     device: Optional[str] = None) -> np.ndarray:
    """All-gather the variable-length set of converged seeds of every rank as packed records
    [seed_id, mu, sigma2, x, y, z] (float64), ordered by rank.  Counts first, then one padded
    all-gather (no per-iteration exchange on the seed path)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = device or ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    n = len(ids)
    rec = np.zeros((n, 6), dtype=np.float64)
    if n:
        rec[:, 0] = ids
        rec[:, 1] = mu
        rec[:, 2] = sigma2
        rec[:, 3:6] = xyz
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    mine = torch.zeros((cap, 6), dtype=torch.float64, device=dev)
    if n:
        mine[:n] = torch.from_numpy(rec).to(dev)
    parts = [torch.zeros((cap, 6), dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = [p[:c].cpu().numpy() for p, c in zip(parts, counts)]
    return np.concatenate(out, axis=0) if out else np.zeros((0, 6))


def gather_converged_device(ctx, seeds, id_offset: int, stream, group=None):
    """Device path of `gather_converged` for a hip.SeedBatch: the converged records are packed on the GPU
    (svo_hip_seed_compact_converged_dev, seed order), the counts are all-gathered, then one padded all-gather of
    device buffers (RCCL over xGMI).  Returns a [total, 6] float64 CUDA tensor ordered by rank, then by seed."""
    import ctypes as C

    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    with torch.cuda.stream(stream):
        rec = torch.empty((max(seeds.n, 1), 6), dtype=torch.float64, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.svo_hip_seed_compact_converged_dev(
        ctx.h, seeds.n, C.c_longlong(id_offset), C.c_void_p(seeds.status.ptr), C.c_void_p(seeds.mu.ptr),
        C.c_void_p(seeds.sigma2.ptr), C.c_void_p(seeds.xyz.ptr), C.c_void_p(rec.data_ptr()), C.c_void_p(cnt.data_ptr())),
        "seed_compact_converged")
    with torch.cuda.stream(stream):
        if world == 1:
            return rec[:int(cnt.item())]
        counts = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(world)]
        dist.all_gather(counts, cnt, group=group)
        counts = [int(c.item()) for c in counts]
        cap = max(max(counts), 1)
        parts = [torch.empty((cap, 6), dtype=torch.float64, device="cuda") for _ in range(world)]
        mine = rec[:cap] if rec.shape[0] >= cap else torch.cat([rec, rec.new_zeros((cap - rec.shape[0], 6))])
        dist.all_gather(parts, mine.contiguous(), group=group)
        return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


class GraphedAllreduceSolver:
    """The patch-sharded solve (`run_allreduce`) with the launch-bound inner loop captured in HIP graphs: one graph per
    pyramid level holding level_begin + n_iter x (accumulate, all-reduce of the normal equations, solve_update).
    The kernels skip frames that have finished, so replaying the full iteration budget keeps the reference's early-exit
    semantics.  Capture happens once per solver configuration; a solve is begin() + one replay per level + finish()."""

    def __init__(self, aligner, max_level: int, min_level: int, n_iter: int, stream, group=None):
        import torch
        import torch.distributed as dist
        self.aligner, self.stream = aligner, stream
        self.graphs = []
        # one eager solve first: creates the communicator and every lazily allocated buffer outside of capture
        with torch.cuda.stream(stream):
            run_allreduce(aligner, max_level, min_level, n_iter, group)
        torch.cuda.synchronize()
        aligner.begin()
        for level in range(max_level, min_level - 1, -1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                aligner.level_begin(level)
                for _ in range(n_iter):
                    aligner.accumulate()
                    dist.all_reduce(aligner.reduce_tensor, op=dist.ReduceOp.SUM, group=group)
                    aligner.solve_update()
            self.graphs.append(g)
        aligner.finish()
        torch.cuda.synchronize()

    def run(self):
        import torch
        with torch.cuda.stream(self.stream):
            self.aligner.begin()
            for g in self.graphs:
                g.replay()
            self.aligner.finish()
