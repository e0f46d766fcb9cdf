"""6000 random non-negative f32 sequences (lengths 1 .. 120 000; magnitudes over up to 60 binades, zeros, mantissas cut to a few
bits, spikes) through svo_hip_ordered_sum_f32_dev against numpy's left-to-right f32 accumulate; prints the number of mismatches.
GPU box: python tools/ordered_sum_stress.py [seed]"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from android_svo_amd import hip
ctx = hip.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
N = 6000
for k in range(N):
    n = int(rng.choice([rng.integers(1, 600), rng.integers(600, 9000), rng.integers(9000, 120000)]))
    x = np.exp2(rng.uniform(-rng.integers(1, 30), rng.integers(1, 30), n)) * rng.random(n)
    if k % 3 == 0: x = x * (rng.random(n) > rng.uniform(0.05, 0.9))
    x = x.astype(np.float32)
    if k % 4 == 1:
        x = (x.view(np.uint32) & np.uint32((0xFFFFFFFF << int(rng.integers(10, 23))) & 0xFFFFFFFF)).view(np.float32)
    if k % 5 == 2:
        x[rng.integers(0, n, max(1, n // 300))] *= np.float32(2.0 ** rng.integers(8, 60))
    got = hip.ordered_sum_f32(ctx, x)
    with np.errstate(over="ignore"):
        want = np.add.accumulate(x, dtype=np.float32)[-1]
    if not ((np.isnan(got) and np.isnan(want)) or got.view(np.uint32) == want.view(np.uint32)):
        bad += 1
        print("MISMATCH", k, n, got, want)
print("sequences", N, "mismatches", bad)
