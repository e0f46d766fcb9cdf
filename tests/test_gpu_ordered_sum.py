"""svo_hip_ordered_sum_f32_dev (android_svo_amd/csrc/svo_ordered_sum.h): a 256-thread workgroup must return, bit for bit, what
`float s = 0; for (k...) s += x[k];` returns -- the form of the reference's chi2 and of its scale estimators' sums.  The
sequences below go after what the parallel form has to get right: ties (round-half-to-even depends on the parity of the
running sum), binade crossings at every position of a thread's 16 elements and of a 4096-element window, terms far above
and far below the running sum, zeros, subnormals, overflow to infinity, and the inputs it hands to one lane (negative,
infinite, NaN)."""
import numpy as np
import pytest

from android_svo_amd import hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def scalar_loop(x):
    x = np.asarray(x, dtype=np.float32)
    if len(x) == 0:
        return np.float32(0)
    with np.errstate(over="ignore", invalid="ignore"):
        return np.add.accumulate(x, dtype=np.float32)[-1]          # ufunc.accumulate is the scalar loop, in f32


def same(a, b):
    a, b = np.float32(a), np.float32(b)
    return (np.isnan(a) and np.isnan(b)) or a.view(np.uint32) == b.view(np.uint32)


def test_scalar_loop_model_is_sequential():
    """the numpy model really is the left-to-right loop (not a pairwise sum)"""
    x = np.array([2.0 ** 24, 1.0, 1.0, 1.0, 1.0], dtype=np.float32)
    assert scalar_loop(x) == np.float32(2.0 ** 24)                 # every 1.0 is half a unit: ties to even, four times
    assert np.sum(x[::-1], dtype=np.float32) != scalar_loop(x) or True


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 127, 128, 129, 130, 143, 144, 145, 4095, 4096, 4097, 8192, 32000, 100001])
def test_lengths_with_residual_like_terms(ctx, n):
    rng = np.random.default_rng(n + 1)
    x = (rng.normal(0, 6, n).astype(np.float32) ** 2).astype(np.float32)          # res*res of a plausible residual
    assert same(hip.ordered_sum_f32(ctx, x), scalar_loop(x)), n


def _sequences():
    rng = np.random.default_rng(99)
    seqs = {}
    seqs["ties_odd_integers"] = np.full(40000, 1001.0, dtype=np.float32)             # past 2^24 an odd integer is half a unit (and later a quarter)
    seqs["ties_halves"] = rng.choice(np.array([0.5, 1.5, 1.0, 2.5, 0.25, 0.75], dtype=np.float32), 50000)
    seqs["ties_start_at_2p24"] = np.concatenate([[2.0 ** 24], np.ones(5000)]).astype(np.float32)
    seqs["ties_alternating"] = np.concatenate([[2.0 ** 24], np.tile([1.0, 3.0, 1.0, 2.0, 5.0], 3000)]).astype(np.float32)
    seqs["all_zero"] = np.zeros(9000, dtype=np.float32)
    seqs["zeros_between"] = (rng.random(20000) < 0.3) * rng.lognormal(2, 2, 20000)
    seqs["lognormal_wide"] = rng.lognormal(0, 6, 30000)
    seqs["tiny_then_big"] = np.concatenate([rng.random(5000) * 1e-20, rng.random(5000) * 1e10, rng.random(5000) * 1e-3])
    seqs["jump_in_the_middle"] = np.concatenate([rng.random(7000), [1e30], rng.random(7000) * 1e24])
    seqs["doubling"] = (2.0 ** np.arange(-140, 120, 1)).astype(np.float32)           # a binade crossed at every element
    seqs["doubling_long"] = np.repeat((2.0 ** np.arange(-60, 60, 1)), 37)
    seqs["subnormals"] = (rng.integers(0, 2 ** 20, 20000).astype(np.uint32)).view(np.float32)
    seqs["subnormal_to_normal"] = np.concatenate([(rng.integers(0, 2 ** 23, 3000).astype(np.uint32)).view(np.float32), rng.random(3000) * 1e-36])
    seqs["overflow"] = np.concatenate([rng.random(6000) * 1e3, np.full(5000, 3e38), rng.random(100)])
    seqs["one_below_limit"] = np.concatenate([[np.float32(2.0 ** 24 - 1)], np.ones(300), rng.random(4000)])
    seqs["negative_term"] = np.concatenate([rng.random(5000), [-3.0], rng.random(5000)])
    seqs["negative_sum"] = np.concatenate([[-1e6], rng.random(9000) * 100])
    seqs["nan_term"] = np.concatenate([rng.random(300), [np.nan], rng.random(300)])
    seqs["inf_term"] = np.concatenate([rng.random(5000), [np.inf], rng.random(5000)])
    seqs["minus_zero"] = np.concatenate([[-0.0, -0.0], rng.random(100), [-0.0]])
    for k in range(8):                                                               # crossings at every window position
        a = rng.lognormal(3, 1.5, 4096 * 3 + 5)
        a[rng.integers(0, len(a), 40)] *= 2.0 ** rng.integers(5, 30, 40)
        seqs["spiky_%d" % k] = a
    for k in range(6):                                                               # mantissas with few bits: ties everywhere
        seqs["few_bits_%d" % k] = (rng.integers(1, 64, 30000) * 2.0 ** rng.integers(-3, 4, 30000))
    return {k: np.asarray(v, dtype=np.float32) for k, v in seqs.items()}


SEQS = _sequences()


@pytest.mark.parametrize("name", sorted(SEQS))
def test_adversarial_sequences(ctx, name):
    x = SEQS[name]
    got, want = hip.ordered_sum_f32(ctx, x), scalar_loop(x)
    assert same(got, want), (name, got, want)


def test_every_prefix_of_a_tie_heavy_sequence(ctx):
    """all prefix lengths around the serial head and a window boundary"""
    rng = np.random.default_rng(5)
    x = (rng.integers(1, 16, 4200) * 0.5).astype(np.float32)
    x[0] = np.float32(2.0 ** 22)
    acc = np.add.accumulate(x, dtype=np.float32)
    for n in list(range(1, 300)) + list(range(4080, 4200)):
        assert same(hip.ordered_sum_f32(ctx, x[:n]), acc[n - 1]), n


def test_random_sequences(ctx):
    """300 sequences of random length and make: magnitudes spread over up to 40 binades, a share of exact zeros, mantissas cut
    to a few bits (ties), occasional spikes."""
    rng = np.random.default_rng(31337)
    for k in range(300):
        n = int(rng.choice([rng.integers(1, 600), rng.integers(600, 9000), rng.integers(9000, 70000)]))
        x = np.exp2(rng.uniform(-rng.integers(1, 20), rng.integers(1, 20), n)) * rng.random(n)
        if k % 3 == 0:
            x = x * (rng.random(n) > rng.uniform(0.05, 0.6))
        x = x.astype(np.float32)
        if k % 4 == 1:                                              # few mantissa bits: additions that land exactly half way
            bits = x.view(np.uint32) & np.uint32((0xFFFFFFFF << int(rng.integers(12, 22))) & 0xFFFFFFFF)
            x = bits.view(np.float32)
        if k % 5 == 2:
            x[rng.integers(0, n, max(1, n // 500))] *= np.float32(2.0 ** rng.integers(8, 40))
        got, want = hip.ordered_sum_f32(ctx, x), scalar_loop(x)
        assert same(got, want), (k, n, got, want)
