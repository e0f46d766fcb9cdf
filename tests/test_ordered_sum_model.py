"""The arithmetic behind android_svo_amd/csrc/svo_ordered_sum.h, modelled in integers on the CPU: inside a binade the f32
running sum is an integer count m of units u, a non-negative term adds q whole units plus a carry that is 0 below half a unit, 1
above it and, on a tie, whatever makes the new count even -- so a term's effect depends on its predecessors through the PARITY
of m only, and everything up to the first addition that leaves the binade can be computed from (q, kind) and a running parity
(on the device: a prefix scan); that one addition is made by the adder and the procedure starts again in the new binade.
The model walks the sequence binade by binade exactly as the kernel does and must return what the scalar f32 loop returns."""
import numpy as np
import pytest

LIMIT = 1 << 24


def unit_of(s):
    bits = int(np.float32(s).view(np.uint32)) & 0x7FFFFFFF
    ef, frac = bits >> 23, bits & 0x7FFFFF
    if ef <= 1:
        return -149, (frac | 0x800000) if ef else frac
    return ef - 150, frac | 0x800000


def make(m, ue):
    if m < 0x800000:
        return np.uint32(m).view(np.float32)
    if m == LIMIT:
        return np.uint32((ue + 151) << 23).view(np.float32)
    return np.uint32(((ue + 150) << 23) | (m & 0x7FFFFF)).view(np.float32)


def decode(x, ue):
    bits = int(np.float32(x).view(np.uint32)) & 0x7FFFFFFF
    ef, M = bits >> 23, bits & 0x7FFFFF
    ex = -149
    if ef:
        M |= 0x800000
        ex = ef - 150
    shift = ex - ue
    if M == 0:
        return 0, 0
    if shift >= 0:
        return M << shift, 0
    sh = -shift
    if sh >= 26:
        return 0, 0
    q = 0 if sh >= 24 else M >> sh
    r, half = M & ((1 << sh) - 1), 1 << (sh - 1)
    return q, (0 if r < half else 1 if r > half else 2)


def ordered_sum_model(x):
    x = np.asarray(x, dtype=np.float32)
    S = np.float32(0)
    k, n, phases = 0, len(x), 0
    while k < n:
        phases += 1
        ue, m = unit_of(S)
        crossed = False
        while k < n:                                   # (the device does this stretch as one prefix scan over (q, kind))
            q, kind = decode(x[k], ue)
            c = ((m + q) & 1) if kind == 2 else kind
            if m + q + c >= LIMIT:                     # this addition leaves the binade: the adder makes it
                with np.errstate(over="ignore"):
                    S = np.float32(make(m, ue) + x[k])
                k += 1
                crossed = True
                break
            m += q + c
            k += 1
        if not crossed:
            S = make(m, ue)
        if not np.isfinite(S):
            return np.float32(np.inf), phases          # inf + (finite, >= 0) stays inf
    return np.float32(S), phases


def scalar_loop(x):
    x = np.asarray(x, dtype=np.float32)
    if len(x) == 0:
        return np.float32(0)
    with np.errstate(over="ignore"):
        return np.add.accumulate(x, dtype=np.float32)[-1]


def _cases():
    rng = np.random.default_rng(2)
    yield "residuals", (rng.normal(0, 6, 5000).astype(np.float32) ** 2).astype(np.float32)
    yield "ties_odd_integers", np.full(20000, 1001.0, dtype=np.float32)
    yield "ties_from_2p24", np.concatenate([[2.0 ** 24], np.tile([1.0, 3.0, 1.0, 2.0, 5.0], 800)]).astype(np.float32)
    yield "halves", rng.choice(np.array([0.5, 1.5, 1.0, 2.5, 0.25, 0.75], dtype=np.float32), 20000)
    yield "zeros_between", ((rng.random(8000) < 0.3) * rng.lognormal(2, 2, 8000)).astype(np.float32)
    yield "wide", rng.lognormal(0, 6, 8000).astype(np.float32)
    yield "doubling", (2.0 ** np.arange(-140, 120)).astype(np.float32)
    yield "subnormals", rng.integers(0, 2 ** 20, 6000).astype(np.uint32).view(np.float32)
    yield "subnormal_to_normal", np.concatenate([rng.integers(0, 2 ** 23, 2000).astype(np.uint32).view(np.float32), (rng.random(2000) * 1e-36).astype(np.float32)])
    yield "overflow", np.concatenate([rng.random(500) * 1e3, np.full(400, 3e38), rng.random(50)]).astype(np.float32)
    yield "few_bits", (rng.integers(1, 64, 8000) * 2.0 ** rng.integers(-3, 4, 8000)).astype(np.float32)
    yield "one_below_limit", np.concatenate([[np.float32(2.0 ** 24 - 1)], np.ones(300), rng.random(1000)]).astype(np.float32)


@pytest.mark.parametrize("name,x", list(_cases()), ids=[n for n, _ in _cases()])
def test_model_equals_the_scalar_loop(name, x):
    got, phases = ordered_sum_model(x)
    want = scalar_loop(x)
    assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32), (name, got, want)
    # the restarts are rare: at most one per binade the sum visits (a few dozen, however long the sequence), except for the
    # sequence built to cross a binade at every element
    assert phases <= (len(x) if name == "doubling" else 64), (name, phases)
