"""An arithmetic identity the fused SparseImgAlign kernel relies on (android_svo_amd/csrc/svo_sia.hip, lpp_project and the
reference-patch precompute): the reference forms its bilinear weights as (float)((1.0 - su) * (1.0 - sv)) -- in double --
and the kernel forms them in f32.  For a position inside the image (u >= 3, so su = u - floor(u) is a multiple of 2^-22
below 1) 1 - su is exact in f32 and the product of two such numbers has at most 44 significant bits: exact in double,
rounded to f32 once on either path.  IEEE arithmetic is the same on the host, so the identity is checked here."""
import numpy as np


def test_f32_bilinear_weights_equal_the_double_form():
    rng = np.random.default_rng(7)
    for lo, hi in [(3, 4), (3, 8), (3, 64), (3, 1300), (1023, 1025)]:
        u = rng.uniform(lo, hi, 400000).astype(np.float32)
        v = rng.uniform(lo, hi, 400000).astype(np.float32)
        u[:2000] = np.float32(lo) + np.float32(2.0 ** -20) * rng.integers(0, 8, 2000).astype(np.float32)   # next to an integer
        su = (u - np.floor(u)).astype(np.float32)
        sv = (v - np.floor(v)).astype(np.float32)
        su64, sv64 = su.astype(np.float64), sv.astype(np.float64)
        ou, ov = (np.float32(1.0) - su).astype(np.float32), (np.float32(1.0) - sv).astype(np.float32)
        pairs = [(((1.0 - su64) * (1.0 - sv64)).astype(np.float32), (ou * ov).astype(np.float32)),
                 ((su64 * (1.0 - sv64)).astype(np.float32), (su * ov).astype(np.float32)),
                 (((1.0 - su64) * sv64).astype(np.float32), (ou * sv).astype(np.float32))]
        for ref, got in pairs:
            assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
            # the kernel halves its weights: a power of two commutes with the rounding
            assert np.array_equal((np.float32(0.5) * ref).view(np.uint32), (np.float32(0.5) * got).view(np.uint32))
