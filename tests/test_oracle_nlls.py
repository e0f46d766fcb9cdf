"""The oracle's Levenberg-Marquardt and robust-cost branches (oracle/svo_oracle.c: sia_optimize_lm, scale_*, robust_weight)
against SparseImgAlign::run executed by the reference's OWN compiled code with method_ = LevenbergMarquardt and / or a robust
cost set through its setRobustCostFunction (tests/golden/sia_nlls_ref.npz, made by oracle/gen_golden.py --nlls-only from
oracle/ref/ref_objects.cpp).  Bit for bit: pose, H_ (for LM the damped matrix of the last trial), chi2_, stop_, the tracked
count and scale_."""
import numpy as np
import pytest

from oracle import gen_golden, orc

CASES = {c[0]: c for c in gen_golden.SIA_REF_CASES}
PAIRS = [(n, c) for n in gen_golden.SIA_NLLS_CASES for c in gen_golden.SIA_NLLS_COMBOS]


@pytest.mark.parametrize("name,combo", PAIRS, ids=[gen_golden.nlls_key(n, c) for n, c in PAIRS])
def test_oracle_equals_the_reference_run(golden, name, combo):
    g = golden("sia_nlls_ref.npz")
    _, kw, max_level, min_level, n_iter = CASES[name]
    fp = gen_golden.make_sia_case(kw)
    o = orc.sparse_img_align(fp, max_level, min_level, n_iter, method=combo[0], scale_estimator=combo[1], weight_function=combo[2])
    k = gen_golden.nlls_key(name, combo)
    assert np.array_equal(np.array(o.T_cur_w), g[k + "_T"])
    assert np.array_equal(np.array(o.H), g[k + "_H"])
    assert o.chi2 == float(g[k + "_chi2"])
    assert o.n_tracked == int(g[k + "_n_tracked"])
    assert o.stop == int(g[k + "_stop"])
    assert np.float32(o.scale) == np.float32(g[k + "_scale_mu_nu"][0])


def test_unit_scale_switches_the_weights_off():
    """setRobustCostFunction(UnitScale, anything) leaves use_weights_ false (nlls_solver_impl.hpp:258-262): the run is the
    plain Gauss-Newton run whatever the weight function."""
    _, kw, max_level, min_level, n_iter = CASES["c0_200"]
    fp = gen_golden.make_sia_case(kw)
    a = orc.sparse_img_align(fp, max_level, min_level, n_iter)
    b = orc.sparse_img_align(fp, max_level, min_level, n_iter, scale_estimator=0, weight_function=2)
    assert np.array_equal(np.array(a.T_cur_w), np.array(b.T_cur_w)) and a.chi2 == b.chi2
