"""DepthFilter::updateSeed / computeTau (S/depth_filter.cpp:359-416) -- rows a-9 / a-10 stay "parity unpinned": depth_filter.cpp
cannot be built here (its logging macros need <android/log.h>, and no stand-in is written), so no reference binary has
confirmed the oracle's arithmetic beyond one known-answer vector.  What CAN be removed as a risk is a typo in the
restatement: the two functions are evaluated here a second time, independently of oracle/svo_oracle.c,

  * statement by statement in numpy with the C++ promotion rules written out (which operand is float, which literal makes
    an expression double, where a double is truncated to float) -- expected equal to the oracle bit for bit, up to the
    last-bit differences of numpy's exp / acos / sin against the C library's;
  * in extended precision (numpy.longdouble) with every intermediate exact to ~1e-19 -- the float results must sit within
    the error an f32 evaluation of the same formulas can accumulate.

1e5 random seeds / measurements each.  This does not pin the rows (they stay unpinned in README / DESIGN)."""
import numpy as np

from oracle import orc

F32, F64 = np.float32, np.float64
SQRT_2_PI = 1.41421356237309505            # depth_filter.cpp:360 -- sqrt(2), not sqrt(2 pi): kept
PI = 3.14159265                            # I/global.h:92


def update_seed_typed(x, tau2, a, b, mu, z_range, sigma2):
    """depth_filter.cpp:368-391 on float32 arrays; every line keeps the type the C++ expression has"""
    d = lambda v: v.astype(F64)
    f = lambda v: v.astype(F32)
    norm_scale = np.sqrt(sigma2 + tau2)                                          # float
    s2 = f(1.0 / (1.0 / d(sigma2) + 1.0 / d(tau2)))                              # 1. literals: double, truncated
    m = s2 * (mu / sigma2 + x / tau2)                                            # float
    exponent = -0.5 * ((d(x) - d(mu)) / d(norm_scale)) ** 2                      # normal_pdf: all double
    pdf = (1.0 / (d(norm_scale) * SQRT_2_PI)) * np.exp(exponent)
    C1 = f(d(a / (a + b)) * pdf)
    C2 = f(d(b / (a + b)) * 1.0 / d(z_range))
    nc = C1 + C2
    C1 = C1 / nc
    C2 = C2 / nc
    ab = a + b                                                                   # float
    ff = f(d(C1) * (d(a) + 1.0) / (d(ab) + 1.0) + d(C2 * a) / (d(ab) + 1.0))
    e = f(d(C1) * (d(a) + 1.0) * (d(a) + 2.0) / ((d(ab) + 1.0) * (d(ab) + 2.0)) +
          d(C2 * a * (a + F32(1.0)) / ((ab + F32(1.0)) * (ab + F32(2.0)))))
    mu_new = C1 * m + C2 * mu
    sigma2_new = C1 * (s2 + m * m) + C2 * (sigma2 + mu * mu) - mu_new * mu_new
    a_new = (e - ff) / (ff - e / ff)
    b_new = a_new * (F32(1.0) - ff) / ff
    return a_new, b_new, mu_new, sigma2_new


def update_seed_exact(x, tau2, a, b, mu, z_range, sigma2):
    L = np.longdouble
    x, tau2, a, b, mu, z_range, sigma2 = (v.astype(L) for v in (x, tau2, a, b, mu, z_range, sigma2))
    ns = np.sqrt(sigma2 + tau2)
    s2 = 1 / (1 / sigma2 + 1 / tau2)
    m = s2 * (mu / sigma2 + x / tau2)
    pdf = (1 / (ns * L(SQRT_2_PI))) * np.exp(-0.5 * ((x - mu) / ns) ** 2)
    C1 = a / (a + b) * pdf
    C2 = b / (a + b) / z_range
    nc = C1 + C2
    C1, C2 = C1 / nc, C2 / nc
    ff = C1 * (a + 1) / (a + b + 1) + C2 * a / (a + b + 1)
    e = C1 * (a + 1) * (a + 2) / ((a + b + 1) * (a + b + 2)) + C2 * a * (a + 1) / ((a + b + 1) * (a + b + 2))
    mu_new = C1 * m + C2 * mu
    sigma2_new = C1 * (s2 + m * m) + C2 * (sigma2 + mu * mu) - mu_new * mu_new
    a_new = (e - ff) / (ff - e / ff)
    return a_new, a_new * (1 - ff) / ff, mu_new, sigma2_new


def random_seeds(n, rng):
    a = rng.uniform(2, 40, n).astype(F32)
    b = rng.uniform(2, 40, n).astype(F32)
    mu = rng.uniform(0.2, 2.0, n).astype(F32)
    z_range = rng.uniform(0.5, 4.0, n).astype(F32)
    sigma2 = (z_range * z_range / rng.uniform(36, 4000, n)).astype(F32)
    tau2 = (10.0 ** rng.uniform(-6, -1, n)).astype(F32)
    x = (mu + rng.normal(0, 1, n) * np.sqrt(sigma2 + tau2) * rng.choice([0.3, 1.0, 4.0], n)).astype(F32)
    return x, tau2, a, b, mu, z_range, sigma2


def test_update_seed_against_two_independent_evaluations():
    n = 100000
    x, tau2, a, b, mu, z_range, sigma2 = random_seeds(n, np.random.default_rng(2025))
    got = np.array([orc.update_seed(x[i], tau2[i], (a[i], b[i], mu[i], z_range[i], sigma2[i])) for i in range(n)])
    o_a, o_b, o_mu, o_s2 = got[:, 0], got[:, 1], got[:, 2], got[:, 4]
    np.testing.assert_array_equal(got[:, 3], z_range)                            # z_range is never written
    # ---- typed evaluation: the same bits (numpy's exp may differ from the C library's in the last bit of a double, which
    # survives the truncation to float about once in 1e7)
    t_a, t_b, t_mu, t_s2 = update_seed_typed(x, tau2, a, b, mu, z_range, sigma2)
    for name, o, t in (("a", o_a, t_a), ("b", o_b, t_b), ("mu", o_mu, t_mu), ("sigma2", o_s2, t_s2)):
        same = (o.view(np.uint32) == t.view(np.uint32)) | (np.isnan(o) & np.isnan(t))
        assert same.mean() > 0.9995, (name, same.mean())
    # ---- extended precision: the float evaluation within the rounding an f32 chain of these formulas can accumulate.  mu and
    # sigma2 are sums of well-scaled terms; a and b come out of (e - f) / (f - e / f), a difference of nearly equal numbers,
    # whose relative error is that of e and f (~1e-7) times the cancellation factor |f| / |f - e / f|
    e_a, e_b, e_mu, e_s2 = (v.astype(F64) for v in update_seed_exact(x, tau2, a, b, mu, z_range, sigma2))
    ok = np.isfinite(e_a) & np.isfinite(o_a.astype(F64)) & (e_s2 > 0)
    assert ok.mean() > 0.99
    rel = lambda o, e: np.abs(o.astype(F64) - e) / np.maximum(np.abs(e), 1e-30)
    assert np.percentile(rel(o_mu, e_mu)[ok], 99.9) < 2e-6 and rel(o_mu, e_mu)[ok].max() < 1e-4
    # sigma2 = E[x^2] - mu^2 in f32: absolute error ~ eps * mu^2
    s2_err = np.abs(o_s2.astype(F64) - e_s2)[ok] / (e_mu[ok] ** 2 + e_s2[ok])
    assert np.percentile(s2_err, 99.9) < 2e-6, np.percentile(s2_err, 99.9)
    amp = (np.abs(e_a) + np.abs(e_b) + 1.0)                                      # cancellation grows with a + b
    assert np.percentile(rel(o_a, e_a)[ok] / amp[ok], 99) < 2e-5
    assert np.percentile(rel(o_b, e_b)[ok] / amp[ok], 99) < 2e-5
    # the known-answer vector the survey recorded from the reference itself
    kat = orc.update_seed(0.52, 0.01, orc.seed_init(2.0, 1.0))
    np.testing.assert_allclose(kat[[0, 1, 2, 4]], [10.4296455, 9.88126183, 0.511521995, 0.0118117034], rtol=3e-7)
    kt = update_seed_typed(*(np.array([v], F32) for v in (0.52, 0.01, 10, 10, 0.5, 1.0, 1.0 / 36)))
    np.testing.assert_allclose([kt[0][0], kt[1][0], kt[2][0], kt[3][0]], [10.4296455, 9.88126183, 0.511521995, 0.0118117034], rtol=3e-7)


def compute_tau_np(t, f, z, px_error_angle, dtype):
    """depth_filter.cpp:396-416"""
    t, f, z = t.astype(dtype), f.astype(dtype), z.astype(dtype)
    a = f * z[:, None] - t
    t_norm = np.sqrt((t * t).sum(axis=1))
    a_norm = np.sqrt((a * a).sum(axis=1))
    alpha = np.arccos((f * t).sum(axis=1) / t_norm)
    beta = np.arccos((a * -t).sum(axis=1) / (t_norm * a_norm))
    beta_plus = beta + dtype(px_error_angle)
    gamma_plus = dtype(PI) - alpha - beta_plus
    return t_norm * np.sin(beta_plus) / np.sin(gamma_plus) - z


def test_compute_tau_against_two_independent_evaluations():
    n = 100000
    rng = np.random.default_rng(77)
    t = rng.normal(0, 0.1, (n, 3))
    f = rng.normal(0, 0.3, (n, 3)) + [0, 0, 1]
    f /= np.linalg.norm(f, axis=1)[:, None]
    z = rng.uniform(0.5, 8.0, n)
    pea = 2.0 * np.arctan(1.0 / (2.0 * 500.0))
    T = np.zeros(7); T[6] = 1.0
    got = np.empty(n)
    for i in range(n):
        T[:3] = t[i]
        got[i] = orc.compute_tau(T, f[i], z[i], pea)
    d = compute_tau_np(t, f, z, pea, np.float64)
    e = compute_tau_np(t, f, z, pea, np.longdouble).astype(np.float64)
    # tau = z_plus - z: compare relative to the two depths.  With almost no parallax gamma_plus is a small angle, z_plus is
    # large and a last-bit difference of acos / sin (numpy's against the C library's) is amplified by 1 / sin(gamma_plus):
    # the bulk must agree to rounding, the few ill-conditioned samples to the conditioning
    scale = np.abs(got) + z
    err_d, err_e = np.abs(got - d) / scale, np.abs(got - e) / scale
    assert (err_d == 0).mean() > 0.5 and np.percentile(err_d, 99.9) < 1e-12 and err_d.max() < 1e-7, (np.percentile(err_d, 99.9), err_d.max())
    assert np.percentile(err_e, 99.9) < 1e-11 and err_e.max() < 1e-7, (np.percentile(err_e, 99.9), err_e.max())
    assert (got > 0).mean() > 0.99
