"""CPU: Ceres' projected Armijo line search as restated in the oracle (oracle/ctvo.c: ctvo_solve, poly_*).  External source
(Ceres 1.14 line_search.cc / polynomial.cc, not under /root/reference): the interpolation step is checked against NumPy's
polynomial machinery, the search itself through its observable contract (sufficient decrease, contraction bounds, counters)."""
import numpy as np
import pytest


def _ref_step(x, v, g, lo, hi):
    """minimiser over [lo, hi] of the polynomial interpolating values and gradients at the sample points (Ceres
    MinimizeInterpolatingPolynomial: midpoint, ends, real parts of the derivative's roots, the samples themselves)"""
    n = len(x); deg = 2 * n - 1
    rows, rhs = [], []
    for xi, vi, gi in zip(x, v, g):
        rows.append([xi ** (deg - j) for j in range(deg + 1)]); rhs.append(vi)
        rows.append([(deg - j) * xi ** (deg - j - 1) if j < deg else 0.0 for j in range(deg + 1)]); rhs.append(gi)
    c = np.linalg.solve(np.array(rows), np.array(rhs))
    cands = [0.5 * (lo + hi), lo, hi] + [r.real for r in np.roots(np.polyder(c)) if lo <= r.real <= hi] + [xi for xi in x if lo <= xi <= hi]
    vals = [np.polyval(c, z) for z in cands]
    return c, min(vals)


@pytest.mark.parametrize("ns", [2, 3])
def test_interpolating_polynomial_minimiser(oracle, ns):
    rng = np.random.default_rng(ns)
    for _ in range(300):
        x = np.concatenate([[0.0], np.sort(rng.uniform(0.05, 1.0, ns - 1))[::-1]])
        v = rng.normal(size=ns); g = rng.normal(size=ns)
        lo, hi = 1e-3 * x[1], 0.6 * x[1]
        got = oracle.ls_interpolate(x, v, g, lo, hi)
        c, best = _ref_step(x, v, g, lo, hi)
        assert lo <= got <= hi
        assert abs(np.polyval(c, got) - best) <= 1e-9 * max(1.0, abs(best))


def test_line_search_only_acts_on_bounded_problems_and_keeps_sufficient_decrease(cv, oracle):
    """config-3 windows (rolling-shutter stress, line delay estimated from 0 inside [0, 35 us]) make the trust-region step
    overshoot: the search shortens it.  With a fixed line delay the reduced program has no bounds and Ceres never searches;
    switching the search off reproduces round 1's alpha = 1 behaviour and a different iterate."""
    hits = 0
    for seed in range(1000, 1012):
        w0 = cv.synth.make_window("config3", seed=seed)
        a = w0.copy(); sa = oracle.OracleWindow(a).solve(15)
        assert sa.num_line_search_steps >= sa.num_line_search_reduced >= 0
        assert all(sa.cost_hist[i + 1] <= sa.cost_hist[i] * (1 + 1e-12) for i in range(sa.iterations))   # monotone: accepted steps only lower the cost
        f = w0.copy(); f.fix_ld = True; f.ld = 2.0e-5
        sf = oracle.OracleWindow(f).solve(15)
        assert sf.num_line_search_steps == 0 and sf.num_line_search_reduced == 0
        if sa.num_line_search_reduced:
            hits += 1
            oracle.set_line_search(False)
            try:
                b = w0.copy(); sb = oracle.OracleWindow(b).solve(15)
            finally:
                oracle.set_line_search(True)
            assert sb.num_line_search_reduced == 0
            assert cv.rel_state_error(a, b)["state"] > 1e-9      # the search changed the iterate sequence
    assert hits >= 2


def test_synthetic_generator_is_deterministic_and_valid(cv):
    a = cv.synth.make_window("config2", seed=1234)
    b = cv.synth.make_window("config2", seed=1234)
    for name in ("quat", "pos", "rho", "imu_gyro", "v_pi", "v_pj", "v_rowi", "v_rowj", "v_lm"):
        np.testing.assert_array_equal(getattr(a, name), getattr(b, name))
    assert a.K == 24 and a.L == 200 and a.M == 2000 and 600 < a.V < 1100
    assert a.v_rowi.min() >= 0 and a.v_rowj.max() < 1024
    assert np.all(a.v_tj > a.v_ti)                               # anchor first, later observations after it
    assert np.all(np.bincount(a.v_lm, minlength=a.L) >= 2)       # every landmark is a candidate (>= 3 observations)
