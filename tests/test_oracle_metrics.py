"""CPU checks of the metrics oracle (oracle/metrics.py, follows mcmc_utils.py:28-111) against literal per-pair loops
and closed forms."""
import numpy as np

from oracle import metrics, targets


def _disc_literal(x, x_, g, g_, d, b):
    diff = x - x_
    r2 = diff @ diff
    return (-4 * b * (b + 1) * r2 / (1 + r2) ** (b + 2) + 2 * b * (d + (g - g_) @ diff) / (1 + r2) ** (1 + b) + (g @ g_) / (1 + r2) ** b)


def test_stein_disc_matches_literal_pair_loop():
    rng = np.random.default_rng(0)
    T, d = 23, 5
    X = rng.standard_normal((T, d))
    grad = lambda x: -x                                  # standard normal target
    u, v = metrics.stein_disc(X, grad, block=7)
    G = grad(X)
    tot = sum(_disc_literal(X[i], X[j], G[i], G[j], d, 0.5) for i in range(T) for j in range(T))
    diag = sum(_disc_literal(X[i], X[i], G[i], G[i], d, 0.5) for i in range(T))
    np.testing.assert_allclose(u, (tot - diag) / (T * (T - 1)), rtol=1e-12)
    np.testing.assert_allclose(v, tot / T ** 2, rtol=1e-12)


def test_stein_v_statistic_is_nonnegative_and_small_for_exact_samples():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((2000, 3))
    u, v = metrics.stein_disc(X, lambda x: -x)
    assert v >= 0 and abs(u) < 0.02                     # unbiased estimate of 0 for exact samples
    Xs = X + 1.5                                         # shifted samples: clearly positive discrepancy
    assert metrics.stein_disc(Xs, lambda x: -x)[0] > 10 * abs(u)


def test_mmd_literal_and_properties():
    rng = np.random.default_rng(2)
    X, Y = rng.standard_normal((40, 4)), rng.standard_normal((40, 4)) + 0.5
    k = lambda a, b: np.exp(-0.5 * ((a - b) ** 2).sum())
    m = 40
    dx = sum(k(X[i], X[j]) for i in range(m) for j in range(m)) - m
    dy = sum(k(Y[i], Y[j]) for i in range(m) for j in range(m)) - m
    dxy = sum(k(X[i], Y[j]) for i in range(m) for j in range(m))
    np.testing.assert_allclose(metrics.max_mean_disc(X, Y, block=16), dx / (m * m - m) - 2 * dxy / (m * m) + dy / (m * m - m), rtol=1e-12)
    np.testing.assert_allclose(metrics.max_mean_disc(X, Y), metrics.max_mean_disc(Y, X), rtol=1e-12)      # symmetric
    assert metrics.max_mean_disc(X, Y) > metrics.max_mean_disc(X, X[::-1].copy())


def test_stein_disc_phi4_gradient_path():
    dist = targets.PhiFour(16)
    rng = np.random.default_rng(3)
    X = rng.uniform(-1, 1, (64, 16))
    u, v = metrics.stein_disc(X, dist.grad_logprob)
    assert np.isfinite(u) and np.isfinite(v) and v > 0
