"""CPU checks of the adaptive tempered SMC restatement (oracle/smc.py; bblackjax/smc/*, exe_others.py:79-111) against
closed forms and independent tools; the reference ships no vectors for it (parity unpinned)."""
import numpy as np
from scipy.special import logsumexp as sp_lse

from oracle import loop, prng, smc, targets


def test_logsumexp_and_log_ess():
    rng = np.random.default_rng(0)
    a = rng.standard_normal(500) * 30
    assert abs(smc.logsumexp(a) - sp_lse(a)) < 1e-12
    w = np.exp(a - sp_lse(a))
    assert abs(np.exp(smc.log_ess(a)) - 1.0 / (w * w).sum()) < 1e-9 / (w * w).sum()     # ESS = 1 / sum w^2 (ess.py:28-43)
    assert abs(np.exp(smc.log_ess(np.zeros(64))) - 64) < 1e-9


def test_dichotomy_branches():
    f = lambda d: 0.3 - d                                   # decreasing, root 0.3
    r = smc.dichotomy(f, 0.0, 0.0, 1.0)
    assert 0.3 - 1e-4 <= r <= 0.3                           # returns the LEFT end of the final bracket (solver.py:74)
    assert smc.dichotomy(lambda d: 1.0 - 0.1 * d, 0.0, 0.0, 1.0) == 1.0      # f(max) > 0: take max_delta (:76-78)
    assert np.isnan(smc.dichotomy(lambda d: -1.0 - d, 0.0, 0.0, 1.0))       # f(min) <= 0: nan (:80)


def test_ess_solver_hits_the_target():
    rng = np.random.default_rng(1)
    ll = rng.standard_normal(2000) * 50 - 100
    for target in (0.5, 0.9, 0.95):
        d = smc.ess_solver(ll, target, 1.0)
        assert 0 < d < 1
        ess = np.exp(smc.log_ess(-d * ll))                  # as written: weights exp(-delta * loglik) (ess.py:83)
        assert abs(ess / 2000 - target) < 2e-3
    assert smc.ess_solver(np.full(100, -3.0), 0.9, 0.25) == 0.25            # flat weights: ESS = n > target -> max_delta


def test_systematic_resampling():
    key = prng.PRNGKey(4)
    n = 257
    assert np.array_equal(smc.systematic(key, np.ones(n) / n, n), np.arange(n))          # uniform weights: identity
    w = np.zeros(n); w[17] = 1.0
    assert np.array_equal(smc.systematic(key, w, n), np.full(n, 17))
    rng = np.random.default_rng(2)
    w = rng.random(n); w /= w.sum()
    idx = smc.systematic(key, w, n)
    assert np.all(np.diff(idx) >= 0) and idx.min() >= 0 and idx.max() <= n - 1
    counts = np.bincount(idx, minlength=n)
    assert np.all(np.abs(counts - n * w) < 1.0 + 1e-9)      # systematic resampling: |N_i - n w_i| < 1


def test_smc_run_tempers_towards_one_and_is_deterministic():
    dist = targets.GaussianMixture(8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4)
    args = loop.default_args(example="4-mode", dim=2, num_chain=128, learning_iter=40, step_size=0.2, seed=3, eval_iter=2)
    a = smc.run(dist, args)
    b = smc.run(dist, args)
    np.testing.assert_array_equal(a["lmbdas"], b["lmbdas"])
    np.testing.assert_array_equal(a["samples"], b["samples"])
    lm = a["lmbdas"]
    assert np.all(np.diff(lm) >= 0) and 0 < lm[0] < lm[-1] <= 1.0
    assert a["samples"].shape == (2 * 128, 2)
    w = a["state"]["weights"]
    assert abs(w.sum() - 1.0) < 1e-12


def test_the_other_resampling_schemes_are_resampling_schemes():
    """stratified / multinomial / residual (resampling.py:55-121) restated: indices are in range and sorted where the scheme
    sorts them, every particle's expected count is N w (checked over many keys), residual repeats each particle at least
    floor(N w) times, and jax.random.permutation's restatement is a permutation that a second call with the same key repeats."""
    from oracle import smc
    rng = np.random.default_rng(5)
    n = 64
    w = rng.random(n) ** 3
    w /= w.sum()
    counts = {k: np.zeros(n) for k in ("stratified", "multinomial", "residual")}
    reps = 300
    for r in range(reps):
        key = prng.PRNGKey(1000 + r)
        for name in counts:
            idx = getattr(smc, name)(key, w, n)
            assert idx.shape == (n,) and idx.min() >= 0 and idx.max() < n
            if name != "residual":
                assert (np.diff(idx) >= 0).all()                      # searchsorted of an increasing sequence
            else:
                assert (np.bincount(idx, minlength=n) >= np.floor(n * w)).all()
            counts[name] += np.bincount(idx, minlength=n)
    for name, c in counts.items():
        err = np.abs(c / reps - n * w)
        assert err.max() < 4 * np.sqrt(n * w.max() / reps) + 0.05, (name, err.max())
    p = smc.permutation(prng.PRNGKey(3), np.arange(1000))
    assert sorted(p.tolist()) == list(range(1000)) and (p != np.arange(1000)).sum() > 900
    np.testing.assert_array_equal(p, smc.permutation(prng.PRNGKey(3), np.arange(1000)))
