"""Pins the oracle's PRNG: Random123 KATs for the block function, jax-documented split values."""
import numpy as np

from oracle import prng


def _tf(k, c):
    y0, y1 = prng.threefry2x32(k, np.array([c[0]], dtype=np.uint32), np.array([c[1]], dtype=np.uint32))
    return int(y0[0]), int(y1[0])


def test_threefry_random123_kat():
    # Random123 kat_vectors, threefry2x32 20 rounds (also used by jax's own test-suite)
    assert _tf((0, 0), (0, 0)) == (0x6B200159, 0x99BA4EFE)
    assert _tf((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF)) == (0x1CB996FC, 0xBB002BE7)
    assert _tf((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3)) == (0xC4923A9C, 0x483DF7A0)


def test_split_matches_jax_documentation_values():
    # values printed in the jax documentation for random.split(PRNGKey(0)) / PRNGKey(42)
    np.testing.assert_array_equal(prng.split(prng.PRNGKey(0), 2),
                                  [[4146024105, 967050713], [2718843009, 1272950319]])
    np.testing.assert_array_equal(prng.split(prng.PRNGKey(42), 2),
                                  [[2465931498, 3679230171], [255383827, 267815257]])


def test_row_helpers_equal_per_key_calls():
    keys = prng.split(prng.PRNGKey(7), 5)
    nr = prng.normal_rows(keys, 6)
    ur = prng.uniform_rows(keys, 6)
    us = prng.uniform_rows(keys)
    sr = prng.split_rows(keys, 4)
    for i, k in enumerate(keys):
        np.testing.assert_array_equal(nr[i], prng.normal(k, (6,)))
        np.testing.assert_array_equal(ur[i], prng.uniform(k, (6,)))
        assert us[i] == prng.uniform(k, ())
        np.testing.assert_array_equal(sr[i], prng.split(k, 4))
    np.testing.assert_array_equal(prng.split_at(prng.PRNGKey(3), 9, [0, 4, 8]), prng.split(prng.PRNGKey(3), 9)[[0, 4, 8]])
    np.testing.assert_array_equal(prng.split_at(prng.PRNGKey(3), 8, np.arange(8)), prng.split(prng.PRNGKey(3), 8))


def test_sharded_draws_equal_full_draw():
    k = prng.PRNGKey(11)
    full = prng.normal(k, (10, 3))
    part = prng.normal(k, (10, 3), start=4 * 3, count=5 * 3).reshape(5, 3)
    np.testing.assert_array_equal(full[4:9], part)


def test_normal_moments_and_range():
    x = prng.normal(prng.PRNGKey(5), (200000,))
    assert abs(x.mean()) < 0.01 and abs(x.std() - 1) < 0.01
    u = prng.uniform(prng.PRNGKey(5), (200000,))
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.005


def test_dirichlet_restatement():
    """jax.random.dirichlet (multi_modal.py:45: the 16-mode mixture's weights) = softmax of log-space gamma draws on
    split(key, n).  jax itself is absent: the sampler is checked as a SAMPLER (Kolmogorov-Smirnov against scipy's gamma law
    for boosted and unboosted shapes), the oracle's and the host module's restatements against each other bit for bit."""
    from scipy import stats
    from mfm_amd import random as jr
    key = prng.split(prng.PRNGKey(0), 3)[2]
    w = prng.dirichlet(key, 4.0 * np.ones(16))
    np.testing.assert_array_equal(w, jr.dirichlet(key, 4.0 * np.ones(16)))
    assert abs(w.sum() - 1.0) < 1e-14 and (w > 0).all()
    for a in (0.3, 1.0, 4.0):
        keys = prng.split(prng.PRNGKey(7), 600)
        x = np.exp([prng.gamma_log(k, a) for k in keys])
        assert stats.kstest(x, stats.gamma(a).cdf).pvalue > 1e-3, a
    d = np.array([prng.dirichlet(k, 4.0 * np.ones(4)) for k in prng.split(prng.PRNGKey(9), 300)])
    np.testing.assert_allclose(d.mean(0), 0.25, atol=0.02)                         # E[w_i] = alpha_i / sum(alpha)
    np.testing.assert_allclose(d.var(0), 0.25 * 0.75 / 17.0, rtol=0.3)             # Var[w_i] = a_i (a_0 - a_i) / (a_0^2 (a_0 + 1))


def test_32_bit_draws_match_values_published_in_the_jax_documentation():
    """Outside pins of the draw conventions (counter layout of random_bits, mantissa fill, the normal's range and erfinv):
    values printed in jax's own documentation for its default 32-bit mode -- the "Pseudorandom numbers" tutorial
    (``random.normal(PRNGKey(42))``, ``random.normal(subkey)`` after one split) and the ``jax.random`` module page
    (``random.uniform(PRNGKey(0))``).  The 64-bit path the reference uses (jax_enable_x64) applies the same conventions to
    64-bit words; it has no published values."""
    assert prng.uniform32(prng.PRNGKey(0)) == np.float32(0.41845703)
    key = prng.PRNGKey(42)
    # (published: -0.18471177; float32 erfinv implementations differ in the last two bits)
    assert abs(float(prng.normal32(key)) - (-0.18471177)) < 1e-7
    _, subkey = prng.split(key)
    assert prng.normal32(subkey) == np.float32(1.3694694)
    # the jax quickstart's first example, ``random.normal(random.PRNGKey(0), (10,))``: TEN draws of one key -- pins the counter layout of a
    # multi-element request (threefry over iota, the two output words laid end to end), which the scalar values above cannot see
    quick = np.array([-0.3721109, 0.26423115, -0.18252768, -0.7368197, -0.44030377, -0.1521442, -0.67135346, -0.5908641, 0.73168886, 0.5673026])
    np.testing.assert_allclose(prng.normal32(prng.PRNGKey(0), (10,)), quick, rtol=0, atol=1.5e-7)      # (erfinv's last bits, as above)
    assert abs(float(prng.normal32(prng.PRNGKey(0))) - (-0.20584226)) < 1e-7                              # ``random.normal(key)``, same page


def test_64_bit_draws_use_the_counter_layout_the_published_32_bit_vector_pins():
    """The reference runs with jax_enable_x64 (``multi_modal.py:14``): 64-bit draws take TWO 32-bit words per element from the same
    counter construction.  No published values exist for that mode; what can be checked is that the 64-bit path is the 32-bit path's
    construction on twice the words: the high words of ``random_bits64`` of n elements are the first n words of ``random_bits32`` of 2 n."""
    k = prng.PRNGKey(7)
    for n in (10, 7, 1):
        b64 = np.asarray(prng.random_bits64(k, n), dtype=np.uint64)
        b32 = prng.random_bits32(k, 2 * n)
        np.testing.assert_array_equal((b64 >> np.uint64(32)).astype(np.uint32), b32[:n])
        np.testing.assert_array_equal((b64 & np.uint64(0xFFFFFFFF)).astype(np.uint32), b32[n:])
