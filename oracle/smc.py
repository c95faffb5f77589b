"""Adaptive tempered SMC on the MALA kernel -- the reference's one baseline without un-vendored dependencies.

ORACLE (test infrastructure; see oracle/__init__.py).  Follows ``bblackjax/smc/base.py:55-134`` (resample -> move ->
weigh), ``resampling.py:50-52,124-135`` (systematic), ``ess.py:28-89`` (log-ESS and the solver's objective, with the
``exp(-delta * loglik)`` weights AS WRITTEN at ``:83``), ``solver.py:20-82`` (dichotomy), ``tempered.py:27-148`` (the move
targets the temperature BEFORE the increment, ``:120-123``), ``adaptive_tempered.py:15-91`` and the driver
``exe_others.py:79-111``.  These files are vendored in the reference (no third-party arithmetic beyond jax.random and
logsumexp), but the reference ships no test vectors for them: PARITY UNPINNED.
"""
import numpy as np

from . import mala, prng
from .targets import Tempered


def logsumexp(a):
    m = np.max(a)
    if not np.isfinite(m):
        m = 0.0
    return m + np.log(np.sum(np.exp(a - m)))


def log_ess(log_weights):
    return 2.0 * logsumexp(log_weights) - logsumexp(2.0 * log_weights)                  # ess.py:41-43


def dichotomy(fun, _delta0, min_delta, max_delta, eps=1e-4, max_iter=100):
    f_a, f_b = fun(min_delta), fun(max_delta)                                           # solver.py:66
    if f_b > 0:
        return max_delta                                                                # :68, :78
    if not f_a > 0:
        return np.nan                                                                   # :80
    a, b, i = min_delta, max_delta, 0
    while i < max_iter and f_a - f_b > eps:                                             # :58-62
        mid = 0.5 * (a + b)
        f_mid = fun(mid)
        if f_mid < 0:
            b, f_b = mid, f_mid
        else:
            a, f_a = mid, f_mid
        i += 1
    return a                                                                            # :74


def ess_solver(logliks, target_ess, max_delta):
    n = logliks.shape[0]
    target_val = np.log(n * target_ess)                                                 # ess.py:80

    def fun(delta):
        with np.errstate(invalid="ignore", over="ignore"):
            return log_ess(np.nan_to_num(-delta * logliks)) - target_val                # :82-86
    return dichotomy(fun, 0.0, 0.0, max_delta)


def systematic(key, weights, num_samples):
    n = weights.shape[0]
    u = prng.uniform(key, ())                                                           # resampling.py:129
    cumsum = np.cumsum(weights)
    linspace = (np.arange(num_samples, dtype=weights.dtype) + u) / num_samples
    idx = np.searchsorted(cumsum, linspace)
    return np.clip(idx, 0, n - 1)


def stratified(key, weights, num_samples):
    n = weights.shape[0]
    u = prng.uniform(key, (num_samples,))                                                # resampling.py:131
    cumsum = np.cumsum(weights)
    linspace = (np.arange(num_samples, dtype=weights.dtype) + u) / num_samples
    return np.clip(np.searchsorted(cumsum, linspace), 0, n - 1)


def _sorted_uniforms(key, n):
    us = prng.uniform(key, (n + 1,))                                                     # :149
    z = np.cumsum(-np.log(us))
    return z[:-1] / z[-1]


def multinomial(key, weights, num_samples):
    n = weights.shape[0]
    linspace = _sorted_uniforms(key, num_samples)                                        # :76
    return np.clip(np.searchsorted(np.cumsum(weights), linspace), 0, n - 1)


def permutation(key, x):
    """jax.random.permutation of a 1-d array (jax 0.4.26 ``_shuffle``: stable sorts by fresh 32-bit keys)."""
    rounds = int(np.ceil(3 * np.log(max(1, x.size)) / np.log(np.iinfo(np.uint32).max)))
    for _ in range(rounds):
        key, sub = prng.split(key)
        x = x[np.argsort(prng.random_bits32(sub, x.size), kind="stable")]
    return x


def residual(key, weights, num_samples):
    key1, key2 = prng.split(key)                                                         # :96
    n = weights.shape[0]
    nw = num_samples * weights
    integer_part = np.floor(nw).astype(np.int32)
    sum_int = int(integer_part.sum())
    residual_sample = multinomial(key1, (nw - integer_part) / (num_samples - sum_int), num_samples)
    residual_sample = permutation(key2, residual_sample)                                 # :114
    integer_idx = np.repeat(np.arange(n + 1), np.concatenate([integer_part, [num_samples - sum_int]]))[:num_samples]
    return np.where(np.arange(num_samples) >= sum_int, residual_sample, integer_idx)     # :122


def init(particles):
    n = particles.shape[0]
    return dict(particles=particles, weights=np.ones(n) / n, lmbda=0.0)                 # tempered.py:45-50


def step(key, state, dist, step_size, target_ess, num_mcmc_steps):
    """One ``adaptive_tempered_smc.step`` (adaptive_tempered.py:80-89 -> tempered.py:88-148 -> base.py:55-134)."""
    lm = state["lmbda"]
    max_delta = 1.0 - lm
    delta = ess_solver(dist.loglik(state["particles"]), target_ess, max_delta)          # adaptive_tempered.py:60-68
    delta = np.clip(delta, 0.0, max_delta)
    updating_key, resampling_key = prng.split(key, 2)                                   # base.py:114
    n = state["weights"].shape[0]
    idx = systematic(resampling_key, state["weights"], n)
    particles = state["particles"][idx]
    keys = prng.split(updating_key, n)
    vg = Tempered(dist, lm).value_and_grad                                              # tempered.py:121-124 (OLD temperature)
    st = mala.init(particles, vg)
    step_keys = prng.split_rows(keys, num_mcmc_steps)
    info = None
    for j in range(num_mcmc_steps):
        st, info, _ = mala.kernel(step_keys[:, j], st, vg, step_size)
    particles = st.position
    log_weights = delta * dist.loglik(particles)                                        # tempered.py:118-119
    lse = logsumexp(log_weights)
    weights = np.exp(log_weights - lse)                                                 # base.py:126-128
    return dict(particles=particles, weights=weights, lmbda=lm + delta), dict(ancestors=idx, lognorm=lse - np.log(n), mcmc=info, delta=delta)


def run(dist, args, n_collect=None):
    """``exe_others.py:79-111``: learning_iter SMC steps, then eval_iter more whose particles are collected."""
    keys = prng.split(prng.PRNGKey(args.seed), args.learning_iter)
    key_dist = prng.split(prng.PRNGKey(args.seed), 6)[3]
    dist.initialize_model(key_dist, args.num_chain)
    state = init(dist.init_params.astype(np.float32).astype(np.float64))
    nsteps = args.anneal_iter // args.num_anneal_temp
    lmbdas, infos = [], []
    for k in keys:
        state, info = step(k, state, dist, args.step_size, args.alpha, nsteps)
        lmbdas.append(state["lmbda"]); infos.append(info)
    collected = []
    for k in prng.split(keys[0], args.eval_iter if n_collect is None else n_collect):
        state, info = step(k, state, dist, args.step_size, args.alpha, nsteps)
        collected.append(state["particles"])
    return dict(lmbdas=np.array(lmbdas), state=state, samples=np.concatenate(collected), infos=infos)
