"""MALA kernel exactly as the reference writes it (float64, batched over chains).

ORACLE (test infrastructure; see oracle/__init__.py).  Follows
``bblackjax/mcmc/mala.py:16-120``, ``bblackjax/mcmc/diffusions.py:19-34``,
``bblackjax/mcmc/proposal.py:80-122,125-161,169-186``, ``bblackjax/util.py:57-82``.

Quirk kept on purpose: the acceptance probability is
``min(1, exp(prev_E - new_E))`` with ``E(a->b) = -logp(a) + |x_b - x_a - eps g_a|^2 / (4 eps)``,
which is the inverse of the textbook Metropolis-Hastings ratio.  ``textbook=True``
flips the sign (a build-side extra; nothing in the reference to compare it with).
"""
from collections import namedtuple

import numpy as np

from . import prng

MALAState = namedtuple("MALAState", "position logdensity logdensity_grad")          # mala.py:16-28
MALAInfo = namedtuple("MALAInfo", "acceptance_rate is_accepted proposed_position proposed_weight")  # :31-48


def init(position, value_and_grad):
    """``mala.py:51-54`` vmapped (``exe_flow_matching.py:316``)."""
    logp, g = value_and_grad(position)
    return MALAState(position, logp, g)


def transition_energy(x_a, logp_a, g_a, x_b, step_size):
    """``mala.py:68-79``: energy of the transition a -> b."""
    theta = x_b - x_a - step_size * g_a
    return -logp_a + 0.25 * (1.0 / step_size) * (theta * theta).sum(1)


def kernel(keys, state, value_and_grad, step_size, textbook=False, noise=None):
    """``mala.py:86-118`` vmapped over chains; ``keys`` is ``[B, 2]`` (one key per chain).

    ``noise`` overrides the Gaussian draw (for tests that feed fixed noise)."""
    x, logp, g = state
    B, d = x.shape
    kk = prng.split_rows(keys, 2)                       # mala.py:93: key_integrator, key_rmh
    if noise is None:
        noise = prng.normal_rows(kk[:, 0], d)           # util.py:80-82
    xn = x + step_size * g + np.sqrt(2.0 * step_size) * noise      # diffusions.py:25-30
    logpn, gn = value_and_grad(xn)                      # diffusions.py:32
    new_E = transition_energy(x, logp, g, xn, step_size)           # proposal.py:157
    prev_E = transition_energy(xn, logpn, gn, x, step_size)        # proposal.py:158
    delta = prev_E - new_E                                         # proposal.py:104
    if textbook:
        delta = -delta
    delta = np.where(np.isnan(delta), -np.inf, delta)              # proposal.py:105
    with np.errstate(over="ignore"):
        p_accept = np.minimum(np.exp(delta), 1.0)                  # proposal.py:178
    u = prng.uniform_rows(kk[:, 1])
    do_accept = u < p_accept                                       # proposal.py:179 (bernoulli)
    theta = x - xn - step_size * gn                                # mala.py:104-112
    with np.errstate(over="ignore"):
        proposed_weight = np.exp(logpn + 0.25 / step_size * (theta * theta).sum(1))   # :113
    acc = do_accept[:, None]
    new_state = MALAState(np.where(acc, xn, x), np.where(do_accept, logpn, logp), np.where(acc, gn, g))
    info = MALAInfo(p_accept, do_accept, xn, proposed_weight)
    return new_state, info, u
