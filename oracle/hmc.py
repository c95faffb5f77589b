"""Hamiltonian Monte Carlo step (float64, batched over chains).

ORACLE (test infrastructure; see oracle/__init__.py).  BUILD-SIDE MODE, NOT ON THE REFERENCE'S MFM PATH: BASELINE.json's north star
names a "MALA/HMC log-density-and-grad step", the reference's loop uses MALA only and vendors no ``hmc.py`` (SURVEY.md note 7), so
there is nothing of the reference to compare this with.  It restates the HMC kernel of blackjax -- the package the reference's
``bblackjax`` was cut from -- in the form that package has beside the vendored ``mala.py`` / ``proposal.py``:

* ``key_momentum, key_integrator = split(rng_key, 2)``; momentum ``p ~ N(0, I)`` (unit mass matrix, drawn like the MALA noise:
  ``util.py:80-82``);
* ``num_integration_steps`` velocity-Verlet steps, each ``p += eps/2 g; x += eps p; (logp, g) = value_and_grad(x); p += eps/2 g``;
* energy ``H = -logp + |p|^2 / 2``; ``delta = H_0 - H_end`` (NaN -> -inf, ``proposal.py:105``); acceptance probability
  ``min(1, exp(delta))`` and ``accept = uniform(key_integrator) < p`` (``proposal.py:178-179``: static binomial sampling).  This is the
  textbook rule: the inverted ratio of the vendored MALA kernel comes from the argument order of its
  ``transition_energy``, which HMC does not have.
"""
from collections import namedtuple

import numpy as np

from . import prng
from .mala import MALAState

HMCInfo = namedtuple("HMCInfo", "acceptance_rate is_accepted proposed_position energy_delta")


def kernel(keys, state, value_and_grad, step_size, num_integration_steps, momentum=None):
    """``keys`` [B, 2]: one key per chain; ``momentum`` overrides the draw (tests)."""
    x, logp, g = state
    B, d = x.shape
    kk = prng.split_rows(keys, 2)
    p = prng.normal_rows(kk[:, 0], d) if momentum is None else np.asarray(momentum, dtype=np.float64)
    h0 = -logp + 0.5 * (p * p).sum(1)
    xn, pn, lpn, gn = x, p, logp, g
    for _ in range(int(num_integration_steps)):
        pn = pn + 0.5 * step_size * gn
        xn = xn + step_size * pn
        lpn, gn = value_and_grad(xn)
        pn = pn + 0.5 * step_size * gn
    h1 = -lpn + 0.5 * (pn * pn).sum(1)
    delta = h0 - h1
    delta = np.where(np.isnan(delta), -np.inf, delta)
    with np.errstate(over="ignore"):
        p_accept = np.minimum(np.exp(delta), 1.0)
    u = prng.uniform_rows(kk[:, 1])
    acc = u < p_accept
    m = acc[:, None]
    return MALAState(np.where(m, xn, x), np.where(acc, lpn, logp), np.where(m, gn, g)), HMCInfo(p_accept, acc, xn, delta), u
