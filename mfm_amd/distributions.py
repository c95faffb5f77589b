"""Targets and reference distributions -- host-side mirror of the reference's ``distributions.py``.

Same class names, constructor arguments and methods (``logprob / loglik / logprior / initialize_model /
sample_model``, ``distributions.py:8-39``).  The reference hands JAX-traceable closures such as
``lambda x: beta * dist.loglik(x) + dist.logprior(x)`` (``exe_flow_matching.py:301,316``) to the MCMC kernels; a Python
closure cannot cross a C ABI, so here the density methods are *symbolic* when called on a ``Sym`` placeholder: they
return a :class:`LogDensityExpr` (``a * loglik + b * logprior`` of one distribution) that the kernels resolve to a
target descriptor + temperature.  Anything else a user-written closure might compute raises ``NotImplementedError``
-- there is no silent CPU fallback.

Called on a CUDA tensor ``[B, d]`` the methods evaluate on the device through the engine the distribution is
attached to.
"""
import os

import numpy as np

from . import random as jr

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


class Sym:
    """Placeholder position handed to a user's ``logdensity_fn`` to recover its structure."""


class LogDensityExpr:
    """``lik_coef * dist.loglik(x) + prior_coef * dist.logprior(x)`` -- what a logdensity closure may evaluate to."""

    def __init__(self, dist, lik_coef=0.0, prior_coef=0.0):
        self.dist, self.lik_coef, self.prior_coef = dist, float(lik_coef), float(prior_coef)

    def _same(self, other):
        if not isinstance(other, LogDensityExpr) or other.dist is not self.dist:
            raise NotImplementedError("log-density closures may only combine loglik/logprior of ONE distribution")

    def __add__(self, other):
        if isinstance(other, (int, float)) and other == 0:
            return self
        self._same(other)
        return LogDensityExpr(self.dist, self.lik_coef + other.lik_coef, self.prior_coef + other.prior_coef)

    __radd__ = __add__

    def __mul__(self, c):
        if not isinstance(c, (int, float, np.floating)):
            raise NotImplementedError("log-density closures may only scale by a Python/NumPy scalar")
        return LogDensityExpr(self.dist, self.lik_coef * float(c), self.prior_coef * float(c))

    __rmul__ = __mul__

    def temperature(self):
        """beta such that the expression is beta * loglik + logprior (the only family the kernels implement)."""
        if self.dist.has_prior and abs(self.prior_coef - 1.0) > 1e-12:
            raise NotImplementedError("only beta * loglik + 1 * logprior is supported")
        return self.lik_coef


def resolve_logdensity(fn):
    """Evaluate a reference-style ``logdensity_fn`` on a placeholder -> (dist, beta)."""
    if isinstance(fn, LogDensityExpr):
        e = fn
    else:
        e = fn(Sym())
    if not isinstance(e, LogDensityExpr):
        raise NotImplementedError("logdensity_fn must be built from dist.loglik / dist.logprior / dist.logprob")
    return e.dist, e.temperature()


class Distribution:
    has_prior = False
    sample_model = None
    _engine = None

    def _dev_loglik(self, x):
        if self._engine is None:
            raise RuntimeError("distribution is not attached to a device engine yet")
        return self._engine.loglik(x)

    def loglik(self, x):
        if isinstance(x, Sym):
            return LogDensityExpr(self, 1.0, 0.0)
        return self._dev_loglik(x)

    def logprior(self, x):
        if isinstance(x, Sym):
            return LogDensityExpr(self, 0.0, 1.0) if self.has_prior else 0.0
        raise NotImplementedError

    def logprob(self, x):
        if isinstance(x, Sym):
            return LogDensityExpr(self, 1.0, 1.0 if self.has_prior else 0.0)
        if self.has_prior:
            raise NotImplementedError
        return self._dev_loglik(x)

    def grad_logprob(self, x):
        """Marker for ``jax.grad(dist.logprob)`` (exe_flow_matching.py:351); evaluated inside the kernels."""
        raise NotImplementedError("evaluated on the device inside the vector-field kernels")

    def log_prob(self, x):
        return self.logprob(x)

    def sample(self, rng_key, n_samples):
        return self.sample_rows(jr.split(rng_key, n_samples))


class PhiFour(Distribution):
    """``distributions.py:114-165`` (Dirichlet boundary; the ``tilt`` branch is dead in the reference)."""

    kind = "phi4"

    def __init__(self, dim, a=0.1, beta=20.0, bc=("dirichlet", 0), tilt=None):
        if bc[0] != "dirichlet" or bc[1] != 0 or tilt is not None:
            raise NotImplementedError("only the Dirichlet-0, untilted PhiFour of multi_modal.py:53 is built")
        self.dim, self.a, self.beta = int(dim), a, beta
        self.log_Z, self.n_plots, self.can_sample = 0.0, 0, False

    def target_block(self):
        return 0, [self.a, self.beta]

    def initialize_model(self, rng_key, n_chain):
        keys = jr.split(rng_key, n_chain)                                   # :163
        self.init_params = jr.uniform_rows(keys, self.dim) * 2.0 - 1.0      # :164


class GaussianMixture(Distribution):
    """``distributions.py:42-77`` (diagonal mixture; ``dim`` forced to 2 as in the reference, :53)."""

    kind = "gmm"

    def __init__(self, modes, covs, weights):
        self.modes = np.asarray(modes, dtype=np.float64)
        self.covs = np.asarray(covs, dtype=np.float64)
        self.chol_covs = np.sqrt(self.covs)                                 # :51
        self.weights = np.asarray(weights, dtype=np.float64)
        self.dim = 2
        self.log_Z, self.n_plots, self.can_sample = 0.0, 0, False
        self.sample_model = self._sample_model

    def target_block(self):
        K = len(self.weights)
        return 1, np.concatenate([[K], self.modes.reshape(-1), self.chol_covs.reshape(-1), self.weights])

    def initialize_model(self, rng_key, n_chain):
        self.init_params = jr.normal_rows(jr.split(rng_key, n_chain), self.dim)     # :70-71

    def sample_rows(self, keys):
        """vmap(sample_model)(keys) (:73-76)."""
        keys = np.atleast_2d(keys)
        kk = np.stack([jr.split(k, 2) for k in keys])
        p_cuml = np.cumsum(self.weights)
        r = p_cuml[-1] * (1.0 - jr.uniform_rows(kk[:, 0], 1)[:, 0])
        choice = np.minimum(np.searchsorted(p_cuml, r, side="left"), len(self.weights) - 1)
        return self.modes[choice] + self.chol_covs[choice] * jr.normal_rows(kk[:, 1], self.dim)

    def _sample_model(self, rng_key):
        return self.sample_rows(rng_key[None])[0]


class IndepGaussian(Distribution):
    """``distributions.py:80-97`` -- the flow's base distribution ('stdgauss' / 'widegauss')."""

    kind = "indep_gauss"

    def __init__(self, dim, mean=0.0, var=1.0):
        self.dim, self.mean, self.std = dim, mean, np.sqrt(var)

    def sample_rows(self, keys):
        return self.mean + self.std * jr.normal_rows(keys, self.dim)

    def sample_model(self, rng_key):
        return self.sample_rows(rng_key[None])[0]

    def initialize_model(self, rng_key, n_chain):
        self.init_params = jr.normal_rows(jr.split(rng_key, n_chain), self.dim)


class LogGaussianCoxPines(Distribution):
    """``distributions.py:231-314`` (unwhitened).  On the device the Cholesky solves of the reference become one
    contraction with the precomputed K^-1 (cond(K) ~ 15 at 32x32) inside the kernels.
    ``file_path`` may point at the reference's ``finpines.csv``; without it the bundled bin counts are used."""

    kind = "lgcp"
    has_prior = True

    def __init__(self, dim, file_path=None, use_whitened=False):
        if use_whitened:
            raise NotImplementedError("whitened LGCP is outside the hot-path scope (SURVEY.md section 2 row 3)")
        n = int(np.sqrt(dim))
        if file_path is not None and os.path.exists(file_path):
            pts = np.genfromtxt(file_path, delimiter=",")
            counts = np.zeros((n, n))
            for e in pts * n:
                r, c = int(np.floor(e[0])), int(np.floor(e[1]))
                counts[r - (r == n), c - (c == n)] += 1
        else:
            z = np.load(os.path.join(_DATA, "pines_counts.npz"))
            if f"counts_{n}" not in z:
                raise FileNotFoundError(f"no bundled pine counts for a {n}x{n} grid; pass file_path=finpines.csv")
            counts = z[f"counts_{n}"].astype(np.float64)
        self.dim, self.n = int(dim), n
        self.counts = counts.reshape(dim)
        self.poisson_a = 1.0 / dim
        idx = np.array([(i, j) for i in range(n) for j in range(n)], dtype=np.float64)
        dist = np.sqrt(((idx[:, None, :] - idx[None]) ** 2).sum(-1))
        self.gram = 1.91 * np.exp(-dist / (n / 33.0))
        self.chol = np.linalg.cholesky(self.gram)
        self.mu = np.log(126.0) - 0.5 * 1.91
        self.log_norm = -0.5 * dim * np.log(2 * np.pi) - np.log(np.abs(np.diag(self.chol))).sum()

    def target_block(self):
        Kinv = np.linalg.inv(self.gram)
        Kinv = 0.5 * (Kinv + Kinv.T)
        return 2, np.concatenate([[self.mu, self.poisson_a, self.log_norm], self.counts, Kinv.reshape(-1)])

    def initialize_model(self, rng_key, n_chain):
        xi = jr.normal_rows(jr.split(rng_key, n_chain), self.dim)
        self.init_params = self.mu + xi @ self.chol.T                        # :313-314
