"""Target and reference densities: value, gradient, Hessian-vector product (float64, batched).

ORACLE (test infrastructure; see oracle/__init__.py).  Follows
``distributions.py:42-97,114-165,231-314`` and ``cox_process_utils.py:29-165``.
The reference obtains gradients by ``jax.grad`` / ``jax.jvp``; here they are the
closed forms (exact calculus), pinned against ``torch.autograd`` in
tests/test_oracle_targets.py.

All functions take ``x`` of shape ``[B, d]`` and return ``[B]`` / ``[B, d]``.
"""
import numpy as np

from . import prng

LOG_2PI = np.log(2.0 * np.pi)


class PhiFour:
    """``distributions.py:114-165`` (Dirichlet boundary, tilt=None)."""

    kind = "phi4"

    def __init__(self, dim, a=0.1, beta=20.0):
        self.dim = int(dim)
        self.a = a
        self.beta = beta
        self.coef = a * dim            # distributions.py:132,150
        self.sample_model = None       # multi_modal.py:61

    def _lap(self, x):
        xp = np.pad(x, ((0, 0), (1, 1)))                      # distributions.py:144
        return 2.0 * x - xp[:, :-2] - xp[:, 2:]

    def loglik(self, x):
        xp = np.pad(x, ((0, 0), (1, 1)))
        diffs = xp[:, 1:] - xp[:, :-1]                         # :148
        U = (diffs * diffs).sum(1) / 2.0 * self.coef           # :149-151
        q = 1.0 - x * x                                        # :133
        V = (q * q).sum(1) / 4.0 / self.coef                   # :134
        return -self.beta * (U + V)                            # :157

    def logprior(self, x):
        return np.zeros(x.shape[0])                            # :159-160

    def logprob(self, x):
        return self.loglik(x) + self.logprior(x)               # :153-154

    def grad_loglik(self, x):
        return -self.beta * (self.coef * self._lap(x) - x * (1.0 - x * x) / self.coef)

    def grad_logprior(self, x):
        return np.zeros_like(x)

    def grad_logprob(self, x):
        return self.grad_loglik(x)

    def hvp_logprob(self, x, v):
        return -self.beta * (self.coef * self._lap(v) - (1.0 - 3.0 * x * x) * v / self.coef)

    def hess_diag(self, x):
        """Diagonal of the Hessian of logprob = hvp_logprob(x, e_j)[j] (exact-trace log-det of the gate term)."""
        return -self.beta * (2.0 * self.coef - (1.0 - 3.0 * x * x) / self.coef)

    def initialize_model(self, key, n_chain, start=0, count=None):
        """``distributions.py:162-164``: U(-1, 1) per chain key."""
        count = n_chain - start if count is None else count
        keys = prng.split_at(key, n_chain, np.arange(start, start + count))
        self.init_params = prng.uniform_rows(keys, self.dim) * 2.0 - 1.0
        return self.init_params


class GaussianMixture:
    """``distributions.py:42-77``: diagonal mixture, density as written (product of pdfs, then log)."""

    kind = "gmm"

    def __init__(self, modes, covs, weights):
        self.modes = np.asarray(modes, dtype=np.float64)       # [K, d]
        self.covs = np.asarray(covs, dtype=np.float64)
        self.chol_covs = np.sqrt(self.covs)                    # :51
        self.weights = np.asarray(weights, dtype=np.float64)
        self.dim = 2                                           # :53

    def _comp(self, x):
        z = (x[:, None, :] - self.modes[None]) / self.chol_covs[None]       # [B,K,d]
        pdf = np.exp(-0.5 * z * z) / (np.sqrt(2.0 * np.pi) * self.chol_covs[None])
        return self.weights[None] * pdf.prod(-1), z                         # :59

    def logprob(self, x):
        with np.errstate(divide="ignore"):
            return np.log(self._comp(x)[0].sum(1))                          # :61

    loglik = logprob                                                        # :63-64

    def logprior(self, x):
        return np.zeros(x.shape[0])                                         # :66-67

    def _resp(self, x):
        p, z = self._comp(x)
        with np.errstate(invalid="ignore", divide="ignore"):
            r = p / p.sum(1, keepdims=True)
        a = -z / self.chol_covs[None]            # d log N_k / dx = -(x - m)/s^2
        return r, a

    def grad_logprob(self, x):
        r, a = self._resp(x)
        return (r[:, :, None] * a).sum(1)

    grad_loglik = grad_logprob

    def grad_logprior(self, x):
        return np.zeros_like(x)

    def hvp_logprob(self, x, v):
        r, a = self._resp(x)
        g = (r[:, :, None] * a).sum(1)
        av = (a * v[:, None, :]).sum(-1)                                    # [B,K]
        t1 = (r[:, :, None] * (a * av[:, :, None] - v[:, None, :] / self.covs[None])).sum(1)
        return t1 - g * (g * v).sum(1, keepdims=True)

    def initialize_model(self, key, n_chain, start=0, count=None):
        """``distributions.py:69-71``: N(0, I) per chain key."""
        count = n_chain - start if count is None else count
        keys = prng.split_at(key, n_chain, np.arange(start, start + count))
        self.init_params = prng.normal_rows(keys, self.dim)
        return self.init_params

    def sample_model_rows(self, keys):
        """vmap of ``distributions.py:73-76``."""
        kk = prng.split_rows(keys, 2)
        p_cuml = np.cumsum(self.weights)
        r = p_cuml[-1] * (1.0 - prng.uniform_rows(kk[:, 0]))
        choice = np.minimum(np.searchsorted(p_cuml, r, side="left"), len(self.weights) - 1)
        return self.modes[choice] + self.chol_covs[choice] * prng.normal_rows(kk[:, 1], self.dim)


REF_VARS = {"stdgauss": 1.0, "widegauss": 5.0}      # exe_flow_matching.py:48-54 (the entries that can be constructed)


class IndepGaussian:
    """``distributions.py:80-97`` (the flow's base distribution, 'stdgauss' by default)."""

    kind = "indep_gauss"

    def __init__(self, dim, mean=0.0, var=1.0):
        self.dim = dim
        self.mean = mean
        self.std = np.sqrt(var)

    def logprob(self, x):
        z = (x - self.mean) / self.std
        return (-0.5 * z * z - np.log(self.std) - 0.5 * LOG_2PI).sum(1)    # :90

    def sample_model_rows(self, keys):
        return self.mean + self.std * prng.normal_rows(keys, self.dim)     # :97


def pines_bin_counts(points, n):
    """``cox_process_utils.py:29-56``."""
    counts = np.zeros((n, n))
    for elem in points * n:
        row, col = int(np.floor(elem[0])), int(np.floor(elem[1]))
        row -= row == n
        col -= col == n
        counts[row, col] += 1
    return counts


class LogGaussianCoxPines:
    """``distributions.py:231-314`` unwhitened (the default, ``:279-281``).

    ``counts``: flat ``[d]`` bin counts of the 126 pine saplings on the sqrt(d) x sqrt(d)
    grid (``cox_process_utils.py:29-56``; committed fixture, see tools/make_pines_counts.py).
    """

    kind = "lgcp"

    def __init__(self, dim, counts):
        self.dim = int(dim)
        n = int(np.sqrt(dim))
        self.n = n
        self.counts = np.asarray(counts, dtype=np.float64).reshape(dim)     # :249
        self.poisson_a = 1.0 / dim                                          # :252
        sv, beta = 1.91, 1.0 / 33                                           # :256-257
        idx = np.array([(i, j) for i in range(n) for j in range(n)], dtype=np.float64)  # cox:59-64
        dist = np.sqrt(((idx[:, None, :] - idx[None]) ** 2).sum(-1))
        self.gram = sv * np.exp(-dist / (n * beta))                         # cox:93-95
        self.chol = np.linalg.cholesky(self.gram)                           # :266
        self.log_norm = -0.5 * dim * LOG_2PI - np.log(np.abs(np.diag(self.chol))).sum()  # :270-272
        self.mu = np.log(126.0) - 0.5 * sv                                  # :274
        self.Kinv = np.linalg.inv(self.gram)
        self.Kinv = 0.5 * (self.Kinv + self.Kinv.T)
        self.sample_model = None                                            # multi_modal.py:98

    def loglik(self, x):
        return (x * self.counts[None] - self.poisson_a * np.exp(x)).sum(1)  # cox:113-115

    def logprior(self, x):
        import scipy.linalg as sla
        white = sla.solve_triangular(self.chol, (x - self.mu).T, lower=True).T   # cox:161-162
        return -0.5 * (white * white).sum(1) + self.log_norm                # :302-303

    def logprob(self, x):
        return self.loglik(x) + self.logprior(x)                            # :309-310

    def grad_loglik(self, x):
        return self.counts[None] - self.poisson_a * np.exp(x)

    def grad_logprior(self, x):
        return -(x - self.mu) @ self.Kinv

    def grad_logprob(self, x):
        return self.grad_loglik(x) + self.grad_logprior(x)

    def hvp_logprob(self, x, v):
        return -self.poisson_a * np.exp(x) * v - v @ self.Kinv

    def hess_diag(self, x):
        """Diagonal of the Hessian of logprob = hvp_logprob(x, e_j)[j]."""
        return -self.poisson_a * np.exp(x) - np.diag(self.Kinv)[None]

    def initialize_model(self, key, n_chain, start=0, count=None):
        """``distributions.py:312-314``: mu + L xi."""
        count = n_chain - start if count is None else count
        keys = prng.split_at(key, n_chain, np.arange(start, start + count))
        self.init_params = self.mu + prng.normal_rows(keys, self.dim) @ self.chol.T
        return self.init_params


class Tempered:
    """``exe_flow_matching.py:301,316``: logprob = beta * loglik + logprior."""

    def __init__(self, dist, beta=1.0):
        self.dist = dist
        self.beta = float(beta)

    def value_and_grad(self, x):
        d = self.dist
        return (self.beta * d.loglik(x) + d.logprior(x),
                self.beta * d.grad_loglik(x) + d.grad_logprior(x))
