"""Pins the oracle's closed-form target gradients / HVPs against torch.autograd (float64),
with the densities re-typed in torch straight from the formulas in distributions.py."""
import os

import numpy as np
import pytest
import torch

from oracle import targets



@pytest.fixture(autouse=True)
def _f64_default():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(dist, logprob_t, x, rtol=1e-10):
    v = np.random.default_rng(0).standard_normal(x.shape)
    xt = torch.tensor(x, requires_grad=True)
    lp = logprob_t(xt)
    np.testing.assert_allclose(dist.logprob(x), lp.detach().numpy(), rtol=rtol, atol=1e-10)
    (g,) = torch.autograd.grad(lp.sum(), xt, create_graph=True)
    np.testing.assert_allclose(dist.grad_logprob(x), g.detach().numpy(), rtol=rtol, atol=1e-9)
    (hv,) = torch.autograd.grad((g * torch.tensor(v)).sum(), xt)
    np.testing.assert_allclose(dist.hvp_logprob(x, v), hv.numpy(), rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("d", [5, 64, 256])
def test_phi4(d):
    dist = targets.PhiFour(d)
    coef, beta = 0.1 * d, 20.0

    def lp(x):
        xp = torch.nn.functional.pad(x, (1, 1))
        diffs = xp[:, 1:] - xp[:, :-1]
        U = (diffs * diffs).sum(1) / 2 * coef
        q = 1 - x * x
        V = (q * q).sum(1) / 4 / coef
        return -beta * (U + V)
    x = np.random.default_rng(1).uniform(-1.5, 1.5, (7, d))
    _check(dist, lp, x)
    np.testing.assert_allclose(dist.grad_loglik(x), dist.grad_logprob(x))


def test_gmm_4mode_and_16mode():
    rng = np.random.default_rng(2)
    for modes, covs, w in [
        (8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4),
        (rng.uniform(-12.8, 12.8, (16, 2)), np.exp(0.5 * rng.standard_normal((16, 2))), rng.dirichlet(4 * np.ones(16))),
    ]:
        dist = targets.GaussianMixture(modes, covs, w)
        m, s, wt = torch.tensor(modes), torch.tensor(np.sqrt(covs)), torch.tensor(w)

        def lp(x):
            z = (x[:, None, :] - m[None]) / s[None]
            pdf = torch.exp(-0.5 * z * z) / (np.sqrt(2 * np.pi) * s[None])
            return torch.log((wt[None] * pdf.prod(-1)).sum(1))
        x = rng.uniform(-10, 10, (9, 2))
        _check(dist, lp, x, rtol=1e-8)


@pytest.mark.parametrize("n", [4, 32])
def test_lgcp(n):
    d = n * n
    if n == 32:
        counts = np.load(os.path.join(ROOT, "mfm_amd", "data", "pines_counts.npz"))["counts_32"]
    else:
        counts = np.random.default_rng(3).poisson(0.5, d)
    dist = targets.LogGaussianCoxPines(d, counts)
    L, c = torch.tensor(dist.chol), torch.tensor(dist.counts)

    def lp(x):
        white = torch.linalg.solve_triangular(L, (x - dist.mu).T, upper=False).T
        prior = -0.5 * (white * white).sum(1) + dist.log_norm
        lik = (x * c - torch.exp(x) / d).sum(1)
        return lik + prior
    x = dist.mu + np.random.default_rng(4).standard_normal((3, d))
    _check(dist, lp, x, rtol=1e-8)
    if n == 32:
        assert np.linalg.cond(dist.gram) < 50
        assert dist.counts.sum() == 126


def test_indep_gaussian_and_initializers():
    from oracle import prng
    ref = targets.IndepGaussian(6)
    x = np.random.default_rng(5).standard_normal((4, 6))
    np.testing.assert_allclose(ref.logprob(x), -0.5 * (x * x).sum(1) - 3 * np.log(2 * np.pi))
    p = targets.PhiFour(16)
    full = p.initialize_model(prng.PRNGKey(1), 12).copy()
    part = p.initialize_model(prng.PRNGKey(1), 12, start=4, count=5)
    np.testing.assert_array_equal(full[4:9], part)
    assert np.abs(full).max() <= 1.0
