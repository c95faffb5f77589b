"""GPU parity: kernelised Stein discrepancy and maximum mean discrepancy (mfm_stein_disc / mfm_max_mean_disc) vs the
float64 oracle (oracle/metrics.py, follows mcmc_utils.py:28-111).  Stated tolerance: 1e-5 relative (float32 pair
terms, float64 accumulation)."""
import numpy as np
import pytest

from oracle import metrics

pytestmark = pytest.mark.gpu


def _dev(x):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()


@pytest.mark.parametrize("n", [256, 333])           # ragged: n not a multiple of the 64-pair tile
def test_stein_disc_phi4_matches_oracle(n):
    from tests import gpu_util as gu
    d = 64
    args, dist, k, model, state = gu.phi4_setup(d=d, B=64, hidden=32, F=16)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=state.params)
    rng = np.random.default_rng(5)
    x32 = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    g32 = dist.grad_logprob(x32.astype(np.float64)).astype(np.float32)
    u, v = ctx.stein_disc(_dev(x32), _dev(g32))
    # oracle on the SAME float32-rounded inputs
    uo, vo = metrics.stein_disc(x32.astype(np.float64), lambda x: g32.astype(np.float64))
    np.testing.assert_allclose([u, v], [uo, vo], rtol=1e-5)
    # permutation invariance (size-independent property)
    p = rng.permutation(n)
    u2, v2 = ctx.stein_disc(_dev(x32[p]), _dev(g32[p]))
    np.testing.assert_allclose([u2, v2], [u, v], rtol=1e-6)
    ctx.close()


def test_stein_and_mmd_mixture_match_oracle():
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.gmm4_setup(B=64)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=state.params)
    rng = np.random.default_rng(6)
    n = 1000
    x32 = (8.0 * rng.choice([-1.0, 1.0], (n, 2)) + rng.standard_normal((n, 2))).astype(np.float32)
    y32 = (8.0 * rng.choice([-1.0, 1.0], (n, 2)) + 1.3 * rng.standard_normal((n, 2))).astype(np.float32)
    g32 = dist.grad_logprob(x32.astype(np.float64)).astype(np.float32)
    u, v = ctx.stein_disc(_dev(x32), _dev(g32))
    uo, vo = metrics.stein_disc(x32.astype(np.float64), lambda x: g32.astype(np.float64))
    np.testing.assert_allclose(v, vo, rtol=1e-5)
    np.testing.assert_allclose(u, uo, rtol=1e-5, atol=1e-5 * abs(vo))
    mmd = ctx.max_mean_disc(_dev(x32), _dev(y32))
    mo = metrics.max_mean_disc(x32.astype(np.float64), y32.astype(np.float64))
    np.testing.assert_allclose(mmd, mo, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(ctx.max_mean_disc(_dev(y32), _dev(x32)), mmd, rtol=1e-6)        # symmetric (tile sums in another order)
    ctx.close()


def test_run_returns_finite_metrics():
    """exe_flow_matching.run fills the reference's result vectors (:561): logpdf, KSD U / V, MMD, train time."""
    from mfm_amd import distributions as D, exe_flow_matching as E
    from oracle import loop
    modes, covs, w = 8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4
    dg = D.GaussianMixture(modes, covs, w)
    args = loop.default_args(example="4-mode", dim=2, num_chain=64, learning_iter=6, mcmc_per_flow_steps=2.0, hutchs=False,
                             fourier_dim=16, hidden_x=[32, 32], hidden_t=[32, 32], hidden_xt=[32, 32], seed=3, eval_iter=2, step_size=0.2)
    res, res_ = E.run(dg, args, dg.sample_model, log_every=1000)
    assert np.all(np.isfinite(res)) and np.all(np.isfinite(res_))
    assert res[2] >= 0 and res_[2] >= 0            # V-statistics
