"""End-to-end smoke of the oracle loop: it runs, flow steps fire on schedule, loss decreases."""
import numpy as np

from oracle import flow, loop, ode, prng, targets


def test_four_mode_mini_run():
    args = loop.default_args(example="4-mode", dim=2, num_chain=64, learning_iter=12, mcmc_per_flow_steps=3.0,
                             eval_iter=2, step_size=0.2, fourier_dim=8, hidden_x=[16, 16], hidden_t=[16, 16],
                             hidden_xt=[16, 16], seed=1)
    dist = targets.GaussianMixture(8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4)
    out = loop.run(dist, args, target_gn=dist.sample_model_rows)
    tr = out["trace"]
    assert len(tr["loss"]) == 12 and len(tr["n_att"]) == 3          # flow steps at counts 4, 8, 12
    assert np.isfinite(tr["loss"]).all() and np.isfinite(tr["target_loss"]).all()
    assert tr["target_loss"][-1] < tr["target_loss"][0]      # the net learns (chain loss grows as chains spread)
    assert 0 < tr["beta"][0] <= 1
    assert tr["learning_rate"][0] == 1e-3 and abs(tr["learning_rate"][-1] - 1e-3 * (1 - 11 / 12)) < 1e-12


def test_phi4_hutch_flow_step_identity_at_init():
    # zero output kernels => v == 0 => flow is the identity, logdet 0, acceptance = exp(logp' - logp)
    args = loop.default_args(example="phi-four", dim=16, num_chain=8, learning_iter=1, mcmc_per_flow_steps=0.0,
                             hutchs=True, step_size=1e-4, fourier_dim=8, hidden_x=[16, 16], hidden_t=[16, 16],
                             hidden_xt=[16, 16], seed=1024)
    dist = targets.PhiFour(16)
    k, model, state, lr_fn, _, _ = loop.setup(dist, args)
    st = flow.init_fn(dist.init_params, dist, 1.0)
    keys = prng.split(prng.PRNGKey(5), 8)
    stats = {}
    new, info = flow.rwmh_step(keys, st, targets.Tempered(dist, 1.0).value_and_grad, model, state.params, args, stats)
    np.testing.assert_allclose(stats["u0"], st.position, atol=1e-12)
    np.testing.assert_allclose(stats["vol0"], 0, atol=1e-12)
    kk = prng.split_rows(keys, 4)
    up = st.position + 2.38 / 4.0 * prng.normal_rows(kk[:, 0], 16)
    np.testing.assert_allclose(info.proposed_position, up, atol=1e-12)
    with np.errstate(over="ignore"):
        np.testing.assert_allclose(info.acceptance_rate, np.exp(dist.logprob(up) - st.logdensity), rtol=1e-9)
