"""GPU parity of the drop-in KERNEL API (SURVEY.md section 8a row M4, 8b iii): ``mfm_amd.bblackjax.mcmc.mala`` --
``init(position, logdensity_fn)``, ``build_kernel()(rng_key, state, logdensity_fn, step_size)`` and
``mala(logdensity_fn, step_size) -> SamplingAlgorithm(init, step)`` (reference ``bblackjax/mcmc/mala.py:51-54,57-120,123-189``,
``bblackjax/base.py:76-103``) -- called the way the reference's callers call them (``exe_flow_matching.py:301,313,316``: the
tempered closure ``lambda x: beta * dist.loglik(x) + dist.logprior(x)``; ``exe_others.py:88-89``: ``mala.build_kernel()`` /
``mala.init`` as plugins) against ``oracle.mala`` on the same keys."""
import numpy as np
import pytest

from oracle import mala as omala, prng, targets

pytestmark = pytest.mark.gpu


def _engine(kind, B, d):
    """A device engine + the product-side distribution, and the oracle twin on the same initial positions."""
    import torch
    from mfm_amd import distributions as D, random as jr
    from mfm_amd.engine import Engine
    from tests import gpu_util as gu
    if kind == "phi4":
        args, odist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=32, F=16)
        dist = D.PhiFour(d)
    elif kind == "gmm16":
        args, odist, k, model, state = gu.gmm16_setup(B=B, hidden=32, F=16)
        dist = D.GaussianMixture(odist.modes, odist.covs, odist.weights)
    else:
        args, odist, k, model, state = gu.lgcp_setup(n=int(np.sqrt(d)), B=B)
        dist = D.LogGaussianCoxPines(d)
    args.ot_cond_flow = False
    eng = Engine(dist, args, model.f)
    x32 = odist.init_params.astype(np.float32)
    return eng, dist, odist, args, torch.as_tensor(x32).cuda(), x32


def _check_step(new, info, o_new, o_info, u, tag):
    np.testing.assert_allclose(info.proposed_position.cpu().numpy(), o_info.proposed_position, rtol=1e-6, atol=1e-6, err_msg=tag)
    np.testing.assert_allclose(info.acceptance_rate.cpu().numpy(), o_info.acceptance_rate, rtol=5e-3, atol=5e-3, err_msg=tag)
    isacc = info.is_accepted.cpu().numpy().astype(bool)
    decided = np.abs(u - o_info.acceptance_rate) > 1e-2                       # decisions may only differ on a knife edge
    np.testing.assert_array_equal(isacc[decided], o_info.is_accepted[decided])
    same = isacc == o_info.is_accepted
    np.testing.assert_allclose(new.position.cpu().numpy()[same], o_new.position[same], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(new.logdensity.cpu().numpy()[same], o_new.logdensity[same], rtol=2e-6, atol=2e-3)
    np.testing.assert_allclose(new.logdensity_grad.cpu().numpy()[same], o_new.logdensity_grad[same], rtol=3e-5, atol=3e-3)
    w_g, w_o = info.proposed_weight.cpu().numpy().astype(np.float64), o_info.proposed_weight          # mala.py:104-113 (diagnostic)
    fin = np.isfinite(w_o) & (w_o > 1e-30) & (w_o < 1e30)
    if fin.any():
        np.testing.assert_allclose(np.log(w_g[fin]), np.log(w_o[fin]), atol=5e-2)


@pytest.mark.parametrize("kind,d,eps,beta", [("phi4", 256, 1e-4, 0.37), ("gmm16", 2, 0.2, 0.6), ("lgcp", 64, 0.01, 0.45)])
def test_build_kernel_with_the_tempered_closure_matches_oracle(kind, d, eps, beta):
    """``kernel(rng_key, state, lambda x: beta * dist.loglik(x) + dist.logprior(x), step_size)`` -- exe_flow_matching.py:301,313."""
    from mfm_amd.bblackjax.mcmc import mala
    B = 64
    eng, dist, odist, args, pos, x32 = _engine(kind, B, d)
    logprob = lambda x: beta * dist.loglik(x) + dist.logprior(x)              # :301
    vg = targets.Tempered(odist, beta).value_and_grad
    state = mala.init(pos, logprob)                                           # mala.py:51-54 / exe_flow_matching.py:316
    o_state = omala.init(x32.astype(np.float64), vg)
    np.testing.assert_allclose(state.logdensity.cpu().numpy(), o_state.logdensity, rtol=2e-6, atol=1e-3)
    np.testing.assert_allclose(state.logdensity_grad.cpu().numpy(), o_state.logdensity_grad, rtol=2e-5, atol=2e-3)
    kernel = mala.build_kernel()                                              # mala.py:57
    key = prng.PRNGKey(123)
    st_in = omala.MALAState(x32.astype(np.float64), state.logdensity.cpu().numpy(), state.logdensity_grad.cpu().numpy().astype(np.float64))
    before = state.position.clone()
    new, info = kernel(key, state, logprob, eps)                              # ONE key: the vmapped call of :303,313
    o_new, o_info, u = omala.kernel(prng.split(key, B), st_in, vg, eps)
    _check_step(new, info, o_new, o_info, u, f"{kind} single key")
    assert (state.position == before).all()                                   # functional: the input state is not modified
    # the caller's own vmap over keys [B, 2] (bblackjax/smc/base.py:122-123)
    keys = prng.split(prng.PRNGKey(9), B)
    new2, info2 = kernel(keys, state, logprob, eps)
    o_new2, o_info2, u2 = omala.kernel(keys, st_in, vg, eps)
    _check_step(new2, info2, o_new2, o_info2, u2, f"{kind} per-chain keys")
    # the kernel as the reference writes it: ONE chain [dim] and its key (mala.py:86-118 un-vmapped)
    b = 5
    one = mala.MALAState(state.position[b], state.logdensity[b], state.logdensity_grad[b])
    new1, info1 = kernel(keys[b], one, logprob, eps)
    assert new1.position.shape == (d,) and new1.logdensity.ndim == 0
    np.testing.assert_array_equal(new1.position.cpu().numpy(), new2.position[b].cpu().numpy())
    np.testing.assert_array_equal(info1.proposed_position.cpu().numpy(), info2.proposed_position[b].cpu().numpy())
    assert float(info1.acceptance_rate) == float(info2.acceptance_rate[b]) and bool(info1.is_accepted) == bool(info2.is_accepted[b])
    st1 = mala.init(state.position[b], logprob)
    assert float(st1.logdensity) == float(state.logdensity[b])
    eng.close()


def test_mala_sampling_algorithm_matches_oracle_over_several_steps():
    """``mala(logdensity_fn, step_size) -> SamplingAlgorithm(init, step)`` (mala.py:123-189), iterated: the chain of states
    follows the oracle's as long as the accept decisions agree (they may only differ on a knife edge)."""
    from mfm_amd.bblackjax.base import SamplingAlgorithm
    from mfm_amd.bblackjax.mcmc.mala import mala
    B, d, eps = 64, 64, 1e-4
    eng, dist, odist, args, pos, x32 = _engine("phi4", B, d)
    algo = mala(dist.logprob, eps)                                            # untempered: logdensity_fn = dist.logprob
    assert isinstance(algo, SamplingAlgorithm) and algo._fields == ("init", "step")
    state = algo.init(pos)
    vg = targets.Tempered(odist, 1.0).value_and_grad
    o_state = omala.MALAState(x32.astype(np.float64), state.logdensity.cpu().numpy(), state.logdensity_grad.cpu().numpy().astype(np.float64))
    key = prng.PRNGKey(2024)
    alive = np.ones(B, bool)                                                  # chains whose decisions have agreed so far
    for it in range(5):
        key, sub = prng.split(key)
        state, info = algo.step(sub, state)
        o_state, o_info, u = omala.kernel(prng.split(sub, B), o_state, vg, eps)
        agree = info.is_accepted.cpu().numpy().astype(bool) == o_info.is_accepted
        knife = np.abs(u - o_info.acceptance_rate) < 1e-2
        assert (agree | knife)[alive].all()
        alive &= agree
        np.testing.assert_allclose(state.position.cpu().numpy()[alive], o_state.position[alive], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(state.logdensity.cpu().numpy()[alive], o_state.logdensity[alive], rtol=2e-6, atol=5e-3)
    assert alive.mean() > 0.9
    eng.close()


def test_counters_report_the_algorithmic_work():
    """``mfm_get_counters``: chain-steps, samples, solves, attempted Dopri5 steps, field evaluations."""
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    B, d = 32, 64
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=32, F=16)
    params = gu.rand_params(model, seed=3, out_scale=0.05)
    params[4]["kernel"] *= 1e-3
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    assert all(v == 0 for v in ctx.counters().values())
    pos = torch.as_tensor(dist.init_params.astype(np.float32)).cuda()
    logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    for i in range(3):
        ctx.mala_step(prng.PRNGKey(i), 1.0, 1e-4, pos, logp, grad)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); g = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(prng.PRNGKey(5), pos, loss, g)
    ctx.adamw_step(g)
    ctx.fm_loss(prng.PRNGKey(6), pos, loss)
    ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(7), 1.0, pos, logp, grad, nsteps=ns)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda")
    ctx.ode_transform(1, pos, out, ldj, key=prng.PRNGKey(8))                 # no nsteps buffer given: the context lends one
    ns2 = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.ode_transform(1, pos, out, ldj, key=prng.PRNGKey(8), nsteps=ns2)
    c = ctx.counters()
    att = int(ns.sum().item()) + 2 * int(ns2.sum().item())
    assert c["mala_chain_steps"] == 3 * B and c["mala_hbm_bytes"] == 3 * B * 4 * (5 * d + 5)
    assert c["fm_train_samples"] == B and c["fm_eval_samples"] == B and c["optimizer_steps"] == 1
    assert c["ode_solves"] == 2 * B + 2 * B and c["dopri_attempts"] == att
    assert c["field_evals"] == 2 * c["ode_solves"] + 6 * att
    ctx.reset_counters()
    assert all(v == 0 for v in ctx.counters().values())
    ctx.close()
