"""The HMC step (build-side mode: mfm_hmc_step / mfm_amd.bblackjax.mcmc.hmc; oracle/hmc.py) through the C ABI against the float64 oracle on the same
keys: trajectories, energies, decisions; the kernel API; the loop with --mcmc_kernel hmc."""
import numpy as np
import pytest

from oracle import hmc, mala, prng, targets

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


@pytest.mark.parametrize("case", ["phi4-64", "phi4-256", "phi4-48", "gmm4", "gmm16"])
def test_hmc_step_matches_oracle(case):
    import torch
    from tests import gpu_util as gu
    if case.startswith("phi4"):
        d = int(case.split("-")[1]); B = 64
        args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=32, F=16)
        eps, L, beta = (2e-4, 12, 0.7)
    elif case == "gmm4":
        B = 64; args, dist, k, model, state = gu.gmm4_setup(B=B); eps, L, beta = (0.15, 8, 1.0)
    else:
        B = 64; args, dist, k, model, state = gu.gmm16_setup(B=B, hidden=32, F=16); eps, L, beta = (0.1, 6, 0.5)
    d = args.dim
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=state.params)
    x32 = dist.init_params.astype(np.float32)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    vg = targets.Tempered(dist, beta).value_and_grad
    st = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda")
    seen = set()
    for it in range(4):
        key = prng.PRNGKey(50 + it)
        ctx.hmc_step(key, beta, eps, L, pos, logp, grad, acc, isacc)
        st_o, info, u = hmc.kernel(prng.split(key, B), st, vg, eps, L)
        pa_g, ia_g = acc.cpu().numpy().astype(np.float64), isacc.cpu().numpy().astype(bool)
        # the acceptance probability is exp(H_0 - H_end) of energies O(|logp|): float32 positions move it by ~1e-7 |logp| in the exponent
        # (the mixtures' log-density is float32 arithmetic per mode on the device: 1e-4 absolute at |logp| ~ 25)
        tol = 5e-6 * max(1.0, np.abs(st.logdensity).max()) + (5e-4 if case.startswith("gmm") else 0.0)
        assert np.abs(np.log(np.maximum(pa_g, 1e-30)) - np.log(np.maximum(info.acceptance_rate, 1e-30))).max() < tol + 1e-5, case
        border = np.abs(u - info.acceptance_rate) < 10 * tol * np.maximum(info.acceptance_rate, 1e-30) + 1e-6
        assert (ia_g == info.is_accepted)[~border].all()
        seen |= set(ia_g.tolist())
        # continue both sides from the KERNEL's decisions (a borderline flip must not cascade): oracle state of the kernel's branch
        m = ia_g[:, None]
        st = mala.MALAState(np.where(m, info.proposed_position, st.position), np.where(ia_g, vg(info.proposed_position)[0], st.logdensity),
                            np.where(m, vg(info.proposed_position)[1], st.logdensity_grad))
        e = np.abs(pos.cpu().numpy().astype(np.float64) - st.position).max()
        assert e < 3e-6 * max(1.0, np.abs(st.position).max()), (case, it, e)
        assert np.abs(logp.cpu().numpy() - st.logdensity).max() < 2e-6 * max(1.0, np.abs(st.logdensity).max()) + (5e-4 if case.startswith("gmm") else 0.0)
        # the oracle continues from float32-rounded positions, as the kernel's state is
        st = mala.MALAState(pos.cpu().numpy().astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    assert True in seen, case                                  # chains move
    ctx.close()


def test_hmc_kernel_api_and_errors():
    import torch
    from mfm_amd import distributions as D
    from mfm_amd.bblackjax.mcmc import hmc as H
    from mfm_amd.engine import Engine
    from oracle import loop
    args = loop.default_args(example="phi-four", dim=64, num_chain=32, hutchs=True, fourier_dim=16, hidden_x=[32, 32], hidden_t=[32, 32], hidden_xt=[32, 32])
    dist = D.PhiFour(64)
    from mfm_amd import random as jr
    dist.initialize_model(jr.PRNGKey(3), 32)
    eng = Engine(dist, args, None)
    alg = H.hmc(dist.logprob, 2e-4, 10)
    st = alg.init(eng.local(dist.init_params))
    st2, info = alg.step(jr.PRNGKey(4), st)
    assert st2.position.shape == (32, 64) and info.acceptance_rate.shape == (32,) and info.is_accepted.dtype == torch.bool
    assert torch.equal(st.position, eng.local(dist.init_params))                       # states are values: the input is untouched
    moved = (st2.position != st.position).any(1)
    assert torch.equal(moved, info.is_accepted)
    with pytest.raises(Exception, match="num_steps"):
        eng.ctx.hmc_step(jr.PRNGKey(1), 1.0, 1e-3, 0, st2.position, st2.logdensity, st2.logdensity_grad)
    eng.close()
    dp = D.LogGaussianCoxPines(64)
    a2 = loop.default_args(example="pines", dim=64, num_chain=16, hutchs=True, fourier_dim=16, hidden_x=[32, 32], hidden_t=[32, 32], hidden_xt=[32, 32])
    dp.initialize_model(jr.PRNGKey(3), 16)
    e2 = Engine(dp, a2, None)
    s = H.init(e2.local(dp.init_params), dp.logprob)
    with pytest.raises(Exception, match="Cox"):
        e2.ctx.hmc_step(jr.PRNGKey(1), 1.0, 1e-3, 3, s.position, s.logdensity, s.logdensity_grad)
    e2.close()


def test_loop_with_the_hmc_kernel_matches_the_oracle_loop():
    """multi_modal's loop with --mcmc_kernel hmc (HMC steps between the flow steps) against the oracle loop in the same mode: losses of the first
    iterations, the chain moments at the end."""
    from mfm_amd import distributions as D, exe_flow_matching as E
    from oracle import loop
    kw = dict(example="phi-four", dim=64, num_chain=64, learning_iter=9, mcmc_per_flow_steps=4.0, hutchs=True, fourier_dim=16, seed=7, eval_iter=1,
              step_size=2e-4, hidden_x=[32, 32], hidden_t=[32, 32], hidden_xt=[32, 32], mcmc_kernel="hmc", hmc_steps=6)
    out = loop.run(targets.PhiFour(64), loop.default_args(**kw))
    res, res_, ex = E.run(D.PhiFour(64), loop.default_args(**kw), None, log_every=1000, return_extras=True)
    lg, lo = ex["metrics"][:, 0], np.asarray(out["trace"]["loss"])
    np.testing.assert_allclose(lg[:4], lo[:4], rtol=2e-5)            # before the first flow step: HMC moves + training only
    np.testing.assert_allclose(lg, lo, rtol=5e-2)
    pg, po = ex["states"].position.cpu().numpy().astype(np.float64), out["states"].position
    close = np.abs(pg - po).max(1) < 1e-3
    # two flow steps and eight HMC moves with annealing: a borderline decision anywhere moves a chain for good (measured 0.77 of 64 chains)
    assert close.mean() > 0.6, close.mean()
    assert np.abs(pg.mean(0) - po.mean(0)).max() < 0.05 and np.abs((pg ** 2).mean(0) - (po ** 2).mean(0)).max() < 0.05
    acc_g = ex["metrics"][:4, 1]
    assert (acc_g > 0.5).all(), acc_g                                # HMC with the textbook rule moves the chains (the as-written MALA rule rejects here)
    ex["engine"].close()
