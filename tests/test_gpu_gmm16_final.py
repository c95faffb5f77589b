"""GPU parity for BASELINE configs[1] (the 16-mode `gaussian-mixture` target, d = 2, exact-trace log-det, 4096 chains,
409,600-sample eval batch), for the end-of-run block N1 (final flow sampling + self-normalised importance resampling,
exe_flow_matching.py:453-459) and for the fractional schedule 0 < mcmc_per_flow_steps < 1 (:304-309)."""
import numpy as np
import pytest

from oracle import flow, fm, loop, mala, ode, prng, targets

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def test_gmm16_fixture_is_what_the_cli_builds():
    from mfm_amd.multi_modal import gmm16_parameters
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.gmm16_setup(B=16, hidden=32, F=16)
    m, c, w = gmm16_parameters()
    np.testing.assert_array_equal(m, dist.modes); np.testing.assert_array_equal(c, dist.covs); np.testing.assert_array_equal(w, dist.weights)
    assert m.shape == (16, 2) and abs(w.sum() - 1) < 1e-12 and np.abs(m).max() <= 12.8 and (c > 0).all()       # multi_modal.py:40-45


def test_gmm16_mala_fm_and_exact_trace_flow_step_match_oracle():
    """The three kernels of one loop iteration on the 16-mode target at the reference's network widths (hidden 128, F = 128)."""
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    B, d = 64, 2
    args, dist, k, model, state = gu.gmm16_setup(B=B, hutchs=False)
    params = gu.rand_params(model, seed=9, out_scale=0.3)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x32 = dist.init_params.astype(np.float32)
    beta, eps = 0.7, args.step_size
    vg = targets.Tempered(dist, beta).value_and_grad
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    st = mala.init(x32.astype(np.float64), vg)
    np.testing.assert_allclose(logp.cpu().numpy(), st.logdensity, rtol=2e-6, atol=2e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), st.logdensity_grad, rtol=2e-5, atol=2e-5)
    # MALA step (mala.py:86-118)
    key = prng.PRNGKey(77)
    st_in = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    new, info, u = mala.kernel(prng.split(key, B), st_in, vg, eps)
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda"); prop = torch.empty(B, d, device="cuda")
    p2, l2, g2 = pos.clone(), logp.clone(), grad.clone()
    ctx.mala_step(key, beta, eps, p2, l2, g2, acc, isacc, prop)
    np.testing.assert_allclose(prop.cpu().numpy(), info.proposed_position, rtol=1e-6, atol=2e-6)
    np.testing.assert_allclose(acc.cpu().numpy(), info.acceptance_rate, rtol=2e-3, atol=2e-3)
    decided = np.abs(u - info.acceptance_rate) > 1e-2
    np.testing.assert_array_equal(isacc.cpu().numpy()[decided].astype(bool), info.is_accepted[decided])
    assert 0.0 < info.is_accepted.mean() < 1.0                               # both branches of the select are exercised
    # flow-matching loss and gradient (exe_flow_matching.py:151-178)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(key, pos, loss, grads)
    lo, go = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma)
    assert abs(loss.item() - lo) < 2e-5 * abs(lo)
    gflat = gu.flat_params(go)
    assert np.abs(grads.cpu().numpy() - gflat).max() < 2e-4 * np.abs(gflat).max()
    # exact-trace flow-MH step (no --hutch: trace(jacfwd(v)), :216-217), on the oracle's step sequence and with own controllers
    B2 = 32
    ctx2 = gu.make_ctx(dist, args, n_local=B2, n_total=B2, fourier=model.f, params=params)
    x2 = x32[:B2]
    pos = _dev(x2); logp = torch.empty(B2, dtype=torch.float64, device="cuda"); grad = torch.empty(B2, d, device="cuda")
    ctx2.mala_init(pos, beta, logp, grad)
    st0 = mala.MALAState(x2.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    keys = prng.split(prng.PRNGKey(31), B2)
    args2 = loop.default_args(**{**vars(args), "num_chain": B2}); args2.n_ts = args.n_ts
    nat = {}
    new_n, info_n = flow.rwmh_step(keys, st0, vg, model, params, args2, nat)
    from tests.test_gpu_replay import _replay_arrays
    dt, ac = _replay_arrays([nat["inv"], nat["fwd"]])
    rp = dict(inv=dict(dt=dt[0].astype(np.float64), acc=ac[0]), fwd=dict(dt=dt[1].astype(np.float64), acc=ac[1]))
    so = {}
    new_o, info_o = flow.rwmh_step(keys, st0, vg, model, params, args2, so, replay=rp)
    ratio = torch.zeros(dt.shape, device="cuda"); own = torch.zeros(dt.shape, device="cuda"); diag = torch.zeros(B2, 4, dtype=torch.float64, device="cuda")
    ctx2.debug_replay(_dev(dt), _dev(ac), ratio, own, diag)
    a = torch.empty(B2, device="cuda"); ia = torch.empty(B2, dtype=torch.uint8, device="cuda"); pr = torch.empty(B2, d, device="cuda"); ns = torch.empty(B2, dtype=torch.int32, device="cuda")
    pc, lc, gc = pos.clone(), logp.clone(), grad.clone()
    ctx2.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(31), beta, pc, lc, gc, a, ia, pr, ns)
    np.testing.assert_array_equal(ns.cpu().numpy(), so["n_att_inv"] + so["n_att_fwd"])
    dg = diag.cpu().numpy()
    e_p = np.abs(pr.cpu().numpy() - info_o.proposed_position).max()
    e_v = max(np.abs(dg[:, 0] - so["vol0"]).max(), np.abs(dg[:, 1] - so["volp"]).max())
    e_a = np.abs(dg[:, 3] - so["log_alpha"]).max()
    print(f"gmm16 exact-trace flow step on prescribed steps: attempts {ns.float().mean().item():.0f}, |dx'| {e_p:.2e}, |dvol| {e_v:.2e}, |d log alpha| {e_a:.2e}")
    assert e_p < 1e-4 * max(1.0, np.abs(info_o.proposed_position).max()) and e_v < 1e-3 and e_a < 5e-3
    sure = np.abs(so["log_alpha"]) > 0.05
    np.testing.assert_array_equal(ia.cpu().numpy().astype(bool)[sure], info_o.is_accepted[sure])
    same = ia.cpu().numpy().astype(bool) == info_o.is_accepted
    np.testing.assert_allclose(pc.cpu().numpy()[same], new_o.position[same], atol=1e-4 * max(1.0, np.abs(new_o.position).max()))
    np.testing.assert_allclose(lc.cpu().numpy()[same], new_o.logdensity[same], rtol=1e-5, atol=1e-3)
    # natural controllers
    ctx2.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(31), beta, pos, logp, grad, a, ia, pr, ns)
    tot = nat["n_att_inv"] + nat["n_att_fwd"]
    assert np.abs(pr.cpu().numpy() - info_n.proposed_position).max() < 5e-3 * max(1.0, np.abs(info_n.proposed_position).max())
    assert abs(ns.float().mean().item() - tot.mean()) < 0.1 * tot.mean()
    ctx.close(); ctx2.close()


def test_gmm16_full_size_iteration_and_eval_batch():
    """configs[1] at its full size: 4096 chains and the 409,600 exact samples of eval_step (:370-374, eval_iter = 100).
    (a) the eval loss on the 409,600 samples the target's own sampler draws is the sum of its chunks and, on one chunk, the
    oracle's; (b) a MALA step + loss/gradient at 4096 chains agree with the oracle on the chains it can afford (draws are
    indexed by global chain id, so a 64-chain shard IS the first 64 chains of the 4096)."""
    import torch
    from mfm_amd import distributions as D, random as jr
    from tests import gpu_util as gu
    B, n_iter = 4096, 100
    n_eval = B * n_iter
    args, dist, k, model, state = gu.gmm16_setup(B=B, hutchs=False)
    params = gu.rand_params(model, seed=3)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, max_eval=n_eval)
    dg = D.GaussianMixture(dist.modes, dist.covs, dist.weights)
    key_gen, key_loss = jr.split(k["target"])                                             # :371
    real = dg.sample_rows(jr.split(key_gen, n_eval)).astype(np.float32)                   # :372-373, the product's host sampler
    np.testing.assert_array_equal(real[:256], dist.sample_model_rows(prng.split(key_gen, n_eval)[:256]).astype(np.float32))
    xs = _dev(real)
    full = torch.zeros(1, dtype=torch.float64, device="cuda")
    ctx.fm_loss(key_loss, xs, full, n_total=n_eval, offset=0)
    part = torch.zeros(1, dtype=torch.float64, device="cuda"); tot = 0.0; first = None
    for c in range(0, n_eval, 16 * B):
        ctx.fm_loss(key_loss, xs[c:c + 16 * B], part, n_total=n_eval, offset=c)
        tot += part.item()
    assert np.isfinite(full.item()) and abs(tot - full.item()) < 1e-9 * abs(full.item())
    ctx.fm_loss(key_loss, xs[:1024], part, n_total=n_eval, offset=0)
    lo, _ = fm.loss_and_grad(model, params, key_loss, real[:1024].astype(np.float64), args.sigma, need_grad=False, n_total=n_eval, start=0)
    assert abs(part.item() - lo) < 2e-5 * abs(lo)
    # large sample sets of the mixtures run the 64-samples-per-workgroup forward kernel (fm_eval64_kernel); MFM_EVAL16 routes
    # them through the 16-row kernel every other call uses: the same per-sample arithmetic, another grouping of the partial sums.
    # Also a sample count that is not a multiple of 64 (the last workgroup is partly empty) and the oracle on that set.
    import os
    n_odd = 16384 + 48
    ctx.fm_loss(key_loss, xs[:n_odd], part, n_total=n_eval, offset=0); l64 = part.item()
    os.environ["MFM_EVAL16"] = "1"
    try:
        ctx.fm_loss(key_loss, xs[:n_odd], part, n_total=n_eval, offset=0); l16 = part.item()
        ctx.fm_loss(key_loss, xs, part, n_total=n_eval, offset=0); f16 = part.item()
    finally:
        del os.environ["MFM_EVAL16"]
    # (each lane first sums its own squared residuals in float32 -- 4 with 16 rows per workgroup, 16 with 64 -- before the float64
    # reduction: measured 1.4e-9 relative)
    assert abs(l64 - l16) < 1e-7 * abs(l16) and abs(full.item() - f16) < 1e-7 * abs(f16), (l64, l16, full.item(), f16)
    lo, _ = fm.loss_and_grad(model, params, key_loss, real[:n_odd].astype(np.float64), args.sigma, need_grad=False, n_total=n_eval, start=0)
    assert abs(l64 - lo) < 2e-5 * abs(lo)
    # (b) one iteration's kernels at 4096 chains vs the oracle on chains [0, 64)
    x32 = dist.init_params.astype(np.float32)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, 2, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    key = prng.PRNGKey(5)
    vg = targets.Tempered(dist, 1.0).value_and_grad
    st_in = mala.MALAState(x32[:64].astype(np.float64), logp[:64].cpu().numpy(), grad[:64].cpu().numpy().astype(np.float64))
    new, info, u = mala.kernel(prng.split(key, B)[:64], st_in, vg, args.step_size)
    acc = torch.empty(B, device="cuda"); prop = torch.empty(B, 2, device="cuda")
    ctx.mala_step(key, 1.0, args.step_size, pos, logp, grad, acc, None, prop)
    np.testing.assert_allclose(prop[:64].cpu().numpy(), info.proposed_position, rtol=1e-6, atol=2e-6)
    np.testing.assert_allclose(acc[:64].cpu().numpy(), info.acceptance_rate, rtol=2e-3, atol=2e-3)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(key, pos, loss, grads)
    assert np.isfinite(loss.item()) and torch.isfinite(grads).all()
    ctx.close()


@pytest.mark.parametrize("case", ["gmm4-exact", "phi4-hutch"])
def test_final_sampling_and_importance_resampling_match_oracle(case):
    """N1 (exe_flow_matching.py:453-459): reference draws -> flow (ONE shared Hutchinson key) -> log-weights -> choice(p = w)."""
    import torch
    from mfm_amd import distributions as D, exe_flow_matching as E, random as jr
    from mfm_amd.engine import Engine
    from tests import gpu_util as gu
    if case == "gmm4-exact":
        args, odist, k, model, state = gu.gmm4_setup(B=64, hidden=32, F=16, hutchs=False, eval_iter=4)
        dist = D.GaussianMixture(odist.modes, odist.covs, odist.weights)
        params = gu.rand_params(model, seed=9, out_scale=0.3)
    else:
        args, odist, k, model, state = gu.phi4_setup(d=64, B=32, hidden=32, F=16, eval_iter=2)
        dist = D.PhiFour(64)
        params = gu.rand_params(model, seed=9, out_scale=0.3)
        params[4]["kernel"] *= 1e-3; params[4]["bias"] *= 1e-3
    n = args.eval_iter * args.num_chain
    eng = Engine(dist, args, model.f, max_eval_samples=n)
    eng.ctx.set_params(gu.flat_params(params))
    gmodel = E.VectorFieldNet(model.f, dist.grad_logprob, args.hidden_x, args.hidden_t, args.hidden_xt).attach(eng)
    _, _, transform_and_logdet = E.create_train_data_gn(dist, gmodel.apply, None, args)
    key_gen = k["gen"]
    fin = E.final_sampling(eng, dist, args, key_gen, transform_and_logdet)
    x_o, ex_o, info = loop.final_sampling(model, params, odist, args, key_gen)
    g = {kk: v.cpu().numpy() for kk, v in fin.items()}
    np.testing.assert_allclose(g["u"], info["u"], atol=1e-6)                                     # :453
    assert np.abs(g["flow_samples"] - x_o).max() < 2e-3 * max(1.0, np.abs(x_o).max())            # :455 (two adaptive solves)
    vol_tol = 5e-2 if case == "phi4-hutch" else 1e-2
    assert np.abs(g["vols"] - info["vols"]).max() < vol_tol * max(1.0, np.abs(info["vols"]).max())
    # log-weights (:457) of two ADAPTIVE solves with their own controllers: log pi(x) inherits |grad log pi| |dx| -- for
    # phi-four O(1e3) * 1e-3 on log-densities of O(1e4) (measured 1.9), for the mixture O(1) * 2e-3 (measured 2.3e-2)
    lw_tol = 5e-4 * np.abs(info["log_weights"]).max() if case == "phi4-hutch" else 5e-2
    assert np.abs(g["log_weights"] - info["log_weights"]).max() < lw_tol, np.abs(g["log_weights"] - info["log_weights"]).max()
    # the categorical draw (:458-459) itself: on the ORACLE's log-weights the kernel's indices are bit-exact (integer output;
    # float64 exp + a cumulative sum taken in index order, as for the SMC resampling)
    idx = torch.empty(n, dtype=torch.int32, device="cuda"); scratch = torch.empty(n, dtype=torch.float64, device="cuda")
    key_hutch, key_choice = jr.split(key_gen)
    eng.ctx.choice_logw(key_choice, _dev(info["log_weights"]), n, scratch, idx)
    np.testing.assert_array_equal(idx.cpu().numpy(), info["idx"])
    w = info["weights"]
    assert 1.0 / (w / w.sum()).max() > 1.5 or case == "phi4-hutch"                              # the draw is not degenerate
    # end to end (own log-weights): the same indices except where a weight difference moves a CDF boundary across a uniform
    agree = (g["idx"] == info["idx"]).mean()
    assert agree > (0.9 if case == "gmm4-exact" else 0.5), agree
    np.testing.assert_array_equal(g["exact_samples"], g["flow_samples"][g["idx"]])              # :459 gather
    eng.close()


def test_fractional_schedule_runs_flow_steps_between_mala_steps():
    """0 < mcmc_per_flow_steps < 1 (exe_flow_matching.py:304-309): with K = 0.5 the generator takes int(1 / K) = 2 flow steps
    per MALA step (flow unless count % 3 == 0).  The product's loop against the oracle's on the same seed."""
    from mfm_amd import distributions as D, exe_flow_matching as E
    common = dict(example="phi-four", dim=64, num_chain=64, learning_iter=6, mcmc_per_flow_steps=0.5, hutchs=True, step_size=1e-4,
                  fourier_dim=16, hidden_x=[32, 32], hidden_t=[32, 32], hidden_xt=[32, 32], seed=1024, eval_iter=1)
    out = loop.run(targets.PhiFour(64), loop.default_args(**common))
    res, res_, ex = E.run(D.PhiFour(64), loop.default_args(**common), None, log_every=1000, return_extras=True)
    tr, m = out["trace"], ex["metrics"]
    assert len(tr["n_att"]) == 4                                           # counts 1, 2, 4, 5 are flow steps; 3 and 6 MALA
    c = ex["engine"].ctx.counters()
    assert c["ode_solves"] >= 4 * 2 * 64 and c["mala_chain_steps"] == 2 * 64
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=5e-3)             # same tolerance as the K >= 1 loop test
    np.testing.assert_allclose(ex["lrs"], tr["learning_rate"], rtol=1e-12)
    mala_it = [2, 5]
    np.testing.assert_allclose(m[mala_it, 1], np.array(tr["acc_mean"])[mala_it], atol=2e-3)
    g, o = ex["states"].position.cpu().numpy().astype(np.float64), out["states"].position
    dmax = np.abs(g - o).max(1)
    assert (dmax > 0.05).sum() <= 6 and dmax[dmax <= 0.05].max() < 2e-2
    ex["engine"].close()
