"""GPU parity of the whole MFM loop (exe_flow_matching.run) against the oracle loop on the same seed:
loss trace, learning rate, annealing temperatures, acceptance statistics and sample moments."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _args(**kw):
    from oracle import loop
    return loop.default_args(**kw)


def _run_both(example, d, B, iters, K, hutch=True, width=32, fourier_dim=16, **kw):
    from mfm_amd import distributions as D, exe_flow_matching as E
    from oracle import loop, targets
    common = dict(example=example, dim=d, num_chain=B, learning_iter=iters, mcmc_per_flow_steps=float(K), hutchs=hutch,
                  fourier_dim=fourier_dim, seed=1024, eval_iter=1, **kw)
    from tests import gpu_util as gu
    common.update(gu.hidden_lists(width))
    if example == "phi-four":
        dg, do = D.PhiFour(d), targets.PhiFour(d)
        tg = to = None
    elif example == "pines":
        dg = D.LogGaussianCoxPines(d)
        do = targets.LogGaussianCoxPines(d, dg.counts)
        tg = to = None
    else:
        modes, covs, w = 8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4
        dg, do = D.GaussianMixture(modes, covs, w), targets.GaussianMixture(modes, covs, w)
        tg, to = dg.sample_model, do.sample_model_rows
    out = loop.run(do, _args(**common), target_gn=to)
    res, res_, ex = E.run(dg, _args(**common), tg, log_every=1000, return_extras=True)
    return out, res, ex


def test_phi4_loop_matches_oracle():
    out, res, ex = _run_both("phi-four", 64, 64, 12, 3, step_size=1e-4)
    tr, m = out["trace"], ex["metrics"]
    # loss is a sum over 64*64 residuals of O(1): float32 forward vs float64 oracle, same noise
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=1e-6)      # before the first flow step: same chains, same noise
    # after it: an accept/reject decision of the flow-MH step (uniform vs an UNCLIPPED ratio of exp(O(1e3)) log-densities)
    # may differ between float32 and float64 for a borderline chain; one flipped chain of 64 moves the summed loss by a few
    # 1e-3 relative (observed 2e-3 when last-bit differences in the AdamW update flipped one), never more than 1/64
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=5e-3)
    np.testing.assert_allclose(ex["lrs"], tr["learning_rate"], rtol=1e-12)
    np.testing.assert_allclose(ex["betas"], tr["beta"], rtol=2e-3)
    mala_it = [i for i in range(12) if (i + 1) % 4 != 0]
    np.testing.assert_allclose(m[mala_it, 1], np.array(tr["acc_mean"])[mala_it], atol=2e-3)
    np.testing.assert_allclose(m[mala_it, 2], np.array(tr["acc_std"])[mala_it], atol=5e-3)
    g = ex["states"].position.cpu().numpy().astype(np.float64)
    o = out["states"].position
    # chains whose three flow-MH decisions all agree with the oracle's end up within the Dopri5 tolerance of it; a chain
    # with a flipped borderline decision is O(1) away, and the first flip perturbs every later batch gradient (so the second
    # and third flow steps integrate a field that differs by ~1e-4 for ALL chains, which flips further borderline cases).
    # The single-step parity of the decision itself is test_flow_rwmh_step_matches_oracle; here allow 6 of 64 chains and
    # compare the rest tightly.
    dmax = np.abs(g - o).max(1)
    flipped = dmax > 0.05
    assert flipped.sum() <= 6, dmax
    same = ~flipped
    assert dmax[same].max() < 2e-2, dmax[same].max()
    np.testing.assert_allclose(g[same].mean(0), o[same].mean(0), atol=5e-3)
    np.testing.assert_allclose((g[same] ** 2).mean(), (o[same] ** 2).mean(), rtol=2e-3)
    np.testing.assert_allclose(ex["states"].logdensity.cpu().numpy()[same].mean(), out["states"].logdensity[same].mean(), rtol=2e-3)
    # parameters after 12 AdamW steps (a flipped chain changes 1/64 of the later batches: looser bound then).  Adam's first
    # updates are lr * g / |g|: a parameter whose gradient is within float32 rounding of zero moves by up to +- lr per step in
    # either run, so the bound is a fraction of 12 lr, not a rounding bound (observed 3e-4 .. 6e-4 over builds whose target
    # gradient differs in the last bit)
    from tests import gpu_util as gu
    po = gu.flat_params(out["state"].params)
    pg = ex["engine"].ctx.get_params()
    assert np.abs(pg - po).max() < (1e-3 if not flipped.any() else 3e-3) * max(1.0, np.abs(po).max())
    s = ex["engine"].ctx.opt_state()
    assert (s["step"], s["count"]) == (out["state"].step, out["state"].count)
    assert np.isfinite(res[0])
    ex["engine"].close()


def test_phi4_loop_at_the_reference_default_lattice_on_the_shape_specialised_solver():
    """The reference's phi-four shape (d = 64, hidden 128, 128 Fourier frequencies: multi_modal.py:52-55,159,178-180) with --hutch:
    its flow steps and the final transform run on the shape-specialised kernels zero-padded to their 128-wide tile (round 4)."""
    out, res, ex = _run_both("phi-four", 64, 32, 8, 3, step_size=1e-4, width=128, fourier_dim=128)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=1e-6)
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=1e-2)           # (a flipped borderline flow-MH decision of 32 chains: see above)
    np.testing.assert_allclose(ex["betas"], tr["beta"], rtol=2e-3)
    g = ex["states"].position.cpu().numpy().astype(np.float64)
    dmax = np.abs(g - out["states"].position).max(1)
    assert (dmax > 0.05).sum() <= 3 and dmax[dmax <= 0.05].max() < 2e-2, dmax
    assert np.isfinite(res[0])
    ex["engine"].close()


@pytest.mark.parametrize("family", ["tile", "wide"])
def test_loop_with_a_chain_count_that_is_not_a_multiple_of_16(monkeypatch, family):
    """--num_chain takes any integer in the reference (multi_modal.py:169).  40 chains: the shard is padded to 48 rows
    (mfm_config.n_chain_valid = 40); the padding rows must not show in the loss, the gradient (through the parameters), the
    annealing temperatures (ESS over the chains) or the acceptance statistics.  Both kernel families."""
    from mfm_amd import _lib
    if family == "wide":
        monkeypatch.setenv("MFM_KERNEL_FAMILY", str(_lib.FAMILY_WIDE))
    out, res, ex = _run_both("phi-four", 64, 40, 9, 3, step_size=1e-4)
    tr, m = out["trace"], ex["metrics"]
    assert ex["states"].position.shape[0] == 48 and ex["engine"].n_valid == 40
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=1e-6)      # before the first flow step: same chains, same noise
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=1e-2)           # (one flipped chain of 40 after a flow step: test_phi4_loop_matches_oracle)
    np.testing.assert_allclose(ex["betas"], tr["beta"], rtol=2e-3)
    mala_it = [i for i in range(9) if (i + 1) % 4 != 0]
    np.testing.assert_allclose(m[mala_it, 1], np.array(tr["acc_mean"])[mala_it], atol=2e-3)
    np.testing.assert_allclose(m[mala_it, 2], np.array(tr["acc_std"])[mala_it], atol=5e-3)
    from tests import gpu_util as gu
    po = gu.flat_params(out["state"].params)
    pg = ex["engine"].ctx.get_params()
    assert np.abs(pg - po).max() < 3e-3 * max(1.0, np.abs(po).max())
    g = ex["states"].position.cpu().numpy().astype(np.float64)[:40]
    dmax = np.abs(g - out["states"].position).max(1)
    assert (dmax > 0.05).sum() <= 4 and dmax[dmax <= 0.05].max() < 2e-2
    assert ex["final"]["flow_samples"].shape[0] == 40 and np.isfinite(res[0])
    ex["engine"].close()


@pytest.mark.parametrize("hutch", [True, False])
def test_four_mode_loop_matches_oracle(hutch):
    """BASELINE configs[0] in miniature; hutch=False is the reference's default (exact trace)."""
    out, res, ex = _run_both("4-mode", 2, 64, 8, 3, hutch=hutch, step_size=0.2)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=5e-3)
    np.testing.assert_allclose(m[:, 3], tr["target_loss"], rtol=1e-3)          # eval_step on the exact samples
    np.testing.assert_allclose(ex["betas"], tr["beta"], rtol=1e-3)
    ex["engine"].close()


def test_four_mode_loop_with_an_eval_set_that_is_not_a_multiple_of_16():
    """--num_chain 50 on a mixture example: eval_step (exe_flow_matching.py:370-374, :444-446) runs on num_chain * eval_iter = 150
    exact samples every iteration -- 9 full 16-row tiles and a partial one, which mfm_fm_loss stages inside the library (it used
    to decline any n % 16 != 0, so every such run aborted at its first iteration with the default eval_iter)."""
    from mfm_amd import distributions as D, exe_flow_matching as E
    from oracle import loop, targets
    from tests import gpu_util as gu
    common = dict(example="4-mode", dim=2, num_chain=50, learning_iter=5, mcmc_per_flow_steps=3.0, hutchs=False, fourier_dim=16, seed=1024,
                  eval_iter=3, step_size=0.2, **gu.hidden_lists(32))
    modes, covs, w = 8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4
    dg, do = D.GaussianMixture(modes, covs, w), targets.GaussianMixture(modes, covs, w)
    out = loop.run(do, _args(**common), target_gn=do.sample_model_rows)
    res, res_, ex = E.run(dg, _args(**common), dg.sample_model, log_every=1000, return_extras=True)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=1e-5)
    np.testing.assert_allclose(m[:, 3], tr["target_loss"], rtol=1e-3)
    ex["engine"].close()


def test_pines_loop_matches_oracle():
    """Log-Gaussian Cox process on a 16 x 16 grid (dim > 128, so the +-1 clip of grad log pi is active as in the
    reference's 40 x 40 example, whose hidden width of 1024 does not fit the 16-chain LDS tile): annealing with a prior
    term, MALA through the K^-1 kernel, flow steps."""
    out, res, ex = _run_both("pines", 256, 32, 9, 3, step_size=0.01)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=1e-5)
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=5e-3)
    np.testing.assert_allclose(ex["betas"], tr["beta"], rtol=2e-3)
    g = ex["states"].position.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(g.mean(0), out["states"].position.mean(0), atol=2e-2)
    np.testing.assert_allclose(ex["states"].logdensity.cpu().numpy().mean(), out["states"].logdensity.mean(), rtol=2e-3)
    assert np.isfinite(res[0])
    ex["engine"].close()


def test_exact_sample_training_matches_oracle():
    """mcmc_per_flow_steps < 0 (exe_flow_matching.py:328,382-386): every iteration trains on fresh exact samples of the
    target; no MCMC, beta = 1, acceptance is NaN."""
    out, res, ex = _run_both("4-mode", 2, 64, 8, -1, hutch=False, step_size=0.2)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=2e-5)
    np.testing.assert_allclose(m[:, 3], tr["target_loss"], rtol=2e-4)
    assert np.isnan(m[:, 1]).all() and np.isnan(np.array(tr["acc_mean"])).all()
    assert ex["betas"][0] == 1.0
    ex["engine"].close()


def test_cis_loop_runs_and_matches_oracle_before_first_flow_step():
    """num_importance_samples > 0 selects conditional importance sampling as the flow step (:298)."""
    out, res, ex = _run_both("4-mode", 2, 64, 6, 2, hutch=False, step_size=0.2, num_importance_samples=3)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:2, 0], tr["loss"][:2], rtol=1e-5)          # MALA iterations before the first CIS step
    assert np.isfinite(m[:, 0]).all() and np.isfinite(tr["loss"]).all()
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=0.05)               # afterwards: near-tie categorical draws may differ
    ex["engine"].close()


def test_run_with_noise_prefetch_is_bit_identical(monkeypatch):
    """Headline network shape (the shape-specialised flow kernel with the noise tail, noise.hip): run() with the draws of
    the coming iterations produced inside the flow step equals run() drawing in line, bit for bit."""
    from mfm_amd import distributions as D, exe_flow_matching as E
    common = dict(example="phi-four", dim=256, num_chain=64, learning_iter=11, mcmc_per_flow_steps=3.0, hutchs=True, seed=7,
                  eval_iter=1, step_size=1e-4)
    out = []
    for off in (True, False):
        if off:
            monkeypatch.setenv("MFM_NO_PREFETCH", "1")
        else:
            monkeypatch.delenv("MFM_NO_PREFETCH")
        res, res_, ex = E.run(D.PhiFour(256), _args(**common), None, log_every=1000, return_extras=True)
        out.append((ex["metrics"].copy(), ex["states"].position.cpu().numpy().copy(), ex["engine"].ctx.get_params()))
        ex["engine"].close()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("example", ["phi-four", "4-mode"])
def test_run_one_call_iterations_equal_the_separate_calls(monkeypatch, example):
    """run() on one rank issues generator + train_step as one library call per iteration (mfm_train_iter: the MALA step inside the
    training kernel); MFM_SPLIT_CALLS=1 keeps the multi-rank call sequence (mfm_mala_step / mfm_flow_step, mfm_fm_loss_grad,
    mfm_adamw_step).  Same traces, chains and parameters, bit for bit -- with the draws prefetched by the flow step (phi-four,
    headline shape) and drawn in line (4-mode)."""
    from mfm_amd import distributions as D, exe_flow_matching as E
    if example == "phi-four":
        dist = D.PhiFour(256)
        common = dict(example="phi-four", dim=256, num_chain=64, learning_iter=11, mcmc_per_flow_steps=3.0, hutchs=True, seed=7, eval_iter=1, step_size=1e-4)
    else:
        dist = D.GaussianMixture(8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4)
        common = dict(example="4-mode", dim=2, num_chain=64, learning_iter=11, mcmc_per_flow_steps=3.0, hutchs=False, seed=7, eval_iter=2, step_size=0.2)
    out = []
    for split in (True, False):
        if split:
            monkeypatch.setenv("MFM_SPLIT_CALLS", "1")
        else:
            monkeypatch.delenv("MFM_SPLIT_CALLS")
        res, res_, ex = E.run(dist, _args(**common), dist.sample_model if example == "4-mode" else None, log_every=1000, return_extras=True)
        out.append((ex["metrics"].copy(), ex["states"].position.cpu().numpy().copy(), ex["engine"].ctx.get_params()))
        ex["engine"].close()
    assert np.isfinite(out[0][0][:, :3]).all()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("force_exchange", [False, True])
def test_train_iter_with_a_non_finite_gradient_equals_the_separate_calls(monkeypatch, force_exchange):
    """apply_if_finite through mfm_train_iter's one-launch reduction + optimizer (optim.hip: reduce_adamw_kernel): a NaN position
    in iteration 3 makes that iteration's gradient non-finite -- the update is skipped, notfinite_count advances and is reset by
    the next finite gradient, exactly as with mfm_fm_loss_grad + mfm_adamw_step.  force_exchange runs EVERY iteration through the
    grid-wide decision on the totals (the path taken when a partial sum is huge or non-finite)."""
    import torch
    from oracle import prng
    from tests import gpu_util as gu
    from mfm_amd._lib import FLOW_RWMH
    if force_exchange:
        monkeypatch.setenv("MFM_DEBUG_FORCE_EXCHANGE", "1")
    args, dist, k, model, state = gu.phi4_setup(d=256, B=64, learning_iter=20)
    params = gu.rand_params(model, seed=3, out_scale=0.05)
    x0 = dist.init_params.astype(np.float32)
    out = []
    for one_call in (False, True):
        ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
        pos = torch.from_numpy(x0).cuda(); logp = torch.empty(64, device="cuda", dtype=torch.float64); grad = torch.empty_like(pos)
        acc = torch.empty(64, device="cuda"); loss = torch.zeros(1, device="cuda", dtype=torch.float64); grads = torch.zeros(ctx.n_params, device="cuda")
        ctx.mala_init(pos, 1.0, logp, grad)
        ks, tr = prng.PRNGKey(5), []
        for count in range(1, 6):
            ks, kg, kt = prng.split(ks, 3)
            if count == 3:
                saved = pos[5, 7].item(); pos[5, 7] = float("nan")
            if one_call:
                ctx.train_iter(count, 100, FLOW_RWMH, kg, kt, 1.0, args.step_size, pos, logp, grad, loss, grads, acc=acc)
            else:
                ctx.mala_step(kg, 1.0, args.step_size, pos, logp, grad, acc); ctx.fm_loss_grad(kt, pos, loss, grads); ctx.adamw_step(grads)
            tr.append((ctx.opt_state(), ctx.get_params().copy(), loss.item()))
            if count == 3:
                pos[5, 7] = saved
                ctx.mala_init(pos, 1.0, logp, grad)
        out.append(tr); ctx.close()
    for i, ((s0, p0, l0), (s1, p1, l1)) in enumerate(zip(*out)):
        assert s0 == s1, (i, s0, s1)
        np.testing.assert_array_equal(p0, p1)
        assert (np.isnan(l0) and np.isnan(l1)) or l0 == l1
    s3, s4 = out[1][2][0], out[1][3][0]
    assert s3["notfinite_count"] == 1 and s3["last_applied"] == 0 and s3["count"] == 2 and s3["step"] == 3
    assert s4["notfinite_count"] == 0 and s4["last_applied"] == 1 and s4["count"] == 3
    np.testing.assert_array_equal(out[1][1][1], out[1][2][1])          # the skipped update left the parameters alone


@pytest.mark.parametrize("case", ["phi4_64", "phi4_256_headline", "gmm4", "gmm16"])
def test_train_iter_equals_the_separate_calls(case):
    """mfm_train_iter (generator :300-314 + train_step :362-368 in one call; its MALA step rides in the training kernel's
    workgroups, fm.hip: fm_fwd_bwd_kernel<.., MALA>) against the same iterations composed from mfm_mala_step / mfm_flow_step /
    mfm_fm_loss_grad / mfm_adamw_step: bit-identical chains, acceptance probabilities, losses and parameters over a schedule with
    a flow iteration in it (K = 3: counts 4 and 8 are flow steps) -- on the generic tile, the headline's shape-specialised
    instance and the 2-d mixtures (one mode per lane)."""
    import torch
    from oracle import prng
    from tests import gpu_util as gu
    from mfm_amd._lib import FLOW_RWMH, MfmError
    if case == "phi4_64":
        args, dist, k, model, state = gu.phi4_setup(d=64, B=64, hidden=32, F=16, learning_iter=20)
    elif case == "phi4_256_headline":
        args, dist, k, model, state = gu.phi4_setup(d=256, B=64, learning_iter=20)
    elif case == "gmm4":
        args, dist, k, model, state = gu.gmm4_setup(B=64, learning_iter=20)
    else:
        args, dist, k, model, state = gu.gmm16_setup(B=64, learning_iter=20)
    params = gu.rand_params(model, seed=3, out_scale=0.05)
    x0 = dist.init_params.astype(np.float32) if hasattr(dist, "init_params") else np.random.default_rng(4).normal(size=(64, dist.dim)).astype(np.float32)
    K = 3
    out = []
    for fused in (False, True):
        ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
        pos = torch.from_numpy(x0).cuda(); logp = torch.empty(64, device="cuda", dtype=torch.float64); grad = torch.empty_like(pos)
        acc = torch.empty(64, device="cuda", dtype=torch.float32); nst = torch.zeros(64, device="cuda", dtype=torch.int32)
        loss = torch.zeros(1, device="cuda", dtype=torch.float64); grads = torch.zeros(ctx.n_params, device="cuda")
        ctx.mala_init(pos, 1.0, logp, grad)
        losses, ks = [], prng.PRNGKey(5)
        for count in range(1, 10):
            ks, kg, kt = prng.split(ks, 3)
            if fused:
                ctx.train_iter(count, K, FLOW_RWMH, kg, kt, 1.0, args.step_size, pos, logp, grad, loss, grads, acc=acc, nsteps=nst)
            else:
                if count % (K + 1) == 0:
                    ctx.flow_step(FLOW_RWMH, kg, 1.0, pos, logp, grad, acc, None, None, nst)
                else:
                    ctx.mala_step(kg, 1.0, args.step_size, pos, logp, grad, acc)
                ctx.fm_loss_grad(kt, pos, loss, grads)
                ctx.adamw_step(grads)
            losses.append(loss.item())
        out.append((pos.cpu().numpy(), logp.cpu().numpy(), np.array(losses), ctx.get_params(), ctx.opt_state(), nst.cpu().numpy(), grad.cpu().numpy(),
                    acc.cpu().numpy()))
        if fused:
            with pytest.raises(MfmError, match="mcmc_per_flow_steps >= 1"):
                ctx.train_iter(1, 0, FLOW_RWMH, kg, kt, 1.0, args.step_size, pos, logp, grad, loss, grads)
        ctx.close()
    a, b = out
    assert a[4] == b[4] and a[4]["step"] == 9
    assert b[5].min() > 0                                     # the flow iterations ran
    for u, v in zip(a[:4] + a[5:], b[:4] + b[5:]):
        np.testing.assert_array_equal(u, v)
