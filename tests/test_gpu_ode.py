"""GPU parity: Dopri5 CNF transforms with Hutchinson log-det and the flow-MH step -- vs the float64 oracle."""
import numpy as np
import pytest

from oracle import flow, mala, ode, prng, targets

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def _setup(d, B, hidden, F, seed=9, out_scale=0.5):
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F)
    params = gu.rand_params(model, seed=seed, out_scale=out_scale)
    # gate * clip(grad log pi): |grad| ~ 1e3, so the clipped term is nearly a sign function of x (and unclipped for
    # d <= 128 it is huge): tame the gate layer so the adaptive solver takes tens, not hundreds, of steps
    gs = 1e-3
    params[4]["kernel"] *= gs; params[4]["bias"] *= gs
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    return args, dist, model, params, ctx


@pytest.mark.parametrize("d,hidden,F", [(256, 128, 128), (128, 128, 128), (64, 128, 128), (192, 128, 128), (64, 32, 16)])     # the shape-specialised solver at its two tile widths and zero-padded to them (d = 64: the reference's phi-four default), one generic
def test_ode_transform_and_inverse_match_oracle(d, hidden, F):
    import torch
    B = 32
    args, dist, model, params, ctx = _setup(d, B, hidden, F)
    x32 = dist.init_params.astype(np.float32)
    keys = prng.split(prng.PRNGKey(21), B)
    for direction, fn in ((1, ode.transform_and_logdet), (-1, ode.inverse_and_logdet)):
        st = {}
        y_o, l_o = fn(model, params, keys, x32.astype(np.float64), True, args.rtol, args.atol, args.mxstep, stats=st)
        out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda")
        ns = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.ode_transform(direction, _dev(x32), out, ldj, keys=_dev(keys.astype(np.uint32).view(np.int32)), nsteps=ns)
        y, l, n = out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy()
        assert np.abs(y - x32).max() > 1e-2                      # the flow actually moves the points
        # two adaptive solves of a ReLU field at rtol = atol = 1e-5 (float32 vs float64 controller decisions):
        # each is within ~1e-5 * sqrt(steps) of the truth, so they agree to a few 1e-4; bound: 2e-3
        assert np.abs(y - y_o).max() < 2e-3 * max(1.0, np.abs(y_o).max()), np.abs(y - y_o).max()
        assert np.abs(y - y_o).mean() < 1e-4, np.abs(y - y_o).mean()
        # the Hutchinson log-det integrates z.(Jz) of a piecewise-linear field: it is the least accurate component
        # of BOTH solvers (the oracle itself is ~3e-2 from a rtol=1e-8 solve on this setup, tools/dbg/debug_ode.py)
        assert np.abs(l - l_o).max() < 5e-2 * max(1.0, np.abs(l_o).max()), (np.abs(l - l_o).max(), np.abs(l_o).max())
        assert np.abs(l - l_o).mean() < 1e-2 * max(1.0, np.abs(l_o).max())
        # own controllers: a float32 and a float64 error norm round differently near ratio = 1 and at ReLU kinks, so individual
        # accept / reject decisions flip (the step-for-step comparison on ONE step sequence is tests/test_gpu_replay.py);
        # the attempt statistics agree
        # (the exact-match share alone was >= 0.5 until cec7b18 changed float associations in the shape-specialised solver --
        # out-layer tangent pull-back, 4-row tail path, DPP sums -- and a few more borderline decisions flipped.
        # What a drifting controller would show is a BIAS or a wide spread of dn, so those are bounded instead.)
        sd = n.astype(np.int64) - st["n_attempted"]
        dn = np.abs(sd)
        rel = dn / st["n_attempted"]
        print(f"natural controllers d={d} dir={direction}: exact attempt counts {(dn == 0).mean():.2f}, |dn| / n: p90 {np.quantile(rel, 0.9):.3f} max {rel.max():.3f}, "
              f"mean signed dn {sd.mean():+.2f} of {st['n_attempted'].mean():.0f}")
        # measured: d = 256 (clipped grad log pi, ~35 attempts) 0.91 exact, no chain beyond 14 %; d = 128 (unclipped, ~30 attempts)
        # 0.44 exact, p90 of |dn| / n 0.18; signed mean -0.2 .. -0.3 attempts (< 1 % of n: no drift of the controller)
        min_exact = 0.8 if d == 256 else 0.3
        assert (dn == 0).mean() >= min_exact and np.quantile(rel, 0.9) < 0.25 and abs(sd.mean()) < 0.02 * st["n_attempted"].mean(), ((dn == 0).mean(), np.quantile(rel, 0.9), sd.mean())
        assert abs(n.mean() - st["n_attempted"].mean()) < 0.1 * st["n_attempted"].mean(), (n, st["n_attempted"])
    # shared key (final sampling, exe_flow_matching.py:455)
    y_o, l_o = ode.transform_and_logdet(model, params, prng.PRNGKey(4), x32.astype(np.float64), True, args.rtol, args.atol, args.mxstep)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda")
    ctx.ode_transform(1, _dev(x32), out, ldj, key=prng.PRNGKey(4))
    assert np.abs(out.cpu().numpy() - y_o).max() < 2e-3 * max(1.0, np.abs(y_o).max())
    # round trip: inverse(transform(x)) == x (size-independent property, deterministic given the probe keys)
    back = torch.empty(B, d, device="cuda"); l2 = torch.empty(B, device="cuda")
    ctx.ode_transform(-1, out, back, l2, key=prng.PRNGKey(4))
    assert np.abs(back.cpu().numpy() - x32).max() < 2e-3
    np.testing.assert_allclose(l2.cpu().numpy(), -ldj.cpu().numpy(), atol=0.15 * max(1.0, np.abs(l_o).max()))
    ctx.close()


def test_ode_identity_flow_at_zero_init():
    import torch
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=256, B=16)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=state.params)
    x32 = dist.init_params.astype(np.float32)
    out = torch.empty(16, 256, device="cuda"); ldj = torch.empty(16, device="cuda"); ns = torch.empty(16, dtype=torch.int32, device="cuda")
    ctx.ode_transform(1, _dev(x32), out, ldj, key=prng.PRNGKey(1), nsteps=ns)
    np.testing.assert_array_equal(out.cpu().numpy(), x32)
    np.testing.assert_array_equal(ldj.cpu().numpy(), 0)
    ctx.close()


@pytest.mark.parametrize("d,hidden,F", [(256, 128, 128), (128, 128, 128), (64, 128, 128), (192, 128, 128), (64, 32, 16)])
def test_flow_rwmh_step_matches_oracle(d, hidden, F):
    import torch
    B = 32
    args, dist, model, params, ctx = _setup(d, B, hidden, F, out_scale=0.05)
    beta = 0.8
    vg = targets.Tempered(dist, beta).value_and_grad
    x32 = dist.init_params.astype(np.float32)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    st = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    key = prng.PRNGKey(31)
    stats = {}
    new, info = flow.rwmh_step(prng.split(key, B), st, vg, model, params, args, stats)
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda")
    prop = torch.empty(B, d, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    from mfm_amd import _lib
    ctx.flow_step(_lib.FLOW_RWMH, key, beta, pos, logp, grad, acc, isacc, prop, ns)
    p = prop.cpu().numpy()
    assert np.abs(p - info.proposed_position).max() < 5e-3 and np.abs(p - info.proposed_position).mean() < 2e-4
    # log acceptance ratio: compare in log space (values are O(1e3) apart in magnitude for a random network)
    with np.errstate(divide="ignore"):
        la_g, la_o = np.log(acc.cpu().numpy().astype(np.float64)), np.log(info.acceptance_rate)
    fin = np.isfinite(la_g) & np.isfinite(la_o)
    if fin.any():
        dla = np.abs(la_g[fin] - la_o[fin])
        print(f"natural-controller flow step: |d log alpha| median {np.median(dla):.2e} max {dla.max():.2e} (|log alpha| up to {np.abs(la_o[fin]).max():.1f})")
        assert dla.max() < 0.05                      # two adaptive solves with their own controllers: measured 2.5e-3 .. 3.6e-3 of |log alpha| <= 97
    np.testing.assert_array_equal(isacc.cpu().numpy().astype(bool)[~fin | (np.abs(la_o) > 1)], info.is_accepted[~fin | (np.abs(la_o) > 1)])
    tot = stats["n_att_inv"] + stats["n_att_fwd"]
    dn = np.abs(ns.cpu().numpy() - tot)
    assert (dn == 0).mean() >= 0.5 and abs(ns.float().mean().item() - tot.mean()) < 0.1 * tot.mean()
    ctx.close()


def test_exact_trace_transform_matches_oracle_on_mixture():
    """No --hutch (the default of the two mixture examples): trace(jacfwd(v)) by d basis probes per RHS evaluation."""
    import torch
    from tests import gpu_util as gu
    B, d = 32, 2
    args, dist, k, model, state = gu.gmm4_setup(B=B, hutchs=False)
    params = gu.rand_params(model, seed=9, out_scale=0.3)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x32 = dist.init_params.astype(np.float32)
    for direction, fn in ((1, ode.transform_and_logdet), (-1, ode.inverse_and_logdet)):
        st = {}
        y_o, l_o = fn(model, params, None, x32.astype(np.float64), False, args.rtol, args.atol, args.mxstep, stats=st)
        out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.ode_transform(direction, _dev(x32), out, ldj, key=prng.PRNGKey(0), nsteps=ns)
        assert np.abs(out.cpu().numpy() - y_o).max() < 2e-3 * max(1.0, np.abs(y_o).max())
        assert np.abs(ldj.cpu().numpy() - l_o).max() < 1e-2 * max(1.0, np.abs(l_o).max())
        assert abs(ns.float().mean().item() - st["n_attempted"].mean()) < 0.1 * st["n_attempted"].mean()
    ctx.close()


def test_flow_imh_step_matches_oracle():
    """Independent MH in latent space (num_importance_samples < 0, exe_flow_matching.py:246-260)."""
    import torch
    from mfm_amd import _lib
    d, B = 64, 32
    args, dist, model, params, ctx = _setup(d, B, 32, 16, out_scale=0.05)
    beta = 1.0
    vg = targets.Tempered(dist, beta).value_and_grad
    x32 = dist.init_params.astype(np.float32)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    st = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    key = prng.PRNGKey(77)
    new, info = flow.imh_step(prng.split(key, B), st, vg, model, params, args)
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda"); prop = torch.empty(B, d, device="cuda")
    ctx.flow_step(_lib.FLOW_IMH, key, beta, pos, logp, grad, acc, isacc, prop, None)
    p = prop.cpu().numpy()
    assert np.abs(p - info.proposed_position).max() < 5e-3 and np.abs(p - info.proposed_position).mean() < 2e-4
    with np.errstate(divide="ignore"):
        la_g, la_o = np.log(acc.cpu().numpy().astype(np.float64)), np.log(info.acceptance_rate)
    fin = np.isfinite(la_g) & np.isfinite(la_o)
    if fin.any():
        dla = np.abs(la_g[fin] - la_o[fin])
        print(f"natural-controller flow step: |d log alpha| median {np.median(dla):.2e} max {dla.max():.2e} (|log alpha| up to {np.abs(la_o[fin]).max():.1f})")
        assert dla.max() < 0.05                      # two adaptive solves with their own controllers: measured 2.5e-3 .. 3.6e-3 of |log alpha| <= 97
    sure = ~fin | (np.abs(la_o) > 1)
    np.testing.assert_array_equal(isacc.cpu().numpy().astype(bool)[sure], info.is_accepted[sure])
    ctx.close()


def test_lgcp_transform_and_flow_step_match_oracle():
    """Log-Gaussian Cox target (8 x 8 grid): grad log pi and its Hessian-vector product need the K^-1 contraction
    inside every RHS evaluation and in the MH target evaluation."""
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    B, d = 32, 64
    args, dist, k, model, state = gu.lgcp_setup(n=8, B=B)
    params = gu.rand_params(model, seed=9, out_scale=0.3)
    params[4]["kernel"] *= 0.05; params[4]["bias"] *= 0.05
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x32 = dist.init_params.astype(np.float32)
    keys = prng.split(prng.PRNGKey(21), B)
    for direction, fn in ((1, ode.transform_and_logdet), (-1, ode.inverse_and_logdet)):
        st = {}
        y_o, l_o = fn(model, params, keys, x32.astype(np.float64), True, args.rtol, args.atol, args.mxstep, stats=st)
        out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.ode_transform(direction, _dev(x32), out, ldj, keys=_dev(keys.astype(np.uint32).view(np.int32)), nsteps=ns)
        assert np.abs(out.cpu().numpy() - y_o).max() < 2e-3 * max(1.0, np.abs(y_o).max())
        assert np.abs(ldj.cpu().numpy() - l_o).max() < 5e-2 * max(1.0, np.abs(l_o).max())
        assert abs(ns.float().mean().item() - st["n_attempted"].mean()) < 0.1 * st["n_attempted"].mean()
    beta = 0.7
    vg = targets.Tempered(dist, beta).value_and_grad
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    st = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    key = prng.PRNGKey(31)
    new, info = flow.rwmh_step(prng.split(key, B), st, vg, model, params, args)
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda"); prop = torch.empty(B, d, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, key, beta, pos, logp, grad, acc, isacc, prop, None)
    p = prop.cpu().numpy()
    assert np.abs(p - info.proposed_position).max() < 5e-3 * max(1.0, np.abs(p).max())
    with np.errstate(divide="ignore"):
        la_g, la_o = np.log(acc.cpu().numpy().astype(np.float64)), np.log(info.acceptance_rate)
    fin = np.isfinite(la_g) & np.isfinite(la_o)
    dla = np.abs(la_g[fin] - la_o[fin])
    print(f"natural-controller flow step (mixture): |d log alpha| median {np.median(dla):.2e} max {dla.max():.2e}")
    assert fin.any() and dla.max() < 0.15         # measured 4.4e-2 (median 2e-4)
    sure = ~fin | (np.abs(la_o) > 1)
    np.testing.assert_array_equal(isacc.cpu().numpy().astype(bool)[sure], info.is_accepted[sure])
    same = isacc.cpu().numpy().astype(bool) == info.is_accepted
    np.testing.assert_allclose(logp.cpu().numpy()[same], new.logdensity[same], rtol=1e-4, atol=5e-2)
    np.testing.assert_allclose(grad.cpu().numpy()[same], new.logdensity_grad[same], rtol=1e-3, atol=5e-2)
    ctx.close()


def test_flow_cis_step_matches_oracle():
    """Conditional importance sampling (num_importance_samples > 0, exe_flow_matching.py:280-296): host composition of the
    ODE / target kernels + mfm_normal_rows + mfm_cis_select vs the oracle step on the same keys (4-mode mixture: its
    log-densities are O(10), so the un-stabilised weights of the reference neither overflow nor vanish)."""
    import torch
    from mfm_amd import random as jr
    from tests import gpu_util as gu
    d, B, n_is = 2, 32, 5
    args, dist, k, model, state = gu.gmm4_setup(B=B, hutchs=False, num_importance_samples=n_is)
    params = gu.rand_params(model, seed=9, out_scale=0.3)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, max_eval=B * n_is)
    beta = 0.9
    vg = targets.Tempered(dist, beta).value_and_grad
    x32 = dist.init_params.astype(np.float32)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    g_before = grad.cpu().numpy().copy()
    st = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), g_before.astype(np.float64))
    key = prng.PRNGKey(55)
    stats = {}
    new, info = flow.cis_step(prng.split(key, B), st, vg, model, params, args, stats)
    # device composition (what exe_flow_matching.create_train_data_gn does)
    kk = jr.split_rows(jr.split(key, B), 4)
    kd = lambda a: _dev(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32))
    u0 = torch.empty(B, d, device="cuda"); vol0 = torch.empty(B, device="cuda")
    ctx.ode_transform(-1, pos, u0, vol0, keys=kd(kk[:, 1]))
    ks = jr.split_rows(kk[:, 0], n_is).reshape(B * n_is, 2); kh = jr.split_rows(kk[:, 2], n_is).reshape(B * n_is, 2)
    refs = torch.empty(B * n_is, d, device="cuda"); ctx.normal_rows(kd(ks), refs)
    np.testing.assert_allclose(refs.cpu().numpy(), stats["refs"], atol=1e-6)
    xs = torch.empty_like(refs); vols = torch.empty(B * n_is, device="cuda")
    ctx.ode_transform(1, refs, xs, vols, keys=kd(kh))
    assert np.abs(xs.cpu().numpy() - stats["xs"]).max() < 5e-3 * max(1.0, np.abs(stats["xs"]).max())
    lps = torch.empty(B * n_is, dtype=torch.float64, device="cuda"); gtmp = torch.empty(B, d, device="cuda"); ltmp = torch.empty(B, dtype=torch.float64, device="cuda")
    for j in range(n_is):
        ctx.mala_init(xs[j * B:(j + 1) * B].contiguous(), beta, ltmp, gtmp); lps[j * B:(j + 1) * B] = ltmp
    np.testing.assert_allclose(lps.cpu().numpy(), stats["lps"], rtol=1e-3, atol=2e-2)
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda"); prop = torch.empty(B, d, device="cuda"); w = torch.empty(B, device="cuda")
    ctx.cis_select(key, n_is, u0, vol0, refs, xs, vols, lps, pos, logp, acc, isacc, prop, w)
    # the categorical draw is an inverse-CDF search: compare the decisions whose uniform is not within 1e-2 of a boundary
    norm, ch = stats["norm"], stats["choice"]
    assert np.isfinite(norm).all()
    cum = np.cumsum(norm, axis=1)
    r = np.array([cum[b, -1] * (1.0 - prng.uniform(prng.split(prng.split(key, B)[b], 4)[3])) for b in range(B)])
    clear = np.abs(cum - r[:, None]).min(1) > 1e-2
    assert clear.mean() > 0.7
    np.testing.assert_array_equal(isacc.cpu().numpy().astype(bool)[clear], info.is_accepted[clear])
    assert 0.1 < info.is_accepted.mean() < 1.0                       # both branches are exercised
    np.testing.assert_allclose(acc.cpu().numpy()[clear], info.acceptance_rate[clear], atol=2e-2)
    np.testing.assert_allclose(w.cpu().numpy()[clear], info.proposed_weight[clear], atol=2e-2)
    np.testing.assert_allclose(pos.cpu().numpy()[clear], new.position[clear], atol=5e-3 * max(1.0, np.abs(new.position).max()))
    np.testing.assert_allclose(logp.cpu().numpy()[clear], new.logdensity[clear], rtol=1e-3, atol=2e-2)
    np.testing.assert_array_equal(grad.cpu().numpy(), g_before)          # :295 the gradient is NOT refreshed
    ctx.close()
