"""The hidden non-linearities of the reference's table (exe_flow_matching.py:39-45; --non_linearity, multi_modal.py:177):
relu (default), tanh, elu, gelu (jax's tanh approximation) and swish on both kernel families.  The backward pass of the fused
tile family takes f' from the stored OUTPUT where the activation is invertible (relu, tanh, elu) and from a packed global
workspace of f'(pre-activation) written by the forward epilogues otherwise (gelu, swish); the wide family keeps pre-activations.  Loss / parameter gradient, vector field / JVP, CNF transform and a short
loop against the oracle with the same activation."""
import numpy as np
import pytest

from oracle import fm, ode, prng

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def _relerr(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


CASES = [("tanh", "tile"), ("elu", "tile"), ("gelu", "tile"), ("swish", "tile"), ("tanh", "wide"), ("elu", "wide"), ("gelu", "wide"), ("swish", "wide"),
         ("gelu", "auto"), ("relu", "wide")]


def _ctx(act, fam, kind="phi4", d=64, B=32, hidden=32, F=16, tame=None):
    from mfm_amd import _lib
    from tests import gpu_util as gu
    setup = gu.phi4_setup if kind == "phi4" else gu.lgcp_setup
    kw = dict(d=d) if kind == "phi4" else dict(n=int(np.sqrt(d)))
    args, dist, k, model, state = setup(B=B, hidden=hidden, F=F, non_linearity=act, **kw)
    params = gu.rand_params(model, seed=3)
    if tame:
        params[4]["kernel"] *= tame; params[4]["bias"] *= tame
    family = {"tile": _lib.FAMILY_TILE, "wide": _lib.FAMILY_WIDE, "auto": None}[fam]
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, family=family)
    if fam == "auto":
        assert act in ("gelu", "swish")
    return args, dist, model, params, ctx


@pytest.mark.parametrize("act,fam", CASES)
@pytest.mark.parametrize("kind,d", [("phi4", 64), ("lgcp", 64)])
def test_loss_grad_field_and_jvp_match_oracle(act, fam, kind, d):
    import torch
    from tests import gpu_util as gu
    B = 32
    args, dist, model, params, ctx = _ctx(act, fam, kind, d, B)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(11)
    loss_o, grads_o = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(key, _dev(x32), loss, grads)
    assert abs(loss.item() - loss_o) <= 2e-5 * abs(loss_o), (loss.item(), loss_o)
    g = gu.unflat_params(model, grads.cpu().numpy())
    for i, (gg, go) in enumerate(zip(g, grads_o)):
        for kk in ("kernel", "bias"):
            assert _relerr(gg[kk], go[kk].astype(np.float64)) < 3e-4, (i, kk, _relerr(gg[kk], go[kk]))
    rng = np.random.default_rng(1)
    t = rng.uniform(0, 1, B).astype(np.float32); z = rng.standard_normal((B, d)).astype(np.float32)
    v_o, jv_o = model.forward(params, x32.astype(np.float64), t.astype(np.float64), tangent=z.astype(np.float64))
    v = torch.empty(B, d, device="cuda"); jv = torch.empty(B, d, device="cuda")
    ctx.vf_apply(_dev(x32), _dev(t), v, _dev(z), jv)
    assert _relerr(v.cpu().numpy(), v_o) < 3e-5
    assert _relerr(jv.cpu().numpy(), jv_o) < 3e-5
    ctx.close()


@pytest.mark.parametrize("act,fam", [("tanh", "tile"), ("elu", "tile"), ("swish", "auto"), ("gelu", "auto")])
def test_cnf_transform_matches_oracle(act, fam):
    import torch
    B, d = 32, 64
    args, dist, model, params, ctx = _ctx(act, fam, "phi4", d, B, tame=1e-3)
    x32 = dist.init_params.astype(np.float32)
    keys = prng.split(prng.PRNGKey(21), B)
    st = {}
    y_o, l_o = ode.transform_and_logdet(model, params, keys, x32.astype(np.float64), True, args.rtol, args.atol, args.mxstep, stats=st)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.ode_transform(1, _dev(x32), out, ldj, keys=_dev(keys.astype(np.uint32).view(np.int32)), nsteps=ns)
    y, l, n = out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy()
    assert np.abs(y - x32).max() > 1e-2
    assert np.abs(y - y_o).max() < 2e-3 * max(1.0, np.abs(y_o).max())
    assert np.abs(l - l_o).max() < 5e-2 * max(1.0, np.abs(l_o).max())
    assert abs(n.mean() - st["n_attempted"].mean()) < 0.1 * st["n_attempted"].mean()
    back = torch.empty(B, d, device="cuda"); l2 = torch.empty(B, device="cuda")
    ctx.ode_transform(-1, out, back, l2, keys=_dev(keys.astype(np.uint32).view(np.int32)))
    assert np.abs(back.cpu().numpy() - x32).max() < 2e-3
    ctx.close()


@pytest.mark.parametrize("act", ["tanh", "swish"])
def test_loop_with_non_default_activation_matches_oracle(act):
    from tests.test_gpu_loop import _run_both
    out, res, ex = _run_both("phi-four", 64, 64, 8, 3, step_size=1e-4, non_linearity=act)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=1e-5)
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=5e-3)
    np.testing.assert_allclose(ex["betas"], tr["beta"], rtol=2e-3)
    assert np.isfinite(res[0])
    ex["engine"].close()


@pytest.mark.parametrize("act", ["gelu", "swish"])
def test_smooth_activations_with_a_mixture_target_and_the_exact_trace(act):
    """exe_flow_matching.py:39-45 is target-agnostic: gelu / swish with the d = 2 mixtures and trace(jacfwd(v)) (no --hutch) --
    declined until round 3 (those activations ran on the wide family only, which serves neither).  Loss / gradient, one
    exact-trace flow-MH step on a prescribed step sequence and a short loop against the oracle."""
    import torch
    from mfm_amd import _lib
    from oracle import flow, mala, targets
    from tests import gpu_util as gu
    from tests.test_gpu_replay import _replay_arrays
    B, d = 32, 2
    args, dist, k, model, state = gu.gmm4_setup(B=B, hidden=32, F=16, hutchs=False, non_linearity=act)
    params = gu.rand_params(model, seed=4, out_scale=0.3)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(3)
    loss_o, grads_o = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(key, _dev(x32), loss, grads)
    assert abs(loss.item() - loss_o) <= 2e-5 * abs(loss_o)
    assert _relerr(grads.cpu().numpy(), gu.flat_params(grads_o)) < 3e-4
    beta = 0.7
    vg = targets.Tempered(dist, beta).value_and_grad
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    st0 = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    kf = prng.PRNGKey(31)
    nat = {}
    flow.rwmh_step(prng.split(kf, B), st0, vg, model, params, args, nat)
    dt, ac = _replay_arrays([nat["inv"], nat["fwd"]])
    rp = dict(inv=dict(dt=dt[0].astype(np.float64), acc=ac[0]), fwd=dict(dt=dt[1].astype(np.float64), acc=ac[1]))
    so = {}
    new_o, info_o = flow.rwmh_step(prng.split(kf, B), st0, vg, model, params, args, so, replay=rp)
    ratio = torch.zeros(dt.shape, device="cuda"); own = torch.zeros(dt.shape, device="cuda"); diag = torch.zeros(B, 4, dtype=torch.float64, device="cuda")
    ctx.debug_replay(_dev(dt), _dev(ac), ratio, own, diag)
    pr = torch.empty(B, d, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, kf, beta, pos, logp, grad, None, None, pr, ns)
    np.testing.assert_array_equal(ns.cpu().numpy(), so["n_att_inv"] + so["n_att_fwd"])
    dg = diag.cpu().numpy()
    assert np.abs(pr.cpu().numpy() - info_o.proposed_position).max() < 1e-4 * max(1.0, np.abs(info_o.proposed_position).max())
    assert max(np.abs(dg[:, 0] - so["vol0"]).max(), np.abs(dg[:, 1] - so["volp"]).max()) < 1e-3 and np.abs(dg[:, 3] - so["log_alpha"]).max() < 5e-3
    ctx.close()
