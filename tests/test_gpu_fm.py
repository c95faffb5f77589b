"""GPU parity: flow-matching loss + parameter gradient, AdamW chain, eval loss, vector field / JVP -- vs the oracle."""
import numpy as np
import pytest

from oracle import fm, optim, prng

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def _relerr(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("setup,d,B,hidden,F", [("phi4", 256, 64, 128, 128), ("phi4", 64, 32, 32, 16), ("phi4", 40, 16, 48, 10),
                                               ("gmm", 2, 64, 32, 16), ("lgcp", 64, 32, 32, 16)])
def test_fm_loss_and_grad_match_oracle(setup, d, B, hidden, F):
    import torch
    from tests import gpu_util as gu
    if setup == "phi4":
        args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F)
    elif setup == "lgcp":
        args, dist, k, model, state = gu.lgcp_setup(n=int(np.sqrt(d)), B=B, hidden=hidden, F=F)
    else:
        args, dist, k, model, state = gu.gmm4_setup(B=B, hidden=hidden, F=F)
    params = gu.rand_params(model, seed=3)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    np.testing.assert_array_equal(ctx.get_params(), gu.flat_params(params))
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(11)
    loss_o, grads_o = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(key, _dev(x32), loss, grads)
    assert abs(loss.item() - loss_o) <= 2e-5 * abs(loss_o), (loss.item(), loss_o)
    g = gu.unflat_params(model, grads.cpu().numpy())
    for i, (gg, go) in enumerate(zip(g, grads_o)):
        for kk in ("kernel", "bias"):
            assert _relerr(gg[kk], go[kk].astype(np.float64)) < 2e-4, (i, kk, _relerr(gg[kk], go[kk]))
    # eval-only loss (eval_step, exe_flow_matching.py:370-374) on the same samples
    l2 = torch.zeros(1, dtype=torch.float64, device="cuda")
    ctx.fm_loss(key, _dev(x32), l2)
    assert abs(l2.item() - loss_o) <= 2e-5 * abs(loss_o)
    ctx.close()


@pytest.mark.parametrize("family", ["tile", "wide"])
@pytest.mark.parametrize("n", [8, 56, 150])
def test_eval_loss_on_a_sample_count_that_is_not_a_multiple_of_16(family, n):
    """mfm_fm_loss (eval_step, exe_flow_matching.py:370-374) takes any n: the last n % 16 samples go through a staged 16-row
    tile with residual 0 on its padding rows; draws are indexed by the global sample index, so a shard [start, start + n) of
    n_total equals the oracle on the same shard."""
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.gmm4_setup(B=160, hidden=32, F=16)
    params = gu.rand_params(model, seed=5)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, max_eval=160, family=_lib.FAMILY_WIDE if family == "wide" else None)
    xs = dist.sample_model_rows(prng.split(prng.PRNGKey(3), 160)).astype(np.float32)
    key = prng.PRNGKey(12)
    for start, n_total in ((0, n), (7, 160)):
        if start + n > n_total:
            continue
        lo, _ = fm.loss_and_grad(model, params, key, xs[start:start + n].astype(np.float64), args.sigma, n_total=n_total, start=start, need_grad=False)
        l = torch.zeros(1, dtype=torch.float64, device="cuda")
        ctx.fm_loss(key, _dev(xs[start:start + n]), l, n_total=n_total, offset=start)
        assert abs(l.item() - lo) <= 2e-5 * abs(lo), (start, n, l.item(), lo)
    assert ctx.counters()["fm_eval_samples"] == n * (1 + (7 + n <= 160))
    ctx.close()


def test_fm_zero_init_loss_is_target_norm():
    import torch
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=256, B=32)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=state.params)      # flax-style init: zero output kernels
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(2)
    t, cond, target = fm.cond_flow_batch(key, x32.astype(np.float64), args.sigma)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(key, _dev(x32), loss, grads)
    assert abs(loss.item() - (target ** 2).sum()) < 1e-5 * (target ** 2).sum()
    ctx.close()


def test_fm_sharded_loss_and_grads_sum_to_unsharded():
    import torch
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=64, B=64, hidden=32, F=16)
    params = gu.rand_params(model, seed=4)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(8)
    tot_l, tot_g = 0.0, 0.0
    for (n_local, off) in [(64, 0), (32, 0), (32, 32)]:
        ctx = gu.make_ctx(dist, args, n_local=n_local, n_total=64, offset=off, fourier=model.f, params=params)
        loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
        ctx.fm_loss_grad(key, _dev(x32[off:off + n_local]), loss, grads)
        if n_local == 64:
            full_l, full_g = loss.item(), grads.cpu().numpy().astype(np.float64)
        else:
            tot_l += loss.item(); tot_g = tot_g + grads.cpu().numpy().astype(np.float64)
        ctx.close()
    assert abs(tot_l - full_l) < 1e-9 * abs(full_l)
    assert _relerr(tot_g, full_g) < 1e-5


def test_adamw_chain_matches_oracle():
    import torch
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=64, B=32, hidden=32, F=16, learning_iter=9)
    params = gu.rand_params(model, seed=5)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    st = optim.TrainState(params, optim.learning_rate_fn(9, 0, args.learning_rate))
    rng = np.random.default_rng(0)
    for it in range(6):
        g = [{kk: (rng.standard_normal(v.shape) * 3).astype(np.float32) for kk, v in p.items()} for p in params]
        if it == 2:
            g[1]["kernel"][0, 0] = np.inf                               # rejected update: state untouched
        st.apply_gradients(g)
        ctx.adamw_step(_dev(gu.flat_params(g)))
        s = ctx.opt_state()
        assert (s["step"], s["count"], s["notfinite_count"]) == (st.step, st.count, st.notfinite_count)
        np.testing.assert_allclose(ctx.get_params(), gu.flat_params(st.params), rtol=3e-6, atol=1e-7)
    # the packed copies the kernels read must follow the master parameters: loss after the updates matches the oracle
    x32 = dist.init_params.astype(np.float32)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda")
    ctx.fm_loss(prng.PRNGKey(1), _dev(x32), loss)
    lo, _ = fm.loss_and_grad(model, st.params, prng.PRNGKey(1), x32.astype(np.float64), args.sigma, need_grad=False)
    assert abs(loss.item() - lo) < 3e-5 * abs(lo)
    ctx.close()


@pytest.mark.parametrize("setup,d,hidden,F", [("phi4", 256, 128, 128), ("phi4", 40, 32, 10), ("gmm", 2, 32, 16), ("lgcp", 64, 32, 16)])
def test_vector_field_and_jvp_match_oracle(setup, d, hidden, F):
    import torch
    from tests import gpu_util as gu
    B = 32
    if setup == "phi4":
        args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F)
    elif setup == "lgcp":
        args, dist, k, model, state = gu.lgcp_setup(n=int(np.sqrt(d)), B=B, hidden=hidden, F=F)
    else:
        args, dist, k, model, state = gu.gmm4_setup(B=B, hidden=hidden, F=F)
    params = gu.rand_params(model, seed=6)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    rng = np.random.default_rng(1)
    x = dist.init_params.astype(np.float32); t = rng.uniform(0, 1, B).astype(np.float32)
    z = rng.standard_normal((B, d)).astype(np.float32)
    v_o, jv_o = model.forward(params, x.astype(np.float64), t.astype(np.float64), tangent=z.astype(np.float64))
    v = torch.empty(B, d, device="cuda"); jv = torch.empty(B, d, device="cuda")
    ctx.vf_apply(_dev(x), _dev(t), v, _dev(z), jv)
    assert _relerr(v.cpu().numpy(), v_o) < 2e-5
    assert _relerr(jv.cpu().numpy(), jv_o) < 2e-5
    v2 = torch.empty(B, d, device="cuda")
    ctx.vf_apply(_dev(x), _dev(t), v2)
    assert _relerr(v2.cpu().numpy(), v_o) < 2e-5
    ctx.close()


@pytest.mark.parametrize("family", ["tile", "wide"])
def test_finite_check_in_the_reduction_matches_separate_check(family):
    """Single rank: mfm_adamw_step(grads) right after mfm_fm_loss_grad(..., grads) takes the apply_if_finite decision
    (exe_flow_matching.py:135-137) from the flag raised in the gradient reduction; a gradient handed over in another
    buffer goes through the separate check kernel.  Same parameters and counters either way, including a rejected
    non-finite step and the recovery after it."""
    import torch
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=64, B=64, hidden=32, F=16, learning_iter=12)
    params = gu.rand_params(model, seed=6)
    x32 = dist.init_params.astype(np.float32)
    bad = x32.copy(); bad[3, 5] = np.inf
    from mfm_amd import _lib
    fam = _lib.FAMILY_WIDE if family == "wide" else _lib.FAMILY_TILE       # wide: the check rides in its weight-gradient kernel
    ctxs = [gu.make_ctx(dist, args, fourier=model.f, params=params, family=fam) for _ in range(2)]
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctxs[0].n_params, device="cuda")
    for it in range(6):
        pos = _dev(bad if it in (2, 3) else x32)
        key = prng.PRNGKey(20 + it)
        ctxs[0].fm_loss_grad(key, pos, loss, grads); ctxs[0].adamw_step(grads)                 # decision in the update kernel
        ctxs[1].fm_loss_grad(key, pos, loss, grads); ctxs[1].adamw_step(grads.clone())         # separate check kernel
        s0, s1 = ctxs[0].opt_state(), ctxs[1].opt_state()
        assert s0 == s1, (it, s0, s1)
        assert s0["step"] == it + 1 and s0["notfinite_count"] == (it - 1 if it in (2, 3) else 0)
        np.testing.assert_array_equal(ctxs[0].get_params(), ctxs[1].get_params())
    assert s0["count"] == 4
    # two gradient evaluations before one update: the flag belongs to the LAST one (a stale non-finite flag must not survive)
    ctxs[0].fm_loss_grad(prng.PRNGKey(40), _dev(bad), loss, grads)
    ctxs[0].fm_loss_grad(prng.PRNGKey(41), _dev(x32), loss, grads); ctxs[0].adamw_step(grads)
    assert ctxs[0].opt_state()["count"] == 5 and ctxs[0].opt_state()["notfinite_count"] == 0
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("kind", ["phi4-256", "gmm", "pines-wide"])
def test_unconditional_flow_matching_batch_matches_oracle(kind):
    """cond_flow = False (exe_flow_matching.py:139-147: x_t = t x1 + (1 - (1 - sigma) t) x0, target x1 - (1 - sigma) x0, two-way key
    split): not reachable from the reference's CLI (multi_modal.py:162-163 fix cond_flow = True) but selectable through
    `create_train_state`'s args; loss and gradient of every kernel family against the oracle's `flow_batch`."""
    import torch
    from tests import gpu_util as gu
    B = 32
    if kind == "phi4-256":
        args, dist, k, model, state = gu.phi4_setup(d=256, B=B, cond_flow=False)
    elif kind == "gmm":
        args, dist, k, model, state = gu.gmm4_setup(B=B, hidden=32, F=16, cond_flow=False)
    else:
        from mfm_amd import _lib
        args, dist, k, model, state = gu.lgcp_setup(n=8, B=B, hidden=48, F=16, cond_flow=False)
    params = gu.rand_params(model, seed=7)
    fam = {}
    if kind == "pines-wide":
        fam = dict(family=_lib.FAMILY_WIDE)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, **fam)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(12)
    lo, go = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma, cond_flow=False)
    lc, _ = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma, cond_flow=True, need_grad=False)
    assert abs(lo - lc) > 1e-3 * abs(lo)                                  # the two batches differ: the flag is not ignored
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(key, _dev(x32), loss, grads)
    assert abs(loss.item() - lo) < 2e-5 * abs(lo), (loss.item(), lo)
    assert _relerr(grads.cpu().numpy(), gu.flat_params(go)) < 3e-4
    ev = torch.zeros(1, dtype=torch.float64, device="cuda")
    ctx.fm_loss(key, _dev(x32), ev)
    assert abs(ev.item() - lo) < 2e-5 * abs(lo)
    ctx.close()
