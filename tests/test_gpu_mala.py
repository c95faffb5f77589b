"""GPU parity: fused MALA step / init / loglik / beta bisection against the float64 oracle (through the C ABI)."""
import numpy as np
import pytest

from oracle import flow, mala, prng, targets

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


@pytest.mark.parametrize("setup,d", [("phi4", 256), ("phi4", 64), ("phi4", 40), ("gmm", 2), ("lgcp", 64), ("lgcp", 256)])
def test_mala_init_and_step_match_oracle(setup, d):
    import torch
    from tests import gpu_util as gu
    B = 64
    if setup == "phi4":
        args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=32, F=16)
        eps, beta = 1e-4, 0.37
    elif setup == "lgcp":
        args, dist, k, model, state = gu.lgcp_setup(n=int(np.sqrt(d)), B=B)
        eps, beta = 0.01, 0.45
    else:
        args, dist, k, model, state = gu.gmm4_setup(B=B)
        eps, beta = 0.2, 0.6
    ctx = gu.make_ctx(dist, args)
    x32 = dist.init_params.astype(np.float32)
    vg = targets.Tempered(dist, beta).value_and_grad
    st = mala.init(x32.astype(np.float64), vg)
    pos, logp, grad = _dev(x32), torch.empty(B, dtype=torch.float64, device="cuda"), torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    np.testing.assert_allclose(logp.cpu().numpy(), st.logdensity, rtol=2e-6, atol=1e-3)
    np.testing.assert_allclose(grad.cpu().numpy(), st.logdensity_grad, rtol=2e-5, atol=2e-3)
    ll = torch.empty(B, dtype=torch.float64, device="cuda")
    ctx.loglik(pos, ll)
    np.testing.assert_allclose(ll.cpu().numpy(), dist.loglik(x32.astype(np.float64)), rtol=2e-6, atol=1e-3)
    # one step on the same (float32-rounded) state, same key
    key = prng.PRNGKey(77)
    keys = prng.split(key, B)
    st_in = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    new, info, u = mala.kernel(keys, st_in, vg, eps)
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda")
    prop = torch.empty(B, d, device="cuda"); w = torch.empty(B, device="cuda")
    ctx.mala_step(key, beta, eps, pos, logp, grad, acc, isacc, prop, w)
    np.testing.assert_allclose(prop.cpu().numpy(), info.proposed_position, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(acc.cpu().numpy(), info.acceptance_rate, rtol=5e-3, atol=5e-3)
    decided = np.abs(u - info.acceptance_rate) > 1e-2          # decisions may only differ on a knife edge
    np.testing.assert_array_equal(isacc.cpu().numpy()[decided].astype(bool), info.is_accepted[decided])
    same = isacc.cpu().numpy().astype(bool) == info.is_accepted
    np.testing.assert_allclose(pos.cpu().numpy()[same], new.position[same], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(logp.cpu().numpy()[same], new.logdensity[same], rtol=2e-6, atol=2e-3)
    np.testing.assert_allclose(grad.cpu().numpy()[same], new.logdensity_grad[same], rtol=3e-5, atol=3e-3)
    ctx.close()


def test_mala_sharded_equals_unsharded():
    """Chains [16, 48) of 64 on a 'second rank' draw exactly what the single-process run draws."""
    import torch
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=64, B=64, hidden=32, F=16)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(5)
    outs = []
    for (n_local, off) in [(64, 0), (32, 16)]:
        ctx = gu.make_ctx(dist, args, n_local=n_local, n_total=64, offset=off)
        pos = _dev(x32[off:off + n_local])
        logp = torch.empty(n_local, dtype=torch.float64, device="cuda"); grad = torch.empty(n_local, 64, device="cuda")
        ctx.mala_init(pos, 1.0, logp, grad)
        for _ in range(3):
            ctx.mala_step(key, 1.0, 1e-4, pos, logp, grad)
        outs.append(pos.cpu().numpy())
        ctx.close()
    np.testing.assert_array_equal(outs[0][16:48], outs[1])


def test_mala_many_steps_statistics():
    """200 steps on phi-four d=64: GPU float32 chain vs oracle float64 chain, same keys -> same moments."""
    import torch
    from tests import gpu_util as gu
    B, d, eps = 256, 64, 1e-4
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=32, F=16)
    ctx = gu.make_ctx(dist, args)
    vg = targets.Tempered(dist, 1.0).value_and_grad
    x32 = dist.init_params.astype(np.float32)
    st = mala.init(x32.astype(np.float64), vg)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    key = prng.PRNGKey(3)
    acc = torch.empty(B, device="cuda")
    accs = []
    for it in range(200):
        key, sub = prng.split(key, 2)
        st, info, _ = mala.kernel(prng.split(sub, B), st, vg, eps)
        ctx.mala_step(sub, 1.0, eps, pos, logp, grad, acc)
        accs.append((acc.mean().item(), info.acceptance_rate.mean()))
    g, o = pos.cpu().numpy().astype(np.float64), st.position
    assert abs(np.mean(accs, 0)[0] - np.mean(accs, 0)[1]) < 2e-3
    np.testing.assert_allclose(g.mean(), o.mean(), atol=2e-3)
    np.testing.assert_allclose((g ** 2).mean(), (o ** 2).mean(), rtol=2e-3)
    np.testing.assert_allclose(logp.cpu().numpy().mean(), st.logdensity.mean(), rtol=1e-3)
    # with identical noise the trajectories stay close except for chains whose accept decision flipped
    close = np.abs(g - o).max(1) < 1e-3
    assert close.mean() > 0.9
    ctx.close()


def test_beta_bisection_matches_oracle():
    import torch
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=64, B=512, hidden=32, F=16)
    ctx = gu.make_ctx(dist, args)
    ll = dist.loglik(dist.init_params)
    b_or = flow.beta_fn(0.0, ll, 0.95, 512)
    b_gpu = ctx.beta_update(0.0, _dev(ll), 0.95)
    assert abs(b_gpu - b_or) <= 1e-9 * max(1.0, abs(b_or)) + 1e-12
    b2 = ctx.beta_update(b_or, _dev(ll), 0.95)
    assert abs(b2 - flow.beta_fn(b_or, ll, 0.95, 512)) < 1e-9
    flat = -1.0 + 1e-3 * np.random.default_rng(0).standard_normal(512)
    assert abs(ctx.beta_update(0.3, _dev(flat), 0.95) - flow.beta_fn(0.3, flat, 0.95, 512)) < 1e-12
    ctx.close()


def test_noise_prefetch_is_bit_identical_to_inline_draws():
    """mfm_noise_prefetch (noise.hip): the draws of coming MALA steps / training batches, produced in the tail of the flow-step
    kernel by the workgroups whose tile is done, are the ones the kernels make in line -- chain states, losses and gradients
    are bit-identical with and without it."""
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    B, d = 512, 256
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
    params = gu.rand_params(model, seed=4, out_scale=0.3)
    params[4]["kernel"] *= 1e-3; params[4]["bias"] *= 1e-3
    x32 = dist.init_params.astype(np.float32)
    keys = np.stack([prng.split(prng.PRNGKey(100 + i), 2) for i in range(6)]).astype(np.uint32)      # [6, 2, 2]: (gn, step) per iteration
    res = []
    for prefetch, one_call in ((False, False), (True, False), (True, True), (False, True)):      # one_call: mfm_train_iter, its MALA step inside the training kernel
        ctx = gu.make_ctx(dist, args, n_local=B // 2, n_total=B, offset=B // 2, fourier=model.f, params=params)
        n = B // 2
        pos = torch.as_tensor(x32[n:]).cuda(); logp = torch.empty(n, dtype=torch.float64, device="cuda"); grad = torch.empty(n, d, device="cuda")
        ctx.mala_init(pos, 1.0, logp, grad)
        acc = torch.empty(n, device="cuda")
        if prefetch:
            assert ctx.noise_prefetch(keys[1:5, 0], keys[1:5, 1])        # iterations 1..4; iteration 5 draws in line
        ctx.flow_step(_lib.FLOW_RWMH, keys[0, 0], 1.0, pos, logp, grad, acc)
        loss = torch.zeros(1, dtype=torch.float64, device="cuda"); g = torch.zeros(ctx.n_params, device="cuda")
        tr = []
        for i in range(1, 6):
            if one_call:
                ctx.train_iter(i, 100, _lib.FLOW_RWMH, keys[i, 0], keys[i, 1], 1.0, 1e-4, pos, logp, grad, loss, g, acc=acc, apply_update=False)
            else:
                ctx.mala_step(keys[i, 0], 1.0, 1e-4, pos, logp, grad, acc)
                ctx.fm_loss_grad(keys[i, 1], pos, loss, g)
            tr.append((loss.item(), g.clone(), acc.clone()))
        res.append((pos.clone(), logp.clone(), grad.clone(), tr))
        ctx.close()
    for r in res[1:]:
        assert torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) and torch.equal(res[0][2], r[2])
        for (l0, g0, a0), (l1, g1, a1) in zip(res[0][3], r[3]):
            assert l0 == l1 and torch.equal(g0, g1) and torch.equal(a0, a1)


def test_acc_stats_kernel():
    import torch
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=64, B=32, hidden=32, F=16)
    ctx = gu.make_ctx(dist, args)
    x = torch.rand(4096, device="cuda") * 3
    out = torch.zeros(2, dtype=torch.float64, device="cuda")
    ctx.acc_stats(x, out)
    xd = x.double()
    assert abs(out[0].item() - xd.sum().item()) < 1e-9 * xd.sum().item()
    assert abs(out[1].item() - (xd * xd).sum().item()) < 1e-9 * (xd * xd).sum().item()
    x[7] = float("nan")
    ctx.acc_stats(x, out)
    assert torch.isnan(out).all()                       # exact-sample training logs NaN acceptance (exe_flow_matching.py:385)
    ctx.close()
