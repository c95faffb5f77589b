"""--ref_dist widegauss (IndepGaussian(dim, var = 5), exe_flow_matching.py:48-54): the flow's reference distribution enters
the training batch (x0, :155), the independent-MH proposal and its density ratio (:249-255), the conditional importance
sampling weights (:283-289) and the final importance weights (:457).  Against the oracle with the same reference."""
import numpy as np
import pytest

from oracle import flow, fm, mala, prng, targets

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


@pytest.mark.parametrize("family", ["tile", "wide"])
def test_widegauss_loss_and_grad(family):
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=64, B=32, hidden=32, F=16, ref_dist="widegauss")
    params = gu.rand_params(model, seed=3)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, family=_lib.FAMILY_TILE if family == "tile" else _lib.FAMILY_WIDE)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(11)
    lo, go = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma, ref_std=np.sqrt(5.0))
    l1, _ = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma)
    assert abs(lo - l1) > 0.1 * abs(l1)                                   # the reference distribution matters
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(key, _dev(x32), loss, grads)
    assert abs(loss.item() - lo) <= 2e-5 * abs(lo)
    gf = gu.flat_params(go).astype(np.float64)
    assert np.abs(grads.cpu().numpy() - gf).max() < 3e-4 * np.abs(gf).max()
    ctx.close()


@pytest.mark.parametrize("shape", ["generic", "headline"])
def test_widegauss_independent_mh_step(shape):
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    d, hidden, F = (64, 32, 16) if shape == "generic" else (256, 128, 128)
    B = 32
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F, ref_dist="widegauss")
    params = gu.rand_params(model, seed=9, out_scale=0.05)
    params[4]["kernel"] *= 1e-3; params[4]["bias"] *= 1e-3
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    beta = 0.8
    vg = targets.Tempered(dist, beta).value_and_grad
    x32 = dist.init_params.astype(np.float32)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    st = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    key = prng.PRNGKey(31)
    new, info = flow.imh_step(prng.split(key, B), st, vg, model, params, args)
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda"); prop = torch.empty(B, d, device="cuda")
    ctx.flow_step(_lib.FLOW_IMH, key, beta, pos, logp, grad, acc, isacc, prop, None)
    p = prop.cpu().numpy()
    assert np.abs(p - info.proposed_position).max() < 5e-3 * max(1.0, np.abs(p).max())
    assert p.std() > 1.5                                                  # proposals come from N(0, 5 I) pushed through the flow
    with np.errstate(divide="ignore"):
        la_g, la_o = np.log(acc.cpu().numpy().astype(np.float64)), np.log(info.acceptance_rate)
    fin = np.isfinite(la_g) & np.isfinite(la_o)
    if fin.any():
        assert np.abs(la_g[fin] - la_o[fin]).max() < 0.5 + 2e-4 * np.abs(la_o[fin]).max()
    sure = ~fin | (np.abs(la_o) > 1)
    np.testing.assert_array_equal(isacc.cpu().numpy().astype(bool)[sure], info.is_accepted[sure])
    ctx.close()


def test_widegauss_loop_with_importance_sampling_matches_oracle():
    from tests.test_gpu_loop import _run_both
    out, res, ex = _run_both("4-mode", 2, 64, 8, 3, hutch=False, step_size=0.2, ref_dist="widegauss", num_importance_samples=4)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=2e-5)
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=2e-2)
    assert np.isfinite(res).all()
    ex["engine"].close()
