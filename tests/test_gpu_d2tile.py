"""The d = 2 exact-trace solver exists on two tilings (16 chains per workgroup: ode.hip `eval_x2`; 4 chains per workgroup:
ode_d2.hip, picked when the 16-chain tiling would leave CUs idle, e.g. BASELINE configs[0]'s 512 chains; MFM_D2_TILE = 4 is
its resident-weight instance for the default widths, 4s the streamed-weight instance for any widths).  Both are held to the
oracle on a PRESCRIBED step sequence (exe_flow_matching.py:206-242, :264-278; oracle/ode.py, oracle/flow.py) and to each other."""
import numpy as np
import pytest

from oracle import flow, loop, mala, ode, prng, targets

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def _setup(which, B):
    from tests import gpu_util as gu
    if which == "gmm16":
        args, dist, k, model, state = gu.gmm16_setup(B=B, hutchs=False)
    else:
        args, dist, k, model, state = gu.gmm4_setup(B=B, hidden=128, F=128, hutchs=False)
    params = gu.rand_params(model, seed=9, out_scale=0.3)
    return gu, args, dist, model, params


@pytest.mark.parametrize("tile", ["4", "4s", "16"])
@pytest.mark.parametrize("which,mode", [("gmm16", "rwmh"), ("gmm4", "rwmh"), ("gmm4", "imh")])
def test_flow_step_on_prescribed_steps(monkeypatch, tile, which, mode):
    import torch
    from mfm_amd import _lib
    monkeypatch.setenv("MFM_D2_TILE", tile)
    B, d = 32, 2
    gu, args, dist, model, params = _setup(which, B)
    if mode == "imh":
        args.num_importance_samples = -1
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x32 = dist.init_params.astype(np.float32)
    beta = 0.7
    vg = targets.Tempered(dist, beta).value_and_grad
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    st0 = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    key = prng.PRNGKey(31)
    keys = prng.split(key, B)
    step = flow.imh_step if mode == "imh" else flow.rwmh_step
    nat = {}
    step(keys, st0, vg, model, params, args, nat)
    from tests.test_gpu_replay import _replay_arrays
    dt, ac = _replay_arrays([nat["inv"], nat["fwd"]])
    rp = dict(inv=dict(dt=dt[0].astype(np.float64), acc=ac[0]), fwd=dict(dt=dt[1].astype(np.float64), acc=ac[1]))
    so = {}
    new_o, info_o = step(keys, st0, vg, model, params, args, so, replay=rp)
    ratio = torch.zeros(dt.shape, device="cuda"); own = torch.zeros(dt.shape, device="cuda"); diag = torch.zeros(B, 4, dtype=torch.float64, device="cuda")
    ctx.debug_replay(_dev(dt), _dev(ac), ratio, own, diag)
    a = torch.empty(B, device="cuda"); ia = torch.empty(B, dtype=torch.uint8, device="cuda"); pr = torch.empty(B, d, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    pc, lc, gc = pos.clone(), logp.clone(), grad.clone()
    ctx.flow_step(_lib.FLOW_IMH if mode == "imh" else _lib.FLOW_RWMH, key, beta, pc, lc, gc, a, ia, pr, ns)
    np.testing.assert_array_equal(ns.cpu().numpy(), so["n_att_inv"] + so["n_att_fwd"])
    dg = diag.cpu().numpy()
    scale = max(1.0, np.abs(info_o.proposed_position).max())
    e_p = np.abs(pr.cpu().numpy() - info_o.proposed_position).max()
    e_v = max(np.abs(dg[:, 0] - so["vol0"]).max(), np.abs(dg[:, 1] - so["volp"]).max())
    e_a = np.abs(dg[:, 3] - so["log_alpha"]).max()
    print(f"{which}/{mode} tile {tile}: attempts {ns.float().mean().item():.0f}, |dx'| {e_p:.2e}, |dvol| {e_v:.2e}, |d log alpha| {e_a:.2e}")
    assert e_p < 1e-4 * scale and e_v < 1e-3 and e_a < 5e-3
    # the float32 controller saw the same error ratios and would have chosen the same steps (attempt by attempt)
    from tests.test_gpu_replay import _controller_diffs
    for s, st in enumerate([so["inv"], so["fwd"]]):
        rr, dd = _controller_diffs(st, ratio[s].cpu().numpy(), own[s].cpu().numpy(), st["n_attempted"])
        # (d + 1 = 3 components: the error estimate is a cancelling sum of seven stage derivatives, so its float32 relative
        # error is larger than on the wide states of tests/test_gpu_replay.py; measured medians 1.3e-4 / 1.6e-5)
        print(f"   solve {s}: error-ratio median rel. diff {np.median(rr):.1e}, chosen-step {np.median(dd):.1e}")
        assert np.median(rr) < 1e-3 and np.median(dd) < 1e-4, (np.median(rr), np.median(dd))
    sure = np.abs(so["log_alpha"]) > 0.05
    np.testing.assert_array_equal(ia.cpu().numpy().astype(bool)[sure], info_o.is_accepted[sure])
    same = ia.cpu().numpy().astype(bool) == info_o.is_accepted
    np.testing.assert_allclose(pc.cpu().numpy()[same], new_o.position[same], atol=1e-4 * scale)
    np.testing.assert_allclose(lc.cpu().numpy()[same], new_o.logdensity[same], rtol=1e-5, atol=1e-3)
    acc_rows = ia.cpu().numpy().astype(bool) & same
    if acc_rows.any():
        np.testing.assert_allclose(gc.cpu().numpy()[acc_rows], new_o.logdensity_grad[acc_rows], rtol=1e-3, atol=1e-3)
    ctx.close()


@pytest.mark.parametrize("direction", [1, -1])
def test_transform_matches_oracle_and_the_two_tilings_agree(monkeypatch, direction):
    import torch
    B, d = 64, 2
    gu, args, dist, model, params = _setup("gmm4", B)
    rng = np.random.default_rng(3)
    x32 = (4.0 * rng.standard_normal((B, d))).astype(np.float32)
    fn = ode.transform_and_logdet if direction > 0 else ode.inverse_and_logdet
    st = {}
    yo, lo = fn(model, params, None, x32.astype(np.float64), False, args.rtol, args.atol, args.mxstep, n_ts=args.n_ts, stats=st)
    res = {}
    for tile in ("4", "4s", "16"):
        monkeypatch.setenv("MFM_D2_TILE", tile)
        ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
        out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.ode_transform(direction, _dev(x32), out, ldj, key=prng.PRNGKey(1), nsteps=ns)
        res[tile] = (out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy())
        ctx.close()
        # two adaptive solves with their own controllers: the tolerance of tests/test_gpu_ode.py
        assert np.abs(res[tile][0] - yo).max() < 2e-3 * max(1.0, np.abs(yo).max())
        assert np.abs(res[tile][1] - lo).max() < 1e-2 * max(1.0, np.abs(lo).max())
        assert abs(res[tile][2].mean() - st["n_attempted"].mean()) < 0.1 * st["n_attempted"].mean()
    # the tilings differ by float reassociation only, but with their OWN controllers that is enough to flip borderline accept
    # decisions of this ReLU field (each agrees with the oracle's attempt count for 35-50 % of the samples, and as often with
    # each other: tools/dbg/d2_tilings.py); the step-for-step comparison is the prescribed-sequence test above.  Here: the same
    # distance to each other as to the oracle.
    for tile in ("4", "4s"):
        assert np.abs(res[tile][0] - res["16"][0]).max() < 4e-3 * max(1.0, np.abs(yo).max())
        assert abs(res[tile][2].mean() - res["16"][2].mean()) < 0.05 * res["16"][2].mean()


def test_small_tile_is_the_default_for_few_chains_and_inverse_undoes_transform(monkeypatch):
    """512 chains (BASELINE configs[0]) take the 4-chain tiling without any environment override; a round trip through both
    directions returns the input and opposite log-dets."""
    import torch
    monkeypatch.delenv("MFM_D2_TILE", raising=False)
    B, d = 512, 2
    gu, args, dist, model, params = _setup("gmm4", B)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    rng = np.random.default_rng(4)
    x = _dev((3.0 * rng.standard_normal((B, d))).astype(np.float32))
    y = torch.empty_like(x); xb = torch.empty_like(x); l1 = torch.empty(B, device="cuda"); l2 = torch.empty(B, device="cuda")
    ctx.ode_transform(1, x, y, l1, key=prng.PRNGKey(1))
    ctx.ode_transform(-1, y, xb, l2, key=prng.PRNGKey(1))
    assert (xb - x).abs().max().item() < 5e-3 * max(1.0, x.abs().max().item())
    assert (l1 + l2).abs().max().item() < 2e-2 * max(1.0, l1.abs().max().item())
    ctx.close()
