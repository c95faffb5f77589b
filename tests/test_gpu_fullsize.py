"""Size-independent properties at BASELINE.json's FULL sizes (the oracle finishes only small cases in seconds):
phi-four d = 256 with 4096 chains (configs[2], one GPU's share of configs[3]), the 409,600-sample eval batch of the
gaussian-mixture example (configs[1]) and the pines per-GPU shape (configs[4]: 1024 chains, 32 x 32 grid, hidden 1024).
Round trips, shard invariance (bit-exact for the chain state), linearity of the summed loss / gradient over shards, and
agreement between kernels that evaluate the same quantity by different routes."""
import numpy as np
import pytest

from oracle import fm, prng

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def _tamed(model, seed=9, out_scale=0.3, gate=1e-3):
    from tests import gpu_util as gu
    p = gu.rand_params(model, seed=seed, out_scale=out_scale)
    p[4]["kernel"] *= gate; p[4]["bias"] *= gate
    return p


def test_phi4_4096_chains_mala_and_loss_are_shard_invariant():
    import torch
    from tests import gpu_util as gu
    B, d = 4096, 256
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
    params = _tamed(model)
    x32 = dist.init_params.astype(np.float32)
    key, kfm = prng.PRNGKey(5), prng.PRNGKey(6)
    res = {}
    for name, (n_local, off) in {"full": (B, 0), "lo": (B // 2, 0), "hi": (B // 2, B // 2)}.items():
        ctx = gu.make_ctx(dist, args, n_local=n_local, n_total=B, offset=off, fourier=model.f, params=params)
        pos = _dev(x32[off:off + n_local]); logp = torch.empty(n_local, dtype=torch.float64, device="cuda"); grad = torch.empty(n_local, d, device="cuda")
        ctx.mala_init(pos, 1.0, logp, grad)
        for it in range(3):
            ctx.mala_step(prng.split(key, 3)[it], 1.0, args.step_size, pos, logp, grad)
        loss = torch.zeros(1, dtype=torch.float64, device="cuda"); g = torch.zeros(ctx.n_params, device="cuda")
        ctx.fm_loss_grad(kfm, pos, loss, g)
        res[name] = (pos.cpu().numpy(), logp.cpu().numpy(), loss.item(), g.cpu().numpy().astype(np.float64))
        ctx.close()
    # chain trajectories do not depend on the sharding: bit-exact
    np.testing.assert_array_equal(np.concatenate([res["lo"][0], res["hi"][0]]), res["full"][0])
    np.testing.assert_array_equal(np.concatenate([res["lo"][1], res["hi"][1]]), res["full"][1])
    # the loss and its gradient are SUMS over chains (exe_flow_matching.py:178): shards add up
    assert abs(res["lo"][2] + res["hi"][2] - res["full"][2]) < 1e-9 * abs(res["full"][2])
    gs = res["lo"][3] + res["hi"][3]
    assert np.abs(gs - res["full"][3]).max() < 2e-5 * np.abs(res["full"][3]).max()


def test_phi4_4096_chains_zero_init_loss_is_target_norm():
    import torch
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=256, B=4096)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=state.params)          # flax-style init: v == 0
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(2)
    t, cond, target = fm.cond_flow_batch(key, x32.astype(np.float64), args.sigma)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(key, _dev(x32), loss, grads)
    assert abs(loss.item() - (target ** 2).sum()) < 1e-5 * (target ** 2).sum()
    ctx.close()


@pytest.mark.parametrize("B,d", [(4096, 256), (1024, 64)])      # BASELINE configs[2]; the reference's own phi-four shape (multi_modal.py:52-55:
def test_phi4_4096_chains_flow_round_trip_and_step_consistency(B, d):      # the shape-specialised kernels zero-padded to their 128-wide tile)
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
    params = _tamed(model, out_scale=0.5)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x32 = dist.init_params.astype(np.float32)
    x = _dev(x32)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.ode_transform(1, x, out, ldj, key=prng.PRNGKey(4), nsteps=ns)
    assert (out - x).abs().max().item() > 1e-2 and ns.min().item() >= 1
    back = torch.empty(B, d, device="cuda"); l2 = torch.empty(B, device="cuda")
    ctx.ode_transform(-1, out, back, l2, key=prng.PRNGKey(4))
    assert (back - x).abs().max().item() < 2e-3                                  # inverse(transform(x)) == x
    assert (l2 + ldj).abs().max().item() < 0.15 * max(1.0, ldj.abs().max().item())
    # the same solve twice is bit-identical (no atomics, no order-dependent reductions)
    out2 = torch.empty_like(out); ldj2 = torch.empty_like(ldj)
    ctx.ode_transform(1, x, out2, ldj2, key=prng.PRNGKey(4))
    assert torch.equal(out, out2) and torch.equal(ldj, ldj2)
    # one flow-MH step: accepted chains sit at their proposal with the target re-evaluated there, rejected ones are untouched
    beta = 0.9
    pos = x.clone(); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    pos0, logp0, grad0 = pos.clone(), logp.clone(), grad.clone()
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda"); prop = torch.empty(B, d, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(31), beta, pos, logp, grad, acc, isacc, prop, ns)
    a = isacc.bool()
    assert 0 < a.sum().item() < B
    assert torch.equal(pos[a], prop[a]) and torch.equal(pos[~a], pos0[~a])
    assert torch.equal(logp[~a], logp0[~a]) and torch.equal(grad[~a], grad0[~a])
    lp2 = torch.empty(B, dtype=torch.float64, device="cuda"); g2 = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, lp2, g2)                                            # the MALA kernel's evaluation of the same target
    assert (lp2 - logp).abs().max().item() < 1e-6 * logp.abs().max().item()
    assert (g2 - grad).abs().max().item() < 1e-4 * max(1.0, grad.abs().max().item())
    ctx.close()


def test_phi4_default_shape_flow_step_agrees_across_chains_per_workgroup(monkeypatch):
    """The reference's phi-four shape (d = 64, 1024 chains): the launch gives every workgroup 4 chains (ode.hip: flow_live_rows).  The same
    flow-MH step with 16, 8 and 4 chains per workgroup: two ADAPTIVE float32 solves of the same chain in layouts that round differently --
    proposals within the solver tolerance of each other, the same decisions but for borderline chains, the same attempt statistics."""
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    B, d = 1024, 64
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
    params = _tamed(model, out_scale=0.5)
    x32 = dist.init_params.astype(np.float32)
    res = {}
    for live in ("16", "8", None):
        monkeypatch.setenv("MFM_FLOW_LIVE", live) if live else monkeypatch.delenv("MFM_FLOW_LIVE", raising=False)
        ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
        pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
        ctx.mala_init(pos, 0.9, logp, grad)
        acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda"); prop = torch.empty(B, d, device="cuda")
        ns = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(31), 0.9, pos, logp, grad, acc, isacc, prop, ns)
        res[live] = (prop.cpu().numpy(), isacc.cpu().numpy().astype(bool), ns.cpu().numpy(), acc.cpu().numpy())
        ctx.close()
    p16, a16, n16, _ = res["16"]
    for live in ("8", None):
        p, a, n, _ = res[live]
        dp = np.abs(p - p16).max(1)
        print(f"chains per workgroup {live or 'auto'} vs 16: |dx'| median {np.median(dp):.2e} max {dp.max():.2e}; decisions equal {np.mean(a == a16):.4f}; attempts {n.mean():.1f} vs {n16.mean():.1f}")
        assert np.median(dp) < 2e-4 and dp.max() < 2e-2, (np.median(dp), dp.max())
        assert np.mean(a == a16) > 0.98
        assert abs(n.mean() - n16.mean()) < 0.02 * n16.mean()
    assert 0 < a16.sum() < B


def test_gaussian_mixture_eval_batch_is_a_sum_of_its_chunks():
    """configs[1]: eval_step on eval_iter * num_chain = 100 * 4096 exact samples (exe_flow_matching.py:370-374)."""
    import torch
    from tests import gpu_util as gu
    B, n_eval = 4096, 409600
    args, dist, k, model, state = gu.gmm4_setup(B=B, hidden=128, F=128)
    params = gu.rand_params(model, seed=3)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, max_eval=n_eval)
    rng = np.random.default_rng(0)
    xs = _dev((8.0 * rng.choice([-1.0, 1.0], size=(n_eval, 2)) + rng.standard_normal((n_eval, 2))).astype(np.float32))
    key = prng.PRNGKey(12)
    full = torch.zeros(1, dtype=torch.float64, device="cuda")
    ctx.fm_loss(key, xs, full, n_total=n_eval, offset=0)
    part = torch.zeros(1, dtype=torch.float64, device="cuda"); tot = 0.0
    for c in range(0, n_eval, 16 * B):
        ctx.fm_loss(key, xs[c:c + 16 * B], part, n_total=n_eval, offset=c)
        tot += part.item()
    assert np.isfinite(full.item()) and abs(tot - full.item()) < 1e-9 * abs(full.item())
    ctx.close()


def test_pines_1024_chains_round_trip_and_shard_sum():
    """configs[4] per-GPU shape on the wide kernel family."""
    import torch
    from tests import gpu_util as gu
    B, d = 1024, 1024
    args, dist, k, model, state = gu.lgcp_setup(n=32, B=B, hidden=1024, F=128)
    params = _tamed(model, seed=2, out_scale=0.2, gate=0.02)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(8)
    res = {}
    for name, (n_local, off) in {"full": (B, 0), "lo": (B // 2, 0), "hi": (B // 2, B // 2)}.items():
        ctx = gu.make_ctx(dist, args, n_local=n_local, n_total=B, offset=off, fourier=model.f, params=params)
        loss = torch.zeros(1, dtype=torch.float64, device="cuda"); g = torch.zeros(ctx.n_params, device="cuda")
        ctx.fm_loss_grad(key, _dev(x32[off:off + n_local]), loss, g)
        res[name] = (loss.item(), g.cpu().numpy().astype(np.float64))
        if name == "full":
            x = _dev(x32)
            out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
            ctx.ode_transform(1, x, out, ldj, key=prng.PRNGKey(4), nsteps=ns)
            back = torch.empty(B, d, device="cuda"); l2 = torch.empty(B, device="cuda")
            ctx.ode_transform(-1, out, back, l2, key=prng.PRNGKey(4))
            assert (out - x).abs().max().item() > 1e-2
            assert (back - x).abs().max().item() < 2e-3 * max(1.0, x.abs().max().item())
        ctx.close()
    assert abs(res["lo"][0] + res["hi"][0] - res["full"][0]) < 1e-9 * abs(res["full"][0])
    assert np.abs(res["lo"][1] + res["hi"][1] - res["full"][1]).max() < 2e-5 * np.abs(res["full"][1]).max()


def test_static_training_kernel_equals_the_runtime_shape_kernel(monkeypatch):
    """The headline configuration runs a shape-static instance of the flow-matching kernel (widths, activation and target kind
    as compile-time constants); MFM_GENERIC_FM routes it through the runtime-shape instance every other configuration uses:
    same arithmetic, so loss, gradient and the eval loss agree to float32 contraction noise -- and both sit on the oracle
    at a size it still finishes (64 of the 4096 chains)."""
    import torch
    from tests import gpu_util as gu
    B, d = 4096, 256
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
    params = _tamed(model)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(17)
    out = {}
    for name in ("static", "generic"):
        if name == "generic":
            monkeypatch.setenv("MFM_GENERIC_FM", "1")
        ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
        loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
        ev = torch.zeros(1, dtype=torch.float64, device="cuda")
        ctx.fm_loss_grad(key, _dev(x32), loss, grads)
        ctx.fm_loss(key, _dev(x32), ev)
        out[name] = (loss.item(), grads.cpu().numpy(), ev.item())
        ctx.close()
    # same arithmetic; hipcc contracts multiply-adds per instance, so the last bits may differ (measured: 2e-10 on the loss)
    for i in (0, 2):
        assert abs(out["static"][i] - out["generic"][i]) <= 1e-8 * abs(out["generic"][i])
    assert np.abs(out["static"][1] - out["generic"][1]).max() <= 1e-5 * np.abs(out["generic"][1]).max()
    assert abs(out["static"][0] - out["static"][2]) <= 1e-8 * abs(out["static"][0])       # same key: eval loss = training loss
    # oracle on the first 64 chains (draws are indexed by global chain id out of 4096)
    ctx = gu.make_ctx(dist, args, n_local=64, n_total=B, offset=0, fourier=model.f, params=params)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(key, _dev(x32[:64]), loss, grads)
    lo, go = fm.loss_and_grad(model, params, key, x32[:64].astype(np.float64), args.sigma, n_total=B, start=0)
    assert abs(loss.item() - lo) < 2e-5 * abs(lo)
    gflat = gu.flat_params(go)
    assert np.abs(grads.cpu().numpy() - gflat).max() < 2e-4 * np.abs(gflat).max()
    ctx.close()


_NATURAL = {}


def test_headline_flow_step_all_4096_chains_against_libmfm_ref(trained_phi4):
    """The WHOLE benchmarked launch against the oracle, live: phi-four d = 256, all 4096 chains and the network bench.py's timed region starts
    from (one trained cycle), one flow-MH step with each side's own controllers -- the float32 kernel through the C ABI, and libmfm_ref
    (oracle/cref: the float64 C / OpenMP restatement, itself held to the numpy oracle in tests/test_oracle_cref.py; ~10 s on 16 threads
    where the numpy oracle needs minutes) on the same chains, keys and parameters.  Two adaptive solves of a clipped, ill-conditioned
    flow do not agree chain by chain (tests/test_gpu_replay.py compares them stage by stage on prescribed steps, 32 chains); over 4096
    chains their STATISTICS must: attempted steps, latent positions, log-determinants, log acceptance ratios, decisions."""
    import torch
    from mfm_amd import _lib
    from oracle import cref, mala, targets
    from tests import gpu_util as gu
    tp = trained_phi4
    B, d = 4096, 256
    dist, model = tp["dist"], tp["model"]
    params = gu.unflat_params(model, tp["params_flat"])
    args = tp["args32"]
    x32 = tp["pos"]
    assert x32.shape == (B, d)
    key = prng.PRNGKey(4242)
    # ---- the kernel ----
    ctx = gu.make_ctx(dist, args, n_local=B, n_total=B, fourier=model.f, params=params)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    a = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda")
    prop = torch.empty(B, d, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, key, 1.0, pos, logp, grad, a, isacc, prop, ns)
    n_g, prop_g, acc_g = ns.cpu().numpy().astype(np.int64), prop.cpu().numpy().astype(np.float64), isacc.cpu().numpy().astype(bool)
    with np.errstate(divide="ignore"):
        la_g = np.log(a.cpu().numpy().astype(np.float64))
    ctx.close()
    # ---- the oracle ----
    cr = cref.CRef(model, params)
    vg = targets.Tempered(dist, 1.0).value_and_grad
    st0 = mala.init(x32.astype(np.float64), vg)
    so = {}
    st1, info = cr.rwmh_step(prng.split(key, B), st0, args, stats=so, record=1001)
    _NATURAL["run"] = (key, st0, so)                  # (the prescribed-step test below replays this run's step sequences)
    n_o = so["n_att_inv"] + so["n_att_fwd"]
    la_o = so["log_alpha"]
    qs = [0.1, 0.5, 0.9, 0.99]
    print(f"full-size flow step: attempts gpu {n_g.mean():.2f} oracle {n_o.mean():.2f} (quantiles {np.quantile(n_g, qs)} vs {np.quantile(n_o, qs)}), "
          f"equal for {(n_g == n_o).mean():.1%}, |dn| median {np.median(np.abs(n_g - n_o)):.0f}")
    # the benchmarked regime: hundreds of attempted steps per chain
    assert n_o.mean() > 250
    # attempted steps: the means within 1 %, the distribution's quantiles within 2 %, a chain's own count within a few per cent
    assert abs(n_g.mean() - n_o.mean()) < 0.01 * n_o.mean()
    assert np.abs(np.quantile(n_g, qs) / np.quantile(n_o, qs) - 1.0).max() < 0.02
    assert np.median(np.abs(n_g - n_o)) <= 0.05 * n_o.mean()
    # proposals: the median chain to the solver's accuracy amplified by the flow, none further than the flow moves a chain
    ep = np.abs(prop_g - info.proposed_position).max(1)
    move = np.abs(info.proposed_position - x32).max(1)
    print(f"   |dx'| median {np.median(ep):.2e} p90 {np.quantile(ep, 0.9):.2e} max {ep.max():.2e}; the flow moves a chain by {np.median(move):.2f} (median)")
    assert np.median(ep) < 2e-2 and np.quantile(ep, 0.9) < 0.2 and ep.max() < move.max()
    # moments of the proposals over the 4096 chains (what the downstream statistics see)
    assert np.abs(prop_g.mean(0) - info.proposed_position.mean(0)).max() < 5e-3
    assert np.abs((prop_g ** 2).mean(0) - (info.proposed_position ** 2).mean(0)).max() < 1e-2
    # log acceptance ratios: O(-1e3) with the network of one trained cycle (the Hutchinson log-det): the kernel reports the ratio itself
    # in float32, which underflows to exactly 0 below exp(-104) -- where the oracle's log ratio is far below that the kernel's ratio
    # must be 0, where it is representable the logarithms must agree to what the clip kinks leave (tests/test_gpu_replay.py)
    print(f"   oracle log alpha: median {np.median(la_o):.1f}, max {la_o.max():.1f}; kernel ratios that are exactly 0: {(~np.isfinite(la_g)).mean():.1%}")
    deep = la_o < -150.0
    assert np.isneginf(la_g[deep]).all()
    shallow = la_o > -60.0      # a handful of 4096 (measured: 7, log ratios -37 ... 889, six of them accepted on both sides)
    fin = shallow & np.isfinite(la_g)
    print("   chains whose oracle log alpha is above -60:", np.round(la_o[shallow], 1), "kernel:", np.round(la_g[shallow], 1), "decisions", info.is_accepted[shallow], acc_g[shallow])
    # chain by chain the two adaptive solves of this clipped flow differ by O(10 - 150) in the log-determinants, either sign
    # (tools/dbg/la_natural.py: term by term, on the kernel's own step sequence); float32 overflows above exp(88.7)
    assert (np.abs(la_g[fin] - la_o[fin]) < 250.0).all() and (la_o[shallow & np.isposinf(la_g)] > 88.7 - 250.0).all()
    # decisions: nearly every proposal is rejected in this regime, on both sides, and the same few are accepted
    assert (acc_g != info.is_accepted).sum() <= 2


def test_headline_mala_steps_loss_and_gradient_all_4096_chains_against_libmfm_ref(trained_phi4):
    """The other half of a benchmarked iteration at its full size, live against the oracle: three MALA steps of all 4096 chains, then the
    flow-matching loss and its 214,272-element gradient on their positions, with the trained network of bench.py's timed region -- the
    kernels through the C ABI against libmfm_ref (oracle/cref) fed the same keys."""
    import torch
    from oracle import cref, mala, targets
    from oracle.vfield import flat_params
    from tests import gpu_util as gu
    tp = trained_phi4
    B, d = 4096, 256
    dist, model, args = tp["dist"], tp["model"], tp["args32"]
    params = gu.unflat_params(model, tp["params_flat"])
    x32 = tp["pos"]
    ctx = gu.make_ctx(dist, args, n_local=B, n_total=B, fourier=model.f, params=params)
    cr = cref.CRef(model, params)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    st = mala.init(x32.astype(np.float64), targets.Tempered(dist, 1.0).value_and_grad)
    assert np.abs(logp.cpu().numpy() - st.logdensity).max() < 1e-7 * np.abs(st.logdensity).max()
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda")
    flips, moved, ok = 0, 0, np.ones(B, bool)
    for it in range(6):        # three steps of the rule as written (far from equilibrium it rejects: min(1, 1 / alpha)), three of the textbook rule
        textbook = it >= 3
        k = prng.split(prng.PRNGKey(99), 6)[it]
        ctx.mala_step(k, 1.0, args.step_size, pos, logp, grad, acc, isacc, textbook=textbook)
        st, info = cr.mala_kernel(prng.split(k, B), st, args.step_size, textbook=textbook)
        same = isacc.cpu().numpy().astype(bool) == info.is_accepted
        flips += int((~same).sum()); moved += int(info.is_accepted.sum())
        ok &= same             # a chain whose decision differs (u within rounding of p) has left the oracle's trajectory: compare the others
        pg, po = acc.cpu().numpy().astype(np.float64)[ok], info.acceptance_rate[ok]
        big = po > 1e-30
        relp = np.abs(pg[big] - po[big]) / po[big] if big.any() else np.zeros(1)
        print(f"   step {it} ({'textbook' if textbook else 'as written'}): acceptance mean {po.mean():.3e}, accepted {int(info.is_accepted.sum())}, "
              f"representable in float32: {int(big.sum())}, p rel. error median {np.median(relp):.1e} max {relp.max():.1e}")
        assert np.median(relp) < 1e-3 and np.quantile(relp, 0.99) < 5e-2
        assert (pg[~big] < 1e-29).all()
    assert moved > 3 * B // 2 and flips <= 8 and ok.mean() > 0.998, (moved, flips, ok.mean())
    e = np.abs(pos.cpu().numpy().astype(np.float64) - st.position)[ok]
    print(f"full-size MALA: six steps, {moved} accepted moves, decisions flipped {flips}, |dx| max {e.max():.2e}")
    assert e.max() < 2e-6 * max(1.0, np.abs(st.position).max())
    # ---- loss and gradient on the oracle's positions (float32-rounded: what the kernel is given) ----
    xk = st.position.astype(np.float32)
    key = prng.PRNGKey(123)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); g = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(key, _dev(xk), loss, g)
    lo, go = cr.fm_loss_grad(*fm.cond_flow_batch(key, xk.astype(np.float64), args.sigma))
    gof = flat_params(go).astype(np.float64)
    gg = g.cpu().numpy().astype(np.float64)
    rel_l = abs(loss.item() - lo) / abs(lo)
    rel_g = np.linalg.norm(gg - gof) / np.linalg.norm(gof)
    print(f"full-size loss {loss.item():.6e} vs {lo:.6e} (rel {rel_l:.1e}); gradient rel. L2 {rel_g:.1e}, max |d| / max |g| {np.abs(gg - gof).max() / np.abs(gof).max():.1e}")
    assert rel_l < 2e-5 and rel_g < 3e-4 and np.abs(gg - gof).max() < 5e-4 * np.abs(gof).max()
    ctx.close()


def test_headline_flow_step_all_4096_chains_on_prescribed_steps_against_libmfm_ref(trained_phi4):
    """The benchmarked launch on PRESCRIBED steps, all 4096 chains: libmfm_ref integrates the flow-MH step with its own controllers and records
    every chain's step sequence (hundreds of attempted steps per solve); the kernel (mfm_debug_replay) and the oracle then both take that
    sequence, rounded to float32.  What tests/test_gpu_replay.py checks on 32 chains, at full size: attempted-step counts of all 8192
    solves EXACT; the typical chain tight; the spread of the clipped flow's log-determinants as that test documents it."""
    import torch
    from mfm_amd import _lib
    from oracle import cref, mala, targets
    from tests import gpu_util as gu
    tp = trained_phi4
    B, d = 4096, 256
    dist, model, args = tp["dist"], tp["model"], tp["args32"]
    params = gu.unflat_params(model, tp["params_flat"])
    x32 = tp["pos"]
    ctx = gu.make_ctx(dist, args, n_local=B, n_total=B, fourier=model.f, params=params)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    cr = cref.CRef(model, params)
    if "run" in _NATURAL:                              # the natural run of the test above (same state, same key): its recorded sequences
        key, st0, nat = _NATURAL["run"]
    else:
        key = prng.PRNGKey(4242)
        st0 = mala.init(x32.astype(np.float64), targets.Tempered(dist, 1.0).value_and_grad)
        nat = {}
        cr.rwmh_step(prng.split(key, B), st0, args, stats=nat, record=1001)
    keys = prng.split(key, B)
    amax = int(max(nat["n_att_inv"].max(), nat["n_att_fwd"].max()))
    cap = amax + 2
    dt = np.zeros((2, B, cap), np.float32); acc = np.zeros((2, B, cap), np.uint8)
    for s_, k_ in enumerate(("inv", "fwd")):
        dt[s_] = nat[k_]["dt_seq"][:, :cap].astype(np.float32); acc[s_] = nat[k_]["acc_seq"][:, :cap]
    rp = dict(inv=dict(dt=dt[0].astype(np.float64), acc=acc[0]), fwd=dict(dt=dt[1].astype(np.float64), acc=acc[1]))
    so = {}
    st1, info = cr.rwmh_step(keys, st0, args, stats=so, replay=rp)
    d_dt, d_acc = _dev(dt), _dev(acc)
    ratio = torch.zeros(dt.shape, device="cuda"); own = torch.zeros(dt.shape, device="cuda")
    diag = torch.zeros(B, 4, dtype=torch.float64, device="cuda")
    ctx.debug_replay(d_dt, d_acc, ratio, own, diag)
    a = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda")
    prop = torch.empty(B, d, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, key, 1.0, pos, logp, grad, a, isacc, prop, ns)
    n_g, n_o = ns.cpu().numpy().astype(np.int64), so["n_att_inv"] + so["n_att_fwd"]
    dg = diag.cpu().numpy()
    np.testing.assert_array_equal(n_g, n_o)                                       # every chain, both solves
    assert n_o.mean() > 250
    ep = np.abs(prop.cpu().numpy().astype(np.float64) - info.proposed_position).max(1)
    e0, e1 = np.abs(dg[:, 0] - so["vol0"]), np.abs(dg[:, 1] - so["volp"])
    sc = max(1.0, np.abs(so["volp"]).max())
    q = lambda v: np.quantile(v, [0.5, 0.9, 0.99, 1.0])
    print(f"full-size replay: attempts {n_o.mean():.1f} (max {n_o.max()}), all equal; |dx'| quantiles 50/90/99/100 % {q(ep)}, |d vol0| {q(e0)}, |d volp| {q(e1)} (scale {sc:.0f})")
    # the typical chain is tight (tests/test_gpu_replay.py, 32 chains: |dx'| 2.8e-5, inverse log-det 2.5e-4 of ~1e3); a chain whose solves cross a
    # clip kink between two stage evaluations differently moves by O(100) in the log-determinant (the float64 oracle does so itself when its
    # stage inputs are rounded to float32): bounded at the 90 % level and by the scale of the log-determinants themselves
    assert np.median(ep) < 3e-4 and np.median(e0) < 5e-3 * sc and np.median(e1) < 5e-3 * sc
    assert np.quantile(ep, 0.9) < 0.1 and np.quantile(e0, 0.9) < 0.05 * sc and np.quantile(e1, 0.9) < 0.05 * sc
    assert ep.max() < np.abs(info.proposed_position - x32).max() and max(e0.max(), e1.max()) < sc
    # decisions: the deeply negative log ratios reject on both sides; where they differ the kernel's ratio is within the log-determinant spread
    mism = isacc.cpu().numpy().astype(bool) != info.is_accepted
    assert mism.sum() <= 3, mism.sum()
    ctx.close()
