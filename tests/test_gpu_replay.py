"""GPU parity of the Dormand-Prince solver and the flow-MH step ON A PRESCRIBED STEP SEQUENCE, in the regime the benchmark runs.

The natural-controller tests (tests/test_gpu_ode.py) compare a float32 adaptive solve with a float64 one: the two controllers
take slightly different steps, so those tests can only bound the distance between two approximations of the same flow (2e-3
on the outputs, 5 % on the Hutchinson log-det, |d log alpha| < 0.5).  Here both sides integrate with the SAME step sizes and
accept decisions (``mfm_debug_replay`` <-> ``oracle.ode.odeint(replay=...)``): the oracle first runs with its own controller,
its step sequence is rounded to float32 and then replayed by the oracle (float64 arithmetic) and by the HIP kernels (float32
arithmetic).  What is compared is then the arithmetic itself -- the six stage evaluations, the 5th-order update, the error
norm, the 4th-order interpolant at t = 1 and the log-det -- at float32 rounding level, over hundreds of attempted steps of an
UNTAMED network (random, and the one the benchmark has after its warm-up cycle: ~300 attempted steps per chain), for the
generic solver tile, the shape-specialised solver (``solve``) and the flow-step kernel with per-row solve phases and tail
compaction (``solve2``).  The controller's own outputs (error ratio per attempt, the step size it would have chosen) are
compared too, so the float32 controller is pinned to the float64 one attempt by attempt.

Stated tolerances (measured maxima in brackets, MI355X, this seed): see the asserts."""
import numpy as np
import pytest

from oracle import flow, mala, ode, prng, targets

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def _replay_arrays(stats_list, cap=None):
    """[S, B, cap] float32 dt / uint8 acc from the oracle's recorded step sequences (one entry per solve)."""
    A = max(s["acc_seq"].shape[1] for s in stats_list)
    cap = cap or A + 2
    B = stats_list[0]["acc_seq"].shape[0]
    dt = np.zeros((len(stats_list), B, cap), np.float32); acc = np.zeros((len(stats_list), B, cap), np.uint8)
    for s, st in enumerate(stats_list):
        dt[s, :, :st["dt_seq"].shape[1]] = st["dt_seq"].astype(np.float32)
        acc[s, :, :st["acc_seq"].shape[1]] = st["acc_seq"]
    return dt, acc


def _check_controller(tag, st_o, ratio_g, own_g, natt):
    """The float32 controller, attempt by attempt, against the float64 one on the same trajectory."""
    B = len(natt)
    worst_r = worst_d = 0.0
    for b in range(B):
        n = int(natt[b])
        ro, rg = st_o["ratio_seq"][b, :n], ratio_g[b, :n].astype(np.float64)
        worst_r = max(worst_r, np.abs(rg - ro).max() / max(1.0, ro.max()) if n else 0.0)
        # the ratio is a cancellation-dominated quantity (5th - 4th order): float32 reproduces it to ~1e-3 of its scale
        np.testing.assert_allclose(rg, ro, rtol=2e-2, atol=2e-2, err_msg=f"{tag}: error ratio, chain {b}")
        do, dg = st_o["dt_own"][b, :n + 1], own_g[b, :n + 1].astype(np.float64)
        worst_d = max(worst_d, (np.abs(dg - do) / do).max())
        np.testing.assert_allclose(dg, do, rtol=1e-2, err_msg=f"{tag}: step size the controller chose, chain {b}")
    return worst_r, worst_d


@pytest.mark.parametrize("d,hidden,F", [(256, 128, 128), (128, 128, 128), (64, 32, 16)])   # shape-specialised solver x2, generic tile
@pytest.mark.parametrize("direction", [1, -1])
def test_transform_on_prescribed_steps_matches_oracle(d, hidden, F, direction):
    import torch
    from tests import gpu_util as gu
    B = 32
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F)
    params = gu.rand_params(model, seed=9, out_scale=0.5)            # UNTAMED gate: gate * clip(grad log pi), hundreds of steps
    if d <= 128:                                                     # no clip below dim 128 (:351): |grad log pi| ~ 1e3 would make
        params[4]["kernel"] *= 2e-2; params[4]["bias"] *= 2e-2       # the random field violently stiff; keep ~100 steps
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x64 = dist.init_params.astype(np.float32).astype(np.float64)
    keys = prng.split(prng.PRNGKey(21), B)
    fn = ode.transform_and_logdet if direction > 0 else ode.inverse_and_logdet
    o = (True, args.rtol, args.atol, args.mxstep)
    st = {}
    fn(model, params, keys, x64, *o, stats=st)                       # the oracle's own controller: records the step sequence
    dt, acc = _replay_arrays([st])
    st_o = {}
    y_o, l_o = fn(model, params, keys, x64, *o, stats=st_o, replay=dict(dt=dt[0].astype(np.float64), acc=acc[0]))
    np.testing.assert_array_equal(st_o["n_attempted"], st["n_attempted"])
    assert st["n_attempted"].mean() > 40, st["n_attempted"].mean()    # a non-trivial integration
    ratio = torch.zeros(dt[0].shape, device="cuda"); own = torch.zeros(dt[0].shape, device="cuda")
    ctx.debug_replay(_dev(dt[0]), _dev(acc[0]), ratio, own)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.ode_transform(direction, _dev(x64.astype(np.float32)), out, ldj, keys=_dev(keys.astype(np.uint32).view(np.int32)), nsteps=ns)
    y, l, n = out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy()
    np.testing.assert_array_equal(n, st["n_attempted"])              # same step sequence => same attempt count, exactly
    assert np.abs(y - x64).max() > 1e-2                              # the flow moves the points
    ey, el = np.abs(y - y_o).max(), np.abs(l - l_o).max()
    # float32 arithmetic over ~100..400 steps: outputs to ~1e-5, log-det (a sum of O(1e2) z.Jz terms of size O(10)) to ~1e-3 rel.
    assert ey < 1e-4 * max(1.0, np.abs(y_o).max()), ey
    assert el < 2e-3 * max(1.0, np.abs(l_o).max()), (el, np.abs(l_o).max())
    assert abs((l - l_o).mean()) < 5e-4 * max(1.0, np.abs(l_o).max())       # no systematic log-det bias
    wr, wd = _check_controller(f"d={d} dir={direction}", st_o, ratio.cpu().numpy(), own.cpu().numpy(), n)
    print(f"replay transform d={d} dir={direction}: attempts {n.mean():.0f}, |dy| {ey:.2e}, |dl| {el:.2e} (|l| {np.abs(l_o).max():.1f}), "
          f"ratio {wr:.2e}, dt_own {wd:.2e}")
    # a replay call is one-shot: the next transform integrates with its own controller again
    ctx.ode_transform(direction, _dev(x64.astype(np.float32)), out, ldj, keys=_dev(keys.astype(np.uint32).view(np.int32)), nsteps=ns)
    assert np.abs(out.cpu().numpy() - y_o).max() < 2e-3 * max(1.0, np.abs(y_o).max())
    ctx.close()


def _flow_replay(ctx, model, params, args, dist, beta, x32, key, label):
    """One flow-MH step of the kernel and of the oracle on the oracle's (float32-rounded) step sequences; returns the
    differences.  Exercises fast::solve2 (per-row solve phases, tail compaction) for the headline shape."""
    import torch
    from mfm_amd import _lib
    B, d = x32.shape
    vg = targets.Tempered(dist, beta).value_and_grad
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    st0 = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    keys = prng.split(key, B)
    nat = {}
    flow.rwmh_step(keys, st0, vg, model, params, args, nat)                          # natural run: records both step sequences
    dt, acc = _replay_arrays([nat["inv"], nat["fwd"]])
    rp = dict(inv=dict(dt=dt[0].astype(np.float64), acc=acc[0]), fwd=dict(dt=dt[1].astype(np.float64), acc=acc[1]))
    so = {}
    new_o, info_o = flow.rwmh_step(keys, st0, vg, model, params, args, so, replay=rp)
    ratio = torch.zeros(dt.shape, device="cuda"); own = torch.zeros(dt.shape, device="cuda")
    diag = torch.zeros(B, 4, dtype=torch.float64, device="cuda")
    ctx.debug_replay(_dev(dt), _dev(acc), ratio, own, diag)
    a = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda")
    prop = torch.empty(B, d, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, key, beta, pos, logp, grad, a, isacc, prop, ns)
    n_o = so["n_att_inv"] + so["n_att_fwd"]
    np.testing.assert_array_equal(ns.cpu().numpy(), n_o)
    dg = diag.cpu().numpy()
    res = dict(
        n=n_o, prop=np.abs(prop.cpu().numpy() - info_o.proposed_position).max(),
        vol0=np.abs(dg[:, 0] - so["vol0"]).max(), volp=np.abs(dg[:, 1] - so["volp"]).max(),
        vol_bias=abs((dg[:, 0] - so["vol0"]).mean()) + abs((dg[:, 1] - so["volp"]).mean()),
        vol_scale=max(1.0, np.abs(so["vol0"]).max(), np.abs(so["volp"]).max()),
        la=np.abs(dg[:, 3] - so["log_alpha"]).max(), la_o=so["log_alpha"], la_g=dg[:, 3],
        isacc=isacc.cpu().numpy().astype(bool), isacc_o=info_o.is_accepted, new_o=new_o,
        pos=pos.cpu().numpy(), logp=logp.cpu().numpy())
    rg, og = ratio.cpu().numpy(), own.cpu().numpy()
    res["ctl_inv"] = _check_controller(label + " inverse", so["inv"], rg[0], og[0], so["n_att_inv"])
    res["ctl_fwd"] = _check_controller(label + " forward", so["fwd"], rg[1], og[1], so["n_att_fwd"])
    return res


def _assert_flow(res, label):
    print(f"replay flow step {label}: attempts {res['n'].mean():.0f} (max {res['n'].max()}), |dx'| {res['prop']:.2e}, |dvol0| {res['vol0']:.2e}, "
          f"|dvolp| {res['volp']:.2e} (scale {res['vol_scale']:.1f}), |d log alpha| {res['la']:.2e}, controller {res['ctl_inv']} {res['ctl_fwd']}")
    assert res["prop"] < 1e-4, res["prop"]                                    # proposal x' = T(T^-1(x) + noise): two solves
    assert res["vol0"] < 2e-3 * res["vol_scale"] and res["volp"] < 2e-3 * res["vol_scale"]
    assert res["vol_bias"] < 1e-3 * res["vol_scale"]                          # no systematic log-det bias
    # log alpha = logp(x') - volp - logp(x) - vol0 (:271-274): logp(x') inherits |grad log pi| |dx'| ~ 1e3 * 1e-5 * sqrt(d)
    assert res["la"] < 0.1, res["la"]
    sure = np.abs(res["la_o"]) > 0.5                                          # decisions whose uniform is not within the noise
    np.testing.assert_array_equal(res["isacc"][sure], res["isacc_o"][sure])


@pytest.mark.parametrize("d,hidden,F", [(256, 128, 128), (64, 32, 16)])
def test_flow_step_on_prescribed_steps_matches_oracle_random_network(d, hidden, F):
    from tests import gpu_util as gu
    B = 32
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F)
    params = gu.rand_params(model, seed=9, out_scale=0.3)
    if d <= 128:
        params[4]["kernel"] *= 2e-2; params[4]["bias"] *= 2e-2
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    res = _flow_replay(ctx, model, params, args, dist, 0.8, dist.init_params.astype(np.float32), prng.PRNGKey(31), f"random d={d}")
    _assert_flow(res, f"random network d={d}")
    ctx.close()


def test_flow_step_in_the_benchmarked_regime_matches_oracle(trained_phi4):
    """The network and chain states the benchmark has when its timed region starts (phi-four d = 256, 4096 chains, one full
    cycle of 101 iterations from the flax-style init: bench.py's warm-up), 32 of its chains: (a) on the oracle's step
    sequence, the flow-step kernel against ``oracle.flow.rwmh_step`` -- outputs, log-dets, log acceptance ratio, decisions,
    attempt counts (exact); (b) with each side's own controller -- attempt counts and outputs as far as two adaptive solves
    of the same flow agree."""
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    tp = trained_phi4
    B, d = 32, 256
    args, dist, model = tp["args32"], tp["dist"], tp["model"]
    params = gu.unflat_params(model, tp["params_flat"])
    ctx = gu.make_ctx(dist, args, n_local=B, n_total=B, fourier=model.f, params=params)
    x32 = tp["pos"][:B]
    res = _flow_replay(ctx, model, params, args, dist, 1.0, x32, prng.PRNGKey(77), "trained d=256")
    assert res["n"].mean() > 150, res["n"].mean()                             # the benchmarked regime: hundreds of attempts
    _assert_flow(res, "trained network d=256 (benchmark state)")
    # (b) natural controllers
    vg = targets.Tempered(dist, 1.0).value_and_grad
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    st0 = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    key = prng.PRNGKey(77)
    so = {}
    new_o, info_o = flow.rwmh_step(prng.split(key, B), st0, vg, model, params, args, so)
    a = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda")
    prop = torch.empty(B, d, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    diag = torch.zeros(B, 4, dtype=torch.float64, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, key, 1.0, pos, logp, grad, a, isacc, prop, ns)
    n_g, n_o = ns.cpu().numpy(), so["n_att_inv"] + so["n_att_fwd"]
    ep = np.abs(prop.cpu().numpy() - info_o.proposed_position)
    print(f"natural flow step (trained): attempts gpu {n_g.mean():.1f} oracle {n_o.mean():.1f}, equal for {(n_g == n_o).mean():.0%}, "
          f"|dx'| max {ep.max():.2e} mean {ep.mean():.2e}")
    assert abs(n_g.mean() - n_o.mean()) < 0.03 * n_o.mean()
    assert np.abs(n_g - n_o).max() <= 0.1 * n_o.max()
    assert ep.max() < 2e-3 and ep.mean() < 1e-4
    ctx.close()
