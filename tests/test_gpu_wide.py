"""GPU parity of the WIDE kernel family (mfm_amd/csrc/wide.hip: per-layer MFMA GEMMs on HBM-resident activations; what
serves the "pines" widths of the reference, multi_modal.py:89-96) against the float64 oracle -- the same cases and
tolerances as the fused 16-chain family (tests/test_gpu_fm.py, test_gpu_ode.py), forced onto small shapes with
kernel_family = WIDE, plus the real pines shape (32 x 32 grid, hidden width 1024)."""
import numpy as np
import pytest

from oracle import flow, fm, mala, ode, prng, targets

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def _relerr(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _setup(kind, d, B, hidden, F, **kw):
    from tests import gpu_util as gu
    if kind == "phi4":
        return gu.phi4_setup(d=d, B=B, hidden=hidden, F=F, **kw)
    return gu.lgcp_setup(n=int(np.sqrt(d)), B=B, hidden=hidden, F=F, **kw)


def _wide_ctx(dist, args, model, params, **kw):
    from mfm_amd import _lib
    from tests import gpu_util as gu
    return gu.make_ctx(dist, args, fourier=model.f, params=params, family=_lib.FAMILY_WIDE, **kw)


# d = 40 / hidden 48 / F = 10: nothing is a multiple of 64 (tile guards); 1024 / 1024 is BASELINE configs[4]'s network
@pytest.mark.parametrize("kind,d,B,hidden,F", [("phi4", 256, 64, 128, 128), ("phi4", 40, 16, 48, 10), ("lgcp", 64, 32, 32, 16),
                                              ("lgcp", 1024, 32, 1024, 128)])
def test_wide_fm_loss_and_grad_match_oracle(kind, d, B, hidden, F):
    import torch
    from tests import gpu_util as gu
    args, dist, k, model, state = _setup(kind, d, B, hidden, F)
    params = gu.rand_params(model, seed=3)
    ctx = _wide_ctx(dist, args, model, params)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(11)
    loss_o, grads_o = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.full((ctx.n_params,), float("nan"), device="cuda")
    ctx.fm_loss_grad(key, _dev(x32), loss, grads)
    assert abs(loss.item() - loss_o) <= 2e-5 * abs(loss_o), (loss.item(), loss_o)
    g = gu.unflat_params(model, grads.cpu().numpy())
    for i, (gg, go) in enumerate(zip(g, grads_o)):
        for kk in ("kernel", "bias"):
            assert np.isfinite(gg[kk]).all()                      # every element of the gradient vector is written
            assert _relerr(gg[kk], go[kk].astype(np.float64)) < 2e-4, (i, kk, _relerr(gg[kk], go[kk]))
    l2 = torch.zeros(1, dtype=torch.float64, device="cuda")
    ctx.fm_loss(key, _dev(x32), l2)
    assert abs(l2.item() - loss_o) <= 2e-5 * abs(loss_o)
    # optimizer step on the wide family's gradient: same AdamW kernels, parameters stay finite and move
    p0 = ctx.get_params(); ctx.adamw_step(grads); p1 = ctx.get_params()
    assert np.isfinite(p1).all() and np.abs(p1 - p0).max() > 0
    ctx.close()


def test_wide_matches_tile_family():
    """The two kernel families are two schedules of the same arithmetic: same loss / gradient / field to float32 rounding."""
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=256, B=64)
    params = gu.rand_params(model, seed=5)
    x32 = dist.init_params.astype(np.float32)
    res = []
    for fam in (_lib.FAMILY_TILE, _lib.FAMILY_WIDE):
        ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, family=fam)
        loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
        ctx.fm_loss_grad(prng.PRNGKey(7), _dev(x32), loss, grads)
        res.append((loss.item(), grads.cpu().numpy()))
        ctx.close()
    assert abs(res[0][0] - res[1][0]) < 1e-6 * abs(res[0][0])
    assert np.abs(res[0][1] - res[1][1]).max() < 2e-5 * np.abs(res[0][1]).max()


@pytest.mark.parametrize("kind,d,hidden,F", [("phi4", 256, 128, 128), ("phi4", 40, 32, 10), ("lgcp", 64, 32, 16), ("lgcp", 1024, 1024, 128)])
def test_wide_vector_field_and_jvp_match_oracle(kind, d, hidden, F):
    import torch
    from tests import gpu_util as gu
    B = 32
    args, dist, k, model, state = _setup(kind, d, B, hidden, F)
    params = gu.rand_params(model, seed=6)
    ctx = _wide_ctx(dist, args, model, params)
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


def _tamed(model, seed, out_scale, gate_scale):
    from tests import gpu_util as gu
    params = gu.rand_params(model, seed=seed, out_scale=out_scale)
    params[4]["kernel"] *= gate_scale; params[4]["bias"] *= gate_scale
    return params


@pytest.mark.parametrize("kind,d,hidden,F,gs", [("phi4", 64, 32, 16, 1e-3), ("lgcp", 64, 32, 16, 0.05)])
def test_wide_ode_transform_and_inverse_match_oracle(kind, d, hidden, F, gs):
    import torch
    B = 32
    args, dist, k, model, state = _setup(kind, d, B, hidden, F)
    params = _tamed(model, 9, 0.5 if kind == "phi4" else 0.3, gs)
    ctx = _wide_ctx(dist, args, model, params)
    x32 = dist.init_params.astype(np.float32)
    keys = prng.split(prng.PRNGKey(21), B)
    for direction, fn in ((1, ode.transform_and_logdet), (-1, ode.inverse_and_logdet)):
        st = {}
        y_o, l_o = fn(model, params, keys, x32.astype(np.float64), True, args.rtol, args.atol, args.mxstep, stats=st)
        out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda")
        ns = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.ode_transform(direction, _dev(x32), out, ldj, keys=_dev(keys.astype(np.uint32).view(np.int32)), nsteps=ns)
        y, l, n = out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy()
        assert np.abs(y - x32).max() > 1e-2
        assert np.abs(y - y_o).max() < 2e-3 * max(1.0, np.abs(y_o).max()), np.abs(y - y_o).max()
        assert np.abs(y - y_o).mean() < 1e-4 * max(1.0, np.abs(y_o).max())
        assert np.abs(l - l_o).max() < 5e-2 * max(1.0, np.abs(l_o).max())
        dn = np.abs(n - st["n_attempted"])
        assert (dn == 0).mean() >= 0.5 and abs(n.mean() - st["n_attempted"].mean()) < 0.1 * st["n_attempted"].mean(), (n, st["n_attempted"])
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda")
    ctx.ode_transform(1, _dev(x32), out, ldj, key=prng.PRNGKey(4))
    back = torch.empty(B, d, device="cuda"); l2 = torch.empty(B, device="cuda")
    ctx.ode_transform(-1, out, back, l2, key=prng.PRNGKey(4))             # inverse(transform(x)) == x
    assert np.abs(back.cpu().numpy() - x32).max() < 2e-3 * max(1.0, np.abs(x32).max())
    ctx.close()


@pytest.mark.parametrize("kind,d,hidden,F,gs,mode", [("lgcp", 64, 32, 16, 0.05, "rwmh"), ("phi4", 64, 32, 16, 1e-3, "imh")])
def test_wide_flow_step_matches_oracle(kind, d, hidden, F, gs, mode):
    import torch
    from mfm_amd import _lib
    B = 32
    args, dist, k, model, state = _setup(kind, d, B, hidden, F)
    params = _tamed(model, 9, 0.3 if kind == "lgcp" else 0.05, gs)
    ctx = _wide_ctx(dist, args, model, params)
    x32 = dist.init_params.astype(np.float32)
    beta = 0.7
    vg = targets.Tempered(dist, beta).value_and_grad
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    st = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    key = prng.PRNGKey(31)
    stats = {}
    step = flow.rwmh_step if mode == "rwmh" else flow.imh_step
    new, info = step(prng.split(key, B), st, vg, model, params, args, stats)
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda")
    prop = torch.empty(B, d, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH if mode == "rwmh" else _lib.FLOW_IMH, key, beta, pos, logp, grad, acc, isacc, prop, ns)
    p = prop.cpu().numpy()
    assert np.abs(p - info.proposed_position).max() < 5e-3 * max(1.0, np.abs(p).max())
    with np.errstate(divide="ignore"):
        la_g, la_o = np.log(acc.cpu().numpy().astype(np.float64)), np.log(info.acceptance_rate)
    fin = np.isfinite(la_g) & np.isfinite(la_o)
    if fin.any():
        assert np.abs(la_g[fin] - la_o[fin]).max() < 0.5
    sure = ~fin | (np.abs(la_o) > 1)
    np.testing.assert_array_equal(isacc.cpu().numpy().astype(bool)[sure], info.is_accepted[sure])
    same = isacc.cpu().numpy().astype(bool) == info.is_accepted
    np.testing.assert_allclose(logp.cpu().numpy()[same], new.logdensity[same], rtol=1e-4, atol=5e-2)
    np.testing.assert_allclose(grad.cpu().numpy()[same], new.logdensity_grad[same], rtol=1e-3, atol=5e-2)
    if mode == "rwmh":
        tot = stats["n_att_inv"] + stats["n_att_fwd"]
        assert abs(ns.float().mean().item() - tot.mean()) < 0.1 * tot.mean()
    ctx.close()


def test_pines_shape_is_served_by_the_wide_family_automatically():
    """BASELINE configs[4]: d = 1024 (32 x 32 grid), hidden width 1024 does not fit the 16-chain LDS tile: mfm_create picks
    the wide family by itself.  One MALA step + training step + CNF round trip at the full width."""
    import torch
    from tests import gpu_util as gu
    B, d = 128, 1024             # >= 128 chains: the MALA step takes the wide propose / K^-1 GEMM / accept split as well
    args, dist, k, model, state = gu.lgcp_setup(n=32, B=B, hidden=1024, F=128)
    params = _tamed(model, 2, 0.2, 0.02)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)           # kernel_family = AUTO
    x32 = dist.init_params.astype(np.float32)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    vg = targets.Tempered(dist, 1.0).value_and_grad
    st = mala.init(x32.astype(np.float64), vg)
    np.testing.assert_allclose(logp.cpu().numpy(), st.logdensity, rtol=1e-5)
    acc = torch.empty(B, device="cuda")
    ctx.mala_step(prng.PRNGKey(3), 1.0, args.step_size, pos, logp, grad, acc)
    new, info, _ = mala.kernel(prng.split(prng.PRNGKey(3), B), st, vg, args.step_size)
    np.testing.assert_allclose(acc.cpu().numpy(), info.acceptance_rate, atol=5e-3)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.ode_transform(1, _dev(x32), out, ldj, key=prng.PRNGKey(4), nsteps=ns)
    assert np.abs(out.cpu().numpy() - x32).max() > 1e-2 and ns.min().item() >= 1
    back = torch.empty(B, d, device="cuda"); l2 = torch.empty(B, device="cuda")
    ctx.ode_transform(-1, out, back, l2, key=prng.PRNGKey(4))
    assert np.abs(back.cpu().numpy() - x32).max() < 2e-3 * max(1.0, np.abs(x32).max())
    ctx.close()


def test_wide_lgcp_mala_at_the_reference_default_grid():
    """The reference's own pines default is the 40 x 40 grid (multi_modal.py:89, d = 1600): beyond the fused LGCP MALA tile,
    served by the wide family's propose / K^-1 GEMM / accept split.  MALAState and MALAInfo vs the oracle."""
    import torch
    from tests import gpu_util as gu
    B, d = 32, 1600
    args, dist, k, model, state = gu.lgcp_setup(n=40, B=B, hidden=1024, F=128)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=state.params)
    x32 = dist.init_params.astype(np.float32)
    beta = 0.6
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    vg = targets.Tempered(dist, beta).value_and_grad
    st = mala.init(x32.astype(np.float64), vg)
    np.testing.assert_allclose(logp.cpu().numpy(), st.logdensity, rtol=1e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), st.logdensity_grad, rtol=1e-3, atol=2e-3)
    st = mala.MALAState(st.position, logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    key = prng.PRNGKey(5)
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda"); prop = torch.empty(B, d, device="cuda")
    ctx.mala_step(key, beta, args.step_size, pos, logp, grad, acc, isacc, prop)
    new, info, u = mala.kernel(prng.split(key, B), st, vg, args.step_size)
    np.testing.assert_allclose(prop.cpu().numpy(), info.proposed_position, atol=1e-5)
    np.testing.assert_allclose(acc.cpu().numpy(), info.acceptance_rate, atol=5e-3)
    sure = np.abs(u - info.acceptance_rate) > 1e-2
    np.testing.assert_array_equal(isacc.cpu().numpy().astype(bool)[sure], info.is_accepted[sure])
    same = isacc.cpu().numpy().astype(bool) == info.is_accepted
    np.testing.assert_allclose(pos.cpu().numpy()[same], new.position[same], atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy()[same], new.logdensity[same], rtol=1e-5)
    ctx.close()


def test_wide_pines_loop_matches_oracle(monkeypatch):
    """The whole loop (annealing, MALA, flow-MH steps, training) on the LGCP target through the wide family."""
    from mfm_amd import _lib
    from tests.test_gpu_loop import _run_both
    monkeypatch.setenv("MFM_KERNEL_FAMILY", str(_lib.FAMILY_WIDE))
    out, res, ex = _run_both("pines", 64, 32, 8, 3, step_size=0.01)
    assert ex["engine"].ctx.cfg.kernel_family == _lib.FAMILY_WIDE
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=1e-5)
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=2e-2)
    np.testing.assert_allclose(ex["betas"], tr["beta"], rtol=2e-3)
    assert np.isfinite(res).all()
    ex["engine"].close()


# ---- exact-trace log-det on the wide family (README.md:58,60: `--example pines` WITHOUT --hutch; exe_flow_matching.py:216-217) ----------
def _tamed_lgcp(model, seed=9, out_scale=3.0):
    from tests import gpu_util as gu
    p = gu.rand_params(model, seed=seed, out_scale=out_scale)      # 16 x 16 / hidden 64: ~16 attempted steps, |log-det| ~ 2, points move by ~8
    p[4]["kernel"] *= 0.05; p[4]["bias"] *= 0.05
    return p


@pytest.mark.parametrize("solver", ["fused-aux", "fused-own"])
@pytest.mark.parametrize("direction", [1, -1])
def test_fused_family_exact_trace_on_both_solvers(monkeypatch, solver, direction):
    """A fused-family context without --hutch at d >= 16 sends its solves to the wide family's solver (api.hip: wide_ex; the reference's
    first phi-four command line trains 6.6 x faster for it); MFM_TILE_EXACT=1 keeps the generic tile's own d-tangent form.  Both against the
    oracle on the same prescribed step sequence (phi-four d = 64, hidden 32: the tile fits)."""
    if solver == "fused-own":
        monkeypatch.setenv("MFM_TILE_EXACT", "1")
    else:
        monkeypatch.delenv("MFM_TILE_EXACT", raising=False)
    test_wide_exact_trace_transform_on_prescribed_steps("phi4", 64, 32, 16, direction, family=None)


@pytest.mark.parametrize("kind,d,hidden,F", [("lgcp", 256, 64, 16), ("phi4", 144, 48, 16)])
@pytest.mark.parametrize("direction", [1, -1])
def test_wide_exact_trace_transform_on_prescribed_steps(kind, d, hidden, F, direction, family="wide"):
    """jnp.trace(jax.jacfwd(v)(x)) as the log-det integrand (no --hutch) on the wide family: hx1 masked tangent rows per chain through
    two GEMMs and a contraction with W_out W_x1 (wide.hip: exact_trace) against the oracle's d tangent columns, step for step on the
    oracle's step sequence: attempt counts exact, outputs at float32 rounding, log-det <= 1e-4 of |l|.  lgcp 16 x 16 / hidden 64;
    phi-four d = 144 > 128 switches the clip (and its 0/1 derivative on the Hessian diagonal) on."""
    import torch
    from tests.test_gpu_replay import _replay_arrays, _check_controller_tight, _tamed
    B = 32
    args, dist, k, model, state = _setup(kind, d, B, hidden, F, hutch=False)
    params = _tamed_lgcp(model) if kind == "lgcp" else _tamed(model)
    if family == "wide":
        ctx = _wide_ctx(dist, args, model, params)
    else:
        from tests import gpu_util as gu
        ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x64 = dist.init_params.astype(np.float32).astype(np.float64)
    keys = prng.split(prng.PRNGKey(21), B)
    fn = ode.transform_and_logdet if direction > 0 else ode.inverse_and_logdet
    o = (False, args.rtol, args.atol, args.mxstep)
    st = {}
    fn(model, params, keys, x64, *o, stats=st)
    dt, acc = _replay_arrays([st])
    st_o = {}
    y_o, l_o = fn(model, params, keys, x64, *o, stats=st_o, replay=dict(dt=dt[0].astype(np.float64), acc=acc[0]))
    assert st["n_attempted"].mean() > 10, st["n_attempted"].mean()
    ratio = torch.zeros(dt[0].shape, device="cuda"); own = torch.zeros(dt[0].shape, device="cuda")
    ctx.debug_replay(_dev(dt[0]), _dev(acc[0]), ratio, own)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    d_keys = _dev(keys.astype(np.uint32).view(np.int32))
    ctx.ode_transform(direction, _dev(x64.astype(np.float32)), out, ldj, keys=d_keys, nsteps=ns)
    y, l, n = out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy()
    np.testing.assert_array_equal(n, st["n_attempted"])
    ey, el, ls = np.abs(y - y_o).max(), np.abs(l - l_o), max(1.0, np.abs(l_o).max())
    mr, md = _check_controller_tight(f"exact {kind} dir={direction}", st_o, ratio.cpu().numpy(), own.cpu().numpy(), n)
    print(f"wide exact-trace transform {kind} d={d} dir={direction}: attempts {n.mean():.0f}, |dy| {ey:.2e}, |dl| max {el.max():.2e} (|l| {ls:.2f}), "
          f"controller medians {mr:.1e} {md:.1e}")
    assert np.abs(l_o).max() > 0.05                                # a log-det worth comparing
    assert ey < 3e-5 * max(1.0, np.abs(y_o).max()), ey
    # measured (MI355X): lgcp 2.4e-6 / 2.1e-6, phi-four 5.4e-7 forward; backward ONE chain of 32 at 1.2e-3: the clip's 0/1 derivative on the
    # Hessian diagonal (gate_i H_ii, |H_ii| ~ 600) flips for an element within float32 rounding of |grad log pi| = 1 -- an isolated
    # kink event as in tests/test_gpu_replay.py, bounded by a quantile; the smooth Cox target has none
    assert np.quantile(el, 0.9) < 1e-4 * ls and el.max() < (1e-4 if kind == "lgcp" else 5e-3) * ls, (np.quantile(el, 0.9), el.max(), ls)
    # natural controller: two adaptive solves of the same flow on their own step sequences (rtol = atol = 1e-5 on the RMS over d + 1
    # components; measured log-det differences 1.2e-2 .. 3.4e-2 of |l| 1.2 .. 2.2: the bound of tests/test_gpu_ode.py, 5 %)
    ctx.ode_transform(direction, _dev(x64.astype(np.float32)), out, ldj, keys=d_keys, nsteps=ns)
    en = np.abs(ldj.cpu().numpy() - l_o)
    print(f"   natural controllers: |dy| {np.abs(out.cpu().numpy() - y_o).max():.2e}, |dl| q90 {np.quantile(en, 0.9):.2e} max {en.max():.2e}")
    assert np.abs(out.cpu().numpy() - y_o).max() < 2e-3 * max(1.0, np.abs(y_o).max())
    assert np.quantile(en, 0.9) < 5e-2 * ls and en.max() < 0.2 * ls, (np.quantile(en, 0.9), en.max(), ls)
    ctx.close()


def test_wide_exact_trace_flow_step_on_prescribed_steps():
    """The flow-MH step with the exact trace on the wide family (lgcp 16 x 16, hidden 64): both solves on the oracle's step
    sequences, log-dets, proposal, log alpha, decisions."""
    from tests.test_gpu_replay import _flow_replay_raw
    B = 32
    args, dist, k, model, state = _setup("lgcp", 256, B, 64, 16, hutch=False)
    params = _tamed_lgcp(model)
    ctx = _wide_ctx(dist, args, model, params)
    r = _flow_replay_raw(ctx, model, params, args, dist, 0.8, dist.init_params.astype(np.float32), prng.PRNGKey(43))
    so, dg, info_o = r["so"], r["diag"], r["info_o"]
    np.testing.assert_array_equal(r["n_g"], r["n_o"])
    e_p = np.abs(r["prop"] - info_o.proposed_position).max()
    vs = max(1.0, np.abs(so["vol0"]).max(), np.abs(so["volp"]).max())
    e_v0, e_vp, e_la = np.abs(dg[:, 0] - so["vol0"]), np.abs(dg[:, 1] - so["volp"]), np.abs(dg[:, 3] - so["log_alpha"])
    print(f"wide exact-trace flow step: attempts {r['n_o'].mean():.0f}, |dx'| {e_p:.2e}, |dvol0| {e_v0.max():.2e}, |dvolp| {e_vp.max():.2e} (scale {vs:.2f}), "
          f"|d log alpha| med {np.median(e_la):.2e} max {e_la.max():.2e}")
    assert e_p < 3e-5 * max(1.0, np.abs(info_o.proposed_position).max())
    assert e_v0.max() < 1e-4 * vs and e_vp.max() < 1e-4 * vs, (e_v0.max(), e_vp.max(), vs)
    vg = targets.Tempered(dist, 0.8).value_and_grad
    gn = vg(info_o.proposed_position.astype(np.float64))[1]
    bound = 2.0 * np.linalg.norm(gn, axis=1) * np.linalg.norm(r["prop"] - info_o.proposed_position, axis=1) + 1e-4 * vs + 1e-3
    assert (e_la <= bound).all(), (e_la / bound).max()
    assert (r["isacc"] == info_o.is_accepted).mean() > 0.9
    ctx.close()


def test_wide_exact_trace_at_the_pines_width_on_prescribed_steps():
    """32 x 32 grid, hidden 1024 (the pines network of BASELINE configs[4] / multi_modal.py:96) with the exact trace: sixteen chains on
    the GPU, the oracle's d = 1024 tangent columns on two of them (75 s of oracle time; the other fourteen replay those two inputs and step
    sequences, so every row of the 16-row GEMM tiles is checked and equal inputs must give equal outputs)."""
    import torch
    from tests.test_gpu_replay import _replay_arrays
    B, Bo, d = 16, 2, 1024
    from tests import gpu_util as gu
    args, dist, k, model, state = _setup("lgcp", d, B, 1024, 128, hutch=False)
    params = gu.rand_params(model, seed=9, out_scale=2.0)          # 12-15 attempted steps, |log-det| ~ 1, points move by ~5
    params[4]["kernel"] *= 0.05; params[4]["bias"] *= 0.05
    ctx = _wide_ctx(dist, args, model, params)
    x4 = dist.init_params[:Bo].astype(np.float32).astype(np.float64)
    k4 = prng.split(prng.PRNGKey(23), B)[:Bo]
    o = (False, args.rtol, args.atol, args.mxstep)
    def natural():                                  # the oracle's own controller (cached: tests/gpu_util.py cached_oracle)
        st_ = {}
        ode.transform_and_logdet(model, params, k4, x4, *o, stats=st_)
        dt_n, ac_n = _replay_arrays([st_])
        return dict(dt=dt_n, acc=ac_n, n_attempted=st_["n_attempted"])
    seq = gu.cached_oracle("natseq_wide_exact_pines_width", natural, x4, k4, gu.flat_params(params))
    dt, acc, st = seq["dt"], seq["acc"], dict(n_attempted=seq["n_attempted"])
    y_o, l_o = ode.transform_and_logdet(model, params, k4, x4, *o, stats={}, replay=dict(dt=dt[0].astype(np.float64), acc=acc[0]))
    rep = lambda a: np.concatenate([a] * (B // Bo), axis=0)
    ratio = torch.zeros(B, dt.shape[2], device="cuda"); own = torch.zeros(B, dt.shape[2], device="cuda")
    ctx.debug_replay(_dev(rep(dt[0])), _dev(rep(acc[0])), ratio, own)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.ode_transform(1, _dev(rep(x4).astype(np.float32)), out, ldj, keys=_dev(rep(k4).astype(np.uint32).view(np.int32)), nsteps=ns)
    y, l, n = out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy()
    np.testing.assert_array_equal(n, rep(st["n_attempted"]))
    ls = max(1.0, np.abs(l_o).max())
    print(f"wide exact trace 32 x 32 / hidden 1024: attempts {n.mean():.1f}, |dy| {np.abs(y - rep(y_o)).max():.2e}, |dl| {np.abs(l - rep(l_o)).max():.2e} (|l| {np.abs(l_o).max():.3f})")
    assert np.abs(l_o).max() > 0.05
    assert np.abs(y - rep(y_o)).max() < 3e-5 * max(1.0, np.abs(y_o).max())
    assert np.abs(l - rep(l_o)).max() < 1e-4 * ls
    np.testing.assert_array_equal(y[:Bo], y[Bo:2 * Bo]); np.testing.assert_array_equal(l[:Bo], l[2 * Bo:3 * Bo])      # equal inputs, equal rows
    ctx.close()


# ---- the reference's OWN pines default: d = 1600 (40 x 40), 128 chains, hidden 1024 (multi_modal.py:89-96) ------------------------------
def test_wide_fm_loss_and_grad_at_the_reference_pines_default():
    """Flow-matching loss and every parameter gradient (exe_flow_matching.py:151-178, :364-365) at the reference's own pines shape,
    next to the MALA case above: same tolerances as the other shapes (loss 2e-5, gradients 2e-4 relative to the tensor's maximum)."""
    import torch
    from tests import gpu_util as gu
    B, d = 128, 1600
    args, dist, k, model, state = gu.lgcp_setup(n=40, B=B, hidden=1024, F=128)
    params = gu.rand_params(model, seed=3)
    ctx = _wide_ctx(dist, args, model, params)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(11)
    loss_o, grads_o = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.full((ctx.n_params,), float("nan"), device="cuda")
    ctx.fm_loss_grad(key, _dev(x32), loss, grads)
    assert abs(loss.item() - loss_o) <= 2e-5 * abs(loss_o), (loss.item(), loss_o)
    g = gu.unflat_params(model, grads.cpu().numpy())
    worst = 0.0
    for i, (gg, go) in enumerate(zip(g, grads_o)):
        for kk in ("kernel", "bias"):
            assert np.isfinite(gg[kk]).all()
            worst = max(worst, _relerr(gg[kk], go[kk].astype(np.float64)))
            assert _relerr(gg[kk], go[kk].astype(np.float64)) < 2e-4, (i, kk, _relerr(gg[kk], go[kk]))
    print(f"pines default (d = 1600, 128 chains, hidden 1024): loss {loss.item():.6e} vs {loss_o:.6e}, worst gradient tensor {worst:.1e}")
    ctx.close()


def test_wide_flow_step_at_the_reference_pines_default_on_prescribed_steps():
    """One Hutchinson flow-MH step (exe_flow_matching.py:264-278) at d = 1600 / 128 chains / hidden 1024 on the oracle's step sequences:
    attempt counts exact, proposal, both log-dets, log alpha term by term, decisions.  The kernels integrate all 128 chains; the float64
    oracle (minutes for 128 chains at this width) checks the first 4, the others replay one of those 4 step sequences, which ends by
    itself (as in tests/test_gpu_rank_slices.py)."""
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    from tests.test_gpu_replay import _replay_arrays
    B, Bo, d = 128, 4, 1600
    args, dist, k, model, state = gu.lgcp_setup(n=40, B=B, hidden=1024, F=128)
    params = gu.rand_params(model, seed=9, out_scale=2.0)
    params[4]["kernel"] *= 0.05; params[4]["bias"] *= 0.05
    ctx = _wide_ctx(dist, args, model, params)
    x32 = dist.init_params.astype(np.float32)
    beta = 0.8
    vg = targets.Tempered(dist, beta).value_and_grad
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    st0 = mala.MALAState(x32[:Bo].astype(np.float64), logp.cpu().numpy()[:Bo], grad.cpu().numpy()[:Bo].astype(np.float64))
    key = prng.PRNGKey(47)
    keys = prng.split(key, B)[:Bo]                                                       # :303: chain b uses split(key, B)[b]
    def natural():                                  # the oracle's own controller (cached: tests/gpu_util.py cached_oracle)
        nat = {}
        flow.rwmh_step(keys, st0, vg, model, params, args, nat)
        dt_n, ac_n = _replay_arrays([nat["inv"], nat["fwd"]])
        return dict(dt=dt_n, acc=ac_n)
    seq = gu.cached_oracle("natseq_wide_flow_pines_default", natural, x32[:Bo], keys, gu.flat_params(params))
    dt_s, ac_s = seq["dt"], seq["acc"]
    rp = dict(inv=dict(dt=dt_s[0].astype(np.float64), acc=ac_s[0]), fwd=dict(dt=dt_s[1].astype(np.float64), acc=ac_s[1]))
    so = {}
    new_o, info_o = flow.rwmh_step(keys, st0, vg, model, params, args, so, replay=rp)
    donor = np.arange(B) % Bo
    dt, ac = np.ascontiguousarray(dt_s[:, donor]), np.ascontiguousarray(ac_s[:, donor])
    ratio = torch.zeros(dt.shape, device="cuda"); own = torch.zeros(dt.shape, device="cuda"); diag = torch.zeros(B, 4, dtype=torch.float64, device="cuda")
    ctx.debug_replay(_dev(dt), _dev(ac), ratio, own, diag)
    a = torch.empty(B, device="cuda"); ia = torch.empty(B, dtype=torch.uint8, device="cuda"); pr = torch.empty(B, d, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, key, beta, pos, logp, grad, a, ia, pr, ns)
    n_o = so["n_att_inv"] + so["n_att_fwd"]
    np.testing.assert_array_equal(ns.cpu().numpy(), n_o[donor])                          # attempt counts: exact; every chain ends where its sequence ends
    dg = diag.cpu().numpy()[:Bo]
    vs = max(1.0, np.abs(so["vol0"]).max(), np.abs(so["volp"]).max())
    e_p = np.abs(pr.cpu().numpy()[:Bo] - info_o.proposed_position).max()
    e_v0, e_vp, e_la = np.abs(dg[:, 0] - so["vol0"]), np.abs(dg[:, 1] - so["volp"]), np.abs(dg[:, 3] - so["log_alpha"])
    print(f"pines default flow step: attempts {n_o.mean():.1f} (max {n_o.max()}), |dx'| {e_p:.2e}, |dvol0| {e_v0.max():.2e}, |dvolp| {e_vp.max():.2e} "
          f"(scale {vs:.2f}), |d log alpha| med {np.median(e_la):.2e} max {e_la.max():.2e}")
    assert n_o.mean() > 15
    assert e_p < 1e-4 * max(1.0, np.abs(info_o.proposed_position).max())                 # the bounds of the pines rank slices (tests/test_gpu_rank_slices.py)
    assert max(e_v0.max(), e_vp.max()) < 5e-3 * vs and np.median(np.maximum(e_v0, e_vp)) < 1e-4 * vs
    gn = vg(info_o.proposed_position.astype(np.float64))[1]
    bound = 2.0 * np.linalg.norm(gn, axis=1) * np.linalg.norm(pr.cpu().numpy()[:Bo] - info_o.proposed_position, axis=1) + e_v0 + e_vp + 1e-3
    assert (e_la <= bound).all(), (e_la / bound).max()
    assert (ia.cpu().numpy().astype(bool)[:Bo] == info_o.is_accepted).mean() > 0.8
    assert np.isfinite(pr.cpu().numpy()).all()
    ctx.close()
