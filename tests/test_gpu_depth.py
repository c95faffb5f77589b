"""Hidden lists of a length other than two (``multi_modal.py:178-180``: ``nargs='+'``; the loops at ``exe_flow_matching.py:74-85``):
one and three hidden layers per branch, and mixed depths, on the wide kernel family (one GEMM launch per layer) against the float64
oracle -- loss and gradient, the optimizer's repack of eleven layers, field and JVP, both log-det integrands on the oracle's step
sequences, a flow-MH step, and the reference's command line ``--hidden_x 128 128 128`` through the Python front end."""
import numpy as np
import pytest

from oracle import fm, ode, optim, prng

pytestmark = pytest.mark.gpu

# (hidden_x, hidden_t, hidden_xt)
ONE = ([32], [32], [32])
THREE = ([32, 48, 32], [48, 32, 16], [64, 32, 48])
MIXED = ([48], [32, 16, 32], [32, 64])          # one x layer: its output IS sx; three t layers; two joint layers
MIXED2 = ([32, 48], [16], [32, 16, 32])
RAGGED = ([20, 50], [30, 24], [100, 36])        # widths that are not multiples of 16 (multi_modal.py:178-180 takes any int): zero-padded; two layers
                                                # per branch, so this one runs on the FUSED tile family (the deeper ragged case below: wide)
RAGGED3 = ([24, 40, 20], [10], [50])
DEPTHS = {"1-1-1": ONE, "3-3-3": THREE, "x1-t3-j2": MIXED, "x2-t1-j3": MIXED2, "ragged": RAGGED, "ragged-x3-t1-j1": RAGGED3}


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


def _tamed(model, out_scale, gate, seed=9):
    from tests import gpu_util as gu
    p = gu.rand_params(model, seed=seed, out_scale=out_scale)
    g = model.zero_layers()[0]                         # the gate layer: t.., x.., GATE, joint.., out
    p[g]["kernel"] *= gate; p[g]["bias"] *= gate
    return p


def test_ragged_two_layer_network_runs_on_the_fused_family_and_matches_the_wide_one():
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=64, B=32, hidden=RAGGED, F=16)
    params = gu.rand_params(model, seed=5)
    x32 = dist.init_params.astype(np.float32)
    res = []
    for fam in (_lib.FAMILY_TILE, _lib.FAMILY_WIDE):
        ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, family=fam)
        assert ctx.cfg.kernel_family == fam
        loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
        ctx.fm_loss_grad(prng.PRNGKey(7), _dev(x32), loss, grads)
        res.append((loss.item(), grads.cpu().numpy()))
        ctx.close()
    assert abs(res[0][0] - res[1][0]) < 1e-6 * abs(res[0][0])
    assert np.abs(res[0][1] - res[1][1]).max() < 2e-5 * np.abs(res[0][1]).max()


def test_two_layer_request_on_the_tile_family_and_bad_depths_are_named():
    from mfm_amd import _lib
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=64, B=16, hidden=THREE, F=16)
    with pytest.raises(_lib.MfmError, match="wide kernel family"):
        gu.make_ctx(dist, args, fourier=model.f, family=_lib.FAMILY_TILE)
    ctx = gu.make_ctx(dist, args, fourier=model.f)      # AUTO: the wide family
    assert ctx.n_params == sum(fi * fo + fo for fi, fo in model.layer_shapes())
    ctx.close()
    with pytest.raises(_lib.MfmError, match="hidden layers"):
        _lib.Context(dim=64, n_chain_local=16, hidden_x=[32, 32, 32, 32])


@pytest.mark.parametrize("kind,d,B,F,act", [("phi4", 40, 16, 10, "relu"), ("lgcp", 64, 32, 16, "tanh"), ("phi4", 144, 32, 16, "gelu")])
@pytest.mark.parametrize("depth", list(DEPTHS))
def test_fm_loss_and_grad_match_oracle(depth, kind, d, B, F, act):
    """Every layer's kernel and bias gradient (a layer list of 5 to 11 entries), relu / tanh (derivative from the stored output) and
    gelu (stored pre-activations); then AdamW steps on oracle-side gradients: master parameters AND the packed copies the GEMMs read."""
    import torch
    from tests import gpu_util as gu
    args, dist, k, model, state = _setup(kind, d, B, DEPTHS[depth], F, non_linearity=act, learning_iter=9)
    params = gu.rand_params(model, seed=3)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(11)
    loss_o, grads_o = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.full((ctx.n_params,), float("nan"), device="cuda")
    ctx.fm_loss_grad(key, _dev(x32), loss, grads)
    assert abs(loss.item() - loss_o) <= 2e-5 * abs(loss_o), (loss.item(), loss_o)
    g = gu.unflat_params(model, grads.cpu().numpy())
    assert len(g) == len(DEPTHS[depth][0]) + len(DEPTHS[depth][1]) + len(DEPTHS[depth][2]) + 2
    for i, (gg, go) in enumerate(zip(g, grads_o)):
        for kk in ("kernel", "bias"):
            assert np.isfinite(gg[kk]).all()
            assert _relerr(gg[kk], go[kk].astype(np.float64)) < 2e-4, (i, kk, _relerr(gg[kk], go[kk]))
    st = optim.TrainState(params, optim.learning_rate_fn(9, 0, args.learning_rate))
    rng = np.random.default_rng(0)
    for it in range(3):
        gr = [{kk: (rng.standard_normal(v.shape) * 3).astype(np.float32) for kk, v in p.items()} for p in params]
        st.apply_gradients(gr)
        ctx.adamw_step(_dev(gu.flat_params(gr)))
        np.testing.assert_allclose(ctx.get_params(), gu.flat_params(st.params), rtol=3e-6, atol=1e-7)
    l2 = torch.zeros(1, dtype=torch.float64, device="cuda")
    ctx.fm_loss(prng.PRNGKey(1), _dev(x32), l2)
    lo, _ = fm.loss_and_grad(model, st.params, prng.PRNGKey(1), x32.astype(np.float64), args.sigma, need_grad=False)
    assert abs(l2.item() - lo) < 3e-5 * abs(lo)
    ctx.close()


@pytest.mark.parametrize("kind,d,F", [("phi4", 40, 10), ("lgcp", 64, 16), ("phi4", 256, 128)])
@pytest.mark.parametrize("depth", list(DEPTHS))
def test_vector_field_and_jvp_match_oracle(depth, kind, d, F):
    import torch
    from tests import gpu_util as gu
    B = 32
    args, dist, k, model, state = _setup(kind, d, B, DEPTHS[depth], F)
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
    ctx.close()


@pytest.mark.parametrize("hutch", [True, False])
@pytest.mark.parametrize("direction", [1, -1])
@pytest.mark.parametrize("depth", list(DEPTHS))
def test_transform_on_prescribed_steps_matches_oracle(depth, direction, hutch):
    """The CNF solve with the Hutchinson estimate and with the exact trace (``exe_flow_matching.py:215-217``; the exact trace's hops
    behind the first x layer change with the depth: x1 alone seeds the first joint layer, a single joint layer closes the trace right
    after the seed) on the oracle's step sequence: attempt counts exact, outputs and log-det at float32 rounding."""
    import torch
    from tests import gpu_util as gu
    from tests.test_gpu_replay import _replay_arrays, _check_controller_tight
    B, d = 32, 64
    args, dist, k, model, state = _setup("lgcp", d, B, DEPTHS[depth], 16, hutch=hutch)
    params = _tamed(model, 0.5, 0.05)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x64 = dist.init_params.astype(np.float32).astype(np.float64)
    keys = prng.split(prng.PRNGKey(21), B)
    fn = ode.transform_and_logdet if direction > 0 else ode.inverse_and_logdet
    o = (hutch, args.rtol, args.atol, args.mxstep)
    st = {}
    fn(model, params, keys, x64, *o, stats=st)
    dt, acc = _replay_arrays([st])
    st_o = {}
    y_o, l_o = fn(model, params, keys, x64, *o, stats=st_o, replay=dict(dt=dt[0].astype(np.float64), acc=acc[0]))
    ratio = torch.zeros(dt[0].shape, device="cuda"); own = torch.zeros(dt[0].shape, device="cuda")
    ctx.debug_replay(_dev(dt[0]), _dev(acc[0]), ratio, own)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.ode_transform(direction, _dev(x64.astype(np.float32)), out, ldj, keys=_dev(keys.astype(np.uint32).view(np.int32)), nsteps=ns)
    y, l, n = out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy()
    np.testing.assert_array_equal(n, st["n_attempted"])
    ey, el, ls = np.abs(y - y_o).max(), np.abs(l - l_o), max(1.0, np.abs(l_o).max())
    _check_controller_tight(f"depth {depth} dir={direction} hutch={hutch}", st_o, ratio.cpu().numpy(), own.cpu().numpy(), n)
    print(f"depth {depth} dir={direction} hutch={hutch}: attempts {n.mean():.0f}, |dx| {np.abs(y_o - x64).max():.2f}, |dy| {ey:.2e}, |dl| max {el.max():.2e} (|l| {ls:.2f})")
    assert n.mean() > 6 and np.abs(l_o).max() > 0.02, (n.mean(), np.abs(l_o).max())
    assert ey < 3e-5 * max(1.0, np.abs(y_o).max()), ey
    assert el.max() < 1e-4 * ls, (el.max(), ls)
    ctx.close()


@pytest.mark.parametrize("act", ["gelu", "tanh", "swish"])
@pytest.mark.parametrize("depth", ["3-3-3", "x1-t3-j2", "ragged"])
def test_exact_trace_with_smooth_activations_on_prescribed_steps(depth, act):
    """The exact trace's masks act'(.) come from the stored outputs (tanh) or from the stored pre-activations (gelu, swish: one stash per
    hidden layer of whatever depth); act'(0) != 0 for all three, so the padded units of the ragged case must be cut by zero weights (GEMM
    hops) and by the seed kernel's row guard.  `ragged` runs on the fused family with its solves on the wide solver, the others on the wide family."""
    import torch
    from tests import gpu_util as gu
    from tests.test_gpu_replay import _replay_arrays
    B, d = 32, 64
    args, dist, k, model, state = _setup("lgcp", d, B, DEPTHS[depth], 16, hutch=False, non_linearity=act)
    params = _tamed(model, 0.5, 0.05)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x64 = dist.init_params.astype(np.float32).astype(np.float64)
    keys = prng.split(prng.PRNGKey(21), B)
    o = (False, args.rtol, args.atol, args.mxstep)
    st = {}
    ode.transform_and_logdet(model, params, keys, x64, *o, stats=st)
    dt, acc = _replay_arrays([st])
    y_o, l_o = ode.transform_and_logdet(model, params, keys, x64, *o, stats={}, replay=dict(dt=dt[0].astype(np.float64), acc=acc[0]))
    ratio = torch.zeros(dt[0].shape, device="cuda"); own = torch.zeros(dt[0].shape, device="cuda")
    ctx.debug_replay(_dev(dt[0]), _dev(acc[0]), ratio, own)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.ode_transform(1, _dev(x64.astype(np.float32)), out, ldj, keys=_dev(keys.astype(np.uint32).view(np.int32)), nsteps=ns)
    y, l, n = out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy()
    np.testing.assert_array_equal(n, st["n_attempted"])
    ey, el = np.abs(y - y_o).max(), np.abs(l - l_o).max()
    print(f"exact trace {act} depth {depth}: attempts {n.mean():.1f}, |l| {np.abs(l_o).max():.2f}, |dy| {ey:.2e}, |dl| {el:.2e}")
    assert n.mean() > 3 and np.abs(l_o).max() > 0.2
    assert ey < 3e-5 * max(1.0, np.abs(y_o).max()) and el < 1e-4 * max(1.0, np.abs(l_o).max()), (ey, el)
    ctx.close()


@pytest.mark.parametrize("depth,hutch", [("1-1-1", False), ("3-3-3", True), ("x1-t3-j2", True), ("x2-t1-j3", False), ("ragged", False), ("ragged-x3-t1-j1", True)])
def test_flow_step_on_prescribed_steps_matches_oracle(depth, hutch):
    from tests import gpu_util as gu
    from tests.test_gpu_replay import _flow_replay_raw
    from oracle import targets
    B = 32
    args, dist, k, model, state = _setup("lgcp", 64, B, DEPTHS[depth], 16, hutch=hutch)
    params = _tamed(model, 0.5, 0.05)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    r = _flow_replay_raw(ctx, model, params, args, dist, 0.8, dist.init_params.astype(np.float32), prng.PRNGKey(43))
    so, dg, info_o = r["so"], r["diag"], r["info_o"]
    np.testing.assert_array_equal(r["n_g"], r["n_o"])
    e_p = np.abs(r["prop"] - info_o.proposed_position).max()
    vs = max(1.0, np.abs(so["vol0"]).max(), np.abs(so["volp"]).max())
    e_v0, e_vp, e_la = np.abs(dg[:, 0] - so["vol0"]), np.abs(dg[:, 1] - so["volp"]), np.abs(dg[:, 3] - so["log_alpha"])
    print(f"depth {depth} flow step: attempts {r['n_o'].mean():.0f}, |dx'| {e_p:.2e}, |dvol0| {e_v0.max():.2e}, |dvolp| {e_vp.max():.2e} (scale {vs:.2f}), |d log alpha| max {e_la.max():.2e}")
    assert e_p < 3e-5 * max(1.0, np.abs(info_o.proposed_position).max())
    assert e_v0.max() < 1e-4 * vs and e_vp.max() < 1e-4 * vs, (e_v0.max(), e_vp.max(), vs)
    vg = targets.Tempered(dist, 0.8).value_and_grad
    gn = vg(info_o.proposed_position.astype(np.float64))[1]
    bound = 2.0 * np.linalg.norm(gn, axis=1) * np.linalg.norm(r["prop"] - info_o.proposed_position, axis=1) + 1e-4 * vs + 1e-3
    assert (e_la <= bound).all(), (e_la / bound).max()
    assert (r["isacc"] == info_o.is_accepted).mean() > 0.9
    ctx.close()


@pytest.mark.parametrize("width", [([32, 32, 32], [32, 32, 32], [32, 32, 32]), ([32], [32], [32]), ([100, 100], [100, 100], [100, 100])], ids=["three", "one", "hundred"])
def test_phi4_loop_with_other_depths_matches_oracle(width):
    """``multi_modal.py --example phi-four --hidden_x h h h --hidden_t h h h --hidden_xt h h h`` (one layer per branch; width 100) through the
    Python front end against the oracle's loop: traces before the first flow step at rounding, then the bounds of tests/test_gpu_loop.py."""
    from tests.test_gpu_loop import _run_both
    out, res, ex = _run_both("phi-four", 64, 64, 12, 3, width=width, step_size=1e-4)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=1e-6)
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=5e-3)
    np.testing.assert_allclose(ex["lrs"], tr["learning_rate"], rtol=1e-12)
    np.testing.assert_allclose(ex["betas"], tr["beta"], rtol=2e-3)
    g = ex["states"].position.cpu().numpy().astype(np.float64)
    o = out["states"].position
    dmax = np.abs(g - o).max(1)
    flipped = dmax > 0.05
    assert flipped.sum() <= 6, dmax
    assert dmax[~flipped].max() < 2e-2
    from tests import gpu_util as gu
    po = gu.flat_params(out["state"].params)
    pg = ex["engine"].ctx.get_params()
    assert pg.shape == po.shape
    assert np.abs(pg - po).max() < (1e-3 if not flipped.any() else 3e-3) * max(1.0, np.abs(po).max())
    s = ex["engine"].ctx.opt_state()
    assert (s["step"], s["count"]) == (out["state"].step, out["state"].count)
    ex["engine"].close()


# ---- the Gaussian mixtures (d = 2) on the wide family: what `--example 4-mode --hidden_x h h h` needs -----------------------------------
GMM_CASES = [("2-2-2", ([32, 32], [32, 32], [32, 32])), ("3-3-3", THREE), ("1-1-1", ONE), ("x1-t3-j2", MIXED), ("ragged", RAGGED)]


def _gmm_ctx(hidden, B, **kw):
    from mfm_amd import _lib
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.gmm4_setup(B=B, hidden=hidden, F=16, **kw)
    params = gu.rand_params(model, seed=9, out_scale=0.3)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, family=_lib.FAMILY_WIDE)
    return args, dist, model, params, ctx


@pytest.mark.parametrize("name,hidden", GMM_CASES)
def test_gmm_loss_gradient_field_and_jvp_on_the_wide_family(name, hidden):
    import torch
    from tests import gpu_util as gu
    B, d = 32, 2
    args, dist, model, params, ctx = _gmm_ctx(hidden, B)
    x32 = dist.init_params.astype(np.float32)
    key = prng.PRNGKey(11)
    loss_o, grads_o = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.full((ctx.n_params,), float("nan"), device="cuda")
    ctx.fm_loss_grad(key, _dev(x32), loss, grads)
    assert abs(loss.item() - loss_o) <= 2e-5 * abs(loss_o), (loss.item(), loss_o)
    for i, (gg, go) in enumerate(zip(gu.unflat_params(model, grads.cpu().numpy()), grads_o)):
        for kk in ("kernel", "bias"):
            assert _relerr(gg[kk], go[kk].astype(np.float64)) < 2e-4, (i, kk, _relerr(gg[kk], go[kk]))
    rng = np.random.default_rng(1)
    t = rng.uniform(0, 1, B).astype(np.float32); z = rng.standard_normal((B, d)).astype(np.float32)
    v_o, jv_o = model.forward(params, x32.astype(np.float64), t.astype(np.float64), tangent=z.astype(np.float64))
    v = torch.empty(B, d, device="cuda"); jv = torch.empty(B, d, device="cuda")
    ctx.vf_apply(_dev(x32), _dev(t), v, _dev(z), jv)
    assert _relerr(v.cpu().numpy(), v_o) < 2e-5 and _relerr(jv.cpu().numpy(), jv_o) < 2e-5
    ctx.close()


@pytest.mark.parametrize("hutch", [True, False])
@pytest.mark.parametrize("name,hidden", GMM_CASES)
def test_gmm_flow_step_on_prescribed_steps_on_the_wide_family(name, hidden, hutch):
    """Hutchinson and exact trace (the mixtures' default): H z / the diagonal of H of the mixture's log-density in the gate term."""
    from tests.test_gpu_replay import _flow_replay_raw
    B = 32
    args, dist, model, params, ctx = _gmm_ctx(hidden, B, hutchs=hutch)
    r = _flow_replay_raw(ctx, model, params, args, dist, 0.7, dist.init_params.astype(np.float32), prng.PRNGKey(31))
    so, dg, info_o = r["so"], r["diag"], r["info_o"]
    np.testing.assert_array_equal(r["n_g"], r["n_o"])
    e_p = np.abs(r["prop"] - info_o.proposed_position).max()
    e_v = max(np.abs(dg[:, 0] - so["vol0"]).max(), np.abs(dg[:, 1] - so["volp"]).max())
    e_a = np.abs(dg[:, 3] - so["log_alpha"]).max()
    print(f"gmm {name} hutch={hutch} flow step (wide): attempts {r['n_o'].mean():.0f}, |dx'| {e_p:.2e}, |dvol| {e_v:.2e}, |d log alpha| {e_a:.2e}")
    assert r["n_o"].mean() > 10
    assert e_p < 1e-4 * max(1.0, np.abs(info_o.proposed_position).max()) and e_v < 1e-3 and e_a < 5e-3      # the bounds of tests/test_gpu_d2tile.py
    assert (r["isacc"] == info_o.is_accepted).mean() > 0.9
    same = r["isacc"] == info_o.is_accepted
    np.testing.assert_allclose(r["logp"][same], r["new_o"].logdensity[same], rtol=1e-4, atol=5e-3)
    ctx.close()


def test_four_mode_loop_with_three_hidden_layers_matches_oracle():
    """``multi_modal.py --example 4-mode --hidden_x h h h --hidden_t h h h --hidden_xt h h h`` (exact trace, n_ts = 5)."""
    from tests.test_gpu_loop import _run_both
    out, res, ex = _run_both("4-mode", 2, 64, 12, 3, hutch=False, width=([32, 32, 32], [32, 32, 32], [32, 32, 32]), step_size=0.2)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=1e-5)
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=5e-2)
    np.testing.assert_allclose(ex["betas"], tr["beta"], rtol=2e-3)
    assert np.isfinite(res).all()
    ex["engine"].close()


DEEP = ([32, 40, 24], [20], [48, 32])


@pytest.mark.parametrize("opts", [dict(num_importance_samples=-1), dict(num_importance_samples=3), dict(ref_dist="widegauss", num_importance_samples=-1),
                                  dict(cond_flow=False), dict(non_linearity="tanh")],
                         ids=["imh", "cis", "widegauss-imh", "no-cond-flow", "tanh"])
def test_phi4_loop_options_with_a_deep_ragged_network(opts):
    """The other flow kernels and options of the loop (exe_flow_matching.py:246-260 independent MH, :280-296 conditional importance
    sampling, the wide reference distribution, the unconditional flow-matching batch, another activation) with hidden lists 3 / 1 / 2 of
    widths that are not multiples of 16: wide family, against the oracle's loop."""
    from tests.test_gpu_loop import _run_both
    out, res, ex = _run_both("phi-four", 64, 64, 8, 3, width=DEEP, step_size=1e-4, **opts)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=2e-5)          # before the first flow step: same chains, same noise
    assert np.isfinite(m[:, 0]).all() and np.isfinite(np.array(tr["loss"])).all()
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=5e-2)               # after it: borderline decisions / near-tie categorical draws may differ
    np.testing.assert_allclose(ex["betas"], tr["beta"], rtol=2e-3)
    assert np.isfinite(res).all()
    ex["engine"].close()
