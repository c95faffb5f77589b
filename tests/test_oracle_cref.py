"""libmfm_ref (oracle/cref/mfm_ref.c: the C / OpenMP float64 restatement of the headline configuration's inner loop) against the numpy
restatement, function by function on the same inputs.  CPU only.  Both are oracles (test infrastructure, PARITY UNPINNED): this
file says they state the same arithmetic -- values to rounding (the summation orders differ: BLAS / pairwise sums vs plain loops),
attempted-step counts of the adaptive solves equal."""
import numpy as np
import pytest

from oracle import cref, flow, fm, mala, ode, prng, targets
from tests import gpu_util as gu


def _setup(d, B, hidden=32, F=16, seed=3):
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F, seed=seed)
    params = gu.rand_params(model, seed=seed, out_scale=0.05)
    return args, dist, model, params, cref.CRef(model, params)


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def test_library_builds_and_reports_its_threads():
    assert cref.build().endswith("libmfm_ref.so")
    assert cref.lib().mfmref_threads() >= 1


@pytest.mark.parametrize("d", [64, 160])
def test_target_and_mala_step_equal_the_numpy_restatement(d):
    args, dist, model, params, cr = _setup(d, 24)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (24, d))
    for temper in (1.0, 0.37):
        vg = targets.Tempered(dist, temper).value_and_grad
        lp, g = vg(x)
        lpc, gc = cr.value_and_grad(x, temper)
        assert _rel(lpc, lp) < 1e-13 and _rel(gc, g) < 1e-13
        keys = prng.split(prng.PRNGKey(5), 24)
        noise = rng.standard_normal((24, d))
        seen = set()
        for step in (1e-4, 2e-5, 1e-5):     # far from equilibrium the rule as written (min(1, 1 / alpha)) rejects at 1e-4 and accepts at 1e-5
            for textbook in (False, True):
                st, info, u = mala.kernel(keys, mala.MALAState(x, lp, g), vg, step, textbook=textbook, noise=noise)
                stc, p, acc = cr.mala_step(mala.MALAState(x, lp, g), noise, u, step, temper, textbook=textbook)
                assert (acc == info.is_accepted).all()
                seen |= set(acc.tolist())
                assert np.abs(p - info.acceptance_rate).max() < 1e-9           # |log p| ~ 1e4 * 1e-13
                assert _rel(stc.position, st.position) < 1e-14 and _rel(stc.logdensity, st.logdensity) < 1e-13 and _rel(stc.logdensity_grad, st.logdensity_grad) < 1e-13
        assert seen == {True, False}


@pytest.mark.parametrize("d", [64, 160])          # 160 > 128: the clipped gate term (exe_flow_matching.py:88-89)
def test_vector_field_forward_and_jvp_equal_the_numpy_restatement(d):
    args, dist, model, params, cr = _setup(d, 16)
    assert (model.grad_clip is not None) == (d > 128)
    rng = np.random.default_rng(1)
    x, t, z = rng.uniform(-1, 1, (16, d)), rng.uniform(0, 1, 16), rng.standard_normal((16, d))
    v, jv = model.forward(params, x, t, tangent=z)
    vc, jvc = cr.forward(x, t, tangent=z)
    assert _rel(vc, v) < 1e-12 and _rel(jvc, jv) < 1e-12
    assert _rel(cr.forward(x, t), v) < 1e-12
    if d > 128:
        g = dist.grad_logprob(x)
        assert (np.abs(g) > 1.0).any() and (np.abs(g) <= 1.0).any()      # both sides of the clip are exercised


def test_flow_matching_loss_and_gradient_equal_the_numpy_restatement():
    args, dist, model, params, cr = _setup(160, 40)
    x1 = np.random.default_rng(2).uniform(-1, 1, (40, 160))
    key = prng.PRNGKey(11)
    loss, grads = fm.loss_and_grad(model, params, key, x1, args.sigma)
    t, cond, target = fm.cond_flow_batch(key, x1, args.sigma)
    lc, gc = cr.fm_loss_grad(t, cond, target)
    assert abs(lc - loss) < 1e-11 * abs(loss)
    for a, b in zip(gc, grads):
        for nm in ("kernel", "bias"):
            assert a[nm].dtype == np.float32 and a[nm].shape == b[nm].shape
            assert np.abs(a[nm].astype(np.float64) - b[nm]).max() <= 2e-7 * max(np.abs(b[nm]).max(), 1e-30), nm      # both round a float64 sum to float32


@pytest.mark.parametrize("sign", [+1, -1])
@pytest.mark.parametrize("strength", ["mild", "stiff"])
def test_adaptive_cnf_solve_equals_the_numpy_restatement(sign, strength):
    args, dist, model, params, cr = _setup(64, 12, hidden=32, F=16)
    # a field strong enough for the controller to work (rejections among the attempted steps); the gate layer is scaled down: unclipped
    # at d <= 128, grad log pi of a lattice far from equilibrium is O(1e3) and the field would blow up
    sc, osc = (1.0, 0.3) if strength == "mild" else (1.5, 1.0)
    params = gu.rand_params(model, seed=4, scale=sc, out_scale=osc)
    gl = model.zero_layers()[0]
    params[gl]["kernel"] *= np.float32(1e-3 / osc); params[gl]["bias"] *= np.float32(1e-3)
    cr.set_params(params)
    rng = np.random.default_rng(3)
    x0, z = rng.uniform(-1, 1, (12, 64)), rng.standard_normal((12, 64))
    st = {}
    f = ode.transform_and_logdet if sign > 0 else ode.inverse_and_logdet
    xo, ldj = f(model, params, None, x0, True, 1e-5, 1e-5, 1000, z=z, stats=st)
    sc_ = {}
    xc, lc = cr.solve(x0, z, sign, 1e-5, 1e-5, 1000, stats=sc_)
    n = st["n_attempted"]
    rejected = sum(int((~st["acc_seq"][b, :n[b]]).sum()) for b in range(12))
    assert n.min() >= 15 and rejected >= 12 and (n.max() >= 100) == (strength == "stiff")
    assert (sc_["n_attempted"] == n).all(), (sc_["n_attempted"], n)
    assert sc_["n_evals_total"] == int((2 + 6 * n).sum())
    # measured: 6e-12 / 5e-11 (mild, ~22 attempted steps), 6e-8 / 1.5e-6 (stiff, up to 130: every relu kink a step crosses amplifies
    # the rounding difference of the two summation orders)
    tol = 1e-9 if strength == "mild" else 1e-4
    assert np.abs(xc - xo).max() < tol and np.abs(lc - ldj).max() < tol * max(1.0, np.abs(ldj).max())


def test_keyed_mala_kernel_and_flow_mh_step_equal_the_numpy_restatement():
    """The two composed steps the CPU baseline of bench.py times (keys -> draws by oracle/prng.py, arithmetic in C)."""
    args, dist, model, params, cr = _setup(64, 12, hidden=32, F=16)
    params = gu.rand_params(model, seed=4, scale=1.0, out_scale=0.3)
    gl = model.zero_layers()[0]
    params[gl]["kernel"] *= np.float32(1e-3 / 0.3); params[gl]["bias"] *= np.float32(1e-3)
    cr.set_params(params)
    vg = targets.Tempered(dist, 1.0).value_and_grad
    x = np.random.default_rng(5).uniform(-1, 1, (12, 64))
    st0 = mala.init(x, vg)
    keys = prng.split(prng.PRNGKey(9), 12)
    a, ia, _ = mala.kernel(keys, st0, vg, 1e-5)
    b, ib = cr.mala_kernel(keys, st0, 1e-5)
    assert (ib.is_accepted == ia.is_accepted).all() and _rel(b.position, a.position) < 1e-14 and _rel(b.logdensity, a.logdensity) < 1e-13
    args.beta = 1.0
    so, sc = {}, {}
    fo, io = flow.rwmh_step(keys, st0, vg, model, params, args, so)
    fc, ic = cr.rwmh_step(keys, st0, args, stats=sc)
    assert (sc["n_att_inv"] == so["n_att_inv"]).all() and (sc["n_att_fwd"] == so["n_att_fwd"]).all()
    assert np.abs(sc["log_alpha"] - so["log_alpha"]).max() < 1e-7 * max(1.0, np.abs(so["log_alpha"]).max())
    assert (ic.is_accepted == io.is_accepted).all()
    assert _rel(fc.position, fo.position) < 1e-9 and _rel(fc.logdensity, fo.logdensity) < 1e-9


def test_recorded_step_sequences_and_their_replay_equal_the_numpy_restatement():
    """ode.odeint's parity instrumentation in C: the recorded step sequence of a natural solve equals numpy's (float64 step sizes to rounding,
    decisions exactly), and replaying a float32-rounded sequence gives numpy's replayed result."""
    args, dist, model, params, cr = _setup(64, 12, hidden=32, F=16)
    params = gu.rand_params(model, seed=4, scale=1.0, out_scale=0.3)
    gl = model.zero_layers()[0]
    params[gl]["kernel"] *= np.float32(1e-3 / 0.3); params[gl]["bias"] *= np.float32(1e-3)
    cr.set_params(params)
    rng = np.random.default_rng(3)
    x0, z = rng.uniform(-1, 1, (12, 64)), rng.standard_normal((12, 64))
    st, sc = {}, {}
    ode.transform_and_logdet(model, params, None, x0, True, 1e-5, 1e-5, 1000, z=z, stats=st)
    cr.solve(x0, z, +1, 1e-5, 1e-5, 1000, stats=sc, record=64)
    A = st["acc_seq"].shape[1]
    np.testing.assert_array_equal(sc["acc_seq"][:, :A], st["acc_seq"])
    assert not sc["acc_seq"][:, A:].any()
    np.testing.assert_allclose(sc["dt_seq"][:, :A + 1], st["dt_seq"], rtol=1e-7, atol=0)      # (the controller amplifies the rounding of the error norm: measured 4e-9)
    rp = dict(dt=st["dt_seq"].astype(np.float32).astype(np.float64), acc=st["acc_seq"])
    so = {}
    xo, lo = ode.transform_and_logdet(model, params, None, x0, True, 1e-5, 1e-5, 1000, z=z, stats=so, replay=rp)
    s2 = {}
    xc, lc = cr.solve(x0, z, +1, 1e-5, 1e-5, 1000, stats=s2, replay=rp)
    assert (s2["n_attempted"] == so["n_attempted"]).all()
    assert np.abs(xc - xo).max() < 1e-11 and np.abs(lc - lo).max() < 1e-10
