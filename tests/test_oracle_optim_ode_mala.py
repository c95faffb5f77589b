"""Pins the oracle's optimizer chain (vs torch.optim.AdamW), Dopri5 (closed forms, scipy RK45),
MALA energy algebra (literal per-chain transcription) and beta bisection."""
import numpy as np
import pytest
import torch
from scipy.integrate import solve_ivp

from oracle import flow, mala, ode, optim, prng, targets


def test_adamw_chain_matches_torch():
    rng = np.random.default_rng(0)
    params = [{"kernel": rng.standard_normal((5, 4)).astype(np.float32), "bias": rng.standard_normal(4).astype(np.float32)}]
    n_steps, lr0 = 7, 1e-2
    st = optim.TrainState(params, optim.learning_rate_fn(n_steps, 0, lr0))
    W = torch.tensor(params[0]["kernel"].copy(), requires_grad=True)
    b = torch.tensor(params[0]["bias"].copy(), requires_grad=True)
    opt = torch.optim.AdamW([{"params": [W], "weight_decay": 1e-4}, {"params": [b], "weight_decay": 0.0}],
                            lr=lr0, betas=(0.9, 0.999), eps=1e-8)
    for s in range(n_steps):
        g = {"kernel": rng.standard_normal((5, 4)).astype(np.float32), "bias": rng.standard_normal(4).astype(np.float32)}
        lr = lr0 * (1 - s / n_steps)
        for grp in opt.param_groups:
            grp["lr"] = lr
        W.grad, b.grad = torch.tensor(g["kernel"]), torch.tensor(g["bias"])
        opt.step()
        assert st.apply_gradients([g])
        # torch decays with p *= 1 - lr*wd BEFORE the adam term; optax adds wd*p to the update: equal to O(lr^2 wd)
        np.testing.assert_allclose(st.params[0]["kernel"], W.detach().numpy(), rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(st.params[0]["bias"], b.detach().numpy(), rtol=2e-6, atol=2e-7)
    assert st.step == n_steps and st.count == n_steps


def test_apply_if_finite_and_clip():
    params = [{"kernel": np.ones((2, 2), np.float32), "bias": np.zeros(2, np.float32)}]
    st = optim.TrainState(params, lambda c: 5.0)     # huge lr: the +-1 clip of the update must bite
    bad = [{"kernel": np.array([[np.nan, 0], [0, 0]], np.float32), "bias": np.zeros(2, np.float32)}]
    assert not st.apply_gradients(bad)
    assert st.step == 1 and st.count == 0 and st.notfinite_count == 1
    np.testing.assert_array_equal(st.params[0]["kernel"], 1.0)
    good = [{"kernel": np.ones((2, 2), np.float32), "bias": np.ones(2, np.float32)}]
    assert st.apply_gradients(good) and st.notfinite_count == 0
    np.testing.assert_allclose(st.params[0]["kernel"], 0.0)      # 1 + clip(-5*(1+1e-4), -1, 1)
    for _ in range(10):
        assert not st.apply_gradients(bad)
    assert st.apply_gradients(bad)                   # 11th consecutive error: update goes through (NaN)
    assert np.isnan(st.params[0]["kernel"]).any()
    lr = optim.learning_rate_fn(100, 0, 1e-3)
    assert lr(0) == 1e-3 and abs(lr(50) - 5e-4) < 1e-18 and lr(100) == 0.0


def test_dopri_tableau_consistency():
    for i, row in enumerate(ode.BETA):
        assert abs(row.sum() - ode.ALPHA[i]) < 1e-15
    assert abs(ode.C_SOL.sum() - 1) < 1e-15 and abs(ode.C_ERR.sum()) < 1e-15 and abs(ode.C_MID.sum() - 0.5) < 1e-15


def test_odeint_linear_closed_form_and_scipy():
    rng = np.random.default_rng(1)
    n, B = 4, 5
    A = rng.standard_normal((n, n)) * 0.7
    y0 = rng.standard_normal((B, n))
    fun = lambda y, t: y @ A.T
    stats = {}
    ys = ode.odeint(fun, y0, np.linspace(0, 1, 5), 1e-5, 1e-5, 1000, stats)
    import scipy.linalg as sla
    for j, tt in enumerate(np.linspace(0, 1, 5)):
        np.testing.assert_allclose(ys[j], y0 @ sla.expm(A * tt).T, rtol=2e-4, atol=2e-4)
    for bidx in range(B):
        sol = solve_ivp(lambda t, y: A @ y, (0, 1), y0[bidx], method="RK45", rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ys[-1][bidx], sol.y[:, -1], rtol=2e-4, atol=2e-4)
    assert stats["n_evals"] >= 2 + 6 * stats["n_attempted"].max() and stats["n_attempted"].min() >= 3
    # per-chain independence: solving one chain alone gives the identical result (vmap semantics)
    y1 = ode.odeint(fun, y0[2:3], np.linspace(0, 1, 5), 1e-5, 1e-5, 1000)
    np.testing.assert_allclose(y1[:, 0], ys[:, 2], rtol=1e-12)
    # time-dependent field with per-chain times
    ys2 = ode.odeint(lambda y, t: y * np.cos(t)[:, None], y0, np.array([0.0, 1.0]), 1e-5, 1e-5, 1000)
    np.testing.assert_allclose(ys2[-1], y0 * np.exp(np.sin(1.0)), rtol=1e-4)


def test_odeint_fifth_order():
    errs = []
    for dt in (0.2, 0.1, 0.05):
        y = np.ones((1, 1)); t = np.zeros(1)
        fun = lambda y, t: y * np.cos(t)[:, None]
        for _ in range(int(round(1 / dt))):
            y, _, _, _ = ode.runge_kutta_step(fun, y, fun(y, t), t, np.full(1, dt)); t = t + dt
        errs.append(abs(y[0, 0] - np.exp(np.sin(1.0))))
    assert errs[0] / errs[1] > 25 and errs[1] / errs[2] > 25


def test_fixed_step_integrators_orders_closed_forms_and_scipy():
    """oracle.ode.odeint_fixed (the build-side RK4 / Euler mode BASELINE.json's north star names; the reference itself integrates
    with the adaptive Dopri5): closed-form linear and time-dependent flows, the classical orders 4 and 1, scipy's RK45 at a
    tight tolerance, and agreement with the oracle's own Dopri5 on the CNF transform of a random network."""
    import scipy.linalg as sla
    from scipy.integrate import solve_ivp
    rng = np.random.default_rng(3)
    n, B = 4, 5
    A = rng.standard_normal((n, n)) * 0.7
    y0 = rng.standard_normal((B, n))
    lin = lambda y, t: y @ A.T
    exact = y0 @ sla.expm(A).T
    e_rk4 = [np.abs(ode.odeint_fixed(lin, y0, np.array([0.0, 1.0]), "rk4", N)[-1] - exact).max() for N in (8, 16, 32)]
    e_eul = [np.abs(ode.odeint_fixed(lin, y0, np.array([0.0, 1.0]), "euler", N)[-1] - exact).max() for N in (64, 128, 256)]
    assert 13 < e_rk4[0] / e_rk4[1] < 19 and 13 < e_rk4[1] / e_rk4[2] < 19 and e_rk4[2] < 1e-6          # order 4
    assert 1.8 < e_eul[0] / e_eul[1] < 2.2 and 1.8 < e_eul[1] / e_eul[2] < 2.2                            # order 1
    # per-chain times and a time-dependent field; intermediate output times on step boundaries (n_ts = 5 of the 4-mode example)
    ys = ode.odeint_fixed(lambda y, t: y * np.cos(t)[:, None], y0, np.linspace(0, 1, 5), "rk4", 64)
    for j, tt in enumerate(np.linspace(0, 1, 5)):
        np.testing.assert_allclose(ys[j], y0 * np.exp(np.sin(tt)), rtol=1e-8)
    for bidx in range(B):
        sol = solve_ivp(lambda t, y: A @ y, (0, 1), y0[bidx], method="RK45", rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(ode.odeint_fixed(lin, y0, np.array([0.0, 1.0]), "rk4", 64)[-1][bidx], sol.y[:, -1], rtol=1e-7, atol=1e-8)
    st = {}
    ode.odeint_fixed(lin, y0, np.array([0.0, 1.0]), "rk4", 10, st)
    assert (st["n_attempted"] == 10).all() and st["n_evals"] == 40
    # the CNF transform with its Hutchinson log-det: fixed RK4 converges to what the adaptive solver gives at a tight tolerance
    from oracle import loop
    args = loop.default_args(example="phi-four", dim=16, num_chain=6, hutchs=True, fourier_dim=8, hidden_x=[16, 16], hidden_t=[16, 16], hidden_xt=[16, 16])
    dist = targets.PhiFour(16)
    k, model, state, lr_fn, _, _ = loop.setup(dist, args)
    prm = [dict(kernel=rng.standard_normal(p["kernel"].shape) / np.sqrt(p["kernel"].shape[0]), bias=rng.standard_normal(p["bias"].shape) * 0.05) for p in state.params]
    prm[4]["kernel"] *= 1e-2; prm[4]["bias"] *= 1e-2                     # the gate of grad log pi (cubic in x): a tame field
    keys = prng.split(prng.PRNGKey(4), 6)
    u = rng.standard_normal((6, 16))
    xa, la = ode.transform_and_logdet(model, prm, keys, u, True, 1e-10, 1e-10, 10000)
    ex, el = {}, {}
    for N in (16, 256):
        xf, lf = ode.transform_and_logdet(model, prm, keys, u, True, 0, 0, 0, fixed=("rk4", N))
        ex[N], el[N] = np.abs(xf - xa).max(), np.abs(lf - la).max()
    # a strong field (points move by ~6, log-det ~6, thousands of adaptive steps at 1e-10) with ReLU kinks: the log-det integrand is
    # only piecewise smooth, so its error falls like h rather than h^4 (measured: x 3.8e-3 -> 8e-5, log-det 4e-2 -> 2e-3)
    assert ex[256] < 1e-3 and el[256] < 5e-3 and ex[256] < ex[16] / 5 and el[256] < el[16] / 5
    xe1, le1 = ode.transform_and_logdet(model, prm, keys, u, True, 0, 0, 0, fixed=("euler", 128))
    xe2, le2 = ode.transform_and_logdet(model, prm, keys, u, True, 0, 0, 0, fixed=("euler", 256))
    assert 1.5 < np.abs(xe1 - xa).max() / np.abs(xe2 - xa).max() < 2.5                                   # order 1
    ub, lb = ode.inverse_and_logdet(model, prm, keys, xf, True, 0, 0, 0, fixed=("rk4", 256))
    np.testing.assert_allclose(ub, u, atol=1e-3)                          # inverse(transform(u)) = u


def test_mala_as_written_matches_literal_transcription():
    d, B, eps = 8, 6, 0.05
    dist = targets.PhiFour(d)
    vg = targets.Tempered(dist, 0.7).value_and_grad
    x = dist.initialize_model(prng.PRNGKey(1), B)
    st = mala.init(x, vg)
    keys = prng.split(prng.PRNGKey(2), B)
    new, info, u = mala.kernel(keys, st, vg, eps)
    for b in range(B):                                   # literal mala.py:86-118 for one chain
        k_int, k_rmh = prng.split(keys[b], 2)
        noise = prng.normal(k_int, (d,))
        xb, lpb, gb = st.position[b], st.logdensity[b], st.logdensity_grad[b]
        xn = xb + eps * gb + np.sqrt(2 * eps) * noise
        lpn, gn = vg(xn[None]); lpn, gn = lpn[0], gn[0]
        E = lambda xa, lpa, ga, xb_: -lpa + 0.25 / eps * ((xb_ - xa - eps * ga) ** 2).sum()
        new_E, prev_E = E(xb, lpb, gb, xn), E(xn, lpn, gn, xb)
        p = min(1.0, np.exp(prev_E - new_E))
        acc = prng.uniform(k_rmh, ()) < p
        assert abs(info.acceptance_rate[b] - p) < 1e-12 and info.is_accepted[b] == acc
        np.testing.assert_allclose(info.proposed_position[b], xn)
        np.testing.assert_allclose(new.position[b], xn if acc else xb)
        np.testing.assert_allclose(new.logdensity[b], lpn if acc else lpb)


def test_mala_as_written_is_inverse_ratio():
    # N(0,1), eps=0.9: stationary variance as written >> 1; textbook == 1
    class N01:
        def value_and_grad(self, x):
            return -0.5 * (x * x).sum(1), -x
    vg = N01().value_and_grad
    out = {}
    for tb in (False, True):
        st = mala.init(np.zeros((4000, 1)), vg)
        key = prng.PRNGKey(9)
        xs = []
        for it in range(150):
            key, sub = prng.split(key, 2)
            st, _, _ = mala.kernel(prng.split(sub, 4000), st, vg, 0.9, textbook=tb)
            if it >= 50:
                xs.append(st.position.copy())
        out[tb] = np.var(np.concatenate(xs))
    assert abs(out[True] - 1.0) < 0.05 and out[False] > 3.0


def test_beta_bisection():
    rng = np.random.default_rng(3)
    ll = -1000 + 300 * rng.standard_normal(512)
    b = flow.beta_fn(0.0, ll, 0.95, 512)
    assert 0 < b < 1 and abs(flow.ess_zero(b, 0.0, ll, 0.95, 512)) < 1e-2
    b2 = flow.beta_fn(b, ll, 0.95, 512)
    assert b < b2 < 1
    # ESS already above target at beta=1: iterate runs to the upper end (jaxopt row)
    ll_flat = -1.0 + 1e-3 * rng.standard_normal(512)
    b3 = flow.beta_fn(0.3, ll_flat, 0.95, 512)
    assert abs(b3 - (1 - 0.7 * 2.0 ** -30)) < 1e-12
    assert flow.beta_fn(b3, ll_flat, 0.95, 512) >= b3
