"""GPU parity of the FIXED-STEP mode of the CNF solver (mfm_config.ode_method / ode_steps; mfm_amd/csrc/ode_fixed.hip): classical RK4 and
forward Euler on N equal steps, the "RK4/Euler ODE integrator" of BASELINE.json's north star.  The reference itself integrates with the
adaptive Dopri5 (exe_flow_matching.py:345-349), so the oracle here is oracle/ode.py: odeint_fixed (float64; pinned by closed forms, scipy
and the oracle's own Dopri5 in tests/test_oracle_optim_ode_mala.py): same steps, same stages, float32 against float64 -- no controller
decisions between the two sides, so the tolerances are those of the prescribed-step replay tests."""
import numpy as np
import pytest

from oracle import flow, mala, ode, prng, targets

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def _setup(d, B, method, steps, out_scale=0.5, gate=1e-3):
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, ode_method=method, ode_steps=steps)
    params = gu.rand_params(model, seed=9, out_scale=out_scale)
    params[4]["kernel"] *= gate; params[4]["bias"] *= gate          # (as in tests/test_gpu_ode.py: a tame gate of the clipped grad log pi)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    return args, dist, model, params, ctx


@pytest.mark.parametrize("d,method,steps", [(256, "rk4", 24), (256, "rk4", 7), (256, "euler", 23), (128, "rk4", 16), (64, "rk4", 16), (64, "euler", 10)])
def test_fixed_step_transform_and_inverse_match_oracle(d, method, steps):
    """Both directions, per-chain probe keys, odd and even step counts (an RK4 time batch serves two steps, an Euler one five: the last
    batch of 7 / 23 steps is a partial one), both tile widths and a lattice zero-padded to one (d = 64)."""
    import torch
    B = 32
    args, dist, model, params, ctx = _setup(d, B, method, steps)
    x32 = dist.init_params.astype(np.float32)
    keys = prng.split(prng.PRNGKey(21), B)
    for direction, fn in ((1, ode.transform_and_logdet), (-1, ode.inverse_and_logdet)):
        y_o, l_o = fn(model, params, keys, x32.astype(np.float64), True, 0, 0, 0, fixed=(method, steps))
        out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.ode_transform(direction, _dev(x32), out, ldj, keys=_dev(keys.astype(np.uint32).view(np.int32)), nsteps=ns)
        y, l = out.cpu().numpy(), ldj.cpu().numpy()
        ey, el, ls = np.abs(y - y_o).max(), np.abs(l - l_o), max(1.0, np.abs(l_o).max())
        print(f"fixed {method} x {steps}, d = {d}, direction {direction:+d}: |dy| {ey:.2e} (|y - x| {np.abs(y_o - x32).max():.2f}), |dl| q90 {np.quantile(el, 0.9):.2e} max {el.max():.2e} (|l| {np.abs(l_o).max():.2f})")
        assert np.abs(y_o - x32).max() > 0.05 and np.abs(l_o).max() > 0.05                      # a non-trivial flow
        # measured: |dy| 2.4e-7 .. 5.4e-7; log-det 1.2e-6 .. 1.6e-6 of |l| ~ 3 at d = 256.  For d <= 128 the reference does not clip grad log pi
        # (exe_flow_matching.py:351), whose Hessian-vector term then multiplies every ReLU kink a stage input lands on within float32
        # rounding: an isolated chain at 4.8e-4 (the prescribed-step replay tests bound the same events by quantiles)
        assert ey < 3e-5 * max(1.0, np.abs(y_o).max()) and np.quantile(el, 0.9) < 2e-5 * ls and el.max() < (1e-3 if d <= 128 else 5e-5) * ls
        np.testing.assert_array_equal(ns.cpu().numpy(), steps)
    ctx.close()


def test_fixed_step_rk4_approaches_the_adaptive_solution():
    """The stated distance to the reference's integrator (both on the GPU, same probes; two discretisations of one flow, not the same
    arithmetic): on this tame field RK4 x 64 ends 4.9e-4 from what Dopri5 at rtol = atol = 1e-5 gives in 35 attempted steps, and its log-det
    6.5e-2 (2.3 % of |l| ~ 2.9) -- the Hutchinson integrand jumps at every ReLU / clip kink, so a fixed grid converges to it like h, not h^4
    (oracle: tests/test_oracle_optim_ode_mala.py)."""
    import torch
    from tests import gpu_util as gu
    B, d = 32, 256
    args, dist, model, params, ctx = _setup(d, B, "rk4", 64)
    args_a, _, k, _, _ = gu.phi4_setup(d=d, B=B)
    ctx_a = gu.make_ctx(dist, args_a, fourier=model.f, params=params)
    x32 = dist.init_params.astype(np.float32)
    keys = _dev(prng.split(prng.PRNGKey(21), B).astype(np.uint32).view(np.int32))
    res = []
    for c in (ctx, ctx_a):
        out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
        c.ode_transform(1, _dev(x32), out, ldj, keys=keys, nsteps=ns)
        res.append((out.cpu().numpy(), ldj.cpu().numpy(), ns.float().mean().item()))
    (yf, lf, nf), (ya, la, na) = res
    print(f"RK4 x 64 vs Dopri5 (mean {na:.1f} attempted steps): |dy| {np.abs(yf - ya).max():.2e}, |dl| {np.abs(lf - la).max():.2e} (|l| {np.abs(la).max():.2f})")
    assert np.abs(yf - ya).max() < 2e-3 * max(1.0, np.abs(ya).max()) and np.abs(lf - la).max() < 5e-2 * max(1.0, np.abs(la).max())
    ctx.close(); ctx_a.close()


@pytest.mark.parametrize("d,method,steps", [(256, "rk4", 20), (256, "euler", 25), (64, "rk4", 12)])
def test_fixed_step_flow_mh_step_matches_oracle(d, method, steps):
    """exe_flow_matching.py:264-278 with both solves on fixed steps: proposal, unclipped acceptance ratio, decisions, accepted states."""
    import torch
    from mfm_amd import _lib
    B = 32
    args, dist, model, params, ctx = _setup(d, B, method, steps, out_scale=0.05)
    beta = 1e-3                            # an early annealing temperature: the random-walk proposals have acceptance ratios of order one
    vg = targets.Tempered(dist, beta).value_and_grad
    x32 = dist.init_params.astype(np.float32)
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    st = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    key = prng.PRNGKey(31)
    so = {}
    new, info = flow.rwmh_step(prng.split(key, B), st, vg, model, params, args, so)
    acc = torch.empty(B, device="cuda"); isacc = torch.empty(B, dtype=torch.uint8, device="cuda")
    prop = torch.empty(B, d, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, key, beta, pos, logp, grad, acc, isacc, prop, ns)
    p = prop.cpu().numpy()
    e_p = np.abs(p - info.proposed_position).max()
    with np.errstate(divide="ignore"):
        la_g, la_o = np.log(acc.cpu().numpy().astype(np.float64)), so["log_alpha"]
    fin = np.isfinite(la_g) & (la_o > -80)
    dla = np.abs(la_g[fin] - la_o[fin])
    gn = np.linalg.norm(vg(info.proposed_position)[1], axis=1)
    print(f"fixed {method} x {steps} flow step, d = {d}: |dx'| {e_p:.2e}, |d log alpha| median {np.median(dla):.2e} max {dla.max():.2e} "
          f"(first-order bound |grad log pi(x')| |dx'| ~ {np.median(gn) * e_p:.1e}); accepted gpu {int(isacc.sum().item())} oracle {int(info.is_accepted.sum())}")
    assert e_p < 3e-5 * max(1.0, np.abs(info.proposed_position).max())
    assert fin.sum() >= B // 2 and (dla <= 3.0 * gn[fin] * np.linalg.norm(p - info.proposed_position, axis=1)[fin] + 2e-3).all()
    same = isacc.cpu().numpy().astype(bool) == info.is_accepted
    assert same.mean() > 0.9
    np.testing.assert_allclose(pos.cpu().numpy()[same], new.position[same], atol=3e-5 * max(1.0, np.abs(new.position).max()))
    np.testing.assert_array_equal(ns.cpu().numpy(), 2 * steps)
    ctx.close()


def test_fixed_step_loop_matches_oracle():
    """`--ode_method rk4 --ode_steps 16` through run(): MALA / flow schedule, training, annealing and the final sampling (whose
    transform runs in the same mode) against the oracle loop with the same integrator."""
    from mfm_amd import distributions as D, exe_flow_matching as E
    from oracle import loop
    from tests import gpu_util as gu
    common = dict(example="phi-four", dim=256, num_chain=64, learning_iter=9, mcmc_per_flow_steps=3.0, hutchs=True, seed=1024, eval_iter=1,
                  step_size=1e-4, ode_method="rk4", ode_steps=16)
    out = loop.run(targets.PhiFour(256), loop.default_args(**common))
    res, res_, ex = E.run(D.PhiFour(256), loop.default_args(**common), None, log_every=1000, return_extras=True)
    tr, m = out["trace"], ex["metrics"]
    np.testing.assert_allclose(m[:3, 0], tr["loss"][:3], rtol=1e-6)
    np.testing.assert_allclose(m[:, 0], tr["loss"], rtol=1e-2)
    np.testing.assert_allclose(ex["betas"], tr["beta"], rtol=2e-3)
    assert ex["engine"].ctx.counters()["dopri_attempts"] == 64 * (2 * 2 * 16 + 16)      # two flow steps of two solves + the final transform
    g = ex["states"].position.cpu().numpy().astype(np.float64)
    dmax = np.abs(g - out["states"].position).max(1)
    assert (dmax > 0.05).sum() <= 4 and dmax[dmax <= 0.05].max() < 2e-2
    assert np.isfinite(res[0])
    ex["engine"].close()


def test_fixed_step_mode_declines_what_it_is_not_built_for():
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    args, dist, k, model, state = gu.gmm4_setup(B=32, hutchs=True, ode_method="rk4", ode_steps=8)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=gu.rand_params(model, seed=1))
    x = _dev(np.zeros((32, 2), np.float32)); out = torch.empty(32, 2, device="cuda"); ldj = torch.empty(32, device="cuda")
    with pytest.raises(_lib.MfmError, match="fixed-step mode"):
        ctx.ode_transform(1, x, out, ldj, key=(0, 1))
    ctx.close()
    with pytest.raises(_lib.MfmError, match="ode_steps"):
        _lib.Context(dim=64, n_chain_local=16, ode_method=_lib.ODE_METHODS["rk4"], ode_steps=0)
    with pytest.raises(_lib.MfmError, match="ode_method"):
        _lib.Context(dim=64, n_chain_local=16, ode_method=7, ode_steps=4)
