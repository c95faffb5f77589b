"""oracle/hmc.py (build-side mode: the HMC step BASELINE.json's north star names beside MALA; nothing of the reference to pin it to):
checked against what an HMC kernel must satisfy -- time reversibility of the velocity-Verlet trajectory, energy error O(eps^2),
the exact acceptance algebra on given draws, and the stationary moments of a Gaussian target.  CPU only."""
import numpy as np

from oracle import hmc, mala, prng, targets


def _gauss_vg(var):
    def vg(x):
        return -0.5 * (x * x).sum(1) / var, -x / var
    return vg


def test_trajectory_is_reversible_and_its_energy_error_is_second_order():
    dist = targets.PhiFour(16)
    vg = targets.Tempered(dist, 1.0).value_and_grad
    rng = np.random.default_rng(0)
    x, p = rng.uniform(-1, 1, (8, 16)), rng.standard_normal((8, 16))
    keys = prng.split(prng.PRNGKey(1), 8)
    errs = []
    for eps in (2e-3, 1e-3, 5e-4):
        st0 = mala.init(x, vg)
        st, info, _ = hmc.kernel(keys, st0, vg, eps, int(round(0.02 / eps)), momentum=p)
        errs.append(np.abs(info.energy_delta).max())
        # integrate back from the proposal with the momentum flipped: the start is recovered
        lpn, gn = vg(info.proposed_position)
        # momentum at the end of the forward trajectory, recomputed step by step
        xn, pn, g = x.copy(), p.copy(), st0.logdensity_grad
        for _ in range(int(round(0.02 / eps))):
            pn = pn + 0.5 * eps * g; xn = xn + eps * pn; _, g = vg(xn); pn = pn + 0.5 * eps * g
        np.testing.assert_allclose(xn, info.proposed_position, rtol=0, atol=1e-14)
        back, binfo, _ = hmc.kernel(keys, mala.MALAState(xn, lpn, gn), vg, eps, int(round(0.02 / eps)), momentum=-pn)
        np.testing.assert_allclose(binfo.proposed_position, x, rtol=0, atol=1e-10)
    assert errs[0] / errs[1] > 3.0 and errs[1] / errs[2] > 3.0, errs          # halving eps quarters the energy error


def test_acceptance_algebra_on_given_draws():
    vg = _gauss_vg(1.0)
    x = np.array([[0.3, -1.2], [2.0, 0.1]]); p = np.array([[1.0, 0.5], [-0.2, 3.0]])
    keys = prng.split(prng.PRNGKey(3), 2)
    st0 = mala.init(x, vg)
    st, info, u = hmc.kernel(keys, st0, vg, 0.3, 4, momentum=p)
    xn, pn = x.copy(), p.copy()
    for _ in range(4):
        pn = pn - 0.5 * 0.3 * xn; xn = xn + 0.3 * pn; pn = pn - 0.5 * 0.3 * xn
    h0 = 0.5 * (x * x).sum(1) + 0.5 * (p * p).sum(1); h1 = 0.5 * (xn * xn).sum(1) + 0.5 * (pn * pn).sum(1)
    np.testing.assert_allclose(info.energy_delta, h0 - h1, atol=1e-13)
    np.testing.assert_allclose(info.acceptance_rate, np.minimum(1.0, np.exp(h0 - h1)), atol=1e-13)
    assert (info.is_accepted == (u < info.acceptance_rate)).all()
    np.testing.assert_array_equal(st.position, np.where(info.is_accepted[:, None], xn, x))


def test_gaussian_target_keeps_its_variance():
    """2048 chains started in N(0, var I) stay there: mean ~ 0, variance ~ var after 30 steps (the textbook rule; the vendored MALA kernel's
    inverted ratio would inflate it: SURVEY.md Q1)."""
    var, d, B = 2.0, 4, 2048
    vg = _gauss_vg(var)
    x = np.sqrt(var) * np.random.default_rng(5).standard_normal((B, d))
    st = mala.init(x, vg)
    key = prng.PRNGKey(9)
    accs = []
    for _ in range(30):
        key, k = prng.split(key, 2)
        st, info, _ = hmc.kernel(prng.split(k, B), st, vg, 0.4, 5)
        accs.append(info.acceptance_rate.mean())
    assert np.mean(accs) > 0.9
    v = st.position.var(0)
    assert np.abs(st.position.mean(0)).max() < 0.12 and np.abs(v / var - 1.0).max() < 0.12, (st.position.mean(0), v)
