"""CPU pins of the round-2 oracle pieces: the prescribed-step-sequence mode of the Dormand-Prince restatement (parity
instrumentation: oracle/ode.py `replay`, `round32`) and the end-of-run block N1 (oracle/loop.py `final_sampling`,
exe_flow_matching.py:453-459)."""
import numpy as np

from oracle import flow, loop, mala, ode, prng, targets
from tests import gpu_util as gu


def _setup(d=16, B=8, hidden=16, F=8):
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F)
    params = gu.rand_params(model, seed=9, out_scale=1.0)
    params[4]["kernel"] *= 1e-3; params[4]["bias"] *= 1e-3
    return args, dist, k, model, params


def test_replaying_a_solve_on_its_own_step_sequence_reproduces_it():
    args, dist, k, model, params = _setup()
    x = dist.init_params
    keys = prng.split(prng.PRNGKey(3), x.shape[0])
    o = (True, args.rtol, args.atol, args.mxstep)
    for fn in (ode.transform_and_logdet, ode.inverse_and_logdet):
        st = {}
        y, l = fn(model, params, keys, x, *o, stats=st)
        n = st["n_attempted"]
        assert st["dt_seq"].shape == (x.shape[0], n.max() + 1) and st["acc_seq"].shape == (x.shape[0], n.max())
        for b in range(x.shape[0]):                                   # a chain's record ends with its last attempt
            assert (st["dt_seq"][b, :n[b]] > 0).all() and (st["dt_seq"][b, n[b]:] == 0).all()
            assert st["acc_seq"][b, n[b] - 1] and not st["acc_seq"][b, n[b]:].any()      # the last attempt is the accepted one that reaches t = 1
            np.testing.assert_array_equal(st["dt_own"][b, :n[b]], st["dt_seq"][b, :n[b]])   # without replay the controller's steps ARE the steps
        st2 = {}
        y2, l2 = fn(model, params, keys, x, *o, stats=st2, replay=dict(dt=st["dt_seq"], acc=st["acc_seq"]))
        np.testing.assert_array_equal(st2["n_attempted"], n)
        np.testing.assert_array_equal(y2, y); np.testing.assert_array_equal(l2, l)           # bit for bit: same arithmetic, same steps
        np.testing.assert_array_equal(st2["ratio_seq"], st["ratio_seq"])
        # float32-rounded steps (what the kernels are given): same attempt counts, results within the rounding of the steps
        dt32 = st["dt_seq"].astype(np.float32).astype(np.float64)
        st3 = {}
        y3, l3 = fn(model, params, keys, x, *o, stats=st3, replay=dict(dt=dt32, acc=st["acc_seq"]))
        np.testing.assert_array_equal(st3["n_attempted"], n)
        assert np.abs(y3 - y).max() < 1e-6 and np.abs(l3 - l).max() < 1e-5
        # the float32-rounding yardstick changes a well-conditioned solve at the rounding level only
        y4, l4 = fn(model, params, keys, x, *o, replay=dict(dt=dt32, acc=st["acc_seq"]), round32=True)
        assert 0 < np.abs(y4 - y3).max() < 1e-5


def test_flow_step_replay_threads_both_solves():
    args, dist, k, model, params = _setup()
    B = dist.init_params.shape[0]
    vg = targets.Tempered(dist, 0.8).value_and_grad
    st0 = mala.init(dist.init_params, vg)
    keys = prng.split(prng.PRNGKey(5), B)
    nat = {}
    new, info = flow.rwmh_step(keys, st0, vg, model, params, args, nat)
    rp = dict(inv=dict(dt=nat["inv"]["dt_seq"], acc=nat["inv"]["acc_seq"]), fwd=dict(dt=nat["fwd"]["dt_seq"], acc=nat["fwd"]["acc_seq"]))
    so = {}
    new2, info2 = flow.rwmh_step(keys, st0, vg, model, params, args, so, replay=rp)
    np.testing.assert_array_equal(new2.position, new.position)
    np.testing.assert_array_equal(info2.acceptance_rate, info.acceptance_rate)
    np.testing.assert_array_equal(so["n_att_inv"], nat["n_att_inv"]); np.testing.assert_array_equal(so["n_att_fwd"], nat["n_att_fwd"])
    with np.errstate(divide="ignore"):
        np.testing.assert_allclose(np.log(info.acceptance_rate), so["log_alpha"], rtol=1e-12, atol=1e-12)


def test_final_sampling_restates_the_importance_resampling():
    """exe_flow_matching.py:453-459: at the flax zero-init the flow is the identity, so x = u, vol = 0 and the weights are
    pi(u) / q0(u); the resampled set is drawn with replacement from the flow samples by inverse-CDF search."""
    args, dist, k, model, state = gu.gmm4_setup(B=32, hidden=16, F=8, hutchs=False, eval_iter=4)
    x, ex, info = loop.final_sampling(model, state.params, dist, args, k["gen"])
    n = args.eval_iter * args.num_chain
    assert x.shape == ex.shape == (n, 2)
    np.testing.assert_allclose(x, info["u"], atol=1e-12); assert np.abs(info["vols"]).max() < 1e-12          # identity flow (up to the rounding of t0 + sum dt)
    ref = targets.IndepGaussian(2)
    np.testing.assert_allclose(info["log_weights"], dist.logprob(x) - ref.logprob(info["u"]), rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(ex, x[info["idx"]])
    # the categorical draw: counts follow the normalised weights (chi-square-free check: heavy samples are drawn, weightless never)
    w = info["weights"] / info["weights"].sum()
    cnt = np.bincount(info["idx"], minlength=n)
    assert cnt[w < 1e-12].sum() == 0 and cnt[np.argmax(w)] >= 1
    key_hutch, key_choice = prng.split(k["gen"])
    np.testing.assert_array_equal(info["idx"], np.minimum(prng.choice_p(key_choice, info["weights"], (n,)), n - 1))


def test_gmm16_fixture_matches_the_cli_recipe():
    import os
    from mfm_amd.multi_modal import gmm16_parameters
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gmm16_params.npz"))
    m, c, w = gmm16_parameters()
    np.testing.assert_array_equal(m, g["modes"]); np.testing.assert_array_equal(c, g["covs"]); np.testing.assert_array_equal(w, g["weights"])
