"""Freeze small input/output vectors of the CPU oracle for every row of SURVEY.md section 8(a).

The reference cannot be run here (no jax) and ships no fixtures, so these are outputs of the ORACLE (oracle/), whose
pieces are pinned independently (tests/test_oracle_*.py).  They give the CPU suite a regression pin of the oracle and
the GPU suite fixed inputs with expected outputs.  Usage: python tools/make_golden.py  ->  tests/golden/mfm_golden.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import flow, fm, loop, mala, ode, optim, prng, targets  # noqa: E402
from tests import gpu_util as gu  # noqa: E402


def build():
    out = {}
    # ---- PRNG (jax conventions) ----
    k = prng.PRNGKey(2026)
    out["prng_split"] = prng.split(k, 4)
    out["prng_normal"] = prng.normal(k, (6,))
    out["prng_uniform"] = prng.uniform(k, (6,))
    # ---- targets: value / grad / hvp ----
    rng = np.random.default_rng(0)
    x = rng.uniform(-1.2, 1.2, (4, 64)); v = rng.standard_normal((4, 64))
    p4 = targets.PhiFour(64)
    out.update(phi4_x=x, phi4_v=v, phi4_logp=p4.logprob(x), phi4_grad=p4.grad_logprob(x), phi4_hvp=p4.hvp_logprob(x, v))
    gm = targets.GaussianMixture(8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4)
    xg = rng.uniform(-10, 10, (6, 2)); vg = rng.standard_normal((6, 2))
    out.update(gmm_x=xg, gmm_v=vg, gmm_logp=gm.logprob(xg), gmm_grad=gm.grad_logprob(xg), gmm_hvp=gm.hvp_logprob(xg, vg))
    counts = np.load(os.path.join(ROOT, "mfm_amd", "data", "pines_counts.npz"))["counts_8"]
    lg = targets.LogGaussianCoxPines(64, counts)
    xl = lg.mu + 0.5 * rng.standard_normal((3, 64)); vl = rng.standard_normal((3, 64))
    out.update(lgcp_x=xl, lgcp_v=vl, lgcp_logp=lg.logprob(xl), lgcp_grad=lg.grad_logprob(xl), lgcp_hvp=lg.hvp_logprob(xl, vl))
    # ---- one phi-four configuration for the kernels (d = 64, 32 chains, hidden 32, F = 16) ----
    args, dist, kk, model, state = gu.phi4_setup(d=64, B=32, hidden=32, F=16, learning_iter=20)
    params = gu.rand_params(model, seed=11, out_scale=0.3)
    params[4]["kernel"] *= 1e-3; params[4]["bias"] *= 1e-3
    x0 = dist.init_params.astype(np.float32).astype(np.float64)
    out.update(cfg_fourier=model.f, cfg_params=gu.flat_params(params), cfg_x0=x0)
    beta = 0.6
    vgf = targets.Tempered(dist, beta).value_and_grad
    st = mala.init(x0, vgf)
    key = prng.PRNGKey(5)
    new, info, u = mala.kernel(prng.split(key, 32), st, vgf, 1e-4)
    out.update(mala_key=key, mala_beta=beta, mala_logp0=st.logdensity, mala_grad0=st.logdensity_grad, mala_pos=new.position,
               mala_logp=new.logdensity, mala_acc=info.acceptance_rate, mala_isacc=info.is_accepted, mala_prop=info.proposed_position, mala_u=u)
    # vector field / jvp
    t = rng.uniform(0, 1, 32); z = rng.standard_normal((32, 64)).astype(np.float32).astype(np.float64)
    vf, jv = model.forward(params, x0, t.astype(np.float32).astype(np.float64), tangent=z)
    out.update(vf_t=t.astype(np.float32), vf_z=z, vf_v=vf, vf_jvp=jv)
    # flow-matching loss / grad, AdamW step
    kf = prng.PRNGKey(6)
    loss, grads = fm.loss_and_grad(model, params, kf, x0, args.sigma)
    ts = optim.TrainState(params, optim.learning_rate_fn(20, 0, args.learning_rate))
    ts.apply_gradients(grads)
    out.update(fm_key=kf, fm_loss=loss, fm_grads=gu.flat_params(grads), adam_params=gu.flat_params(ts.params))
    # Dopri5 transforms (Hutchinson), flow-MH step
    keys = prng.split(prng.PRNGKey(7), 32)
    s1 = {}
    yf, lf = ode.transform_and_logdet(model, params, keys, x0, True, args.rtol, args.atol, args.mxstep, stats=s1)
    s2 = {}
    yi, li = ode.inverse_and_logdet(model, params, keys, x0, True, args.rtol, args.atol, args.mxstep, stats=s2)
    out.update(ode_keys=keys, ode_fwd=yf, ode_fwd_ldj=lf, ode_fwd_natt=s1["n_attempted"], ode_inv=yi, ode_inv_ldj=li, ode_inv_natt=s2["n_attempted"])
    kfl = prng.PRNGKey(8)
    st1 = mala.init(x0, targets.Tempered(dist, 1.0).value_and_grad)
    nf, inf_ = flow.rwmh_step(prng.split(kfl, 32), st1, targets.Tempered(dist, 1.0).value_and_grad, model, params, args)
    with np.errstate(divide="ignore"):
        out.update(flow_key=kfl, flow_prop=inf_.proposed_position, flow_logacc=np.log(inf_.acceptance_rate), flow_isacc=inf_.is_accepted)
    # beta bisection
    ll = dist.loglik(x0)
    out.update(beta_ll=ll, beta_0=flow.beta_fn(0.0, ll, 0.95, 32))
    return out


def build_ext():
    """Second set (tests/golden/mfm_golden_ext.npz): the rows added after the first set was frozen -- the tempered-SMC pieces
    (N4), the non-default activations, the 'widegauss' reference distribution."""
    from oracle import smc
    out = {}
    rng = np.random.default_rng(1)
    # ---- SMC pieces (bblackjax/smc) ----
    ll = rng.standard_normal(256) * 40 - 300
    key = prng.PRNGKey(9)
    delta = float(np.clip(smc.ess_solver(ll, 0.95, 1.0), 0.0, 1.0))
    lw = delta * ll
    w = np.exp(lw - smc.logsumexp(lw))
    out.update(smc_ll=ll, smc_delta=delta, smc_weights=w, smc_lognorm=smc.logsumexp(lw) - np.log(256), smc_key=key,
               smc_idx=smc.systematic(key, w, 256))
    # ---- activations: vector field / JVP and loss for tanh and gelu on one small configuration ----
    for act in ("tanh", "gelu"):
        args, dist, kk, model, state = gu.phi4_setup(d=64, B=32, hidden=32, F=16, non_linearity=act)
        params = gu.rand_params(model, seed=12)
        x0 = dist.init_params.astype(np.float32).astype(np.float64)
        t = rng.uniform(0, 1, 32).astype(np.float32); z = rng.standard_normal((32, 64)).astype(np.float32)
        v, jv = model.forward(params, x0, t.astype(np.float64), tangent=z.astype(np.float64))
        kf = prng.PRNGKey(13)
        loss, grads = fm.loss_and_grad(model, params, kf, x0, args.sigma)
        out.update({f"{act}_fourier": model.f, f"{act}_params": gu.flat_params(params), f"{act}_x0": x0, f"{act}_t": t, f"{act}_z": z,
                    f"{act}_v": v, f"{act}_jvp": jv, f"{act}_key": kf, f"{act}_loss": loss, f"{act}_grads": gu.flat_params(grads)})
    # ---- widegauss reference distribution: training batch loss, independent-MH step ----
    args, dist, kk, model, state = gu.phi4_setup(d=64, B=32, hidden=32, F=16, ref_dist="widegauss")
    params = gu.rand_params(model, seed=14, out_scale=0.05)
    params[4]["kernel"] *= 1e-3; params[4]["bias"] *= 1e-3
    x0 = dist.init_params.astype(np.float32).astype(np.float64)
    kf = prng.PRNGKey(15)
    loss, grads = fm.loss_and_grad(model, params, kf, x0, args.sigma, ref_std=np.sqrt(5.0))
    vg = targets.Tempered(dist, 0.8).value_and_grad
    st = mala.init(x0, vg)
    ki = prng.PRNGKey(16)
    new, info = flow.imh_step(prng.split(ki, 32), st, vg, model, params, args)
    with np.errstate(divide="ignore"):
        out.update(wg_fourier=model.f, wg_params=gu.flat_params(params), wg_x0=x0, wg_key=kf, wg_loss=loss, wg_grads=gu.flat_params(grads),
                   wg_imh_key=ki, wg_imh_prop=info.proposed_position, wg_imh_logacc=np.log(info.acceptance_rate), wg_imh_isacc=info.is_accepted)
    return out


if __name__ == "__main__":
    for name, fn in (("mfm_golden.npz", build), ("mfm_golden_ext.npz", build_ext)):
        path = os.path.join(ROOT, "tests", "golden", name)
        if name == "mfm_golden.npz" and os.path.exists(path) and "--all" not in sys.argv:
            continue                     # the first set stays frozen unless asked for
        o = fn()
        np.savez_compressed(path, **o)
        print(path, os.path.getsize(path), "bytes,", len(o), "arrays")
