"""Flow-based Metropolis-Hastings steps, the MALA/flow scheduler and the beta annealing.

ORACLE (test infrastructure; see oracle/__init__.py).  Follows
``exe_flow_matching.py:246-260`` (independent MH), ``:264-278`` (random-walk MH in latent
space, the default), ``:280-296`` (conditional importance sampling), ``:300-316``
(scheduler / init_fn) and ``:391-417`` (beta_fn / beta_gen, with ``jaxopt.Bisection``
0.8.3 restated from its published algorithm -- PARITY UNPINNED).

Quirk kept: flow-MH acceptance probabilities are NOT clipped to 1 and are
what the loop logs as the acceptance rate.
"""
import numpy as np

from . import mala, ode, prng
from .mala import MALAInfo, MALAState
from .targets import REF_VARS, IndepGaussian, Tempered


def _accept(keys_acc, a, prev, prop):
    u = prng.uniform_rows(keys_acc)
    with np.errstate(invalid="ignore"):
        acc = u <= a                                   # uniform <= NaN is False -> reject
    m = acc[:, None]
    x, lp, g = prev
    xn, lpn, gn = prop
    state = MALAState(np.where(m, xn, x), np.where(acc, lpn, lp), np.where(m, gn, g))
    info = MALAInfo(a, acc, xn, np.zeros_like(a))
    return state, info


def fixed_mode(args):
    """``(method, nsteps)`` of the build-side fixed-step mode (``--ode_method rk4|euler --ode_steps N``), or None: the reference's Dopri5."""
    n = int(getattr(args, "ode_steps", 0) or 0)
    return (getattr(args, "ode_method", "rk4"), n) if n > 0 else None


def rwmh_step(keys, prev, value_and_grad, model, params, args, stats=None, replay=None, round32=False):
    """``exe_flow_matching.py:264-278``.  ``replay = dict(inv=..., fwd=...)``: prescribed step sequences of the two
    solves (parity instrumentation, see ``ode.odeint``)."""
    B, d = prev.position.shape
    kk = prng.split_rows(keys, 4)                      # :265 key_gen, key_acc, key_hutch1, key_hutch2
    o = dict(hutch=args.hutchs, rtol=args.rtol, atol=args.atol, mxstep=args.mxstep, n_ts=args.n_ts, fixed=fixed_mode(args))
    st_inv = {} if stats is not None else None
    st_fwd = {} if stats is not None else None
    rp = replay or {}
    o["round32"] = round32                             # test yardstick only, see ode._augmented
    u0, vol0 = ode.inverse_and_logdet(model, params, kk[:, 3], prev.position, stats=st_inv, replay=rp.get("inv"), **o)   # :267
    up = u0 + (2.38 / np.sqrt(d)) * prng.normal_rows(kk[:, 0], d)                                  # :262,268
    xp, volp = ode.transform_and_logdet(model, params, kk[:, 2], up, stats=st_fwd, replay=rp.get("fwd"), **o)            # :269
    lpn, gn = value_and_grad(xp)                                                                    # :270
    with np.errstate(over="ignore", invalid="ignore"):
        a = np.exp(lpn - volp - prev.logdensity - vol0)                                            # :271-274
    if stats is not None:
        stats["n_att_inv"], stats["n_att_fwd"] = st_inv["n_attempted"], st_fwd["n_attempted"]
        stats["u0"], stats["vol0"], stats["up"], stats["volp"] = u0, vol0, up, volp
        stats["inv"], stats["fwd"], stats["log_alpha"] = st_inv, st_fwd, lpn - volp - prev.logdensity - vol0
    return _accept(kk[:, 1], a, prev, (xp, lpn, gn))                                               # :275-278


def imh_step(keys, prev, value_and_grad, model, params, args, stats=None, replay=None):
    """``exe_flow_matching.py:246-260``.  ``stats`` / ``replay`` as in ``rwmh_step`` (parity instrumentation)."""
    B, d = prev.position.shape
    ref = IndepGaussian(d, var=REF_VARS[getattr(args, "ref_dist", "stdgauss")])
    kk = prng.split_rows(keys, 4)                      # :247
    o = dict(hutch=args.hutchs, rtol=args.rtol, atol=args.atol, mxstep=args.mxstep, n_ts=args.n_ts, fixed=fixed_mode(args))
    st_inv = {} if stats is not None else None
    st_fwd = {} if stats is not None else None
    rp = replay or {}
    up = ref.sample_model_rows(kk[:, 0])                                                           # :249
    xp, volp = ode.transform_and_logdet(model, params, kk[:, 2], up, stats=st_fwd, replay=rp.get("fwd"), **o)            # :250
    u0, vol0 = ode.inverse_and_logdet(model, params, kk[:, 3], prev.position, stats=st_inv, replay=rp.get("inv"), **o)   # :251
    lpn, gn = value_and_grad(xp)                                                                    # :252
    la = lpn - ref.logprob(up) - volp + ref.logprob(u0) - vol0 - prev.logdensity                  # :253-256
    with np.errstate(over="ignore", invalid="ignore"):
        a = np.exp(la)
    if stats is not None:
        stats["n_att_inv"], stats["n_att_fwd"] = st_inv["n_attempted"], st_fwd["n_attempted"]
        stats["u0"], stats["vol0"], stats["up"], stats["volp"] = u0, vol0, up, volp
        stats["inv"], stats["fwd"], stats["log_alpha"] = st_inv, st_fwd, la
    return _accept(kk[:, 1], a, prev, (xp, lpn, gn))


def cis_step(keys, prev, value_and_grad, model, params, args, stats=None):
    """``exe_flow_matching.py:280-296``: conditional importance sampling with ``args.num_importance_samples`` fresh
    flow samples per chain.  Quirk kept: an accepted state carries the STALE ``prev_state.logdensity_grad`` (``:295``)."""
    B, d = prev.position.shape
    n_is = int(args.num_importance_samples)
    ref = IndepGaussian(d, var=REF_VARS[getattr(args, "ref_dist", "stdgauss")])
    kk = prng.split_rows(keys, 4)                      # :281 key_sample, key_hutch_prev, key_hutch, key_choice
    o = dict(hutch=args.hutchs, rtol=args.rtol, atol=args.atol, mxstep=args.mxstep, n_ts=args.n_ts, fixed=fixed_mode(args))
    u0, vol0 = ode.inverse_and_logdet(model, params, kk[:, 1], prev.position, **o)                 # :282
    with np.errstate(over="ignore", invalid="ignore"):
        w_prev = np.exp(prev.logdensity - ref.logprob(u0) - vol0)                                  # :283
    ks = np.stack([prng.split(kk[b, 0], n_is) for b in range(B)]).reshape(B * n_is, 2)           # :284
    kh = np.stack([prng.split(kk[b, 2], n_is) for b in range(B)]).reshape(B * n_is, 2)           # :286
    refs = ref.sample_model_rows(ks)                                                               # :285
    xs, vols = ode.transform_and_logdet(model, params, kh, refs, **o)                              # :287
    lps, _ = value_and_grad(xs)                                                                    # :288
    with np.errstate(over="ignore", invalid="ignore"):
        w = np.exp(lps - ref.logprob(refs) - vols).reshape(B, n_is)                                # :289
    allw = np.concatenate([w_prev[:, None], w], axis=1)
    norm = allw / allw.sum(1, keepdims=True)                                                       # :290-291
    choice = np.array([prng.choice_p(kk[b, 3], norm[b]) for b in range(B)])                        # :292
    choice = np.minimum(choice, n_is)
    acc = choice != 0
    pick = np.maximum(choice - 1, 0) + np.arange(B) * n_is
    m = acc[:, None]
    wsel = norm[np.arange(B), choice]
    state = MALAState(np.where(m, xs[pick], prev.position), np.where(acc, lps[pick], prev.logdensity), prev.logdensity_grad)   # :293-295
    info = MALAInfo(wsel, acc, np.where(m, xs[pick], prev.position), wsel)
    if stats is not None:
        stats.update(u0=u0, vol0=vol0, refs=refs, xs=xs, vols=vols, lps=lps, norm=norm, choice=choice)
    return state, info


def train_data_generator(key, states, count, model, params, dist, args, beta=1.0, n_total=None,
                         start=0, stats=None):
    """``exe_flow_matching.py:300-314``; chain b uses ``split(key, B_total)[b]`` (``:303``)."""
    B = states.position.shape[0]
    n_total = B if n_total is None else n_total
    vg = Tempered(dist, beta).value_and_grad                                                       # :301
    keys = prng.split_at(key, n_total, np.arange(start, start + B))                                # :303
    K = args.mcmc_per_flow_steps
    if 0 < K < 1:                                                                                  # :304-309
        do_flow = count % (int(1 / K) + 1) != 0
    else:
        do_flow = count % (int(K) + 1) == 0                                                        # :311
    if do_flow:
        step = cis_step if args.num_importance_samples > 0 else imh_step if args.num_importance_samples < 0 else rwmh_step   # :298
        return step(keys, states, vg, model, params, args, stats)
    if getattr(args, "mcmc_kernel", "mala") == "hmc":       # build-side mode (oracle/hmc.py), not in the reference
        from . import hmc
        st, hi, _ = hmc.kernel(keys, states, vg, args.step_size, int(args.hmc_steps))
        return st, MALAInfo(hi.acceptance_rate, hi.is_accepted, hi.proposed_position, np.zeros_like(hi.acceptance_rate))
    st, info, _ = mala.kernel(keys, states, vg, args.step_size)                                    # :313
    return st, info


def init_fn(positions, dist, beta=1.0):
    """``exe_flow_matching.py:316``."""
    return mala.init(positions, Tempered(dist, beta).value_and_grad)


def ess_zero(beta, prev_beta, logliks, alpha, n_chain):
    """``exe_flow_matching.py:393-399``."""
    logw = logliks * (beta - prev_beta)
    w = np.exp(logw - logw.max())
    w = w / w.sum()
    return 1.0 / (w * w).sum() - alpha * n_chain


def beta_fn(prev_beta, logliks, alpha, n_chain, maxiter=30, tol=1e-5):
    """``exe_flow_matching.py:391-402``: jaxopt.Bisection(lower=prev_beta, upper=1, maxiter=30,
    tol=1e-5, check_bracket=False).run().params -- the last midpoint evaluated."""
    f = lambda b: ess_zero(b, prev_beta, logliks, alpha, n_chain)
    low, high = float(prev_beta), 1.0
    fl, fh = f(low), f(high)
    sign = 1 if (fl < 0 and fh >= 0) else (-1 if (fl > 0 and fh <= 0) else 0)
    params, err, it = 0.5 * (low + high), np.inf, 0
    while err > tol and it < maxiter:
        params = 0.5 * (high + low)
        value = f(params)
        too_large = sign * value > 0
        if too_large:
            high = params
        else:
            low = params
        err = abs(value)
        it += 1
    return params


def beta_gen(beta, states, dist, alpha, n_chain):
    """``exe_flow_matching.py:410-417``."""
    if beta < 1.0:
        beta = beta_fn(beta, dist.loglik(states.position), alpha, n_chain)                          # :413
        states = init_fn(states.position, dist, beta)                                              # :415
    return beta, states
