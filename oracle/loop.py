"""The MFM training loop (``exe_flow_matching.py:321-449``) on the CPU oracle.

ORACLE (test infrastructure; see oracle/__init__.py).  Key plumbing follows SURVEY.md
Appendix A line by line.  Returns the per-iteration metrics the reference sends to wandb
(``:367,442-449``) plus the final chain states, so tests can compare loss traces and
sample moments with the HIP path on the same seeds.
"""
import time
from types import SimpleNamespace

import numpy as np

from . import flow, fm, optim, prng, targets
from .vfield import VectorFieldNet


def default_args(**kw):
    """argparse defaults of ``multi_modal.py:147-220``."""
    a = dict(seed=1, dim=64, num_modes=16, example="phi-four", sigma=1e-4, fourier_dim=128,
             fourier_std=1.0, hutchs=False, ref_dist="stdgauss", cond_flow=True, ot_cond_flow=False,
             num_importance_samples=0, mcmc_per_flow_steps=10.0, num_chain=128, learning_iter=400,
             eval_iter=100, alpha=0.95, anneal_iter=200, num_anneal_temp=200, non_linearity="relu",
             hidden_x=[128, 128], hidden_t=[128, 128], hidden_xt=[128, 128], step_size=0.2,
             learning_rate=1e-3, weight_decay=1e-4, adam_beta1=0.9, adam_beta2=0.999,
             adam_epsilon=1e-8, gradient_clip=1.0, warmup_steps=0, rtol=1e-5, atol=1e-5, mxstep=1000.0,
             mcmc_kernel="mala", hmc_steps=10)      # (the last two: build-side mode, mfm_amd/multi_modal.py)
    a.update(kw)
    ns = SimpleNamespace(**a)
    ns.n_ts = 5 if ns.example == "4-mode" else 2          # exe_flow_matching.py:347
    return ns


def setup(dist, args, target_gn=None):
    """``exe_flow_matching.py:333-360``: keys, chain init, network init, train state."""
    keys = prng.split(prng.PRNGKey(args.seed), 6)                                        # :333
    k = dict(zip("target sample init dist fourier gen".split(), keys))
    dist.initialize_model(k["dist"], args.num_chain)                                     # :334
    fourier = args.fourier_std * prng.normal(k["fourier"], (args.fourier_dim,))          # :350
    model = VectorFieldNet(fourier, dist, args.hidden_x, args.hidden_t, args.hidden_xt,
                           args.non_linearity, args.gradient_clip if args.dim > 128 else None)   # :351
    params = model.init(k["init"])                                                       # :353
    lr_fn = optim.learning_rate_fn(args.learning_iter, args.warmup_steps, args.learning_rate)    # :355-359
    state = optim.TrainState(params, lr_fn, args.adam_beta1, args.adam_beta2, args.adam_epsilon,
                             args.weight_decay, args.gradient_clip)                      # :360
    real = key_loss = None
    if target_gn is not None:                                                            # :370-374
        k["gen"], key_loss = prng.split(k["target"], 2)
        real = target_gn(prng.split(k["gen"], args.eval_iter * args.num_chain))
    return k, model, state, lr_fn, real, key_loss


def run(dist, args, target_gn=None, params_override=None, beta_override=None, timer=None):
    """Hot loop ``exe_flow_matching.py:425-449``.  ``params_override`` replaces the initial
    parameters (tests use non-zero output kernels so the flow is not the identity)."""
    k, model, state, lr_fn, real, key_loss = setup(dist, args, target_gn)
    if params_override is not None:
        state.params = [{kk: v.astype(np.float32).copy() for kk, v in p.items()} for p in params_override]
    n_chain = args.num_chain
    ref_std = float(np.sqrt(targets.REF_VARS[args.ref_dist]))                            # :149 (ref_dists, :48-54)
    iter_per_temp = args.anneal_iter // args.num_anneal_temp                             # :330
    use_real_samples = args.mcmc_per_flow_steps < 0                                       # :328
    if use_real_samples:
        beta = 1.0                                                                       # :429-430
    elif beta_override is not None:
        beta = beta_override
    else:
        beta = flow.beta_fn(0.0, dist.loglik(dist.init_params), args.alpha, n_chain)     # :426
    if use_real_samples:
        states = flow.MALAState(dist.init_params, None, None)                            # :386
    else:
        states = flow.init_fn(dist.init_params, dist, beta)                              # :431
    key_sample = k["sample"]
    trace = dict(loss=[], learning_rate=[], acc_mean=[], acc_std=[], target_loss=[], beta=[], n_att=[], n_moved=[])
    t0 = time.perf_counter()
    for count in range(1, args.learning_iter + 1):                                       # :432
        key_sample, key_gn, key_step = prng.split(key_sample, 3)                         # :433
        stats = {}
        if use_real_samples:                                                             # :382-385
            states = flow.MALAState(target_gn(prng.split(key_gn, n_chain)), None, None)
            infos = flow.MALAInfo(np.full(n_chain, np.nan), None, None, None)
        else:
            before = states.position
            states, infos = flow.train_data_generator(key_gn, states, count, model, state.params, dist,
                                                      args, beta, stats=stats)           # :438
            trace["n_moved"].append(int((states.position != before).any(1).sum()))        # (test bookkeeping: chains whose proposal was accepted)
        loss, grads = fm.loss_and_grad(model, state.params, key_step, states.position, args.sigma,
                                       args.cond_flow, ref_std=ref_std)                  # :364-365
        lr = lr_fn(state.step)                                                           # :367
        state.apply_gradients(grads)                                                     # :366
        if not use_real_samples and count % iter_per_temp == 0:                          # :440-441
            beta, states = flow.beta_gen(beta, states, dist, args.alpha, n_chain)
        trace["loss"].append(loss); trace["learning_rate"].append(lr)
        trace["acc_mean"].append(infos.acceptance_rate.mean())                           # :442
        trace["acc_std"].append(infos.acceptance_rate.std())                             # :443
        trace["beta"].append(beta)
        if "n_att_inv" in stats:
            trace["n_att"].append((stats["n_att_inv"].mean(), stats["n_att_fwd"].mean()))
        if real is not None:                                                             # :444-446
            tl, _ = fm.loss_and_grad(model, state.params, key_loss, real, args.sigma, args.cond_flow,
                                     need_grad=False, ref_std=ref_std)
            trace["target_loss"].append(tl)
        if timer is not None:
            timer(count, time.perf_counter() - t0)
    trace["train_time"] = time.perf_counter() - t0
    return dict(trace=trace, states=states, state=state, model=model, beta=beta, keys=k)


def final_sampling(model, params, dist, args, key_gen, stats=None):
    """``exe_flow_matching.py:453-459``: the flow's samples and their self-normalised importance resampling.

    ``key_gen`` is the key the reference reuses here (``:333``, or ``:371`` when exact samples exist): the
    reference draws split into ``eval_iter * num_chain`` (``:389,453``), and it is split AGAIN into the ONE Hutchinson key
    shared by all samples and the key of the categorical draw (``:454-455``).  Returns
    ``(flow_samples, exact_samples, info)``; ``info`` carries every intermediate (``u, vols, samples_logdensity,
    log_weights, weights, idx``)."""
    from . import ode
    n = args.eval_iter * args.num_chain
    ref = targets.IndepGaussian(dist.dim, var=targets.REF_VARS[args.ref_dist])           # :388 (ref_dists, :48-54)
    u = ref.sample_model_rows(prng.split(key_gen, n))                                    # :453 (:389)
    key_hutch, key_choice = prng.split(key_gen)                                          # :454
    x, vols = ode.transform_and_logdet(model, params, key_hutch, u, args.hutchs, args.rtol, args.atol, args.mxstep,
                                       n_ts=args.n_ts, stats=stats, fixed=flow.fixed_mode(args))   # :455 (one shared probe key)
    lp = dist.logprob(x)                                                                 # :456
    logw = lp - ref.logprob(u) - vols                                                    # :457
    w = np.exp(logw - logw.max())                                                        # :458
    idx = prng.choice_p(key_choice, w, (n,))                                             # :459 (with replacement)
    idx = np.minimum(idx, n - 1)
    return x, x[idx], dict(u=u, vols=vols, samples_logdensity=lp, log_weights=logw, weights=w, idx=idx)
