"""Baseline runners -- drop-in for the reference's ``exe_others.py``, for the ONE baseline that shares the MFM hot path:
adaptive tempered SMC on the MALA kernel (``exe_others.py:79-111``).  The other baselines
(flowMC, pocoMC, DDS, FAB: ``:44-76,114-296``) wrap third-party samplers outside the scope and raise.

``run(dist, args, target_gn=None) -> (res[5], res_[5])`` as ``exe_others.py:22,376``: logpdf, Stein U / V, MMD, train time
for the collected particles (the reference reports the same MCMC particles as "flow" and "exact" samples, ``:108-109``).
"""
import logging
import time

import numpy as np

from . import random as jr
from . import wandb_shim as wandb
from .bblackjax.mcmc import mala
from .bblackjax.smc import adaptive_tempered, base as smc_base, resampling
from .engine import Engine, allgather_cat
from .exe_flow_matching import _logprob_any, stein_disc

logger = logging.getLogger(__name__)


def run(dist, args, target_gn=None, return_extras=False):
    import torch
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s - %(message)s", datefmt="%m/%d/%Y %H:%M:%S", level=logging.INFO)
    if not getattr(args, "do_smc", False):
        raise NotImplementedError("only --do_smc (adaptive tempered SMC on the MALA kernel) is built; flowMC / pocoMC / DDS / FAB "
                                  "wrap third-party samplers outside the hot-path scope (SURVEY.md section 2)")
    learning_iter, n_iter, n_chain = args.learning_iter, args.eval_iter, args.num_chain
    key_target, key_sample, key_init, key_dist, key_fourier, key_gen = jr.split(jr.PRNGKey(args.seed), 6)     # :33
    dist.initialize_model(key_dist, n_chain)                                                # :34
    eng = Engine(dist, args, None, max_eval_samples=n_iter * n_chain)
    if eng.n_valid != eng.n_local:
        raise NotImplementedError("the SMC baseline needs num_chain to be a multiple of 16 per GPU (no padding rows among the particles)")
    smc_base.attach(eng)
    real_samples = None
    if target_gn is not None:                                                               # :36-39
        key_gen, key_loss = jr.split(key_target)
        real_samples = torch.as_tensor(np.ascontiguousarray(dist.sample_rows(jr.split(key_gen, n_iter * n_chain)), dtype=np.float32), device=eng.dev)
    logger.info(f"===== Starting training seed {args.seed} w/ {learning_iter} iterations =====")
    logger.info("Adaptive tempered SMC")
    tempered = adaptive_tempered.adaptive_tempered_smc(                                     # :85-94
        dist.logprior, dist.loglik, mala.build_kernel(), mala.init, dict(step_size=args.step_size), resampling.systematic,
        args.alpha, num_mcmc_steps=args.anneal_iter // args.num_anneal_temp)
    keys = jr.split(jr.PRNGKey(args.seed), learning_iter)                                   # :101
    state = tempered.init(eng.local(dist.init_params))                                      # :102
    lmbdas = []
    train_start = time.time()
    for k in keys:                                                                          # :104 (lax.scan of one_step)
        state, info = tempered.step(k, state)
        lmbdas.append(state.lmbda)
    eng.ctx.sync()
    train_time = time.time() - train_start
    logger.info(f"Final temp= {state.lmbda}")
    keys2 = jr.split(keys[0], n_iter)                                                       # :107
    collected = []
    for k in keys2:                                                                         # :108
        state, info = tempered.step(k, state)
        collected.append(allgather_cat(state.particles.contiguous()))                        # (more than one rank: every rank evaluates ALL particles)
    flow_samples = torch.cat(collected)                                                     # :109 "not really flow but MCMC"
    exact_samples = flow_samples                                                            # :110

    logpdf = _logprob_any(eng, flow_samples).mean().item()                                  # :308
    stein = stein_disc(eng, flow_samples)
    logpdf_, stein_ = logpdf, stein                                                         # same particles (:108-109)
    if real_samples is not None:
        mmd = mmd_ = eng.ctx.max_mean_disc(real_samples, flow_samples)
    else:
        mmd = mmd_ = 0.0
    wandb.log({"train_time": train_time, "logpdf": logpdf, "KSD U-stat": stein[0], "KSD V-stat": stein[1]})
    res = np.array([logpdf, stein[0], stein[1], mmd, train_time])
    res_ = np.array([logpdf_, stein_[0], stein_[1], mmd_, train_time])
    wandb.finish()
    if return_extras:
        return res, res_, dict(lmbdas=np.array(lmbdas), state=state, engine=eng, samples=flow_samples)
    eng.close()
    return res, res_
