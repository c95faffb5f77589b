"""CLI / experiment driver -- drop-in for the reference's ``multi_modal.py`` (same flags, defaults and per-example
overrides, ``multi_modal.py:21-101,147-220``), running the MFM loop on MI355X.

Additions (defaults leave the reference behaviour untouched): ``--force_dim`` / ``--force_num_chain`` override the
values ``main`` hard-codes per example (needed for BASELINE.json's phi-four d=256 / 4096-chain configuration;
``multi_modal.py:52,55`` fix 64 / 1024), ``--log_every`` sets how often metrics are copied to the host,
``--ode_method rk4|euler --ode_steps N`` integrates the flow on N equal steps instead of the reference's adaptive Dopri5.
``--do_smc`` runs the tempered-SMC baseline on the same MALA kernel (``exe_others.py:79-111``); the other baselines
(``--do_flowmc`` ... ``--do_fab``) wrap third-party samplers outside the hot-path scope and raise.

Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N -m mfm_amd.multi_modal ...``; chains are
sharded over ranks and the flow-matching gradient is all-reduced over RCCL.
"""
import argparse
import os

import numpy as np

from . import random as jr
from . import wandb_shim as wandb
from .distributions import GaussianMixture, LogGaussianCoxPines, PhiFour
from .exe_flow_matching import run
from .exe_others import run as run_others


def gmm16_parameters(num_modes=16, dim=2, lim=(-16, 16)):
    """(modes, covs, weights) of the `gaussian-mixture` example, multi_modal.py:39-47 (frozen in tests/golden/gmm16_params.npz)."""
    key_mode, key_cov, key_weight = jr.split(jr.PRNGKey(0), 3)
    modes = jr.uniform(key_mode, (num_modes, dim), lim[0] * .8, lim[1] * .8)
    covs = np.exp(.5 * jr.normal(key_cov, (num_modes, dim)))
    weights = jr.dirichlet(key_weight, 4. * np.ones(num_modes))                # :45 (jax's gamma sampler restated: random.py)
    return modes, covs, weights


def main(args):
    if args.example == "gaussian-mixture":                                              # :23-47
        print("Setting up Gaussian mixture density...")
        args.dim, args.num_modes, args.lim, args.levels, args.step_size = 2, 16, [-16, 16], 20, 0.2
        dist = GaussianMixture(*gmm16_parameters(args.num_modes, args.dim, args.lim))
    elif args.example == "phi-four":                                                    # :50-63
        print("Setting up Phi four example density...")
        args.dim = 64
        args.lim, args.num_chain, args.eval_iter, args.step_size = [-1.6, 1.6], 1024, 1, 0.0001
        if args.force_dim:
            args.dim = args.force_dim
        dist = PhiFour(args.dim)
        dist.sample_model = None
    elif args.example == "4-mode":                                                      # :65-85
        print("Setting up 4-mode Gaussian mixture density...")
        args.dim, args.lim, args.levels, args.step_size = 2, [-16, 16], 20, 0.2
        dist = GaussianMixture(8. * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1]]), np.ones((4, 2)), np.ones(4) / 4)
    elif args.example == "pines":                                                       # :87-98
        print("Setting up Log Gaussian Cox density...")
        args.dim, args.lim, args.num_chain, args.eval_iter, args.step_size = 1600, None, 128, 1, 0.01
        args.hidden_x = args.hidden_t = args.hidden_xt = [1024, 1024]
        if args.force_dim:
            args.dim = args.force_dim
        dist = LogGaussianCoxPines(args.dim)
        dist.sample_model = None
    else:
        raise Exception("Example not found.")
    if args.force_num_chain:
        args.num_chain = args.force_num_chain
    if args.do_flowmc or args.do_pocomc or args.do_dds or args.do_fab:
        raise NotImplementedError("the flowMC / pocoMC / DDS / FAB runners (exe_others.py) wrap third-party samplers outside the "
                                  "hot-path scope (SURVEY.md section 2); --do_smc (tempered SMC on the MALA kernel) is built")

    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as td
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if not td.is_initialized():
            td.init_process_group("nccl")

    N_PARAM = args.dim
    job_type = "mcmc_per_flow_steps=" + str(args.mcmc_per_flow_steps) + ",learning_iter=" + str(args.learning_iter) + (",hutchs" if args.hutchs else "")
    if args.do_smc:
        job_type = "Adaptive tempered SMC"                                              # :110-111
    seeds = [args.seed] if args.seed else [i ** 10 for i in range(10)]                  # :118
    res, res_ = [], []
    for seed in seeds:
        args.seed = seed
        wandb.init(project=args.example, config=args, group="dim=" + str(N_PARAM), job_type=job_type)
        if args.do_smc:
            _res, _res_ = run_others(dist, args, dist.sample_model)                     # :126-127
        else:
            _res, _res_ = run(dist, args, dist.sample_model, log_every=args.log_every)  # :129
        res.append(_res); res_.append(_res_)
    res, res_ = np.array(res), np.array(res_)
    print(job_type)
    print("-" * 100)
    print("logprob\t & stein-u\t & stein-v\t & mmd  \t & time \t")
    print(*[f"{m:.2e} \\pm {s * 1.96:.2e}" for m, s in zip(res.mean(0), res.std(0))], sep="$ & $")
    print(*[f"{m:.2e} \\pm {s * 1.96:.2e}" for m, s in zip(res_.mean(0), res_.std(0))], sep="$ & $")
    print("-" * 100)
    return res, res_


def build_parser():
    parser = argparse.ArgumentParser()                                                  # :148-219, same defaults
    parser.add_argument("--seed", type=int, default=None)
    parser.add_argument('--dim', type=int, default=64)
    parser.add_argument('--num_modes', type=int, default=16)
    parser.add_argument("--example", type=str, default="pines")
    parser.add_argument("--sigma", type=float, default=1e-4)
    parser.add_argument("--fourier_dim", type=int, default=128)
    parser.add_argument("--fourier_std", type=float, default=1.0)
    parser.add_argument('--hutchs', dest='hutchs', action='store_true')
    parser.set_defaults(hutchs=False)
    parser.add_argument("--ref_dist", type=str, default='stdgauss')
    parser.add_argument('--cond_flow', dest='cond_flow', action='store_true')
    parser.set_defaults(cond_flow=True)
    parser.add_argument('--ot_cond_flow', dest='ot_cond_flow', action='store_true')
    parser.set_defaults(ot_cond_flow=False)
    parser.add_argument("--num_importance_samples", type=int, default=0)
    parser.add_argument("--mcmc_per_flow_steps", type=float, default=10)
    parser.add_argument('--num_chain', type=int, default=128)
    parser.add_argument("--learning_iter", type=int, default=400)
    parser.add_argument("--eval_iter", type=int, default=100)
    parser.add_argument("--alpha", type=float, default=0.95)
    parser.add_argument("--anneal_iter", type=int, default=200)
    parser.add_argument('--num_anneal_temp', type=int, default=200)
    parser.add_argument('--non_linearity', type=str, default='relu')
    parser.add_argument('--hidden_x', type=int, nargs='+', default=[128, 128])
    parser.add_argument('--hidden_t', type=int, nargs='+', default=[128, 128])
    parser.add_argument('--hidden_xt', type=int, nargs='+', default=[128, 128])
    parser.add_argument('--step_size', type=float, default=0.2)
    for flag in ("flowmc", "pocomc", "dds", "smc", "fab"):
        parser.add_argument(f'--do_{flag}', dest=f'do_{flag}', action='store_true')
        parser.set_defaults(**{f'do_{flag}': False})
    parser.add_argument('--learning_rate', type=float, default=1e-3)
    parser.add_argument('--weight_decay', type=float, default=0.0001)
    parser.add_argument('--adam_beta1', type=float, default=0.9)
    parser.add_argument('--adam_beta2', type=float, default=0.999)
    parser.add_argument('--adam_epsilon', type=float, default=1e-8)
    parser.add_argument('--gradient_clip', type=float, default=1.0)
    parser.add_argument('--warmup_steps', type=int, default=0)
    parser.add_argument('--rtol', type=float, default=1e-5)
    parser.add_argument('--atol', type=float, default=1e-5)
    parser.add_argument('--mxstep', type=float, default=1_000)
    parser.add_argument('--lim', type=float, nargs=2, default=[-16, 16])
    parser.add_argument('--grid_width', type=int, default=400)
    parser.add_argument('--levels', type=int, default=50)
    parser.add_argument('--check', dest='check', action='store_true')
    parser.set_defaults(check=False)
    # build-side additions
    parser.add_argument('--force_dim', type=int, default=None)
    parser.add_argument('--force_num_chain', type=int, default=None)
    parser.add_argument('--log_every', type=int, default=1)
    # the fixed-step mode of the CNF solver (BASELINE.json's "RK4/Euler ODE integrator"; the reference integrates with the adaptive
    # Dopri5 of jax.experimental.ode.odeint, exe_flow_matching.py:345-349, which stays the default): --ode_method rk4 --ode_steps 64
    parser.add_argument('--ode_method', type=str, default='dopri5', choices=['dopri5', 'rk4', 'euler'])
    parser.add_argument('--ode_steps', type=int, default=0)
    # the MCMC move of a non-flow iteration: the reference's MALA step (exe_flow_matching.py:313), or -- BASELINE.json's "MALA/HMC" step,
    # not in the reference -- an HMC step of --hmc_steps velocity-Verlet steps of size --step_size (mfm_amd/bblackjax/mcmc/hmc.py)
    parser.add_argument('--mcmc_kernel', type=str, default='mala', choices=['mala', 'hmc'])
    parser.add_argument('--hmc_steps', type=int, default=10)
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
