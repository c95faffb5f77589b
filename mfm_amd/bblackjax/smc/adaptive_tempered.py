"""``bblackjax/smc/adaptive_tempered.py``: tempered SMC with the temperature increment chosen by the ESS solver
(``:15-91``) and the user-facing ``adaptive_tempered_smc`` (``:94-173``)."""
from ..base import SamplingAlgorithm
from . import ess, solver, tempered


def build_kernel(logprior_fn, loglikelihood_fn, mcmc_step_fn, mcmc_init_fn, resampling_fn, target_ess, root_solver=solver.dichotomy):
    def compute_delta(state):
        max_delta = 1 - state.lmbda                                                     # :60-61
        delta = ess.ess_solver(loglikelihood_fn, state.particles, target_ess, max_delta, root_solver)   # :62-68
        return min(max(delta, 0.0), max_delta) if delta == delta else delta             # :70 (clip; NaN propagates)

    tempered_kernel = tempered.build_kernel(logprior_fn, loglikelihood_fn, mcmc_step_fn, mcmc_init_fn, resampling_fn)

    def kernel(rng_key, state, num_mcmc_steps, mcmc_parameters):
        delta = compute_delta(state)
        lmbda = delta + state.lmbda                                                     # :87-88
        return tempered_kernel(rng_key, state, num_mcmc_steps, lmbda, mcmc_parameters)

    return kernel


class adaptive_tempered_smc:
    init = staticmethod(tempered.init)
    build_kernel = staticmethod(build_kernel)

    def __new__(cls, logprior_fn, loglikelihood_fn, mcmc_step_fn, mcmc_init_fn, mcmc_parameters, resampling_fn, target_ess,
                root_solver=solver.dichotomy, num_mcmc_steps=10):
        kernel = cls.build_kernel(logprior_fn, loglikelihood_fn, mcmc_step_fn, mcmc_init_fn, resampling_fn, target_ess, root_solver)
        return SamplingAlgorithm(cls.init, lambda rng_key, state: kernel(rng_key, state, num_mcmc_steps, mcmc_parameters))
