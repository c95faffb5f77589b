"""``bblackjax/smc/tempered.py``: tempered SMC kernel (``:27-148``) on the device MALA kernels.

The MCMC move targets ``logprior + state.lmbda * loglikelihood`` -- the temperature BEFORE the increment, as the reference
writes it (``:120-123``) -- and the particles are weighed with ``delta * loglikelihood`` (``:118-119``)."""
from typing import NamedTuple

from ... import random as jr
from ..base import SamplingAlgorithm
from . import base
from .base import SMCState


class TemperedSMCState(NamedTuple):
    particles: object
    weights: object
    lmbda: float


def init(particles):
    s = base.init(particles)
    return TemperedSMCState(s.particles, s.weights, 0.0)                                # :45-50


def build_kernel(logprior_fn, loglikelihood_fn, mcmc_step_fn, mcmc_init_fn, resampling_fn):
    def kernel(rng_key, state: TemperedSMCState, num_mcmc_steps: int, lmbda: float, mcmc_parameters: dict):
        delta = lmbda - state.lmbda                                                     # :116

        def log_weights_fn(particles):
            return delta * loglikelihood_fn(particles)                                  # :118-119 (device, float64)

        def tempered_logposterior_fn(position):
            return logprior_fn(position) + state.lmbda * loglikelihood_fn(position)     # :121-124

        def mcmc_kernel(keys, particles):                                               # :126-137, batched over particles
            st = mcmc_init_fn(particles, tempered_logposterior_fn)
            info = None
            step_keys = jr.split_rows(keys, num_mcmc_steps)                             # [N, steps, 2]
            for j in range(num_mcmc_steps):
                st, info = mcmc_step_fn(step_keys[:, j], st, tempered_logposterior_fn, **mcmc_parameters)
            return st.position, info

        smc_state, info = base.step(rng_key, SMCState(state.particles, state.weights), mcmc_kernel, log_weights_fn, resampling_fn)
        return TemperedSMCState(smc_state.particles, smc_state.weights, state.lmbda + delta), info    # :145-148

    return kernel


class tempered_smc:
    init = staticmethod(init)
    build_kernel = staticmethod(build_kernel)

    def __new__(cls, logprior_fn, loglikelihood_fn, mcmc_step_fn, mcmc_init_fn, mcmc_parameters, resampling_fn, num_mcmc_steps=10):
        kernel = cls.build_kernel(logprior_fn, loglikelihood_fn, mcmc_step_fn, mcmc_init_fn, resampling_fn)
        return SamplingAlgorithm(cls.init, lambda rng_key, state, lmbda: kernel(rng_key, state, num_mcmc_steps, lmbda, mcmc_parameters))
