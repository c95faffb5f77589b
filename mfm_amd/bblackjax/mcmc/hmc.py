"""Hamiltonian Monte Carlo kernel on MI355X -- a BUILD-SIDE MODE beside ``mala.py``.

BASELINE.json's north star names a "MALA/HMC log-density-and-grad step"; the reference's MFM loop uses MALA only and its vendored
``bblackjax/mcmc`` has no ``hmc.py`` (SURVEY.md note 7), so this module has no counterpart to be a drop-in for.  It follows the kernel of
blackjax (the package ``bblackjax`` was cut from) with the same conventions as ``mala.py`` here: ``state.position`` is
``[n_chain_local, dim]`` (CUDA float32), ``rng_key`` the key BEFORE the per-chain split, ``logdensity_fn`` built from a device target;
unit mass matrix, velocity Verlet, acceptance ``min(1, exp(H_0 - H_end))`` (``oracle/hmc.py``; device: ``mfm_hmc_step``)."""
from typing import Callable, NamedTuple

from ...distributions import resolve_logdensity
from ..base import SamplingAlgorithm
from .mala import MALAState as HMCState, _engine, init

__all__ = ["HMCState", "HMCInfo", "init", "build_kernel", "hmc"]


class HMCInfo(NamedTuple):
    acceptance_rate: object
    is_accepted: object


def build_kernel():
    def kernel(rng_key, state: HMCState, logdensity_fn: Callable, step_size: float, num_integration_steps: int):
        dist, beta = resolve_logdensity(logdensity_fn)
        eng = _engine(dist)
        t = eng.torch
        pos, logp, grad = state.position.clone(), state.logdensity.clone(), state.logdensity_grad.clone()      # states are values
        acc = t.empty(pos.shape[0], device=pos.device, dtype=t.float32)
        isacc = t.empty(pos.shape[0], device=pos.device, dtype=t.uint8)
        eng.ctx.hmc_step(rng_key, beta, step_size, num_integration_steps, pos, logp, grad, acc, isacc)
        return HMCState(pos, logp, grad), HMCInfo(acc, isacc.bool())

    return kernel


class hmc:
    init = staticmethod(init)
    build_kernel = staticmethod(build_kernel)

    def __new__(cls, logdensity_fn: Callable, step_size: float, num_integration_steps: int) -> SamplingAlgorithm:
        kernel = cls.build_kernel()

        def init_fn(position):
            return cls.init(position, logdensity_fn)

        def step_fn(rng_key, state):
            return kernel(rng_key, state, logdensity_fn, step_size, num_integration_steps)

        return SamplingAlgorithm(init_fn, step_fn)
