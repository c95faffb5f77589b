"""``bblackjax/smc/base.py``: the general SMC step (``:55-134``) -- resample, move, weigh -- on device tensors.

``update_fn`` / ``weigh_fn`` are batched here (the reference passes ``jax.vmap``-ed callables, ``tempered.py:141-142``):
``update_fn(keys [N, 2], particles [N, d]) -> (particles, info)``, ``weigh_fn(particles) -> log-weights [N]`` (float64).
"""
from typing import NamedTuple

import numpy as np

from ... import random as jr


class SMCState(NamedTuple):
    particles: object
    weights: object


class SMCInfo(NamedTuple):
    ancestors: object
    log_likelihood_increment: float
    update_info: object


def init(particles):
    import torch
    n = particles.shape[0]
    return SMCState(particles, torch.full((n,), 1.0 / n, device=particles.device, dtype=torch.float64))


def step(rng_key, state: SMCState, update_fn, weigh_fn, resample_fn, num_resampled=None):
    import torch
    updating_key, resampling_key = jr.split(rng_key, 2)                                 # :114
    num_particles = state.weights.shape[0]
    if num_resampled is None:
        num_resampled = num_particles
    resampling_idx = resample_fn(resampling_key, state.weights, num_resampled)          # :119
    eng = getattr(resample_fn, "_engine", None) or _engine_of(state.particles)
    particles = torch.empty_like(state.particles)
    eng.ctx.gather_rows(state.particles, resampling_idx, particles)                     # :120
    keys = jr.split(updating_key, num_resampled)                                        # :122
    particles, update_info = update_fn(keys, particles)                                 # :123
    log_weights = weigh_fn(particles)                                                   # :125
    weights = torch.empty(num_particles, device=particles.device, dtype=torch.float64)
    normalizing_constant = eng.ctx.smc_weights(log_weights, 1.0, weights)               # :126-128
    return SMCState(particles, weights), SMCInfo(resampling_idx, normalizing_constant, update_info)


_ENGINE = [None]


def attach(engine):
    """The device engine the SMC pieces run on (one per process)."""
    _ENGINE[0] = engine


def _engine_of(_tensor):
    if _ENGINE[0] is None:
        raise RuntimeError("attach a device engine first: mfm_amd.bblackjax.smc.base.attach(Engine(dist, args))")
    return _ENGINE[0]
