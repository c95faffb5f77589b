"""``bblackjax/smc/base.py``: the general SMC step (``:55-134``) -- resample, move, weigh -- on device tensors.

``update_fn`` / ``weigh_fn`` are batched here (the reference passes ``jax.vmap``-ed callables, ``tempered.py:141-142``):
``update_fn(keys [N, 2], particles [N, d]) -> (particles, info)``, ``weigh_fn(particles) -> log-weights [N]`` (float64).

More than one rank (build-side: the reference is single-device): every rank holds a contiguous shard of the N particles and ALL N
weights.  Resampling is over all particles (``:119``: every rank computes the same N indices from the same key and keeps its
slice), so a step all-gathers the particles (ancestors live anywhere) and the new log-weights; the MCMC move works on the shard
with particle b's key ``split(updating_key, N)[b]`` (``:122``) -- a sharded run draws what the single process draws.
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


def _sharded(eng):
    from ...engine import _collective, _dist
    return eng is not None and _collective(_dist()) and eng.world > 1


def init(particles):
    import torch
    eng = _ENGINE[0]
    n = eng.n_total if _sharded(eng) else particles.shape[0]
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
    keys = jr.split(updating_key, num_resampled)                                        # :122
    if _sharded(eng):
        from ...engine import allgather_cat
        lo, hi = eng.offset, eng.offset + eng.n_local
        eng.ctx.gather_rows(allgather_cat(state.particles.contiguous()), resampling_idx[lo:hi].contiguous(), particles)   # :120
        particles, update_info = update_fn(keys[lo:hi], particles)                      # :123
        log_weights = allgather_cat(weigh_fn(particles).contiguous())                   # :125
    else:
        eng.ctx.gather_rows(state.particles, resampling_idx, particles)                 # :120
        particles, update_info = update_fn(keys, particles)                             # :123
        log_weights = weigh_fn(particles)                                               # :125
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
