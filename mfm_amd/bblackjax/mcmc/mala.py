"""Public API for Metropolis Adjusted Langevin kernels -- drop-in for ``bblackjax/mcmc/mala.py`` on MI355X.

Same names and argument meaning as the reference (``MALAState`` / ``MALAInfo`` ``:16-48``, ``init`` ``:51-54``,
``build_kernel`` ``:57-120``, ``mala`` ``:123-189``).  Differences forced by the C-ABI boundary:

* the reference kernel is written per chain and batched by the CALLER with ``jax.vmap`` over
  ``keys = jax.random.split(rng_key, n_chain)`` (``exe_flow_matching.py:303,313``); here the kernel is batched itself:
  ``state.position`` is ``[n_chain_local, dim]`` (CUDA float32) and ``rng_key`` is the key BEFORE that split;
* ``logdensity_fn`` must be built from ``dist.loglik / dist.logprior / dist.logprob`` of one of the built targets
  (``mfm_amd.distributions``); arbitrary closures raise ``NotImplementedError`` (no CPU fallback);
* ``logdensity`` is float64, positions and gradients float32;
* a caller that vmaps over its OWN keys (``bblackjax/smc/base.py:122-123``) passes ``rng_key`` of shape ``[n_chain, 2]``.

The acceptance rule is the reference's AS WRITTEN.  ``build_kernel(textbook=True)`` flips it.
"""
from typing import Callable, NamedTuple

import numpy as np

from ...distributions import resolve_logdensity
from ..base import SamplingAlgorithm

__all__ = ["MALAState", "MALAInfo", "init", "build_kernel", "mala"]


class MALAState(NamedTuple):
    position: object
    logdensity: object
    logdensity_grad: object


class MALAInfo(NamedTuple):
    acceptance_rate: object
    is_accepted: object
    proposed_position: object
    proposed_weight: object


def _engine(dist):
    if dist._engine is None:
        raise RuntimeError("attach the distribution to a device engine first (mfm_amd.engine.Engine(dist, args))")
    return dist._engine


def _rows(eng, x):
    """A single chain ``[dim]`` (the reference's un-vmapped call shape) rides as every row of one n_chain_local batch."""
    return x[None].expand(eng.n_local, -1).contiguous()


def init(position, logdensity_fn: Callable) -> MALAState:
    dist, beta = resolve_logdensity(logdensity_fn)
    eng = _engine(dist)
    t = eng.torch
    single = position.ndim == 1
    pos = _rows(eng, position) if single else position
    logp = t.empty(pos.shape[0], device=pos.device, dtype=t.float64)
    grad = t.empty_like(pos)
    eng.ctx.mala_init(pos, beta, logp, grad)
    if single:
        return MALAState(position, logp[0], grad[0])
    return MALAState(position, logp, grad)


def build_kernel(textbook: bool = False):
    def kernel(rng_key, state: MALAState, logdensity_fn: Callable, step_size: float):
        """``mala.py:86-118``.  ``state.position`` ``[n_chain_local, dim]`` with ONE key: the vmapped call of
        ``exe_flow_matching.py:303,313`` (chain b draws from ``split(rng_key, n_chain_total)[chain_offset + b]``); with keys
        ``[n_chain_local, 2]``: the caller's own vmap (``smc/base.py:122-123``); ``state.position`` ``[dim]``: the kernel
        as the reference writes it, one chain and its key."""
        dist, beta = resolve_logdensity(logdensity_fn)
        eng = _engine(dist)
        t = eng.torch
        single = state.position.ndim == 1
        if single:
            pos, grad = _rows(eng, state.position), _rows(eng, state.logdensity_grad)
            logp = state.logdensity.reshape(1).expand(eng.n_local).contiguous()
            rng_key = np.tile(np.asarray(rng_key, dtype=np.uint32).reshape(1, 2), (eng.n_local, 1))
        else:
            pos, logp, grad = state.position.clone(), state.logdensity.clone(), state.logdensity_grad.clone()
        n = pos.shape[0]
        acc = t.empty(n, device=pos.device, dtype=t.float32)
        isacc = t.empty(n, device=pos.device, dtype=t.uint8)
        prop = t.empty_like(pos)
        w = t.empty(n, device=pos.device, dtype=t.float32)
        if getattr(rng_key, "ndim", 1) == 2:          # [n_chain, 2]: the caller already split its key per chain (smc/base.py:122-123)
            keys = rng_key if t.is_tensor(rng_key) else t.as_tensor(np.ascontiguousarray(rng_key, dtype=np.uint32).view(np.int32), device=pos.device)
            eng.ctx.mala_step_keys(keys, beta, step_size, pos, logp, grad, acc, isacc, prop, w, textbook=textbook)
        else:
            eng.ctx.mala_step(rng_key, beta, step_size, pos, logp, grad, acc, isacc, prop, w, textbook=textbook)
        if single:
            return MALAState(pos[0], logp[0], grad[0]), MALAInfo(acc[0], isacc.bool()[0], prop[0], w[0])
        return MALAState(pos, logp, grad), MALAInfo(acc, isacc.bool(), prop, w)

    return kernel


class mala:
    """``mala(logdensity_fn, step_size) -> SamplingAlgorithm(init, step)`` (``mala.py:123-189``)."""

    init = staticmethod(init)
    build_kernel = staticmethod(build_kernel)

    def __new__(cls, logdensity_fn: Callable, step_size: float) -> SamplingAlgorithm:
        kernel = cls.build_kernel()

        def init_fn(position):
            return cls.init(position, logdensity_fn)

        def step_fn(rng_key, state):
            return kernel(rng_key, state, logdensity_fn, step_size)

        return SamplingAlgorithm(init_fn, step_fn)
