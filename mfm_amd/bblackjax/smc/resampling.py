"""``bblackjax/smc/resampling.py``: the four resampling schemes on device tensors.

``systematic`` (``:50-52,124-135``: the one ``exe_others.py:91`` uses), ``stratified`` (``:55-57``) and ``multinomial`` (``:60-80``) are one
kernel each (sequential float64 cumulative sums + one binary search per output: the indices are integer outputs and must not depend
on a scan tree).  ``residual`` (``:83-121``) is composed here as the reference composes it: the integer parts of ``N w`` as repeats,
the rest by a multinomial draw on the residual weights, shuffled by ``jax.random.permutation`` (restated in ``mfm_amd/random.py``).
"""
import numpy as np

from ... import random as jr
from .base import _engine_of


def _cumsum_scheme(scheme, rng_key, weights, num_samples):
    import torch
    if num_samples != weights.shape[0]:
        raise NotImplementedError("num_samples != number of particles (waste-free SMC) is not built")
    eng = _engine_of(weights)
    idx = torch.empty(num_samples, device=weights.device, dtype=torch.int32)
    scratch = torch.empty(2 * num_samples + 2, device=weights.device, dtype=torch.float64)
    eng.ctx.smc_resample_scheme(scheme, rng_key, weights, scratch, idx)
    return idx


def systematic(rng_key, weights, num_samples: int):
    return _cumsum_scheme(0, rng_key, weights, num_samples)


def stratified(rng_key, weights, num_samples: int):
    return _cumsum_scheme(1, rng_key, weights, num_samples)


def multinomial(rng_key, weights, num_samples: int):
    return _cumsum_scheme(2, rng_key, weights, num_samples)


def residual(rng_key, weights, num_samples: int):
    import torch
    key1, key2 = jr.split(rng_key)                                                       # :96
    n = weights.shape[0]
    nw = num_samples * weights                                                           # :98
    integer_part = torch.floor(nw).to(torch.int32)                                       # :101
    sum_int = int(integer_part.sum().item())                                             # :102
    residual_part = nw - integer_part                                                    # :104
    residual_sample = multinomial(key1, (residual_part / (num_samples - sum_int)).contiguous(), num_samples)   # :105-107
    perm = torch.as_tensor(jr.permutation_indices(key2, num_samples), device=weights.device)
    residual_sample = residual_sample[perm]                                              # :114
    counts = torch.cat([integer_part, torch.tensor([num_samples - sum_int], device=weights.device, dtype=torch.int32)])
    integer_idx = torch.repeat_interleave(torch.arange(n + 1, device=weights.device, dtype=torch.int32), counts.long())[:num_samples]   # :116-120
    pos = torch.arange(num_samples, device=weights.device)
    return torch.where(pos >= sum_int, residual_sample, integer_idx)                     # :122
