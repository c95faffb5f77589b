"""``bblackjax/smc/resampling.py``: systematic resampling (``:50-52,124-135``) on the device; the other schemes of the
reference file (stratified, multinomial, residual) are not used by the SMC baseline (``exe_others.py:91``)."""
from .base import _engine_of


def systematic(rng_key, weights, num_samples: int):
    import torch
    if num_samples != weights.shape[0]:
        raise NotImplementedError("num_samples != number of particles (waste-free SMC) is not built")
    eng = _engine_of(weights)
    idx = torch.empty(num_samples, device=weights.device, dtype=torch.int32)
    scratch = torch.empty(num_samples, device=weights.device, dtype=torch.float64)
    eng.ctx.smc_resample(rng_key, weights, scratch, idx)
    return idx


def stratified(*_a, **_k):
    raise NotImplementedError("only systematic resampling is built (the one exe_others.py:91 uses)")


multinomial = residual = stratified
