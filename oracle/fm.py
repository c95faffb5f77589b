"""Flow-matching batch construction, loss and parameter gradient.

ORACLE (test infrastructure; see oracle/__init__.py).  Follows
``exe_flow_matching.py:139-147`` (flow_fn), ``:151-169`` (cond_flow_fn, the default
path: ``multi_modal.py:162-163``), ``:171-179`` (loss = SUM of squared residuals) and
``:362-366`` (value_and_grad w.r.t. the parameters).  The OT branch (``:156-165``)
references names the reference never imports and is dead.

``n_total`` / ``start`` describe a shard of the chain axis: the draws are indexed by
GLOBAL chain id, so a rank holding chains ``[start, start+B)`` of ``n_total`` draws
what a single process would (build-side extension; the reference is single-device).
"""
import numpy as np

from . import prng


def cond_flow_batch(key, samples, sigma, n_total=None, start=0, ref_std=1.0):
    """``exe_flow_matching.py:151-169``; ref_dist = IndepGaussian(dim, var = ref_std**2) (``:48-54,149,155``)."""
    B, d = samples.shape
    n_total = B if n_total is None else n_total
    key_time, key_ref, key_gauss, _key_ot = prng.split(key, 4)                       # :153
    t = prng.uniform(key_time, (n_total, 1), start=start, count=B)                   # :154
    ref_keys = prng.split_at(key_ref, n_total, np.arange(start, start + B))          # :155
    x0 = ref_std * prng.normal_rows(ref_keys, d)                                     # distributions.py:96-97
    eps = prng.normal(key_gauss, (n_total, d), start=start * d, count=B * d).reshape(B, d)   # :166
    tt = t[:, None]
    cond = sigma * eps + tt * samples + (1.0 - tt) * x0                              # :167
    target = samples - x0                                                            # :168
    return t, cond, target


def flow_batch(key, samples, sigma, n_total=None, start=0):
    """``exe_flow_matching.py:139-147`` (only reachable with cond_flow off; not from the CLI)."""
    B, d = samples.shape
    n_total = B if n_total is None else n_total
    key_time, key_ref = prng.split(key, 2)
    t = prng.uniform(key_time, (n_total, 1), start=start, count=B)
    x0 = prng.normal(key_ref, (n_total, d), start=start * d, count=B * d).reshape(B, d)
    tt = t[:, None]
    sds = 1.0 - (1.0 - sigma) * tt
    return t, tt * samples + sds * x0, samples - (1.0 - sigma) * x0


def loss_and_grad(model, params, key, samples, sigma, cond_flow=True, n_total=None, start=0,
                  need_grad=True, ref_std=1.0):
    """``exe_flow_matching.py:171-178`` + ``:364-365``."""
    batch = cond_flow_batch if cond_flow else flow_batch
    t, cond, target = batch(key, samples, sigma, n_total, start, ref_std) if cond_flow else batch(key, samples, sigma, n_total, start)
    if not need_grad:
        v = model.forward(params, cond, t)
        return ((v - target) ** 2).sum(), None
    v, cache = model.forward(params, cond, t, cache=True)
    diffs = v - target                                                               # :177
    loss = (diffs * diffs).sum()                                                     # :178
    grads = model.backward(params, cache, 2.0 * diffs)
    return loss, grads
