"""Counter-based PRNG: Threefry-2x32 with the key/counter conventions of jax.random.

ORACLE (test infrastructure; see oracle/__init__.py).  PARITY UNPINNED: jax is
not importable here, so the conventions below are the published ones of
``jax._src.prng`` / ``jax._src.random`` at the reference's pin (jax 0.4.26,
``environment.yaml:101``, non-partitionable threefry, ``jax_enable_x64`` on as
set by ``multi_modal.py:14``).  The block function itself is pinned by the
Random123 known-answer vectors (tests/test_oracle_prng.py).

Reference call sites served: ``exe_flow_matching.py:141-143,153-155,166,212,
232,247,265,268,275,303,333,350,433``; ``bblackjax/mcmc/mala.py:93``;
``bblackjax/util.py:80-82``; ``bblackjax/mcmc/proposal.py:179``;
``distributions.py:70-76,93-97,163-164,313-314``.

The HIP kernels implement exactly these index conventions
(mfm_amd/csrc/prng.hip.h), so a GPU chain and an oracle chain given the same key
draw the same numbers (up to float64 erfinv rounding and the final cast to
float32).
"""
import numpy as np
from scipy.special import erfinv

_U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return (x << _U32(r)) | (x >> _U32(32 - r))


def threefry2x32(key, x0, x1):
    """Threefry-2x32, 20 rounds (Random123).  key: (k0, k1); x0, x1 uint32 arrays."""
    old = np.seterr(over="ignore")
    try:
        k0 = _U32(key[0])
        k1 = _U32(key[1])
        k2 = k0 ^ k1 ^ _U32(0x1BD11BDA)
        ks = (k0, k1, k2)
        x0 = np.asarray(x0, dtype=_U32).copy()
        x1 = np.asarray(x1, dtype=_U32).copy()
        x0 += ks[0]
        x1 += ks[1]
        for blk in range(5):
            for r in _ROT[blk % 2]:
                x0 += x1
                x1 = _rotl(x1, r)
                x1 ^= x0
            x0 += ks[(blk + 1) % 3]
            x1 += ks[(blk + 2) % 3] + _U32(blk + 1)
        return x0, x1
    finally:
        np.seterr(**old)


def PRNGKey(seed):
    """jax.random.PRNGKey with x64 enabled: [seed >> 32, seed & 0xffffffff]."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=_U32)


def _threefry_2x32_counts(key, n):
    """jax ``threefry_2x32(key, iota(n))``: counters split in halves, outputs concatenated."""
    odd = n % 2
    cnt = np.arange(n + odd, dtype=_U32)
    if odd:
        cnt[-1] = 0
    half = (n + odd) // 2
    y0, y1 = threefry2x32(key, cnt[:half], cnt[half:])
    out = np.concatenate([y0, y1])
    return out[:-1] if odd else out


def split(key, num=2):
    """jax.random.split: keys[j] = (out[2j], out[2j+1]), out = threefry_2x32(key, iota(2 num))."""
    return _threefry_2x32_counts(key, 2 * num).reshape(num, 2)


def split_at(key, num, idx):
    """Rows ``idx`` of ``split(key, num)`` without materialising the rest (shard-friendly)."""
    idx = np.asarray(idx, dtype=np.int64)
    out = np.empty(idx.shape + (2,), dtype=_U32)
    for w in range(2):
        m = 2 * idx + w  # position in the concatenated output
        lo = m < num
        c0 = np.where(lo, m, m - num).astype(_U32)
        y0, y1 = threefry2x32(key, c0, (c0 + _U32(num)).astype(_U32))
        out[..., w] = np.where(lo, y0, y1)
    return out


def random_bits64(key, size, start=0, count=None):
    """64-bit draws: sample i = (y0 << 32) | y1 with (y0, y1) = threefry(key, (i, i + size)).

    ``start``/``count`` select a contiguous sub-range of the ``size`` samples
    (what a rank owning a shard of the chains draws)."""
    if count is None:
        count = size - start
    i = (np.arange(count, dtype=np.uint64) + np.uint64(start)).astype(_U32)
    y0, y1 = threefry2x32(key, i, (i + _U32(size)).astype(_U32))
    return (y0.astype(np.uint64) << np.uint64(32)) | y1.astype(np.uint64)


def _bits_to_unit(bits):
    """Mantissa fill: 52 random bits -> float64 in [0, 1)."""
    fb = (bits >> np.uint64(12)) | np.float64(1.0).view(np.uint64)
    return fb.view(np.float64) - 1.0


def uniform(key, shape=(), minval=0.0, maxval=1.0, start=0, count=None):
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    size = int(np.prod(shape)) if shape else 1
    u = _bits_to_unit(random_bits64(key, size, start, count))
    u = np.maximum(minval, u * (maxval - minval) + minval)
    return u.reshape(shape) if count is None else u


_LO = np.nextafter(np.float64(-1.0), np.float64(0.0))


def normal(key, shape=(), start=0, count=None):
    """jax.random.normal (float64): sqrt(2) * erfinv(uniform(nextafter(-1, 0), 1))."""
    u = uniform(key, shape, _LO, 1.0, start, count)
    return np.sqrt(2.0) * erfinv(u)


# ---- 32-bit draw mode --------------------------------------------------------------------------------------------------------
# What jax.random does WITHOUT jax_enable_x64.  The reference never draws this way (multi_modal.py:14 enables x64) and neither do the
# kernels; these three functions exist to pin the CONVENTIONS the 64-bit path shares with it -- the counter layout of
# ``_random_bits`` (``threefry_2x32(key, iota(n))``, odd sizes padded), the mantissa fill ``bits >> (nbits - nmant) | 1.0`` minus one,
# the (nextafter(-1, 0), 1) range and sqrt(2) erfinv of ``normal`` -- against values PUBLISHED in the jax documentation
# (tests/test_oracle_prng.py): jax itself cannot be imported here.
def random_bits32(key, size):
    return _threefry_2x32_counts(key, size)


def uniform32(key, shape=(), minval=0.0, maxval=1.0):
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    size = int(np.prod(shape)) if shape else 1
    fb = (random_bits32(key, size) >> _U32(9)) | np.float32(1.0).view(_U32)
    f = fb.view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, (f * (hi - lo) + lo).astype(np.float32)).reshape(shape)


def normal32(key, shape=()):
    u = uniform32(key, shape, np.nextafter(np.float32(-1.0), np.float32(0.0)), 1.0)
    return (np.float32(np.sqrt(2.0)) * erfinv(u.astype(np.float64)).astype(np.float32)).astype(np.float32)


def bernoulli(key, p):
    p = np.asarray(p, dtype=np.float64)
    return uniform(key, p.shape) < p


def truncated_normal(key, lower, upper, shape):
    """jax.random.truncated_normal (used by flax's lecun_normal)."""
    from scipy.special import erf
    a = erf(lower / np.sqrt(2.0))
    b = erf(upper / np.sqrt(2.0))
    u = uniform(key, shape, a, b)
    out = np.sqrt(2.0) * erfinv(u)
    return np.clip(out, np.nextafter(lower, np.inf), np.nextafter(upper, -np.inf))


def choice_p(key, p, shape=()):
    """jax.random.choice(key, n, shape, replace=True, p=p): inverse-CDF search."""
    p_cuml = np.cumsum(np.asarray(p, dtype=np.float64))
    r = p_cuml[-1] * (1.0 - uniform(key, shape))
    return np.searchsorted(p_cuml, r, side="left")


# Per-chain helpers with the vmapped reference shape -------------------------------------------

def normal_rows(keys, d):
    """vmap(lambda k: normal(k, (d,)))(keys): keys [n, 2] -> [n, d]."""
    keys = np.asarray(keys, dtype=_U32)
    n = keys.shape[0]
    i = np.arange(d, dtype=_U32)[None, :].repeat(n, 0)
    y0, y1 = threefry2x32((keys[:, 0:1], keys[:, 1:2]), i, (i + _U32(d)).astype(_U32))
    bits = (y0.astype(np.uint64) << np.uint64(32)) | y1.astype(np.uint64)
    u = np.maximum(_LO, _bits_to_unit(bits) * (1.0 - _LO) + _LO)
    return np.sqrt(2.0) * erfinv(u)


def uniform_rows(keys, d=None):
    """vmap(lambda k: uniform(k, (d,)))(keys) (d=None: scalar draw per key)."""
    keys = np.asarray(keys, dtype=_U32)
    n = keys.shape[0]
    dd = 1 if d is None else d
    i = np.arange(dd, dtype=_U32)[None, :].repeat(n, 0)
    y0, y1 = threefry2x32((keys[:, 0:1], keys[:, 1:2]), i, (i + _U32(dd)).astype(_U32))
    bits = (y0.astype(np.uint64) << np.uint64(32)) | y1.astype(np.uint64)
    u = np.maximum(0.0, _bits_to_unit(bits))
    return u[:, 0] if d is None else u


def split_rows(keys, num):
    """vmap(lambda k: split(k, num))(keys): [n, 2] -> [n, num, 2]."""
    keys = np.asarray(keys, dtype=_U32)
    n = keys.shape[0]
    cnt = np.arange(2 * num, dtype=_U32)
    x0 = cnt[None, :num].repeat(n, 0)
    x1 = cnt[None, num:].repeat(n, 0)
    y0, y1 = threefry2x32((keys[:, 0:1], keys[:, 1:2]), x0, x1)
    return np.concatenate([y0, y1], axis=1).reshape(n, num, 2)


# jax.random.dirichlet (multi_modal.py:45) ------------------------------------------------------------------------------------

def gamma_log(key, alpha):
    """``jax._src.random._gamma_one(key, alpha, log_space=True)`` (jax 0.4.26; third-party, restated from the published source
    -- PARITY UNPINNED; the sampler is checked distributionally against scipy.stats.gamma in tests/test_oracle_prng.py):
    Marsaglia-Tsang with ``d = alpha - 1/3``, ``c = 1 / (3 sqrt(d))``, squeeze ``U < 1 - 0.0331 x^4`` or
    ``log U < x^2 / 2 + d (1 - v^3 + log v^3)``; alpha < 1 is boosted through ``log U' / alpha`` from the FIRST sub-key."""
    a = np.float64(alpha)
    boost = a >= 1.0
    a0 = a
    a = a if boost else a + 1.0
    d = a - np.float64(1.0 / 3.0)
    c = np.float64(1.0 / 3.0) / np.sqrt(d)
    key, subkey = split(key)
    X, V, U = np.float64(0.0), np.float64(1.0), np.float64(2.0)
    while True:
        cond = (U >= 1.0 - 0.0331 * (X * X)) and (np.log(U) >= X * 0.5 + d * ((1.0 - V) + np.log(V)))
        if not cond:
            break
        key, x_key, u_key = split(key, 3)
        kx, x, v = x_key, np.float64(0.0), np.float64(-1.0)
        while v <= 0.0:
            kx, sub = split(kx)
            x = np.float64(normal(sub, ()).reshape(())[()])
            v = 1.0 + x * c
        X, V = x * x, (v * v) * v
        U = np.float64(uniform(u_key, ()).reshape(())[()])
    log_samples = np.log1p(-np.float64(uniform(subkey, ()).reshape(())[()]))
    log_boost = 0.0 if (boost or log_samples == 0.0) else log_samples * (1.0 / a0)
    return (np.log(d) + log_boost) + np.log(V)


def dirichlet(key, alpha):
    """``jax.random.dirichlet``: ``softmax(loggamma(key, alpha))`` with one sub-key per component (``split(key, n)``)."""
    alpha = np.asarray(alpha, dtype=np.float64)
    keys = split(key, alpha.shape[0])
    ls = np.array([gamma_log(keys[i], alpha[i]) for i in range(alpha.shape[0])])
    w = np.exp(ls - ls.max())
    return w / w.sum()
