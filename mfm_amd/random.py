"""Host-side key plumbing with jax.random conventions (threefry2x32, 64-bit draws under x64).

Only what the host needs: deriving / splitting keys, one-off draws at set-up time (chain initialisation, Fourier
frequencies, network initialisation, exact mixture samples).  Per-iteration noise is drawn inside the HIP kernels
(mfm_amd/csrc/prng.hip.h) with the same conventions, so a key means the same stream on both sides.
Reference call sites: exe_flow_matching.py:333,350,353,371-372,433; distributions.py:70-76,163-164.
"""
import numpy as np
from scipy.special import erf, erfinv

U32 = np.uint32


def _rot(x, r):
    return (x << U32(r)) | (x >> U32(32 - r))


def _threefry(k0, k1, x0, x1):
    with np.errstate(over="ignore"):
        k0, k1 = U32(k0) if np.isscalar(k0) else k0.astype(U32), U32(k1) if np.isscalar(k1) else k1.astype(U32)
        ks = (k0, k1, k0 ^ k1 ^ U32(0x1BD11BDA))
        x0 = (np.asarray(x0, U32) + ks[0]).astype(U32)
        x1 = (np.asarray(x1, U32) + ks[1]).astype(U32)
        rots = ((13, 15, 26, 6), (17, 29, 16, 24))
        for b in range(5):
            for r in rots[b % 2]:
                x0 = (x0 + x1).astype(U32)
                x1 = _rot(x1, r) ^ x0
            x0 = (x0 + ks[(b + 1) % 3]).astype(U32)
            x1 = (x1 + ks[(b + 2) % 3] + U32(b + 1)).astype(U32)
    return x0, x1


def PRNGKey(seed):
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=U32)


def split(key, num=2):
    cnt = np.arange(2 * num, dtype=U32)
    y0, y1 = _threefry(key[0], key[1], cnt[:num], cnt[num:])
    return np.concatenate([y0, y1]).reshape(num, 2)


def split_rows(keys, num):
    """vmap(lambda k: split(k, num))(keys): keys [n, 2] -> [n, num, 2]."""
    keys = np.atleast_2d(np.asarray(keys, U32))
    cnt = np.arange(2 * num, dtype=U32)[None, :]
    y0, y1 = _threefry(keys[:, 0:1], keys[:, 1:2], np.broadcast_to(cnt[:, :num], (keys.shape[0], num)),
                       np.broadcast_to(cnt[:, num:], (keys.shape[0], num)))
    return np.concatenate([y0, y1], axis=1).reshape(keys.shape[0], num, 2)


def _bits64(keys, size):
    """keys [n, 2] -> [n, size] uint64 (one independent draw of `size` samples per key)."""
    keys = np.atleast_2d(np.asarray(keys, U32))
    i = np.arange(size, dtype=U32)[None, :]
    y0, y1 = _threefry(keys[:, 0:1], keys[:, 1:2], np.broadcast_to(i, (keys.shape[0], size)),
                       np.broadcast_to((i + U32(size)).astype(U32), (keys.shape[0], size)))
    return (y0.astype(np.uint64) << np.uint64(32)) | y1.astype(np.uint64)


def _unit(bits):
    return ((bits >> np.uint64(12)) | np.float64(1.0).view(np.uint64)).view(np.float64) - 1.0


def uniform(key, shape=(), minval=0.0, maxval=1.0):
    shape = tuple(np.atleast_1d(shape)) if np.ndim(shape) or shape != () else ()
    size = int(np.prod(shape)) if shape else 1
    u = np.maximum(minval, _unit(_bits64(key, size))[0] * (maxval - minval) + minval)
    return u.reshape(shape)


_LO = np.nextafter(np.float64(-1.0), 0.0)


def normal(key, shape=()):
    return np.sqrt(2.0) * erfinv(uniform(key, shape, _LO, 1.0))


def uniform_rows(keys, d):
    return np.maximum(0.0, _unit(_bits64(keys, d)))


def normal_rows(keys, d):
    u = np.maximum(_LO, _unit(_bits64(keys, d)) * (1.0 - _LO) + _LO)
    return np.sqrt(2.0) * erfinv(u)


def truncated_normal(key, lower, upper, shape):
    a, b = erf(lower / np.sqrt(2.0)), erf(upper / np.sqrt(2.0))
    out = np.sqrt(2.0) * erfinv(uniform(key, shape, a, b))
    return np.clip(out, np.nextafter(lower, np.inf), np.nextafter(upper, -np.inf))


def _gamma_one_log(key, alpha):
    """log of one Gamma(alpha, 1) draw: ``jax._src.random._gamma_one(key, alpha, log_space=True)`` @ jax 0.4.26 (third-party,
    restated from the published source: Marsaglia & Tsang's squeeze / rejection loop on threefry sub-keys, float64).  Key
    plumbing: ``key, subkey = split(key)`` (the boost draw, used when alpha < 1), then per rejection round
    ``key, x_key, U_key = split(key, 3)``; the proposal ``x`` is redrawn from ``split(x_key)`` while ``1 + c x <= 0``."""
    alpha = float(alpha)
    boost = alpha >= 1.0
    alpha_orig = alpha
    if not boost:
        alpha = alpha + 1.0
    d = alpha - 1.0 / 3.0
    c = (1.0 / 3.0) / np.sqrt(d)
    key, subkey = split(key)
    X, V, U = 0.0, 1.0, 2.0                               # initial state: the loop condition holds
    while U >= 1.0 - 0.0331 * (X * X) and np.log(U) >= X * 0.5 + d * ((1.0 - V) + np.log(V)):
        key, x_key, U_key = split(key, 3)
        kx, x, v = x_key, 0.0, -1.0
        while v <= 0.0:
            kx, sk = split(kx)
            x = float(normal(sk, ()))
            v = 1.0 + x * c
        X, V, U = x * x, (v * v) * v, float(uniform(U_key, ()))
    log_samples = np.log1p(-float(uniform(subkey, ())))   # -jax.random.exponential(subkey)
    log_boost = 0.0 if (boost or log_samples == 0.0) else log_samples * (1.0 / alpha_orig)
    return (np.log(d) + log_boost) + np.log(V)


def dirichlet(key, alpha):
    """``jax.random.dirichlet(key, alpha)`` for a 1-d ``alpha`` (multi_modal.py:45): log-space gamma draws on ``split(key, n)``
    (``_gamma_impl``), then a softmax (``_dirichlet``)."""
    alpha = np.asarray(alpha, dtype=np.float64)
    keys = split(key, alpha.shape[0])
    ls = np.array([_gamma_one_log(keys[i], alpha[i]) for i in range(alpha.shape[0])])
    w = np.exp(ls - ls.max())
    return w / w.sum()


def random_bits32(key, size):
    """``jax._src.prng.threefry_random_bits`` for 32-bit words: ``threefry_2x32(key, iota(size))`` (odd sizes padded)."""
    odd = size % 2
    cnt = np.arange(size + odd, dtype=np.uint32)
    if odd:
        cnt[-1] = 0
    half = (size + odd) // 2
    y0, y1 = _threefry(np.uint32(key[0]), np.uint32(key[1]), cnt[:half], cnt[half:])
    out = np.concatenate([y0, y1])
    return out[:-1] if odd else out


def permutation_indices(key, n):
    """Indices p with ``jax.random.permutation(key, x) == x[p]`` (jax 0.4.26 ``_shuffle``: rounds of a STABLE sort by fresh 32-bit
    keys, ``ceil(3 ln n / ln(2^32 - 1))`` of them)."""
    p = np.arange(n)
    rounds = int(np.ceil(3 * np.log(max(1, n)) / np.log(np.iinfo(np.uint32).max)))
    for _ in range(rounds):
        key, sub = split(key)
        p = p[np.argsort(random_bits32(sub, n), kind="stable")]
    return p
