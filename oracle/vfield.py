"""VectorFieldNet: forward, x-JVP and parameter gradient (float64 compute, float32 params).

ORACLE (test infrastructure; see oracle/__init__.py).  Follows
``exe_flow_matching.py:56-90``.  Dense layers are numbered in flax ``@nn.compact``
creation order (``Dense_0..``): time branch, x branch, gate (zero kernel), joint
branch, output (zero kernel).  flax's ``Dense`` is ``y = x @ kernel + bias`` with
``kernel [in, out]``, float32 parameters promoted to the float64 inputs
(``multi_modal.py:14``).  PARITY UNPINNED for the initialiser's key derivation
(flax hashes module paths); the two output kernels are zero in any case so the
initial field is identically zero.
"""
import numpy as np

from . import prng

ACTS = {
    "relu": (lambda z: np.maximum(z, 0.0), lambda z: (z > 0).astype(z.dtype)),
    "tanh": (np.tanh, lambda z: 1.0 - np.tanh(z) ** 2),
}


def _sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


ACTS["swish"] = (lambda z: z * _sigmoid(z), lambda z: _sigmoid(z) * (1.0 + z * (1.0 - _sigmoid(z))))
ACTS["elu"] = (lambda z: np.where(z > 0, z, np.expm1(np.minimum(z, 0.0))),
               lambda z: np.where(z > 0, 1.0, np.exp(np.minimum(z, 0.0))))
_C = np.sqrt(2.0 / np.pi)
ACTS["gelu"] = (  # jax.nn.gelu default approximate=True
    lambda z: 0.5 * z * (1.0 + np.tanh(_C * (z + 0.044715 * z ** 3))),
    lambda z: 0.5 * (1.0 + np.tanh(_C * (z + 0.044715 * z ** 3)))
    + 0.5 * z * (1.0 - np.tanh(_C * (z + 0.044715 * z ** 3)) ** 2) * _C * (1.0 + 3 * 0.044715 * z * z),
)


class VectorFieldNet:
    """``exe_flow_matching.py:56-90``.

    ``dist`` supplies ``grad_logprob`` / ``hvp_logprob`` of the UNTEMPERED target
    (``:351``); ``grad_clip`` is ``args.gradient_clip if dim > 128 else None``."""

    def __init__(self, fourier_random, dist, hidden_x, hidden_t, hidden_xt, act="relu", grad_clip=None):
        self.f = np.asarray(fourier_random, dtype=np.float64)
        self.dist = dist
        self.hidden_x, self.hidden_t, self.hidden_xt = list(hidden_x), list(hidden_t), list(hidden_xt)
        self.act, self.dact = ACTS[act]
        self.grad_clip = grad_clip
        self.dim = dist.dim

    # ---- parameter structure -----------------------------------------------------------------
    def layer_shapes(self):
        F2, d = 2 * self.f.shape[0], self.dim
        shapes, prev = [], F2
        for h in self.hidden_t:
            shapes.append((prev, h)); prev = h
        ht = prev
        prev = d
        for h in self.hidden_x:
            shapes.append((prev, h)); prev = h
        hx = prev
        shapes.append((ht, d))                       # gate, zero kernel (:81)
        prev = hx + ht
        for h in self.hidden_xt:
            shapes.append((prev, h)); prev = h
        shapes.append((prev, d))                     # output, zero kernel (:86)
        return shapes

    def zero_layers(self):
        lt, lx, lxt = len(self.hidden_t), len(self.hidden_x), len(self.hidden_xt)
        return (lt + lx, lt + lx + 1 + lxt)

    def init(self, key):
        """lecun_normal kernels (truncated normal, variance 1/fan_in), zero biases; float32."""
        shapes = self.layer_shapes()
        keys = prng.split(key, len(shapes))
        params = []
        for i, (fi, fo) in enumerate(shapes):
            if i in self.zero_layers():
                W = np.zeros((fi, fo), dtype=np.float32)
            else:
                std = np.sqrt(1.0 / fi) / 0.87962566103423978
                W = (prng.truncated_normal(keys[i], -2.0, 2.0, (fi, fo)) * std).astype(np.float32)
            params.append({"kernel": W, "bias": np.zeros(fo, dtype=np.float32)})
        return params

    # ---- forward / jvp / backward --------------------------------------------------------------
    def _gterm(self, x):
        g = self.dist.grad_logprob(x)
        if self.grad_clip:
            return np.clip(g, -self.grad_clip, self.grad_clip), (np.abs(g) <= self.grad_clip)
        return g, None

    def forward(self, params, x, t, cache=False, tangent=None):
        """v(x, t) for x [B, d], t [B].  ``tangent`` [B, d] adds the x-JVP (returned second)."""
        lt, lx, lxt = len(self.hidden_t), len(self.hidden_x), len(self.hidden_xt)
        W = [p["kernel"].astype(np.float64) for p in params]
        b = [p["bias"].astype(np.float64) for p in params]
        degt = 2.0 * np.pi * self.f[None, :] * np.asarray(t, dtype=np.float64)[:, None]      # :70
        ffat = np.concatenate([np.cos(degt), np.sin(degt)], axis=1)                          # :71
        acts_in, pre = [], []
        li = 0
        s = ffat
        for _ in range(lt):                                                                  # :74-75
            acts_in.append(s); z = s @ W[li] + b[li]; pre.append(z); s = self.act(z); li += 1
        st = s
        s = x
        ts = tangent
        for _ in range(lx):                                                                  # :78-79
            acts_in.append(s); z = s @ W[li] + b[li]; pre.append(z); s = self.act(z)
            if ts is not None:
                ts = self.dact(z) * (ts @ W[li])
            li += 1
        sx = s
        acts_in.append(st); nn_t = st @ W[li] + b[li]; pre.append(nn_t); li += 1             # :81
        s = np.concatenate([sx, st], axis=1)                                                 # :83
        if ts is not None:
            ts = np.concatenate([ts, np.zeros_like(st)], axis=1)
        for _ in range(lxt):                                                                 # :84-85
            acts_in.append(s); z = s @ W[li] + b[li]; pre.append(z); s = self.act(z)
            if ts is not None:
                ts = self.dact(z) * (ts @ W[li])
            li += 1
        acts_in.append(s); nn_xt = s @ W[li] + b[li]; pre.append(nn_xt)                      # :86
        g, inside = self._gterm(x)
        v = nn_xt + nn_t * g                                                                 # :88-90
        out = [v]
        if tangent is not None:
            hv = self.dist.hvp_logprob(x, tangent)
            if inside is not None:
                hv = hv * inside
            out.append(ts @ W[li] + nn_t * hv)
        if cache:
            out.append((acts_in, pre, g))
        return out[0] if len(out) == 1 else tuple(out)

    def jacobian_trace_columns(self, params, x, t):
        """trace(d v / d x) by d forward-mode columns, one ``forward`` per basis vector: the literal reading of
        ``jnp.trace(jax.jacfwd(v)(x))`` (``exe_flow_matching.py:216-217``).  O(d) forwards; kept as the cross-check of
        :meth:`jacobian_trace` (``tests/test_oracle_vfield.py``)."""
        B, d = x.shape
        tr = np.zeros(B)
        for j in range(d):
            e = np.zeros_like(x); e[:, j] = 1.0
            _, jv = self.forward(params, x, t, tangent=e)
            tr += jv[:, j]
        return tr

    def jacobian_trace(self, params, x, t, block=256):
        """trace(d v / d x) (``exe_flow_matching.py:216-217``, ``:236-237``): what ``jax.jacfwd`` does -- ALL d basis tangents pushed
        through the layers at once (per chain a [d, h] tangent matrix, in blocks of ``block`` basis vectors to bound memory) on ONE
        value pass -- then the diagonal of the result.  Same arithmetic per column as :meth:`jacobian_trace_columns`."""
        lt, lx, lxt = len(self.hidden_t), len(self.hidden_x), len(self.hidden_xt)
        W = [p["kernel"].astype(np.float64) for p in params]
        _, (acts_in, pre, g) = self.forward(params, x, t, cache=True)
        B, d = x.shape
        nn_t = pre[lt + lx]
        _, inside = self._gterm(x)
        tr = np.zeros(B)
        for j0 in range(0, d, block):
            j1 = min(d, j0 + block)
            li = lt
            ts = np.broadcast_to(W[li][j0:j1][None], (B, j1 - j0, W[li].shape[1]))          # e_j W_x1: rows j0..j1 of the kernel
            ts = self.dact(pre[li])[:, None, :] * ts
            li += 1
            for _ in range(lx - 1):
                ts = self.dact(pre[li])[:, None, :] * (ts @ W[li]); li += 1
            li += 1                                                                           # gate layer: no x-tangent
            hx = ts.shape[2]
            for k in range(lxt):
                Wk = W[li][:hx] if k == 0 else W[li]                                          # the st half of the joint input has no tangent
                ts = self.dact(pre[li])[:, None, :] * (ts @ Wk); li += 1
            jv = ts @ W[li][:, j0:j1]                                                         # [B, blk, blk]: columns j0..j1 of the out layer
            tr += np.einsum("bjj->b", jv)
        # + the gate term: d/dx_j [nn_t_j clip(g_j(x))] = nn_t_j 1[|g_j| <= clip] H_jj
        if hasattr(self.dist, "hess_diag"):
            hd = self.dist.hess_diag(x)
        else:                                                                                 # (the 2-d mixtures)
            hd = np.stack([self.dist.hvp_logprob(x, np.broadcast_to(np.eye(d)[j][None], x.shape))[:, j] for j in range(d)], axis=1)
        if inside is not None:
            hd = hd * inside
        return tr + (nn_t * hd).sum(1)

    def backward(self, params, cache, dv):
        """Parameter gradients of sum(dv * v) given the forward cache; float32 like the params."""
        lt, lx, lxt = len(self.hidden_t), len(self.hidden_x), len(self.hidden_xt)
        acts_in, pre, g = cache
        W = [p["kernel"].astype(np.float64) for p in params]
        n = len(W)
        dW, db = [None] * n, [None] * n
        li = n - 1
        dz = dv                                                  # output layer
        dW[li] = acts_in[li].T @ dz; db[li] = dz.sum(0)
        ds = dz @ W[li].T
        for k in range(lxt):                                     # joint branch, reversed
            li -= 1
            dz = ds * self.dact(pre[li])
            dW[li] = acts_in[li].T @ dz; db[li] = dz.sum(0)
            ds = dz @ W[li].T
        hx = W[lt + lx - 1].shape[1] if lx else self.dim
        d_sx, d_st = ds[:, :hx], ds[:, hx:]
        li -= 1                                                  # gate layer
        dz = dv * g
        dW[li] = acts_in[li].T @ dz; db[li] = dz.sum(0)
        d_st = d_st + dz @ W[li].T
        ds = d_sx
        for k in range(lx):                                      # x branch, reversed
            li -= 1
            dz = ds * self.dact(pre[li])
            dW[li] = acts_in[li].T @ dz; db[li] = dz.sum(0)
            ds = dz @ W[li].T
        ds = d_st
        for k in range(lt):                                      # time branch, reversed
            li -= 1
            dz = ds * self.dact(pre[li])
            dW[li] = acts_in[li].T @ dz; db[li] = dz.sum(0)
            ds = dz @ W[li].T
        return [{"kernel": dW[i].astype(np.float32), "bias": db[i].astype(np.float32)} for i in range(n)]


def flat_params(params):
    """The canonical flat float32 vector (flax creation order; kernel [in][out] then bias, per layer) of a parameter list."""
    return np.concatenate([np.concatenate([p["kernel"].reshape(-1), p["bias"].reshape(-1)]) for p in params]).astype(np.float32)


def unflat_params(model, flat):
    """Inverse of ``flat_params`` for ``model``'s layer shapes."""
    out, o = [], 0
    for (fi, fo) in model.layer_shapes():
        W = flat[o:o + fi * fo].reshape(fi, fo); o += fi * fo
        b = flat[o:o + fo]; o += fo
        out.append({"kernel": W.astype(np.float32), "bias": b.astype(np.float32)})
    return out
