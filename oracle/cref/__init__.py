"""ctypes binding of ``libmfm_ref`` (``mfm_ref.c``): the float64 C / OpenMP restatement of the headline configuration's inner loop.

ORACLE (test infrastructure; see oracle/__init__.py): loaded by ``tests/`` (checked against the numpy restatement) and by
``bench.py``'s ``cpu_baseline`` leg (the timed CPU port), never by ``mfm_amd``.  PARITY UNPINNED like the rest of ``oracle/``.

``CRef(model, params)`` wraps an ``oracle.vfield.VectorFieldNet`` on a ``PhiFour`` target with relu activations; the methods mirror
the numpy functions they restate (``targets.Tempered.value_and_grad``, ``mala.kernel`` with given draws, ``VectorFieldNet.forward``,
``fm.loss_and_grad`` on a given batch, ``ode.transform_and_logdet`` / ``inverse_and_logdet`` with given Hutchinson probes).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .. import prng
from ..mala import MALAInfo, MALAState
from ..vfield import flat_params, unflat_params

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "_build", "libmfm_ref.so")
_lib = None


def build(force=False):
    """Compile ``mfm_ref.c`` (gcc, OpenMP) into ``oracle/_build/libmfm_ref.so`` unless it is up to date."""
    src = os.path.join(HERE, "mfm_ref.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-B", "-C", HERE], check=True, stdout=subprocess.DEVNULL)
    return LIB


class _Net(C.Structure):
    _fields_ = [("d", C.c_int), ("F", C.c_int), ("lt", C.c_int), ("lx", C.c_int), ("lxt", C.c_int),
                ("shapes", C.c_void_p), ("flat", C.c_void_p), ("fourier", C.c_void_p),
                ("grad_clip", C.c_double), ("coef", C.c_double), ("beta", C.c_double)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.mfmref_threads.restype = C.c_int
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class CRef:
    def __init__(self, model, params):
        dist = model.dist
        assert getattr(dist, "kind", None) == "phi4", "libmfm_ref covers the PhiFour target only"
        assert model.act(np.array([-1.0, 2.0])).tolist() == [0.0, 2.0], "libmfm_ref covers relu only"
        self.model, self.dist = model, dist
        self.lib = lib()
        self.d = int(model.dim)
        self._shapes = np.ascontiguousarray(np.array(model.layer_shapes(), dtype=np.int32))
        self._fourier = _f64(model.f)
        self.set_params(params)

    def set_params(self, params):
        self._flat = np.ascontiguousarray(flat_params(params), dtype=np.float32)
        m = self.model
        self._net = _Net(self.d, int(m.f.shape[0]), len(m.hidden_t), len(m.hidden_x), len(m.hidden_xt), _p(self._shapes), _p(self._flat),
                         _p(self._fourier), float(m.grad_clip or 0.0), float(self.dist.coef), float(self.dist.beta))

    @property
    def threads(self):
        return int(self.lib.mfmref_threads())

    def set_threads(self, n):
        self.lib.mfmref_set_threads(int(n))

    # targets.Tempered(dist, temper).value_and_grad
    def value_and_grad(self, x, temper=1.0):
        x = _f64(x); B = x.shape[0]
        logp, grad = np.empty(B), np.empty_like(x)
        assert self.lib.mfmref_phi4_value_grad(_p(x), B, self.d, C.c_double(self.dist.coef), C.c_double(self.dist.beta), C.c_double(temper), _p(logp), _p(grad)) == 0
        return logp, grad

    # mala.kernel with the Gaussian draws `noise` [B, d] and the uniforms `u` [B] given
    def mala_step(self, state, noise, u, step_size, temper=1.0, textbook=False):
        x, lp, g = (_f64(a).copy() for a in state)
        B = x.shape[0]
        p, acc = np.empty(B), np.empty(B, dtype=np.uint8)
        assert self.lib.mfmref_mala_step(_p(x), _p(lp), _p(g), _p(_f64(noise)), _p(_f64(u)), B, self.d, C.c_double(step_size), C.c_double(self.dist.coef),
                                         C.c_double(self.dist.beta), C.c_double(temper), int(bool(textbook)), _p(p), _p(acc)) == 0
        return MALAState(x, lp, g), p, acc.astype(bool)

    # VectorFieldNet.forward(params, x, t, tangent=...)
    def forward(self, x, t, tangent=None):
        x, t = _f64(x), _f64(t).reshape(-1); B = x.shape[0]
        v = np.empty_like(x)
        tg = _f64(tangent) if tangent is not None else None
        jv = np.empty_like(x) if tangent is not None else None
        assert self.lib.mfmref_vfield(C.byref(self._net), _p(x), _p(t), _p(tg), B, _p(v), _p(jv)) == 0
        return v if tangent is None else (v, jv)

    # fm.loss_and_grad on a batch (t, cond, target) already built from its draws
    def fm_loss_grad(self, t, cond, target):
        cond, target, t = _f64(cond), _f64(target), _f64(t).reshape(-1)
        loss = C.c_double(0.0)
        g = np.empty(self._flat.shape[0], dtype=np.float32)
        assert self.lib.mfmref_fm_loss_grad(C.byref(self._net), _p(cond), _p(target), _p(t), cond.shape[0], C.byref(loss), _p(g)) == 0
        return loss.value, unflat_params(self.model, g)

    # ode.transform_and_logdet (sign = +1) / inverse_and_logdet (-1), Hutchinson probes z [B, d] given, one output time (t = 1).
    # replay = dict(dt=[B, cap], acc=[B, cap]): the prescribed step sequence of ode.odeint's parity instrumentation; record = cap: the
    # chain's own sequence into stats["dt_seq"] [B, cap] / stats["acc_seq"] [B, cap] (zero past its last attempt), as ode.odeint records it
    def solve(self, x0, z, sign, rtol, atol, mxstep, stats=None, replay=None, record=0):
        x0, z = _f64(x0), _f64(z); B = x0.shape[0]
        xo, ldj, natt = np.empty_like(x0), np.empty(B), np.empty(B, dtype=np.int64)
        nev = C.c_longlong(0)
        rp_dt = rp_acc = rec_dt = rec_acc = None
        cap = 0
        if replay is not None:
            rp_dt = _f64(replay["dt"]); cap = rp_dt.shape[1]
            acc_in = np.asarray(replay["acc"]).astype(np.uint8)          # (numpy's sequences: A + 1 step sizes, A decisions)
            rp_acc = np.zeros((B, cap), dtype=np.uint8); rp_acc[:, :min(cap, acc_in.shape[1])] = acc_in[:, :cap]
            assert rp_dt.shape == (B, cap)
        elif record:
            cap = int(record)
            rec_dt = np.zeros((B, cap)); rec_acc = np.zeros((B, cap), dtype=np.uint8)
        assert self.lib.mfmref_cnf_solve(C.byref(self._net), _p(x0), _p(z), int(sign), C.c_double(rtol), C.c_double(atol), int(mxstep), B,
                                         _p(xo), _p(ldj), _p(natt), C.byref(nev), _p(rp_dt), _p(rp_acc), _p(rec_dt), _p(rec_acc), cap) == 0
        if stats is not None:
            stats["n_attempted"], stats["n_evals_total"] = natt, int(nev.value)
            if rec_dt is not None:
                assert natt.max() < cap, "record capacity too small"
                stats["dt_seq"], stats["acc_seq"] = rec_dt, rec_acc.astype(bool)
        return xo, ldj

    # mala.kernel: one key per chain (mala.py:93 key_integrator, key_rmh), draws by oracle/prng.py, arithmetic in C
    def mala_kernel(self, keys, state, step_size, temper=1.0, textbook=False):
        kk = prng.split_rows(keys, 2)
        noise = prng.normal_rows(kk[:, 0], self.d)                   # util.py:80-82
        u = prng.uniform_rows(kk[:, 1])
        st, p, acc = self.mala_step(state, noise, u, step_size, temper, textbook)
        return st, MALAInfo(p, acc, None, None)

    # flow.rwmh_step (exe_flow_matching.py:264-278): keys and draws by oracle/prng.py, the two CNF solves and the target in C
    def rwmh_step(self, keys, prev, args, temper=1.0, stats=None, replay=None, record=0):
        """``replay = dict(inv=..., fwd=...)`` / ``record``: see ``solve`` (``stats["inv"]`` / ``stats["fwd"]`` carry the recorded sequences)."""
        d = self.d
        kk = prng.split_rows(keys, 4)                                # :265 key_gen, key_acc, key_hutch1, key_hutch2
        o = (args.rtol, args.atol, args.mxstep)
        si, sf = {}, {}
        rp = replay or {}
        u0, vol0 = self.solve(prev.position, prng.normal_rows(kk[:, 3], d), -1, *o, stats=si, replay=rp.get("inv"), record=record)     # :267
        up = u0 + (2.38 / np.sqrt(d)) * prng.normal_rows(kk[:, 0], d)                                   # :262,268
        xp, volp = self.solve(up, prng.normal_rows(kk[:, 2], d), +1, *o, stats=sf, replay=rp.get("fwd"), record=record)              # :269
        lpn, gn = self.value_and_grad(xp, temper)                                                       # :270
        with np.errstate(over="ignore", invalid="ignore"):
            a = np.exp(lpn - volp - prev.logdensity - vol0)                                             # :271-274
            acc = prng.uniform_rows(kk[:, 1]) <= a                                                      # :275 (NaN: reject)
        m = acc[:, None]
        if stats is not None:
            stats.update(n_att_inv=si["n_attempted"], n_att_fwd=sf["n_attempted"], u0=u0, vol0=vol0, up=up, volp=volp,
                         log_alpha=lpn - volp - prev.logdensity - vol0, inv=si, fwd=sf)
        state = MALAState(np.where(m, xp, prev.position), np.where(acc, lpn, prev.logdensity), np.where(m, gn, prev.logdensity_grad))
        return state, MALAInfo(a, acc, xp, np.zeros_like(a))
