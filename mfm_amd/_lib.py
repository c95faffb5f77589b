"""ctypes binding of libmfm_hip (include/mfm.h) -- the reference-side stub a maintainer adds.

Every device buffer is a ``torch.Tensor`` on the GPU; only its ``data_ptr()`` crosses the boundary.  The library
fails loudly: a missing ``.so`` raises at load, a missing GPU raises at ``Context`` creation, and every non-zero
status code becomes ``MfmError`` carrying ``mfm_last_error()``.
"""
import ctypes as C
import os

import numpy as np

from .build import LIB

PHI4, GMM, LGCP = 0, 1, 2
FLOW_RWMH, FLOW_IMH = 0, 1
FAMILY_AUTO, FAMILY_TILE, FAMILY_WIDE = 0, 1, 2
ACTIVATIONS = {"relu": 0, "tanh": 1, "elu": 2, "gelu": 3, "swish": 4}          # exe_flow_matching.py:39-45


class MfmError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("fourier_dim", C.c_int32),
        ("hidden_t", C.c_int32 * 2), ("hidden_x", C.c_int32 * 2), ("hidden_xt", C.c_int32 * 2),
        ("n_chain_local", C.c_int32), ("n_chain_total", C.c_int32), ("chain_offset", C.c_int32),
        ("grad_clip", C.c_float), ("sigma", C.c_float), ("cond_flow", C.c_int32), ("hutch", C.c_int32),
        ("rtol", C.c_double), ("atol", C.c_double), ("mxstep", C.c_int32), ("n_ts", C.c_int32),
        ("learning_rate", C.c_double), ("adam_b1", C.c_double), ("adam_b2", C.c_double), ("adam_eps", C.c_double),
        ("weight_decay", C.c_double), ("update_clip", C.c_double),
        ("learning_iter", C.c_int32), ("warmup_steps", C.c_int32), ("max_eval_samples", C.c_int32),
        ("kernel_family", C.c_int32), ("activation", C.c_int32), ("ref_std", C.c_double), ("n_chain_valid", C.c_int32),
        ("depth_t", C.c_int32), ("depth_x", C.c_int32), ("depth_xt", C.c_int32),
        ("hidden_t3", C.c_int32), ("hidden_x3", C.c_int32), ("hidden_xt3", C.c_int32),
        ("ode_method", C.c_int32), ("ode_steps", C.c_int32),
    ]


MAX_DEPTH = 3          # include/mfm.h: MFM_MAX_DEPTH
ODE_METHODS = {"dopri5": 0, "rk4": 1, "euler": 2}      # include/mfm.h: MFM_ODE_*


_P = C.c_void_p
_U32 = C.c_uint32
_SIGS = {
    "mfm_last_error": (C.c_char_p, []),
    "mfm_version": (C.c_int, []),
    "mfm_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "mfm_destroy": (C.c_int, [_P]),
    "mfm_set_stream": (C.c_int, [_P, _P]),
    "mfm_sync": (C.c_int, [_P]),
    "mfm_num_params": (C.c_int, [_P]),
    "mfm_set_target": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.c_size_t]),
    "mfm_set_fourier": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "mfm_set_params": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "mfm_get_params": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "mfm_reset_optimizer": (C.c_int, [_P]),
    "mfm_mala_init": (C.c_int, [_P, _P, C.c_double, _P, _P]),
    "mfm_mala_step": (C.c_int, [_P, _U32, _U32, C.c_double, C.c_double, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "mfm_mala_step_keys": (C.c_int, [_P, _P, C.c_double, C.c_double, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "mfm_hmc_step": (C.c_int, [_P, _U32, _U32, C.c_double, C.c_double, C.c_int, _P, _P, _P, _P, _P]),
    "mfm_loglik": (C.c_int, [_P, _P, _P]),
    "mfm_smc_delta": (C.c_int, [_P, _P, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "mfm_smc_weights": (C.c_int, [_P, _P, C.c_int, C.c_double, _P, C.POINTER(C.c_double)]),
    "mfm_smc_resample": (C.c_int, [_P, _U32, _U32, _P, C.c_int, _P, _P]),
    "mfm_smc_resample_scheme": (C.c_int, [_P, C.c_int, _U32, _U32, _P, C.c_int, _P, _P]),
    "mfm_gather_rows": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "mfm_acc_stats": (C.c_int, [_P, _P, C.c_int, _P]),
    "mfm_choice_logw": (C.c_int, [_P, _U32, _U32, _P, C.c_int, C.c_int, _P, _P]),
    "mfm_fm_loss_grad": (C.c_int, [_P, _U32, _U32, _P, _P, _P]),
    "mfm_fm_loss": (C.c_int, [_P, _U32, _U32, _P, C.c_int, C.c_int, C.c_int, _P]),
    "mfm_adamw_step": (C.c_int, [_P, _P]),
    "mfm_comm_unique_id": (C.c_int, [_P]),
    "mfm_comm_init": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "mfm_comm_destroy": (C.c_int, [_P]),
    "mfm_comm_count": (C.c_int, [_P, _P]),
    "mfm_grad_allreduce_begin": (C.c_int, [_P, _P]),
    "mfm_opt_state": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "mfm_vf_apply": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P]),
    "mfm_ode_transform": (C.c_int, [_P, C.c_int, C.c_int, _P, _U32, _U32, _P, C.c_int, _P, _P, _P]),
    "mfm_flow_step": (C.c_int, [_P, C.c_int, _U32, _U32, C.c_double, _P, _P, _P, _P, _P, _P, _P]),
    "mfm_train_iter": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _U32, _U32, _U32, _U32, C.c_double, C.c_double,
                                 _P, _P, _P, _P, _P, _P, _P, C.c_int]),
    "mfm_beta_update": (C.c_int, [_P, C.c_double, _P, C.c_int, C.c_double, C.POINTER(C.c_double)]),
    "mfm_normal_rows": (C.c_int, [_P, _P, C.c_int, _P]),
    "mfm_cis_select": (C.c_int, [_P, _U32, _U32, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfm_stein_disc": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.POINTER(C.c_double)]),
    "mfm_max_mean_disc": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_double)]),
    "mfm_noise_prefetch": (C.c_int, [_P, C.c_int, C.POINTER(_U32), C.POINTER(_U32)]),
    "mfm_noise_drop": (C.c_int, [_P]),
    "mfm_get_counters": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "mfm_reset_counters": (C.c_int, [_P]),
    "mfm_debug_replay": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "mfm_profile": (C.c_int, [_P, C.c_int]),
    "mfm_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mfm_pack_index": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mfm_pack_index_T": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mfm_threefry2x32": (C.c_int, [_U32, _U32, _U32, _U32, C.POINTER(_U32)]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Load libmfm_hip.so (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise MfmError(f"{LIB} not found: build it with `python -m mfm_amd.build` (hipcc, gfx950)")
        # torch first: libmfm_hip.so depends on libamdhip64 by SONAME, and PyTorch-ROCm bundles its own copy.  Loaded after
        # torch the dependency resolves to the runtime torch already initialised (one HIP runtime per process: device
        # pointers and streams are shared with torch); loaded BEFORE torch it would bind /opt/rocm's copy, and the second
        # runtime in the process finds no device ("no HIP device available" from mfm_create).
        import torch  # noqa: F401
        lib = C.CDLL(os.environ.get("MFM_LIB", LIB))      # MFM_LIB: development override (A/B of kernel variants)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _chk(rc):
    if rc != 0:
        raise MfmError(f"libmfm_hip error {rc}: {load().mfm_last_error().decode()}")


def _ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA tensor (dtype-checked: a wrong dtype would be silent garbage)."""
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous()):
        raise MfmError("expected a contiguous CUDA tensor")
    if dtype is not None and str(t.dtype).split(".")[-1] != dtype:
        raise MfmError(f"expected a {dtype} tensor, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


F32, F64, U8, I32 = "float32", "float64", "uint8", "int32"


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Context:
    """Owner of one ``mfm_ctx`` (one GPU, one process)."""

    def __init__(self, **kw):
        import torch
        self.lib = load()
        if not torch.cuda.is_available():
            raise MfmError("no GPU visible: the MFM hot path has no CPU fallback")
        cfg = Config()
        defaults = dict(fourier_dim=128, hidden_t=(128, 128), hidden_x=(128, 128), hidden_xt=(128, 128),
                        chain_offset=0, grad_clip=0.0, sigma=1e-4, cond_flow=1, hutch=0, rtol=1e-5, atol=1e-5,
                        mxstep=1000, n_ts=2, learning_rate=1e-3, adam_b1=0.9, adam_b2=0.999, adam_eps=1e-8,
                        weight_decay=1e-4, update_clip=1.0, learning_iter=400, warmup_steps=0, max_eval_samples=0,
                        kernel_family=int(os.environ.get("MFM_KERNEL_FAMILY", FAMILY_AUTO)), activation=0)
        defaults.update(kw)
        defaults.setdefault("n_chain_total", defaults["n_chain_local"])
        for k, v in defaults.items():
            if k in ("hidden_t", "hidden_x", "hidden_xt"):      # lists of 1 .. MAX_DEPTH widths (exe_flow_matching.py:74-85)
                hs = [int(h) for h in v]
                if not 1 <= len(hs) <= MAX_DEPTH:
                    raise MfmError(f"{k}: {len(hs)} hidden layers; the kernels take 1 to {MAX_DEPTH} per branch")
                setattr(cfg, k, (C.c_int32 * 2)(*(hs + hs)[:2]))
                setattr(cfg, "depth_" + k[7:], len(hs))
                setattr(cfg, k + "3", hs[2] if len(hs) > 2 else 0)
            else:
                setattr(cfg, k, v)
        self.cfg = cfg
        self.dim, self.n_local = int(cfg.dim), int(cfg.n_chain_local)
        h = _P()
        _chk(self.lib.mfm_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.n_params = self.lib.mfm_num_params(h)
        self.use_current_stream()

    before_params = None      # set by the engine: called before any call that reads or writes the network parameters
    # (a deferred optimizer step whose gradient all-reduce is still in flight is applied there)

    def _p(self):
        if self.before_params is not None:
            self.before_params()

    def close(self):
        if getattr(self, "h", None):
            self.lib.mfm_destroy(self.h)
            self.h = None

    __del__ = close

    def use_current_stream(self):
        import torch
        _chk(self.lib.mfm_set_stream(self.h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def sync(self):
        _chk(self.lib.mfm_sync(self.h))

    # ---- target / parameters ------------------------------------------------------------------------------
    def set_target(self, kind, params):
        p = np.ascontiguousarray(params, dtype=np.float64)
        _chk(self.lib.mfm_set_target(self.h, kind, p.ctypes.data_as(C.POINTER(C.c_double)), p.size))

    def set_fourier(self, f):
        f = _f32(f)
        assert f.size == self.cfg.fourier_dim
        _chk(self.lib.mfm_set_fourier(self.h, f.ctypes.data_as(C.POINTER(C.c_float))))

    def set_params(self, flat):
        self._p()
        flat = _f32(flat)
        assert flat.size == self.n_params, (flat.size, self.n_params)
        _chk(self.lib.mfm_set_params(self.h, flat.ctypes.data_as(C.POINTER(C.c_float))))

    def get_params(self):
        self._p()
        out = np.empty(self.n_params, dtype=np.float32)
        _chk(self.lib.mfm_get_params(self.h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def reset_optimizer(self):
        self._p()
        _chk(self.lib.mfm_reset_optimizer(self.h))

    # ---- kernels ----------------------------------------------------------------------------------------------
    def mala_init(self, pos, beta, logp, grad):
        _chk(self.lib.mfm_mala_init(self.h, _ptr(pos, F32), float(beta), _ptr(logp, F64), _ptr(grad, F32)))

    def mala_step(self, key, beta, step_size, pos, logp, grad, acc=None, is_acc=None, proposed=None, weight=None,
                  textbook=False):
        _chk(self.lib.mfm_mala_step(self.h, int(key[0]), int(key[1]), float(beta), float(step_size), int(textbook),
                                    _ptr(pos, F32), _ptr(logp, F64), _ptr(grad, F32), _ptr(acc, F32), _ptr(is_acc, U8),
                                    _ptr(proposed, F32), _ptr(weight, F32)))

    def mala_step_keys(self, keys, beta, step_size, pos, logp, grad, acc=None, is_acc=None, proposed=None, weight=None,
                       textbook=False):
        """``keys``: int32/uint32 device tensor [n_chain_local, 2], one key per chain (the caller vmaps over its own keys)."""
        _chk(self.lib.mfm_mala_step_keys(self.h, _ptr(keys, I32), float(beta), float(step_size), int(textbook),
                                         _ptr(pos, F32), _ptr(logp, F64), _ptr(grad, F32), _ptr(acc, F32), _ptr(is_acc, U8),
                                         _ptr(proposed, F32), _ptr(weight, F32)))

    def hmc_step(self, key, beta, step_size, num_steps, pos, logp, grad, acc=None, is_acc=None):
        """Build-side mode (``mfm_hmc_step``): one HMC step of every local chain, state updated in place."""
        _chk(self.lib.mfm_hmc_step(self.h, int(key[0]), int(key[1]), float(beta), float(step_size), int(num_steps),
                                   _ptr(pos, F32), _ptr(logp, F64), _ptr(grad, F32), _ptr(acc, F32), _ptr(is_acc, U8)))

    def loglik(self, pos, out):
        _chk(self.lib.mfm_loglik(self.h, _ptr(pos, F32), _ptr(out, F64)))

    # ---- adaptive tempered SMC pieces (bblackjax/smc) -----------------------------------------------------------
    def smc_delta(self, logliks, target_ess, max_delta):
        out = C.c_double()
        _chk(self.lib.mfm_smc_delta(self.h, _ptr(logliks, F64), logliks.shape[0], float(target_ess), float(max_delta), C.byref(out)))
        return out.value

    def smc_weights(self, logliks, delta, weights):
        out = C.c_double()
        _chk(self.lib.mfm_smc_weights(self.h, _ptr(logliks, F64), logliks.shape[0], float(delta), _ptr(weights, F64), C.byref(out)))
        return out.value

    def smc_resample(self, key, weights, scratch, idx):
        _chk(self.lib.mfm_smc_resample(self.h, int(key[0]), int(key[1]), _ptr(weights, F64), weights.shape[0], _ptr(scratch, F64), _ptr(idx, I32)))

    def choice_logw(self, key, logw, m, scratch, idx):
        """idx[j] = jax.random.choice(key, n, (m,), p=exp(logw - max logw))[j] (``exe_flow_matching.py:458-459``)."""
        _chk(self.lib.mfm_choice_logw(self.h, int(key[0]), int(key[1]), _ptr(logw, F64), logw.shape[0], int(m), _ptr(scratch, F64), _ptr(idx, I32)))

    def acc_stats(self, x, out):
        """out[0:2] (float64) = sum, sum of squares of the float32 vector x."""
        _chk(self.lib.mfm_acc_stats(self.h, _ptr(x, F32), x.shape[0], _ptr(out, F64)))

    def smc_resample_scheme(self, scheme, key, weights, scratch, idx):
        """scheme: 0 systematic, 1 stratified, 2 multinomial (resampling.py); scratch: 2 n + 2 float64."""
        _chk(self.lib.mfm_smc_resample_scheme(self.h, int(scheme), int(key[0]), int(key[1]), _ptr(weights, F64), weights.shape[0],
                                              _ptr(scratch, F64), _ptr(idx, I32)))

    def gather_rows(self, src, idx, dst):
        """dst[r] = src[idx[r]] for the ``idx.shape[0]`` rows of ``dst`` (``src`` may hold more rows: the all-gathered particles of a sharded SMC step)."""
        _chk(self.lib.mfm_gather_rows(self.h, _ptr(src, F32), _ptr(idx, I32), idx.shape[0], src.shape[1], _ptr(dst, F32)))

    def fm_loss_grad(self, key, pos, loss, grads):
        self._p()
        _chk(self.lib.mfm_fm_loss_grad(self.h, int(key[0]), int(key[1]), _ptr(pos, F32), _ptr(loss, F64), _ptr(grads, F32)))

    def fm_loss(self, key, samples, loss, n_total=None, offset=0):
        self._p()
        n = samples.shape[0]
        _chk(self.lib.mfm_fm_loss(self.h, int(key[0]), int(key[1]), _ptr(samples, F32), n, n if n_total is None else n_total,
                                  offset, _ptr(loss, F64)))

    def adamw_step(self, grads):
        self._p()
        _chk(self.lib.mfm_adamw_step(self.h, _ptr(grads, F32)))

    # ---- context-owned RCCL communicator (include/mfm.h: mfm_comm_*) ---------------------------------------------
    def comm_unique_id(self):
        """128 bytes from ncclGetUniqueId (rank 0; ship them to the other ranks out of band)."""
        buf = (C.c_uint8 * 128)()
        _chk(self.lib.mfm_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _chk(self.lib.mfm_comm_init(self.h, int(nranks), int(rank), buf))
        self.has_comm = True

    def comm_destroy(self):
        _chk(self.lib.mfm_comm_destroy(self.h))
        self.has_comm = False

    def comm_count(self):
        """Ranks of the context's communicator as RCCL reports them (ncclCommCount); 0 without one."""
        n = C.c_int32(0)
        _chk(self.lib.mfm_comm_count(self.h, C.byref(n)))
        return int(n.value)

    def grad_allreduce_begin(self, grads):
        """Asynchronous SUM all-reduce of the gradient on the context's communication stream; the next adamw_step(grads) waits."""
        _chk(self.lib.mfm_grad_allreduce_begin(self.h, _ptr(grads, F32)))

    def opt_state(self):
        self._p()
        out = (C.c_int32 * 4)()
        lr = C.c_float()
        _chk(self.lib.mfm_opt_state(self.h, out, C.byref(lr)))
        return dict(step=out[0], count=out[1], notfinite_count=out[2], last_applied=out[3], last_lr=lr.value)

    def vf_apply(self, x, t, v, tangent=None, jvp=None):
        self._p()
        _chk(self.lib.mfm_vf_apply(self.h, _ptr(x, F32), _ptr(t, F32), _ptr(tangent, F32), x.shape[0], _ptr(v, F32), _ptr(jvp, F32)))

    def ode_transform(self, direction, x, out, ldj, keys=None, key=(0, 0), nsteps=None):
        self._p()
        _chk(self.lib.mfm_ode_transform(self.h, direction, 0 if keys is None else 1, _ptr(keys, I32), int(key[0]), int(key[1]),
                                        _ptr(x, F32), x.shape[0], _ptr(out, F32), _ptr(ldj, F32), _ptr(nsteps, I32)))

    def flow_step(self, mode, key, beta, pos, logp, grad, acc=None, is_acc=None, proposed=None, nsteps=None):
        self._p()
        _chk(self.lib.mfm_flow_step(self.h, mode, int(key[0]), int(key[1]), float(beta), _ptr(pos, F32), _ptr(logp, F64), _ptr(grad, F32),
                                    _ptr(acc, F32), _ptr(is_acc, U8), _ptr(proposed, F32), _ptr(nsteps, I32)))

    def train_iter(self, count, mcmc_per_flow_steps, flow_mode, key_gen, key_train, beta, step_size, pos, logp, grad, loss, grads,
                   acc=None, nsteps=None, apply_update=True):
        """One loop iteration (exe_flow_matching.py:432-439): generator (flow-MH step when ``count % (K+1) == 0``, MALA step
        otherwise) + train_step on the new positions, in one library call."""
        self._p()
        _chk(self.lib.mfm_train_iter(self.h, int(count), int(mcmc_per_flow_steps), int(flow_mode), int(key_gen[0]), int(key_gen[1]),
                                     int(key_train[0]), int(key_train[1]), float(beta), float(step_size), _ptr(pos, F32), _ptr(logp, F64),
                                     _ptr(grad, F32), _ptr(acc, F32), _ptr(nsteps, I32), _ptr(loss, F64), _ptr(grads, F32),
                                     1 if apply_update else 0))

    def noise_prefetch(self, keys_gn, keys_step):
        """Produce the draws of the iterations keyed ``keys_gn[j]`` (MALA step) / ``keys_step[j]`` (training batch) on the side
        stream; returns False where the configuration is not served (the kernels then draw in line as always)."""
        kg = np.ascontiguousarray(keys_gn, dtype=np.uint32).reshape(-1, 2)
        ks = np.ascontiguousarray(keys_step, dtype=np.uint32).reshape(-1, 2)
        assert kg.shape == ks.shape
        if kg.shape[0] == 0:
            return False
        rc = self.lib.mfm_noise_prefetch(self.h, kg.shape[0], kg.ctypes.data_as(C.POINTER(_U32)), ks.ctypes.data_as(C.POINTER(_U32)))
        if rc == -2:            # MFM_EUNSUPPORTED
            return False
        _chk(rc)
        return True

    def noise_drop(self):
        _chk(self.lib.mfm_noise_drop(self.h))

    COUNTERS = ("mala_chain_steps", "fm_train_samples", "fm_eval_samples", "ode_solves", "dopri_attempts", "field_evals",
                "optimizer_steps", "mala_hbm_bytes")

    def counters(self):
        """Algorithmic work done through this context since creation / ``reset_counters`` (``mfm_get_counters``)."""
        out = (C.c_int64 * 8)()
        _chk(self.lib.mfm_get_counters(self.h, out))
        return dict(zip(self.COUNTERS, [int(v) for v in out]))

    def reset_counters(self):
        _chk(self.lib.mfm_reset_counters(self.h))

    def debug_replay(self, dt, acc, ratio, dt_own, diag=None):
        """Arm the next ``ode_transform`` / ``flow_step`` with a prescribed step sequence (parity instrumentation,
        ``mfm_debug_replay``): float32 ``dt``, uint8 ``acc``, float32 outputs ``ratio`` / ``dt_own``, all CUDA tensors of shape
        [n, cap] (transform) or [2, n, cap] (flow step)."""
        cap = dt.shape[-1]
        for t_ in (acc, ratio, dt_own):
            assert t_.shape == dt.shape
        self._replay_keep = (dt, acc, ratio, dt_own, diag)      # the library holds raw pointers until the armed call has run
        _chk(self.lib.mfm_debug_replay(self.h, cap, _ptr(dt, F32), _ptr(acc, U8), _ptr(ratio, F32), _ptr(dt_own, F32), _ptr(diag, F64)))

    PROF_CLASSES = ("mala_step", "fm_fwd_bwd", "wgrad", "adamw", "flow_step", "fm_eval", "reduce", "_")

    def profile(self, enable=True, classes=None):
        """HIP-event timing of the kernel classes named in ``classes`` (default: all) on the context's stream."""
        mask = 0 if not enable else (-1 if classes is None else sum(1 << self.PROF_CLASSES.index(c) for c in classes))
        _chk(self.lib.mfm_profile(self.h, mask))

    def profile_read(self):
        ms, cnt = (C.c_double * 8)(), (C.c_int64 * 8)()
        _chk(self.lib.mfm_profile_read(self.h, ms, cnt))
        return {n: dict(ms=ms[i], launches=cnt[i]) for i, n in enumerate(self.PROF_CLASSES) if cnt[i]}

    def normal_rows(self, keys, out):
        """out[i] = normal(keys[i], (dim,)); keys: int32 CUDA tensor [n, 2] holding uint32 bit patterns."""
        _chk(self.lib.mfm_normal_rows(self.h, _ptr(keys, I32), out.shape[0], _ptr(out, F32)))

    def cis_select(self, key, n_is, u0, vol0, refs, xs, vols, lps, pos, logp, acc=None, is_acc=None, proposed=None, weight=None):
        _chk(self.lib.mfm_cis_select(self.h, int(key[0]), int(key[1]), int(n_is), _ptr(u0, F32), _ptr(vol0, F32), _ptr(refs, F32),
                                     _ptr(xs, F32), _ptr(vols, F32), _ptr(lps, F64), _ptr(pos, F32), _ptr(logp, F64), _ptr(acc, F32),
                                     _ptr(is_acc, U8), _ptr(proposed, F32), _ptr(weight, F32)))

    def stein_disc(self, x, grad, beta=-0.5):
        """(U, V) statistics of ``mcmc_utils.py:28-85`` for CUDA samples x [n, d] and grad log p at x."""
        out = (C.c_double * 2)()
        _chk(self.lib.mfm_stein_disc(self.h, _ptr(x, F32), _ptr(grad, F32), x.shape[0], float(beta), out))
        return out[0], out[1]

    def max_mean_disc(self, x, y):
        """``mcmc_utils.py:88-111`` for CUDA sample sets x, y [m, d]."""
        if x.shape != y.shape:
            raise ValueError("max_mean_disc: both sample sets must have the same shape (the reference uses m = X.shape[0] for both)")
        out = C.c_double()
        _chk(self.lib.mfm_max_mean_disc(self.h, _ptr(x, F32), _ptr(y, F32), x.shape[0], C.byref(out)))
        return out.value

    def beta_update(self, prev_beta, logliks, alpha):
        out = C.c_double()
        _chk(self.lib.mfm_beta_update(self.h, float(prev_beta), _ptr(logliks, F64), logliks.numel(), float(alpha), C.byref(out)))
        return out.value
