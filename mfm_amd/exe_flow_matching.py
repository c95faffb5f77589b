"""MFM training loop and its building blocks -- drop-in for the reference's ``exe_flow_matching.py`` on MI355X.

Keeps the reference's names and call shapes (``VectorFieldNet``, ``create_train_state``, ``create_learning_rate_fn``,
``create_train_data_gn -> (train_data_generator, init_fn, transform_and_logdet)``, ``run(dist, args, target_gn)``;
``exe_flow_matching.py:56-90,93-198,201-318,321-561``) while every array operation of the hot loop (``:432-449``) runs
in hand-written HIP kernels behind the C ABI (``include/mfm.h``).  What changes at the boundary is listed in
INTEGRATION.md: arrays are CUDA tensors batched over chains (the reference batches with ``jax.vmap``), parameters
and optimizer state live on the device (``state.params`` copies them back on demand), and ``jax.random`` keys are
uint32[2] NumPy arrays with the same split conventions.
"""
import logging
import time
from typing import Callable

import numpy as np

from . import random as jr
from . import wandb_shim as wandb
from ._lib import FLOW_IMH, FLOW_RWMH
from .bblackjax.mcmc.mala import MALAInfo, MALAState, build_kernel, init  # noqa: F401  (same import as :28)
from .distributions import IndepGaussian
from .engine import Engine, allgather_cat, allreduce_sum_

logger = logging.getLogger(__name__)

non_lins = {"tanh": "tanh", "elu": "elu", "relu": "relu", "gelu": "gelu", "swish": "swish"}     # :39-45 (names: evaluated in the kernels)
def _dead_ref(name):
    def make(dim):
        raise NotImplementedError(f"ref_dist={name!r} cannot be constructed in the reference either (exe_flow_matching.py:48-54: "
                                  "GaussianMixture(dim) / FlatDistribution.sample_model / PhiFourBase do not fit their call sites)")
    return make


ref_dists = {"stdgauss": lambda dim: IndepGaussian(dim), "widegauss": lambda dim: IndepGaussian(dim, var=5.),      # :48-54
             "bimodal": _dead_ref("bimodal"), "flat": _dead_ref("flat"), "phifour": _dead_ref("phifour")}


# ---- parameters: flax-style pytree <-> canonical flat vector (include/mfm.h) ---------------------------------------
def layer_shapes(dim, fourier_dim, hidden_x, hidden_t, hidden_xt):
    """(in, out) of the Dense layers in flax creation order (``exe_flow_matching.py:74-86``): the time branch, the x branch, the
    gate, the joint branch, the output; the hidden lists have any length >= 1."""
    shapes, prev = [], 2 * fourier_dim
    for h in hidden_t:
        shapes.append((prev, h)); prev = h
    ht, prev = prev, dim
    for h in hidden_x:
        shapes.append((prev, h)); prev = h
    hx = prev
    shapes.append((ht, dim))
    prev = hx + ht
    for h in hidden_xt:
        shapes.append((prev, h)); prev = h
    shapes.append((prev, dim))
    return shapes


def flatten_params(params):
    p = params["params"]
    return np.concatenate([np.concatenate([np.asarray(p[f"Dense_{i}"]["kernel"], np.float32).reshape(-1),
                                           np.asarray(p[f"Dense_{i}"]["bias"], np.float32).reshape(-1)])
                           for i in range(len(p))])


def unflatten_params(flat, shapes):
    out, o = {}, 0
    for i, (fi, fo) in enumerate(shapes):
        W = flat[o:o + fi * fo].reshape(fi, fo); o += fi * fo
        b = flat[o:o + fo]; o += fo
        out[f"Dense_{i}"] = {"kernel": W.copy(), "bias": b.copy()}
    return {"params": out}


class VectorFieldNet:
    """``exe_flow_matching.py:56-90``.  ``grad_logporob`` is ``dist.grad_logprob`` (the reference passes
    ``jax.grad(dist.logprob)``, :351); the gradient is evaluated inside the kernels."""

    def __init__(self, fourier_random, grad_logporob, hidden_x, hidden_t, hidden_xt, act_fn="relu", grad_clip=None):
        if act_fn not in non_lins:
            raise NotImplementedError(f"unknown activation {act_fn!r}")
        self.fourier_random = np.asarray(fourier_random, dtype=np.float64)
        self.dist = getattr(grad_logporob, "__self__", None)
        if self.dist is None:
            raise NotImplementedError("grad_logporob must be dist.grad_logprob of a built target")
        self.hidden_x, self.hidden_t, self.hidden_xt = list(hidden_x), list(hidden_t), list(hidden_xt)
        self.grad_clip = grad_clip
        self.engine = None

    def shapes(self):
        return layer_shapes(self.dist.dim, len(self.fourier_random), self.hidden_x, self.hidden_t, self.hidden_xt)

    def init(self, rng_key, x=None, t=None):
        """flax ``Module.init``: lecun_normal kernels, zero biases, ZERO gate / output kernels (:81,86), float32."""
        shapes = self.shapes()
        keys = jr.split(rng_key, len(shapes))
        out = {}
        gate = len(self.hidden_t) + len(self.hidden_x)
        for i, (fi, fo) in enumerate(shapes):
            if i in (gate, len(shapes) - 1):
                W = np.zeros((fi, fo), np.float32)
            else:
                W = (jr.truncated_normal(keys[i], -2.0, 2.0, (fi, fo)) * np.sqrt(1.0 / fi) / 0.87962566103423978).astype(np.float32)
            out[f"Dense_{i}"] = {"kernel": W, "bias": np.zeros(fo, np.float32)}
        return {"params": out}

    def attach(self, engine):
        self.engine = engine
        return self

    def apply(self, params, x, t):
        """v(x, t) for CUDA tensors x [n, d], t [n] (n multiple of 16)."""
        eng = self.engine
        _maybe_upload(eng, params)
        v = eng.torch.empty_like(x)
        eng.ctx.vf_apply(x, t, v)
        return v


class DeviceParams:
    """Handle for the parameters resident in the device context (what ``state.params`` is during training)."""

    def __init__(self, engine, shapes):
        self.engine, self.shapes = engine, shapes

    def to_host(self):
        return unflatten_params(self.engine.ctx.get_params(), self.shapes)


def _maybe_upload(engine, params):
    if isinstance(params, DeviceParams) or params is None:
        return
    engine.ctx.set_params(flatten_params(params))


def create_learning_rate_fn(num_train_steps, num_warmup_steps: int, learning_rate: float) -> Callable:
    """``exe_flow_matching.py:189-198`` (evaluated on the device by the optimizer kernel; this is the host twin)."""
    def schedule_fn(step):
        step = float(step)
        if num_warmup_steps > 0 and step < num_warmup_steps:
            return learning_rate * step / num_warmup_steps
        ts = num_train_steps - num_warmup_steps
        if ts <= 0:
            return learning_rate
        return learning_rate * (1.0 - min(max(step - num_warmup_steps, 0.0), ts) / ts)
    return schedule_fn


class TrainState:
    """``flax.training.train_state.TrainState`` twin (:101-110,181-186): parameters, AdamW moments and the step
    counters live in the device context; ``apply_gradients`` runs the fused apply_if_finite/AdamW/clip kernel."""

    def __init__(self, engine, apply_fn, shapes):
        self.engine, self.apply_fn = engine, apply_fn
        self.params = DeviceParams(engine, shapes)

    @property
    def step(self):
        return self.engine.ctx.opt_state()["step"]

    def loss_fn(self, rng_key, samples, params=None):
        """flow_matching_loss (:171-178) on CUDA samples [n, d]; returns a 1-element float64 CUDA tensor."""
        _maybe_upload(self.engine, params)
        return self.engine.eval_loss(rng_key, samples)

    def apply_gradients(self, grads):
        self.engine.ctx.adamw_step(grads)
        return self


def create_train_state(vector_field_apply, vector_field_param, learning_rate_fn, args) -> TrainState:
    """``exe_flow_matching.py:93-186``.  The optimizer hyper-parameters were given to the engine with ``args``."""
    model = vector_field_apply.__self__
    eng = model.engine
    eng.ctx.set_params(flatten_params(vector_field_param))
    eng.ctx.reset_optimizer()
    return TrainState(eng, vector_field_apply, model.shapes())


def create_train_data_gn(dist, vector_field_apply, ode_integrator, args):
    """``exe_flow_matching.py:201-318``.  ``ode_integrator`` is accepted for signature parity; the integrator is the
    in-kernel Dopri5 with the reference's rtol / atol / mxstep (given to the engine with ``args``)."""
    model = vector_field_apply.__self__
    eng = model.engine
    t = eng.torch
    n_is = int(args.num_importance_samples)
    mode = FLOW_IMH if n_is < 0 else FLOW_RWMH                                             # :298
    ref_dist = ref_dists[getattr(args, "ref_dist", "stdgauss")](args.dim)                  # :244
    n, d = eng.n_local, eng.dim
    info = dict(acc=t.empty(n, device=eng.dev, dtype=t.float32), isacc=t.empty(n, device=eng.dev, dtype=t.uint8),
                prop=t.empty(n, d, device=eng.dev, dtype=t.float32), w=t.zeros(n, device=eng.dev, dtype=t.float32),
                nsteps=t.zeros(n, device=eng.dev, dtype=t.int32))

    def transform_and_logdet(key, ref_sample, vector_field_param=None):
        """:206-221, batched over samples with ONE shared Hutchinson key (as used at :455)."""
        _maybe_upload(eng, vector_field_param)
        out = t.empty_like(ref_sample)
        ldj = t.empty(ref_sample.shape[0], device=eng.dev, dtype=t.float32)
        eng.ctx.ode_transform(1, ref_sample, out, ldj, key=key)
        return out, ldj

    def _keys_dev(k):
        return t.as_tensor(np.ascontiguousarray(k, dtype=np.uint32).view(np.int32), device=eng.dev)

    def conditional_importance_sampling(rng_key, beta, pos, logp):
        """:280-296.  The solves and target evaluations are the kernels the other flow steps use; the per-chain
        weights, the categorical draw and the state update run in ``mfm_cis_select``."""
        if eng.n_valid != eng.n_local:
            raise NotImplementedError("num_importance_samples > 0 with a chain count that is not a multiple of 16 per GPU")
        keys = jr.split(rng_key, eng.n_total)[eng.offset:eng.offset + n]                   # :303
        kk = jr.split_rows(keys, 4)                                                        # :281
        u0 = t.empty_like(pos); vol0 = t.empty(n, device=eng.dev, dtype=t.float32)
        eng.ctx.ode_transform(-1, pos, u0, vol0, keys=_keys_dev(kk[:, 1]))                  # :282
        ks = jr.split_rows(kk[:, 0], n_is).reshape(n * n_is, 2)                            # :284
        kh = jr.split_rows(kk[:, 2], n_is).reshape(n * n_is, 2)                            # :286
        refs = t.empty(n * n_is, d, device=eng.dev, dtype=t.float32)
        eng.ctx.normal_rows(_keys_dev(ks), refs)                                           # :285
        if ref_dist.std != 1.0 or ref_dist.mean != 0.0:
            refs.mul_(float(ref_dist.std)).add_(float(ref_dist.mean))                      # distributions.py:96-97
        xs = t.empty_like(refs); vols = t.empty(n * n_is, device=eng.dev, dtype=t.float32)
        eng.ctx.ode_transform(1, refs, xs, vols, keys=_keys_dev(kh))                        # :287
        lps = _logprob_any(eng, xs, beta=beta)                                             # :288
        eng.ctx.cis_select(rng_key, n_is, u0, vol0, refs, xs, vols, lps, pos, logp, info["acc"], info["isacc"], info["prop"], info["w"])

    use_hmc = getattr(args, "mcmc_kernel", "mala") == "hmc"

    def train_data_generator(rng_key, states, count, vector_field_param=None, beta=1.0):
        """:300-314.  States are updated IN PLACE (and returned); infos are views of reused device buffers."""
        _maybe_upload(eng, vector_field_param)
        K = args.mcmc_per_flow_steps
        if 0 < K < 1:
            do_flow = count % (int(1 / K) + 1) != 0                                        # :304-309
        else:
            do_flow = count % (int(K) + 1) == 0                                            # :311
        pos, logp, grad = states
        if do_flow and n_is > 0:
            conditional_importance_sampling(rng_key, beta, pos, logp)
        elif do_flow:
            eng.ctx.flow_step(mode, rng_key, beta, pos, logp, grad, info["acc"], info["isacc"], info["prop"], info["nsteps"])
        elif use_hmc:          # build-side mode (--mcmc_kernel hmc): the iteration's MCMC move is an HMC step (mfm_hmc_step) instead of :313
            eng.ctx.hmc_step(rng_key, beta, args.step_size, int(args.hmc_steps), pos, logp, grad, info["acc"], info["isacc"])
        else:
            eng.ctx.mala_step(rng_key, beta, args.step_size, pos, logp, grad, info["acc"], info["isacc"], info["prop"], info["w"])
        return MALAState(pos, logp, grad), MALAInfo(info["acc"], info["isacc"], info["prop"], info["w"])

    def init_fn(init_positions, beta=1.0):
        """:316."""
        pos = init_positions
        logp = t.empty(pos.shape[0], device=eng.dev, dtype=t.float64)
        grad = t.empty_like(pos)
        eng.ctx.mala_init(pos, beta, logp, grad)
        return MALAState(pos, logp, grad)

    train_data_generator.info_buffers = info
    # what run() needs to issue generator + train_step as ONE library call (mfm_train_iter) where that is the same computation
    train_data_generator.flow_mode = mode
    train_data_generator.one_call_ok = n_is <= 0 and args.mcmc_per_flow_steps >= 1 and float(args.mcmc_per_flow_steps).is_integer() and not use_hmc
    return train_data_generator, init_fn, transform_and_logdet


def _finish_metric_rows(eng, metrics, lo, hi):
    """Rows [lo, hi) of the per-iteration metrics hold per-rank partial sums (loss, sum acc, sum acc^2, target loss):
    sum them over ranks in one collective (off the per-iteration critical path) and turn the acceptance sums into the
    mean / population std logged at exe_flow_matching.py:442-443."""
    if hi <= lo:
        return
    rows = metrics[lo:hi]
    allreduce_sum_(rows)
    mean = rows[:, 1] / eng.n_total
    rows[:, 2] = (rows[:, 2] / eng.n_total - mean * mean).clamp_min(0).sqrt()
    rows[:, 1] = mean


def final_sampling(eng, dist, args, key_gen, transform_and_logdet, params=None):
    """``exe_flow_matching.py:453-459``: N = eval_iter * num_chain draws of the reference distribution pushed through the flow
    with ONE shared Hutchinson key, self-normalised importance weights against the target, resampling with replacement.

    The transform is per-sample independent: every rank integrates its contiguous slice of the N draws.  The resampling is
    not -- weights are normalised by the GLOBAL maximum and ``jax.random.choice`` searches the GLOBAL cumulative sum with N
    uniforms -- so flow samples, log-densities and log-weights are all-gathered (N x dim floats: small) and every rank draws
    the same N indices from the same key.  Returns tensors over all N samples, identical on every rank."""
    t = eng.torch
    n_final = args.eval_iter * args.num_chain
    if n_final % eng.world:
        raise ValueError(f"eval_iter * num_chain = {n_final} must be a multiple of the world size")
    ref = ref_dists[args.ref_dist](args.dim)                                                # :388
    u_host = ref.sample_rows(jr.split(key_gen, n_final))                                    # :453 (:389)
    key_hutch, key_choice = jr.split(key_gen)                                               # :454
    per = n_final // eng.world
    lo = eng.rank * per
    u = t.as_tensor(np.ascontiguousarray(u_host[lo:lo + per], dtype=np.float32), device=eng.dev)
    if per % 16:                                                                            # (tiles of 16 samples: pad with copies, drop them again)
        u_pad = t.cat([u, u[-1:].expand(16 - per % 16, -1)]).contiguous()
        flow_pad, vols_pad = transform_and_logdet(key_hutch, u_pad, params)
        flow_local, vols = flow_pad[:per].contiguous(), vols_pad[:per].contiguous()
    else:
        flow_local, vols = transform_and_logdet(key_hutch, u, params)                       # :455
    lp_local = _logprob_any(eng, flow_local)                                                # :456
    ref_lp = (-0.5 * (((u.double() - ref.mean) / ref.std) ** 2).sum(1) - args.dim * np.log(ref.std) - 0.5 * args.dim * np.log(2 * np.pi))   # distributions.py:89-90
    logw_local = lp_local - ref_lp - vols.double()                                          # :457
    flow_samples = allgather_cat(flow_local.contiguous())
    samples_logdensity = allgather_cat(lp_local.contiguous())
    log_weights = allgather_cat(logw_local.contiguous())
    idx = t.empty(n_final, device=eng.dev, dtype=t.int32)
    scratch = t.empty(n_final, device=eng.dev, dtype=t.float64)
    eng.ctx.choice_logw(key_choice, log_weights, n_final, scratch, idx)                     # :458-459
    exact_samples = t.empty_like(flow_samples)
    eng.ctx.gather_rows(flow_samples, idx, exact_samples)
    return dict(flow_samples=flow_samples, exact_samples=exact_samples, samples_logdensity=samples_logdensity,
                log_weights=log_weights, idx=idx, vols=vols, u=u)


def run(dist, args, target_gn=None, log_every=1, return_extras=False):
    """``exe_flow_matching.py:321-561``: the hot loop runs entirely on the device; metrics are fetched every
    ``log_every`` iterations (the reference syncs to the host every iteration for wandb, :442-449)."""
    import torch
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s - %(message)s", datefmt="%m/%d/%Y %H:%M:%S", level=logging.INFO)
    use_real_samples = args.mcmc_per_flow_steps < 0                                        # :328
    if use_real_samples and target_gn is None:
        raise ValueError("mcmc_per_flow_steps < 0 trains on exact samples and needs a target with sample_model (:382-386)")
    learning_iter = args.learning_iter
    iter_per_temp = args.anneal_iter // args.num_anneal_temp                                # :330
    n_iter, n_chain = args.eval_iter, args.num_chain
    key_target, key_sample, key_init, key_dist, key_fourier, key_gen = jr.split(jr.PRNGKey(args.seed), 6)    # :333
    dist.initialize_model(key_dist, n_chain)                                                # :334
    fourier_random = args.fourier_std * jr.normal(key_fourier, (args.fourier_dim,))         # :350
    model = VectorFieldNet(fourier_random, dist.grad_logprob, args.hidden_x, args.hidden_t, args.hidden_xt,
                           non_lins[args.non_linearity], args.gradient_clip if args.dim > 128 else None)     # :351
    n_eval = n_iter * n_chain if target_gn is not None else 0
    eng = Engine(dist, args, fourier_random, max_eval_samples=-(-max(n_eval, n_iter * n_chain, n_chain * max(int(args.num_importance_samples), 0)) // 16) * 16)
    model.attach(eng)
    vector_field_param = model.init(key_init, dist.init_params[0], 0.0)                     # :353
    learning_rate_fn = create_learning_rate_fn(learning_iter, args.warmup_steps, args.learning_rate)          # :355-359
    state = create_train_state(model.apply, vector_field_param, learning_rate_fn, args)     # :360

    real_samples = None
    if target_gn is not None:                                                               # :370-374
        key_gen, key_loss = jr.split(key_target)
        keys_target = jr.split(key_gen, n_eval)
        real_host = dist.sample_rows(keys_target)
        lo = eng.rank * (n_eval // eng.world)
        real_samples = torch.as_tensor(np.ascontiguousarray(real_host[lo:lo + n_eval // eng.world], dtype=np.float32), device=eng.dev)
        eval_loss = torch.zeros(1, device=eng.dev, dtype=torch.float64)

    logger.info(f"===== Starting training seed {args.seed} w/ {learning_iter} iterations =====")
    train_data_generator, init_fn, transform_and_logdet = create_train_data_gn(dist, model.apply, None, args)  # :380-381
    train_start = time.time()                                                               # :421

    pos0 = eng.local(dist.init_params)
    if use_real_samples:                                                                    # :382-386, :429-430
        def train_data_generator(key, states, count, *_):
            """positions = vmap(target_gn)(split(key, n_chain)); no MCMC state, acceptance is NaN."""
            rows = dist.sample_rows(jr.split(key, n_chain))[eng.offset:eng.offset + eng.n_valid]
            states.position.copy_(torch.as_tensor(eng.pad_rows(rows), device=eng.dev))
            return states, MALAInfo(nan_acc, None, None, None)
        nan_acc = torch.full((eng.n_local,), float("nan"), device=eng.dev, dtype=torch.float32)
        init_fn = lambda positions, *_: MALAState(positions, None, None)
        beta = 1.0
    else:
        beta = eng.ctx.beta_update(0.0, eng.all_logliks(pos0), args.alpha)                  # :426
        logger.info(f"Initial beta= {beta}")
    train_states = init_fn(pos0, beta)                                                      # :431
    metrics = torch.zeros(learning_iter, 4, device=eng.dev, dtype=torch.float64)            # loss, acc mean, acc std, target loss
    betas, lrs = [], []
    n_reduced = 0
    import os
    K_int = int(args.mcmc_per_flow_steps) if args.mcmc_per_flow_steps >= 1 else 0
    prefetch = K_int >= 1 and not use_real_samples and args.num_importance_samples <= 0 and not os.environ.get("MFM_NO_PREFETCH")
    # one rank: generator + train_step in one call (more ranks keep them apart: the MALA step overlaps the gradient all-reduce)
    one_call = (not use_real_samples and eng.world == 1 and not eng._split_calls and getattr(train_data_generator, "one_call_ok", False))
    for count in range(1, learning_iter + 1):                                               # :432
        key_sample, key_train_gn, key_train_step = jr.split(key_sample, 3)                  # :433
        if prefetch and count % (K_int + 1) == 0:
            # a flow step comes: let the workgroups of its kernel that finish early produce the random draws of the K
            # MALA + training iterations that follow (noise.hip); their keys are the next K splits of key_sample.  Slot 0 is
            # this iteration's own training batch (its generator key is the flow step's: those MALA draws are never asked for)
            ks, kg, kt = key_sample, [key_train_gn], [key_train_step]
            for _ in range(min(K_int, learning_iter - count)):
                ks, a_, b_ = jr.split(ks, 3)
                kg.append(a_); kt.append(b_)
            if kg:
                prefetch = eng.ctx.noise_prefetch(np.stack(kg), np.stack(kt))               # False: not served for this configuration
        row = metrics[count - 1]                                                            # loss | sum acc | sum acc^2 | target loss
        if one_call:
            # :438-439 as one library call: on a MALA iteration the step runs inside the training kernel's workgroups (fm.hip)
            _maybe_upload(eng, state.params)
            ib = train_data_generator.info_buffers
            eng.train_iter(count, K_int, train_data_generator.flow_mode, key_train_gn, key_train_step, beta, args.step_size,
                           train_states.position, train_states.logdensity, train_states.logdensity_grad, acc=ib["acc"], nsteps=ib["nsteps"],
                           loss_out=row[0:1])
            infos = MALAInfo(ib["acc"], None, None, None)
        else:
            train_states, infos = train_data_generator(key_train_gn, train_states, count, state.params, beta)    # :438
            eng.train_step(key_train_step, train_states.position, loss_out=row[0:1])        # :439 (:362-368)
            eng.reseed_padding(train_states.position, train_states.logdensity, train_states.logdensity_grad)
        lrs.append(learning_rate_fn(count - 1))                                             # :367 (pre-increment step)
        if not use_real_samples and count % iter_per_temp == 0 and beta < 1.0:              # :440-441, :417
            beta = eng.ctx.beta_update(beta, eng.all_logliks(train_states.position), args.alpha)              # :413
            train_states = init_fn(train_states.position, beta)                             # :415
        eng.ctx.acc_stats(infos.acceptance_rate[:eng.n_valid], row[1:3])                    # :442-443, per-rank partial sums (chains only: no padding rows)
        if real_samples is not None:                                                        # :444-446
            eng.eval_loss(key_loss, real_samples, row[3:4], n_total=n_eval, offset=eng.rank * (n_eval // eng.world))
        betas.append(beta)
        if count % log_every == 0 or count == learning_iter:
            _finish_metric_rows(eng, metrics, n_reduced, count); n_reduced = count          # ONE collective per log interval
            row = metrics[count - 1].tolist()                                               # host sync
            wandb.log({"loss": row[0], "learning_rate": lrs[-1], "acceptance avg.": row[1], "acceptance std.": row[2],
                       "target_loss": row[3], "train_time": time.time() - train_start})     # :447-449
    eng.ctx.sync()
    train_time = time.time() - train_start
    logger.info(f"Final beta= {beta}")

    # ---- final flow samples + importance resampling (:453-459), over ALL N samples on every rank --------------------
    fin = final_sampling(eng, dist, args, key_gen, transform_and_logdet, state.params)
    flow_samples, exact_samples, samples_logdensity = fin["flow_samples"], fin["exact_samples"], fin["samples_logdensity"]
    if getattr(args, "check", False) and real_samples is not None:                          # :462-467 (--check: the exact samples' own scores)
        real_chk = allgather_cat(real_samples)
        logger.info(f"Logpdf of real samples= {_logprob_any(eng, real_chk).mean().item()}")
        stein_chk = stein_disc(eng, real_chk)
        logger.info(f"Stein U, V disc of real samples= {stein_chk[0]}, {stein_chk[1]}")
        logger.info(f"Max mean disc of NF+MCMC samples= {eng.ctx.max_mean_disc(real_chk, real_chk)}")       # (the reference's label)
    logpdf = samples_logdensity.mean().item()                                               # :469
    logger.info(f"Logpdf of flow samples= {logpdf}")
    stein = stein_disc(eng, flow_samples)                                                   # :471 (mcmc_utils.py:28-85)
    logger.info(f"Stein U, V disc of flow samples= {stein[0]}, {stein[1]}")
    logpdf_ = _logprob_any(eng, exact_samples).mean().item()                                # :473
    logger.info(f"Logpdf of exact samples= {logpdf_}")
    stein_ = stein_disc(eng, exact_samples)                                                 # :475
    logger.info(f"Stein U, V disc of exact samples= {stein_[0]}, {stein_[1]}")
    if target_gn is not None:                                                               # :480-487 (mcmc_utils.py:88-111)
        real_all = allgather_cat(real_samples)                                              # every rank holds a slice of the exact samples
        mmd = eng.ctx.max_mean_disc(real_all, flow_samples)
        logger.info(f"Max mean disc of flow samples= {mmd}")
        mmd_ = eng.ctx.max_mean_disc(real_all, exact_samples)
        logger.info(f"Max mean disc of exact samples= {mmd_}")
    else:
        mmd = mmd_ = 0.0                                                                    # :490
    res = np.array([logpdf, stein[0], stein[1], mmd, train_time])                           # :561
    res_ = np.array([logpdf_, stein_[0], stein_[1], mmd_, train_time])
    wandb.finish()
    if return_extras:
        return res, res_, dict(metrics=metrics.cpu().numpy(), betas=np.array(betas), lrs=np.array(lrs), states=train_states,
                               engine=eng, state=state, flow_samples=flow_samples, exact_samples=exact_samples, model=model,
                               final=fin, key_gen=key_gen)
    eng.close()
    return res, res_


def stein_disc(eng, x, beta=-0.5):
    """``mcmc_utils.stein_disc(X, dist.logprob)`` (mcmc_utils.py:28-85) for CUDA samples [n, d] -> (U, V)."""
    _, grad = _logprob_any(eng, x, want_grad=True)
    return eng.ctx.stein_disc(x.contiguous(), grad, beta)


def max_mean_disc(eng, x, y):
    """``mcmc_utils.max_mean_disc`` (mcmc_utils.py:88-111) for CUDA sample sets of equal size."""
    return eng.ctx.max_mean_disc(x.contiguous(), y.contiguous())


def _logprob_any(eng, x, want_grad=False, beta=1.0):
    """vmap(dist.logprob) for any sample count: the MALA init kernel at beta = 1 returns loglik + logprior for every
    built target; samples are processed in chunks of the engine's chain count (zero padded)."""
    t = eng.torch
    out = t.empty(x.shape[0], device=eng.dev, dtype=t.float64)
    gout = t.empty(x.shape[0], x.shape[1], device=eng.dev, dtype=t.float32) if want_grad else None
    n = eng.n_local
    lp = t.empty(n, device=eng.dev, dtype=t.float64)
    gr = t.empty(n, x.shape[1], device=eng.dev, dtype=t.float32)
    for s in range(0, x.shape[0], n):
        chunk = x[s:s + n]
        m = chunk.shape[0]
        if m < n:
            pad = t.zeros(n, x.shape[1], device=eng.dev, dtype=x.dtype); pad[:m] = chunk
            chunk = pad
        eng.ctx.mala_init(chunk.contiguous(), float(beta), lp, gr)
        out[s:s + m] = lp[:m]
        if want_grad:
            gout[s:s + m] = gr[:m]
    return (out, gout) if want_grad else out
