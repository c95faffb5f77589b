"""Device engine: one ``mfm_ctx`` (libmfm_hip) per GPU/process plus the chain-sharding and RCCL plumbing.

Chains are sharded contiguously over ranks (rank r owns global chains ``[r*B/W, (r+1)*B/W)``); parameters, optimizer
state and target constants are replicated.  PRNG draws are indexed by global chain id inside the kernels, so chain
trajectories do not depend on the world size.  The only data-path collective is ONE all-reduce(SUM) of the
flow-matching gradient (+ the loss scalar) per iteration, issued through ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests of the sharding logic).
"""
import numpy as np

from . import _lib


def _dist():
    try:
        import torch.distributed as td
        if td.is_available() and td.is_initialized():
            return td
    except Exception:
        pass
    return None


def _collective(td):
    """Whether the collectives are issued: more than one rank -- or MFM_COLLECTIVES_AT_WORLD1=1, the rehearsal of the RCCL
    call pattern (async all-reduce on RCCL's stream, waits, all-gathers) on a ONE-rank communicator, which is all a
    one-GPU box can host (RCCL refuses two ranks on one device)."""
    import os
    return td is not None and (td.get_world_size() > 1 or bool(os.environ.get("MFM_COLLECTIVES_AT_WORLD1")))


def shard(n_total, rank, world):
    """Contiguous shard of the chain axis: (rows resident on this GPU, global id of its first chain, chains among those rows).

    ``--num_chain`` takes any integer (multi_modal.py:169).  The kernels work on tiles of 16 chains (one MFMA M-tile), so a
    shard that is not a multiple of 16 is padded with copies of its last chain: those rows are integrated like any other and
    contribute nothing to the loss, the gradient or any statistic (``mfm_config.n_chain_valid``)."""
    if n_total % world:
        raise ValueError(f"num_chain={n_total} is not divisible by world size {world}")
    n_valid = n_total // world
    return -(-n_valid // 16) * 16, rank * n_valid, n_valid


def allreduce_sum_(*tensors):
    """In-place SUM all-reduce of each tensor over the default process group (no-op without one)."""
    td = _dist()
    if _collective(td):
        for t in tensors:
            td.all_reduce(t, op=td.ReduceOp.SUM)
    return tensors


def allgather_cat(t):
    """Concatenate equally sized per-rank shards in rank order."""
    import torch
    td = _dist()
    if _collective(td):
        parts = [torch.empty_like(t) for _ in range(td.get_world_size())]
        td.all_gather(parts, t)
        return torch.cat(parts)
    return t


def global_mean_std(x, n_total):
    """Mean / population std over ALL chains of a per-chain quantity held as per-rank shards."""
    import torch
    s = torch.stack([x.double().sum(), (x.double() ** 2).sum()])
    allreduce_sum_(s)
    mean = s[0] / n_total
    return mean, (s[1] / n_total - mean * mean).clamp_min(0).sqrt()


class DeferredAllReduce:
    """The gradient all-reduce of iteration i kept in flight while iteration i+1 starts.

    ``submit(t)`` issues ``all_reduce(t, SUM, async_op=True)``: RCCL runs it on its own stream, ordered after the work
    already queued on the caller's stream.  ``flush()`` makes the caller's stream wait for it and then runs ``apply``
    (the optimizer step).  The engine flushes before every call that reads or writes the network parameters, so a MALA
    step (which needs neither the parameters nor the gradient buffer) overlaps the collective; a flow step or the next
    loss/gradient evaluation waits for it.  Without a process group ``submit`` applies immediately."""

    def __init__(self, apply, begin_in_lib=None):
        self.apply = apply
        self.begin_in_lib = begin_in_lib     # context-owned communicator: the library starts and awaits the all-reduce
        self.work = None
        self.armed = False

    def submit(self, *tensors):
        td = _dist()
        if not _collective(td):
            self.apply()
            return
        self.flush()
        if self.begin_in_lib is not None:
            for t in tensors:
                self.begin_in_lib(t)         # mfm_grad_allreduce_begin: asynchronous, on the context's communication stream
            self.work = []
        else:
            self.work = [td.all_reduce(t, op=td.ReduceOp.SUM, async_op=True) for t in tensors]
        self.armed = True

    def flush(self):
        if not self.armed:
            return
        self.armed = False                   # first: apply() may re-enter through the parameter hook
        for w in self.work:
            w.wait()
        self.work = None
        self.apply()                         # (in-library form: mfm_adamw_step waits for the all-reduce itself)


class Engine:
    def __init__(self, dist, args, fourier_random=None, max_eval_samples=0):
        import torch
        self.torch = torch
        td = _dist()
        self.world = td.get_world_size() if td else 1
        self.rank = td.get_rank() if td else 0
        self.n_total = int(args.num_chain)
        self.n_local, self.offset, self.n_valid = shard(self.n_total, self.rank, self.world)
        self.dim = int(args.dim)
        for name in ("hidden_x", "hidden_t", "hidden_xt"):       # exe_flow_matching.py:74-85 loop over lists of any length
            if not 1 <= len(getattr(args, name)) <= _lib.MAX_DEPTH:
                raise NotImplementedError(f"--{name}: {len(getattr(args, name))} hidden layers; the kernels take 1 to {_lib.MAX_DEPTH} per branch")
        if getattr(args, "non_linearity", "relu") not in _lib.ACTIVATIONS:
            raise NotImplementedError(f"unknown non_linearity {args.non_linearity!r} (exe_flow_matching.py:39-45)")
        ref_vars = {"stdgauss": 1.0, "widegauss": 5.0}                               # exe_flow_matching.py:48-54, distributions.py:80-97
        if getattr(args, "ref_dist", "stdgauss") not in ref_vars:
            raise NotImplementedError(f"ref_dist={args.ref_dist!r}: of the reference's table (exe_flow_matching.py:48-54) only "
                                      "'stdgauss' and 'widegauss' can be constructed, there as here")
        if getattr(args, "ot_cond_flow", False):
            raise NotImplementedError("ot_cond_flow is dead code in the reference (un-imported ott)")
        self.args = args
        self.ctx = _lib.Context(
            dim=self.dim, fourier_dim=int(args.fourier_dim), hidden_t=args.hidden_t, hidden_x=args.hidden_x,
            hidden_xt=args.hidden_xt, n_chain_local=self.n_local, n_chain_valid=self.n_valid, n_chain_total=self.n_total, chain_offset=self.offset,
            grad_clip=float(args.gradient_clip) if self.dim > 128 else 0.0,          # exe_flow_matching.py:351
            sigma=float(args.sigma), cond_flow=int(bool(args.cond_flow)), hutch=int(bool(args.hutchs)),
            rtol=float(args.rtol), atol=float(args.atol), mxstep=int(args.mxstep),
            n_ts=5 if getattr(args, "example", "") == "4-mode" else 2,               # :347
            learning_rate=float(args.learning_rate), adam_b1=float(args.adam_beta1), adam_b2=float(args.adam_beta2),
            adam_eps=float(args.adam_epsilon), weight_decay=float(args.weight_decay),
            update_clip=float(args.gradient_clip), learning_iter=int(args.learning_iter),
            warmup_steps=int(args.warmup_steps), max_eval_samples=int(max_eval_samples),
            activation=_lib.ACTIVATIONS[getattr(args, "non_linearity", "relu")],
            ref_std=float(np.sqrt(ref_vars[getattr(args, "ref_dist", "stdgauss")])),
            # build-side mode (not in the reference): the CNF solves on N equal RK4 / Euler steps (include/mfm.h: ode_method)
            ode_method=_lib.ODE_METHODS[getattr(args, "ode_method", "dopri5")] if int(getattr(args, "ode_steps", 0) or 0) > 0 else 0,
            ode_steps=int(getattr(args, "ode_steps", 0) or 0))
        kind, blk = dist.target_block()
        self.ctx.set_target(kind, blk)
        self.dist = dist
        dist._engine = self
        if fourier_random is not None:
            self.ctx.set_fourier(np.asarray(fourier_random, dtype=np.float32))
        self.n_params = self.ctx.n_params
        dev = torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        self.grads = torch.zeros(self.n_params, device=dev, dtype=torch.float32)
        self.loss = torch.zeros(1, device=dev, dtype=torch.float64)
        import os
        self._split_calls = bool(os.environ.get("MFM_SPLIT_CALLS")) or _collective(td)   # development: the multi-rank call sequence on one rank
        begin = None
        # More than one rank: the gradient all-reduce runs inside the library, on a communicator the context owns (include/mfm.h:
        # mfm_comm_*), IN LINE on the context's stream right behind the weight-gradient kernel, and every iteration is ONE
        # mfm_train_iter like on one rank (MALA step inside the training kernel).  Measured on a one-rank communicator (tools/dbg/
        # rehearse_n.sh): 84.1 us per MALA + training iteration against 78.8 without collectives -- torch.distributed's all-reduce on RCCL's
        # own stream with the optimizer step deferred behind the next MALA step: 109.3 (two cross-stream event hops cost more than the 9 us
        # of overlap they buy).  MFM_TORCH_ALLREDUCE=1 keeps that form; so does a process group that is not RCCL.  The 128-byte id travels
        # over the process group that is already up (control plane only); a rank-0 failure to produce it sends every rank to the fallback.
        if _collective(td) and td.get_backend() == "nccl" and not os.environ.get("MFM_TORCH_ALLREDUCE"):
            ids = [None]
            if self.rank == 0:
                try:
                    ids = [self.ctx.comm_unique_id()]
                except Exception as err:                      # (librccl not loadable: every rank falls back together)
                    import warnings
                    warnings.warn(f"in-library RCCL communicator unavailable ({err}); using torch.distributed's all-reduce")
            td.broadcast_object_list(ids, src=0)
            if ids[0] is not None:
                self.ctx.comm_init(self.world, self.rank, ids[0])
                begin = self.ctx.grad_allreduce_begin
        self.rccl_in_lib = begin is not None
        self._fused_n = self.rccl_in_lib and not os.environ.get("MFM_NO_FUSED_AT_N")
        self._deferred = DeferredAllReduce(lambda: self.ctx.adamw_step(self.grads), begin)
        self.ctx.before_params = self._deferred.flush

    # ---- helpers --------------------------------------------------------------------------------------------
    def local(self, full):
        """Rows of a [n_total, ...] host array owned by this rank, as a float32 device tensor."""
        return self.torch.as_tensor(self.pad_rows(np.asarray(full)[self.offset:self.offset + self.n_valid]), device=self.dev)

    def pad_rows(self, a):
        """[n_valid, ...] host rows -> [n_local, ...] float32: the padding rows repeat the last chain (finite, never counted)."""
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.shape[0] < self.n_local:
            a = np.concatenate([a, np.repeat(a[-1:], self.n_local - a.shape[0], axis=0)])
        return a

    def empty_state(self):
        t = self.torch
        return (t.empty(self.n_local, self.dim, device=self.dev, dtype=t.float32),
                t.empty(self.n_local, device=self.dev, dtype=t.float64),
                t.empty(self.n_local, self.dim, device=self.dev, dtype=t.float32))

    def loglik(self, pos):
        out = self.torch.empty(pos.shape[0], device=self.dev, dtype=self.torch.float64)
        self.ctx.loglik(pos, out)
        return out

    def all_logliks(self, pos):
        return allgather_cat(self.loglik(pos)[:self.n_valid].contiguous())

    # ---- one training step on the local chains (exe_flow_matching.py:362-368) -----------------------------------
    def train_step(self, key, positions, loss_out=None):
        """Loss + gradient on the local chains, ONE all-reduce(SUM) of the gradient, AdamW.  Returns the LOCAL loss (a
        sum over this rank's chains, :178) in ``loss_out`` (default ``self.loss``); callers that log it sum it over
        ranks off the critical path (``allreduce_sum_`` on a batch of iterations, see ``run()``).  With more than one
        rank the optimizer step is deferred until the parameters are next needed, so the collective overlaps the
        following MALA step (``DeferredAllReduce``)."""
        loss = self.loss if loss_out is None else loss_out
        self.ctx.fm_loss_grad(key, positions, loss, self.grads)  # flushes the previous iteration's deferred step first
        self._deferred.submit(self.grads)
        return loss

    def train_iter(self, count, K, flow_mode, key_gen, key_train, beta, step_size, pos, logp, grad, acc=None, nsteps=None,
                   loss_out=None):
        """One loop iteration (exe_flow_matching.py:432-439): generator + train_step.  A single library call (``mfm_train_iter``) on one
        rank, and on more with the library's own communicator (the gradient all-reduce in line behind the weight-gradient kernel).  The
        split forms (MFM_NO_FUSED_AT_N / MFM_TORCH_ALLREDUCE, or a process group that is not RCCL): the separate calls, so that the MALA
        step — which needs neither the parameters nor the gradient buffer — runs while the previous iteration's all-reduce is in flight."""
        loss = self.loss if loss_out is None else loss_out
        if (self.world == 1 and not self._split_calls) or self._fused_n:
            self.ctx.train_iter(count, K, flow_mode, key_gen, key_train, beta, step_size, pos, logp, grad, loss, self.grads,
                                acc=acc, nsteps=nsteps)
            self.reseed_padding(pos, logp, grad)
            return loss
        if count % (int(K) + 1) == 0:
            self.ctx.flow_step(flow_mode, key_gen, beta, pos, logp, grad, acc, None, None, nsteps)
        else:
            self.ctx.mala_step(key_gen, beta, step_size, pos, logp, grad, acc)
        out = self.train_step(key_train, pos, loss_out)
        self.reseed_padding(pos, logp, grad)
        return out

    def reseed_padding(self, pos, logp, grad):
        """Padding rows of a shard whose chain count is not a multiple of 16 restart every iteration from the LAST CHAIN's state.  They
        are integrated with draws of their own (global ids past the chains'), so left alone they would wander like chains nobody looks at --
        and a padding row that ever went non-finite would poison the weight-gradient GEMM (0 x NaN) and make apply_if_finite reject
        every step.  Restarted from a chain, a padding row is never more than one step away from a state the run itself depends on."""
        if self.n_valid < self.n_local:
            v = self.n_valid
            for t in (pos, logp, grad):
                if t is not None:
                    t[v:] = t[v - 1]

    def flush(self):
        """Apply a deferred optimizer step now (before reading parameters / optimizer state from outside the ctx)."""
        self._deferred.flush()

    def eval_loss(self, key, samples, out=None, n_total=None, offset=0):
        out = self.torch.zeros(1, device=self.dev, dtype=self.torch.float64) if out is None else out
        self.ctx.fm_loss(key, samples, out, n_total=n_total, offset=offset)
        return out

    def mean_std(self, x):
        """Global mean / population std of a per-chain quantity (acceptance rate, exe_flow_matching.py:442-443)."""
        return global_mean_std(x, self.n_total)

    def close(self):
        self._deferred.flush()
        self.ctx.before_params = None
        self.ctx.close()
