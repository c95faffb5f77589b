"""Helpers shared by the -m gpu parity tests: build a device context that mirrors an oracle setup."""
import numpy as np

from oracle import loop, prng, targets
from oracle.vfield import VectorFieldNet


from oracle.vfield import flat_params, unflat_params      # noqa: E402,F401  (re-exported: the tests' historical home of the two)


def rand_params(model, seed=0, scale=1.0, out_scale=0.3):
    """Non-trivial parameters (the flax init has zero output kernels => v == 0, useless for parity)."""
    rng = np.random.default_rng(seed)
    ps = []
    zl = model.zero_layers()
    for i, (fi, fo) in enumerate(model.layer_shapes()):
        s = (out_scale if i in zl else scale) / np.sqrt(fi)
        ps.append({"kernel": (rng.standard_normal((fi, fo)) * s).astype(np.float32),
                   "bias": (rng.standard_normal(fo) * 0.05).astype(np.float32)})
    return ps


def target_block(dist):
    from mfm_amd import _lib
    if dist.kind == "phi4":
        return _lib.PHI4, [dist.a, dist.beta]
    if dist.kind == "gmm":
        K = len(dist.weights)
        return _lib.GMM, np.concatenate([[K], dist.modes.reshape(-1), dist.chol_covs.reshape(-1), dist.weights])
    if dist.kind == "lgcp":
        return _lib.LGCP, np.concatenate([[dist.mu, dist.poisson_a, dist.log_norm], dist.counts, dist.Kinv.reshape(-1)])
    raise NotImplementedError(dist.kind)


def make_ctx(dist, args, n_local=None, n_total=None, offset=0, fourier=None, params=None, max_eval=0, family=None):
    from mfm_amd import _lib
    n_local = args.num_chain if n_local is None else n_local
    ctx = _lib.Context(
        dim=args.dim, fourier_dim=args.fourier_dim, hidden_t=args.hidden_t, hidden_x=args.hidden_x,
        hidden_xt=args.hidden_xt, n_chain_local=n_local, n_chain_total=n_total or n_local, chain_offset=offset,
        grad_clip=(args.gradient_clip if args.dim > 128 else 0.0), sigma=args.sigma, cond_flow=int(args.cond_flow),
        hutch=int(args.hutchs), rtol=args.rtol, atol=args.atol, mxstep=int(args.mxstep), n_ts=args.n_ts,
        learning_rate=args.learning_rate, adam_b1=args.adam_beta1, adam_b2=args.adam_beta2, adam_eps=args.adam_epsilon,
        weight_decay=args.weight_decay, update_clip=args.gradient_clip, learning_iter=args.learning_iter,
        warmup_steps=args.warmup_steps, max_eval_samples=max_eval, activation=_lib.ACTIVATIONS[args.non_linearity],
        ref_std=float(np.sqrt(targets.REF_VARS[getattr(args, "ref_dist", "stdgauss")])),
        ode_method=_lib.ODE_METHODS[getattr(args, "ode_method", "dopri5")] if int(getattr(args, "ode_steps", 0) or 0) > 0 else 0,
        ode_steps=int(getattr(args, "ode_steps", 0) or 0),
        **({} if family is None else {"kernel_family": family}))
    kind, blk = target_block(dist)
    ctx.set_target(kind, blk)
    if fourier is not None:
        ctx.set_fourier(fourier)
    if params is not None:
        ctx.set_params(flat_params(params))
    return ctx


def hidden_lists(hidden):
    """An int: two layers of that width per branch (the reference's default depth); a triple of lists: (hidden_x, hidden_t, hidden_xt)."""
    if isinstance(hidden, int):
        return dict(hidden_x=[hidden, hidden], hidden_t=[hidden, hidden], hidden_xt=[hidden, hidden])
    hx, ht, hxt = hidden
    return dict(hidden_x=list(hx), hidden_t=list(ht), hidden_xt=list(hxt))


def phi4_setup(d=256, B=64, seed=1, hutch=True, hidden=128, F=128, **kw):
    args = loop.default_args(example="phi-four", dim=d, num_chain=B, hutchs=hutch, step_size=1e-4, seed=seed,
                             fourier_dim=F, **hidden_lists(hidden), **kw)
    dist = targets.PhiFour(d)
    k, model, state, lr_fn, _, _ = loop.setup(dist, args)
    return args, dist, k, model, state


def gmm4_setup(B=64, seed=1, hidden=32, F=16, **kw):
    args = loop.default_args(example="4-mode", dim=2, num_chain=B, step_size=0.2, seed=seed, fourier_dim=F, **hidden_lists(hidden), **kw)
    dist = targets.GaussianMixture(8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4)
    k, model, state, lr_fn, _, _ = loop.setup(dist, args)
    return args, dist, k, model, state


def lgcp_setup(n=8, B=32, seed=1, hidden=32, F=16, hutch=True, **kw):
    import os
    d = n * n
    counts = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mfm_amd", "data", "pines_counts.npz"))[f"counts_{n}"]
    args = loop.default_args(example="pines", dim=d, num_chain=B, hutchs=hutch, step_size=0.01, seed=seed, fourier_dim=F,
                             **hidden_lists(hidden), **kw)
    dist = targets.LogGaussianCoxPines(d, counts)
    k, model, state, lr_fn, _, _ = loop.setup(dist, args)
    return args, dist, k, model, state


def train_phi4_like_bench(n_iter=101, chains=4096, d=256, seed=1):
    """The state bench.py's timed region starts from: phi-four d = 256, `chains` chains, K = 100, --hutch, beta = 1, the
    flax-style initial network trained for `n_iter` iterations of the product's loop (101 = one full cycle incl. its flow step).
    Returns the oracle-side objects for the same configuration plus the trained parameters and the chain positions."""
    import sys, os
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from mfm_amd import exe_flow_matching as E, random as jr
    from mfm_amd._lib import FLOW_RWMH
    from mfm_amd.distributions import PhiFour
    from mfm_amd.engine import Engine
    args = bench.make_args(chains, learning_iter=10000)
    args.dim = d
    dist = PhiFour(d)
    key_target, key_sample, key_init, key_dist, key_fourier, key_gen = jr.split(jr.PRNGKey(seed), 6)
    dist.initialize_model(key_dist, chains)
    fourier = args.fourier_std * jr.normal(key_fourier, (args.fourier_dim,))
    eng = Engine(dist, args, fourier)
    model = E.VectorFieldNet(fourier, dist.grad_logprob, args.hidden_x, args.hidden_t, args.hidden_xt).attach(eng)
    eng.ctx.set_params(E.flatten_params(model.init(key_init)))
    pos = eng.local(dist.init_params)
    logp = torch.empty(chains, device=eng.dev, dtype=torch.float64); grad = torch.empty_like(pos)
    acc = torch.empty(chains, device=eng.dev, dtype=torch.float32); nst = torch.zeros(chains, device=eng.dev, dtype=torch.int32)
    eng.ctx.mala_init(pos, 1.0, logp, grad)
    ks = key_sample
    for count in range(1, n_iter + 1):
        ks, k_gn, k_step = jr.split(ks, 3)
        eng.train_iter(count, 100, FLOW_RWMH, k_gn, k_step, 1.0, args.step_size, pos, logp, grad, acc=acc, nsteps=nst)
    eng.ctx.sync()
    out = dict(params_flat=eng.ctx.get_params(), pos=pos.cpu().numpy(), fourier=np.asarray(fourier, dtype=np.float64),
               n_att_last_flow=nst.cpu().numpy().copy(), counters=eng.ctx.counters())
    eng.close()
    o_dist = targets.PhiFour(d)
    oargs = loop.default_args(example="phi-four", dim=d, num_chain=32, hutchs=True, step_size=1e-4, seed=seed, mcmc_per_flow_steps=100.0)
    out["dist"] = o_dist
    out["args32"] = oargs
    out["model"] = VectorFieldNet(out["fourier"], o_dist, oargs.hidden_x, oargs.hidden_t, oargs.hidden_xt, "relu", oargs.gradient_clip)
    return out


def gmm16_setup(B=64, seed=1, hidden=128, F=128, **kw):
    """BASELINE configs[1]: the 16-mode `gaussian-mixture` target (parameters: tests/golden/gmm16_params.npz), d = 2."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gmm16_params.npz"))
    args = loop.default_args(example="gaussian-mixture", dim=2, num_chain=B, step_size=0.2, seed=seed, fourier_dim=F,
                             hidden_x=[hidden, hidden], hidden_t=[hidden, hidden], hidden_xt=[hidden, hidden], **kw)
    dist = targets.GaussianMixture(g["modes"], g["covs"], g["weights"])
    k, model, state, lr_fn, _, _ = loop.setup(dist, args)
    return args, dist, k, model, state


def cached_oracle(tag, compute, *inputs):
    """Frozen outputs of an expensive, DETERMINISTIC oracle computation: ``compute() -> dict of arrays``.

    The four slowest GPU tests spent 400 of the suite's 600 s in the float64 oracle's NATURAL-controller pass at the pines widths
    (hidden 1024, d = 1024 / 1600), whose only product is the step sequence the prescribed-step comparison then replays on both
    sides -- a pure function of positions, keys and parameters.  It is read from ``tests/golden/oracle_<tag>.npz`` when the digest
    of ``inputs`` stored with it matches (a changed setup recomputes: slower, never wrong); ``MFM_WRITE_ORACLE_CACHE=<dir>``
    writes the file (tools/readme: run the GPU tests once with it on the GPU box, copy the files into tests/golden).  The
    REPLAYED oracle pass -- the one the GPU is compared with -- always runs."""
    import hashlib
    import os
    h = hashlib.sha1()
    for a in inputs:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    digest = h.hexdigest()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"oracle_{tag}.npz")
    if os.path.exists(path):
        z = np.load(path)
        if str(z["input_digest"]) == digest:
            return {k: z[k] for k in z.files if k != "input_digest"}
        print(f"cached_oracle({tag}): inputs changed since the file was written -- recomputing")
    out = compute()
    wd = os.environ.get("MFM_WRITE_ORACLE_CACHE")
    if wd:
        os.makedirs(wd, exist_ok=True)
        np.savez_compressed(os.path.join(wd, f"oracle_{tag}.npz"), input_digest=np.array(digest), **out)
    return out
