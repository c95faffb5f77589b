#!/usr/bin/env python3
"""bench.py -- MFM train-steps/s x chains on phi-four (d=256, 4096 chains per GPU, K=100, --hutch).

One "step" = one iteration of the reference's hot loop (exe_flow_matching.py:432-449) at beta = 1, excluding host
logging: the MCMC half (a fused MALA step, or on every 101st iteration a flow-MH step = two Dopri5 CNF solves with
Hutchinson log-det) + the learning half (flow-matching loss forward/backward, weight gradients, [RCCL all-reduce],
AdamW).  Inputs are synthetic and resident in HBM: chains start from the target's own initialiser U(-1,1)^d, the
network from the flax-style initialiser; the warm-up (default 201 iterations: one full cycle and the MALA + training half of the
next) trains it so the timed flow steps integrate a non-trivial field.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

K should be a multiple of the 101-iteration cycle (default 1010 = ten cycles: 10 flow steps + 1000 MALA iterations, the mix the
metric is defined on); other values are timed as asked, and "flow_steps_timed" in the output says what the window held.

`python bench.py --gpus N` launched bare starts the N ranks itself (a child `torch.distributed.run`, before this process touches
the GPU) and relays rank 0's line.  `--workload` selects another BASELINE config (default: the metric's own, phi-four).

Prints ONE JSON line (rank 0).  value = (chains over all GPUs) * K / (max-over-ranks wall time of the K steps);
config.cycle_weighted_value = the same run's per-iteration and per-flow-step times composed into one full (K + 1)-iteration cycle
(what `value` converges to when --steps is a multiple of the cycle).
"roofline": the kernel with the largest share of GPU time in the timed region (HIP events recorded by the library on
its stream, mfm_profile); "cpu_baseline": libmfm_ref, the float64 C / OpenMP oracle (a port of the reference semantics, NOT JAX/XLA;
the numpy oracle beside it on the headline workload, alone on the others)
timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 (matrix), dense
PEAK_HBM_GBS = 8000.0


WORKLOADS = {   # name: (example, dim, hidden width, MALA step size, chains per GPU, K)
    "phi-four": ("phi-four", 256, 128, 1e-4, 4096, 100),       # BASELINE configs[2] (the headline; configs[3] = 8 x this)
    "pines": ("pines", 1024, 1024, 1e-2, 1024, 100),            # BASELINE configs[4]: 8192 chains over 8 GPUs
    "gaussian-mixture": ("gaussian-mixture", 2, 128, 0.2, 4096, 100),   # BASELINE configs[1]: 16 modes, exact trace, eval_step on 409,600 samples
    "4-mode": ("4-mode", 2, 128, 0.2, 512, 10),                 # BASELINE configs[0]: the reference's own CPU-runnable case (51,200 eval samples)
}
D2 = ("gaussian-mixture", "4-mode")      # the d = 2 mixtures: no --hutch (exact trace), eval_step every iteration (exe_flow_matching.py:444-446)


def make_args(n_total, learning_iter, workload="phi-four"):
    from types import SimpleNamespace
    ex, dim, h, eps, _, K = WORKLOADS[workload]
    return SimpleNamespace(
        example=ex, dim=dim, num_chain=n_total, seed=1, sigma=1e-4, fourier_dim=128, fourier_std=1.0, hutchs=workload not in D2,
        ref_dist="stdgauss", cond_flow=True, ot_cond_flow=False, num_importance_samples=0, mcmc_per_flow_steps=float(K),
        learning_iter=learning_iter, eval_iter=100 if workload in D2 else 1, alpha=0.95, anneal_iter=200, num_anneal_temp=200, non_linearity="relu",
        hidden_x=[h, h], hidden_t=[h, h], hidden_xt=[h, h], step_size=eps, learning_rate=1e-3,
        weight_decay=1e-4, adam_beta1=0.9, adam_beta2=0.999, adam_epsilon=1e-8, gradient_clip=1.0, warmup_steps=0,
        rtol=1e-5, atol=1e-5, mxstep=1000.0)


def flops_per_chain(d=256, h=128, F=128, lgcp=False):
    P_w = 2 * F * h + 5 * h * h + 3 * d * h                 # all kernels
    P_x = 2 * d * h + 3 * h * h                             # weights on the x-tangent path
    dgrad = 2 * d * h + 5 * h * h                           # out, gate (d*h each), j1 (2h*h), j2, x2, t2 (h*h each)
    kinv = 2 * d * d if lgcp else 0                         # grad log pi of the LGCP target: one K^-1 contraction per evaluation
    return dict(fwd=2 * P_w + kinv, fm_fwd_bwd=2 * P_w + 2 * dgrad + kinv, wgrad=2 * P_w, field_eval=2 * P_w + 2 * P_x + kinv)


def fixed_step_report(dist, args, fourier, params_flat, pos, logp, grad, field_eval_flops, key):
    """Flow-MH step time in the fixed-step mode (RK4 x 64, Euler x 256: 256 field evaluations per solve each) on a context of its own
    that shares nothing with the timed one but copies of its parameters and chain states."""
    import copy
    import torch
    from mfm_amd._lib import FLOW_RWMH
    from mfm_amd.engine import Engine
    res = {}
    for method, steps in (("rk4", 64), ("euler", 256)):
        a2 = copy.copy(args); a2.ode_method, a2.ode_steps = method, steps
        d2_ = copy.copy(dist)
        eng2 = Engine(d2_, a2, fourier)
        eng2.ctx.set_params(params_flat)
        B = pos.shape[0]
        acc = torch.empty(B, device=pos.device); ns = torch.empty(B, dtype=torch.int32, device=pos.device)
        ms = []
        for rep in range(4):
            p, l, g = pos.clone(), logp.clone(), grad.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            eng2.ctx.sync(); torch.cuda.synchronize()
            eng2.ctx.profile(True, classes=["flow_step"])
            eng2.ctx.flow_step(FLOW_RWMH, key, 1.0, p, l, g, acc, None, None, ns)
            pr = eng2.ctx.profile_read(); eng2.ctx.profile(False)
            if rep:
                ms.append(pr["flow_step"]["ms"])
        t = sum(ms) / len(ms)
        evals = 2 * steps * (4 if method == "rk4" else 1)
        # executed matrix work per chain: the x branch of every evaluation (320 MFMAs per wave and 16 chains: 327,680 flop per chain) and the
        # time branch once per slot of a five-slot batch (196,608 flop per chain and stage time: one batch per two RK4 / five Euler steps)
        batches = 2 * -(-steps // (2 if method == "rk4" else 5))
        executed = evals * 327680.0 + batches * 5 * 196608.0
        res[f"{method}_x{steps}"] = {"flow_step_ms": round(t, 4), "field_evaluations_per_chain": evals,
                                    "tflops_on_algorithmic_flops": round(B * evals * field_eval_flops / (t * 1e-3) / 1e12, 1),
                                    "executed_matrix_flops_frac_of_f32_mfma_peak": round(B * executed / (t * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
        eng2.close()
    res["note"] = ("build-side mode (BASELINE north star's RK4/Euler integrator), not the reference's adaptive Dopri5 (tests/test_gpu_fixed.py states its distance "
                   "to it).  tflops_on_algorithmic_flops uses the roofline's convention, (2 P_w + 2 P_x) per field evaluation, and may exceed the matrix peak: "
                   "the time branch is evaluated once per distinct stage time of a batch, the tangent skips the output layer")
    return res


def cpu_baseline(params_flat, fourier, steps_mala, chains, seed=1, c_threads=0):
    """Oracle on a bounded sample: `chains` chains, `steps_mala` MALA+train iterations and ONE flow-MH step + train step, with the
    network the GPU run has after its warm-up.  Composed into one 101-iteration cycle: 100 * t(MALA+train) + t(flow+train).
    c_threads = 0: the float64 numpy restatement (multi-threaded BLAS); > 0: libmfm_ref, its C / OpenMP restatement (oracle/cref:
    the target, the MALA arithmetic, the network and the adaptive solves in C on that many threads; keys, draws and the AdamW step by
    the numpy modules).  Same keys, same chains: the two legs do the same work."""
    import numpy as np
    from oracle import flow, fm, loop, mala, optim, prng, targets
    from oracle.vfield import VectorFieldNet, unflat_params
    args = loop.default_args(example="phi-four", dim=256, num_chain=chains, hutchs=True, step_size=1e-4, seed=seed,
                             mcmc_per_flow_steps=100.0, learning_iter=10000)
    dist = targets.PhiFour(256)
    dist.initialize_model(prng.PRNGKey(seed), chains)
    model = VectorFieldNet(fourier, dist, args.hidden_x, args.hidden_t, args.hidden_xt, "relu", 1.0)
    params = unflat_params(model, params_flat)
    state = optim.TrainState(params, optim.learning_rate_fn(10000, 0, 1e-3))
    vg = targets.Tempered(dist, 1.0).value_and_grad
    st = mala.init(dist.init_params, vg)
    key = prng.PRNGKey(seed + 1)

    cr = None
    if c_threads:
        from oracle import cref
        cr = cref.CRef(model, params)
        cr.set_threads(c_threads)

    def train(st, k):
        if cr is not None:
            cr.set_params(state.params)
            loss, grads = cr.fm_loss_grad(*fm.cond_flow_batch(k, st.position, args.sigma))
        else:
            loss, grads = fm.loss_and_grad(model, state.params, k, st.position, args.sigma)
        state.apply_gradients(grads)

    t0 = time.perf_counter()
    for i in range(steps_mala):
        key, k1, k2 = prng.split(key, 3)
        if cr is not None:
            st, _ = cr.mala_kernel(prng.split(k1, chains), st, args.step_size)
        else:
            st, _, _ = mala.kernel(prng.split(k1, chains), st, vg, args.step_size)
        train(st, k2)
    t_mala = (time.perf_counter() - t0) / steps_mala
    key, k1, k2 = prng.split(key, 3)
    t0 = time.perf_counter()
    stats = {}
    if cr is not None:
        cr.set_params(state.params)
        st, _ = cr.rwmh_step(prng.split(k1, chains), st, args, stats=stats)
    else:
        st, _ = flow.rwmh_step(prng.split(k1, chains), st, vg, model, state.params, args, stats)
    train(st, k2)
    t_flow = time.perf_counter() - t0
    cycle = 100 * t_mala + t_flow
    return dict(value=chains * 101 / cycle, t_mala_train_s=t_mala, t_flow_train_s=t_flow,
                n_att=float(stats["n_att_inv"].mean() + stats["n_att_fwd"].mean()))


def cpu_baseline_d2(workload, params_flat, fourier, chains, n_eval, K, seed=1):
    """Oracle on a bounded sample of a d = 2 mixture iteration (MALA or exact-trace flow step + train step + eval_step on the
    n_eval exact samples): the MALA + train part on all `chains`, eval_step on n_eval / 25 samples (x 25), the flow step on 128
    chains (x chains / 128), composed into one (K + 1)-iteration cycle."""
    import numpy as np
    from oracle import flow, fm, loop, mala, optim, prng, targets
    from oracle.vfield import VectorFieldNet, unflat_params
    if workload == "gaussian-mixture":
        g = np.load(os.path.join(ROOT, "tests", "golden", "gmm16_params.npz"))
        dist = targets.GaussianMixture(g["modes"], g["covs"], g["weights"])
    else:
        dist = targets.GaussianMixture(8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4)
    args = loop.default_args(example=workload, dim=2, num_chain=chains, hutchs=False, step_size=0.2, seed=seed, mcmc_per_flow_steps=float(K), learning_iter=10000)
    dist.initialize_model(prng.PRNGKey(seed), chains)
    model = VectorFieldNet(fourier, dist, args.hidden_x, args.hidden_t, args.hidden_xt, "relu", None)
    params = unflat_params(model, params_flat)
    state = optim.TrainState(params, optim.learning_rate_fn(10000, 0, 1e-3))
    vg = targets.Tempered(dist, 1.0).value_and_grad
    st = mala.init(dist.init_params, vg)
    key = prng.PRNGKey(seed + 1)
    n_ev = max(1024, n_eval // 25)
    real = dist.sample_model_rows(prng.split(prng.PRNGKey(seed + 2), n_ev))
    t0 = time.perf_counter()
    for _ in range(2):
        key, k1, k2 = prng.split(key, 3)
        st, _, _ = mala.kernel(prng.split(k1, chains), st, vg, args.step_size)
        loss, grads = fm.loss_and_grad(model, state.params, k2, st.position, args.sigma)
        state.apply_gradients(grads)
    t_mala = (time.perf_counter() - t0) / 2
    t0 = time.perf_counter()
    fm.loss_and_grad(model, state.params, key, real, args.sigma, need_grad=False)
    t_eval = (time.perf_counter() - t0) * n_eval / n_ev
    nf = min(128, chains)
    sub = mala.MALAState(st.position[:nf], st.logdensity[:nf], st.logdensity_grad[:nf])
    stats = {}
    t0 = time.perf_counter()
    flow.rwmh_step(prng.split(key, nf), sub, vg, model, state.params, args, stats)
    t_flow = (time.perf_counter() - t0) * chains / nf
    cycle = K * t_mala + (t_flow + t_mala) + (K + 1) * t_eval
    return dict(value=chains * (K + 1) / cycle,
                sample=f"{chains} chains: 2 MALA+train iterations ({t_mala:.3f}s each), eval_step on {n_ev} of {n_eval} samples (scaled: {t_eval:.3f}s), one exact-trace "
                       f"flow-MH step on {nf} chains (scaled: {t_flow:.3f}s, mean {float(stats['n_att_inv'].mean() + stats['n_att_fwd'].mean()):.1f} Dopri5 attempts); "
                       f"composed into a {K + 1}-iteration cycle; float64 numpy oracle")


def cpu_baseline_pines(params_flat, fourier, dim, hidden, chains, K, seed=1):
    """Oracle on a bounded sample of the pines iteration (LGCP 32 x 32, hidden 1024): ONE MALA + train iteration on 256 chains
    (scaled to `chains`) and one Hutchinson flow-MH step + train step on 32 chains (scaled), with the network the GPU run has
    after its warm-up; composed into one (K + 1)-iteration cycle."""
    import numpy as np
    from oracle import flow, fm, loop, mala, optim, prng, targets
    from oracle.vfield import VectorFieldNet, unflat_params
    n = int(round(dim ** 0.5))
    counts = np.load(os.path.join(ROOT, "mfm_amd", "data", "pines_counts.npz"))[f"counts_{n}"]
    nm, nf = min(256, chains), min(32, chains)
    args = loop.default_args(example="pines", dim=dim, num_chain=nm, hutchs=True, step_size=0.01, seed=seed, mcmc_per_flow_steps=float(K),
                             learning_iter=10000, hidden_x=[hidden] * 2, hidden_t=[hidden] * 2, hidden_xt=[hidden] * 2)
    dist = targets.LogGaussianCoxPines(dim, counts)
    dist.initialize_model(prng.PRNGKey(seed), nm)
    model = VectorFieldNet(fourier, dist, args.hidden_x, args.hidden_t, args.hidden_xt, "relu", 1.0)
    params = unflat_params(model, params_flat)
    state = optim.TrainState(params, optim.learning_rate_fn(10000, 0, 1e-3))
    vg = targets.Tempered(dist, 1.0).value_and_grad
    st = mala.init(dist.init_params, vg)
    key = prng.PRNGKey(seed + 1)
    key, k1, k2 = prng.split(key, 3)
    t0 = time.perf_counter()
    st, _, _ = mala.kernel(prng.split(k1, nm), st, vg, args.step_size)
    loss, grads = fm.loss_and_grad(model, state.params, k2, st.position, args.sigma)
    state.apply_gradients(grads)
    t_mala = (time.perf_counter() - t0) * chains / nm
    sub = mala.MALAState(st.position[:nf], st.logdensity[:nf], st.logdensity_grad[:nf])
    stats = {}
    t0 = time.perf_counter()
    flow.rwmh_step(prng.split(key, nf), sub, vg, model, state.params, args, stats)
    t_flow = (time.perf_counter() - t0) * chains / nf
    cycle = K * t_mala + (t_flow + t_mala)
    return dict(value=chains * (K + 1) / cycle,
                sample=f"one MALA+train iteration on {nm} of {chains} chains (scaled: {t_mala:.2f}s), one Hutchinson flow-MH step on {nf} chains "
                       f"(scaled: {t_flow:.1f}s, mean {float(stats['n_att_inv'].mean() + stats['n_att_fwd'].mean()):.1f} Dopri5 attempts), same network as the GPU "
                       f"after warm-up; composed into a {K + 1}-iteration cycle; float64 numpy oracle")


def usable_cores():
    """Host threads this process may actually use: CPU affinity capped by the cgroup CPU quota (a GPU box hands a job a share
    of its cores), not the machine's core count."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def pmc_traffic(kernel_class, workload="phi-four"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (separate --pmc passes of this
    same command, tools/prof_round.sh): 2 * FETCH_SIZE + WRITE_SIZE, both reported in KB; the factor 2 is the gfx950
    correction for wide coalesced reads (MI355X_MICROARCH.md, HBM section).  None when no summary is present."""
    files = {"phi-four": ("profiles/r05_pmc_summary.json", "profiles/r04_pmc_summary.json", "profiles/r03_pmc_summary.json", "profiles/r02_pmc_summary.json", "profiles/r01_pmc_summary.json"),    # newest committed summary first
             "gaussian-mixture": ("profiles/r05_gmm_pmc_summary.json", "profiles/r04_gmm_pmc_summary.json", "profiles/r03_gmm_pmc_summary.json", "profiles/r02_gmm_pmc_summary.json"),
             "4-mode": ("profiles/r05_4mode_pmc_summary.json", "profiles/r04_4mode_pmc_summary.json", "profiles/r03_4mode_pmc_summary.json", "profiles/r02_4mode_pmc_summary.json"),
             "pines": ("profiles/r03_pines_pmc_summary.json", "profiles/r02_pines_pmc_summary.json")}.get(workload, ())
    # kernel class -> what its kernel is called in the summaries (the d = 2 flow step runs d2::flow_kernel since round 3)
    patterns = {"fm_eval": ("fm_eval_kernel", "fm_eval64"), "flow_step": ("flow_step", "d2::flow_kernel")}.get(kernel_class, (kernel_class,))
    if workload == "pines":
        # its roofline entry is the whole training step (many launches): the summary's `_fm_train_step` entry sums FETCH / WRITE
        # over the dispatches from fm_prologue_kernel to adamw_vec_kernel of every training step of the PMC passes (tools/prof_workload.sh)
        for rel in ("profiles/r05_pines_pmc_summary.json", "profiles/r04_pines_pmc_summary.json"):
            try:
                c = json.load(open(os.path.join(ROOT, rel)))["_fm_train_step"]
                return int((2.0 * c["FETCH_SIZE"]["mean_per_step"] + c["WRITE_SIZE"]["mean_per_step"]) * 1024), rel
            except Exception:
                continue
        return None, None
    for rel in files:
        try:
            d = json.load(open(os.path.join(ROOT, rel)))
        except Exception:
            continue
        for name, c in d.items():
            if any(pt in name for pt in patterns) and isinstance(c.get("FETCH_SIZE"), dict) and isinstance(c.get("WRITE_SIZE"), dict):
                return int((2.0 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024), rel
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1010)      # ten full (K+1)-cycles: a flow step lasts as long as its SLOWEST chain (an
                                                             # extreme statistic of 4096 adaptive solves: 510..620 attempted steps from
                                                             # one flow step to the next), so two cycles are a noisy sample of it
    ap.add_argument("--warmup", type=int, default=201)      # one full cycle (incl. its flow step: first launch, scratch set-up) + the
                                                             # MALA iterations of the next: the timed region then holds two complete
                                                             # cycles, each STARTING with its flow step (whose tail produces the draws
                                                             # of the 100 iterations after it, noise.hip: produced and used inside)
    ap.add_argument("--chains-per-gpu", type=int, default=0)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="phi-four",
                    help="phi-four: BASELINE configs[2] (the metric's configuration, default); pines: configs[4] per-GPU shape; "
                         "gaussian-mixture / 4-mode: the d = 2 mixtures of configs[1] / configs[0] (exact trace + eval_step every iteration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fixed-step", action="store_true", help="skip the fixed-step (RK4 / Euler) flow-step report that follows the timed region")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if launched and world != a.gpus:
        # already under a launcher with another world size (e.g. --nproc-per-node 4 with --gpus 8): every rank spawning its own
        # nested N-rank job would oversubscribe the GPUs -- a clean error instead
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if a.gpus > 1 and not launched:
        # launched bare (`python bench.py --gpus N`): start one rank per GPU as a CHILD torch.distributed.run job -- before this
        # process has touched the GPU (nothing above imports torch), never by re-exec -- relay its output and exit with its code
        import subprocess
        # --standalone lets the launcher's own rendezvous pick (and hold) a free port: probing one here and closing the socket
        # before the child binds it would be a race
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               f"--nproc-per-node={a.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.run(cmd, env=env).returncode)
    import numpy as np
    import torch
    # MFM_BENCH_SHARE_GPU=1 + MFM_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path on a ONE-GPU box (all ranks on device 0,
    # gloo instead of RCCL, which refuses two ranks on one device); never used by the driver's runs
    torch.cuda.set_device(0 if os.environ.get("MFM_BENCH_SHARE_GPU") else local_rank)
    td = None
    backend = None
    if world > 1 or os.environ.get("MFM_COLLECTIVES_AT_WORLD1"):                      # the latter: rehearsal of the RCCL call
        import torch.distributed as td                                                # pattern on a one-rank communicator
        backend = os.environ.get("MFM_BENCH_BACKEND", "nccl")                         # "nccl" = RCCL on ROCm
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        td.init_process_group(backend, **({"device_id": torch.device("cuda", torch.cuda.current_device())} if backend == "nccl" else {}))

    from mfm_amd import exe_flow_matching as E, random as jr
    from mfm_amd._lib import FLOW_RWMH
    from mfm_amd.distributions import GaussianMixture, LogGaussianCoxPines, PhiFour
    from mfm_amd.engine import Engine

    wl_example, wl_dim, wl_h, _, wl_chains, wl_K = WORKLOADS[a.workload]
    if not a.chains_per_gpu:
        a.chains_per_gpu = wl_chains
    if a.workload == "pines" and a.steps == 1010 and a.warmup == 201:     # two cycles after two cycles of training: the timed flow
        a.steps, a.warmup = 202, 202                                       # steps integrate a field that has trained for 200+ iterations
    if a.workload == "4-mode" and a.steps == 1010 and a.warmup == 201:    # K = 10: twenty 11-iteration cycles after two
        a.steps, a.warmup = 220, 22
    n_total = a.chains_per_gpu * world            # weak scaling: per-GPU work fixed
    args = make_args(n_total, learning_iter=10000, workload=a.workload)
    d2 = a.workload in D2
    if a.workload == "phi-four":
        dist = PhiFour(wl_dim)
    elif a.workload == "pines":
        dist = LogGaussianCoxPines(wl_dim)
    elif a.workload == "gaussian-mixture":                  # multi_modal.py:39-47 (parameters: tests/golden/gmm16_params.npz)
        from mfm_amd.multi_modal import gmm16_parameters
        dist = GaussianMixture(*gmm16_parameters())
    else:                                                   # multi_modal.py:79-85
        dist = GaussianMixture(8. * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1]]), np.ones((4, 2)), np.ones(4) / 4)
    key_target, key_sample, key_init, key_dist, key_fourier, key_gen = jr.split(jr.PRNGKey(args.seed), 6)
    dist.initialize_model(key_dist, n_total)
    fourier = args.fourier_std * jr.normal(key_fourier, (args.fourier_dim,))
    n_eval = args.eval_iter * n_total if d2 else 0         # eval_step's batch (exe_flow_matching.py:370-374): eval_iter * num_chain
    eng = Engine(dist, args, fourier, max_eval_samples=n_eval // world)
    real_samples = key_loss = eval_out = None
    if d2:
        key_gen_t, key_loss = jr.split(key_target)          # :371
        lo = eng.rank * (n_eval // world)
        rows = dist.sample_rows(jr.split(key_gen_t, n_eval)[lo:lo + n_eval // world])       # :372-373, this rank's slice
        real_samples = torch.as_tensor(np.ascontiguousarray(rows, dtype=np.float32), device=eng.dev)
        eval_out = torch.zeros(1, device=eng.dev, dtype=torch.float64)
    model = E.VectorFieldNet(fourier, dist.grad_logprob, args.hidden_x, args.hidden_t, args.hidden_xt).attach(eng)
    eng.ctx.set_params(E.flatten_params(model.init(key_init)))
    ctx = eng.ctx
    pos = eng.local(dist.init_params)
    logp = torch.empty(eng.n_local, device=eng.dev, dtype=torch.float64)
    grad = torch.empty_like(pos)
    acc = torch.empty(eng.n_local, device=eng.dev, dtype=torch.float32)
    nst = torch.zeros(eng.n_local, device=eng.dev, dtype=torch.int32)
    beta = 1.0                                   # steady state
    ctx.mala_init(pos, beta, logp, grad)

    EXTRA = 40                                               # untimed, fully instrumented iterations after the timed region
    # The metric is a steady-state rate: the timed flow steps must integrate a TRAINED field (at the flax zero-init the flow is
    # the identity and a flow step costs a tenth of its steady-state time), and one iteration in K+1 is a flow step.  So,
    # whatever --warmup / --steps are: (1) at least one full cycle (a flow step + K training iterations) runs untimed --
    # `prep` state-preparation iterations are added in front of a shorter --warmup -- and (2) the loop counter is aligned so
    # that the timed region STARTS with a flow step: it then holds ceil(steps / (K+1)) of them, never fewer than the
    # schedule's share (a short --steps understates the rate, it cannot overstate it).  Defaults: prep = 0, alignment = 0.
    prep = max(0, wl_K + 1 - a.warmup)
    untimed = prep + a.warmup
    count0 = (-(untimed + 1)) % (wl_K + 1)
    total = untimed + a.steps
    keys = np.empty((total + EXTRA + 1 + wl_K, 2, 2), dtype=np.uint32)  # key plumbing of :433, precomputed off the clock
    ks = key_sample
    for i in range(total + EXTRA + 1 + wl_K):
        ks, k_gn, k_step = jr.split(ks, 3)
        keys[i, 0], keys[i, 1] = k_gn, k_step
    natt_sum = torch.zeros(1, device=eng.dev, dtype=torch.float64)
    n_flow = [0]

    def step(i, count):
        flow = count % (wl_K + 1) == 0                                           # :311
        if flow and not os.environ.get("MFM_NO_PREFETCH"):   # draws of this iteration's training batch and of the next K iterations, in the flow step's tail (noise.hip)
            ctx.noise_prefetch(keys[i:i + 1 + wl_K, 0], keys[i:i + 1 + wl_K, 1])
        # generator (:300-314) + loss/grad + [all-reduce] + AdamW (:362-368): mfm_train_iter on one rank
        eng.train_iter(count, wl_K, FLOW_RWMH, keys[i, 0], keys[i, 1], beta, args.step_size, pos, logp, grad, acc=acc, nsteps=nst)
        if d2:                                               # eval_step (:370-374, :444-446): same key, same exact samples, every iteration
            eng.eval_loss(key_loss, real_samples, eval_out, n_total=n_eval, offset=eng.rank * (n_eval // world))
        if flow:
            natt_sum.add_(nst.double().sum()); n_flow[0] += 1

    def fence():
        if td is not None:
            td.barrier()
        torch.cuda.synchronize()

    # Prime every kernel of the loop once on COPIES of the chain state (first launch of a kernel = code-object load, scratch and
    # LDS set-up: ~10 ms for the flow-step kernel), so that the timed region does not depend on whether --warmup happens to
    # contain a flow step.  Nothing of the benchmark state changes: parameters, optimizer and chains are untouched.
    _p, _l, _g = pos.clone(), logp.clone(), grad.clone()
    ctx.mala_step(keys[0, 0], beta, args.step_size, _p, _l, _g, acc)
    # (the untimed iterations hold a full cycle: they launch the flow-step kernel themselves)
    ctx.fm_loss(keys[0, 1], _p, torch.zeros(1, device=eng.dev, dtype=torch.float64))
    natt_sum.add_(nst.double().sum()); natt_sum.zero_()      # torch loads its reduction kernels lazily, too
    fence()
    del _p, _l, _g
    count = count0
    for i in range(untimed):
        count += 1
        step(i, count)
    fence()
    natt_sum.zero_(); n_flow[0] = 0
    if not os.environ.get("MFM_NO_PREFETCH"):
        ctx.noise_prefetch(keys[:wl_K, 0], keys[:wl_K, 1])   # one-time allocation of the draw buffers (2.5 GB) off the clock ...
    ctx.noise_drop()      # ... and nothing produced before the timed region is used inside it: all work of a timed step is timed
    fence()
    # HIP events on the library's stream around every launch of the DOMINANT kernel class only: an event pair per launch of
    # every class costs ~38 us of stream time per iteration (measured: 0.911 -> 0.872 ms/step), which would be charged to
    # `value`.  The other classes are timed in a separate instrumented pass after the timed region.
    dom_cls = {"phi-four": "flow_step", "pines": "fm_fwd_bwd"}.get(a.workload, "fm_eval")
    ctx.profile(True, classes=sorted({dom_cls, "flow_step"}))      # + the flow step (one launch in K + 1): cycle_weighted_value
    t0 = time.perf_counter()
    for i in range(untimed, total):
        count += 1
        step(i, count)
    fence()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    # instrumented pass (not part of `value`): every class, on MALA + training iterations in the state the timed ones run in --
    # after an (uninstrumented) flow step whose tail has produced their draws; without it a timed region that ends with its
    # cycle leaves these iterations drawing in line (mala_step 38 us instead of 10, the training kernel 65 instead of 43)
    ctx.profile(False)
    natt_timed, nflow_timed = natt_sum.clone(), n_flow[0]    # the tallies of the timed region: the flow step below is not part of it
    count += (wl_K + 1) - count % (wl_K + 1)
    step(total, count)
    natt_sum.copy_(natt_timed); n_flow[0] = nflow_timed
    ctx.profile(True)
    for extra in range(1, EXTRA + 1):                        # MALA + training iterations only (a flow count is skipped)
        count += 2 if (count + 1) % (wl_K + 1) == 0 else 1
        step(total + extra, count)
    fence()
    prof_all = ctx.profile_read()
    ctx.profile(False)
    # proof of what the collectives spanned (a SCALE record must show that RCCL saw N ranks, not N one-rank jobs): the process
    # group's own size, and an all-reduce(SUM) of the rank ids, which is world (world - 1) / 2 only if every rank took part
    comm = {"world": world, "backend": backend, "group_size": 1, "rank_id_sum": 0, "rank_id_sum_expected": 0, "ok": world == 1}
    if td is not None:
        ids = torch.tensor([float(rank), 1.0], device=eng.dev, dtype=torch.float64)
        td.all_reduce(ids, op=td.ReduceOp.SUM)
        comm.update(group_size=td.get_world_size(), rank_id_sum=int(ids[0].item()), ranks_counted=int(ids[1].item()),
                    rank_id_sum_expected=world * (world - 1) // 2)
        comm["ok"] = comm["group_size"] == world and comm["rank_id_sum"] == comm["rank_id_sum_expected"] and comm["ranks_counted"] == world
        if getattr(eng, "rccl_in_lib", False):
            comm["in_library_comm_ranks"] = eng.ctx.comm_count()
    if td is not None:
        tmax = torch.tensor([dt], device=eng.dev, dtype=torch.float64)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        dt = tmax.item()
        td.all_reduce(natt_sum, op=td.ReduceOp.SUM)
    value = n_total * a.steps / dt

    if rank == 0:
        fl = flops_per_chain(wl_dim, wl_h, 128, lgcp=a.workload == "pines")
        B = eng.n_local
        natt_mean = natt_sum.item() / max(1, n_flow[0] * n_total)               # attempted Dopri5 steps per chain per flow step (2 solves)
        alg = {  # algorithmic FLOPs per launch
            "fm_fwd_bwd": B * fl["fm_fwd_bwd"], "wgrad": B * fl["wgrad"],
            "flow_step": B * (4 + 6 * natt_mean) * fl["field_eval"],
            "fm_eval": (n_eval // world) * fl["fwd"],        # eval_step: forward only on eval_iter * num_chain exact samples (2 P_w each)
        }
        if d2:                                               # exact trace: d tangent passes per field evaluation (:216-217)
            P_x = 2 * wl_dim * wl_h + 3 * wl_h * wl_h
            alg["flow_step"] = B * (4 + 6 * natt_mean) * (fl["fwd"] + wl_dim * 2 * P_x)
        if a.workload == "pines":
            alg["fm_fwd_bwd"] += B * fl["wgrad"]          # the wide family's class 1 covers forward, data and weight gradients
        dom = max((k for k in prof if k in alg and prof[k]["launches"]), key=lambda k: prof[k]["ms"], default=None)
        src = prof
        if dom is None:      # no launch of the dominant class inside a very short timed region: the instrumented pass stands in
            src = prof_all
            dom = max((k for k in prof_all if k in alg and prof_all[k]["launches"]), key=lambda k: prof_all[k]["ms"], default=None)
        roof = None
        if dom is not None:
            avg_ms = src[dom]["ms"] / src[dom]["launches"]
            ach = alg[dom] / (avg_ms * 1e-3) / 1e12
            traffic, traffic_src = pmc_traffic(dom, a.workload)
            roof = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_ms": round(avg_ms, 5), "algorithmic_flop_per_launch": alg[dom]}
        desc = {
            "phi-four": ("phi-four d=256, 4096 chains per GPU", "phi-four d=256, 4096 chains/GPU, mcmc_per_flow_steps=100, --hutch, beta=1 (BASELINE configs[2])"),
            "pines": ("pines d=1024, 1024 chains per GPU", f"pines LGCP d={wl_dim} (32x32), hidden {wl_h}, {a.chains_per_gpu} chains/GPU, mcmc_per_flow_steps=100, --hutch, "
                      "beta=1 (BASELINE configs[4] per-GPU shape; wide kernel family)"),
            "gaussian-mixture": ("gaussian-mixture d=2, 4096 chains per GPU", f"gaussian-mixture (16 modes, d=2), {a.chains_per_gpu} chains/GPU, mcmc_per_flow_steps=100, exact trace, "
                                 f"eval_step on {n_eval} exact samples every iteration, beta=1 (BASELINE configs[1])"),
            "4-mode": ("4-mode d=2, 512 chains per GPU", f"4-mode mixture (d=2), {a.chains_per_gpu} chains/GPU, mcmc_per_flow_steps=10, exact trace, eval_step on {n_eval} exact "
                       "samples every iteration, beta=1 (BASELINE configs[0], the reference's CPU-runnable case)"),
        }[a.workload]
        # The metric is defined on the (K + 1)-iteration cycle (one flow step + K MALA iterations).  A --steps that is not a
        # multiple of the cycle over-represents the flow step (the window starts with one), so besides `value` -- the K timed
        # steps as asked -- the same run's numbers are also composed into one full cycle: every iteration's non-flow part
        # (MALA kernel on K of them, training step, eval) at its average inside the timed region + one average flow-step launch.
        flow_ms = prof.get("flow_step", {"ms": 0.0, "launches": 0})
        cw = None
        if flow_ms["launches"]:
            t_rest = (dt * 1e3 - flow_ms["ms"]) / a.steps
            cycle_ms = (wl_K + 1) * t_rest + flow_ms["ms"] / flow_ms["launches"]
            cw = {"cycle_weighted_value": round(n_total * (wl_K + 1) / (cycle_ms * 1e-3), 1), "cycle_ms": round(cycle_ms, 4),
                  "iteration_ms_excluding_flow_kernel": round(t_rest, 5), "flow_step_avg_ms": round(flow_ms["ms"] / flow_ms["launches"], 4)}
        par = ("single GPU" + (f" (REHEARSAL: the multi-rank call sequence with its collectives on a one-rank '{backend}' communicator)" if td is not None else "")) if world == 1 else (f"chains sharded x{world}, one gradient all-reduce(SUM) per iteration over torch.distributed backend "
                                               f"'{backend}'" + (" (RCCL over xGMI)" if backend == "nccl" else " (rehearsal backend, NOT RCCL)"))
        if getattr(eng, "rccl_in_lib", False):
            par += ("; gradient all-reduce inside the library on a context-owned RCCL communicator (mfm_comm_init)"
                    + (", in line behind the weight-gradient kernel of one mfm_train_iter per iteration" if getattr(eng, "_fused_n", False) else ", on its communication stream, optimizer step deferred"))
        out = {
            "metric": f"MFM train-steps/s x chains ({desc[0]})", "value": round(value, 1),
            "unit": "chain-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc[1],
                       "chains_total": n_total, "chains_per_gpu": B, "parallelism": par,
                       "state_prep_iterations": prep, "flow_steps_timed": n_flow[0], "dopri_attempts_per_chain_per_flow_step": round(natt_mean, 2),
                       "chain_dim_updates_per_s": round(value * wl_dim, 1), **(cw or {})},
            "roofline": roof,
            "kernels_ms_total": {k: {"ms": round(v["ms"], 3), "launches": v["launches"]} for k, v in prof.items() if v["launches"]},
            "kernels_avg_us_instrumented_pass": {k: round(v["ms"] / v["launches"] * 1e3, 2) for k, v in prof_all.items() if v["launches"]},
            "counters": ctx.counters(),
            "comm": comm,
        }
        if world == 1 and a.workload == "phi-four" and not a.no_fixed_step:
            # NS1, reported BESIDE the benchmarked (adaptive Dopri5) flow step, never instead of it: the same flow-MH step on the same
            # chains and network with both solves on N equal steps (mfm_config.ode_method; mfm_amd/csrc/ode_fixed.hip)
            try:
                out["fixed_step_flow_step"] = fixed_step_report(dist, args, fourier, ctx.get_params(), pos, logp, grad, fl["field_eval"], keys[0, 0])
            except Exception as e:
                out["fixed_step_flow_step"] = {"failed": repr(e)}
        if world == 1 and not a.no_cpu_baseline:
            try:
                from threadpoolctl import threadpool_limits
                cores = usable_cores()
                params_flat = ctx.get_params()
                with threadpool_limits(limits=cores):
                    if a.workload == "pines":
                        cb = cpu_baseline_pines(params_flat, fourier, wl_dim, wl_h, chains=a.chains_per_gpu, K=wl_K)
                        sample = cb.pop("sample")
                    elif d2:
                        cb = cpu_baseline_d2(a.workload, params_flat, fourier, chains=a.chains_per_gpu, n_eval=n_eval, K=wl_K)
                        sample = cb.pop("sample")
                    else:
                        # two legs on the same 512 chains, keys and network: libmfm_ref (C / OpenMP, the reported value) and the numpy restatement
                        cb = cpu_baseline(params_flat, fourier, steps_mala=4, chains=512, c_threads=cores)
                        cn = cpu_baseline(params_flat, fourier, steps_mala=4, chains=512)
                        sample = (f"512 chains: 4 MALA+train iterations and 1 flow-MH+train iteration (mean {cb['n_att']:.1f} "
                                  f"Dopri5 attempts), same network as the GPU after warm-up; composed into a 101-iteration cycle "
                                  f"(t_mala_train={cb['t_mala_train_s']:.3f}s, t_flow_train={cb['t_flow_train_s']:.3f}s); libmfm_ref: float64 C / OpenMP "
                                  f"restatement (oracle/cref), one chain per task; the float64 numpy restatement on the same sample and threads: "
                                  f"{cn['value']:.1f} chain-steps/s (t_mala_train={cn['t_mala_train_s']:.3f}s, t_flow_train={cn['t_flow_train_s']:.3f}s, "
                                  f"mean {cn['n_att']:.1f} attempts)")
                        extra = {"numpy_value": round(cn["value"], 1)}
                out["cpu_baseline"] = {"value": round(cb["value"], 1), "unit": "chain-steps/s", "cores": cores, "kind": "port", "sample": sample}
                if a.workload == "phi-four":
                    out["cpu_baseline"].update(extra)
            except Exception as e:  # the baseline must never hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "chain-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if td is not None:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
