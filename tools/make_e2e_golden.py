"""Freeze END-TO-END runs of the CPU oracle at the reference's real shapes -> tests/golden/e2e_*.npz.

Two cases, each on three seeds (the spread over seeds is what the statistical tolerances of
tests/test_gpu_e2e.py are derived from):

* ``4mode``  BASELINE configs[0]: ``--example 4-mode --num_chain 512 --learning_iter 100 --mcmc_per_flow_steps 10``
  (``multi_modal.py:65-85``; every other flag at its argparse default, ``:147-220``: hidden 128, Fourier 128, exact
  trace, n_ts = 5, eval_iter 100 -> ``eval_step`` on 51,200 exact samples every iteration, final sampling of 51,200
  flow samples + importance resampling, logpdf / KSD-U / KSD-V / MMD: ``exe_flow_matching.py:432-449,453-490``);
* ``phi4``   the reference's own phi-four defaults (``multi_modal.py:50-63``: d = 64, 1024 chains, step 1e-4,
  eval_iter 1; exact trace, K = 10) over three MALA/flow cycles (``--learning_iter 33``);
* ``phi4_256``  BASELINE configs[2] (the headline): phi-four d = 256, 4096 chains, K = 100, ``--hutch``, 103 iterations
  (one full cycle with its flow step at iteration 101, two iterations after it).

The reference itself cannot be run here (no jax) and ships no fixtures: these are outputs of the ORACLE (oracle/),
frozen because they take minutes to hours of CPU time; they are not outputs of the reference.

Usage: python tools/make_e2e_golden.py [4mode|phi4] [--seeds 1 2 3] [--jobs 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    "4mode": dict(example="4-mode", dim=2, num_chain=512, learning_iter=100, mcmc_per_flow_steps=10.0, step_size=0.2,
                  eval_iter=100, hutchs=False),
    "phi4": dict(example="phi-four", dim=64, num_chain=1024, learning_iter=33, mcmc_per_flow_steps=10.0, step_size=1e-4,
                 eval_iter=1, hutchs=False),
    # BASELINE configs[2], the headline: phi-four d = 256, 4096 chains, K = 100, --hutch -- one full MALA / flow cycle + 2 iterations
    "phi4_256": dict(example="phi-four", dim=256, num_chain=4096, learning_iter=103, mcmc_per_flow_steps=100.0, step_size=1e-4,
                     eval_iter=1, hutchs=True),
    # The headline SHAPE with flow proposals that are ACCEPTED: at lr 1e-3 the Hutchinson log-det of the d = 256 field makes every
    # flow-MH proposal of phi4_256 a rejection (log alpha ~ -700 .. -9000 after one cycle, on both sides), so that case never
    # exercises the accepted-state path.  With --learning_rate 1e-4 (a flag of the reference, multi_modal.py:199) the field grows
    # slowly enough: over the first 24 iterations of the annealing (beta 1e-4 .. 1.3e-3) the six flow steps (K = 3) integrate
    # non-trivial fields (8 .. 10 attempted steps per inverse solve, 17 .. 45 per forward solve) and a sizeable share of their
    # proposals is accepted (mean unclipped ratio 0.02 .. 4).  256 chains, --hutch.
    "phi4_256_accept": dict(example="phi-four", dim=256, num_chain=256, learning_iter=24, mcmc_per_flow_steps=3.0, step_size=1e-4,
                            eval_iter=1, hutchs=True, learning_rate=1e-4),
    # BASELINE configs[1]: the 16-mode mixture, 4096 chains, K = 100, exact trace, eval_step on 409,600 exact samples every iteration
    # (the LOOP only: its final sampling is 409,600 exact-trace solves, hours of oracle time)
    "gmm16": dict(example="gaussian-mixture", dim=2, num_chain=4096, learning_iter=103, mcmc_per_flow_steps=100.0, step_size=0.2,
                  eval_iter=100, hutchs=False),
    # BASELINE configs[4], one GPU's share: pines 32 x 32 grid (d = 1024), 1024 chains, hidden 1024, K = 100, --hutch
    "pines": dict(example="pines", dim=1024, num_chain=1024, learning_iter=103, mcmc_per_flow_steps=100.0, step_size=0.01,
                  eval_iter=1, hutchs=True, hidden_x=[1024, 1024], hidden_t=[1024, 1024], hidden_xt=[1024, 1024]),
}
NO_FINAL = ("gmm16",)
# cases that keep EVERY chain's final position of seed 1, the seed the GPU test runs (per-chain comparison of the accepted flow proposals)
FULL_POS = ("phi4_256_accept",)


def make_dist(case):
    import numpy as np
    from oracle import targets
    if case == "4mode":
        modes, covs, w = 8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4
        d = targets.GaussianMixture(modes, covs, w)
        return d, d.sample_model_rows
    if case == "gmm16":
        g = np.load(os.path.join(ROOT, "tests", "golden", "gmm16_params.npz"))
        d = targets.GaussianMixture(g["modes"], g["covs"], g["weights"])
        return d, d.sample_model_rows
    if case == "pines":
        counts = np.load(os.path.join(ROOT, "mfm_amd", "data", "pines_counts.npz"))["counts_32"]
        return targets.LogGaussianCoxPines(1024, counts), None
    return targets.PhiFour(CASES[case]["dim"]), None


def run_one(job):
    case, seed, threads = job
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = str(threads)
    import numpy as np
    from oracle import loop, metrics
    dist, target_gn = make_dist(case)
    args = loop.default_args(seed=seed, **CASES[case])
    t0 = time.time()

    def timer(count, el):
        if count % 10 == 0:
            print(f"[{case} seed {seed}] iteration {count}: {el:.0f} s", flush=True)
    out = loop.run(dist, args, target_gn=target_gn, timer=timer)
    tr = out["trace"]
    res = dict(seed=seed, loss=np.array(tr["loss"]), learning_rate=np.array(tr["learning_rate"]), beta=np.array(tr["beta"]),
               acc_mean=np.array(tr["acc_mean"]), acc_std=np.array(tr["acc_std"]), n_att=np.array(tr["n_att"]), n_moved=np.array(tr["n_moved"]),
               target_loss=np.array(tr["target_loss"]) if tr["target_loss"] else np.zeros(0))
    pos = out["states"].position
    res.update(chain_mean=pos.mean(0), chain_second=(pos[:, :, None] * pos[:, None, :]).mean(0) if pos.shape[1] <= 8 else (pos ** 2).mean(0),
               chain_logdensity_mean=out["states"].logdensity.mean(), chain_pos=pos.astype(np.float32) if pos.shape[1] == 2 or (case in FULL_POS and seed == 1) else pos[:64, :64].astype(np.float32))
    print(f"[{case} seed {seed}] loop done in {time.time() - t0:.0f} s; final sampling", flush=True)
    if case in NO_FINAL:
        res["oracle_seconds"] = time.time() - t0
        return case, seed, res
    st = {}
    x, ex, info = loop.final_sampling(out["model"], out["state"].params, dist, args, out["keys"]["gen"], stats=st)   # :453-459
    res.update(final_natt_mean=st["n_attempted"].mean(), logpdf=info["samples_logdensity"].mean(),                  # :469
               logpdf_exact=dist.logprob(ex).mean(),                                                                # :473
               flow_mean=x.mean(0), flow_second=(x ** 2).mean(0), exact_mean=ex.mean(0), exact_second=(ex ** 2).mean(0),
               logw_max=info["log_weights"].max(), ess=1.0 / ((info["weights"] / info["weights"].sum()) ** 2).sum())
    ksd = metrics.stein_disc(x, dist.grad_logprob)                                                                  # :471
    ksd_ = metrics.stein_disc(ex, dist.grad_logprob)                                                                # :475
    res.update(ksd_u=ksd[0], ksd_v=ksd[1], ksd_u_exact=ksd_[0], ksd_v_exact=ksd_[1])
    if target_gn is not None:                                                                                       # :480-487
        import oracle.prng as prng
        k = out["keys"]
        # loop.setup re-split k["target"] into (k["gen"], key_loss) and drew the exact samples from split(k["gen"], n)
        real = target_gn(prng.split(k["gen"], args.eval_iter * args.num_chain))
        res.update(mmd=metrics.max_mean_disc(real, x), mmd_exact=metrics.max_mean_disc(real, ex))
    res["oracle_seconds"] = time.time() - t0
    print(f"[{case} seed {seed}] done in {time.time() - t0:.0f} s: logpdf {res['logpdf']:.4f} ksd_v {res['ksd_v']:.4e}", flush=True)
    return case, seed, res


def main():
    import numpy as np
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=["4mode", "phi4"])
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--jobs", type=int, default=3)
    a = ap.parse_args()
    jobs = [(c, s, max(1, 8 // a.jobs)) for c in a.cases for s in a.seeds]
    if a.jobs > 1:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(a.jobs) as pool:
            results = pool.map(run_one, jobs, chunksize=1)
    else:
        results = [run_one(j) for j in jobs]
    for c in a.cases:
        flat = {}
        for case, seed, res in results:
            if case == c:
                flat.update({f"s{seed}_{k}": np.asarray(v) for k, v in res.items()})
        flat["seeds"] = np.array(a.seeds)
        path = os.path.join(ROOT, "tests", "golden", f"e2e_{c}.npz")
        np.savez_compressed(path, **flat)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
