"""Predicted weak-scaling efficiency of the headline configuration (phi-four d = 256, 4096 chains per GPU, K = 100) from the
per-chain distribution of Dormand-Prince attempts, measured on ONE GPU.

With the gradient all-reduce every iteration, all ranks leave a flow-step iteration together: an N-rank flow step lasts as long
as the slowest of N x 4096 chains.  Model: t_flow(N) = a + b * E[max attempts over N x 4096 chains] with (a, b) fitted on the
measured launches of this GPU (one launch per key on the saved benchmark state, tools/flow_ab.py prepare), the expectation
taken over the pooled empirical distribution of attempts (chains x keys; attempts of a chain are nearly independent of its
position, tools/att_corr.py), plus the measured per-iteration cost of the other 100 iterations and the all-reduce exposure
given on the command line.  Usage (GPU box): python tools/scaling_model.py [--keys 48] [--iter_us 90.8] [--iter1_us 82.1] [--allreduce_us 20]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keys", type=int, default=48)
    ap.add_argument("--iter_us", type=float, default=90.8, help="MALA + training iteration as separate launches, the call sequence of N > 1 ranks (bench with MFM_NO_FUSED_MALA=1: iteration_ms_excluding_flow_kernel)")
    ap.add_argument("--iter1_us", type=float, default=82.1, help="the same iteration on ONE rank, MALA step inside the training kernel, reduction + optimizer in one launch (bench: iteration_ms_excluding_flow_kernel)")
    ap.add_argument("--allreduce_us", type=float, default=20.0, help="exposed all-reduce time per iteration at N > 1 (857 KB over xGMI)")
    a = ap.parse_args()
    import torch
    from mfm_amd import _lib
    from oracle import prng
    from tests import gpu_util as gu
    z = np.load(os.path.join(ROOT, "tools", "data", "flow_ab_state.npz"))
    B, d = z["pos"].shape
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
    ctx = gu.make_ctx(dist, args, fourier=z["fourier"]); ctx.set_params(z["params"])
    pos0 = torch.as_tensor(z["pos"]).cuda(); logp0 = torch.empty(B, dtype=torch.float64, device="cuda"); grad0 = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos0, 1.0, logp0, grad0)
    acc = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    att, ms = [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for j in range(a.keys + 1):
        pos, logp, grad = pos0.clone(), logp0.clone(), grad0.clone()
        torch.cuda.synchronize()
        e0.record()
        ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(1000 + j), 1.0, pos, logp, grad, acc, None, None, ns)
        e1.record(); torch.cuda.synchronize()
        if j:                                   # first launch: code-object load
            att.append(ns.cpu().numpy().copy()); ms.append(e0.elapsed_time(e1))
    att = np.stack(att); ms = np.array(ms)
    mx = att.max(1)
    b, a0 = np.polyfit(mx, ms, 1)
    pool = att.reshape(-1)
    rng = np.random.default_rng(0)
    out = dict(keys=a.keys, chains=B, attempts_mean=float(pool.mean()), attempts_p999=float(np.quantile(pool, 0.999)),
               fit_ms=dict(a=float(a0), b_per_attempt=float(b), resid_ms=float(np.std(ms - (a0 + b * mx)))),
               measured_flow_ms=float(ms.mean()), measured_max_attempts=float(mx.mean()), model={})
    K = 100
    t1 = None
    for N in (1, 2, 4, 8):
        n = N * B
        emax = float(np.mean([rng.choice(pool, n).max() for _ in range(400)]))
        t_flow = a0 + b * emax
        t_iter = a.iter1_us * 1e-3 if N == 1 else (a.iter_us + a.allreduce_us) * 1e-3
        cyc = K * t_iter + t_flow + t_iter            # K MALA iterations + the flow step and its iteration's training step
        rate = N * B * (K + 1) / (cyc * 1e-3)
        if N == 1:
            t1 = rate
        out["model"][N] = dict(expected_max_attempts=emax, flow_ms=t_flow, cycle_ms=cyc, chain_steps_per_s=rate, efficiency=rate / (N * t1))
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
