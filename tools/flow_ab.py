"""Same-trajectory A/B of flow-step kernel variants (development aid).  The duration of a flow step is an extreme statistic (its
slowest chain), which moves by +-5 % from one flow step to the next: two builds are only comparable on the SAME chain states,
parameters and keys.

    python tools/flow_ab.py prepare                 # bench.py's state after its warm-up cycle -> gpurun_out/flow_ab_state.npz
    python tools/flow_ab.py run [libname ...]       # each variant (mfm_amd/lib/libmfm_hip_<name>.so; '' = the product library) in its
                                                    # own process: 8 flow steps from the saved state with 8 fixed keys
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STATE = os.path.join(ROOT, "gpurun_out", "flow_ab_state.npz")        # written on the GPU box; copy it to tools/data/ to reuse it in later calls
STATE_IN = os.path.join(ROOT, "tools", "data", "flow_ab_state.npz")


def prepare():
    from tests import gpu_util as gu
    tp = gu.train_phi4_like_bench()
    os.makedirs(os.path.dirname(STATE), exist_ok=True)
    np.savez(STATE, params=tp["params_flat"], pos=tp["pos"], fourier=tp["fourier"])
    print("saved", STATE, "last flow step attempts", tp["n_att_last_flow"].mean())


def run_one(nkeys=8):
    import torch
    from mfm_amd import _lib
    from oracle import prng
    from tests import gpu_util as gu
    z = np.load(STATE_IN if os.path.exists(STATE_IN) else STATE)
    B, d = z["pos"].shape
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
    ctx = gu.make_ctx(dist, args, fourier=z["fourier"])
    ctx.set_params(z["params"])
    pos0 = torch.as_tensor(z["pos"]).cuda()
    logp0 = torch.empty(B, dtype=torch.float64, device="cuda"); grad0 = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos0, 1.0, logp0, grad0)
    acc = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ts, na, nm = [], [], []
    for j in range(-1, nkeys):
        pos, logp, grad = pos0.clone(), logp0.clone(), grad0.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(100 + max(j, 0)), 1.0, pos, logp, grad, acc, None, None, ns); e1.record()
        torch.cuda.synchronize()
        if j >= 0:
            ts.append(e0.elapsed_time(e1)); na.append(ns.float().mean().item()); nm.append(ns.max().item())
    print(f"{os.environ.get('MFM_LIB', 'product'):60s} mean {np.mean(ts):7.3f} ms  per key {np.round(ts, 2)}  attempts {np.mean(na):.1f}  max {nm}")


if __name__ == "__main__":
    if sys.argv[1] == "prepare":
        prepare()
    elif sys.argv[1] == "one":
        run_one()
    else:
        for name in (sys.argv[2:] or [""]):
            env = dict(os.environ)
            if name:
                env["MFM_LIB"] = os.path.join(ROOT, "mfm_amd", "lib", f"libmfm_hip_{name}.so")
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, check=False)
