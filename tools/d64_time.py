"""Development aid: the flow step at the reference's phi-four default shape (d = 64, 1024 chains, --hutch) on the shape-specialised
kernel (zero-padded to its 128-wide tile) and on the generic tile (MFM_GENERIC_ODE=1), same network, same keys.  The network is a random one
with a STRONG field (every variant walks the same ~280 attempted steps per chain: a kernel-speed comparison at equal work, not a statement about
a trained flow; the float64 oracle runs into its step limit on it).  D64_EXACT=1: exact trace; D64_B: chain count; MFM_FLOW_LIVE: chains per workgroup."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def one():
    import torch
    from mfm_amd import _lib
    from oracle import prng
    from tests import gpu_util as gu
    B, d = int(os.environ.get("D64_B", 1024)), 64
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=128, F=128, hutch=not os.environ.get("D64_EXACT"))
    params = gu.rand_params(model, seed=1, out_scale=0.05)
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    pos0 = torch.from_numpy(dist.init_params.astype(np.float32)).cuda()
    logp0 = torch.empty(B, dtype=torch.float64, device="cuda"); grad0 = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos0, 1.0, logp0, grad0)
    acc = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ts = []
    for j in range(-1, 6):
        pos, logp, grad = pos0.clone(), logp0.clone(), grad0.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(100 + max(j, 0)), 1.0, pos, logp, grad, acc, None, None, ns); e1.record()
        torch.cuda.synchronize()
        if j >= 0: ts.append(e0.elapsed_time(e1))
    print(f"{'generic tile' if os.environ.get('MFM_GENERIC_ODE') else 'shape-specialised (padded)':28s} flow step {np.mean(ts):7.3f} ms  attempts {ns.float().mean().item():.1f} (max {ns.max().item()})")

if __name__ == "__main__":
    if len(sys.argv) > 1: one()
    else:
        for env in ({}, {"MFM_GENERIC_ODE": "1"}):
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env), check=False)
