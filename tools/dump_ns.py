"""Development aid: per-chain attempted steps of 8 flow steps from the saved A/B state -> gpurun_out/ns_keys.npz (input of
tools/sched_model.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mfm_amd import _lib
from oracle import prng
from tests import gpu_util as gu
p = os.path.join(ROOT, "tools", "data", "flow_ab_state.npz")
if not os.path.exists(p): p = os.path.join(ROOT, "gpurun_out", "flow_ab_state.npz")
z = np.load(p)
B, d = z["pos"].shape
args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
ctx = gu.make_ctx(dist, args, fourier=z["fourier"]); ctx.set_params(z["params"])
pos0 = torch.as_tensor(z["pos"]).cuda(); logp0 = torch.empty(B, dtype=torch.float64, device="cuda"); grad0 = torch.empty(B, d, device="cuda")
ctx.mala_init(pos0, 1.0, logp0, grad0)
acc = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
out = []
for j in range(8):
    pos, logp, grad = pos0.clone(), logp0.clone(), grad0.clone()
    ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(100 + j), 1.0, pos, logp, grad, acc, None, None, ns)
    out.append(ns.cpu().numpy().copy())
np.savez(os.path.join(ROOT, "gpurun_out", "ns_keys.npz"), ns=np.stack(out))
print("saved", np.stack(out).shape, np.stack(out).mean())
