"""How long does the slowest tile of a flow step run with few rows?  Per-tile order statistics of the attempted steps per chain
(development aid; uses the state saved by tools/flow_ab.py prepare)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
for j in range(8):
    pos, logp, grad = pos0.clone(), logp0.clone(), grad0.clone()
    ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(100 + j), 1.0, pos, logp, grad, acc, None, None, ns)
    n = np.sort(ns.cpu().numpy().reshape(-1, 16), axis=1)[:, ::-1]          # per tile, descending
    s = np.argmax(n[:, 0])
    t = n[s]
    print(f"key {j}: slowest tile: n1 {t[0]} n2 {t[1]} n3 {t[2]} n4 {t[3]} n5 {t[4]} n9 {t[8]} | attempts with <=1 row {t[0]-t[1]}, <=2 rows {t[0]-t[2]}, <=3 {t[0]-t[3]}, <=4 {t[0]-t[4]}, <=8 {t[0]-t[8]} "
          f"| all tiles mean: <=2 rows {np.mean(n[:,0]-n[:,2]):.0f}, <=4 {np.mean(n[:,0]-n[:,4]):.0f}, <=8 {np.mean(n[:,0]-n[:,8]):.0f}, n1 {n[:,0].mean():.0f}")
