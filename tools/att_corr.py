"""Can a chain's number of attempted Dopri5 steps be predicted (for a chain -> tile assignment that balances the tiles)?
Same chain state, different keys (the part of the count the position determines), and a cost model of a tile's run time
from the order statistics of its rows (full / compact / micro attempt costs of DESIGN 4.1).  Development aid."""
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
N = []
for j in range(6):
    pos, logp, grad = pos0.clone(), logp0.clone(), grad0.clone()
    ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(100 + j), 1.0, pos, logp, grad, acc, None, None, ns)
    N.append(ns.cpu().numpy().astype(np.float64))
N = np.array(N)
print("corr of per-chain attempts between keys (same state):", np.round(np.corrcoef(N)[0], 3))
print("mean %.1f std %.1f max %s" % (N.mean(), N.std(), N.max(1)))


def tile_time(n):                     # n: [tiles, 16] -> k cycles per tile
    s = np.sort(n, axis=1)[:, ::-1]
    return 130 * s[:, 0] + 90 * s[:, 2] + 45 * s[:, 8]


def assign(pred):
    """two slow + six medium + eight fast chains per tile; the slowest tiles get the fastest of the other groups"""
    o = np.argsort(-pred)
    T = B // 16
    slow, med, fast = o[:2 * T].reshape(T, 2), o[2 * T:8 * T], o[8 * T:]
    med = med[::-1].reshape(T, 6); fast = fast[::-1].reshape(T, 8)
    return np.concatenate([slow, med, fast], axis=1)


for j in range(1, 6):
    cur = tile_time(N[j].reshape(-1, 16))
    per = tile_time(N[j][assign(N[j])])
    prd = tile_time(N[j][assign(N[j - 1])])
    avg = tile_time(N[j][assign(N[:j].mean(0))])
    print(f"key {j}: model max tile time (M cycles): contiguous {cur.max()/1e3:.1f} (mean {cur.mean()/1e3:.1f}) | perfect foresight {per.max()/1e3:.1f} | "
          f"predicted from the previous key {prd.max()/1e3:.1f} | from the mean of all previous {avg.max()/1e3:.1f}")
