"""Per-chain differences of the flow step on a prescribed step sequence (development aid for the tail modes of fast::solve2)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import prng
from tests import gpu_util as gu
from tests.test_gpu_replay import _flow_replay_raw, _tamed
B = 16
args, dist, k, model, state = gu.phi4_setup(d=256, B=B)
params = _tamed(model, out_scale=2.0)
ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
r = _flow_replay_raw(ctx, model, params, args, dist, 0.8, dist.init_params.astype(np.float32), prng.PRNGKey(31))
so, dg = r["so"], r["diag"]
order = np.argsort(r["n_o"])
print("chain n_inv n_fwd total | |dx'| dvol0 dvolp")
for b in order:
    print(f"{b:3d} {so['n_att_inv'][b]:4d} {so['n_att_fwd'][b]:4d} {r['n_o'][b]:4d} | {np.abs(r['prop'][b] - r['info_o'].proposed_position[b]).max():.2e} {abs(dg[b,0]-so['vol0'][b]):.2e} {abs(dg[b,1]-so['volp'][b]):.2e}  gpu n {r['n_g'][b]}")
