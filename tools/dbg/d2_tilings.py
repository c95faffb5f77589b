"""Attempt counts / outputs of the three d = 2 tilings against each other and the oracle (natural controllers)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ode, prng
from tests.test_gpu_d2tile import _setup, _dev
B, d = 64, 2
gu, args, dist, model, params = _setup("gmm4", B)
rng = np.random.default_rng(3)
x32 = (4.0 * rng.standard_normal((B, d))).astype(np.float32)
for direction in (1, -1):
    fn = ode.transform_and_logdet if direction > 0 else ode.inverse_and_logdet
    st = {}
    yo, lo = fn(model, params, None, x32.astype(np.float64), False, args.rtol, args.atol, args.mxstep, n_ts=args.n_ts, stats=st)
    res = {}
    for tile in ("4", "4s", "16"):
        os.environ["MFM_D2_TILE"] = tile
        ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
        out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.ode_transform(direction, _dev(x32), out, ldj, key=prng.PRNGKey(1), nsteps=ns)
        res[tile] = (out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy())
        ctx.close()
        print(direction, tile, "vs oracle: same counts", (res[tile][2] == st["n_attempted"]).mean(), "mean", res[tile][2].mean(), st["n_attempted"].mean(),
              "|dy|", np.abs(res[tile][0] - yo).max(), "|dl|", np.abs(res[tile][1] - lo).max())
    for a, b in (("4", "16"), ("4s", "16"), ("4", "4s")):
        same = res[a][2] == res[b][2]
        print("   ", a, b, "same", same.mean(), "|dn| max", np.abs(res[a][2] - res[b][2]).max(), "|dy| same", np.abs(res[a][0][same] - res[b][0][same]).max(), "|dy| all", np.abs(res[a][0] - res[b][0]).max())
    print("   counts 4 :", res["4"][2][:24]); print("   counts 16:", res["16"][2][:24]); print("   oracle   :", st["n_attempted"][:24])
