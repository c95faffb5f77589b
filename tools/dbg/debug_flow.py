import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import flow, mala, prng, targets
from tests import gpu_util as gu
from mfm_amd import _lib
d, hidden, F, B = 64, 32, 16, 32
args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F)
params = gu.rand_params(model, seed=9, out_scale=0.05)
params[4]["kernel"] *= 1e-3; params[4]["bias"] *= 1e-3
beta = 0.8
vg = targets.Tempered(dist, beta).value_and_grad
x32 = dist.init_params.astype(np.float32)
key = prng.PRNGKey(31)
res = []
for rep in range(3):
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    pos = torch.from_numpy(x32).cuda(); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda", dtype=torch.float32)
    ctx.mala_init(pos, beta, logp, grad)
    if rep == 0:
        st = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
        stats = {}
        new, info = flow.rwmh_step(prng.split(key, B), st, vg, model, params, args, stats)
    acc = torch.empty(B, device="cuda", dtype=torch.float32); prop = torch.empty(B, d, device="cuda", dtype=torch.float32); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, key, beta, pos, logp, grad, acc, None, prop, ns)
    p = prop.cpu().numpy(); res.append(p)
    err = np.abs(p - info.proposed_position).max(1)
    print("rep", rep, "row maxerr:", np.array2string(err, precision=1, max_line_width=250))
    print("   natt gpu", ns.cpu().numpy(), "\n   natt ora", stats["n_att_inv"] + stats["n_att_fwd"])
    ctx.close()
print("deterministic:", np.array_equal(res[0], res[1]), np.array_equal(res[1], res[2]))
