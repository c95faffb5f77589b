import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import mala, prng, targets
from tests import gpu_util as gu
for B, d in ((64, 64), (64, 256)):
  args, dist, k, model, state = gu.lgcp_setup(n=int(np.sqrt(d)), B=B)
  if True:
    ctx = gu.make_ctx(dist, args)
    x32 = dist.init_params.astype(np.float32)
    beta, eps = 0.45, 0.01
    vg = targets.Tempered(dist, beta).value_and_grad
    pos = torch.from_numpy(x32).cuda(); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda", dtype=torch.float32)
    ctx.mala_init(pos, beta, logp, grad)
    st = mala.init(x32.astype(np.float64), vg)
    print("init logp err", np.abs(logp.cpu().numpy() - st.logdensity).max(), "grad err cols", np.nonzero(np.abs(grad.cpu().numpy() - st.logdensity_grad).max(0) > 1e-2)[0])
    ll = torch.empty(B, dtype=torch.float64, device='cuda'); ctx.loglik(pos, ll); print('loglik err', np.abs(ll.cpu().numpy() - dist.loglik(x32.astype(np.float64))).max())
    key = prng.PRNGKey(77)
    st_in = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    new, info, u = mala.kernel(prng.split(key, B), st_in, vg, eps)
    prop = torch.empty(B, d, device="cuda", dtype=torch.float32)
    acc = torch.empty(B, device='cuda'); isacc = torch.empty(B, dtype=torch.uint8, device='cuda'); w = torch.empty(B, device='cuda')
    ctx.mala_step(key, beta, eps, pos, logp, grad, acc, isacc, prop, w)
    e = np.abs(prop.cpu().numpy() - info.proposed_position)
    print("bad cols", np.nonzero(e.max(0) > 1e-5)[0], "bad rows", np.nonzero(e.max(1) > 1e-5)[0][:20], e.max())
