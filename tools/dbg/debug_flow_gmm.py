import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import flow, mala, prng, targets, ode
from tests import gpu_util as gu
from mfm_amd import _lib
B = 32
args, dist, k, model, state = gu.gmm4_setup(B=B, hutchs=True)
d = 2
params = gu.rand_params(model, seed=9, out_scale=0.3)
beta = 0.8
vg = targets.Tempered(dist, beta).value_and_grad
x32 = dist.init_params.astype(np.float32)
key = prng.PRNGKey(31)
ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
# plain transforms first
keys = prng.split(prng.PRNGKey(21), B)
for direction, fn in ((1, ode.transform_and_logdet), (-1, ode.inverse_and_logdet)):
    st = {}
    y_o, l_o = fn(model, params, keys, x32.astype(np.float64), True, args.rtol, args.atol, args.mxstep, stats=st)
    out = torch.empty(B, d, device="cuda", dtype=torch.float32); ldj = torch.empty(B, device="cuda", dtype=torch.float32); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.ode_transform(direction, torch.from_numpy(x32).cuda(), out, ldj, keys=torch.from_numpy(keys.astype(np.uint32).view(np.int32)).cuda(), nsteps=ns)
    print("dir", direction, "max|dy|", np.abs(out.cpu().numpy() - y_o).max(), "max|dl|", np.abs(ldj.cpu().numpy() - l_o).max(), "natt", ns.cpu().numpy()[:8], st["n_attempted"][:8])
pos = torch.from_numpy(x32).cuda(); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda", dtype=torch.float32)
ctx.mala_init(pos, beta, logp, grad)
st = mala.MALAState(x32.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
stats = {}
new, info = flow.rwmh_step(prng.split(key, B), st, vg, model, params, args, stats)
acc = torch.empty(B, device="cuda", dtype=torch.float32); prop = torch.empty(B, d, device="cuda", dtype=torch.float32); ns = torch.empty(B, dtype=torch.int32, device="cuda")
isacc = torch.empty(B, device="cuda", dtype=torch.uint8)
ctx.flow_step(_lib.FLOW_RWMH, key, beta, pos, logp, grad, acc, isacc, prop, ns)
p = prop.cpu().numpy()
print("prop err per row", np.array2string(np.abs(p - info.proposed_position).max(1), precision=1, max_line_width=250))
print("acc gpu", acc.cpu().numpy()[:8], "\nacc ora", info.acceptance_rate[:8])
print("isacc", isacc.cpu().numpy()[:16], info.is_accepted[:16].astype(int))
print("pos err", np.abs(pos.cpu().numpy() - new.position).max(1)[:16])
print("logp", logp.cpu().numpy()[:6], new.logdensity[:6])
print("grad err", np.abs(grad.cpu().numpy() - new.logdensity_grad).max(1)[:16])
