import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import fm, prng
from tests import gpu_util as gu
d,B,hidden,F = [int(x) for x in sys.argv[1:5]] if len(sys.argv)>4 else (256,64,128,128)
args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F)
params = gu.rand_params(model, seed=3)
ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
x32 = dist.init_params.astype(np.float32)
key = prng.PRNGKey(11)
loss_o, grads_o = fm.loss_and_grad(model, params, key, x32.astype(np.float64), args.sigma)
loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
for rep in range(2):
    ctx.fm_loss_grad(key, torch.from_numpy(x32).cuda(), loss, grads)
    g = gu.unflat_params(model, grads.cpu().numpy())
    print("loss", loss.item(), loss_o)
    for i,(gg,go) in enumerate(zip(g,grads_o)):
        for kk in ("kernel","bias"):
            a,b = gg[kk].astype(np.float64), go[kk].astype(np.float64)
            print(i, kk, "relerr %.3e"%(np.abs(a-b).max()/(np.abs(b).max()+1e-30)), "max|g| %.3e"%np.abs(b).max(), "nan", np.isnan(a).sum())
