import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ode, prng
from tests import gpu_util as gu
for (d,hidden,F,gscale) in [(256,128,128,0.02),(64,32,16,1e-3)]:
    B=32
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F)
    params = gu.rand_params(model, seed=9, out_scale=0.5)
    params[4]["kernel"] *= gscale; params[4]["bias"] *= gscale
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x32 = dist.init_params.astype(np.float32)
    keys = prng.split(prng.PRNGKey(21), B)
    for direction, fn in ((1, ode.transform_and_logdet), (-1, ode.inverse_and_logdet)):
        st = {}
        y_o, l_o = fn(model, params, keys, x32.astype(np.float64), True, args.rtol, args.atol, args.mxstep, stats=st)
        st2 = {}
        y_t, l_t = fn(model, params, keys, x32.astype(np.float64), True, 1e-8, 1e-8, 100000, stats=st2)
        out = torch.empty(B, d, device="cuda", dtype=torch.float32); ldj = torch.empty(B, device="cuda", dtype=torch.float32)
        ns = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.ode_transform(direction, torch.from_numpy(x32).cuda(), out, ldj, keys=torch.from_numpy(keys.astype(np.uint32).view(np.int32)).cuda(), nsteps=ns)
        y, l, n = out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy()
        print(f"d={d} gscale={gscale} dir={direction}: natt oracle mean {st['n_attempted'].mean():.1f} gpu {n.mean():.1f} | "
              f"|y_gpu-y_or| {np.abs(y-y_o).max():.2e} |y_or-y_true| {np.abs(y_o-y_t).max():.2e} |y_gpu-y_true| {np.abs(y-y_t).max():.2e} | "
              f"|l_gpu-l_or| {np.abs(l-l_o).max():.2e} |l_or-l_true| {np.abs(l_o-l_t).max():.2e} |l_gpu-l_true| {np.abs(l-l_t).max():.2e} max|l| {np.abs(l_t).max():.2e} mean|dy| {np.abs(y-y_o).mean():.2e} dn {np.abs(n-st['n_attempted']).max()}", flush=True)
    ctx.close()
