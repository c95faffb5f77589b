import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mfm_amd import exe_flow_matching as E, random as jr
from mfm_amd._lib import FLOW_RWMH
from mfm_amd.distributions import PhiFour
from mfm_amd.engine import Engine
args = bench.make_args(4096, 10000)
dist = PhiFour(256)
k = jr.split(jr.PRNGKey(1), 6)
dist.initialize_model(k[3], 4096)
fourier = jr.normal(k[4], (128,))
eng = Engine(dist, args, fourier)
model = E.VectorFieldNet(fourier, dist.grad_logprob, args.hidden_x, args.hidden_t, args.hidden_xt).attach(eng)
eng.ctx.set_params(E.flatten_params(model.init(k[2])))
ctx = eng.ctx
pos = eng.local(dist.init_params); logp = torch.empty(4096, device="cuda", dtype=torch.float64); grad = torch.empty_like(pos)
acc = torch.empty(4096, device="cuda", dtype=torch.float32); nst = torch.zeros(4096, device="cuda", dtype=torch.int32)
ctx.mala_init(pos, 1.0, logp, grad)
ks = k[1]; hist = []
for count in range(1, 405):
    ks, kg, kt = jr.split(ks, 3)
    if count % 101 == 0:
        ctx.flow_step(FLOW_RWMH, kg, 1.0, pos, logp, grad, acc, None, None, nst)
        n = nst.cpu().numpy().astype(float); hist.append(n)
        t = n.reshape(-1, 16)
        print(f"flow step @ {count}: mean natt {n.mean():.1f} tile-max mean {t.max(1).mean():.1f} ratio {t.max(1).mean()/n.mean():.3f} | sorted-oracle ratio {np.sort(n).reshape(-1,16).max(1).mean()/n.mean():.3f} acc mean {acc.mean().item():.3g}")
        if len(hist) > 1:
            print("   corr with previous:", np.corrcoef(hist[-2], hist[-1])[0, 1], " ratio if sorted by previous:", n[np.argsort(hist[-2])].reshape(-1, 16).max(1).mean() / n.mean())
    else:
        ctx.mala_step(kg, 1.0, args.step_size, pos, logp, grad, acc)
    eng.train_step(kt, pos)
