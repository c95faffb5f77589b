"""Development aid: the deferred (suspicious-gradient) path of wgrad_sk_kernel against the separate calls, iteration by iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MFM_DEBUG_FORCE_EXCHANGE"] = "1"
import numpy as np, torch
from oracle import prng
from tests import gpu_util as gu
from mfm_amd._lib import FLOW_RWMH
args, dist, k, model, state = gu.phi4_setup(d=256, B=64, learning_iter=20)
params = gu.rand_params(model, seed=3, out_scale=0.05)
x0 = dist.init_params.astype(np.float32)
out = []
for one_call in (False, True):
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    pos = torch.from_numpy(x0).cuda(); logp = torch.empty(64, device="cuda", dtype=torch.float64); grad = torch.empty_like(pos)
    acc = torch.empty(64, device="cuda"); loss = torch.zeros(1, device="cuda", dtype=torch.float64); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    ks, tr = prng.PRNGKey(5), []
    for count in range(1, 6):
        ks, kg, kt = prng.split(ks, 3)
        if one_call:
            ctx.train_iter(count, 100, FLOW_RWMH, kg, kt, 1.0, args.step_size, pos, logp, grad, loss, grads, acc=acc)
        else:
            ctx.mala_step(kg, 1.0, args.step_size, pos, logp, grad, acc); ctx.fm_loss_grad(kt, pos, loss, grads); ctx.adamw_step(grads)
        tr.append((ctx.opt_state(), ctx.get_params().copy(), grads.cpu().numpy().copy(), loss.item()))
    out.append(tr); ctx.close()
for i, ((s0, p0, g0, l0), (s1, p1, g1, l1)) in enumerate(zip(*out)):
    bad = np.flatnonzero(p0 != p1)
    print(i, s0 == s1, "loss", l0 == l1, "grads differ", (g0 != g1).sum(), "params differ", len(bad), "first", bad[:5], "max", np.abs(p0 - p1).max())
    if len(bad):
        sh = model.layer_shapes(); off = 0
        for li, (K, N) in enumerate(sh):
            nk = K * N; cnt = ((bad >= off) & (bad < off + nk)).sum(); cb = ((bad >= off + nk) & (bad < off + nk + N)).sum()
            print("   layer", li, (K, N), "kernel diffs", cnt, "bias diffs", cb); off += nk + N
