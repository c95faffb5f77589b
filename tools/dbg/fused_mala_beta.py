"""Development aid: mfm_train_iter (MALA step inside the training kernel) against the separate calls at beta < 1, iteration by iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import prng
from tests import gpu_util as gu
from mfm_amd._lib import FLOW_RWMH
args, dist, k, model, state = gu.phi4_setup(d=256, B=64, learning_iter=20)
params = gu.rand_params(model, seed=3, out_scale=0.05)
x0 = dist.init_params.astype(np.float32)
beta = float(sys.argv[1]) if len(sys.argv) > 1 else 0.37
res = []
for fused in (False, True):
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    pos = torch.from_numpy(x0).cuda(); logp = torch.empty(64, device="cuda", dtype=torch.float64); grad = torch.empty_like(pos)
    acc = torch.empty(64, device="cuda"); loss = torch.zeros(1, device="cuda", dtype=torch.float64); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.mala_init(pos, beta, logp, grad)
    ks = prng.PRNGKey(5); tr = []
    import ctypes as C
    dbg = torch.zeros(64 * 8, dtype=torch.float64, device="cuda")
    if hasattr(ctx.lib, "mfm_debug_mala_buffer"):
        ctx.lib.mfm_debug_mala_buffer.argtypes = [C.c_void_p]; ctx.lib.mfm_debug_mala_buffer(C.c_void_p(dbg.data_ptr()))
    for count in range(1, 4):
        ks, kg, kt = prng.split(ks, 3)
        if fused:
            ctx.train_iter(count, 100, FLOW_RWMH, kg, kt, beta, args.step_size, pos, logp, grad, loss, grads, acc=acc)
        else:
            ctx.mala_step(kg, beta, args.step_size, pos, logp, grad, acc); ctx.fm_loss_grad(kt, pos, loss, grads); ctx.adamw_step(grads)
        if count == 1: dbgs = globals().setdefault("dbgs", []); dbgs.append(dbg.cpu().numpy().reshape(64, 8).copy())
        tr.append((acc.cpu().numpy().copy(), pos.cpu().numpy().copy(), logp.cpu().numpy().copy(), grad.cpu().numpy().copy(), loss.item()))
    res.append(tr); ctx.close()
for i, (a, b) in enumerate(zip(*res)):
    print(i, [int((u != v).sum()) for u, v in zip(a[:4], b[:4])], a[4] == b[4], np.abs(a[0].astype(np.float64) - b[0]).max())
a, b = res[0][0][0], res[1][0][0]
print("acc separate", a[:6], "\nacc one call", b[:6], "\nrelative", (np.abs(a.astype(np.float64) - b) / np.abs(a))[:6])

if "dbgs" in globals() and len(dbgs) == 2:
    np.set_printoptions(precision=17, linewidth=200)
    for nm, col in zip(["th1", "th2", "lpn", "lp", "delta", "inv4e", "gn0", "xn0"], range(8)):
        u, v = dbgs[0][:, col], dbgs[1][:, col]
        print(nm, "max rel diff", np.max(np.abs(u - v) / np.maximum(np.abs(u), 1e-300)), u[0], v[0])
