"""Per-layer gradient error of a rank slice (debug aid for tests/test_gpu_rank_slices.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import fm, prng
from tests.test_gpu_rank_slices import _setup, _dev
from tests import gpu_util as gu
for case, rank in [("pines-8192", 0), ("pines-8192", 7)]:
    g_, args, dist, model, params, x32, ctx, off, c = _setup(case, rank)
    kf = prng.PRNGKey(60 + rank)
    pos = _dev(x32)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(kf, pos, loss, grads)
    lo, go = fm.loss_and_grad(model, params, kf, x32.astype(np.float64), args.sigma, n_total=c["n_total"], start=off)
    print(case, rank, "loss", loss.item(), lo)
    gg = gu.unflat_params(model, grads.cpu().numpy())
    for i, (a, b) in enumerate(zip(gg, go)):
        for kk in ("kernel", "bias"):
            e = np.abs(a[kk].astype(np.float64) - b[kk]).max(); m = np.abs(b[kk]).max()
            print(f"  layer {i} {kk}: max err {e:.3e} max {m:.3e} rel {e / max(m, 1e-30):.2e}")
    ctx.close()
