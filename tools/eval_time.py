"""eval_step kernel on the configs[1] batch (409,600 samples): average launch time under a few settings (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import prng
from tests import gpu_util as gu
n_eval = 409600
args, dist, k, model, state = gu.gmm16_setup(B=4096, hutchs=False)
params = gu.rand_params(model, seed=3)
ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, max_eval=n_eval)
xs = torch.randn(n_eval, 2, device="cuda") * 8
loss = torch.zeros(1, dtype=torch.float64, device="cuda")
for setting in sys.argv[1:] or ["MFM_EVAL_STAGGER=0"]:
    for kv in setting.split(","):
        kk, vv = kv.split("="); os.environ[kk] = vv
    for _ in range(3): ctx.fm_loss(prng.PRNGKey(1), xs, loss, n_total=n_eval)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ctx.fm_loss(prng.PRNGKey(1), xs, loss, n_total=n_eval)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{setting}: {(t1 - t0) / 20 * 1e3:.4f} ms per call, loss {loss.item():.6e}")
