"""Ad-hoc kernel timings on the GPU box (development aid; bench.py is the judged benchmark)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import prng
from tests import gpu_util as gu
from mfm_amd import _lib

B, d = 4096, 256
args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
params = gu.rand_params(model, seed=1, out_scale=0.05)
ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
pos = torch.from_numpy(dist.init_params.astype(np.float32)).cuda()
logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
acc = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
ctx.mala_init(pos, 1.0, logp, grad)
key = prng.PRNGKey(1)

def timeit(name, fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:20s} {e0.elapsed_time(e1) / n * 1e3:10.1f} us")

timeit("mala_step", lambda: ctx.mala_step(key, 1.0, 1e-4, pos, logp, grad, acc), 50)
timeit("fm_loss_grad", lambda: ctx.fm_loss_grad(key, pos, loss, grads), 20)
timeit("fm_loss(eval)", lambda: ctx.fm_loss(key, pos, loss), 20)
timeit("adamw", lambda: ctx.adamw_step(grads), 20)
if not os.environ.get("QT_NOFLOW"): timeit("flow_step", lambda: ctx.flow_step(_lib.FLOW_RWMH, key, 1.0, pos, logp, grad, acc, None, None, ns), 3)
print("mean attempts/2 solves", ns.float().mean().item())
