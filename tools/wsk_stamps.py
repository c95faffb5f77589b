"""Section time stamps of wgrad_sk_kernel (development build: tools/build_variant.sh wsk -DMFM_WSK_STAMPS; run with
MFM_LIB=.../libmfm_hip_wsk.so).  Stamps: 0 start, 1 ring primed (issue only), 2 second stage, 3 main loop done, 4 partials published
(drained + barrier), 5 ticket drawn, 6 last arriver: partials summed, 7 block updated, 8 end."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import prng
from tests import gpu_util as gu
from mfm_amd import _lib
B, d = 4096, 256
args, dist, k, model, state = gu.phi4_setup(d=d, B=B, learning_iter=10000)
params = gu.rand_params(model, seed=1, out_scale=0.05)
ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
pos = torch.from_numpy(dist.init_params.astype(np.float32)).cuda()
logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda"); acc = torch.empty(B, device="cuda")
loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
ctx.mala_init(pos, 1.0, logp, grad)
buf = torch.zeros(512 * 16, dtype=torch.int64, device="cuda")
ctx.lib.mfm_debug_wsk_buffer.argtypes = [C.c_void_p]
assert ctx.lib.mfm_debug_wsk_buffer(C.c_void_p(buf.data_ptr())) == 0
for i in range(40):
    kg, kt = prng.split(prng.PRNGKey(i), 2)
    ctx.train_iter(i + 1, 100, _lib.FLOW_RWMH, kg, kt, 1.0, 1e-4, pos, logp, grad, loss, grads, acc=acc)
torch.cuda.synchronize()
z = buf.cpu().numpy().reshape(512, 16).astype(np.float64)
t0 = z[:, 0].min()
names = ["start", "ring primed", "2nd stage", "loop done", "published", "ticket", "summed (last arrivers)", "updated (last arrivers)", "end"]
print("stamp                      mean      min      max   [cycles since the first workgroup's start; 100 MHz s_memtime? see below]")
for i, nm in enumerate(names):
    v = z[:, i][z[:, i] > 0] - t0
    if len(v): print(f"{nm:24s} {v.mean():8.0f} {v.min():8.0f} {v.max():8.0f}   n={len(v)}")
print("per-workgroup sections (mean): prime %.0f  first stage wait %.0f  loop %.0f  publish %.0f  ticket %.0f" % (
    (z[:, 1] - z[:, 0]).mean(), (z[:, 2] - z[:, 1]).mean(), (z[:, 3] - z[:, 2]).mean(), (z[:, 4] - z[:, 3]).mean(), (z[:, 5] - z[:, 4]).mean()))
la = z[:, 6] > 0
print("last arrivers: %d; ticket -> summed %.0f, summed -> updated %.0f" % (la.sum(), (z[la, 6] - z[la, 5]).mean(), (z[la, 7] - z[la, 6]).mean()))
