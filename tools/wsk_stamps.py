"""Section time stamps of wgrad_sk_kernel (development build: tools/build_variant.sh wsk -DMFM_WSK_STAMPS; run with
MFM_LIB=.../libmfm_hip_wsk.so).  Stamps (s_memrealtime: one 100 MHz clock for the device): 0 start, 1 ring primed (issue only), 2 second
stage, 3 main loop done, 4 partials published (drained + barrier), 5 ticket drawn, 6 every contributor of the block has published,
7 this workgroup's slice combined and updated, 8 end."""
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
buf = torch.zeros(1024 * 16, dtype=torch.int64, device="cuda")
ctx.lib.mfm_debug_wsk_buffer.argtypes = [C.c_void_p]
assert ctx.lib.mfm_debug_wsk_buffer(C.c_void_p(buf.data_ptr())) == 0
for i in range(41):
    if i == 40:
        torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    kg, kt = prng.split(prng.PRNGKey(i), 2)
    ctx.train_iter(i + 1, 100, _lib.FLOW_RWMH, kg, kt, 1.0, 1e-4, pos, logp, grad, loss, grads, acc=acc)
torch.cuda.synchronize()
z = buf.cpu().numpy().reshape(1024, 16)[: int(os.environ.get("MFM_WSK_G", 512))].astype(np.float64) * 0.01       # s_memrealtime ticks (100 MHz) -> us
t0 = z[:, 0].min()
names = ["start", "ring primed", "2nd stage", "loop done", "published", "ticket drawn", "block complete", "slice combined", "end"]
print("stamp [us since the first workgroup's start]     mean      min      max")
for i, nm in enumerate(names):
    v = z[:, i][z[:, i] > 0] - t0
    if len(v): print(f"{nm:24s} {v.mean():8.2f} {v.min():8.2f} {v.max():8.2f}   n={len(v)}")
print("per-workgroup sections (mean us): prime %.2f  first stage wait %.2f  loop %.2f  publish %.2f  ticket %.2f  ticket->end %.2f" % (
    (z[:, 1] - z[:, 0]).mean(), (z[:, 2] - z[:, 1]).mean(), (z[:, 3] - z[:, 2]).mean(), (z[:, 4] - z[:, 3]).mean(), (z[:, 5] - z[:, 4]).mean(), (z[:, 8] - z[:, 5]).mean()))
print("after the ticket (mean us): wait for the block %.2f  combine + update %.2f  rest %.2f" % ((z[:, 6] - z[:, 5]).mean(), (z[:, 7] - z[:, 6]).mean(), (z[:, 8] - z[:, 7]).mean()))
st = np.sort(z[:, 0] - t0)
print("start times: quantiles 0/25/50/75/100 %%: %s; workgroups started within 1 us: %d" % (np.quantile(st, [0, .25, .5, .75, 1]).round(2), (st < 1).sum()))
print("end times quantiles:", np.quantile(z[:, 8] - t0, [0, .25, .5, .75, 1]).round(2))
tot = z[:, 8] - z[:, 0]
for i in np.argsort(tot)[-5:]:
    print("slow workgroup", i, "sections:", np.diff(z[i, :9]).round(2))
