"""Section cycle counts of fm_fwd_bwd_kernel (development build: tools/build_variant.sh fmstamps -DMFM_FM_STAMPS; run with
MFM_LIB=.../libmfm_hip_fmstamps.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import prng
from tests import gpu_util as gu
B, d = 4096, 256
args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
params = gu.rand_params(model, seed=1, out_scale=0.05)
ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
pos = torch.from_numpy(dist.init_params.astype(np.float32)).cuda()
loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
buf = torch.zeros(256 * 32, dtype=torch.int64, device="cuda")
ctx.lib.mfm_debug_fm_buffer.argtypes = [C.c_void_p]
assert ctx.lib.mfm_debug_fm_buffer(C.c_void_p(buf.data_ptr())) == 0
for _ in range(3):
    ctx.fm_loss_grad(prng.PRNGKey(1), pos, loss, grads)
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(256, 32)[:, :6].astype(np.float64)
dlt = np.diff(s, axis=1)
names = ["prologue draws+cond", "fourier+gmm", "forward 7 layers", "out layer+loss", "backward 6 dgrads"]
for n, v in zip(names, dlt.mean(0)):
    print(f"{n:24s} {v:10.0f} shader-clock cycles")
print("total cycles per workgroup", (s[:, 5] - s[:, 0]).mean())
