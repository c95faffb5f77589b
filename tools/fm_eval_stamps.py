"""Section cycle counts of fm_eval64_kernel on the configs[1] eval batch (development build: tools/build_variant.sh fmstamps
-DMFM_FM_STAMPS; run with MFM_LIB=.../libmfm_hip_fmstamps.so)."""
import ctypes as C, os, sys
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
buf = torch.zeros(12800 * 32, dtype=torch.int64, device="cuda")
ctx.lib.mfm_debug_fm_buffer.argtypes = [C.c_void_p]
assert ctx.lib.mfm_debug_fm_buffer(C.c_void_p(buf.data_ptr())) == 0
for _ in range(2):
    ctx.fm_loss(prng.PRNGKey(1), xs, loss, n_total=n_eval)
torch.cuda.synchronize()
rows = int(os.environ.get("MFM_EVAL_ROWS", "32"))
s = buf.cpu().numpy().reshape(12800, 32)[:n_eval // rows, :11].astype(np.float64)
dlt = np.diff(s, axis=1)
names = ["zero + t + cond/target draws", "fourier features + barrier", "gmm grad (8 lanes per wave)", "t1 job", "x1 job + barrier", "st, sx jobs + barrier",
         "gate, j1 jobs + barrier", "j2 job + barrier", "out job + loss", "loss reduction"]
for nme, v in zip(names, np.median(dlt, axis=0)):
    print(f"{nme:32s} {v:10.0f} cycles")
print("total cycles per workgroup (median)", np.median(s[:, 10] - s[:, 0]), " start spread of the grid (cycles)", s[:, 0].max() - s[:, 0].min())
