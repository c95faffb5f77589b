"""Development aid: per-phase cycle stamps of ONE field evaluation (steady state) in the solver tile."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MFM_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mfm_amd/lib/libmfm_hip_stamps.so")
import numpy as np, torch
np.set_printoptions(suppress=True, linewidth=250)
from tests import gpu_util as gu
from mfm_amd import _lib
if os.environ.get("EVAL_STAMPS_CFG") == "gmm":          # the 16-mode mixture (d = 2), Hutchinson-style single tangent pass
    B, d = 4096, 2
    args, dist, k, model, state = gu.gmm16_setup(B=B, hutchs=True)
else:
    B, d = 4096, 256
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
params = gu.rand_params(model, seed=1, out_scale=0.05)
ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
x = torch.from_numpy(dist.init_params.astype(np.float32)).cuda(); t = torch.rand(B, device="cuda"); z = torch.randn(B, d, device="cuda")
NW = int(os.environ.get("ODE_NW", "8"))
st = torch.zeros(B // 16 * NW * 16, dtype=torch.int64, device="cuda")
fn = ctx.lib.mfm_debug_eval_stamps
fn.restype = C.c_int; fn.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]
for reps in (20,):
    rc = fn(ctx.h, x.data_ptr(), t.data_ptr(), z.data_ptr(), B, reps, st.data_ptr()); assert rc == 0
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(B // 16, NW, 16)[:, :, :14].astype(np.float64)
    rel = s - s[:, :, :1].min(axis=1, keepdims=True)         # relative to the earliest wave of the workgroup
    names = ["start", "", "pre-bar1(four+gmm)", "post-bar1", "pre-bar2(t1,x1)", "post-bar2", "pre-bar3(t2,x2)", "post-bar3", "pre-bar4(gate,j1)", "post-bar4",
             "pre-bar5(j2)", "post-bar5", "pre-reduce(out)", "post-reduce"]
    med = np.median(rel, axis=0)        # [wave, stamp]
    print("median cycles since eval start, per wave (rows) / stamp (cols):")
    print("stamps:", [n for i, n in enumerate(names) if i != 1])
    for w in range(NW):
        print("wave", w, np.delete(med[w], 1).astype(int))
    tot = np.median(rel[:, :, 13].max(1))
    dur = np.diff(np.delete(med, 1, axis=1), axis=1).astype(int)
    print("phase durations wave0:", dur[0], "\nphase durations wave4:", dur[4])
    print("median eval cycles (last wave):", tot, " ideal MFMA cycles/SIMD: 38912")
