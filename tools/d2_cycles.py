"""In-kernel cycle accounting of the d = 2 flow step (stamps build: tools/build_variant.sh stamps -DMFM_STAMPS): cycles per field
evaluation, per time batch, and outside both, per tiling (development aid)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MFM_LIB"] = os.path.join(ROOT, "mfm_amd/lib/libmfm_hip_stamps.so")
import numpy as np, torch
from mfm_amd import _lib
from oracle import prng
from tests.test_gpu_d2tile import _setup, _dev
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
gu, args, dist, model, params = _setup("gmm4", B)
rng = np.random.default_rng(0)
x32 = (8.0 * rng.choice([-1.0, 1.0], (B, 2)) + rng.standard_normal((B, 2))).astype(np.float32)
for tile in ("4", "4s"):
    os.environ["MFM_D2_TILE"] = tile
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    nwg = B // 4
    dbg = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
    fn = ctx.lib.mfm_debug_flow_buffer; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
    assert fn(dbg.data_ptr()) == 0
    pos = _dev(x32); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, 2, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    ns = torch.empty(B, dtype=torch.int32, device="cuda")
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p2, l2, g2 = pos.clone(), logp.clone(), grad.clone()
        e0.record(); ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(5), 1.0, p2, l2, g2, None, None, None, ns); e1.record(); torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(nwg, 64).astype(float)
    tot, nev, cev, nprep, cprep = d[:, 0], d[:, 2], d[:, 3], d[:, 5], d[:, 6]
    print(f"tile {tile}: {e0.elapsed_time(e1):.3f} ms, attempts mean {ns.float().mean().item():.1f} max {ns.max().item()}; per WG: evals {nev.mean():.0f} (max {nev.max():.0f}), "
          f"cycles/eval {np.median(cev / nev):.0f}, cycles/time batch {np.median(cprep / np.maximum(nprep, 1)):.0f}, total/WG max {tot.max() / 1e6:.2f} M, "
          f"share eval {np.median(cev / tot):.2f} batch {np.median(cprep / tot):.2f} other {1 - np.median((cev + cprep) / tot):.2f}")
    ctx.close()
