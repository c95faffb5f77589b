"""MALA + training iteration of the headline configuration in the benchmarked regime (development aid; bench.py is the judged one).

A flow step produces the draws of the K iterations that follow (noise.hip), then K = 100 mfm_train_iter calls are timed as a block
with HIP events (wall per iteration: what bench.py's `iteration_ms_excluding_flow_kernel` sees) and, in a second pass, kernel by
kernel (`mfm_profile`).  Prints the parameters' checksum so that two builds / switches can be compared for bit-identity.
Usage: python tools/iter_time.py [--cycles N]          (environment switches of common.hip.h apply: MFM_WGRAD_SLABS=1, MFM_WSK_XCD=1 ...)"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import prng
from tests import gpu_util as gu
from mfm_amd import _lib

B, d, K = 4096, 256, 100
cycles = int(sys.argv[sys.argv.index("--cycles") + 1]) if "--cycles" in sys.argv else 3
args, dist, k, model, state = gu.phi4_setup(d=d, B=B, learning_iter=10000)
params = gu.rand_params(model, seed=1, out_scale=0.05)
params[4]["kernel"] *= 1e-3              # a tame field: the flow steps here only produce draws, they are not what is timed
ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
pos = torch.from_numpy(dist.init_params.astype(np.float32)).cuda()
logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
acc = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
ctx.mala_init(pos, 1.0, logp, grad)


def cycle(c, timed):
    kk = np.stack([prng.split(prng.PRNGKey(1000 * c + i), 2) for i in range(K)]).astype(np.uint32)
    assert ctx.noise_prefetch(kk[:, 0], kk[:, 1])
    ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(100 + c), 1.0, pos, logp, grad, acc, None, None, ns)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.sync(); torch.cuda.synchronize()
    e0.record()
    for i in range(K):
        ctx.train_iter(i + 1, K, _lib.FLOW_RWMH, kk[i, 0], kk[i, 1], 1.0, 1e-4, pos, logp, grad, loss, grads, acc=acc)
    e1.record(); ctx.sync(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3


cycle(0, False)
walls = [cycle(c, True) for c in range(1, 1 + cycles)]
print("wall per iteration [us]:", " ".join(f"{w:.1f}" for w in walls), " min %.1f" % min(walls))
ctx.profile(True)
cycle(99, True)
pr = ctx.profile_read()
ctx.profile(False)
print("kernels (HIP events around each launch) [us]:", {n: round(v["ms"] / v["launches"] * 1e3, 2) for n, v in pr.items() if n != "flow_step"})
p = ctx.get_params()
print("loss %.10g  params sha %s  finite %s  opt %s" % (loss.item(), hashlib.sha1(p.tobytes()).hexdigest()[:12], np.isfinite(p).all(), ctx.opt_state()))
