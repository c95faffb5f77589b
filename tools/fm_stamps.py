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
from mfm_amd import _lib
key = prng.PRNGKey(1)
if "--prefetch" in sys.argv:      # the benchmarked regime: the draws come from the buffer the preceding flow step filled (noise.hip)
    logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    acc = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    assert ctx.noise_prefetch(np.stack([prng.PRNGKey(7)]), np.stack([key]))
    p2 = pos.clone()
    ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(100), 1.0, p2, logp, grad, acc, None, None, ns)
if "--lg" in sys.argv:              # a build with four stamps per layer_gemm tile and wave (development only; see git history of this tool)
    lg = torch.zeros(256 * 8 * 128, dtype=torch.int64, device="cuda")
    ctx.lib.mfm_debug_lg_buffer.argtypes = [C.c_void_p]
    assert ctx.lib.mfm_debug_lg_buffer(C.c_void_p(lg.data_ptr())) == 0
    for _ in range(40):                    # the loop's regime: sustained clocks, the weights rewritten before every training kernel
        ctx.fm_loss_grad(key, pos, loss, grads)
        if "--no-update" not in sys.argv:
            ctx.adamw_step(grads)
    torch.cuda.synchronize()
    z = lg.cpu().numpy().reshape(256, 8, 32, 4).astype(np.float64)
    names = ["t1", "x1", "t2", "x2", "gate q0", "gate q1", "j1", "j2", "out q0", "out q1", "d j2", "d j1", "d cat q0", "d cat q1", "d st", "d x1", "d t1"]
    print("tile            first data   MFMAs   epilogue | wave skew at entry / at exit (max - min over the 8 waves)   [cycles, mean over workgroups]")
    for i, nm in enumerate(names):
        t = z[:, :, i, :]
        if not t[:, :, 3].all(): break
        d = np.diff(t, axis=2).mean((0, 1))
        print(f"{nm:12s} {d[0]:10.0f} {d[1]:9.0f} {d[2]:9.0f} | {np.ptp(t[:, :, 0], axis=1).mean():8.0f} {np.ptp(t[:, :, 3], axis=1).mean():8.0f}")
    sys.exit(0)
if "--loop" in sys.argv:            # the benchmarked regime end to end: a flow step that produces the draws, then K iterations of
    K = 100                        # mfm_train_iter (MALA step inside the training kernel, optimizer step); stamps of the LAST iteration
    logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    acc = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    kk = np.stack([prng.split(prng.PRNGKey(1000 + i), 2) for i in range(K)]).astype(np.uint32)
    assert ctx.noise_prefetch(kk[:, 0], kk[:, 1])
    ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(100), 1.0, pos, logp, grad, acc, None, None, ns)
    for i in range(K):
        ctx.train_iter(i + 1, K, _lib.FLOW_RWMH, kk[i, 0], kk[i, 1], 1.0, 1e-4, pos, logp, grad, loss, grads, acc=acc)
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(256, 32).astype(np.float64)
    s = raw[:, [0, 6, 7, 8, 1, 2, 3, 4, 5]]
    for n, v in zip(["zero pads + issue loads", "MALA step (wave 0)", "barrier", "batch construction", "fourier", "forward 7 layers", "out layer+loss", "backward 6 dgrads"],
                    np.diff(s, axis=1).mean(0)):
        print(f"{n:26s} {v:10.0f}")
    tot = s[:, -1] - s[:, 0]
    print("total: mean", tot.mean(), "min", tot.min(), "max", tot.max())
    if raw[:, 10].all():           # stamps inside mala_chain_step (wave 0)
        f = raw[:, [6, 10, 11, 12, 13, 14, 7]]
        for n, v in zip(["  MALA: issue own loads", "  MALA: batch loads issued, times, Fourier features, pads", "  MALA: proposal (waits for the loads)", "  MALA: value + gradient at the proposal",
                         "  MALA: four float64 wave sums", "  MALA: accept + stores"], np.diff(f, axis=1).mean(0)):
            print(f"{n:44s} {v:10.0f}")
    sys.exit(0)
if "--mala" in sys.argv:            # mfm_train_iter: the MALA step inside the training kernel (stamps 6 / 7 / 8 around it)
    if "--prefetch" not in sys.argv:
        logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda")
        ctx.mala_init(pos, 1.0, logp, grad)
        acc = torch.empty(B, device="cuda")
    for _ in range(1 if "--prefetch" in sys.argv else 3):
        # with the optimizer step, as in the loop: the kernel then finds the weights freshly rewritten (cold in every XCD's L2)
        ctx.train_iter(1, 100, _lib.FLOW_RWMH, prng.PRNGKey(7), key, 1.0, 1e-4, pos, logp, grad, loss, grads, acc=acc, apply_update="--no-update" not in sys.argv)
    torch.cuda.synchronize()
    if "--fine" in sys.argv:       # a build with stamps 10..13 inside mala_chain_step
        s = buf.cpu().numpy().reshape(256, 32).astype(np.float64)[:, [6, 10, 11, 12, 13, 7]]
        for n, v in zip(["issue loads, keys", "proposal (waits for the loads)", "value + gradient at the proposal", "two float64 wave sums per chain", "accept + stores"], np.diff(s, axis=1).mean(0)):
            print(f"{n:36s} {v:10.0f}")
        sys.exit(0)
    s = buf.cpu().numpy().reshape(256, 32).astype(np.float64)[:, [0, 6, 7, 8, 1, 2, 3, 4, 5]]
    for n, v in zip(["zero pads + issue loads", "MALA step (wave 0)", "barrier", "batch construction", "fourier", "forward 7 layers", "out layer+loss", "backward 6 dgrads"],
                    np.diff(s, axis=1).mean(0)):
        print(f"{n:26s} {v:10.0f}")
    tot = s[:, -1] - s[:, 0]
    print("total: mean", tot.mean(), "min", tot.min(), "max", tot.max())
    t0 = s[:, 0].min()
    print("workgroup start after the first one: mean", (s[:, 0] - t0).mean(), "max", (s[:, 0] - t0).max(), "| last end", (s[:, -1] - t0).max(),
          "(if the counter is shared by the CUs)")
    sys.exit(0)
for _ in range(3):
    ctx.fm_loss_grad(key, pos, loss, grads)
torch.cuda.synchronize()
if "--fine" in sys.argv:           # a build whose every layer_gemm call and barrier is followed by a stamp (ids in program order below)
    order = [0, 1, 2, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 3, 18, 19, 4, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 5]
    what = ["prologue", "fourier", "-", "t1", "x1", "barrier", "t2", "x2", "barrier", "gate", "j1", "barrier", "j2", "barrier", "-", "out+loss",
            "wave_sum+barrier", "loss write", "d j2", "barrier", "d j1", "barrier", "d cat", "barrier", "d st (gate)", "barrier", "d x1", "d t1", "-"]
    s = buf.cpu().numpy().reshape(256, 32).astype(np.float64)[:, order]
    for n, v in zip(what, np.diff(s, axis=1).mean(0)):
        print(f"{n:20s} {v:9.0f}")
    print("total", (s[:, -1] - s[:, 0]).mean())
    sys.exit(0)
s = buf.cpu().numpy().reshape(256, 32)[:, :6].astype(np.float64)
dlt = np.diff(s, axis=1)
names = ["prologue draws+cond", "fourier+gmm", "forward 7 layers", "out layer+loss", "backward 6 dgrads"]
for n, v in zip(names, dlt.mean(0)):
    print(f"{n:24s} {v:10.0f} shader-clock cycles")
print("total cycles per workgroup", (s[:, 5] - s[:, 0]).mean())
