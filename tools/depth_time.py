"""Development aid: what leaving the fused tile family costs.  The headline shape (phi-four d = 256, 4096 chains, F = 128, --hutch) with two
hidden layers of 128 per branch on the fused tile kernels and on the wide family, and with three hidden layers (wide family only):
one flow step and the MALA + training iteration, same parameters' scale, same keys."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(tag):
    import torch
    from mfm_amd import _lib
    from oracle import prng
    from tests import gpu_util as gu
    B, d = 4096, 256
    hid = 128 if tag != "three" else ([128] * 3, [128] * 3, [128] * 3)
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hid, F=128)
    params = gu.rand_params(model, seed=1, out_scale=0.05)
    fam = {"tile": _lib.FAMILY_TILE, "wide": _lib.FAMILY_WIDE, "three": _lib.FAMILY_AUTO}[tag]
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params, family=fam)
    pos0 = torch.from_numpy(dist.init_params.astype(np.float32)).cuda()
    logp0 = torch.empty(B, dtype=torch.float64, device="cuda"); grad0 = torch.empty(B, d, device="cuda")
    ctx.mala_init(pos0, 1.0, logp0, grad0)
    acc = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ts = []
    for j in range(-1, 3):
        pos, logp, grad = pos0.clone(), logp0.clone(), grad0.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.flow_step(_lib.FLOW_RWMH, prng.PRNGKey(100 + max(j, 0)), 1.0, pos, logp, grad, acc, None, None, ns); e1.record()
        torch.cuda.synchronize()
        if j >= 0: ts.append(e0.elapsed_time(e1))
    pos, logp, grad = pos0.clone(), logp0.clone(), grad0.clone()
    it = []
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(50):
            ctx.mala_step(prng.PRNGKey(7 + i), 1.0, 1e-4, pos, logp, grad, acc)
            ctx.fm_loss_grad(prng.PRNGKey(900 + i), pos, loss, grads)
            ctx.adamw_step(grads)
        e1.record(); torch.cuda.synchronize()
        it.append(e0.elapsed_time(e1) / 50)
    print(f"{tag:6s} flow step {np.mean(ts):8.2f} ms (attempts {ns.float().mean().item():.1f}, max {ns.max().item()}) | MALA + training iteration as separate calls {1e3 * min(it):7.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1: one(sys.argv[1])
    else:
        for tag in ("tile", "wide", "three"):
            subprocess.run([sys.executable, os.path.abspath(__file__), tag], check=False)
