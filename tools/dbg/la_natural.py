"""Development aid: the terms of log alpha of the kernel's NATURAL flow step (its own controllers), by fixed-point iteration of the replay
instrumentation (prescribe the step sequence the kernel itself chose along the previous prescription until nothing changes), for one
tile of 16 chains of the benchmarked state; beside them libmfm_ref's natural terms for the same chains."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import cref, mala, prng, targets
from tests import gpu_util as gu
from mfm_amd import _lib
tp = gu.train_phi4_like_bench()
B, d = 4096, 256
dist, model, args = tp["dist"], tp["model"], tp["args32"]
params = gu.unflat_params(model, tp["params_flat"])
x32 = tp["pos"]
key = prng.PRNGKey(4242)
cr = cref.CRef(model, params)
vg = targets.Tempered(dist, 1.0).value_and_grad
np.set_printoptions(linewidth=220, precision=2, suppress=True)
for b0 in [int(a) for a in sys.argv[1:]] or [1264]:
    xs = x32[b0:b0 + 16]
    st0 = mala.init(xs.astype(np.float64), vg)
    so = {}
    cr.rwmh_step(prng.split(key, B)[b0:b0 + 16], st0, args, stats=so)
    ctx = gu.make_ctx(dist, args, n_local=16, n_total=B, offset=b0, fourier=model.f, params=params)
    cap = 700
    dt = torch.zeros(2, 16, cap, device="cuda"); acc = torch.zeros(2, 16, cap, dtype=torch.uint8, device="cuda")
    def run(armed):
        pos = torch.as_tensor(xs).cuda(); logp = torch.empty(16, dtype=torch.float64, device="cuda"); grad = torch.empty(16, d, device="cuda")
        ctx.mala_init(pos, 1.0, logp, grad)
        ratio = torch.zeros_like(dt); own = torch.zeros_like(dt); diag = torch.zeros(16, 4, dtype=torch.float64, device="cuda")
        if armed:
            ctx.debug_replay(dt, acc, ratio, own, diag)
        a = torch.empty(16, device="cuda"); isacc = torch.empty(16, dtype=torch.uint8, device="cuda"); prop = torch.empty(16, d, device="cuda"); ns = torch.empty(16, dtype=torch.int32, device="cuda")
        ctx.flow_step(_lib.FLOW_RWMH, key, 1.0, pos, logp, grad, a, isacc, prop, ns)
        torch.cuda.synchronize()
        return ratio, own, diag.cpu().numpy(), a.cpu().numpy().astype(np.float64), ns.cpu().numpy(), prop.cpu().numpy()
    _, _, _, a_nat, n_nat, prop_nat = run(False)
    for it in range(800):
        ratio, own, diag, a_rp, n_rp, prop_rp = run(True)
        ndt = own.clone(); nacc = ((ratio <= 1.0) & (ratio > 0)).to(torch.uint8)      # (ratio 0: column not reached)
        same = bool(torch.equal(ndt, dt) and torch.equal(nacc, acc))
        dt, acc = ndt, nacc
        if same:
            break
    print(f"tile {b0}: fixed point after {it} iterations; attempts natural {n_nat} replayed {n_rp}")
    print("   proposals equal to the natural run's:", np.abs(prop_rp - prop_nat).max(1))
    with np.errstate(divide="ignore"):
        print("   kernel natural log(ratio) ", np.log(a_nat))
    print("   kernel (fixed point) la   ", diag[:, 3])
    print("   oracle natural la         ", so["log_alpha"])
    lp_old = st0.logdensity
    lpn_o = so["log_alpha"] + so["volp"] + lp_old + so["vol0"]
    print("   vol0 kernel - oracle      ", diag[:, 0] - so["vol0"])
    print("   volp kernel - oracle      ", diag[:, 1] - so["volp"])
    print("   lpn  kernel - oracle      ", diag[:, 2] - lpn_o)
    print("   la   kernel - oracle      ", diag[:, 3] - so["log_alpha"])
    ctx.close()
