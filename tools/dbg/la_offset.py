"""Development aid: signed differences kernel - oracle of the terms of log alpha (inverse log-det, forward log-det, target at the proposal) on the
oracle's step sequences, for a tile of 16 chains of the benchmarked state (global chain ids: the draws of the full run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import cref, flow, mala, prng, targets
from tests import gpu_util as gu
from tests import test_gpu_replay as tr
from mfm_amd import _lib
tp = gu.train_phi4_like_bench()
B, d = 4096, 256
dist, model, args = tp["dist"], tp["model"], tp["args32"]
params = gu.unflat_params(model, tp["params_flat"])
x32 = tp["pos"]
key = prng.PRNGKey(4242)
cr = cref.CRef(model, params)
vg = targets.Tempered(dist, 1.0).value_and_grad
st0 = mala.init(x32.astype(np.float64), vg)
so = {}
cr.rwmh_step(prng.split(key, B), st0, args, stats=so)
la = so["log_alpha"]
sh = np.flatnonzero(la > -60)
print("shallow chains", sh, np.round(la[sh], 1))
for b in list(sh[:2]) + [100]:
    b0 = (int(b) // 16) * 16
    ctx = gu.make_ctx(dist, args, n_local=16, n_total=B, offset=b0, fourier=model.f, params=params)
    xs = x32[b0:b0 + 16]
    pos = torch.as_tensor(xs).cuda(); logp = torch.empty(16, dtype=torch.float64, device="cuda"); grad = torch.empty(16, d, device="cuda")
    ctx.mala_init(pos, 1.0, logp, grad)
    s0 = mala.MALAState(xs.astype(np.float64), logp.cpu().numpy(), grad.cpu().numpy().astype(np.float64))
    keys = prng.split(key, B)[b0:b0 + 16]
    nat = {}
    flow.rwmh_step(keys, s0, vg, model, params, args, nat)
    dt, acc = tr._replay_arrays([nat["inv"], nat["fwd"]])
    rp = dict(inv=dict(dt=dt[0].astype(np.float64), acc=acc[0]), fwd=dict(dt=dt[1].astype(np.float64), acc=acc[1]))
    s = {}
    new_o, info_o = flow.rwmh_step(keys, s0, vg, model, params, args, s, replay=rp)
    d_dt, d_acc = torch.as_tensor(dt).cuda(), torch.as_tensor(acc).cuda()
    ratio = torch.zeros(dt.shape, device="cuda"); own = torch.zeros(dt.shape, device="cuda")
    diag = torch.zeros(16, 4, dtype=torch.float64, device="cuda")
    ctx.debug_replay(d_dt, d_acc, ratio, own, diag)
    a = torch.empty(16, device="cuda"); isacc = torch.empty(16, dtype=torch.uint8, device="cuda"); prop = torch.empty(16, d, device="cuda"); ns = torch.empty(16, dtype=torch.int32, device="cuda")
    ctx.flow_step(_lib.FLOW_RWMH, key, 1.0, pos, logp, grad, a, isacc, prop, ns)
    dg = diag.cpu().numpy()
    lpn_o = vg(info_o.proposed_position)[0]
    np.set_printoptions(linewidth=200, precision=2, suppress=True)
    print(f"tile at {b0}: oracle log alpha (natural, C)", la[b0:b0 + 16])
    print("   replay oracle log alpha   ", s["log_alpha"])
    print("   kernel log alpha (replay) ", dg[:, 3])
    print("   d vol0 ", dg[:, 0] - s["vol0"])
    print("   d volp ", dg[:, 1] - s["volp"])
    print("   d lpn  ", dg[:, 2] - lpn_o)
    print("   d la   ", dg[:, 3] - s["log_alpha"])
    print("   |dx'|  ", np.abs(prop.cpu().numpy() - info_o.proposed_position).max(1))
    ctx.close()
