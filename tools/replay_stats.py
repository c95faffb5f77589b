"""Distribution of the float32-vs-float64 differences on a prescribed step sequence (tests/test_gpu_replay.py): which tolerances
the replay parity tests can state.  Run on the GPU box: python tools/replay_stats.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import ode, prng  # noqa: E402
from tests import gpu_util as gu  # noqa: E402
from tests.test_gpu_replay import _dev, _flow_replay_raw, _replay_arrays  # noqa: E402


def q(a):
    a = np.asarray(a, dtype=np.float64).ravel()
    return "min %.1e med %.1e p90 %.1e p99 %.1e max %.1e" % tuple(np.quantile(a, [0, .5, .9, .99, 1]))


def ctl(tag, st_o, ratio_g, own_g, natt):
    rr, dd = [], []
    for b in range(len(natt)):
        n = int(natt[b])
        ro, rg = st_o["ratio_seq"][b, :n], ratio_g[b, :n]
        rr.append(np.abs(rg - ro) / np.maximum(ro, 1e-3))
        do, dg = st_o["dt_own"][b, :n + 1], own_g[b, :n + 1]
        dd.append(np.abs(dg - do) / do)
    rr, dd = np.concatenate(rr), np.concatenate(dd)
    print(f"  {tag}: ratio rel diff {q(rr)} | frac > 1e-3: {(rr > 1e-3).mean():.3f} > 1e-2: {(rr > 1e-2).mean():.3f}")
    print(f"  {tag}: dt_own rel diff {q(dd)} | frac > 1e-3: {(dd > 1e-3).mean():.3f}")


def transform_case(d, hidden, F, direction, gate, out_scale=0.5):
    B = 32
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B, hidden=hidden, F=F)
    params = gu.rand_params(model, seed=9, out_scale=out_scale)
    params[4]["kernel"] *= gate; params[4]["bias"] *= gate
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x64 = dist.init_params.astype(np.float32).astype(np.float64)
    keys = prng.split(prng.PRNGKey(21), B)
    fn = ode.transform_and_logdet if direction > 0 else ode.inverse_and_logdet
    o = (True, args.rtol, args.atol, args.mxstep)
    st = {}
    fn(model, params, keys, x64, *o, stats=st)
    dt, acc = _replay_arrays([st])
    st_o = {}
    y_o, l_o = fn(model, params, keys, x64, *o, stats=st_o, replay=dict(dt=dt[0].astype(np.float64), acc=acc[0]))
    ratio = torch.zeros(dt[0].shape, device="cuda"); own = torch.zeros(dt[0].shape, device="cuda")
    ctx.debug_replay(_dev(dt[0]), _dev(acc[0]), ratio, own)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    ctx.ode_transform(direction, _dev(x64.astype(np.float32)), out, ldj, keys=_dev(keys.astype(np.uint32).view(np.int32)), nsteps=ns)
    y, l, n = out.cpu().numpy(), ldj.cpu().numpy(), ns.cpu().numpy()
    print(f"transform d={d} dir={direction} gate={gate}: attempts oracle {st['n_attempted'].mean():.0f} replay {st_o['n_attempted'].mean():.0f} gpu {n.mean():.0f} equal {(n == st['n_attempted']).all()}")
    print(f"  |dy| per chain max: {q(np.abs(y - y_o).max(1))}  (|y| {np.abs(y_o).max():.2f}, moved {np.abs(y_o - x64).max():.2f})")
    print(f"  |dl|: {q(np.abs(l - l_o))}  (|l| max {np.abs(l_o).max():.1f}) bias {np.mean(l - l_o):.2e}")
    ctl("ctl", st_o, ratio.cpu().numpy(), own.cpu().numpy(), n)
    ctx.close()


def flow_case(label, ctx, model, params, args, dist, beta, x32, key):
    r = _flow_replay_raw(ctx, model, params, args, dist, beta, x32, key)
    so, dg = r["so"], r["diag"]
    print(f"flow {label}: attempts {r['n_o'].mean():.0f} (max {r['n_o'].max()}), gpu equal {(r['n_g'] == r['n_o']).all()}")
    print(f"  |dx'| per chain: {q(np.abs(r['prop'] - r['info_o'].proposed_position).max(1))}")
    print(f"  |dvol0|: {q(np.abs(dg[:, 0] - so['vol0']))} (|vol0| {np.abs(so['vol0']).max():.1f}) bias {np.mean(dg[:, 0] - so['vol0']):.2e}")
    print(f"  |dvolp|: {q(np.abs(dg[:, 1] - so['volp']))} (|volp| {np.abs(so['volp']).max():.1f}) bias {np.mean(dg[:, 1] - so['volp']):.2e}")
    print(f"  |d log alpha|: {q(np.abs(dg[:, 3] - so['log_alpha']))}; log alpha range [{so['log_alpha'].min():.1f}, {so['log_alpha'].max():.1f}]")
    ctl("inv", so["inv"], r["ratio"][0], r["own"][0], so["n_att_inv"])
    ctl("fwd", so["fwd"], r["ratio"][1], r["own"][1], so["n_att_fwd"])


if __name__ == "__main__":
    for d, h, F, gate in [(256, 128, 128, 1.0), (256, 128, 128, 1e-3), (128, 128, 128, 1e-3), (64, 32, 16, 1e-3)]:
        for direction in (1, -1):
            transform_case(d, h, F, direction, gate)
    for d, h, F, gate in [(256, 128, 128, 1.0), (64, 32, 16, 1e-3)]:
        args, dist, k, model, state = gu.phi4_setup(d=d, B=32, hidden=h, F=F)
        params = gu.rand_params(model, seed=9, out_scale=0.3)
        params[4]["kernel"] *= gate; params[4]["bias"] *= gate
        ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
        flow_case(f"random d={d} gate={gate}", ctx, model, params, args, dist, 0.8, dist.init_params.astype(np.float32), prng.PRNGKey(31))
        ctx.close()
    tp = gu.train_phi4_like_bench()
    params = gu.unflat_params(tp["model"], tp["params_flat"])
    ctx = gu.make_ctx(tp["dist"], tp["args32"], n_local=32, n_total=32, fourier=tp["model"].f, params=params)
    flow_case("trained d=256", ctx, tp["model"], params, tp["args32"], tp["dist"], 1.0, tp["pos"][:32], prng.PRNGKey(77))
    print("counters of the training cycle:", tp["counters"], "last flow step attempts mean", tp["n_att_last_flow"].mean())
