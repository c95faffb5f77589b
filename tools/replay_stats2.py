"""(1) how many attempted steps a gate-tamed random network takes as its output scale grows (a well-conditioned field with a
long integration for the replay parity test); (2) the float64 oracle's OWN sensitivity, on a fixed step sequence, to a one-ulp
(float32) perturbation of the initial positions in the benchmark regime -- the yardstick for float32-vs-float64 differences of
the clipped field (the log-det integrand has spikes of height |H z| ~ 2e3 wherever |grad log pi| crosses the clip)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import flow, mala, ode, prng, targets  # noqa: E402
from tests import gpu_util as gu  # noqa: E402
from tests.test_gpu_replay import _replay_arrays  # noqa: E402
from tools.replay_stats import q, transform_case  # noqa: E402

if __name__ == "__main__":
    for osc in (1.0, 2.0, 4.0):
        transform_case(256, 128, 128, 1, 1e-3, out_scale=osc)
    transform_case(64, 32, 16, 1, 1e-3, out_scale=4.0)
    # (2) oracle sensitivity, trained regime
    tp = gu.train_phi4_like_bench()
    model, dist, args = tp["model"], tp["dist"], tp["args32"]
    params = gu.unflat_params(model, tp["params_flat"])
    B = 32
    x = tp["pos"][:B].astype(np.float64)
    vg = targets.Tempered(dist, 1.0).value_and_grad
    keys = prng.split(prng.PRNGKey(77), B)
    st0 = mala.init(x, vg)
    nat = {}
    flow.rwmh_step(keys, st0, vg, model, params, args, nat)
    dt, acc = _replay_arrays([nat["inv"], nat["fwd"]])
    rp = dict(inv=dict(dt=dt[0].astype(np.float64), acc=acc[0]), fwd=dict(dt=dt[1].astype(np.float64), acc=acc[1]))
    so = {}
    flow.rwmh_step(keys, st0, vg, model, params, args, so, replay=rp)
    rng = np.random.default_rng(0)
    xp = x * (1.0 + 6e-8 * rng.choice([-1.0, 1.0], size=x.shape))
    sp = {}
    flow.rwmh_step(keys, mala.init(xp, vg), vg, model, params, args, sp, replay=rp)
    print("oracle vs oracle(x * (1 +- 6e-8)), same step sequence, trained d=256:")
    print("  |dup|  :", q(np.abs(sp["up"] - so["up"]).max(1)))
    print("  |dvol0|:", q(np.abs(sp["vol0"] - so["vol0"])))
    print("  |dvolp|:", q(np.abs(sp["volp"] - so["volp"])))
    print("  |dla|  :", q(np.abs(sp["log_alpha"] - so["log_alpha"])))
