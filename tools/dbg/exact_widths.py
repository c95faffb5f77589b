"""Development aid: the wide family's exact-trace transform against the oracle with a gate term that is NOT tamed away (phi-four: the
Hessian diagonal of the target enters the trace through gate_i H_ii)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_gpu_wide as tw, test_gpu_replay as tr
from tests import gpu_util as gu


def tamed(model, out_scale=0.5, seed=9, gate=float(os.environ.get("GATE", 0.05))):
    p = gu.rand_params(model, seed=seed, out_scale=out_scale)
    p[4]["kernel"] *= gate; p[4]["bias"] *= gate
    return p


tr._tamed = tamed
for hid, F in ((128, 128), (64, 16)):
    try:
        tw.test_wide_exact_trace_transform_on_prescribed_steps("phi4", 64, hid, F, 1)
        print("OK", hid, F)
    except AssertionError as e:
        print("FAIL", hid, F, str(e)[:300])
