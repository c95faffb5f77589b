"""Development aid: run-to-run determinism of the phi-four loop (generic solver shape) and of the headline shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.test_gpu_loop import _args
from mfm_amd import distributions as D, exe_flow_matching as E

def run(d, B, hidden, F, iters=8, K=3):
    common = dict(example="phi-four", dim=d, num_chain=B, learning_iter=iters, mcmc_per_flow_steps=float(K), hutchs=True,
                  fourier_dim=F, hidden_x=[hidden] * 2, hidden_t=[hidden] * 2, hidden_xt=[hidden] * 2, seed=1024, eval_iter=1, step_size=1e-4)
    res, res_, ex = E.run(D.PhiFour(d), _args(**common), None, log_every=1000, return_extras=True)
    m = ex["metrics"][:, 0].copy(); ex["engine"].close()
    return m

for shape in ((64, 64, 32, 16), (256, 64, 128, 128)):
    a = [run(*shape) for _ in range(4)]
    print(shape, "loss traces identical across 4 runs:", all(np.array_equal(a[0], x) for x in a[1:]))
    for x in a: print("   ", np.array2string(x[:6], precision=6))
