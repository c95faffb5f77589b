"""Exact-trace transform of N reference draws through the d = 2 flow (final sampling, exe_flow_matching.py:453-455): the three
tilings timed on the same inputs (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import prng
from tests.test_gpu_d2tile import _setup, _dev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 409600
gu, args, dist, model, params = _setup("gmm16", 64)
rng = np.random.default_rng(0)
u = _dev(rng.standard_normal((N, 2)).astype(np.float32))
for tile in ("16", "4s", "4"):
    os.environ["MFM_D2_TILE"] = tile
    ctx = gu.make_ctx(dist, args, n_local=64, n_total=64, fourier=model.f, params=params, max_eval=N)
    out = torch.empty(N, 2, device="cuda"); ldj = torch.empty(N, device="cuda"); ns = torch.empty(N, dtype=torch.int32, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.ode_transform(1, u, out, ldj, key=prng.PRNGKey(1), nsteps=ns)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"N={N} tile {tile}: {1e3 * (t1 - t0):.1f} ms, attempts mean {ns.float().mean().item():.1f} max {ns.max().item()}")
    ctx.close()
