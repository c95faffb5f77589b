"""Development aid: shape-specialised solver (ode_fast.hip) vs the generic solver tile on the same inputs."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import prng
from tests import gpu_util as gu

def run(tag):
    B, d = 32, 256
    args, dist, k, model, state = gu.phi4_setup(d=d, B=B)
    params = gu.rand_params(model, seed=9, out_scale=0.5)
    params[4]["kernel"] *= 1e-3; params[4]["bias"] *= 1e-3
    ctx = gu.make_ctx(dist, args, fourier=model.f, params=params)
    x32 = dist.init_params.astype(np.float32)
    keys = prng.split(prng.PRNGKey(21), B)
    out = torch.empty(B, d, device="cuda"); ldj = torch.empty(B, device="cuda"); ns = torch.empty(B, dtype=torch.int32, device="cuda")
    x = torch.from_numpy(x32).cuda(); kk = torch.from_numpy(keys.astype(np.uint32).view(np.int32)).cuda()
    res = {}
    for direction in (1, -1):
        ctx.ode_transform(direction, x, out, ldj, keys=kk, nsteps=ns)
        res[direction] = (out.cpu().numpy().copy(), ldj.cpu().numpy().copy(), ns.cpu().numpy().copy())
    t = torch.rand(B, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)); z = torch.randn(B, d, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    np.savez(f"/tmp/dbg_{tag}.npz", y1=res[1][0], l1=res[1][1], n1=res[1][2], y2=res[-1][0], l2=res[-1][1], n2=res[-1][2])

if len(sys.argv) > 1:
    run(sys.argv[1])
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "fast"], env=env)
    env["MFM_GENERIC_ODE"] = "1"
    subprocess.check_call([sys.executable, __file__, "gen"], env=env)
    a, b = np.load("/tmp/dbg_fast.npz"), np.load("/tmp/dbg_gen.npz")
    np.set_printoptions(linewidth=200, suppress=True)
    for s in ("1", "2"):
        print("solve", s, "natt fast", a["n" + s]); print("        natt gen ", b["n" + s])
        print("   |dy| per row max", np.abs(a["y" + s] - b["y" + s]).max(1).round(6))
        print("   |dl|", np.abs(a["l" + s] - b["l" + s]).round(5))
