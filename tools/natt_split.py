"""How much of the flow step's tile time is the lock-step BETWEEN the two solves?  Per chain: attempted steps of the inverse
solve (from the current position) and of the forward solve (from the latent proposal), after one training cycle of the
bench workload.  Tile time today ~ max_tile(inv) + max_tile(fwd); with per-row solve phases ~ max_tile(inv + fwd)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mfm_amd import exe_flow_matching as E, random as jr
from mfm_amd.distributions import PhiFour
from mfm_amd.engine import Engine
B = 4096
args = bench.make_args(B, 10000)
dist = PhiFour(256)
k = jr.split(jr.PRNGKey(1), 6)
dist.initialize_model(k[3], B)
fourier = jr.normal(k[4], (128,))
eng = Engine(dist, args, fourier)
model = E.VectorFieldNet(fourier, dist.grad_logprob, args.hidden_x, args.hidden_t, args.hidden_xt).attach(eng)
eng.ctx.set_params(E.flatten_params(model.init(k[2])))
ctx = eng.ctx
pos = eng.local(dist.init_params); logp = torch.empty(B, device="cuda", dtype=torch.float64); grad = torch.empty_like(pos)
acc = torch.empty(B, device="cuda"); ctx.mala_init(pos, 1.0, logp, grad)
ks = k[1]
for count in range(1, 304):
    ks, kg, kt = jr.split(ks, 3)
    if count % 101 == 0:
        keys = jr.split(kg, B); kk = jr.split_rows(keys, 4)
        kd = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32), device="cuda")
        u0 = torch.empty_like(pos); v0 = torch.empty(B, device="cuda"); n_inv = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.ode_transform(-1, pos, u0, v0, keys=kd(kk[:, 3]), nsteps=n_inv)
        z = torch.empty_like(pos); ctx.normal_rows(kd(kk[:, 0]), z)
        up = u0 + (2.38 / 16.0) * z
        xp = torch.empty_like(pos); vp = torch.empty(B, device="cuda"); n_fwd = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.ode_transform(1, up, xp, vp, keys=kd(kk[:, 2]), nsteps=n_fwd)
        a, b = n_inv.cpu().numpy().astype(float), n_fwd.cpu().numpy().astype(float)
        ta, tb, ts = a.reshape(-1, 16), b.reshape(-1, 16), (a + b).reshape(-1, 16)
        now = ta.max(1) + tb.max(1); alt = ts.max(1)
        print(f"count {count}: mean inv {a.mean():.1f} fwd {b.mean():.1f} corr {np.corrcoef(a, b)[0, 1]:.3f} | tile time today: mean {now.mean():.1f} max {now.max():.1f}"
              f" | per-row phases: mean {alt.mean():.1f} max {alt.max():.1f} | chain max {(a + b).max():.0f} | gain on max {now.max() / alt.max():.3f}, on mean {now.mean() / alt.mean():.3f}")
        nst = torch.empty(B, dtype=torch.int32, device="cuda")
        from mfm_amd._lib import FLOW_RWMH
        ctx.flow_step(FLOW_RWMH, kg, 1.0, pos, logp, grad, acc, None, None, nst)
    else:
        ctx.mala_step(kg, 1.0, args.step_size, pos, logp, grad, acc)
    eng.train_step(kt, pos)
