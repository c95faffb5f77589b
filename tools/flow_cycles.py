"""Development aid: in-situ cycle accounting of the flow-MH step (stamps build): cycles per executed field evaluation,
share of the solver loop outside the field evaluation, shader clock, attempt statistics."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MFM_LIB"] = os.path.join(ROOT, "mfm_amd/lib/" + os.environ.get("MFM_CYC_LIB", "libmfm_hip_stamps.so") + "")
import numpy as np, torch
import bench
from mfm_amd import exe_flow_matching as E, random as jr
from mfm_amd._lib import FLOW_RWMH
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
dbg = torch.zeros(2 * B // 16 * 64, dtype=torch.int64, device="cuda")
fn = ctx.lib.mfm_debug_flow_buffer; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
assert fn(dbg.data_ptr()) == 0
pos = eng.local(dist.init_params); logp = torch.empty(B, device="cuda", dtype=torch.float64); grad = torch.empty_like(pos)
acc = torch.empty(B, device="cuda", dtype=torch.float32); nst = torch.zeros(B, device="cuda", dtype=torch.int32)
ctx.mala_init(pos, 1.0, logp, grad)
ks = k[1]
for count in range(1, 304):
    ks, kg, kt = jr.split(ks, 3)
    if count % 101 == 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.flow_step(FLOW_RWMH, kg, 1.0, pos, logp, grad, acc, None, None, nst); e1.record()
        torch.cuda.synchronize()
        n = nst.cpu().numpy().astype(float); t = n.reshape(-1, 16)
        dall = dbg.cpu().numpy().reshape(2, -1, 64).astype(float)
        d = dall[0]
        cyc, rt, nev, cev = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
        ms = e0.elapsed_time(e1)
        print(f"flow step @ {count}: {ms:.1f} ms | natt mean {n.mean():.1f} p50 {np.median(n):.0f} p99 {np.percentile(n,99):.0f} max {n.max():.0f} | tile-max mean {t.max(1).mean():.1f} max {t.max(1).max():.0f}")
        print(f"   per WG: evals mean {nev.mean():.0f} max {nev.max():.0f} | cycles mean {cyc.mean()/1e6:.1f}M max {cyc.max()/1e6:.1f}M | clock {np.median(cyc/rt)*0.1:.3f} GHz "
              f"| cycles/eval in eval {np.median(cev/nev):.0f}, total/eval {np.median(cyc/nev):.0f} -> outside-eval share {1-np.median(cev/cyc):.3f}")
        if d[:, 5].max() > 0:
            ntb, ctb = d[:, 5], d[:, 6]
            print(f"   time batches per WG mean {ntb.mean():.0f}; cycles/batch {np.median(ctb/ntb):.0f}; share of WG cycles: eval {np.median(cev/cyc):.3f} batch {np.median(ctb/cyc):.3f} other {1-np.median((cev+ctb)/cyc):.3f}")
        if d[:, 16:28].max() > 0:
            print("   eval sections, cycles per eval: jobs x1,x2,j1,j2,out:", (np.median(d[:, 16:21] / d[:, 2:3], axis=0)).astype(int), " epilogue before barrier 1..4:", (np.median(d[:, 21:25] / d[:, 2:3], axis=0)).astype(int), " out epilogue:", int(np.median(d[:, 25] / d[:, 2])), " 4 barriers:", int(np.median(d[:, 26] / d[:, 2])))
        if d[:, 8:16].max() > 0:
            print("   time-batch sections, cycles per batch (sincos, bar, cos job, bar+sin write+bar, sin job+epi, bar+t2+epi, bar+gate/j1t, bar):", (np.median(d[:, 8:16] / d[:, 5:6], axis=0)).astype(int))
        if d[:, 16:28].max() > 0:
            print("   eval sections, cycles per eval: jobs x1,x2,j1,j2,out:", (np.median(d[:, 16:21] / d[:, 2:3], axis=0)).astype(int), " epilogue before barrier 1..4:", (np.median(d[:, 21:25] / d[:, 2:3], axis=0)).astype(int), " out epilogue:", int(np.median(d[:, 25] / d[:, 2])), " 4 barriers:", int(np.median(d[:, 26] / d[:, 2])))
        d4 = dall[1]
        if d4[:, 16:28].max() > 0:
            print("   WAVE 4 eval sections: jobs x1,x2,j1,j2,out:", (np.median(d4[:, 16:21] / d4[:, 2:3], axis=0)).astype(int), " epilogue before barrier 1..4:", (np.median(d4[:, 21:25] / d4[:, 2:3], axis=0)).astype(int), " out epilogue:", int(np.median(d4[:, 25] / d4[:, 2])), " 4 barriers:", int(np.median(d4[:, 26] / d4[:, 2])), "eval total", int(np.median(d4[:, 3] / d4[:, 2])))
            print("   WAVE 4 time-batch sections:", (np.median(d4[:, 8:16] / d4[:, 5:6], axis=0)).astype(int))
        if d[:, 28].max() > 0:
            nct, cct = d[:, 28], d[:, 29]
            f = nct > 0
            print(f"   compacted time batches (<= 8 rows): per WG mean {nct.mean():.0f}, slowest WG {nct[np.argmax(cyc)]:.0f}; cycles/batch {np.median(cct[f] / nct[f]):.0f}")
            nec, cec = d[:, 30], d[:, 31]
            f = nec > 0
            if f.any():
                print(f"   compact evaluations (<= 8 rows): per WG mean {nec.mean():.0f}, slowest WG {nec[np.argmax(cyc)]:.0f}; cycles/evaluation {np.median(cec[f] / nec[f]):.0f}")
        if d[:, 48].max() > 0:
            f = d[:, 48] > 0; f1 = d[:, 50] > 0
            print(f"   micro evaluations (<= 2 rows): per WG mean {d[:, 48].mean():.0f}, slowest WG {d[np.argmax(cyc), 48]:.0f}; cycles/evaluation {np.median(d[f, 49] / d[f, 48]):.0f} | "
                  f"single-tile time batches (<= 3 rows): per WG mean {d[:, 50].mean():.0f}, slowest WG {d[np.argmax(cyc), 50]:.0f}; cycles/batch {np.median(d[f1, 51] / d[f1, 50]):.0f}")
        for wname, dd in (("wave 0", d), ("wave 4", d4)):
            f = dd[:, 30] > 0
            if f.any() and dd[:, 32:48].max() > 0:
                c = np.median(dd[f, 32:48] / dd[f, 30:31], axis=0).astype(int)
                print(f"   COMPACT eval sections, {wname}, cycles per evaluation: gathers {c[0]} target(pre) {c[1]} x1 job {c[2]} target(post)+epi {c[3]} barriers(4) {c[4]} "
                      f"x2 job {c[5]} epilogues x2/j1/j2 {c[6]} j1 job {c[7]} j2 job {c[8]} out job+epi {c[9]} partials+readback {c[10]}")
        done = d[:, 53]
        tl = d[:, 54:60].copy()
        if "tl_prev" in globals(): tl_now = tl - tl_prev
        else: tl_now = tl
        tl_prev = tl
        for P in (1, 2, 3):
            cc, nn = tl_now[:, 2 * (P - 1)], tl_now[:, 2 * (P - 1) + 1]
            if nn.sum() > 0: print(f"   tail loop P = {P}: attempts per WG mean {nn.mean():.1f}, cycles per attempt {cc.sum() / nn.sum() / 1e3:.1f}k")
        t4 = dall[1][:, 54:58].copy()
        t4_now = t4 - t4_prev if "t4_prev" in globals() else t4
        t4_prev = t4
        if t4_now[:, 0].sum() > 0:
            print(f"   inside the P = 1 tail: cycles per evaluation {t4_now[:, 1].sum() / t4_now[:, 0].sum() / 1e3:.2f}k, per time batch {t4_now[:, 3].sum() / max(t4_now[:, 2].sum(), 1) / 1e3:.2f}k "
                  f"(6 evaluations + 1 batch = {(6 * t4_now[:, 1].sum() / t4_now[:, 0].sum() + t4_now[:, 3].sum() / max(t4_now[:, 2].sum(), 1)) / 1e3:.1f}k of the attempt)")
        print(f"   tile done (before the noise work): mean {done.mean()/1e6:.1f}M max {done.max()/1e6:.1f}M mean/max {done.mean()/done.max():.3f} | time in the tail loops: mean {d[:,52].mean()/1e6:.1f}M")
        natt_main = d[:, 5] + d[:, 28]
        for nm, dd in (("wave 0", d), ("wave 4", d4)):
            print(f"   main loop outside batch / evaluations, {nm}, cycles per attempt: stage inputs (6) {np.median(dd[:,60]/natt_main):.0f}, norms {np.median(dd[:,61]/natt_main):.0f}, leaders + barrier {np.median(dd[:,62]/natt_main):.0f}, decision (+ solve switches) {np.median(dd[:,63]/natt_main):.0f}")
        for w in np.argsort(-done)[:4]:
            r = d[w]
            full = r[6] + r[3]; cmp8 = r[29] + r[31]; tail = r[52]
            ts = np.sort(t[w])[::-1].astype(int)
            natt_tail = ts[0] + 4 - r[5] - r[28]
            print(f"   slow WG {w}: done {r[53]/1e6:.1f}M | full: {r[5]:.0f} att {full/1e6:.1f}M ({full/max(r[5],1)/1e3:.0f}k/att) | compact: {r[28]:.0f} att {cmp8/1e6:.1f}M ({cmp8/max(r[28],1)/1e3:.0f}k/att) | "
                  f"tails: ~{natt_tail:.0f} att {tail/1e6:.1f}M ({tail/max(natt_tail,1)/1e3:.0f}k/att) | rest {(r[53]-full-cmp8-tail)/1e6:.1f}M ({(r[53]-full-cmp8-tail)/max(r[5]+r[28],1)/1e3:.0f}k/att) | tile natt sorted {ts[:9]}")
        print(f"   WG time: mean/max {cyc.mean()/cyc.max():.3f}; alg evals (4+6 natt) mean {4+6*n.mean():.0f}; executed/alg {nev.mean()/(4+6*n.mean()):.3f}; max-WG/alg {nev.max()/(4+6*n.mean()):.3f}")
    else:
        ctx.mala_step(kg, 1.0, args.step_size, pos, logp, grad, acc)
    eng.train_step(kt, pos)
