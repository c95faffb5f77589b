"""Golden vectors (tests/golden/mfm_golden.npz, made by tools/make_golden.py from the CPU oracle): the CPU suite
re-derives them (regression pin of the oracle), the GPU suite checks the HIP kernels against them through the C ABI."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "mfm_golden.npz"))


def test_oracle_reproduces_golden_vectors():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    now = make_golden.build()
    assert set(now) == set(G.files)
    for k in G.files:
        a, b = np.asarray(now[k], dtype=np.float64), np.asarray(G[k], dtype=np.float64)
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9, err_msg=k)


def _ctx():
    from tests import gpu_util as gu
    args, dist, kk, model, state = gu.phi4_setup(d=64, B=32, hidden=32, F=16, learning_iter=20)
    params = gu.unflat_params(model, G["cfg_params"])
    return args, dist, model, gu.make_ctx(dist, args, fourier=G["cfg_fourier"], params=params)


@pytest.mark.gpu
def test_hip_kernels_match_golden_vectors():
    import torch
    from mfm_amd import _lib
    args, dist, model, ctx = _ctx()
    dev = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).cuda()
    x0 = dev(G["cfg_x0"], torch.float32)
    B, d = 32, 64
    # MALA
    pos = x0.clone(); logp = torch.empty(B, dtype=torch.float64, device="cuda"); grad = torch.empty(B, d, device="cuda", dtype=torch.float32)
    ctx.mala_init(pos, float(G["mala_beta"]), logp, grad)
    np.testing.assert_allclose(logp.cpu().numpy(), G["mala_logp0"], rtol=2e-6, atol=1e-3)
    np.testing.assert_allclose(grad.cpu().numpy(), G["mala_grad0"], rtol=2e-5, atol=2e-3)
    acc = torch.empty(B, device="cuda", dtype=torch.float32); isacc = torch.empty(B, dtype=torch.uint8, device="cuda"); prop = torch.empty_like(pos)
    ctx.mala_step(G["mala_key"], float(G["mala_beta"]), 1e-4, pos, logp, grad, acc, isacc, prop, None)
    np.testing.assert_allclose(prop.cpu().numpy(), G["mala_prop"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(acc.cpu().numpy(), G["mala_acc"], atol=5e-3)
    sure = np.abs(G["mala_u"] - G["mala_acc"]) > 1e-2
    np.testing.assert_array_equal(isacc.cpu().numpy().astype(bool)[sure], G["mala_isacc"][sure])
    # vector field + JVP
    v = torch.empty(B, d, device="cuda", dtype=torch.float32); jv = torch.empty_like(v)
    ctx.vf_apply(x0, dev(G["vf_t"]), v, dev(G["vf_z"], torch.float32), jv)
    assert np.abs(v.cpu().numpy() - G["vf_v"]).max() < 2e-5 * np.abs(G["vf_v"]).max()
    assert np.abs(jv.cpu().numpy() - G["vf_jvp"]).max() < 2e-5 * np.abs(G["vf_jvp"]).max()
    # flow-matching loss / gradient, AdamW
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda", dtype=torch.float32)
    ctx.fm_loss_grad(G["fm_key"], x0, loss, grads)
    assert abs(loss.item() - float(G["fm_loss"])) < 2e-5 * float(G["fm_loss"])
    assert np.abs(grads.cpu().numpy() - G["fm_grads"]).max() < 2e-4 * np.abs(G["fm_grads"]).max()
    ctx.adamw_step(grads)
    np.testing.assert_allclose(ctx.get_params(), G["adam_params"], rtol=1e-4, atol=2e-6)
    ctx.set_params(G["cfg_params"])
    # Dopri5 transforms
    keys = dev(G["ode_keys"].astype(np.uint32).view(np.int32))
    for direction, ky, kl, kn in ((1, "ode_fwd", "ode_fwd_ldj", "ode_fwd_natt"), (-1, "ode_inv", "ode_inv_ldj", "ode_inv_natt")):
        out = torch.empty(B, d, device="cuda", dtype=torch.float32); ldj = torch.empty(B, device="cuda", dtype=torch.float32)
        ns = torch.empty(B, dtype=torch.int32, device="cuda")
        ctx.ode_transform(direction, x0, out, ldj, keys=keys, nsteps=ns)
        assert np.abs(out.cpu().numpy() - G[ky]).max() < 2e-3
        assert np.abs(ldj.cpu().numpy() - G[kl]).max() < 5e-2 * max(1.0, np.abs(G[kl]).max())
        assert abs(ns.float().mean().item() - G[kn].mean()) < 0.1 * G[kn].mean()
    # flow-MH step
    pos = x0.clone(); ctx.mala_init(pos, 1.0, logp, grad)
    ctx.flow_step(_lib.FLOW_RWMH, G["flow_key"], 1.0, pos, logp, grad, acc, isacc, prop, None)
    assert np.abs(prop.cpu().numpy() - G["flow_prop"]).max() < 5e-3
    sure = ~np.isfinite(G["flow_logacc"]) | (np.abs(G["flow_logacc"]) > 1)
    np.testing.assert_array_equal(isacc.cpu().numpy().astype(bool)[sure], G["flow_isacc"][sure])
    # beta bisection
    assert abs(ctx.beta_update(0.0, dev(G["beta_ll"]), 0.95) - float(G["beta_0"])) < 1e-9
    ctx.close()


# ---- second set: SMC pieces, non-default activations, widegauss reference distribution ------------------------------------
GX = np.load(os.path.join(ROOT, "tests", "golden", "mfm_golden_ext.npz"))


def test_oracle_reproduces_extended_golden_vectors():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    now = make_golden.build_ext()
    assert set(now) == set(GX.files)
    for k in GX.files:
        np.testing.assert_allclose(np.asarray(now[k], dtype=np.float64), np.asarray(GX[k], dtype=np.float64), rtol=1e-9, atol=1e-9, err_msg=k)


@pytest.mark.gpu
def test_hip_kernels_match_extended_golden_vectors():
    import torch
    from mfm_amd import _lib
    from tests import gpu_util as gu
    dev = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).cuda()
    # SMC pieces
    args, dist, kk, model, state = gu.phi4_setup(d=64, B=256, hidden=32, F=16)
    ctx = gu.make_ctx(dist, args)
    ll = dev(GX["smc_ll"])
    assert abs(ctx.smc_delta(ll, 0.95, 1.0) - float(GX["smc_delta"])) < 1e-9
    w = torch.empty(256, dtype=torch.float64, device="cuda")
    lognorm = ctx.smc_weights(ll, float(GX["smc_delta"]), w)
    np.testing.assert_allclose(w.cpu().numpy(), GX["smc_weights"], rtol=1e-12)
    assert abs(lognorm - float(GX["smc_lognorm"])) < 1e-10
    idx = torch.empty(256, dtype=torch.int32, device="cuda"); scr = torch.empty(256, dtype=torch.float64, device="cuda")
    ctx.smc_resample(GX["smc_key"], dev(GX["smc_weights"]), scr, idx)
    np.testing.assert_array_equal(idx.cpu().numpy(), GX["smc_idx"])
    ctx.close()
    # activations (tanh: fused family, gelu: wide family)
    for act in ("tanh", "gelu"):
        args, dist, kk, model, state = gu.phi4_setup(d=64, B=32, hidden=32, F=16, non_linearity=act)
        ctx = gu.make_ctx(dist, args, fourier=GX[f"{act}_fourier"], params=gu.unflat_params(model, GX[f"{act}_params"]))
        x0 = dev(GX[f"{act}_x0"], torch.float32)
        v = torch.empty(32, 64, device="cuda"); jv = torch.empty_like(v)
        ctx.vf_apply(x0, dev(GX[f"{act}_t"]), v, dev(GX[f"{act}_z"]), jv)
        assert np.abs(v.cpu().numpy() - GX[f"{act}_v"]).max() < 3e-5 * np.abs(GX[f"{act}_v"]).max()
        assert np.abs(jv.cpu().numpy() - GX[f"{act}_jvp"]).max() < 3e-5 * np.abs(GX[f"{act}_jvp"]).max()
        loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
        ctx.fm_loss_grad(GX[f"{act}_key"], x0, loss, grads)
        assert abs(loss.item() - float(GX[f"{act}_loss"])) < 2e-5 * float(GX[f"{act}_loss"])
        assert np.abs(grads.cpu().numpy() - GX[f"{act}_grads"]).max() < 3e-4 * np.abs(GX[f"{act}_grads"]).max()
        ctx.close()
    # widegauss reference distribution
    args, dist, kk, model, state = gu.phi4_setup(d=64, B=32, hidden=32, F=16, ref_dist="widegauss")
    ctx = gu.make_ctx(dist, args, fourier=GX["wg_fourier"], params=gu.unflat_params(model, GX["wg_params"]))
    x0 = dev(GX["wg_x0"], torch.float32)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); grads = torch.zeros(ctx.n_params, device="cuda")
    ctx.fm_loss_grad(GX["wg_key"], x0, loss, grads)
    assert abs(loss.item() - float(GX["wg_loss"])) < 2e-5 * float(GX["wg_loss"])
    pos = x0.clone(); logp = torch.empty(32, dtype=torch.float64, device="cuda"); grad = torch.empty(32, 64, device="cuda")
    ctx.mala_init(pos, 0.8, logp, grad)
    acc = torch.empty(32, device="cuda"); isacc = torch.empty(32, dtype=torch.uint8, device="cuda"); prop = torch.empty(32, 64, device="cuda")
    ctx.flow_step(_lib.FLOW_IMH, GX["wg_imh_key"], 0.8, pos, logp, grad, acc, isacc, prop, None)
    assert np.abs(prop.cpu().numpy() - GX["wg_imh_prop"]).max() < 5e-3 * max(1.0, np.abs(GX["wg_imh_prop"]).max())
    with np.errstate(divide="ignore"):
        la = np.log(acc.cpu().numpy().astype(np.float64))
    fin = np.isfinite(la) & np.isfinite(GX["wg_imh_logacc"])
    if fin.any():
        assert np.abs(la[fin] - GX["wg_imh_logacc"][fin]).max() < 0.5 + 2e-4 * np.abs(GX["wg_imh_logacc"][fin]).max()
    ctx.close()


def test_end_to_end_fixtures_are_complete_and_self_consistent():
    """tests/golden/e2e_*.npz (tools/make_e2e_golden.py): three oracle seeds per case with every trace the GPU tests compare;
    learning-rate traces equal the schedule (exe_flow_matching.py:189-198), temperatures are non-decreasing and end at <= 1,
    the flow iterations carry attempt counts, and the seeds really differ."""
    import os
    from oracle import optim
    from tools.make_e2e_golden import CASES
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for case, cfg in CASES.items():
        z = np.load(os.path.join(gold, f"e2e_{case}.npz"))
        assert list(z["seeds"]) == [1, 2, 3]
        n_it, K = cfg["learning_iter"], int(cfg["mcmc_per_flow_steps"])
        lr = optim.learning_rate_fn(n_it, 0, cfg.get("learning_rate", 1e-3))
        for s in (1, 2, 3):
            for key in ("loss", "learning_rate", "beta", "acc_mean", "acc_std"):
                assert z[f"s{s}_{key}"].shape == (n_it,), (case, s, key)
            np.testing.assert_allclose(z[f"s{s}_learning_rate"], [lr(i) for i in range(n_it)], rtol=1e-12)
            b = z[f"s{s}_beta"]
            assert (np.diff(b) >= 0).all() and 0 < b[0] and b[-1] <= 1.0
            assert z[f"s{s}_n_att"].shape == (n_it // (K + 1), 2) and (z[f"s{s}_n_att"] > 1).all()
            assert np.isfinite(z[f"s{s}_loss"]).all() and np.isfinite(z[f"s{s}_chain_mean"]).all()
            if cfg["example"] in ("4-mode", "gaussian-mixture"):
                assert z[f"s{s}_target_loss"].shape == (n_it,)
        assert abs(z["s1_loss"][0] - z["s2_loss"][0]) > 1e-6 * abs(z["s1_loss"][0])
