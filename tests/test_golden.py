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
