"""Pins the oracle's VectorFieldNet forward / JVP / parameter gradient and FM loss against torch."""
import numpy as np
import pytest
import torch

from oracle import fm, prng, targets
from oracle.vfield import VectorFieldNet



@pytest.fixture(autouse=True)
def _f64_default():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)


def _rand_params(model, seed=0, scale=0.3):
    rng = np.random.default_rng(seed)
    return [{"kernel": (rng.standard_normal(s) * scale / np.sqrt(s[0])).astype(np.float32),
             "bias": (rng.standard_normal(s[1]) * 0.1).astype(np.float32)} for s in model.layer_shapes()]


def _torch_forward(model, params, x, t, act, glp):
    W = [torch.tensor(p["kernel"].astype(np.float64), requires_grad=True) for p in params]
    b = [torch.tensor(p["bias"].astype(np.float64), requires_grad=True) for p in params]
    f = torch.tensor(model.f)
    degt = 2 * np.pi * f[None] * t[:, None]
    s = torch.cat([torch.cos(degt), torch.sin(degt)], 1)
    li = 0
    for _ in model.hidden_t:
        s = act(s @ W[li] + b[li]); li += 1
    st = s
    s = x
    for _ in model.hidden_x:
        s = act(s @ W[li] + b[li]); li += 1
    sx = s
    nn_t = st @ W[li] + b[li]; li += 1
    s = torch.cat([sx, st], 1)
    for _ in model.hidden_xt:
        s = act(s @ W[li] + b[li]); li += 1
    nn_xt = s @ W[li] + b[li]
    g = glp(x)
    if model.grad_clip:
        g = torch.clamp(g, -model.grad_clip, model.grad_clip)
    return nn_xt + nn_t * g, W, b


def _phi4_grad_t(d):
    coef, beta = 0.1 * d, 20.0

    def glp(x):
        xp = torch.nn.functional.pad(x, (1, 1))
        lap = 2 * x - xp[:, :-2] - xp[:, 2:]
        return -beta * (coef * lap - x * (1 - x * x) / coef)
    return glp


@pytest.mark.parametrize("act,clip,hid", [("relu", None, [32, 32]), ("relu", 1.0, [32, 48]), ("tanh", None, [16, 16]),
                                          ("swish", 1.0, [16]), ("elu", None, [16, 16]), ("gelu", None, [16, 16])])
def test_forward_jvp_backward(act, clip, hid):
    d, B, F = 12, 6, 8
    dist = targets.PhiFour(d)
    fr = np.random.default_rng(1).standard_normal(F)
    model = VectorFieldNet(fr, dist, hid, hid, hid, act, clip)
    params = _rand_params(model)
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (B, d)); t = rng.uniform(0, 1, B); z = rng.standard_normal((B, d))
    tact = {"relu": torch.relu, "tanh": torch.tanh, "swish": torch.nn.functional.silu,
            "elu": torch.nn.functional.elu, "gelu": lambda u: torch.nn.functional.gelu(u, approximate="tanh")}[act]
    xt = torch.tensor(x, requires_grad=True)
    vt, W, b = _torch_forward(model, params, xt, torch.tensor(t), tact, _phi4_grad_t(d))
    v, jv, cache = model.forward(params, x, t, cache=True, tangent=z)
    np.testing.assert_allclose(v, vt.detach().numpy(), rtol=1e-10, atol=1e-10)
    # JVP via double-backward trick: J z = d/ds [ (v . s) grad wrt x ]^T ... use autograd.functional
    _, jvt = torch.autograd.functional.jvp(
        lambda xx: _torch_forward(model, params, xx, torch.tensor(t), tact, _phi4_grad_t(d))[0], (torch.tensor(x),), (torch.tensor(z),))
    np.testing.assert_allclose(jv, jvt.numpy(), rtol=1e-8, atol=1e-8)
    # trace of the Jacobian
    tr = model.jacobian_trace(params, x, t)
    J = torch.autograd.functional.jacobian(
        lambda xx: _torch_forward(model, params, xx[None], torch.tensor(t[:1]), tact, _phi4_grad_t(d))[0][0], torch.tensor(x[0]))
    np.testing.assert_allclose(tr[0], torch.trace(J).item(), rtol=1e-8, atol=1e-8)
    # ... whose blocked all-columns-at-once form equals the one-forward-per-basis-vector reading of jnp.trace(jax.jacfwd(v)(x))
    np.testing.assert_allclose(model.jacobian_trace(params, x, t, block=3), model.jacobian_trace_columns(params, x, t), rtol=1e-12, atol=1e-12)
    # parameter gradient of sum((v - target)^2)
    target = rng.standard_normal((B, d))
    loss = ((vt - torch.tensor(target)) ** 2).sum()
    gs = torch.autograd.grad(loss, W + b)
    grads = model.backward(params, cache, 2 * (v - target))
    for i in range(len(W)):
        np.testing.assert_allclose(grads[i]["kernel"], gs[i].numpy().astype(np.float32), rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(grads[i]["bias"], gs[len(W) + i].numpy().astype(np.float32), rtol=2e-6, atol=1e-6)


def test_zero_init_invariant_and_loss():
    d, B = 10, 16
    dist = targets.PhiFour(d)
    key = prng.PRNGKey(3)
    model = VectorFieldNet(prng.normal(key, (8,)), dist, [16, 16], [16, 16], [16, 16])
    params = model.init(prng.PRNGKey(4))
    shapes = model.layer_shapes()
    assert shapes == [(16, 16), (16, 16), (10, 16), (16, 16), (16, 10), (32, 16), (16, 16), (16, 10)]
    assert all(np.abs(params[i]["kernel"]).max() == 0 for i in model.zero_layers())
    assert abs(params[0]["kernel"].std() - np.sqrt(1 / 16)) < 0.05 and np.abs(params[0]["kernel"]).max() <= 2 / np.sqrt(16) / 0.8796 + 1e-6
    x = dist.initialize_model(prng.PRNGKey(5), B)
    v = model.forward(params, x, np.linspace(0, 1, B))
    assert np.abs(v).max() == 0.0                       # v == 0 at init (exe_flow_matching.py:81,86)
    loss, grads = fm.loss_and_grad(model, params, prng.PRNGKey(6), x, 1e-4)
    t, cond, target = fm.cond_flow_batch(prng.PRNGKey(6), x, 1e-4)
    np.testing.assert_allclose(loss, (target ** 2).sum())
    assert t.shape == (B,) and 0 <= t.min() and t.max() < 1
    # sharded draws equal the full-batch draws
    t2, cond2, target2 = fm.cond_flow_batch(prng.PRNGKey(6), x[4:9], 1e-4, n_total=B, start=4)
    np.testing.assert_array_equal(t[4:9], t2); np.testing.assert_array_equal(cond[4:9], cond2)
    np.testing.assert_array_equal(target[4:9], target2)


def test_jacobian_trace_with_clip_and_cox_target():
    """The exact-trace log-det integrand (exe_flow_matching.py:216-217) where the gate term matters: the clipped phi-four gradient
    (d > 128: the 0/1 derivative of the clip) and the Cox target (dense K^-1 on the Hessian diagonal); blocked form = column form."""
    from tests import gpu_util as gu
    for setup in ("phi4", "lgcp"):
        if setup == "phi4":
            args, dist, k, model, state = gu.phi4_setup(d=144, B=6, hidden=32, F=16, hutch=False)
        else:
            args, dist, k, model, state = gu.lgcp_setup(n=4, B=6, hidden=32, F=16, hutch=False)
        params = gu.rand_params(model, seed=5)
        x = dist.init_params.astype(np.float64)
        t = np.linspace(0.05, 0.95, 6)
        a, b = model.jacobian_trace_columns(params, x, t), model.jacobian_trace(params, x, t, block=11)
        assert np.abs(a).max() > 1e-3
        np.testing.assert_allclose(b, a, rtol=1e-12, atol=1e-12)
