"""Error behaviour at the C-ABI boundary: unsupported or inconsistent requests raise MfmError carrying mfm_last_error()
-- they never fall back to another path ("Semantics that change at the boundary", item 7)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(**over):
    from mfm_amd import _lib
    kw = dict(dim=64, fourier_dim=16, hidden_t=(32, 32), hidden_x=(32, 32), hidden_xt=(32, 32), n_chain_local=32, hutch=1)
    kw.update(over)
    return _lib.Context(**kw)


def test_create_rejects_bad_configurations():
    from mfm_amd import _lib
    with pytest.raises(_lib.MfmError, match="multiple of 16"):
        _ctx(n_chain_local=30)
    _ctx(hidden_x=(30, 32)).close()                                   # served since round 4: zero-padded widths, either family
    _ctx(hidden_x=(30, 32), kernel_family=_lib.FAMILY_TILE).close()
    _ctx(hidden_x=(30, 32), kernel_family=_lib.FAMILY_WIDE).close()
    with pytest.raises(_lib.MfmError, match="wide kernel family only"):
        _ctx(hidden_x=(32, 32, 32), kernel_family=_lib.FAMILY_TILE)
    with pytest.raises(_lib.MfmError, match="hidden widths must be positive"):
        _ctx(hidden_x=(0, 32))
    with pytest.raises(_lib.MfmError, match="hidden layers"):
        _ctx(hidden_t=(32, 32, 32, 32))
    with pytest.raises(_lib.MfmError, match="outside n_chain_total"):
        _ctx(n_chain_total=32, chain_offset=16)
    with pytest.raises(_lib.MfmError, match="does not fit"):
        _ctx(dim=1024, hidden_x=(1024, 1024), hidden_t=(1024, 1024), hidden_xt=(1024, 1024), kernel_family=_lib.FAMILY_TILE)
    _ctx(dim=1024, hidden_x=(1024, 1024), hidden_t=(1024, 1024), hidden_xt=(1024, 1024), hutch=0).close()      # wide family + exact trace: served since round 4
    _ctx(activation=_lib.ACTIVATIONS["gelu"], kernel_family=_lib.FAMILY_TILE).close()      # served since round 3 (stored f'(pre-activation))
    with pytest.raises(_lib.MfmError, match="unknown activation"):
        _ctx(activation=9)
    with pytest.raises(_lib.MfmError, match="kernel_family"):
        _ctx(kernel_family=7)


def test_calls_in_the_wrong_state_or_with_bad_sizes_raise():
    import torch
    from mfm_amd import _lib
    ctx = _ctx()
    pos = torch.zeros(32, 64, device="cuda"); logp = torch.zeros(32, dtype=torch.float64, device="cuda"); grad = torch.zeros(32, 64, device="cuda")
    with pytest.raises(_lib.MfmError, match="mfm_set_target"):
        ctx.mala_init(pos, 1.0, logp, grad)
    with pytest.raises(_lib.MfmError, match="phi4 target takes"):
        ctx.set_target(_lib.PHI4, [0.1])
    with pytest.raises(_lib.MfmError, match="unknown target"):
        ctx.set_target(9, [0.1, 20.0])
    ctx.set_target(_lib.PHI4, [0.1, 20.0])
    with pytest.raises(_lib.MfmError, match="step_size"):
        ctx.mala_step((0, 1), 1.0, 0.0, pos, logp, grad)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda"); g = torch.zeros(ctx.n_params, device="cuda")
    with pytest.raises(_lib.MfmError, match="mfm_set_fourier"):
        ctx.fm_loss_grad((0, 1), pos, loss, g)
    ctx.set_fourier(np.ones(16, dtype=np.float32))
    ctx.fm_loss((0, 1), pos[:24], loss)                       # any n > 0 (a last partial tile is staged inside the library) ...
    assert np.isfinite(loss.item())
    with pytest.raises(_lib.MfmError, match="must be positive"):
        ctx.fm_loss((0, 1), pos[:0], loss)                    # ... but not none
    with pytest.raises(_lib.MfmError, match="max_eval_samples"):
        ctx.fm_loss((0, 1), torch.zeros(64, 64, device="cuda"), loss)
    out = torch.zeros(32, 64, device="cuda"); ldj = torch.zeros(32, device="cuda")
    with pytest.raises(_lib.MfmError, match="direction"):
        ctx.ode_transform(0, pos, out, ldj, key=(0, 1))
    with pytest.raises(_lib.MfmError, match="unknown flow step mode"):
        ctx.flow_step(5, (0, 1), 1.0, pos, logp, grad)
    with pytest.raises(_lib.MfmError, match="contiguous CUDA tensor"):
        ctx.mala_init(pos.cpu(), 1.0, logp, grad)
    with pytest.raises(_lib.MfmError, match="float64"):
        ctx.mala_init(pos, 1.0, logp.float(), grad)
    assert ctx.noise_prefetch(np.zeros((4, 2), np.uint32), np.zeros((4, 2), np.uint32)) is False      # not the headline shape: declined
    ctx.close()


def test_wide_family_takes_the_mixtures_and_declines_wide_ones():
    import torch
    from mfm_amd import _lib
    ctx = _ctx(kernel_family=_lib.FAMILY_WIDE, dim=2)
    ctx.set_target(_lib.GMM, np.concatenate([[1], np.zeros(2), np.ones(2), [1.0]]))      # served since round 4 (tests/test_gpu_depth.py)
    ctx.close()
    ctx = _ctx(kernel_family=_lib.FAMILY_WIDE, dim=16)
    with pytest.raises(_lib.MfmError, match="dim <= 8"):
        ctx.set_target(_lib.GMM, np.concatenate([[1], np.zeros(16), np.ones(16), [1.0]]))
    ctx.close()
