"""CPU tests of the host layer: symbolic log-density resolution, parameter (un)flattening, CLI defaults, C-ABI
exports, chain sharding -- no GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_declared_in_header():
    from mfm_amd import _lib, build
    build.build()
    lib = ctypes.CDLL(build.LIB)
    hdr = open(os.path.join(ROOT, "include", "mfm.h")).read()
    declared = set(re.findall(r"\b(mfm_[A-Za-z0-9_]+)\s*\(", hdr))
    declared -= {"mfm_ctx", "mfm_config"}
    assert declared and declared == set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert _lib.load().mfm_version() == 1


def test_build_retries_without_the_scheduler_flag_when_hipcc_rejects_it(tmp_path, monkeypatch, capfd):
    """mfm_amd/build.py: `-mllvm -amdgpu-sched-strategy=max-ilp` is an internal LLVM option; a toolchain that no longer knows it
    must still produce the library (default strategy), and say so."""
    from mfm_amd import build
    (tmp_path / "tiny.hip").write_text('extern "C" int tiny_answer(void) { return 42; }\n')
    monkeypatch.setattr(build, "CSRC", str(tmp_path))
    monkeypatch.setattr(build, "SOURCES", ["tiny.hip"])
    monkeypatch.setattr(build, "LIB", str(tmp_path / "libtiny.so"))
    monkeypatch.setattr(build, "SCHED_FLAGS", ["-mllvm", "-amdgpu-sched-strategy-of-a-future-rocm=max-ilp"])
    assert build.build(force=True) == str(tmp_path / "libtiny.so")
    assert ctypes.CDLL(build.LIB).tiny_answer() == 42
    assert "default scheduling strategy" in capfd.readouterr().err
    monkeypatch.setattr(build, "SOURCES", ["missing.hip"])          # any other failure still raises
    with pytest.raises(Exception):
        build.build(force=True)


def test_ctypes_config_mirrors_the_header_struct_field_for_field():
    """mfm_config crosses the C ABI by value layout: the ctypes mirror must list the header's members in order, with the same types."""
    from mfm_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mfm.h")).read()
    body = hdr[hdr.index("typedef struct mfm_config {"):hdr.index("} mfm_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    ctype = {"int32_t": ctypes.c_int32, "float": ctypes.c_float, "double": ctypes.c_double}
    fields = []
    for ty, names in re.findall(r"\b(int32_t|float|double)\s+([^;]+);", body):
        for nm in names.split(","):
            m = re.fullmatch(r"\s*(\w+)\s*(?:\[(\d+)\])?\s*", nm)
            fields.append((m.group(1), ctype[ty] * int(m.group(2)) if m.group(2) else ctype[ty]))
    mirror = [(n, t) for n, t in _lib.Config._fields_]
    assert [n for n, _ in fields] == [n for n, _ in mirror]
    for (n, t), (_, tm) in zip(fields, mirror):
        assert ctypes.sizeof(t) == ctypes.sizeof(tm) and (t is tm or getattr(t, "_type_", None) is getattr(tm, "_type_", 0)), n
    assert _lib.MAX_DEPTH == int(re.search(r"#define MFM_MAX_DEPTH (\d+)", hdr).group(1))


def test_layer_shapes_follow_the_hidden_lists_of_any_length():
    """exe_flow_matching.py:74-86: Dense layers in creation order for hidden lists of length 1, 2 and 3; host mirror == oracle."""
    from mfm_amd import exe_flow_matching as E
    from oracle import targets
    from oracle.vfield import VectorFieldNet
    assert E.layer_shapes(6, 4, [16, 32], [48, 64], [80, 96]) == [(8, 48), (48, 64), (6, 16), (16, 32), (64, 6), (96, 80), (80, 96), (96, 6)]
    for hx, ht, hxt in (([16], [32], [48]), ([16, 32, 48], [16], [32, 32]), ([16, 16, 16], [32, 32, 32], [48, 48, 48])):
        m = VectorFieldNet(np.zeros(4), targets.PhiFour(6), hx, ht, hxt)
        assert E.layer_shapes(6, 4, hx, ht, hxt) == m.layer_shapes()
        assert len(m.layer_shapes()) == len(hx) + len(ht) + len(hxt) + 2


def test_no_gpu_fails_loudly():
    import torch
    from mfm_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.MfmError, match="no GPU"):
        _lib.Context(dim=4, n_chain_local=16)


def test_pack_index_is_a_bijection_and_matches_the_documented_layout():
    from mfm_amd import _lib
    lib = _lib.load()
    K, N = 48, 32
    idx = np.array([[lib.mfm_pack_index(k, n, K // 16) for n in range(N)] for k in range(K)])
    assert sorted(idx.reshape(-1)) == list(range(K * N))
    k, n = 37, 21                                  # Wp[nt][kb][16 g + c][s] = W[16 kb + 4 g + s][16 nt + c]
    nt, c, kb, g, s = n // 16, n % 16, k // 16, (k % 16) // 4, k % 4
    assert idx[k, n] == (((nt * (K // 16) + kb) * 64) + g * 16 + c) * 4 + s
    idxT = np.array([[lib.mfm_pack_index_T(k, n, N // 16) for n in range(N)] for k in range(K)])
    assert sorted(idxT.reshape(-1)) == list(range(K * N))
    # the transposed pack is the pack of W^T
    assert all(idxT[k, n] == lib.mfm_pack_index(n, k, N // 16) for k in range(0, K, 5) for n in range(0, N, 3))


def test_logdensity_closures_resolve_to_descriptors():
    from mfm_amd.distributions import GaussianMixture, LogGaussianCoxPines, PhiFour, resolve_logdensity
    d = PhiFour(64)
    beta = 0.3
    dist, b = resolve_logdensity(lambda position: beta * d.loglik(position) + d.logprior(position))   # exe_flow_matching.py:301
    assert dist is d and b == beta
    assert resolve_logdensity(d.logprob)[1] == 1.0
    g = GaussianMixture(np.zeros((2, 2)), np.ones((2, 2)), np.ones(2) / 2)
    with pytest.raises(NotImplementedError):
        resolve_logdensity(lambda x: d.loglik(x) + g.loglik(x))
    with pytest.raises(NotImplementedError):
        resolve_logdensity(lambda x: 3.0)
    lg = LogGaussianCoxPines(1024)
    assert lg.counts.sum() == 126 and lg.has_prior
    assert resolve_logdensity(lambda x: 0.5 * lg.loglik(x) + lg.logprior(x))[1] == 0.5
    with pytest.raises(NotImplementedError):
        resolve_logdensity(lambda x: 0.5 * lg.logprob(x))                      # would temper the prior too


def test_distribution_initialisers_match_oracle():
    from mfm_amd import distributions as D, random as jr
    from oracle import prng, targets
    a, b = D.PhiFour(32), targets.PhiFour(32)
    a.initialize_model(jr.PRNGKey(3), 8); b.initialize_model(prng.PRNGKey(3), 8)
    np.testing.assert_array_equal(a.init_params, b.init_params)
    modes, covs, w = 8.0 * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1.0]]), np.ones((4, 2)), np.ones(4) / 4
    a, b = D.GaussianMixture(modes, covs, w), targets.GaussianMixture(modes, covs, w)
    a.initialize_model(jr.PRNGKey(4), 8); b.initialize_model(prng.PRNGKey(4), 8)
    np.testing.assert_array_equal(a.init_params, b.init_params)
    keys = jr.split(jr.PRNGKey(5), 16)
    np.testing.assert_array_equal(a.sample_rows(keys), b.sample_model_rows(keys))
    lg, lo = D.LogGaussianCoxPines(1024), None
    lg.initialize_model(jr.PRNGKey(6), 4)
    ref = targets.LogGaussianCoxPines(1024, np.load(os.path.join(ROOT, "mfm_amd", "data", "pines_counts.npz"))["counts_32"])
    np.testing.assert_allclose(lg.init_params, ref.initialize_model(prng.PRNGKey(6), 4), rtol=1e-12)


def test_param_pytree_roundtrip_and_init():
    from mfm_amd import exe_flow_matching as E, random as jr
    from mfm_amd.distributions import PhiFour
    d = PhiFour(40)
    m = E.VectorFieldNet(np.zeros(10), d.grad_logprob, [32, 48], [16, 32], [64, 16])
    p = m.init(jr.PRNGKey(1))
    assert [p["params"][f"Dense_{i}"]["kernel"].shape for i in range(8)] == [(20, 16), (16, 32), (40, 32), (32, 48), (32, 40), (80, 64), (64, 16), (16, 40)]
    assert np.abs(p["params"]["Dense_4"]["kernel"]).max() == 0 and np.abs(p["params"]["Dense_7"]["kernel"]).max() == 0
    flat = E.flatten_params(p)
    q = E.unflatten_params(flat, m.shapes())
    for i in range(8):
        np.testing.assert_array_equal(p["params"][f"Dense_{i}"]["kernel"], q["params"][f"Dense_{i}"]["kernel"])
    # same initial parameters as the oracle's restatement of the flax initialiser
    from oracle import prng, targets
    from oracle.vfield import VectorFieldNet as OV
    po = OV(np.zeros(10), targets.PhiFour(40), [32, 48], [16, 32], [64, 16]).init(prng.PRNGKey(1))
    np.testing.assert_array_equal(po[3]["kernel"], p["params"]["Dense_3"]["kernel"])


def test_cli_defaults_and_overrides_match_reference():
    from mfm_amd.multi_modal import build_parser
    a = build_parser().parse_args([])
    assert (a.example, a.dim, a.num_chain, a.learning_iter, a.mcmc_per_flow_steps, a.eval_iter) == ("pines", 64, 128, 400, 10, 100)
    assert (a.sigma, a.fourier_dim, a.cond_flow, a.hutchs, a.alpha, a.anneal_iter, a.num_anneal_temp) == (1e-4, 128, True, False, 0.95, 200, 200)
    assert (a.learning_rate, a.weight_decay, a.gradient_clip, a.rtol, a.atol, a.mxstep, a.seed) == (1e-3, 1e-4, 1.0, 1e-5, 1e-5, 1000, None)
    assert build_parser().parse_args(["--hutch"]).hutchs                       # README's --hutch works by prefix matching
    lr = __import__("mfm_amd.exe_flow_matching", fromlist=["x"]).create_learning_rate_fn(400, 0, 1e-3)
    assert lr(0) == 1e-3 and abs(lr(100) - 7.5e-4) < 1e-18 and lr(400) == 0.0


def test_chain_sharding():
    from mfm_amd.engine import shard
    assert shard(32768, 3, 8) == (4096, 12288, 4096)
    with pytest.raises(ValueError):
        shard(100, 0, 8)
    assert shard(64, 3, 8) == (16, 24, 8)    # 8 chains per GPU: padded to one MFMA M-tile, global ids stay those of the chains
    assert shard(100, 0, 1) == (112, 0, 100)  # --num_chain takes any integer (multi_modal.py:169)
