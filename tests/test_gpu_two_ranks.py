"""Two ranks on ONE GPU (gloo with device tensors; RCCL refuses two ranks on the same device) run the sharded loop
(`exe_flow_matching.run` through `Engine`: chain sharding, deferred gradient all-reduce + AdamW, batched metric
reduction) and must reproduce the single-process run of the same seed: MALA chains bit-for-bit before the first flow
step, losses / parameters within float32 summation-order noise."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _args(num_chain):
    from oracle import loop
    deep = os.environ.get("MFM_TEST_HIDDEN") == "deep"      # three layers on the x branch, one on the t branch, ragged widths: the wide family
    hid = dict(hidden_x=[32, 40, 24], hidden_t=[20], hidden_xt=[48, 32]) if deep else dict(hidden_x=[32, 32], hidden_t=[32, 32], hidden_xt=[32, 32])
    return loop.default_args(example="phi-four", dim=64, num_chain=num_chain, learning_iter=7, mcmc_per_flow_steps=3.0,
                             hutchs=True, fourier_dim=16, seed=1024, eval_iter=1, step_size=1e-4, **hid)


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from mfm_amd import distributions as D, exe_flow_matching as E
    res, res_, ex = E.run(D.PhiFour(64), _args(64), None, log_every=3, return_extras=True)
    eng = ex["engine"]
    assert (eng.world, eng.n_local, eng.offset) == (world, 32, 32 * rank)
    np.savez(out % rank, metrics=ex["metrics"], pos=ex["states"].position.cpu().numpy(), params=eng.ctx.get_params(),
             opt=np.array([eng.ctx.opt_state()[k] for k in ("step", "count")]), res=res, res_=res_,
             idx=ex["final"]["idx"].cpu().numpy(), flow=ex["flow_samples"].cpu().numpy(), logw=ex["final"]["log_weights"].cpu().numpy())
    eng.close()
    td.destroy_process_group()


def test_two_ranks_reproduce_single_process(tmp_path):
    import torch.multiprocessing as mp
    from mfm_amd import distributions as D, exe_flow_matching as E
    out = str(tmp_path / "r%d.npz")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res, res_, ex = E.run(D.PhiFour(64), _args(64), None, log_every=1000, return_extras=True)
    m1 = ex["metrics"]
    p1 = ex["engine"].ctx.get_params()
    pos1 = ex["states"].position.cpu().numpy()
    opt1 = ex["engine"].ctx.opt_state()
    fin1 = {k: v.cpu().numpy() for k, v in ex["final"].items()}
    ex["engine"].close()
    z = [np.load(out % r) for r in range(2)]
    np.testing.assert_array_equal(z[0]["metrics"], z[1]["metrics"])            # every rank logs the global numbers
    np.testing.assert_array_equal(z[0]["params"], z[1]["params"])              # replicas stay identical
    m2 = z[0]["metrics"]
    np.testing.assert_allclose(m2[:3, 0], m1[:3, 0], rtol=1e-6)                # loss: same chains, same noise, other summation order
    np.testing.assert_allclose(m2[:, 0], m1[:, 0], rtol=5e-3)                  # after a flow step a borderline decision may flip
    np.testing.assert_allclose(m2[:3, 1:3], m1[:3, 1:3], atol=1e-6)            # acceptance mean / std over ALL chains
    assert tuple(z[0]["opt"]) == (opt1["step"], opt1["count"])
    assert np.abs(z[0]["params"] - p1).max() < 2e-3 * max(1.0, np.abs(p1).max())
    pos2 = np.concatenate([z[0]["pos"], z[1]["pos"]])
    close = np.abs(pos2 - pos1).max(1) < 2e-2
    assert close.sum() >= 60, close.sum()
    # end-of-run evaluation (exe_flow_matching.py:453-490) is GLOBAL: every rank integrates its slice of the N reference draws,
    # the samples and log-weights are all-gathered and the resampling / metrics run over all N on every rank
    for k in ("idx", "flow", "logw"):
        np.testing.assert_array_equal(z[0][k], z[1][k], err_msg=k)
    for k in ("res", "res_"):                                                           # [logpdf, KSD-U, KSD-V, MMD | wall-clock train_time]
        np.testing.assert_array_equal(z[0][k][:4], z[1][k][:4], err_msg=k)
    assert z[0]["flow"].shape == fin1["flow_samples"].shape == (64, 64)
    # against the single-process run: the same draws through parameters that differ by float32 summation order amplified by
    # seven Adam updates (bounded by 2e-3 above): O(1) samples move by up to a few 1e-2 (observed 1.3e-2 .. 3.1e-2 over builds)
    assert np.abs(z[0]["flow"] - fin1["flow_samples"]).max() < 5e-2
    np.testing.assert_allclose(z[0]["res"][:3], res[:3], rtol=5e-2, atol=1e-3)          # logpdf, KSD U / V of the flow samples


def test_two_ranks_with_a_deep_ragged_network_reproduce_single_process(tmp_path, monkeypatch):
    """The same through the wide family (hidden lists 3 / 1 / 2 with widths that are not multiples of 16): an eight-layer gradient vector
    summed over two ranks, deferred AdamW, replicas identical, losses against the single-process run."""
    import torch.multiprocessing as mp
    from mfm_amd import distributions as D, exe_flow_matching as E
    monkeypatch.setenv("MFM_TEST_HIDDEN", "deep")
    out = str(tmp_path / "r%d.npz")
    port = 29500 + (os.getpid() + 7) % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res, res_, ex = E.run(D.PhiFour(64), _args(64), None, log_every=1000, return_extras=True)
    m1, p1, opt1 = ex["metrics"], ex["engine"].ctx.get_params(), ex["engine"].ctx.opt_state()
    assert ex["engine"].ctx.n_params == sum(fi * fo + fo for fi, fo in E.layer_shapes(64, 16, [32, 40, 24], [20], [48, 32]))
    ex["engine"].close()
    z = [np.load(out % r) for r in range(2)]
    np.testing.assert_array_equal(z[0]["metrics"], z[1]["metrics"])
    np.testing.assert_array_equal(z[0]["params"], z[1]["params"])
    np.testing.assert_allclose(z[0]["metrics"][:3, 0], m1[:3, 0], rtol=1e-6)
    np.testing.assert_allclose(z[0]["metrics"][:, 0], m1[:, 0], rtol=5e-3)
    assert tuple(z[0]["opt"]) == (opt1["step"], opt1["count"])
    assert np.abs(z[0]["params"] - p1).max() < 2e-3 * max(1.0, np.abs(p1).max())


def _rccl_worker(rank, world, port, out, in_lib):
    import torch
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MFM_COLLECTIVES_AT_WORLD1="1")
    os.environ.update({0: {"MFM_TORCH_ALLREDUCE": "1"}, 1: {"MFM_NO_FUSED_AT_N": "1"}, 2: {}}[in_lib])
    torch.cuda.set_device(0)
    td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from mfm_amd import distributions as D, exe_flow_matching as E
    res, res_, ex = E.run(D.PhiFour(64), _args(64), None, log_every=3, return_extras=True)
    eng = ex["engine"]
    assert eng._split_calls                                   # the multi-rank configuration (collectives issued)
    assert eng.rccl_in_lib == bool(in_lib)                    # torch.distributed's RCCL backend, or the context's own communicator
    assert eng._fused_n == (in_lib == 2)                      # one mfm_train_iter per iteration with the all-reduce in line (the default)
    np.savez(out % rank, metrics=ex["metrics"], pos=ex["states"].position.cpu().numpy(), params=eng.ctx.get_params(),
             opt=np.array([eng.ctx.opt_state()[k] for k in ("step", "count")]), res=res,
             idx=ex["final"]["idx"].cpu().numpy(), flow=ex["flow_samples"].cpu().numpy())
    eng.close()
    td.destroy_process_group()


@pytest.mark.parametrize("in_lib", [0, 1, 2])
def test_rccl_call_pattern_on_a_one_rank_communicator(tmp_path, in_lib):
    """RCCL refuses two ranks on one device, so what a one-GPU box can check of the backend the 8-GPU runs use is the CALL
    PATTERN on a one-rank `nccl` communicator.  A sum over one rank is the identity, so each form must reproduce the single-call
    run bit for bit.  in_lib = 2, the DEFAULT with more than one rank: the context owns the communicator (mfm_comm_init), every
    iteration is one mfm_train_iter -- MALA step inside the training kernel, ncclAllReduce in line on the context's stream behind the
    weight-gradient kernel, AdamW behind it -- plus the all-gathers of the final evaluation.  in_lib = 1 (MFM_NO_FUSED_AT_N): separate
    MALA / loss-gradient calls, the all-reduce started on the context's communication stream (mfm_grad_allreduce_begin), AdamW
    deferred behind the next MALA step.  in_lib = 0 (MFM_TORCH_ALLREDUCE): the same split sequence through torch.distributed."""
    import torch.multiprocessing as mp
    from mfm_amd import distributions as D, exe_flow_matching as E
    out = str(tmp_path / "n%d.npz")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_rccl_worker, args=(1, port + in_lib, out, in_lib), nprocs=1, join=True)
    res, res_, ex = E.run(D.PhiFour(64), _args(64), None, log_every=1000, return_extras=True)
    z = np.load(out % 0)
    np.testing.assert_array_equal(z["params"], ex["engine"].ctx.get_params())
    np.testing.assert_array_equal(z["pos"], ex["states"].position.cpu().numpy())
    np.testing.assert_array_equal(z["metrics"][:, :3], ex["metrics"][:, :3])
    np.testing.assert_array_equal(z["idx"], ex["final"]["idx"].cpu().numpy())
    np.testing.assert_array_equal(z["flow"], ex["flow_samples"].cpu().numpy())
    np.testing.assert_array_equal(z["res"][:4], res[:4])
    opt = ex["engine"].ctx.opt_state()
    assert tuple(z["opt"]) == (opt["step"], opt["count"])
    ex["engine"].close()


def _smc_worker(rank, world, port, out):
    import torch
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from mfm_amd import distributions as D, exe_others as X
    a = _args(128); a.do_smc = True; a.learning_iter = 10; a.eval_iter = 2
    res, res_, ex = X.run(D.PhiFour(64), a, None, return_extras=True)
    eng = ex["engine"]
    assert (eng.world, eng.n_local, eng.offset) == (world, 64, 64 * rank)
    assert ex["state"].particles.shape == (64, 64) and ex["state"].weights.shape == (128,)
    np.savez(out % rank, lmbdas=ex["lmbdas"], samples=ex["samples"].cpu().numpy(), res=res, shard=ex["state"].particles.cpu().numpy(),
             weights=ex["state"].weights.cpu().numpy())
    eng.close()
    td.destroy_process_group()


def test_two_ranks_run_the_smc_baseline_like_one_process(tmp_path):
    """Adaptive tempered SMC on the MALA kernel (exe_others.py:79-111) with the particles sharded over two ranks: resampling and the
    ESS solver work on ALL particles (weights and log-likelihoods all-gathered, ancestors fetched from the gathered particles), the
    MCMC moves on each rank's shard with the particle's own key -- so the run is the single-process run, bit for bit."""
    import torch.multiprocessing as mp
    from mfm_amd import distributions as D, exe_others as X
    out = str(tmp_path / "s%d.npz")
    port = 33500 + os.getpid() % 2000
    mp.spawn(_smc_worker, args=(2, port, out), nprocs=2, join=True)
    a = _args(128); a.do_smc = True; a.learning_iter = 10; a.eval_iter = 2
    res, res_, ex = X.run(D.PhiFour(64), a, None, return_extras=True)
    z = [np.load(out % r) for r in range(2)]
    assert ex["lmbdas"][-1] > ex["lmbdas"][0] >= 0.0
    for r in range(2):
        np.testing.assert_array_equal(z[r]["lmbdas"], ex["lmbdas"])
        np.testing.assert_array_equal(z[r]["samples"], ex["samples"].cpu().numpy())
        np.testing.assert_array_equal(z[r]["weights"], ex["state"].weights.cpu().numpy())
        np.testing.assert_array_equal(z[r]["shard"], ex["state"].particles.cpu().numpy()[64 * r:64 * r + 64])
        np.testing.assert_array_equal(z[r]["res"][:4], res[:4])
    ex["engine"].close()
