"""World-size-2 (gloo, CPU) test of the N>1 path: chain sharding, shard-invariant draws, the SUM all-reduce of the
flow-matching gradient / loss, the log-likelihood all-gather for the beta bisection and global acceptance statistics.
The per-shard computation is the oracle here (no GPU in this container); on the GPU box the same host plumbing
(mfm_amd/engine.py) wraps the HIP kernels, whose own shard invariance is covered by tests/test_gpu_*.py."""
import os

import numpy as np
import torch
import torch.distributed as td
import torch.multiprocessing as mp


def _worker(rank, world, port, out, n_total):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from mfm_amd import engine
    from oracle import flow, fm, loop, mala, prng, targets
    from oracle.vfield import VectorFieldNet
    from tests import gpu_util as gu
    d = 16
    _, off, n_local = engine.shard(n_total, rank, world)      # rows on the GPU (a multiple of 16), first global id, CHAINS of this rank
    dist = targets.PhiFour(d)
    x = dist.initialize_model(prng.PRNGKey(3), n_total, start=off, count=n_local)
    model = VectorFieldNet(prng.normal(prng.PRNGKey(4), (8,)), dist, [16, 16], [16, 16], [16, 16])
    params = gu.rand_params(model, seed=2)
    key = prng.PRNGKey(5)
    # MALA on the shard: keys are indexed by GLOBAL chain id
    vg = targets.Tempered(dist, 1.0).value_and_grad
    st = mala.init(x, vg)
    keys = prng.split_at(key, n_total, np.arange(off, off + n_local))
    st, info, _ = mala.kernel(keys, st, vg, 1e-3)
    # FM loss / grad on the shard, then ONE all-reduce(SUM)
    loss, grads = fm.loss_and_grad(model, params, key, st.position, 1e-4, n_total=n_total, start=off)
    g = torch.from_numpy(gu.flat_params(grads).astype(np.float64))
    l = torch.tensor([loss], dtype=torch.float64)
    engine.allreduce_sum_(g, l)
    ll = engine.allgather_cat(torch.from_numpy(dist.loglik(st.position)))
    beta = flow.beta_fn(0.0, ll.numpy(), 0.95, n_total)
    m, s = engine.global_mean_std(torch.from_numpy(info.acceptance_rate), n_total)
    # deferred optimizer step: the all-reduce is in flight until the parameters are next needed
    applied = []
    buf = torch.full((5,), float(rank + 1), dtype=torch.float64)
    d = engine.DeferredAllReduce(lambda: applied.append(buf.clone()))
    d.submit(buf)
    assert applied == [] and d.armed
    d.flush(); d.flush()
    assert len(applied) == 1 and torch.equal(applied[0], torch.full((5,), 3.0, dtype=torch.float64))
    buf.fill_(float(rank)); d.submit(buf); buf2 = buf          # a second submit flushes nothing (already flushed)
    d.submit(buf2)                                              # ... but a submit while armed flushes the first one
    d.flush()
    assert len(applied) == 3
    # the in-library split form (context-owned communicator, MFM_NO_FUSED_AT_N=1): submit() starts the reduction through the library's
    # begin call and issues no torch collective; the optimizer step (which awaits the reduction inside the library) runs at flush
    order = []
    buf3 = torch.full((3,), float(rank + 1), dtype=torch.float64)

    def begin(t):                                              # stands in for mfm_grad_allreduce_begin
        order.append("begin"); td.all_reduce(t, op=td.ReduceOp.SUM)

    d2 = engine.DeferredAllReduce(lambda: order.append(("apply", buf3.clone())), begin)
    d2.submit(buf3)
    assert order == ["begin"] and d2.armed and d2.work == []
    d2.flush(); d2.flush()
    assert len(order) == 2 and order[1][0] == "apply" and torch.equal(order[1][1], torch.full((3,), 3.0, dtype=torch.float64))
    if rank == 0:
        np.savez(out, g=g.numpy(), l=l.numpy(), pos=st.position, beta=beta, m=m.item(), s=s.item(), ll=ll.numpy())
    td.destroy_process_group()


import pytest


@pytest.mark.parametrize("n_total", [64, 40])      # 40: 20 chains per rank, shards padded to 32 rows on the GPU (engine.shard)
def test_two_ranks_match_single_process(tmp_path, n_total):
    out = str(tmp_path / "r0.npz")
    port = 29500 + (os.getpid() + n_total) % 2000
    mp.spawn(_worker, args=(2, port, out, n_total), nprocs=2, join=True)
    z = np.load(out)
    from oracle import flow, fm, mala, prng, targets
    from oracle.vfield import VectorFieldNet
    from tests import gpu_util as gu
    d = 16
    dist = targets.PhiFour(d)
    x = dist.initialize_model(prng.PRNGKey(3), n_total)
    model = VectorFieldNet(prng.normal(prng.PRNGKey(4), (8,)), dist, [16, 16], [16, 16], [16, 16])
    params = gu.rand_params(model, seed=2)
    key = prng.PRNGKey(5)
    vg = targets.Tempered(dist, 1.0).value_and_grad
    st, info, _ = mala.kernel(prng.split(key, n_total), mala.init(x, vg), vg, 1e-3)
    np.testing.assert_array_equal(z["pos"], st.position[:n_total // 2])      # rank 0's chains == the first half
    loss, grads = fm.loss_and_grad(model, params, key, st.position, 1e-4)
    np.testing.assert_allclose(z["l"][0], loss, rtol=1e-12)
    gf = gu.flat_params(grads).astype(np.float64)              # float32 gradients summed over two shards
    np.testing.assert_allclose(z["g"], gf, rtol=1e-5, atol=1e-6 * np.abs(gf).max())
    np.testing.assert_allclose(z["ll"], dist.loglik(st.position), rtol=1e-13)
    assert abs(z["beta"] - flow.beta_fn(0.0, dist.loglik(st.position), 0.95, n_total)) < 1e-12
    assert abs(z["m"] - info.acceptance_rate.mean()) < 1e-12 and abs(z["s"] - info.acceptance_rate.std()) < 1e-10
