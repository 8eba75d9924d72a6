"""Data-parallel path on CPU with the gloo backend, world_size 2 (the N>1 code path of model_helper/dp.py):
shard the global batch, scale each replica's loss by 1/N, clip per tensor LOCALLY, SUM across replicas
(CrossShardOptimizer order, model_helper.py:405-417), then identical Adam everywhere.  Gradients come from the
oracle (the HIP path needs a GPU); the exchange, the flat-buffer layout and the sharding are the product's."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import las_oracle as O
from tests.helpers import make_hparams, make_batch


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_clipped(ohp, op, batch, n):
    out = O.train_step(ohp, op, None, None, 1, batch, n_replicas=n)
    return out['clipped']


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from phones_las_amd import dp, model_helper as mh
    rk, w, _ = dp.init_from_env(backend='gloo')
    assert (rk, w) == (rank, world) and dp.world_size() == world
    ohp, params = make_hparams(F=5, L=2, H=8, V=9)
    op = O.init_params(ohp, bias_scale=0.1)
    gb = make_batch(B=4, T=10, F=5, V=9, U=5, src_len=[10, 6, 8, 3], tgt_len=[5, 3, 4, 2])
    shard = dp.shard_batch(gb, rank, world)
    assert shard['encoder_inputs'].shape[0] == 2
    v = mh.Variables(mh.param_table(params), device='cpu')
    v.load(op)
    mine = _local_clipped(ohp, op, shard, world)
    for n in v.grads:
        v.grads[n].copy_(mine[n].float())
    dp.all_reduce_sum_(v.grad)
    # serial reference: sum over replicas of locally clipped gradients of loss_r / N
    ref = None
    for r in range(world):
        c = _local_clipped(ohp, op, dp.shard_batch(gb, r, world), world)
        ref = c if ref is None else {k: ref[k] + c[k] for k in c}
    err = max(float((v.grads[n].double() - ref[n]).abs().max()) for n in ref)
    # identical Adam on every replica
    zeros = {k: torch.zeros_like(x) for k, x in op.items()}
    newp, _, _ = O.adam_apply(op, zeros, zeros, {n: v.grads[n].double() for n in op}, 1, 1e-3)
    flat = torch.cat([newp[n].reshape(-1) for n in op])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    m = dp.mean_scalar(float(rank))
    ret[rank] = (err, same, m)
    dist.destroy_process_group()


def test_two_replica_gradient_sum_matches_serial_reference():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        err, same, m = ret[r]
        assert err < 1e-6, err
        assert same
        assert abs(m - 0.5) < 1e-12


def test_single_process_is_a_no_op():
    from phones_las_amd import dp
    x = torch.arange(4.0)
    assert dp.world_size() == 1 and dp.rank() == 0
    assert torch.equal(dp.all_reduce_sum_(x.clone()), x)
    assert dp.mean_scalar(3.0) == 3.0


def _bucket_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from phones_las_amd import dp, model_helper as mh
    dp.init_from_env(backend='gloo')
    _, params = make_hparams(F=5, L=3, H=8, V=9)
    table = mh.param_table(params)
    res = {}
    for split in (False, True):
        v = mh.Variables(table, device='cpu')
        if split:
            buckets = v.split_buckets(2 * 4)          # [top listener layer + speller] first, [layers 0..1] second
            assert len(buckets) == 2 and buckets[1]['begin'] == 0 and buckets[0]['end'] == v.total
        g = torch.Generator().manual_seed(100 + rank)
        v.grad.copy_(torch.randn(v.total, generator=g))
        mine = v.grad.clone()
        v.skip_flag.fill_(1.0 if rank == 1 else 0.0)          # replica 1's persistent kernels "timed out"
        for b in (v.buckets if split else [None]):            # bucket order = the order the backward pass completes them
            dp.all_reduce_sum_(v.exchange_view(b))
        other = torch.Generator().manual_seed(100 + (1 - rank))
        want = mine + torch.randn(v.total, generator=other)
        res[split] = (float((v.grad - want).abs().max()), float(v.skip_flag))
        # per-tensor views still address the summed values
        name = table[-1][0]
        assert torch.equal(v.grads[name].reshape(-1), v.grad[v.offsets[-2]:v.offsets[-2] + v.grads[name].numel()])
    ret[rank] = res
    dp.barrier()
    assert dp.broadcast_int(7 if rank == 0 else 99) == 7
    assert dp.sum_floats([1.0, rank]) == [2.0, 1.0]
    dist.destroy_process_group()


def test_two_replica_bucket_exchange_carries_the_timeout_flag():
    """LasModel's exchange pieces (Variables.exchange_view / split_buckets) on two gloo ranks: the whole buffer and the
    two-bucket form both sum the gradients, and the timeout flag set on ONE replica reaches BOTH (flag > 0 makes every
    replica's Adam kernel skip the update)."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bucket_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        for split in (False, True):
            err, flag = ret[r][split]
            assert err < 1e-6 and flag == 1.0, (r, split, err, flag)
