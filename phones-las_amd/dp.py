"""Data parallelism for the LAS train step: one process per GPU, RCCL (torch.distributed backend "nccl") over
xGMI; `gloo` on CPU for tests.  Mirrors what the reference gets from tf.tpu.CrossShardOptimizer
(model_helper.py:405-406, train.py:129-140,157-160): every replica holds all variables, trains its own shard of
the global batch with its loss scaled by 1/N, clips ITS OWN per-tensor gradients, then gradients are SUMMED
across replicas and Adam is applied identically everywhere.  The only exchange per step is one all-reduce over
the flat fp32 gradient buffer (26 MB for the metric-M model)."""
import os

import torch
import torch.distributed as dist

__all__ = ['init_from_env', 'shard_batch', 'all_reduce_sum_', 'mean_scalar', 'world_size', 'rank', 'barrier',
           'broadcast_int', 'sum_floats']


def init_from_env(backend=None, device=None):
    """Join the job torch.distributed.run started (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  Returns (rank, world,
    local_rank).  A single process (WORLD_SIZE unset or 1) does not create a process group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rk = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = device
        dist.init_process_group(backend, rank=rk, world_size=world, **kw)
    return rk, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_batch(tensors, rk=None, world=None):
    """Split every [B, ...] tensor of a dict along the batch into `world` equal contiguous shards and return this
    rank's (the TPUEstimator splits train_batch_size the same way, train.py:157-160).  B must divide evenly."""
    rk = rank() if rk is None else rk
    world = world_size() if world is None else world
    out = {}
    for k, v in tensors.items():
        B = v.shape[0]
        if B % world:
            raise ValueError('global batch %d is not divisible by %d replicas' % (B, world))
        n = B // world
        out[k] = v[rk * n:(rk + 1) * n].contiguous()
    return out


def all_reduce_sum_(flat, group=None):
    """In-place cross-replica SUM of a flat buffer (no-op for a single process)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def mean_scalar(value, group=None):
    """Mean of a python float / 0-d tensor over replicas (logging only)."""
    t = torch.as_tensor(float(value), dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == 'nccl':
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t = t / dist.get_world_size(group)
    return float(t)


def _dev_of(group=None):
    return 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'


def barrier(group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group=group)


def broadcast_int(value, src=0, group=None):
    """Every replica returns rank `src`'s integer (decisions taken by one clock: when to evaluate, when to stop)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=_dev_of(group))
    dist.broadcast(t, src=src, group=group)
    return int(t.item())


def sum_floats(values, group=None):
    """Element-wise sum of a short list of python floats over the replicas (evaluation metrics of sharded batches)."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.to(_dev_of(group))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [float(x) for x in t.cpu()]
