"""N-replica arithmetic of the HIP path on ONE device ("N ranks x B vs 1 rank x N*B ... emulated on one device by
sequential micro-batches"; VERDICT r5 missing #2).

The reference wraps Adam in CrossShardOptimizer (`model_helper.py:405-406`): every replica takes the gradient of ITS shard's
loss / N (+ the L2 term / N), clips each tensor LOCALLY to norm 2 (`:411-416`), the clipped gradients are SUMMED across replicas
and every replica applies the same Adam update (`:417`, `train.py:157-160`).  `tests/test_dp_gloo.py` checks the exchange with
ORACLE gradients on CPU, and every other `-m gpu` data-parallel test runs on a 1-rank group where 1 / N = 1.  Here two
`LasModel(world_size=2)` replicas run one after the other on the two halves of a B = 8 batch: forward, backward, norms, clip on
the device with `grad_scale = 1/2` and `l2 / 2`, the two flat clipped buffers are added by hand (what the all-reduce does), then
`las_adam_update` on both.  Compared with `O.train_step(..., n_replicas=2)` per replica, with the serial sum of the gloo test,
and -- second test -- with the same step taken through the two exchange buckets of `enable_exchange_overlap()` on a 1-rank RCCL
group, the peer's clipped gradient pre-filled."""
import socket

import pytest
import torch

from tests.helpers import make_hparams, make_batch, to_device

pytestmark = pytest.mark.gpu
WORLD = 2
TOL = 1e-2          # worst gradient element against the oracle's bf16-storage model, relative to the tensor's max-abs (as test_gpu_model)


def _setup(att='luong'):
    from oracle import las_oracle as O
    ohp, params = make_hparams(F=13, L=2, H=64, V=11, att=att, lr=1e-3, l2=1e-4)
    # the reference's U(-0.075, 0.075) initialisation leaves every gradient norm of this toy far below the clip; ten times larger
    # weights give a mix: the listener kernels are clipped (norms 3-10), speller and biases are not (0.3-1.5)
    op = {k: v * 10.0 for k, v in O.init_params(ohp, bias_scale=0.1).items()}
    gb = make_batch(B=8, T=12, F=13, V=11, U=6, src_len=[12, 7, 10, 12, 9, 12, 5, 11], tgt_len=[6, 4, 5, 6, 3, 6, 2, 5], seed=3)
    return O, ohp, params, op, gb


def _replica_backward(model, shard, overlap=False):
    """One replica's local part of the step: returns (audio loss of the shard as the device reports it, flat clipped gradients)."""
    feats, labels = to_device(shard)
    model.vars.grad.zero_()
    audio, _, dlogits = model.forward_train(feats, labels)
    if overlap:
        model.backward_exchange_end(model.backward_exchange_begin(dlogits))      # clip per bucket; the 1-rank all-reduce is the identity
    else:
        model.backward(dlogits)
        model.collect_status(zero_norms=True)
        model.clip_gradients()
    torch.cuda.synchronize()
    model.check_device_status()
    return float(audio), model.vars.grad.clone()


def _check_against_oracle(O, ohp, op, gb, dp, models, locals_, lr):
    # per replica: the oracle's clipped gradient of loss_r / N on that replica's shard
    refs = []
    for r in range(WORLD):
        out = O.train_step(ohp, op, None, None, 1, dp.shard_batch(gb, r, WORLD), mxu='bf16', n_replicas=WORLD)
        refs.append(out)
        m = models[r]
        for name in op:
            ref = out['clipped'][name]
            got = m.vars.grads[name].double().cpu()
            err = float((got - ref).abs().max() / (ref.abs().max() + 1e-30))
            assert err < TOL, (r, name, err)
        # the REPORTED loss is the shard's own (CrossShardOptimizer scales what it differentiates, not what the estimator logs) ...
        assert abs(locals_[r][0] - float(out['audio_loss'])) < 2e-3 * abs(float(out['audio_loss'])), (r, locals_[r][0], float(out['audio_loss']))
        # ... and the gradient is that of loss_r / N: tensors the clip leaves alone are HALF the single-replica gradient
        single = O.train_step(ohp, op, None, None, 1, dp.shard_batch(gb, r, WORLD), mxu='bf16', n_replicas=1)
        free = [k for k in op if float(single['grads'][k].norm()) < O.GRAD_NORM]
        assert len(free) >= 3 and len(op) - len(free) >= 3, 'the case must hold clipped tensors and tensors the clip leaves alone'
        for name in free:
            ref = single['clipped'][name] / WORLD
            got = m.vars.grads[name].double().cpu()
            assert float((got - ref).abs().max() / (ref.abs().max() + 1e-30)) < TOL, (r, name)
    # the serial sum of tests/test_dp_gloo.py
    serial = {k: sum(refs[r]['clipped'][k] for r in range(WORLD)) for k in op}
    return refs, serial


def test_two_replicas_on_one_device_match_the_oracle_and_the_serial_sum():
    from phones_las_amd import dp, model_helper as mh
    O, ohp, params, op, gb = _setup()
    models = [mh.LasModel(params, world_size=WORLD, rank=r) for r in range(WORLD)]
    for m in models:
        m.load_variables(op)
        assert m.tail_buckets is None                       # (the single-replica tail split must not be taken)
    locals_ = [_replica_backward(models[r], dp.shard_batch(gb, r, WORLD)) for r in range(WORLD)]
    refs, serial = _check_against_oracle(O, ohp, op, gb, dp, models, locals_, 1e-3)
    # the all-reduce, by hand: every replica ends up with the SUM of the clipped buffers
    total = locals_[0][1] + locals_[1][1]
    for m in models:
        m.vars.grad.copy_(total)
    for name in op:
        got = models[0].vars.grads[name].double().cpu()
        err = float((got - serial[name]).abs().max() / (serial[name].abs().max() + 1e-30))
        assert err < TOL, (name, err)
    for m in models:
        m.adam_update()
    torch.cuda.synchronize()
    assert torch.equal(models[0].vars.flat, models[1].vars.flat)           # identical parameters on both replicas
    assert torch.equal(models[0].vars.m, models[1].vars.m) and torch.equal(models[0].vars.v, models[1].vars.v)
    assert int(models[0].step_dev.item()) == 2 and int(models[1].step_dev.item()) == 2
    zeros = {k: torch.zeros_like(x) for k, x in op.items()}
    newp, _, _ = O.adam_apply(op, zeros, zeros, serial, 1, 1e-3)
    for name in op:
        moved_ref = newp[name] - op[name]
        moved = models[0].vars.params[name].double().cpu() - op[name]
        # first Adam step: every element moves by ~lr * sign(g); elements whose summed gradient is at rounding level may flip
        big = serial[name].abs() > 1e-2 * serial[name].abs().max()
        assert float((moved - moved_ref)[big].abs().max()) < 2e-5, name


def test_two_bucket_exchange_with_a_prefilled_peer_equals_the_plain_sum():
    """The same step through `enable_exchange_overlap()`'s two buckets on a 1-rank RCCL group (its all-reduce is the identity;
    the peer's clipped gradient is added by hand): bucket-wise norms + clip with 1/N scaling must leave the same buffers as
    the plain pass."""
    import torch.distributed as dist
    from phones_las_amd import dp, model_helper as mh
    O, ohp, params, op, gb = _setup()
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        group = dist.group.WORLD
        plain = [mh.LasModel(params, world_size=WORLD, rank=r) for r in range(WORLD)]
        bucketed = [mh.LasModel(params, world_size=WORLD, rank=r, process_group=group) for r in range(WORLD)]
        for m in plain + bucketed:
            m.load_variables(op)
        for m in bucketed:
            assert len(m.enable_exchange_overlap()) == 2
        lp = [_replica_backward(plain[r], dp.shard_batch(gb, r, WORLD)) for r in range(WORLD)]
        lb = [_replica_backward(bucketed[r], dp.shard_batch(gb, r, WORLD), overlap=True) for r in range(WORLD)]
        for r in range(WORLD):
            assert lp[r][0] == lb[r][0]
            assert torch.equal(lp[r][1], lb[r][1]), 'replica %d: bucket-wise clip differs from the plain pass' % r
        _check_against_oracle(O, ohp, op, gb, dp, bucketed, lb, 1e-3)
        total = lb[0][1] + lb[1][1]                         # the peer's buffer, pre-filled
        for m in bucketed + plain:
            m.vars.grad.copy_(total)
            m.adam_update()
        torch.cuda.synchronize()
        for m in bucketed[1:] + plain:
            assert torch.equal(bucketed[0].vars.flat, m.vars.flat)
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()
