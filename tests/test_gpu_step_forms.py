"""The step forms bench.py times (bench.StepForms: part_a / part_b launched eagerly or replayed as HIP graphs, with the
tail-bucket split of the train op; the two-bucket exchange forms of a data-parallel run) against LasModel.train_step:
same parameters, Adam slots and loss, BIT FOR BIT, after K optimiser steps from the same weights.  The reference runs one
train op per step (model_helper.py:403-417); a captured-and-replayed step must be that step and nothing else.  bench.py
refuses to time a form that is not listed in bench.COVERED_FORMS, and this file covers every entry of that list."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

K = 6


def _setup(cfg, **model_kw):
    import bench
    from phones_las_amd import model_helper as mh
    c = bench.CONFIGS[cfg]
    dev = torch.device('cuda', 0)
    model = mh.LasModel(bench.build_params(c), **model_kw)
    feats, labels = bench.synthetic_batch(c, 1234, dev)
    feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])
    return bench, c, model, feats, labels


def _same_state(a, b, what):
    torch.cuda.synchronize()
    for name in ('flat', 'm', 'v'):
        ta, tb = getattr(a.vars, name), getattr(b.vars, name)
        assert torch.equal(ta, tb), '%s: vars.%s differs (max |d| = %g)' % (what, name, float((ta - tb).abs().max()))
    assert int(a.step_dev.item()) == int(b.step_dev.item()), what
    assert not a.read_and_clear_status() and not b.read_and_clear_status(), what


@pytest.mark.parametrize('cfg', ['tiny', 'metric-M'])
def test_plain_forms_are_the_train_step_bit_for_bit(cfg):
    """plain_graph (what the driver's bench run timed in round 4) and plain_eager, single replica, tail-bucket split."""
    bench, c, ref, feats, labels = _setup(cfg)
    _, _, mod, _, _ = _setup(cfg)
    assert torch.equal(ref.vars.flat, mod.vars.flat)
    forms = bench.StepForms(mod, feats, labels, c['U'], multi=False)
    forms.warm_up()                                                   # two real optimiser steps
    cands = forms.build(want_eager=True, want_graph=True)              # captures execute nothing
    assert set(cands) == {'plain_eager', 'plain_graph'} and set(cands) <= set(bench.COVERED_FORMS)
    assert mod.tail_buckets is not None                               # the form bench.py times splits the train op
    loss = None
    for _ in range(2):
        loss = ref.train_step(feats, labels, num_steps=c['U'])
    _same_state(ref, mod, 'after the warm-up steps')
    for name in ('plain_graph', 'plain_eager', 'plain_graph'):
        step = cands[name][0]
        for _ in range(K):
            step()
            loss = ref.train_step(feats, labels, num_steps=c['U'])
        _same_state(ref, mod, '%s after %d steps' % (name, K))
        assert torch.equal(loss.reshape(1), forms.loss_buf), (name, float(loss), float(forms.loss_buf))
    assert torch.isfinite(loss).all()


def test_tail_split_train_op_equals_the_one_pass_train_op(monkeypatch):
    """norms + clip + Adam of everything above the bottom listener layer beside that layer's weight-gradient
    products (the default) against the one-pass train op behind a full join (LAS_TAIL_OVERLAP=0)."""
    bench, c, split, feats, labels = _setup('tiny')
    monkeypatch.setenv('LAS_TAIL_OVERLAP', '0')
    _, _, whole, _, _ = _setup('tiny')
    monkeypatch.delenv('LAS_TAIL_OVERLAP')
    assert split.tail_buckets is not None and whole.tail_buckets is None
    for i in range(4):
        la = split.train_step(feats, labels, num_steps=c['U'])
        lb = whole.train_step(feats, labels, num_steps=c['U'])
        _same_state(split, whole, 'step %d' % i)
        assert torch.equal(la, lb)


def test_data_parallel_forms_are_the_train_step_bit_for_bit():
    """The forms of a run with several ranks -- clip, all-reduce, Adam as separate passes; the exchange as one all-reduce
    (plain_*) or in two buckets beside the backward pass (overlap_*) -- on a 1-rank RCCL group."""
    import socket
    import torch.distributed as dist
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        group = dist.group.WORLD
        for overlap in (False, True):
            bench, c, ref, feats, labels = _setup('tiny', process_group=group)
            _, _, mod, _, _ = _setup('tiny', process_group=group)
            if overlap:
                assert len(ref.enable_exchange_overlap()) == 2 and len(mod.enable_exchange_overlap()) == 2
            forms = bench.StepForms(mod, feats, labels, c['U'], multi=True)
            forms.warm_up()
            cands = forms.build(want_eager=True, want_graph=True, overlap_exchange=overlap, plain_too=not overlap)
            names = ['overlap_graph', 'overlap_eager'] if overlap else ['plain_graph', 'plain_eager']
            assert sorted(cands) == sorted(names) and set(cands) <= set(bench.COVERED_FORMS)
            loss = None
            for _ in range(4 if overlap else 2):                      # the eager steps in front of the captures
                loss = ref.train_step(feats, labels, num_steps=c['U'])
            _same_state(ref, mod, 'after the warm-up steps (overlap=%s)' % overlap)
            for name in names:
                for _ in range(K):
                    cands[name][0]()
                    loss = ref.train_step(feats, labels, num_steps=c['U'])
                _same_state(ref, mod, '%s after %d steps' % (name, K))
                assert torch.equal(loss.reshape(1), forms.loss_buf), name
    finally:
        dist.destroy_process_group()


def test_bench_lists_no_form_without_a_test():
    import bench
    covered_here = {'plain_eager', 'plain_graph', 'overlap_eager', 'overlap_graph'}
    assert set(bench.COVERED_FORMS) == covered_here
