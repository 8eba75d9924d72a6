"""Convergence twin: the device trains on the same toy corpus, from the same initial weights, on the same
batch sequence as the oracle did in the build container (tests/golden/make_convergence_twin.py -> the fixture
tests/golden/convergence_twin.json: windowed loss-curve checkpoints and the held-out greedy PER of the exact f64 model and of
the oracle's bf16 storage model), 2 400 optimiser steps, dropout and sampling off.

What can be asked of it.  2 400 Adam steps amplify every rounding difference (tests/test_gpu_trajectory.py follows the oracle
step for step over 20), so the three runs -- f64 oracle, bf16-model oracle, device -- are three samples of where training of
this model on this corpus ends, not three copies of one trajectory.  The distance between the two ORACLE runs is the scale:
the device must stay inside a band a few times that wide around the f64 curve (stated below), and its held-out PER -- the
median over six late checkpoints, on 128 utterances / about 640 phones -- within max(1 point, the two oracle runs' own
distance) + 0.5 of the f64 oracle's.
This is the only available stand-in for north_star's "matched TIMIT PER": no TF, no corpus."""
import json
import os

import numpy as np
import pytest
import torch

from tests import twin_corpus as TC
from tests.helpers import make_hparams, to_device

pytestmark = pytest.mark.gpu
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'convergence_twin.json')


def _tensors(b):
    return {k: torch.from_numpy(v.astype(np.float64) if k == 'encoder_inputs' else v.astype(np.int64)) for k, v in b.items()}


def test_the_corpus_survives_the_tfrecord_round_trip(tmp_path):
    """The corpus the twin trains on, written as the reference's SequenceExample records (preprocess_all.py:31-50) and read
    back through the product's reader: same frames, same labels -- the batches below are what the input path would feed."""
    from phones_las_amd.utils import tfrecord
    utts = TC.utterances(40, 11)
    path = os.path.join(str(tmp_path), 'twin.tfr')
    with tfrecord.TFRecordWriter(path) as w:
        for x, ys in utts:
            w.write(tfrecord.make_example(x, ['p%d' % y for y in ys]))
    back = [tfrecord.parse_sequence_example(r, num_channels=TC.F) for r in tfrecord.tf_record_iterator(path, verify=True)]
    assert len(back) == len(utts)
    for (x, ys), (bx, by) in zip(utts, back):
        assert np.array_equal(np.asarray(bx, np.float32).reshape(-1, TC.F), x)
        assert [l.decode() if isinstance(l, bytes) else l for l in by] == ['p%d' % y for y in ys]


def test_device_training_ends_where_the_oracle_does():
    from oracle import las_oracle as O
    from phones_las_amd import model_helper as mh
    fx = json.load(open(FIXTURE))
    assert fx['corpus']['steps'] == TC.STEPS and fx['corpus']['n_train'] == TC.N_TRAIN and fx['corpus']['window'] == TC.WINDOW
    m = TC.MODEL
    ohp, params = make_hparams(F=m['F'], L=m['L'], H=m['H'], Hd=m['Hd'], V=m['V'], att=m['att'], lr=m['lr'], l2=m['l2'])
    model = mh.LasModel(params)
    model.load_variables(O.init_params(ohp, seed=4321))
    batches = [to_device(_tensors(b)) for b in TC.train_batches()]
    tb, refs = TC.test_batches()
    tb = [to_device(_tensors(b))[0] for b in tb]

    def held_out_per():
        hyps = []
        for feats in tb:
            hyps += [row.tolist() for row in model.predict(feats)['sample_ids'].cpu()]
        return TC.per(hyps, refs)

    losses, curve, pers = [], {}, {}
    for t in range(TC.STEPS):
        feats, labels = batches[t % len(batches)]
        losses.append(float(model.train_step(feats, labels)))
        if t + 1 in TC.CHECKPOINTS:
            curve[t + 1] = float(np.mean(losses[-TC.WINDOW:]))
        if t + 1 in TC.PER_STEPS:
            pers[t + 1] = held_out_per()
    model.check_device_status()
    per = float(np.median(list(pers.values())))
    f64, bf = fx['f64'], fx['bf16']
    report = {'per': {'device': per, 'f64': f64['per'], 'bf16': bf['per'], 'device_at': pers, 'f64_at': f64['per_at'], 'bf16_at': bf['per_at']},
              'curve': {k: (round(curve[int(k)], 5), round(f64['curve'][k], 5), round(bf['curve'][k], 5)) for k in f64['curve']}}
    print(json.dumps(report))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out')
    if os.path.isdir(out):
        json.dump(report, open(os.path.join(out, 'convergence_twin_device.json'), 'w'), indent=1)
    # the loss curve: within max(3 x the distance between the two oracle runs, 50 % of the larger oracle value, 5e-3) of the f64
    # curve (late checkpoints sit near zero with occasional Adam spikes on either side: the band is relative to what the two
    # oracle runs themselves do there)
    for k, ref in f64['curve'].items():
        band = max(3.0 * abs(ref - bf['curve'][k]), 0.5 * max(ref, bf['curve'][k]), 5e-3)
        assert abs(curve[int(k)] - ref) <= band, (k, curve[int(k)], ref, bf['curve'][k], band)
    assert curve[TC.STEPS] < 0.05 * curve[TC.CHECKPOINTS[0]]
    # held-out PER (median over the six late checkpoints): within max(1, the two oracle runs' own distance) + 0.5 points of f64
    gap = abs(f64['per'] - bf['per'])
    assert abs(per - f64['per']) <= max(1.0, gap) + 0.5, report['per']
