"""Every BASELINE.json configuration at the shape bench.py times it (B, T, U of bench.CONFIGS), for 30 optimiser steps on one
fixed synthetic batch: the loss stays finite and goes down, every parameter stays finite and no persistent kernel reports a
timeout (round 2's only cfg5 run at this size ended in NaN and nothing noticed).  One-step parity at these
shapes is in test_gpu_golden_shapes.py; this is the multi-step check."""
import math

import pytest
import torch

import bench

pytestmark = pytest.mark.gpu

STEPS = 30


@pytest.mark.parametrize('cfg', ['cfg1', 'metric-M', 'metric-L', 'cfg4', 'cfg5'])
def test_thirty_train_steps_at_the_bench_shape_stay_finite_and_learn(cfg):
    from phones_las_amd import model_helper as mh
    c = bench.CONFIGS[cfg]
    dev = torch.device('cuda', 0)
    model = mh.LasModel(bench.build_params(c), binf2phone=bench.binf_matrix(c['binf']) if c.get('binf') else None)
    feats, labels = bench.synthetic_batch(c, 1234, dev)
    feats['encoder_inputs'] = model.listener.pad_features(feats['encoder_inputs'])
    losses = []
    for _ in range(STEPS):
        losses.append(model.train_step(feats, labels, num_steps=c['U']))
    losses = [float(x) for x in torch.cat(losses).cpu()]
    model.check_device_status()                   # raises on a timeout of any persistent kernel
    print(cfg, 'loss %.4f -> %.4f' % (losses[0], losses[-1]))
    assert all(math.isfinite(x) for x in losses), losses
    for name, p in model.vars.params.items():
        assert bool(torch.isfinite(p).all()), name
    assert int(model.step_dev.item()) == STEPS + 1            # no step was skipped by the timeout gate
    first, last = sum(losses[:3]) / 3, sum(losses[-3:]) / 3
    assert last < first - 0.02 * abs(first), (first, last, losses)
