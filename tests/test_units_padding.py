"""Any --encoder_units / --decoder_units (las/ops.py:10-12 takes any integer): widths the kernels are not built for run
zero-padded to the next supported one (model_helper.physical_params).  CPU part: the logical -> physical index maps.
GPU part: a padded model against the oracle at the LOGICAL widths, with the tolerances of tests/test_gpu_model.py, and the
padding staying at exactly zero through gradients, clip, Adam and weight noise."""
import numpy as np
import pytest
import torch

from tests.helpers import make_hparams, make_batch, to_device, relerr


CASES = {
    'luong_96_96': dict(att='luong', H=96, Hd=96),                                      # both -> 128, state handed over
    'bahdanau_80_48': dict(att='bahdanau', H=80, Hd=48, pass_hidden=False),             # 128 / 64
    'luong_64_40': dict(att='luong', H=64, Hd=40, pass_hidden=False),                   # only the decoder is padded
    'general_72_104': dict(att='luong_monotonic', H=72, Hd=104, pass_hidden=False, dec_layers=2, als=24, emb=16,
                           ctc=0.3, L=3),                                              # general decoder, CTC head, 3 layers
    'stacked_40_40': dict(att='custom', H=40, Hd=40, pyramidal=False, pass_hidden=False),
    'luong_600_520': dict(att='luong', H=600, Hd=520, pass_hidden=False),               # both -> 1024 (32-member chains)
}


@pytest.mark.parametrize('case', sorted(CASES))
def test_index_maps_cover_the_logical_model(case):
    from oracle import las_oracle as O
    from phones_las_amd import model_helper as mh
    ohp, params = make_hparams(**CASES[case])
    phys = mh.physical_params(params)
    assert phys is not params and params.encoder.num_units == CASES[case]['H']      # the caller's hparams are left alone
    assert phys.encoder.num_units in mh.SUPPORTED_UNITS and phys.decoder.num_units in mh.SUPPORTED_UNITS
    logical, physical = mh.param_table(params), mh.param_table(phys)
    assert [(n, tuple(s)) for n, s, _ in logical] == [(n, tuple(s)) for n, s, _ in O.param_table(ohp)]
    maps = mh.pad_index_maps(params, phys)
    for (name, ls, _), (pname, ps, _) in zip(logical, physical):
        assert name == pname and len(ls) == len(ps) == len(maps[name])
        for n_log, n_phys, m in zip(ls, ps, maps[name]):
            assert len(m) == n_log and (np.diff(m) > 0).all() and m[0] >= 0 and m[-1] < n_phys
    # a gate block of the first listener kernel: logical column g*H + u sits at g*H' + u
    H, Hp = params.encoder.num_units, phys.encoder.num_units
    cols = maps[logical[0][0]][1]
    assert [int(cols[g * H + 3]) for g in range(4)] == [g * Hp + 3 for g in range(4)]
    # supported widths are not touched
    _, plain = make_hparams(att='luong', H=128, Hd=64, pass_hidden=False)
    assert mh.physical_params(plain) is plain
    with pytest.raises(ValueError):
        mh.physical_units(1030)


def _padded_model(case):
    from oracle import las_oracle as O
    from phones_las_amd import model_helper as mh
    ohp, params = make_hparams(**CASES[case])
    op = O.init_params(ohp, bias_scale=0.1)
    model = mh.LasModel(params)
    assert model.padded and model.vars.padding_is_zero('params')
    model.load_variables(op)
    assert model.vars.padding_is_zero('params')
    for n, t in model.vars.logical('params').items():
        assert torch.equal(t.cpu(), op[n].float()), n
    return O, ohp, op, model


@pytest.mark.gpu
@pytest.mark.parametrize('case', sorted(CASES))
def test_padded_model_matches_the_oracle_at_the_logical_widths(case):
    O, ohp, op, model = _padded_model(case)
    L = CASES[case].get('L', 2)
    T = 12 if L < 3 else 48                                 # (the CTC head needs T / 2^(L-1) frames >= the label count)
    batch = make_batch(T=T, src_len=[T, T - 5, T - 2], tgt_len=[6, 4, 5])
    feats, labels = to_device(batch)
    out = O.train_step(ohp, op, None, None, 1, batch, mxu='bf16')
    model.vars.grad.zero_()
    loss, logits, dlogits = model.forward_train(feats, labels)
    V = ohp.decoder.target_vocab_size
    ref_logits = out['aux']['logits']
    for b, n in enumerate([6, 4, 5]):
        assert relerr(logits[b, :n, :V], ref_logits[b, :n]) < 2e-2
    total = loss + model.l2_loss()
    assert abs(float(total) - float(out['loss'])) < 2e-2 * abs(float(out['loss']))
    model.backward(dlogits)
    torch.cuda.synchronize()
    assert model.vars.padding_is_zero('grads')              # exactly: the padded units never receive a gradient
    tol = 2e-2 if 'general' in case else 1e-2               # tests/test_gpu_model.py: 2x for the general decoder
    grads = model.vars.logical('grads')
    l2 = float(model.params.l2_reg_scale)
    for name, g in grads.items():
        r = out['grads'][name] - l2 * op[name]              # the device adds the L2 term in the norms pass
        assert relerr(g, r) < tol, name


@pytest.mark.gpu
def test_padding_stays_zero_through_the_train_op_and_weight_noise():
    O, ohp, op, model = _padded_model('bahdanau_80_48')
    p = model.params
    p.add_hparam('add_noise', 1) if not hasattr(p, 'add_noise') else p.set_hparam('add_noise', 1)
    p.set_hparam('noise_std', 0.05)
    batch = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    feats, labels = to_device(batch)
    before = {n: t.clone() for n, t in model.vars.logical('params').items()}
    losses = [float(model.train_step(feats, labels)) for _ in range(3)]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses))
    for which in ('params', 'grads', 'm', 'v'):
        assert model.vars.padding_is_zero(which), which
    after = model.vars.logical('params')
    assert any(not torch.equal(before[n], after[n]) for n in before)
    # predictions speak the logical widths
    pred = model.predict(feats)
    assert pred['encoder_out'].shape[-1] == 2 * 2 * CASES['bahdanau_80_48']['H']
    assert int((pred['encoder_out'] != 0).sum()) > 0
