"""GPU parity at the BENCHMARKED and BASELINE shapes against committed oracle fixtures (tests/golden/shape_<case>.npz,
made by tests/golden/make_golden.py in the build container).  The oracle is NOT imported here: weights, batches and
hyper-parameters come from tests/golden_cases.py, expected values from the fixtures.

These are the kernel instantiations bench.py times: lstm_fwd/bwd_kernel<256, 4> over 800-step chains (companions, lean
path), dec_persist_fwd/bwd at Hd=256 / M=1024 / T'=200 / U=80, the 128x128 NT and the fused TN products at K = B*T.

Stated tolerances (fraction of the reference tensor's max-abs; the fixtures hold two models of the reference arithmetic:
'bf16' = float64 arithmetic with the device's bf16 storage points forward AND backward, 'f64' = exact):
  logits 2e-2 vs both models; audio loss 1e-3 relative;
  encoder memory 1.6e-2 (two bf16 ulps), final encoder states 5e-3;
  gradients vs the bf16 model: 1e-2 of the per-tensor max-abs on the sampled elements (2e-2 for metricL_ctc: 66 decoder
  rows), per-tensor norm within 2 %;
  gradients vs the EXACT f64 model (round 3; what bf16 operand storage costs, nothing fitted): 5e-2 of the per-tensor max-abs,
  norm within 1.5 % (measured over the ten cases: worst element 3.1e-2 -- metricL_full speller/query_layer/kernel -- worst
  norm 7.5e-3).
Measured on MI355X (round 2): at bench.py's exact shape (B=64, T=800, U=80) logits 1.3e-3 / 4.5e-3 of max-abs against the
bf16 / exact model, audio loss 2e-9 relative, worst gradient element 1.4e-3, worst norm 9e-5.  The measured errors of the run are written to gpurun_out/golden_shapes_<case>.json (DESIGN.md quotes them)."""
import json
import os

import numpy as np
import pytest
import torch

from tests import golden_cases as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOL = dict(logits=2e-2, loss=1e-3, memory=1.6e-2, state=5e-3, grad=1e-2, gradnorm=2e-2, grad_f64=5e-2, gradnorm_f64=1.5e-2)
# metricL_ctc: 66 decoder rows in all (B*U ragged) -- single bf16 flips of d(logits) / context show in the projection kernel
GRAD_TOL = {'metricL_ctc': 2e-2}


def _fixture(case):
    return np.load(os.path.join(G.GOLDEN, 'shape_%s.npz' % case))


def _model(case):
    from phones_las_amd import model_helper as mh
    c = G.CASES[case]
    binf = G.binf_matrix(c['binf']) if c.get('binf') else None
    model = mh.LasModel(G.product_params(case), seed=G.SEED_PARAMS, binf2phone=binf)
    w = G.weights(case)
    assert list(w) == [n for n, _, _ in model.vars.table]
    model.load_variables({k: torch.from_numpy(v).cuda() for k, v in w.items()})
    nb = G.batch(case)
    feats = {'encoder_inputs': torch.from_numpy(nb['encoder_inputs']).cuda(),
             'source_sequence_length': torch.from_numpy(nb['source_sequence_length']).to(torch.int32).cuda()}
    labels = {k: torch.from_numpy(nb[k]).to(torch.int32).cuda()
              for k in ('targets_inputs', 'targets_outputs', 'target_sequence_length')}
    return model, feats, labels, nb


def _rel(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(np.asarray(got, dtype=np.float64) - ref).max() / (np.abs(ref).max() + 1e-30))


def _run_case(case):
    g = _fixture(case)
    model, feats, labels, nb = _model(case)
    assert abs(float(np.abs(nb['encoder_inputs']).sum()) - float(g['x_checksum'])) < 1e-3      # same inputs as the fixture
    model.vars.grad.zero_()
    loss, logits, dlogits = model.forward_train(feats, labels)
    model.backward(dlogits)
    torch.cuda.synchronize()
    model.check_device_status()
    V = G.CASES[case]['V']
    tl = nb['target_sequence_length']
    lg = logits[..., :V].float().cpu().numpy()
    rep = {'case': case}
    for mxu in ('bf16', 'f64'):
        ref = g[mxu + '_logits']
        err = max(float(np.abs(lg[b, :tl[b]].astype(np.float64) - ref[b, :tl[b]]).max()) for b in range(len(tl)))
        rep['logits_vs_' + mxu] = err / float(np.abs(ref).max())
        rep['audio_loss_vs_' + mxu] = abs(float(loss) - float(g[mxu + '_audio_loss'])) / abs(float(g[mxu + '_audio_loss']))
    names = [str(n) for n in g['names']]
    assert names == [n for n, _, _ in model.vars.table]
    # gradients against BOTH models: 'bf16' (the oracle with the device's storage points, forward and backward) and 'f64' (the
    # exact model: nothing in it was fitted to the device -- ADVICE r2)
    per_tensor = {}
    for mxu, tag in (('bf16', ''), ('f64', '_f64')):
        worst_g, worst_n = ('', 0.0), ('', 0.0)
        for i, n in enumerate(names):
            got = model.vars.grads[n].reshape(-1).cpu().numpy()
            ref = g['%s_grad_%02d' % (mxu, i)]
            e = _rel(got[G.grad_sample(n, got.size)], ref)
            rn = float(g[mxu + '_gradnorm'][i])
            en = abs(float(np.linalg.norm(got.astype(np.float64))) - rn) / (rn if rn > 0 else 1.0)
            per_tensor.setdefault(n, []).extend([e, en])
            if e > worst_g[1]:
                worst_g = (n, e)
            if en > worst_n[1]:
                worst_n = (n, en)
        rep['grad_worst' + tag] = list(worst_g)
        rep['gradnorm_worst' + tag] = list(worst_n)
    rep['grad_per_tensor'] = per_tensor          # name -> [element err, norm err] vs bf16 model, then vs f64 model
    return g, model, rep, (loss, logits)


def _write(rep):
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'golden_shapes_%s.json' % rep['case']), 'w') as f:
        json.dump(rep, f, indent=1)
    print(json.dumps({k: v for k, v in rep.items() if k != 'grad_per_tensor'}))


def _check(rep):
    assert rep['logits_vs_bf16'] < TOL['logits'] and rep['logits_vs_f64'] < TOL['logits'], rep
    assert rep['audio_loss_vs_bf16'] < TOL['loss'] and rep['audio_loss_vs_f64'] < TOL['loss'], rep
    assert rep['grad_worst'][1] < GRAD_TOL.get(rep['case'], TOL['grad']), rep['grad_worst']
    assert rep['gradnorm_worst'][1] < TOL['gradnorm'], rep['gradnorm_worst']
    # ... and against the EXACT model (nothing in it was fitted to the device): what bf16 operand storage costs
    assert rep['grad_worst_f64'][1] < TOL['grad_f64'], rep['grad_worst_f64']
    assert rep['gradnorm_worst_f64'][1] < TOL['gradnorm_f64'], rep['gradnorm_worst_f64']


@pytest.mark.parametrize('case', ['metricM_dense', 'metricM_ragged', 'metricM_bench', 'metricM_ragged64', 'cfg1_timit', 'metricL_ctc', 'metricL_full'])
def test_train_step_matches_the_oracle_fixture(case):
    """Forward + backward of one train step at the case's shape: logits, loss and every gradient tensor against the
    fixture.  metricM_*: the persistent decoder and the 4-row recurrent kernels must be the ones that ran."""
    from phones_las_amd import hip
    c = G.CASES[case]
    g, model, rep, _ = _run_case(case)
    if case.startswith('metricM'):
        assert hip.lib().las_lstm_slice_rows(c['B'], c['H'], 2) == 4
        assert getattr(model.speller, '_persist_ws', None) is not None and getattr(model.speller, '_persist_ws_bwd', None) is not None
    _write(rep)
    _check(rep)


@pytest.mark.parametrize('case', ['metricM_dense', 'metricM_ragged', 'cfg1_timit', 'metricL_ctc', 'cfg5_binf'])
def test_encoder_memory_and_states_match_the_fixture(case):
    """The listener alone (PREDICT graph = same kernels, no tape): every 4th frame of the encoder memory and the final
    (c, h) of the last layer's two directions; frames beyond the reduced length are exactly zero."""
    g = _fixture(case)
    model, feats, labels, nb = _model(case)
    from phones_las_amd.las.ops import PREDICT
    (mem, mem_len), state = model.listener.forward(feats['encoder_inputs'], feats['source_sequence_length'], PREDICT)
    torch.cuda.synchronize()
    model.check_device_status()
    m = mem.float().cpu().numpy()
    ref = g['bf16_memory'].astype(np.float32)
    e_mem = _rel(m[:, ::4], ref)
    ml = mem_len.cpu().numpy()
    L = G.CASES[case]['L']
    want = nb['source_sequence_length'].copy()
    for _ in range(L - 1):
        want = want // 2 + want % 2
    assert ml.tolist() == want.tolist()
    for b in range(m.shape[0]):
        if ml[b] < m.shape[1]:
            assert float(np.abs(m[b, ml[b]:]).max()) == 0.0
    e_c = max(_rel(state[d].c.cpu().numpy(), g['bf16_state_c'][d]) for d in range(2))
    e_h = max(_rel(state[d].h.cpu().numpy(), g['bf16_state_h'][d]) for d in range(2))
    print(json.dumps({'case': case, 'memory': e_mem, 'state_c': e_c, 'state_h': e_h,
                      'memory_vs_f64': _rel(m[:, ::4], g['f64_memory'].astype(np.float32))}))
    assert e_mem < TOL['memory'] and e_c < TOL['state'] and e_h < 1.6e-2, (e_mem, e_c, e_h)


def test_cfg5_real_binf_map_bahdanau_monotonic_matches_the_fixture():
    """BASELINE configs[4]: --binary_outputs --binf_projection with the reference's misc/binf_map.csv (40 x 197, as the
    reference's load_binf2phone returned it) + bahdanau_monotonic; the TRAIN-mode score noise of the fixture is the numpy
    restatement of the device generator, checked here against las_normal_fill."""
    from phones_las_amd import hip
    from phones_las_amd.las.speller_general import GeneralSpeller
    assert G.NOISE_STREAM == GeneralSpeller.NOISE_STREAM
    n = 12 * 4 * 16
    dev = torch.empty(n, dtype=torch.float32, device='cuda')
    hip.check(hip.lib().las_normal_fill(hip.p(dev), n, G.first_step_seed(), G.NOISE_STREAM, hip.stream()))
    assert float(np.abs(dev.cpu().numpy() - G.device_normal(G.first_step_seed(), G.NOISE_STREAM, n)).max()) < 2e-5
    g, model, rep, _ = _run_case('cfg5_binf')
    assert model.last_seed == G.first_step_seed() and tuple(G.binf_matrix('binf_map.csv').shape) == (40, 197)
    _write(rep)
    assert rep['logits_vs_bf16'] < TOL['logits'], rep
    assert rep['audio_loss_vs_bf16'] < 2e-3, rep
    assert rep['grad_worst'][1] < 3e-2, rep['grad_worst']         # general decoder path: d(attention) rounding points not modelled
    assert rep['gradnorm_worst'][1] < 5e-2, rep['gradnorm_worst']


def test_persistent_decoder_matches_per_step_launches_at_the_benchmarked_shape(monkeypatch):
    """dec_persist_fwd_kernel<false,true> / dec_persist_bwd_kernel<false> at Hd=256, M=1024, T'=200, U=80 (what bench.py
    times) against the per-step launches on the same inputs, and against itself.
      * two persistent runs are BIT-identical (logits and every gradient whose reduction order is fixed): no exchange
        of the launch depends on timing;
      * step 0 agrees with the per-step path to rounding: the one-launch kernel writes the step out with its own
        arithmetic (v_dot2c scores, per-wave softmax partials, DPP sums), so fp32 values differ in the last bits and a
        context element may round to the neighbouring bf16 (one such flip is ~1e-4 of the largest logit);
      * later steps: a 1-ulp fp32 difference flips a bf16 rounding of h_t / context now and then, and the flip feeds the
        next step -- over 80 steps the two paths drift apart by a few 1e-4 of the largest logit (measured 4.7e-4), both
        inside the 2e-2 tolerance against the fixture."""
    model, feats, labels, nb = _model('metricM_ragged')
    g = _fixture('metricM_ragged')
    outs = {}
    for flag in ('1', '1b', '0'):
        monkeypatch.setenv('LAS_DEC_PERSIST', flag[0])
        model.vars.grad.zero_()
        model.speller._persist_ws = model.speller._persist_ws_bwd = None
        loss, logits, dlogits = model.forward_train(feats, labels)
        model.backward(dlogits)
        torch.cuda.synchronize()
        model.check_device_status()
        assert (model.speller._persist_ws is not None) == (flag[0] == '1')
        outs[flag] = (float(loss), logits.clone(), {n: t.clone() for n, t in model.vars.grads.items()})
    assert torch.equal(outs['1'][1], outs['1b'][1])                       # run-to-run: bit-identical logits
    run_to_run = max(float((outs['1'][2][n] - outs['1b'][2][n]).abs().max() / (outs['1'][2][n].abs().max() + 1e-30)) for n in outs['1'][2])
    scale = float(outs['0'][1].abs().max())
    d_step0 = float((outs['1'][1][:, 0] - outs['0'][1][:, 0]).abs().max()) / scale
    d_logits = float((outs['1'][1] - outs['0'][1]).abs().max()) / scale
    worst = max((float((outs['1'][2][n] - outs['0'][2][n]).abs().max() / (outs['0'][2][n].abs().max() + 1e-30)), n)
                for n in outs['1'][2])
    print(json.dumps({'persist_run_to_run_grad': run_to_run, 'persist_vs_per_step_logits_step0': d_step0,
                      'persist_vs_per_step_logits': d_logits, 'persist_vs_per_step_grad_worst': list(worst)}))
    assert run_to_run < 1e-5          # (split-K atomics of the speller's weight-gradient products: fp32 summation order)
    assert d_step0 < 5e-4
    assert d_logits < 2e-3
    assert worst[0] < 5e-3, worst
    V, tl, ref = 64, nb['target_sequence_length'], g['bf16_logits']
    for flag in ('1', '0'):
        lg = outs[flag][1][..., :V].float().cpu().numpy()
        err = max(float(np.abs(lg[b, :tl[b]].astype(np.float64) - ref[b, :tl[b]]).max()) for b in range(len(tl)))
        assert err / float(np.abs(ref).max()) < TOL['logits']


def test_cfg5_at_its_stated_size_matches_the_fixture():
    """BASELINE configs[4] at T=800 / U=80 (T'=200 frames): the monotonic 'parallel' chain p * cp * cumsum(prev / clip(cp,
    1e-10, 1)) runs far past the frame where the exclusive cumprod drops under its clip, the attention mass of late decoder
    steps shrinks to 1e-8 (as the reference's formula makes it: SURVEY A.6), and the normaliser's backward scans cross
    waves.  Round 2's only run at this size ended in NaN (a missing barrier in front of the right-to-left scans, invisible
    at T' <= 64); the T=64 fixture could not see it.  Gradients must be finite and match the oracle."""
    g, model, rep, _ = _run_case('cfg5_full')
    _write(rep)
    for n, t in model.vars.grads.items():
        assert bool(torch.isfinite(t).all()), n
    assert rep['logits_vs_bf16'] < TOL['logits'] and rep['logits_vs_f64'] < TOL['logits'], rep
    assert rep['audio_loss_vs_bf16'] < 2e-3 and rep['audio_loss_vs_f64'] < 2e-3, rep
    # Gradients are asserted against the EXACT (f64) model here: 3e-2 of the per-tensor max-abs, norms within 3 % (measured
    # 1.4e-2 / 9.6e-3).  The oracle's bf16 model is reported, not asserted: d/dc of prev / clip(c, 1e-10, 1) is -prev / c^2 (up
    # to 1e20 prev) just inside the clip and 0 outside, so the gradient of the reference's formula is DISCONTINUOUS where the
    # exclusive cumprod crosses 1e-10 under attention mass; in this fixture utterance 0 crosses at decoder step 31 and the
    # bf16 model's forward roundings put it on the other side of that edge than the exact model and the device (its score-path
    # gradients -- memory_layer, query_layer, attention_v, score_bias -- are 18-20 % larger from that single (utterance, step)
    # entry; every other step agrees to 1 %: measured with the oracle alone).
    # Round 4: which side of that edge the DEVICE lands on depends on its own forward roundings -- with the bottom layer's
    # input projection accumulated inside the recurrence (one more change of summation order) it moved from the exact model's
    # side (1.4e-2 / 9.6e-3 against f64) to the bf16 model's (9.4e-3 / 6.0e-3 against bf16; 0.27 against f64 on the
    # score-path tensors).  Both are restatements of the same formula, so the assertion is: EVERY tensor agrees with one of the
    # two models within that model's tolerance (bf16 model: 1e-2 / 2 %; exact model: 3e-2 / 3 %), and all tensors with the SAME
    # one (the device cannot be on both sides of the edge at once).
    per = rep['grad_per_tensor']
    fits_bf16 = all(v[0] < TOL['grad'] and v[1] < TOL['gradnorm'] for v in per.values())
    fits_f64 = all(v[2] < 3e-2 and v[3] < 3e-2 for v in per.values())
    assert fits_bf16 or fits_f64, (rep['grad_worst'], rep['gradnorm_worst'], rep['grad_worst_f64'], rep['gradnorm_worst_f64'])
    # ... and an absolute backstop on BOTH models whichever side the device is on: the edge entry is worth 0.27
    # of a score-path tensor's max-abs against the model on the other side (measured, above); nothing may be further than 0.4
    # from either.  `side` is written into the report so that a flip from one run to the next is seen.
    rep['grad_side'] = 'bf16' if fits_bf16 else 'f64'
    _write(rep)
    assert all(v[0] < 0.4 and v[2] < 0.4 for v in per.values()), (rep['grad_worst'], rep['grad_worst_f64'])


@pytest.mark.parametrize('case', ['dec512_groups', 'dec512_groups_luong'])
def test_512_unit_one_launch_decoders_over_several_groups(case):
    """The written-out 512-unit decoders metric-L / cfg3 / cfg4 time (dec_persist_fwd_lean_kernel<att, 4, 8, 20>,
    dec_persist_bwd_kernel<wq, 16, ...>: streamed K chunks, column-split h Wq with the granule all-to-all) at the benchmarked
    decoder shape M=2048, T'=100, U=80 on TWO groups of 8 utterances with ragged lengths, against the oracle fixture."""
    g, model, rep, _ = _run_case(case)
    assert getattr(model.speller, '_persist_ws', None) is not None and getattr(model.speller, '_persist_ws_bwd', None) is not None
    _write(rep)
    _check(rep)


@pytest.mark.parametrize('parts', ['4', '1'], ids=['four_workgroups', 'one_workgroup'])
@pytest.mark.parametrize('case', ['cfg5_binf', 'cfg5_full'])
def test_one_launch_decoder_with_attention_layer_and_monotonic_normaliser_matches_the_step_launches(case, parts, monkeypatch):
    """cfg5's decoder (attention layer of 2 * binf_count outputs + bahdanau_monotonic) forward in ONE launch
    (dec_persist_fwd_kernel<..., AL>: the monotonic chain inside the shared step body, the attention layer as a 16-column-tile
    product on the group's members behind one more group barrier) against the same model on the per-step launches
    (LAS_DEC_PERSIST_AL=0): same noise stream, so the same numbers up to bf16 flips of h / context / attention (the frame
    split of the score phase changes the fp32 summation order of nothing, but the attention layer's K chunks are summed per
    wave): logits within 2e-3 of the largest, saved p_choose / alignments within 1e-3, gradients within 1e-2 of each tensor's
    largest.  And both against the fixture (the case's own test)."""
    # parts: the one-launch backward with four workgroups per utterance (three take quarters of the query path and of the two
    # products off the one that walks the chain; the default where it fits) or with one (LAS_DEC_SEQ_PARTS=1: d(keys) of all
    # frames in one workgroup's registers)
    monkeypatch.setenv('LAS_DEC_SEQ_PARTS', parts)
    outs = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('LAS_DEC_PERSIST_AL', flag)
        monkeypatch.setenv('LAS_DEC_SEQ_BWD', flag)         # ... and the backward: one launch (one workgroup per utterance) or step by step
        model, feats, labels, nb = _model(case)
        model.vars.grad.zero_()
        loss, logits, dlogits = model.forward_train(feats, labels)
        took = getattr(model.speller, '_persist_ws', None) is not None
        assert took == (flag == '1')
        sv = model.speller.saved
        keep = {k: sv[k].clone() for k in ('align', 'p', 'att')}
        model.backward(dlogits)
        torch.cuda.synchronize()
        model.check_device_status()
        outs[flag] = (float(loss), logits.clone(), keep, {n: t.clone() for n, t in model.vars.grads.items()})
    scale = float(outs['0'][1].abs().max())
    d_logits = float((outs['1'][1] - outs['0'][1]).abs().max()) / scale
    d_align = float((outs['1'][2]['align'] - outs['0'][2]['align']).abs().max())
    d_p = float((outs['1'][2]['p'] - outs['0'][2]['p']).abs().max())
    worst = max((float((outs['1'][3][n] - outs['0'][3][n]).abs().max() / (outs['0'][3][n].abs().max() + 1e-30)), n) for n in outs['1'][3])
    print(json.dumps({'case': case, 'one_launch_vs_steps_logits': d_logits, 'align': d_align, 'p_choose': d_p, 'grad_worst': list(worst),
                      'loss': [outs['1'][0], outs['0'][0]]}))
    assert d_logits < 2e-3 and d_align < 1e-3 and d_p < 1e-3, (d_logits, d_align, d_p)
    assert worst[0] < 1e-2, worst
