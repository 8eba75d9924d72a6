"""Known-answer and cross-implementation checks of the oracle's speller, losses, metric, optimiser."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import las_oracle as O

DT = torch.float64


def _hp(att='luong', dec_layers=1, bottom_only=True, pass_hidden=True, V=11, H=8, L=2, F_=5, als=None,
        emb=0):
    return O.HP(encoder=O.EncoderHP(num_layers=L, num_units=H), num_channels=F_,
                decoder=O.DecoderHP(num_layers=dec_layers, num_units=H, target_vocab_size=V,
                                    attention_type=att, bottom_only=bottom_only,
                                    pass_hidden_state=pass_hidden, attention_layer_size=als,
                                    embedding_size=emb))


def _batch(B=3, T=12, F_=5, V=11, U=6, seed=0):
    b = O.synthetic_batch(B, T, F_, V, U, ragged=False, seed=seed)
    b['source_sequence_length'] = torch.tensor([T, T - 5, T - 2][:B])
    b['target_sequence_length'] = torch.tensor([U, U - 2, U - 1][:B])
    return b


def test_zero_keys_give_uniform_attention_and_ln_v_loss():
    hp = _hp()
    p = O.init_params(hp)
    p['speller/memory_layer/kernel'].zero_()
    p['speller/projection_layer/kernel'].zero_()
    b = _batch()
    (mem, ml), st = O.listener(b['encoder_inputs'], b['source_sequence_length'], p, hp.encoder)
    logits, sp = O.speller_train(hp, p, mem, ml, st, b['targets_inputs'], b['target_sequence_length'])
    for bi in range(3):
        n = int(ml[bi])
        a = sp.align_hist[0][bi]
        assert torch.allclose(a[:n], torch.full((n,), 1.0 / n, dtype=DT))
        assert float(a[n:].abs().sum()) == 0.0
        # wrapper output = context (attention_layer_size None): mean of valid frames
        assert torch.allclose(sp.attention[bi], mem[bi, :n].mean(0), atol=1e-12)
    loss = O.compute_loss_train(logits, b['targets_outputs'], b['target_sequence_length'])
    assert abs(float(loss) - math.log(11)) < 1e-12


def test_sequence_loss_matches_torch_cross_entropy():
    torch.manual_seed(0)
    B, U, V = 4, 7, 9
    logits = torch.randn(B, U, V, dtype=DT)
    tg = torch.randint(0, V, (B, U))
    ln = torch.tensor([7, 3, 1, 5])
    ours = O.compute_loss_train(logits, tg, ln)
    w = (torch.arange(U)[None] < ln[:, None])
    ref = F.cross_entropy(logits[w], tg[w], reduction='sum') / w.sum()
    assert abs(float(ours - ref)) < 1e-12


def test_pass_hidden_state_uses_encoder_fw_then_bw():
    hp = _hp(dec_layers=2)
    p = O.init_params(hp)
    b = _batch()
    (mem, ml), st = O.listener(b['encoder_inputs'], b['source_sequence_length'], p, hp.encoder)
    sp = O.Speller(hp, p, mem, ml, st)
    assert torch.equal(sp.cells[0][0], st[0][0]) and torch.equal(sp.cells[1][1], st[1][1])
    hp2 = _hp(dec_layers=2, bottom_only=False)
    sp2 = O.Speller(hp2, O.init_params(hp2), mem, ml, st)
    assert float(sp2.cells[0][0].abs().max()) == 0.0      # silently ignored (las/model.py:260)


def test_all_attention_types_run_and_normalise():
    for att in ('luong', 'bahdanau', 'luong_monotonic', 'bahdanau_monotonic', 'custom'):
        hp = _hp(att=att, als=6)
        p = O.init_params(hp)
        b = _batch()
        (mem, ml), st = O.listener(b['encoder_inputs'], b['source_sequence_length'], p, hp.encoder)
        logits, sp = O.speller_train(hp, p, mem, ml, st, b['targets_inputs'], b['target_sequence_length'])
        assert logits.shape == (3, 6, 11)
        a = sp.align_hist[-1]
        for bi in range(3):
            assert float(a[bi, int(ml[bi]):].abs().sum()) == 0.0
        if 'monotonic' not in att:
            assert torch.allclose(a.sum(-1), torch.ones(3, dtype=DT))
        else:
            assert bool((a.sum(-1) <= 1 + 1e-9).all())


def test_monotonic_parallel_known_answer():
    # p = 1 everywhere and previous attention one-hot at 0 => stays at 0
    p = torch.ones(1, 4, dtype=DT)
    prev = torch.tensor([[1.0, 0, 0, 0]], dtype=DT)
    a = O.monotonic_attention(p, prev, 'parallel')
    assert torch.allclose(a, prev)
    # p = [0, 1, ...]: moves to position 1
    a = O.monotonic_attention(torch.tensor([[0.0, 1, 1, 1]], dtype=DT), prev, 'parallel')
    assert torch.allclose(a, torch.tensor([[0.0, 1, 0, 0]], dtype=DT), atol=1e-9)
    a = O.monotonic_attention(torch.tensor([[0.0, 1, 1, 1]], dtype=DT), prev, 'hard')
    assert torch.allclose(a, torch.tensor([[0.0, 1, 0, 0]], dtype=DT))


def test_greedy_stops_and_lengths():
    hp = _hp()
    p = O.init_params(hp)
    # force EOS at the first step
    p['speller/projection_layer/bias'][O.EOS_ID] = 50.0
    b = _batch()
    (mem, ml), st = O.listener(b['encoder_inputs'], b['source_sequence_length'], p, hp.encoder)
    logits, ids, fl, _ = O.speller_greedy(hp, p, mem, ml, st)
    assert logits.shape[1] == 1 and fl.tolist() == [1, 1, 1] and ids[:, 0].tolist() == [2, 2, 2]
    # never EOS: runs to round(max(len'))
    p['speller/projection_layer/bias'][O.EOS_ID] = -50.0
    logits, ids, fl, _ = O.speller_greedy(hp, p, mem, ml, st)
    assert logits.shape[1] == int(ml.max()) and fl.tolist() == [int(ml.max())] * 3


def test_eval_loss_pads_shorter_side():
    B, V = 2, 5
    logits = torch.zeros(B, 3, V, dtype=DT)
    tg = torch.tensor([[3, 4, 2, 2, 2], [3, 2, 2, 2, 2]])
    l = O.compute_loss_eval(logits, tg, torch.tensor([3, 1]), torch.tensor([5, 2]))
    assert abs(float(l) - math.log(V)) < 1e-12        # all-zero logits => ln V on every weighted step


def test_ctc_matches_torch():
    torch.manual_seed(3)
    B, T, C = 3, 9, 6
    logits = torch.randn(B, T, C, dtype=DT)
    labels = torch.tensor([[1, 2, 2, 3], [4, 5, 0, 0], [3, 0, 0, 0]])
    ll = torch.tensor([4, 2, 1])
    tl = torch.tensor([9, 7, 4])
    ours = O.ctc_loss_dense(logits, labels, ll, tl, blank=0)
    ref = F.ctc_loss(torch.log_softmax(logits, -1).transpose(0, 1), labels, tl, ll, blank=0,
                     reduction='none')
    assert torch.allclose(ours, ref, atol=1e-9)
    assert O.ctc_greedy_decode(torch.eye(4, dtype=DT)[[0, 0, 3, 1, 1, 3, 1]][None], torch.tensor([7])) == [[0, 1, 1]]


def test_edit_distance_merge_and_trim():
    # utils/metrics_utils.py:10-23: repeats merged in hyp AND truth, cut at first EOS, -1 dropped
    assert O.dense_to_sparse_merge([5, 5, 6, 2, 7], 2) == [5, 6]
    assert O.dense_to_sparse_merge([5, -1, 6, 6], 2) == [5, 6]
    d = O.edit_distance([[5, 5, 6, 2, 2]], [[5, 7, 6, 2, 2]])
    assert abs(d[0] - 1.0 / 3.0) < 1e-12
    assert O.edit_distance([[2, 2]], [[2, 2]]) == [0.0]


def test_adam_tf_form_and_per_tensor_clip():
    p = {'a': torch.tensor([1.0, -2.0], dtype=DT)}
    m = {'a': torch.zeros(2, dtype=DT)}
    v = {'a': torch.zeros(2, dtype=DT)}
    g = {'a': torch.tensor([0.5, -0.25], dtype=DT)}
    np_, nm, nv = O.adam_apply(p, m, v, g, 1, 1e-3)
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    exp = p['a'] - lr_t * (0.1 * g['a']) / (torch.sqrt(0.001 * g['a'] ** 2) + 1e-8)
    assert torch.allclose(np_['a'], exp, atol=1e-15)
    hp = _hp()
    params = O.init_params(hp)
    out = O.train_step(hp, params, None, None, 1, _batch())
    for k, gc in out['clipped'].items():
        n = float(out['grads'][k].norm())
        assert float(gc.norm()) <= 2.0 + 1e-9
        if n <= 2.0:
            assert torch.equal(gc, out['grads'][k])


def test_param_count_matches_survey_metric_m():
    # metric-M encoder 4 806 656, decoder 1 705 024
    hp = O.HP(encoder=O.EncoderHP(num_layers=3, num_units=256), num_channels=40,
              decoder=O.DecoderHP(num_layers=1, num_units=256, target_vocab_size=64, bottom_only=True,
                                  pass_hidden_state=True))
    tab = O.param_table(hp)
    enc = sum(int(np.prod(s)) for n, s, _ in tab if n.startswith('listener'))
    dec = sum(int(np.prod(s)) for n, s, _ in tab if n.startswith('speller'))
    assert enc == 4806656 and dec == 1705024


def test_sigmoid_losses_known_answers():
    """sequence_loss_sigmoid / compute_loss_sigmoid (model_helper.py:81-130): zero logits give ln 2 whatever the targets;
    masked steps do not count; the EVAL form pads the shorter of (decoded, targets) with zeros and weighs the longer."""
    import math
    from oracle import las_oracle as O
    B, U, nf = 3, 5, 6
    g = torch.Generator().manual_seed(0)
    tg = (torch.rand(B, U, nf, generator=g) < 0.4).double()
    ln = torch.tensor([5, 2, 4])
    z = torch.zeros(B, U, nf, dtype=torch.float64)
    assert abs(float(O.compute_loss_sigmoid_train(z, tg, ln)) - math.log(2.0)) < 1e-12
    lg = torch.randn(B, U, nf, generator=g, dtype=torch.float64)
    ref = 0.0
    for b in range(B):
        for t in range(int(ln[b])):
            x, y = lg[b, t], tg[b, t]
            ref += float((torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))).mean())
    assert abs(float(O.compute_loss_sigmoid_train(lg, tg, ln)) - ref / float(ln.sum())) < 1e-12
    lg2 = lg.clone(); lg2[1, 2:] = 99.0                                # beyond the length: ignored
    assert abs(float(O.compute_loss_sigmoid_train(lg2, tg, ln)) - ref / float(ln.sum())) < 1e-12
    # EVAL: decoded 3 steps, targets up to 5: logits zero-padded, weights over max(len) = target lengths here
    fl = torch.tensor([3, 3, 1])
    ev = O.compute_loss_sigmoid_eval(lg[:, :3], tg, fl, ln)
    lgp = torch.cat([lg[:, :3], torch.zeros(B, 2, nf, dtype=torch.float64)], 1)
    w = (torch.arange(5).unsqueeze(0) < torch.maximum(ln, fl).unsqueeze(1)).double()
    assert abs(float(ev) - float(O.sequence_loss_sigmoid(lgp, tg, w))) < 1e-12


def test_binary_greedy_decode_known_answers():
    """las/model.py:320-336 (InferenceHelper): with a zero projection kernel the sample is the sign of the bias; the decode
    stops at the first step when the bias of the last (</s>) feature is positive, and runs to maximum_iterations when it is
    negative; the first input is the <s> feature vector."""
    from oracle import las_oracle as O
    nf, V = 6, 9
    binf = (torch.rand(nf, V, generator=torch.Generator().manual_seed(1)) < 0.5).float().numpy()
    hp = O.HP(encoder=O.EncoderHP(num_layers=2, num_units=8), num_channels=5,
              decoder=O.DecoderHP(num_layers=1, num_units=8, target_vocab_size=V, bottom_only=True, pass_hidden_state=True,
                                  binary_outputs=True, binf_count=nf, binf_map=binf))
    p = O.init_params(hp)
    assert p['speller/projection_layer/kernel'].shape == (32, nf) and p['speller/decoder_cell_0/lstm_cell/kernel'].shape[0] == nf + 32 + 8
    p['speller/projection_layer/kernel'] = torch.zeros_like(p['speller/projection_layer/kernel'])
    b = O.synthetic_batch(2, 8, 5, V, 4)
    (mem, ml), st = O.listener(b['encoder_inputs'], b['source_sequence_length'], p, hp.encoder)
    p['speller/projection_layer/bias'] = torch.tensor([1., -1., 1., -1., 1., 2.], dtype=torch.float64)
    lg, smp, fl, _ = O.speller_greedy_binary(hp, p, mem, ml, st)
    assert lg.shape == (2, 1, nf) and fl.tolist() == [1, 1] and smp[0, 0].tolist() == [1, 0, 1, 0, 1, 1]
    p['speller/projection_layer/bias'] = torch.tensor([1., -1., 1., -1., 1., -2.], dtype=torch.float64)
    lg, smp, fl, _ = O.speller_greedy_binary(hp, p, mem, ml, st)
    assert lg.shape[1] == int(ml.max()) and fl.tolist() == [int(ml.max())] * 2


def _simulate_monotonic(p, hard_scores=None):
    """The stochastic process monotonic attention is the expectation of (Raffel et al. 2017, what
    tf.contrib.seq2seq.monotonic_attention computes in closed form): at output step i the decoder starts at the memory
    position it attended at step i-1 (position 0 before the first step) and walks right; at position j it stops and attends
    with probability p[i, j], else moves on; walking off the end attends nothing, at this and every later step.
    BRUTE FORCE: every outcome z in {0,1}^(U x T) of the independent stop / move-on draws is enumerated, the walk is
    simulated for it, and the attended positions are weighted with the outcome's probability.  Returns alpha [U, T]."""
    U, T = p.shape
    alpha = np.zeros((U, T))
    for code in range(1 << (U * T)):
        z = np.array([(code >> k) & 1 for k in range(U * T)]).reshape(U, T)
        w = float(np.prod(np.where(z == 1, p, 1.0 - p)))
        if w == 0.0:
            continue
        pos = 0
        for i in range(U):
            j = pos
            while j < T and z[i, j] == 0:
                j += 1
            if j >= T:
                break                      # fell off the end: nothing attended from here on
            alpha[i, j] += w
            pos = j
    return alpha


def test_monotonic_attention_parallel_is_the_expectation_of_the_attend_or_skip_process():
    """Pins oracle.monotonic_attention('parallel') -- p * cumprod_excl(1-p) * cumsum(prev / clip(cumprod_excl(1-p))) chained
    over decoder steps from the one-hot initial alignment -- on an INDEPENDENT statement of what it means: the marginals of
    the stop / move-on walk, by enumeration of all 2^(U T) outcomes (U=3 steps, T'=5 frames; and T'=6 with U=2).  Also the
    'hard' mode: with p in {0,1} the walk is deterministic and the alignment one-hot (or empty)."""
    rng = np.random.default_rng(11)
    for U, T in ((3, 5), (2, 6)):
        p = rng.uniform(0.05, 0.95, size=(U, T))
        p[0, 1] = 0.0                      # a frame that is never chosen, and one that almost always stops the walk
        p[U - 1, T - 2] = 0.97
        want = _simulate_monotonic(p)
        prev = torch.zeros(1, T, dtype=DT)
        prev[0, 0] = 1.0
        for i in range(U):
            a = O.monotonic_attention(torch.tensor(p[i:i + 1], dtype=DT), prev, 'parallel')
            assert float((a[0] - torch.tensor(want[i])).abs().max()) < 1e-12, (U, T, i)
            prev = a
        assert abs(float(want.sum(1)[-1]) - float(prev.sum())) < 1e-12 and float(prev.sum()) < 1.0      # mass lost off the end
    # Where the closed form LEAVES the process -- and the reference with it, this is tf.contrib's formula: once the exclusive
    # cumprod of (1 - p) falls under the 1e-10 clip, prev / clip(cp) no longer cancels cp and the mass of walkers that START
    # beyond that point is lost (a frame with p = 1 in front of the previous position is enough).  This is the regime cfg5 at
    # T' = 200 is in from about the 30th decoder step (alignments summing to 1e-8: DESIGN.md 2); the device kernels reproduce
    # the formula, not the process.
    p = np.array([[0.5, 0.5, 0.5, 0.5], [0.3, 1.0, 0.4, 0.6]])
    want = _simulate_monotonic(p)
    prev = torch.zeros(1, 4, dtype=DT)
    prev[0, 0] = 1.0
    a0 = O.monotonic_attention(torch.tensor(p[0:1], dtype=DT), prev, 'parallel')
    a1 = O.monotonic_attention(torch.tensor(p[1:2], dtype=DT), a0, 'parallel')
    assert float((a0[0] - torch.tensor(want[0])).abs().max()) < 1e-12
    assert float((a1[0, :2] - torch.tensor(want[1, :2])).abs().max()) < 1e-12
    assert want[1, 2] > 0.04 and float(a1[0, 2]) < 1e-20 and want[1, 3] > 0.05 and float(a1[0, 3]) < 1e-20
    # hard mode = the same walk with deterministic draws
    z = (rng.uniform(size=(3, 5)) < 0.35).astype(np.float64)
    want = _simulate_monotonic(z)
    prev = torch.zeros(1, 5, dtype=DT)
    prev[0, 0] = 1.0
    for i in range(3):
        a = O.monotonic_attention(torch.tensor(z[i:i + 1], dtype=DT), prev, 'hard')
        assert torch.equal(a[0], torch.tensor(want[i])), i
        prev = a
