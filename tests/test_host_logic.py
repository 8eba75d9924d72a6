"""Host-side logic that needs no GPU: hyper-parameter files, vocabulary, edit distance, the variable table and
the data-parallel helpers."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from oracle import las_oracle as O
from tests.helpers import make_hparams


def test_param_table_matches_oracle_for_several_models():
    from phones_las_amd import model_helper as mh
    for kw in (dict(), dict(att='bahdanau', L=3, H=128), dict(unidirectional=True, L=1), dict(dec_layers=2, bottom_only=False, pass_hidden=False)):
        ohp, params = make_hparams(**kw)
        assert mh.param_table(params) == [(n, tuple(s), i) for n, s, i in O.param_table(ohp)]


def test_variables_layout_and_init_match_oracle_seed():
    from phones_las_amd import model_helper as mh
    ohp, params = make_hparams()
    v = mh.Variables(mh.param_table(params), device='cpu')
    v.initialize(4321)
    ref = O.init_params(ohp, 4321)
    for n in ref:
        assert torch.equal(v.params[n].double(), ref[n]), n
    assert all(o % 4 == 0 for o in v.offsets)
    assert v.params['listener/bilstm_0/fw/lstm_cell/kernel'].data_ptr() == v.flat.data_ptr()
    assert v.num_parameters() == sum(int(np.prod(s)) for _, s, _ in O.param_table(ohp))


def test_hparams_json_is_double_encoded_and_existing_file_wins(tmp_path):
    from phones_las_amd.utils import params_utils as pu
    d = str(tmp_path / 'model')
    args = argparse.Namespace(model_dir=d, encoder_units=256, learning_rate=1e-3, reset=False)
    p = pu.create_hparams(args, target_vocab_size=64)
    assert p.encoder.num_units == 256 and p.decoder.target_vocab_size == 64 and p.learning_rate == 1e-3
    raw = json.load(open(os.path.join(d, 'hparams.json')))
    assert isinstance(raw, str) and json.loads(raw)['encoder_units'] == 256          # JSON string holding JSON
    # rerun with a different learning rate: the saved file wins (reference quirk B6) ...
    args2 = argparse.Namespace(model_dir=d, encoder_units=128, learning_rate=5e-4, reset=False)
    p2 = pu.create_hparams(args2)
    assert p2.encoder.num_units == 256 and p2.learning_rate == 1e-3
    # ... unless --reset
    args3 = argparse.Namespace(model_dir=d, encoder_units=128, learning_rate=5e-4, reset=True)
    p3 = pu.create_hparams(args3, target_vocab_size=64)
    assert p3.encoder.num_units == 128 and p3.learning_rate == 5e-4
    with pytest.raises(ValueError):
        pu.create_hparams(argparse.Namespace(model_dir=str(tmp_path / 'new'), reset=False))


def test_vocab_ids_start_after_specials(tmp_path):
    from phones_las_amd.utils import vocab_utils as vu
    f = tmp_path / 'vocab.txt'
    f.write_text('aa\nb\nə\n', encoding='utf-8')
    assert vu.load_vocab(str(f)) == ['<unk>', '<s>', '</s>', 'aa', 'b', 'ə']
    t = vu.create_vocab_table(str(f))
    assert t.lookup(['b', 'zz', '</s>']) == [4, 0, 2]


def test_edit_distance_matches_oracle_on_random_rows():
    from phones_las_amd.utils import metrics_utils as mu
    rng = np.random.default_rng(0)
    hyp = rng.integers(-1, 6, size=(50, 9)).tolist()
    tru = rng.integers(2, 6, size=(50, 7)).tolist()
    assert mu.edit_distance(hyp, tru, 2) == O.edit_distance(hyp, tru, 2)
    mapping = [0, 1, 2, 3, 3, 4]
    hyp = [[max(v, 0) for v in r] for r in hyp]
    assert mu.edit_distance(hyp, tru, 2, mapping) == O.edit_distance(hyp, tru, 2, mapping)


def test_shard_batch_splits_evenly():
    from phones_las_amd import dp
    x = {'a': torch.arange(12).reshape(6, 2), 'b': torch.arange(6)}
    s1 = dp.shard_batch(x, 1, 3)
    assert s1['a'].tolist() == [[4, 5], [6, 7]] and s1['b'].tolist() == [2, 3]
    with pytest.raises(ValueError):
        dp.shard_batch(x, 0, 4)


def test_gather_tree_matches_oracle_and_hand_example():
    """BeamSearchDecoder.finalize's gather_tree: product host routine vs the oracle's, plus a hand-checked case."""
    import numpy as np
    import torch
    from oracle import las_oracle as O
    from phones_las_amd.las.speller_general import gather_tree
    # T=3, B=1, K=2; end_token 9.  beam 0 at t=2 came from beam 1 at t=1, which came from beam 0 at t=0
    step = np.array([[[1, 2]], [[3, 4]], [[5, 9]]])
    par = np.array([[[0, 0]], [[0, 0]], [[1, 0]]])
    out = gather_tree(step, par, np.array([3]), 9)
    assert out[:, 0, 0].tolist() == [1, 4, 5]
    assert out[:, 0, 1].tolist() == [1, 3, 9]
    rng = np.random.default_rng(0)
    step = rng.integers(0, 6, size=(7, 3, 4))
    par = rng.integers(0, 4, size=(7, 3, 4))
    ml = np.array([7, 4, 0])
    ref = O.gather_tree(torch.from_numpy(step), torch.from_numpy(par), torch.from_numpy(ml), 2).numpy()
    assert (gather_tree(step, par, ml, 2) == ref).all()
    assert (ref[4:, 1] == 2).all() and (ref[:, 2] == 2).all()


def test_oracle_beam_width_one_is_greedy():
    import torch
    from oracle import las_oracle as O
    from tests.helpers import make_hparams, make_batch
    ohp, _ = make_hparams(att='luong')
    op = O.init_params(ohp, bias_scale=0.1)
    g = torch.Generator().manual_seed(3)
    op['speller/projection_layer/bias'] = torch.randn(11, generator=g, dtype=torch.float64) * 1.5
    b = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    (mem, ml), st = O.listener(b['encoder_inputs'], b['source_sequence_length'], op, ohp.encoder, 'f64')
    rl, rids, rfl, _ = O.speller_greedy(ohp, op, mem, ml, st, 'f64')
    p1, s1, l1 = O.speller_beam(ohp, op, mem, ml, st, 1, 'f64')
    T = p1.shape[1]
    for i in range(3):
        n = int(rfl[i])
        assert p1[i, :n, 0].tolist() == rids[i, :n].tolist()
        assert int(l1[i, 0]) == n


def test_oracle_beam_from_partial_targets_continues_the_teacher_forced_state():
    """speller_beam(partial_targets=...) (las/model.py:299-307,351-361): with beam width 1 the search is greedy decoding
    from the state a teacher-forced pass over the prefix leaves behind, fed partial_targets[:, 0] first."""
    import torch
    from oracle import las_oracle as O
    from tests.helpers import make_hparams, make_batch
    ohp, _ = make_hparams(att='luong')
    op = O.init_params(ohp, bias_scale=0.1)
    g = torch.Generator().manual_seed(5)
    op['speller/projection_layer/bias'] = torch.randn(11, generator=g, dtype=torch.float64) * 1.5
    b = make_batch(src_len=[12, 7, 10], tgt_len=[6, 4, 5])
    (mem, ml), st = O.listener(b['encoder_inputs'], b['source_sequence_length'], op, ohp.encoder, 'f64')
    partial = torch.tensor([[1, 5, 7], [1, 3, 3], [1, 9, 4]])
    p1, _, l1 = O.speller_beam(ohp, op, mem, ml, st, 1, 'f64', partial_targets=partial)
    sp = O.Speller(ohp, op, mem, ml, st, 'f64', False)
    for t in range(partial.shape[1]):
        sp.step(sp.embed(partial[:, t]))
    ids = partial[:, 0]
    done = torch.zeros(3, dtype=torch.bool)
    for t in range(p1.shape[1]):
        ids = sp.step(sp.embed(ids)).argmax(-1)
        for i in range(3):
            if not done[i]:
                assert int(p1[i, t, 0]) == int(ids[i]), (i, t)
        done |= ids == ohp.decoder.eos_id


def test_exchange_buckets_partition_the_flat_buffer():
    """Variables.split_buckets (the two exchange buckets of the overlapped data-parallel step): contiguous pieces of the
    flat buffers in the order the backward pass completes them, boundaries relative to each piece, 16-byte aligned."""
    from phones_las_amd import model_helper as mh
    table = [('a', (5, 3), 'zeros'), ('b', (7,), 'zeros'), ('c', (4, 4), 'zeros'), ('d', (1,), 'zeros'), ('e', (9, 2), 'zeros')]
    v = mh.Variables(table, device='cpu')
    assert len(v.buckets) == 1 and v.buckets[0]['begin'] == 0 and v.buckets[0]['end'] == v.total
    b0, b1 = v.split_buckets(3)
    assert (b0['lo'], b0['hi'], b1['lo'], b1['hi']) == (3, 5, 0, 3)
    assert b1['begin'] == 0 and b1['end'] == b0['begin'] and b0['end'] == v.total
    assert b0['begin'] % 4 == 0 and (b0['slot'], b1['slot']) == (0, 1)
    assert b0['seg'].tolist() == [0, v.offsets[4] - v.offsets[3], v.total - v.offsets[3]]
    assert b1['seg'].tolist() == [0, v.offsets[1], v.offsets[2], v.offsets[3]]
    # the views of the tensors tile the pieces exactly
    assert v.grads['d'].data_ptr() == v.grad.data_ptr() + 4 * v.offsets[3]
    # a split at either end keeps the single bucket
    w = mh.Variables(table, device='cpu')
    assert len(w.split_buckets(0)) == 1 and len(w.split_buckets(5)) == 1


def test_tf_variable_name_map_listener_names_follow_survey_a1():
    """`listener/bilstm_{l}/bidirectional_rnn/{fw,bw}/lstm_cell/{kernel,bias}` are the names the reference's graph gives
    the pyramidal listener (`las/ops.py:23-46`); the stacked listener and the unidirectional forms by the same TF scoping rules."""
    from phones_las_amd import model_helper as mh
    from tests.helpers import make_hparams
    _, p = make_hparams(L=3, H=16)
    m = mh.tf_variable_name_map(p)
    assert set(m) == {n for n, _, _ in mh.param_table(p)}                    # every variable, nothing else
    assert len(set(m.values())) == len(m)                                     # one-to-one
    for l in range(3):
        for d in ('fw', 'bw'):
            for x in ('kernel', 'bias'):
                assert m['listener/bilstm_%d/%s/lstm_cell/%s' % (l, d, x)] == 'listener/bilstm_%d/bidirectional_rnn/%s/lstm_cell/%s' % (l, d, x)
    _, p = make_hparams(L=2, H=16, unidirectional=True)
    assert mh.tf_variable_name_map(p)['listener/bilstm_1/fw/lstm_cell/kernel'] == 'listener/bilstm_1/rnn/lstm_cell/kernel'
    _, p = make_hparams(L=2, H=16, pyramidal=False)
    m = mh.tf_variable_name_map(p)
    assert m['listener/bw/multi_rnn_cell/cell_1/lstm_cell/bias'] == 'listener/bidirectional_rnn/bw/multi_rnn_cell/cell_1/lstm_cell/bias'
    _, p = make_hparams(L=2, H=16, pyramidal=False, unidirectional=True)
    assert mh.tf_variable_name_map(p)['listener/fw/multi_rnn_cell/cell_0/lstm_cell/kernel'] == 'listener/rnn/multi_rnn_cell/cell_0/lstm_cell/kernel'
    # the speller's names are derived, not verified: they are listed as such, and the listener's are not
    _, p = make_hparams(L=2, H=16, att='bahdanau', dec_layers=2, ctc=0.3)
    unv = mh.UNVERIFIED_TF_NAMES(p)
    assert unv and all(n.startswith('speller') for n in unv)
    m = mh.tf_variable_name_map(p)
    assert m['speller/memory_layer/kernel'] == 'speller/memory_layer/kernel'
    assert m['speller/attention_v'] == 'speller/decoder/attention_wrapper/bahdanau_attention/attention_v'
    assert m['ctc_logits/kernel'] == 'ctc_logits/kernel'
