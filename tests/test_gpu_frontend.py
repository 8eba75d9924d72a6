"""GPU parity of the acoustic front-end kernels against the float64 oracle (oracle/frontend_oracle.py) on seeded
synthetic audio.  Tolerance: fp32 arithmetic over 320-term DFT sums and log compression: 2e-3 absolute on dB / log
features (values span ~80 dB), 1e-4 relative elsewhere."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _audio(n=16000, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    y = 0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 2500 * t * (1 + 0.1 * t)) + 0.05 * rng.standard_normal(n)
    return y.astype(np.float32)


@pytest.mark.parametrize('feature_type,energy,deltas,F', [('mfcc', False, True, 39), ('mfcc', True, True, 42),
                                                         ('mfe', False, False, 40), ('mfe', True, True, 123)])
def test_librosa_pipeline_vs_oracle(feature_type, energy, deltas, F):
    from oracle import frontend_oracle as FO
    from phones_las_amd import frontend
    y = _audio()
    args = argparse.Namespace(feature_type=feature_type, backend='librosa', n_mfcc=13, n_mels=40, window=20, step=10,
                              energy=energy, deltas=deltas)
    got = frontend.calculate_acoustic_features(args, y).cpu().double().numpy()
    ref = FO.librosa_features(y, feature_type, 13, 40, 20, 10, energy, deltas)
    assert got.shape == ref.shape == (101, F)
    assert np.abs(got - ref).max() < 5e-3, np.abs(got - ref).max()


def test_tf_mfcc_op_vs_oracle():
    from oracle import frontend_oracle as FO
    from phones_las_amd import frontend
    y = _audio(8000, 3)
    got = frontend.calculate_mfcc_op(16000, 13, 320, 160, 40)(y).cpu().double().numpy()
    ref = FO.tf_mfcc(y, 16000, 13, 320, 160, 40)
    assert got.shape == ref.shape == (49, 13)
    assert np.abs(got - ref).max() < 2e-3


def test_short_signal_and_unsupported_backend():
    from phones_las_amd import frontend
    with pytest.raises(ValueError):
        frontend.calculate_mfcc_op(16000, 13, 320, 160, 40)(np.zeros(100, np.float32))
    with pytest.raises(ValueError):
        frontend.calculate_acoustic_features(argparse.Namespace(feature_type='mfcc', backend='kaldi'), np.zeros(4000))


@pytest.mark.parametrize('feature_type,energy,deltas,window,F', [('mfcc', False, False, 20, 13), ('mfcc', True, True, 25, 39),
                                                                ('mfe', True, False, 20, 41), ('mfe', True, True, 25, 123)])
def test_speechpy_backend_vs_oracle(feature_type, energy, deltas, window, F):
    """preprocess_all.py:73-79, 88-91, 122-123 (--backend speechpy): speechpy==2.4's mfe / mfcc / extract_derivative_feature on the
    front-end kernels against the float64 restatement (oracle/frontend_oracle.py: rectangular frames, one frame fewer than fit,
    the filterbank from 300 Hz on the lower half of the bins, log frame energy as first cepstral coefficient, differences along
    the FEATURE axis).  Natural-log features of fp32 sums over 320 / 400 terms: 2e-3 absolute."""
    from oracle import frontend_oracle as FO
    from phones_las_amd import frontend
    y = _audio()
    args = argparse.Namespace(feature_type=feature_type, backend='speechpy', n_mfcc=13, n_mels=40, window=window, step=10,
                              energy=energy, deltas=deltas)
    got = frontend.calculate_acoustic_features(args, y).cpu().double().numpy()
    ref = FO.speechpy_features(y, feature_type, 13, 40, window, 10, energy, deltas)
    T = (16000 - window * 16) // 160
    assert got.shape == ref.shape == (T, F)
    assert np.abs(got - ref).max() < 2e-3, np.abs(got - ref).max()
    # the batched entry point goes utterance by utterance for this backend: the same tensors
    both = frontend.calculate_acoustic_features_batch(args, [y, y[:9000]])
    assert torch.equal(both[0], frontend.calculate_acoustic_features(args, y)) and both[1].shape[0] == (9000 - window * 16) // 160


def test_speechpy_mfe_without_energy_dies_as_the_reference_does():
    """preprocess_all.py:77-79 assigns `acoustic_features` under --energy only: UnboundLocalError without it, in the reference and here."""
    from phones_las_amd import frontend
    args = argparse.Namespace(feature_type='mfe', backend='speechpy', n_mfcc=13, n_mels=40, window=20, step=10, energy=False,
                              deltas=False)
    with pytest.raises(UnboundLocalError):
        frontend.calculate_acoustic_features(args, _audio())


def test_preprocess_all_cli_end_to_end(tmp_path):
    import wave
    import preprocess_all
    from phones_las_amd.utils import tfrecord, load_vocab, load_normalization
    d = str(tmp_path)
    lines = []
    for i in range(3):
        y = (_audio(8000 + 1600 * i, i) * 32767 * 0.5).astype('<i2')
        p = '%s/a%d.wav' % (d, i)
        with wave.open(p, 'wb') as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(y.tobytes())
        lines.append('%s,arpabet,%s' % (p, ['hh ah l ow', 'w er l d', 'ah ah'][i]))
    lines.append('%s/missing.wav,arpabet,x y' % d)                      # skipped like the reference does
    open(d + '/list.csv', 'w').write('\n'.join(lines) + '\n')
    n = preprocess_all.main(preprocess_all.parse_args(['--input_file', d + '/list.csv', '--output_file', d + '/train.tfr',
                                                       '--targets', 'phones', '--deltas', '--energy', '--save_norm',
                                                       '--save_vocab']))
    assert n == 3
    recs = [tfrecord.parse_sequence_example(r, 42) for r in tfrecord.tf_record_iterator(d + '/train.tfr', verify=True)]
    assert [x.shape for x, _ in recs] == [(51, 42), (61, 42), (71, 42)]
    assert recs[0][1] == ['hh', 'ah', 'l', 'ow']
    assert load_vocab(d + '/vocab.txt')[3] == 'ah'                      # most common first
    m, s = load_normalization(d + '/norm.dmp')
    assert m.shape == (42,) and np.allclose(m, np.mean([x.mean(0) for x, _ in recs], 0), atol=1e-4)   # quirk B5
    # --backend speechpy through the same CLI (preprocess_all.py:88-91, 122-123, 228-230): 13 cepstra x 3, one frame fewer than fit
    n = preprocess_all.main(preprocess_all.parse_args(['--input_file', d + '/list.csv', '--output_file', d + '/sp.tfr', '--targets', 'phones',
                                                       '--deltas', '--backend', 'speechpy', '--n_jobs', '4', '--save_norm']))
    assert n == 3                                                       # (the count of utterances in the norm statistics)
    recs = [tfrecord.parse_sequence_example(r, 39) for r in tfrecord.tf_record_iterator(d + '/sp.tfr', verify=True)]
    assert [x.shape for x, _ in recs] == [((8000 + 1600 * i - 320) // 160, 39) for i in range(3)]
    from oracle import frontend_oracle as FO
    y0 = (_audio(8000, 0) * 32767 * 0.5).astype('<i2').astype(np.float32) / 32768.0
    assert np.abs(recs[0][0] - FO.speechpy_features(y0, 'mfcc', 13, 40, 20, 10, False, True)).max() < 2e-3


@pytest.mark.parametrize('feature_type,energy,deltas', [('mfcc', True, True), ('mfcc', False, False), ('mfe', True, True), ('mfe', False, True)])
def test_batched_front_end_is_bit_identical_to_the_per_utterance_path(feature_type, energy, deltas):
    """calculate_acoustic_features_batch (three launches over the frames of ALL utterances: las_fe_batch_melspec / _finish /
    _delta) against calculate_acoustic_features utterance by utterance: reflect padding at every signal's own ends, the
    top_db floor under every utterance's own maximum and the Savitzky-Golay edge windows are per utterance, and a frame's
    arithmetic runs in the same order -- torch.equal, not a tolerance.  Ragged lengths, one utterance much louder than the
    others (its maximum must not floor its neighbours)."""
    from phones_las_amd import frontend
    args = argparse.Namespace(feature_type=feature_type, backend='librosa', n_mfcc=13, n_mels=40, window=20, step=10,
                              energy=energy, deltas=deltas)
    waves = [_audio(16000, 0), 30.0 * _audio(4321, 1), 1e-3 * _audio(9999, 2), _audio(1600, 3), _audio(24000, 4)]
    got = frontend.calculate_acoustic_features_batch(args, waves)
    assert len(got) == 5
    for w, g in zip(waves, got):
        ref = frontend.calculate_acoustic_features(args, w)
        assert g.shape == ref.shape and torch.equal(g, ref)
    assert frontend.calculate_acoustic_features_batch(args, []) == []
