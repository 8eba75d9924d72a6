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
        frontend.calculate_acoustic_features(argparse.Namespace(feature_type='mfcc', backend='speechpy'), np.zeros(4000))
