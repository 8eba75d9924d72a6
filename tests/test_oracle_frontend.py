"""Pin the front-end oracle's building blocks against scipy / torch (librosa and TF are not installed)."""
import numpy as np
import scipy.fft
import scipy.signal
import torch

from oracle import frontend_oracle as FO


def test_window_stft_dct_match_scipy_and_torch():
    rng = np.random.default_rng(0)
    y = rng.standard_normal(4000)
    assert np.allclose(FO.hann_periodic(320), scipy.signal.get_window('hann', 320, fftbins=True))
    S = FO.stft_mag(y, 320, 160, center=True, power=1.0)
    ref = torch.stft(torch.tensor(y), 320, 160, window=torch.hann_window(320, periodic=True, dtype=torch.float64),
                     center=True, pad_mode='reflect', return_complex=True).abs().T.numpy()
    assert S.shape == ref.shape == (1 + 4000 // 160, 161)
    assert np.allclose(S, ref, atol=1e-9)
    S2 = FO.stft_mag(y, 320, 160, center=False, power=1.0)
    assert S2.shape[0] == 1 + (4000 - 320) // 160
    x = rng.standard_normal((7, 40))
    assert np.allclose(x @ FO.dct2_matrix(13, 40).T, scipy.fft.dct(x, type=2, norm='ortho', axis=1)[:, :13])
    # TF's mfccs_from_log_mel_spectrograms = unnormalised DCT-II * (2N)^-1/2: equals ortho for k>=1, sqrt(2) larger at k=0
    tfm = x @ (FO.dct2_matrix(40, 40, ortho=False) * np.sqrt(1 / 80.0)).T
    orth = scipy.fft.dct(x, type=2, norm='ortho', axis=1)
    assert np.allclose(tfm[:, 1:], orth[:, 1:]) and np.allclose(tfm[:, 0], orth[:, 0] * np.sqrt(2))


def test_delta_is_savgol_interp():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((40, 5))
    for order in (1, 2):
        ref = scipy.signal.savgol_filter(x, 9, polyorder=order, deriv=order, axis=0, mode='interp')
        assert np.allclose(FO.delta(x, 9, order), ref, atol=1e-10), order


def test_mel_filterbanks_basic_properties():
    w = FO.mel_htk_matrix(40, 161, 16000, 80.0, 7600.0)
    assert w.shape == (161, 40) and float(w[0].sum()) == 0.0 and float(w.max()) <= 1.0 + 1e-12
    assert (w.sum(0) > 0).all()
    m = FO.mel_slaney_matrix(40, 320, 16000)
    assert m.shape == (40, 161) and (m >= 0).all()
    # Slaney area normalisation: each filter integrates to ~1 over frequency (bin spacing 50 Hz)
    assert np.allclose(m.sum(1) * 50.0, 1.0, atol=0.25)
    assert abs(FO._mel_to_hz_slaney(FO._hz_to_mel_slaney(3000.0)) - 3000.0) < 1e-9


def test_pipelines_shapes_and_db_clipping():
    rng = np.random.default_rng(2)
    y = (rng.standard_normal(16000) * 0.1).astype(np.float32)
    f = FO.librosa_features(y, 'mfcc', energy=True, deltas=True)
    assert f.shape == (101, 42)                                       # (13 + energy) * 3, frames = 1 + N // hop
    base = FO.librosa_features(y, 'mfcc', energy=True, deltas=False)
    assert np.allclose(f[:, 0::3], base)                              # interleaved layout [c, dc, ddc]
    mfe = FO.librosa_features(y, 'mfe', n_mels=40)
    assert mfe.shape == (101, 40) and mfe.max() - mfe.min() <= 80.0 + 1e-9
    t = FO.tf_mfcc(y)
    assert t.shape == (99, 13) and np.isfinite(t).all()
    tone = np.sin(2 * np.pi * 1000 * np.arange(16000) / 16000.0)      # a 1 kHz tone peaks in the mel band holding 1 kHz
    mel = FO.stft_mag(tone, 320, 160, True, 2.0) @ FO.mel_slaney_matrix(40, 320, 16000).T
    peak = int(mel.mean(0).argmax())
    centers = FO._mel_to_hz_slaney(np.linspace(FO._hz_to_mel_slaney(0), FO._hz_to_mel_slaney(8000), 42))[1:-1]
    assert abs(centers[peak] - 1000.0) < 120.0


def test_speechpy_restatement_building_blocks():
    """speechpy==2.4 is not installed (PARITY UNPINNED): the restatement is checked against what its published functions are
    defined to do, computed another way -- scipy's rfft / dct, an explicit loop over frames, the triangle definition -- and the
    quirks oracle/frontend_oracle.py says it keeps are asserted as such, so that a later 'fix' of one of them fails here."""
    rng = np.random.default_rng(4)
    y = rng.standard_normal(5000) * 0.1
    fr = FO.speechpy_stack_frames(y, 16000, 0.025, 0.010)
    assert fr.shape == ((5000 - 400) // 160, 400)                     # floor((N - L) / stride): one frame fewer than fit
    assert all(np.array_equal(fr[i], y[160 * i:160 * i + 400]) for i in (0, 7, fr.shape[0] - 1))
    bank = FO.speechpy_filterbanks(40, 201, 16000, 0, 8000)
    assert bank.shape == (40, 201) and bank.min() >= 0.0 and bank.max() <= 1.0
    lo = int(np.floor(202 * 300.0 / 16000))                           # `low_freq or 300`: the bank starts at 300 Hz ...
    assert not bank[:, :lo].any() and bank[0, lo + 1:].any()
    assert not bank[:, 102:].any() and bank[-1, 95:100].all()         # ... and ends at bin (coefficients + 1) / 2: half of the spectrum
    spec, en = FO.speechpy_mfe(y, 16000, 0.025, 0.010, 40, 400)
    P = np.abs(scipy.fft.rfft(fr, n=400, axis=1)) ** 2 / 400
    assert np.allclose(en, P.sum(1)) and np.allclose(spec, P @ bank.T)
    c = FO.speechpy_mfcc(y, 16000, 0.025, 0.010, 13, 40, 400)
    full = scipy.fft.dct(np.log(spec), type=2, norm='ortho', axis=1)
    assert np.allclose(c[:, 1:], full[:, 1:13]) and np.allclose(c[:, 0], np.log(en))      # dc_elimination
    silent = FO.speechpy_mfcc(np.zeros(2000), 16000, 0.02, 0.01, 13, 40, 320)
    assert np.isfinite(silent).all() and np.allclose(silent[:, 0], np.log(np.finfo(float).eps))   # zero_handling
    x = rng.standard_normal((6, 9))
    d = FO.speechpy_derivative(x)
    pad = np.pad(x, ((0, 0), (2, 2)), 'edge')
    assert np.allclose(d, (pad[:, 3:12] + 2 * pad[:, 4:13]) / 10.0)   # along the FEATURE axis; nothing subtracted (as published)
    f = FO.speechpy_features(y, 'mfcc', 13, 40, 25, 10, False, True)
    assert f.shape == (fr.shape[0], 39) and np.allclose(f[:, 0::3], c) and np.allclose(f[:, 2::3], FO.speechpy_derivative(FO.speechpy_derivative(c)))
    import pytest
    with pytest.raises(UnboundLocalError):                            # preprocess_all.py:77-79
        FO.speechpy_features(y, 'mfe', energy=False)
    e = FO.speechpy_features(y, 'mfe', 13, 40, 25, 10, True, False)
    assert np.allclose(e, np.log(np.hstack([spec, en[:, None]]) + 1e-8))


def test_speechpy_tables_of_the_product_are_the_oracle_s():
    """phones-las_amd/frontend.py builds the speechpy filterbank and the feature-axis difference matrix on the host (numpy, no
    GPU): they must be the oracle's, entry for entry."""
    import importlib
    import sys
    import types
    if 'phones_las_amd.frontend' in sys.modules:
        fe = sys.modules['phones_las_amd.frontend']
    else:
        fe = importlib.import_module('phones_las_amd.frontend')
    for n_mels, bins in ((40, 161), (40, 201), (23, 129), (80, 257)):
        assert np.array_equal(fe._speechpy_filterbanks(n_mels, bins, 16000), FO.speechpy_filterbanks(n_mels, bins, 16000, 0, 8000))
    rng = np.random.default_rng(5)
    for F in (13, 41, 5):
        x = rng.standard_normal((4, F))
        d1 = FO.speechpy_derivative(x)
        ref = np.stack([x, d1, FO.speechpy_derivative(d1)], axis=-1).reshape(4, -1)
        assert np.allclose(x @ fe._speechpy_delta_matrix(F), ref, atol=1e-12)
    assert isinstance(fe, types.ModuleType)
