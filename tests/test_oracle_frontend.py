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
