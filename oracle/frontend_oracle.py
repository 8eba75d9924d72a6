"""CPU oracle of the acoustic front-end (TEST INFRASTRUCTURE ONLY; numpy float64).

Two pipelines of the reference:
  * ``tf_mfcc``  — utils/features_utils.py:5-20 ``calculate_mfcc_op`` (TF contrib.signal; dead code in the reference but
    named by the north star): periodic-Hann STFT without centering -> magnitude -> HTK mel (80..7600 Hz, no area
    normalisation) -> log(x + 1e-6) -> DCT-II * (2N)^-1/2 -> first `coeffs`.
  * ``librosa_features`` — preprocess_all.py:69-130 with ``--backend librosa`` (librosa==0.7.1 semantics, SURVEY.md
    A.9): centred reflect-padded periodic-Hann STFT power spectrogram -> Slaney mel basis (fmin 0, fmax sr/2, area
    normalised) -> MFCC = ortho DCT-II(power_to_db)[:n_mfcc] | MFE = amplitude_to_db(mel power) (the reference's
    quirk) -> optional RMS energy column -> optional Savitzky-Golay deltas (width 9, orders 1 and 2, mode 'interp'),
    interleaved [c0, d c0, dd c0, c1, ...].

  * ``speechpy_features`` — preprocess_all.py:69-130 with ``--backend speechpy`` (round 6): ``speechpy.feature.mfe`` /
    ``mfcc`` / ``extract_derivative_feature`` of speechpy==2.4 (requirements.txt:19), a third-party dependency that is not
    under /root/reference and not installed: its published algorithm is restated below FROM KNOWLEDGE OF THAT RELEASE, quirks
    included (each is named where it is restated).  Anchored on the reference's call sites (preprocess_all.py:73-79, 88-91,
    122-123) -- the reference holds no test or golden vector for it.

PARITY UNPINNED against librosa / TensorFlow / speechpy themselves (none is installed).  The building blocks
are pinned in tests/test_oracle_frontend.py against scipy (get_window, fft.dct, savgol_filter, rfft) and torch.stft.
"""
import numpy as np

SAMPLE_RATE = 16000


def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def frame_signal(y, n_fft, hop, center):
    y = np.asarray(y, dtype=np.float64)
    if center:
        y = np.pad(y, n_fft // 2, mode='reflect')
    n = 1 + (len(y) - n_fft) // hop
    if n <= 0:
        return np.zeros((0, n_fft))
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n)[:, None]
    return y[idx]


def stft_mag(y, n_fft, hop, center, power):
    fr = frame_signal(y, n_fft, hop, center) * hann_periodic(n_fft)
    spec = np.fft.rfft(fr, n=n_fft, axis=1)
    return np.abs(spec) ** power


# ---- mel filterbanks -----------------------------------------------------------------------------
def hz_to_mel_htk(f):
    return 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)


def mel_htk_matrix(n_mels, n_bins, sr, lo, hi):
    """tf.contrib.signal.linear_to_mel_weight_matrix: [n_bins, n_mels], triangles in mel space, DC bin zeroed."""
    nyq = sr / 2.0
    lin = np.linspace(0.0, nyq, n_bins)[1:]
    spec_mel = hz_to_mel_htk(lin)[:, None]
    edges = np.linspace(hz_to_mel_htk(lo), hz_to_mel_htk(hi), n_mels + 2)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lo_slope = (spec_mel - lower) / (center - lower)
    up_slope = (upper - spec_mel) / (upper - center)
    w = np.maximum(0.0, np.minimum(lo_slope, up_slope))
    return np.pad(w, [[1, 0], [0, 0]])


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_slaney_matrix(n_mels, n_fft, sr, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, htk=False, norm=1): [n_mels, 1 + n_fft//2], Slaney area normalisation."""
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return w * enorm[:, None]


def dct2_matrix(n_out, n_in, ortho=True):
    """DCT-II as a matrix [n_out, n_in]: y_k = 2 sum_n x_n cos(pi k (2n+1) / 2N) (scipy convention), ortho-scaled."""
    k = np.arange(n_out)[:, None]
    n = np.arange(n_in)[None, :]
    m = 2.0 * np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in))
    if ortho:
        m = m * np.sqrt(1.0 / (2.0 * n_in))
        m[0] *= np.sqrt(0.5)
    return m


def power_to_db(s, amin=1e-10, top_db=80.0):
    db = 10.0 * np.log10(np.maximum(amin, s))
    return np.maximum(db, db.max() - top_db) if top_db is not None and db.size else db


def amplitude_to_db(s, amin=1e-5, top_db=80.0):
    return power_to_db(np.abs(s) ** 2, amin=amin ** 2, top_db=top_db)


# ---- Savitzky-Golay deltas (librosa.feature.delta = scipy.signal.savgol_filter(mode='interp')) --------------------
def savgol_coeffs(width, order, deriv):
    half = width // 2
    x = np.arange(-half, half + 1, dtype=np.float64)
    A = np.vander(x, order + 1, increasing=True)            # [width, order+1]
    pinv = np.linalg.pinv(A)                                # [order+1, width]
    fact = float(np.prod(np.arange(1, deriv + 1))) if deriv else 1.0
    return pinv[deriv] * fact                               # correlate with the data window


def savgol_edge_matrix(width, order, deriv):
    """Rows of the 'interp' edge handling: value of the deriv-th derivative of the LS polynomial fitted to the first
    `width` samples, evaluated at positions 0..half-1 (mirror for the end).  [half, width]."""
    half = width // 2
    x = np.arange(width, dtype=np.float64)
    A = np.vander(x, order + 1, increasing=True)
    pinv = np.linalg.pinv(A)                                # coefficients = pinv @ window
    rows = []
    for p in range(half):
        # d^deriv/dx^deriv of sum_j c_j x^j at x = p
        d = np.zeros(order + 1)
        for j in range(deriv, order + 1):
            d[j] = np.prod(np.arange(j - deriv + 1, j + 1)) * (p ** (j - deriv))
        rows.append(d @ pinv)
    return np.array(rows)


def delta(x, width=9, order=1):
    """Along axis 0 ([T, F])."""
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[0]
    if T < width:
        raise ValueError('delta needs at least %d frames' % width)
    half = width // 2
    c = savgol_coeffs(width, order, order)
    out = np.zeros_like(x)
    for t in range(half, T - half):
        out[t] = c @ x[t - half:t + half + 1]
    E = savgol_edge_matrix(width, order, order)
    out[:half] = E @ x[:width]
    # end: fit on the last window, evaluate at the last `half` positions
    x_idx = np.arange(width, dtype=np.float64)
    A = np.vander(x_idx, order + 1, increasing=True)
    pinv = np.linalg.pinv(A)
    for i, p in enumerate(range(width - half, width)):
        d = np.zeros(order + 1)
        for j in range(order, order + 1 + 0):
            pass
        for j in range(order, A.shape[1]):
            d[j] = np.prod(np.arange(j - order + 1, j + 1)) * (p ** (j - order))
        out[T - half + i] = (d @ pinv) @ x[T - width:]
    return out


# ---- pipelines ----------------------------------------------------------------------------------------------------
def tf_mfcc(y, sample_rate=SAMPLE_RATE, coeffs=13, window=320, step=160, mels=40):
    mag = stft_mag(y, window, step, center=False, power=1.0)
    mel = mag @ mel_htk_matrix(mels, mag.shape[1], sample_rate, 80.0, 7600.0)
    logmel = np.log(mel + 1e-6)
    n = logmel.shape[1]
    dct = dct2_matrix(n, n, ortho=False) * np.sqrt(1.0 / (2.0 * n))    # mfccs_from_log_mel_spectrograms scaling
    return (logmel @ dct.T)[:, :coeffs]


def rms(y, frame_length, hop):
    fr = frame_signal(y, frame_length, hop, center=True)
    return np.sqrt(np.mean(fr ** 2, axis=1, keepdims=True))


def librosa_features(y, feature_type='mfcc', n_mfcc=13, n_mels=40, window_ms=20, step_ms=10, energy=False, deltas=False,
                     sr=SAMPLE_RATE):
    n_fft = int(window_ms * sr / 1000.0)
    hop = int(step_ms * sr / 1000.0)
    S = stft_mag(y, n_fft, hop, center=True, power=2.0)                       # [T, bins]
    mel = S @ mel_slaney_matrix(n_mels, n_fft, sr).T                            # [T, n_mels]
    if feature_type == 'mfcc':
        feats = power_to_db(mel) @ dct2_matrix(n_mfcc, n_mels, ortho=True).T
    elif feature_type == 'mfe':
        feats = amplitude_to_db(mel)                                            # quirk: amplitude_to_db of a POWER mel
    else:
        raise ValueError('Unexpected features type.')
    if energy:
        feats = np.hstack([feats, rms(y, n_fft, hop)])
    if deltas:
        d1, d2 = delta(feats, order=1), delta(feats, order=2)
        feats = np.stack([feats, d1, d2], axis=-1).reshape(feats.shape[0], -1)  # interleaved [c0,dc0,ddc0,c1,...]
    return feats


# ---- speechpy==2.4 (third-party; restated from the published source of that release) ------------------------------
SPEECHPY_EPS = np.finfo(float).eps


def speechpy_zero_handling(x):
    """speechpy.functions.zero_handling: exact zeros become machine epsilon (so that the log exists)."""
    return np.where(x == 0, SPEECHPY_EPS, x)


def speechpy_stack_frames(sig, sampling_frequency, frame_length, frame_stride):
    """speechpy.processing.stack_frames(..., filter=ones, zero_padding=False) as speechpy.feature.mfe calls it: frames of
    round(fs * frame_length) samples every round(fs * frame_stride), rectangular window.  QUIRK kept: numframes =
    floor((N - L) / stride) -- one frame FEWER than fit (the frame that ends exactly at or before the last sample is dropped)."""
    sig = np.asarray(sig, dtype=np.float64)
    L = int(np.round(sampling_frequency * frame_length))
    stride = float(np.round(sampling_frequency * frame_stride))
    numframes = int(np.floor((sig.shape[0] - L) / stride))
    if numframes <= 0:
        return np.zeros((0, L))
    idx = np.arange(L)[None, :] + (np.arange(numframes) * stride).astype(np.int32)[:, None]
    return sig[idx]


def speechpy_filterbanks(num_filter, coefficients, sampling_freq, low_freq=None, high_freq=None):
    """speechpy.feature.filterbanks: [num_filter, coefficients] triangles between HTK-mel-spaced points.  QUIRKS kept:
    `low_freq = low_freq or 300` -- the low_frequency=0 that mfe passes is falsy, so the bank starts at 300 Hz; the bin of a
    frequency is floor((coefficients + 1) * f / fs) with coefficients = fft_length / 2 + 1 (the number of rfft bins, not the
    FFT length), so the bank covers the lower half of the bins only."""
    high_freq = high_freq or sampling_freq / 2
    low_freq = low_freq or 300
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    mels = np.linspace(mel(low_freq), mel(high_freq), num_filter + 2)
    hertz = 700.0 * (np.exp(mels / 1127.0) - 1.0)
    freq_index = np.floor((coefficients + 1) * hertz / sampling_freq).astype(int)
    bank = np.zeros((num_filter, coefficients))
    for i in range(num_filter):
        left, middle, right = int(freq_index[i]), int(freq_index[i + 1]), int(freq_index[i + 2])
        z = np.linspace(left, right, num=right - left + 1)
        tri = np.zeros(z.shape)                                   # speechpy.functions.triangle
        up = np.logical_and(left < z, z <= middle)
        tri[up] = (z[up] - left) / (middle - left) if middle > left else 0.0
        down = np.logical_and(middle <= z, z < right)
        tri[down] = (right - z[down]) / (right - middle) if right > middle else 0.0
        bank[i, left:right + 1] = tri
    return bank


def speechpy_mfe(signal, sampling_frequency, frame_length, frame_stride, num_filters, fft_length):
    """speechpy.feature.mfe: (filterbank energies [T, num_filters], frame energies [T]); power spectrum = |rfft(frame, n =
    fft_length)|^2 / fft_length (a frame longer than fft_length is cropped by rfft, a shorter one zero-padded)."""
    frames = speechpy_stack_frames(signal, sampling_frequency, frame_length, frame_stride)
    power = np.abs(np.fft.rfft(frames, n=fft_length, axis=-1)) ** 2 / fft_length
    energy = speechpy_zero_handling(power.sum(axis=1))
    bank = speechpy_filterbanks(num_filters, power.shape[1], sampling_frequency, 0, sampling_frequency / 2)
    return speechpy_zero_handling(power @ bank.T), energy


def speechpy_mfcc(signal, sampling_frequency, frame_length, frame_stride, num_cepstral, num_filters, fft_length):
    """speechpy.feature.mfcc (dc_elimination=True): ortho DCT-II of the log filterbank energies, first num_cepstral; the first
    coefficient REPLACED by the log frame energy."""
    feat, energy = speechpy_mfe(signal, sampling_frequency, frame_length, frame_stride, num_filters, fft_length)
    if len(feat) == 0:
        return np.empty((0, num_cepstral))
    feat = np.log(feat) @ dct2_matrix(num_cepstral, num_filters, ortho=True).T
    feat[:, 0] = np.log(energy)
    return feat


def speechpy_derivative(feat, delta_windows=2):
    """speechpy.processing.derivative_extraction(feat, DeltaWindows=2).  QUIRKS kept: the differences run along the FEATURE axis
    (axis 1, edge-padded), not along time; and the published loop body is the two lines
        dif = Range * FEAT[:, offset + Range:offset + Range + cols]
        - FEAT[:, offset - Range:offset - Range + cols]
    of which the second is an expression statement of its own (a line break, no backslash): nothing is subtracted, so
    DIF[c] = (1 * f[c + 1] + 2 * f[c + 2]) / 10.  SPEECHPY_DELTA_SUBTRACTS = True gives the reading the author meant
    (Range * f[c + R] - f[c - R]); the product's table follows this function, whichever reading it is set to."""
    rows, cols = feat.shape
    dif = np.zeros(feat.shape)
    scale = 0.0
    pad = np.pad(feat, ((0, 0), (delta_windows, delta_windows)), 'edge')
    for i in range(delta_windows):
        r = i + 1
        d = r * pad[:, delta_windows + r:delta_windows + r + cols]
        if SPEECHPY_DELTA_SUBTRACTS:
            d = d - pad[:, delta_windows - r:delta_windows - r + cols]
        scale += 2 * r ** 2
        dif += d
    return dif / scale


SPEECHPY_DELTA_SUBTRACTS = False


def speechpy_features(y, feature_type='mfcc', n_mfcc=13, n_mels=40, window_ms=20, step_ms=10, energy=False, deltas=False,
                      sr=SAMPLE_RATE):
    """preprocess_all.py:69-130 with --backend speechpy.  mfe: log(hstack(filterbank energies, frame energy) + 1e-8) -- the
    reference assigns `acoustic_features` only under --energy (:77-79), without it the function dies with UnboundLocalError, and so
    does this one; mfcc: speechpy's mfcc, --energy ignored (:88-91); --deltas: extract_derivative_feature's [T, F, 3] cube
    reshaped to [T, 3F], i.e. interleaved [c0, d c0, dd c0, c1, ...] (:122-128)."""
    n_fft = int(window_ms * sr / 1000.0)
    if feature_type == 'mfe':
        spec, en = speechpy_mfe(y, sr, window_ms * 1e-3, step_ms * 1e-3, n_mels, n_fft)
        if not energy:
            raise UnboundLocalError("local variable 'acoustic_features' referenced before assignment")
        feats = np.log(np.hstack((spec, en[:, None])) + 1e-8)
    elif feature_type == 'mfcc':
        feats = speechpy_mfcc(y, sr, window_ms * 1e-3, step_ms * 1e-3, n_mfcc, n_mels, n_fft)
    else:
        raise ValueError('Unexpected features type.')
    if deltas:
        d1 = speechpy_derivative(feats)
        d2 = speechpy_derivative(d1)
        feats = np.stack([feats, d1, d2], axis=-1).reshape(feats.shape[0], -1)
    return feats
