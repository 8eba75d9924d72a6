"""Acoustic front-end on the GPU, mirroring the reference's two entry points:

  * ``calculate_acoustic_features(args, waveform)`` — preprocess_all.py:69-130, ``--backend librosa`` and ``--backend speechpy``
    (feature_type mfcc | mfe, --n_mfcc, --n_mels, --window, --step, --energy, --deltas);
  * ``calculate_mfcc_op(sample_rate, coeffs, window, step, mels)`` — utils/features_utils.py:5-20.

The signal-processing tables (periodic Hann window, DFT twiddles, Slaney / HTK mel bases, DCT-II basis,
Savitzky-Golay taps) are built here in float64 and handed to the table-driven kernels of csrc/frontend.hip.
The speechpy backend (round 6) runs on the same kernels with speechpy==2.4's tables -- rectangular frames without centering,
its filterbank, its feature-axis differences -- restated from the published source of that release (requirements.txt:19; the
package is not under /root/reference): see oracle/frontend_oracle.py for the quirks that are kept.  The lyon cochlear model and
audio decoding are out of scope."""
import functools

import numpy as np
import torch

from . import hip

SAMPLE_RATE = 16000
__all__ = ['calculate_acoustic_features', 'calculate_acoustic_features_batch', 'calculate_mfcc_op', 'SAMPLE_RATE']


# ---- tables (float64 on the host) -------------------------------------------------------------------------------------
def _hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def _twiddles(n_fft):
    bins = n_fft // 2 + 1
    ang = 2.0 * np.pi * np.outer(np.arange(n_fft), np.arange(bins)) / n_fft
    return np.cos(ang), -np.sin(ang)


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, brk, step = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    return np.where(f >= brk, brk / f_sp + np.log(np.maximum(f, 1e-10) / brk) / step, f / f_sp)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, brk, step = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    return np.where(m >= brk / f_sp, brk * np.exp(step * (m - brk / f_sp)), f_sp * m)


def _mel_basis_slaney(n_mels, n_fft, sr):
    """librosa.filters.mel defaults (Slaney scale, fmin 0, fmax sr/2, area normalised), returned as [bins, n_mels]."""
    freqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    pts = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(0.0), _slaney_hz_to_mel(sr / 2.0), n_mels + 2))
    w = np.zeros((1 + n_fft // 2, n_mels))
    for i in range(n_mels):
        up = (freqs - pts[i]) / (pts[i + 1] - pts[i])
        down = (pts[i + 2] - freqs) / (pts[i + 2] - pts[i + 1])
        w[:, i] = np.maximum(0.0, np.minimum(up, down)) * (2.0 / (pts[i + 2] - pts[i]))
    return w


def _mel_basis_htk(n_mels, bins, sr, lo, hi):
    """tf.contrib.signal.linear_to_mel_weight_matrix: [bins, n_mels], DC row zero, no normalisation."""
    mel = lambda f: 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)
    spec = mel(np.linspace(0.0, sr / 2.0, bins)[1:])[:, None]
    e = np.linspace(mel(lo), mel(hi), n_mels + 2)
    w = np.maximum(0.0, np.minimum((spec - e[:-2][None]) / (e[1:-1] - e[:-2])[None], (e[2:][None] - spec) / (e[2:] - e[1:-1])[None]))
    return np.vstack([np.zeros((1, n_mels)), w])


def _dct_basis(n_out, n_in, ortho):
    k, n = np.arange(n_out)[None, :], np.arange(n_in)[:, None]
    m = 2.0 * np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in))          # [n_in, n_out]
    m = m * np.sqrt(1.0 / (2.0 * n_in))
    if ortho:
        m[:, 0] *= np.sqrt(0.5)
    return m


def _savgol(width, order):
    """Interior taps and the two edge matrices of scipy.signal.savgol_filter(deriv=order, polyorder=order, 'interp')."""
    half = width // 2
    fact = float(np.prod(np.arange(1, order + 1)))
    xs = np.arange(-half, half + 1, dtype=np.float64)
    taps = np.linalg.pinv(np.vander(xs, order + 1, increasing=True))[order] * fact
    pinv = np.linalg.pinv(np.vander(np.arange(width, dtype=np.float64), order + 1, increasing=True))

    def rows(positions):
        out = []
        for p in positions:
            d = np.zeros(order + 1)
            for j in range(order, order + 1):
                d[j] = fact * (p ** (j - order))
            out.append(d @ pinv)
        return np.array(out)
    return taps, rows(range(half)), rows(range(width - half, width))


@functools.lru_cache(maxsize=16)
def _tables(kind, n_fft, n_mels, n_out, sr):
    dev = 'cuda'
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    cos, sin = _twiddles(n_fft)
    d = {'window': t(_hann_periodic(n_fft)), 'cos': t(cos), 'sin': t(sin)}
    if kind == 'librosa':
        d['mel'] = t(_mel_basis_slaney(n_mels, n_fft, sr))
        d['dct'] = t(_dct_basis(n_out, n_mels, ortho=True))
    else:
        d['mel'] = t(_mel_basis_htk(n_mels, n_fft // 2 + 1, sr, 80.0, 7600.0))
        d['dct'] = t(_dct_basis(n_mels, n_mels, ortho=False))
    for order in (1, 2):
        taps, lo, hi = _savgol(9, order)
        d['sg%d' % order] = (t(taps), t(lo), t(hi))
    return d


# ---- speechpy==2.4 tables (preprocess_all.py:73-79, 88-91, 122-123) ---------------------------------------------------
def _speechpy_filterbanks(num_filter, coefficients, sampling_freq):
    """speechpy.feature.filterbanks as speechpy.feature.mfe calls it (low_frequency = 0, high_frequency = fs / 2), [num_filter,
    coefficients].  Two properties of the published code are kept: `low_freq = low_freq or 300` turns the 0 into 300 Hz, and the
    bin of a frequency is floor((coefficients + 1) * f / fs) with coefficients = the number of rfft bins."""
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    hertz = 700.0 * (np.exp(np.linspace(mel(300.0), mel(sampling_freq / 2.0), num_filter + 2) / 1127.0) - 1.0)
    idx = np.floor((coefficients + 1) * hertz / sampling_freq).astype(int)
    bank = np.zeros((num_filter, coefficients))
    for i in range(num_filter):
        left, middle, right = int(idx[i]), int(idx[i + 1]), int(idx[i + 2])
        for z in range(left, right + 1):
            if left < z <= middle:
                bank[i, z] = (z - left) / (middle - left)
            elif middle <= z < right:
                bank[i, z] = (right - z) / (right - middle)
    return bank


# speechpy.processing.derivative_extraction as published: the loop body's second line (`- FEAT[...]`) is a statement of its own,
# nothing is subtracted.  True: the reading its author meant (Range * f[c + R] - f[c - R]).
SPEECHPY_DELTA_SUBTRACTS = False


def _speechpy_delta_matrix(F):
    """extract_derivative_feature as ONE matrix: [F, 3F], columns interleaved [c0, d c0, dd c0, c1, ...].  The differences run
    along the FEATURE axis (edge-padded), DeltaWindows = 2: d f[c] = sum_R R f[c + R] (- f[c - R]) / 10."""
    M = np.zeros((F, F))
    for c in range(F):
        for r in (1, 2):
            M[min(c + r, F - 1), c] += r / 10.0
            if SPEECHPY_DELTA_SUBTRACTS:
                M[max(c - r, 0), c] -= 1.0 / 10.0
    W = np.zeros((F, 3 * F))
    W[:, 0::3] = np.eye(F)
    W[:, 1::3] = M
    W[:, 2::3] = M @ M
    return W


@functools.lru_cache(maxsize=16)
def _tables_speechpy(n_fft, frame_len, n_mels, n_cep, sr, subtracts):
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device='cuda')
    bins = n_fft // 2 + 1
    cos, sin = _twiddles(n_fft)
    window = np.zeros(n_fft)
    window[:min(frame_len, n_fft)] = 1.0                         # rectangular; rfft(n = fft_length) crops or zero-pads the frame
    # power spectrum = |X|^2 / fft_length; one more column sums it: the frame energy
    mel = np.hstack([_speechpy_filterbanks(n_mels, bins, sr).T, np.ones((bins, 1))]) / n_fft
    # ortho DCT-II of the log filterbank energies; its first coefficient replaced by the log frame energy (dc_elimination)
    dct = np.zeros((n_mels + 1, n_cep))
    dct[:n_mels] = _dct_basis(n_cep, n_mels, ortho=True)
    dct[:, 0] = 0.0
    dct[n_mels, 0] = 1.0
    return {'window': t(window), 'cos': t(cos), 'sin': t(sin), 'mel': t(mel), 'dct': t(dct)}


@functools.lru_cache(maxsize=16)
def _speechpy_delta_table(F, subtracts):
    return torch.tensor(_speechpy_delta_matrix(F), dtype=torch.float32, device='cuda')


def _speechpy_features(args, waveform):
    """preprocess_all.py:69-130 with --backend speechpy (speechpy.feature.mfe / mfcc / extract_derivative_feature)."""
    lib, st = hip.lib(), hip.stream()
    n_fft = int(args.window * SAMPLE_RATE / 1000.0)
    frame_len = int(np.round(SAMPLE_RATE * (args.window * 1e-3)))
    stride = int(np.round(SAMPLE_RATE * (args.step * 1e-3)))
    if args.feature_type == 'mfe' and not args.energy:
        # preprocess_all.py:77-79: `acoustic_features` is assigned under --energy only; the reference dies here with this error
        raise UnboundLocalError("local variable 'acoustic_features' referenced before assignment (preprocess_all.py:79: --feature_type mfe "
                                "--backend speechpy works with --energy only)")
    wave = _as_wave(waveform)
    N = wave.numel()
    frames = int(np.floor((N - frame_len) / float(stride)))      # stack_frames(zero_padding=False): one frame fewer than fit
    if frames <= 0 or stride <= 0:
        raise ValueError('signal shorter than two frames')
    bins = n_fft // 2 + 1
    tb = _tables_speechpy(n_fft, frame_len, args.n_mels, args.n_mfcc, SAMPLE_RATE, SPEECHPY_DELTA_SUBTRACTS)
    dev = wave.device
    spec = torch.empty(frames, bins, device=dev)
    hip.check(lib.las_fe_stft(hip.p(wave), N, n_fft, stride, 0, 2, hip.p(tb['window']), hip.p(tb['cos']), hip.p(tb['sin']), bins,
                              hip.p(spec), bins, frames, st))
    M1 = args.n_mels + 1
    logmel = torch.empty(frames, M1, device=dev)
    if args.feature_type == 'mfcc':
        # log(zero_handling(.)): exact zeros become machine epsilon
        hip.check(lib.las_fe_matmul(hip.p(spec), bins, hip.p(tb['mel']), M1, hip.p(logmel), M1, frames, M1, bins, 1,
                                    float(np.finfo(float).eps), st))
        feats = torch.empty(frames, args.n_mfcc, device=dev)
        hip.check(lib.las_fe_matmul(hip.p(logmel), M1, hip.p(tb['dct']), args.n_mfcc, hip.p(feats), args.n_mfcc, frames,
                                    args.n_mfcc, M1, 0, 0.0, st))
    else:
        # np.log(hstack(spec, energy) + 1e-8)
        hip.check(lib.las_fe_matmul(hip.p(spec), bins, hip.p(tb['mel']), M1, hip.p(logmel), M1, frames, M1, bins, 1, 1e-8, st))
        feats = logmel
    if args.deltas:
        F = feats.shape[1]
        W = _speechpy_delta_table(F, SPEECHPY_DELTA_SUBTRACTS)
        out = torch.empty(frames, 3 * F, device=dev)
        hip.check(lib.las_fe_matmul(hip.p(feats), F, hip.p(W), 3 * F, hip.p(out), 3 * F, frames, 3 * F, F, 0, 0.0, st))
        feats = out
    return feats


def _as_wave(waveform):
    w = torch.as_tensor(waveform, dtype=torch.float32)
    return w.cuda().contiguous() if not w.is_cuda else w.contiguous()


def calculate_acoustic_features(args, waveform):
    """preprocess_all.py:69-130.  ``args`` needs feature_type, n_mfcc, n_mels, window, step, energy, deltas and backend
    ('librosa', the default, or 'speechpy').  Returns a CUDA fp32 tensor [T, F]."""
    backend = getattr(args, 'backend', 'librosa')
    if backend not in ('librosa', 'speechpy'):
        raise ValueError('backend must be librosa or speechpy (got %r)' % (backend,))
    if args.feature_type not in ('mfcc', 'mfe'):
        raise ValueError('Unexpected features type.' if args.feature_type != 'lyon' else 'lyon features are out of scope')
    if backend == 'speechpy':
        return _speechpy_features(args, waveform)
    lib, st = hip.lib(), hip.stream()
    n_fft = int(args.window * SAMPLE_RATE / 1000.0)
    hop = int(args.step * SAMPLE_RATE / 1000.0)
    wave = _as_wave(waveform)
    N = wave.numel()
    frames = 1 + N // hop                                  # center=True
    bins = n_fft // 2 + 1
    tb = _tables('librosa', n_fft, args.n_mels, args.n_mfcc, SAMPLE_RATE)
    dev = wave.device
    spec = torch.empty(frames, bins, device=dev)
    hip.check(lib.las_fe_stft(hip.p(wave), N, n_fft, hop, 1, 2, hip.p(tb['window']), hip.p(tb['cos']), hip.p(tb['sin']), bins,
                              hip.p(spec), bins, frames, st))
    scratch = torch.empty(1, device=dev)
    mel_db = torch.empty(frames, args.n_mels, device=dev)
    if args.feature_type == 'mfcc':
        # power_to_db(S, amin=1e-10, top_db=80) then ortho DCT-II, first n_mfcc
        hip.check(lib.las_fe_matmul(hip.p(spec), bins, hip.p(tb['mel']), args.n_mels, hip.p(mel_db), args.n_mels, frames,
                                    args.n_mels, bins, 2, 1e-10, st))
        hip.check(lib.las_fe_top_db(hip.p(mel_db), args.n_mels, frames, args.n_mels, 80.0, hip.p(scratch), st))
        base = torch.empty(frames, args.n_mfcc, device=dev)
        hip.check(lib.las_fe_matmul(hip.p(mel_db), args.n_mels, hip.p(tb['dct']), args.n_mfcc, hip.p(base), args.n_mfcc, frames,
                                    args.n_mfcc, args.n_mels, 0, 0.0, st))
    else:
        # amplitude_to_db applied to the POWER mel spectrogram (the reference's quirk): 20 log10(max(S, 1e-5)), top_db 80
        hip.check(lib.las_fe_matmul(hip.p(spec), bins, hip.p(tb['mel']), args.n_mels, hip.p(mel_db), args.n_mels, frames,
                                    args.n_mels, bins, 2, 1e-5, st))
        mel_db.mul_(2.0)
        hip.check(lib.las_fe_top_db(hip.p(mel_db), args.n_mels, frames, args.n_mels, 80.0, hip.p(scratch), st))
        base = mel_db
    F = base.shape[1]
    if args.energy:
        feats = torch.empty(frames, F + 1, device=dev)
        feats[:, :F] = base
        hip.check(lib.las_fe_rms(hip.p(wave), N, n_fft, hop, hip.addr(feats, F), F + 1, frames, st))
        F += 1
    else:
        feats = base
    if args.deltas:
        out = torch.empty(frames, 3 * F, device=dev)
        hip.check(lib.las_fe_delta(hip.p(feats), F, frames, F, None, None, None, 9, hip.p(out), 3 * F, 3, 0, st))
        for order in (1, 2):
            taps, lo, hi = tb['sg%d' % order]
            hip.check(lib.las_fe_delta(hip.p(feats), F, frames, F, hip.p(taps), hip.p(lo), hip.p(hi), 9, hip.p(out), 3 * F, 3,
                                       order, st))
        feats = out
    return feats


def calculate_acoustic_features_batch(args, waveforms):
    """calculate_acoustic_features over a LIST of waveforms in three kernel launches (las_fe_batch_melspec -- preceded by one small fill of its per-utterance maxima -- / _finish / _delta over the
    frames of all utterances; a fourth of two small kernels when --energy / --deltas are off is not needed).  The per-frame
    arithmetic is that of the per-utterance kernels in the same order: the result is bit-identical to calling
    calculate_acoustic_features on every waveform.  Returns a list of CUDA fp32 tensors [T_u, F] (views of one buffer).
    preprocess_all.py:69-130 of the reference runs this chain file by file on the host (librosa); here the utterances of
    a batch share every launch."""
    if getattr(args, 'backend', 'librosa') == 'speechpy':
        # (the reference forces n_jobs = 1 for this backend, preprocess_all.py:228-230: utterance by utterance here too)
        return [calculate_acoustic_features(args, w) for w in waveforms]
    if getattr(args, 'backend', 'librosa') != 'librosa':
        raise ValueError('backend must be librosa or speechpy')
    if args.feature_type not in ('mfcc', 'mfe'):
        raise ValueError('Unexpected features type.' if args.feature_type != 'lyon' else 'lyon features are out of scope')
    if not len(waveforms):
        return []
    lib, st = hip.lib(), hip.stream()
    n_fft = int(args.window * SAMPLE_RATE / 1000.0)
    hop = int(args.step * SAMPLE_RATE / 1000.0)
    bins = n_fft // 2 + 1
    waves = [torch.as_tensor(w, dtype=torch.float32).reshape(-1) for w in waveforms]
    lens = [int(w.numel()) for w in waves]
    frames = [1 + n // hop for n in lens]                      # center=True
    if args.deltas and min(frames) < 9:
        raise ValueError('deltas need at least 9 frames per utterance (Savitzky-Golay window)')
    if min(lens) < n_fft // 2 + 1:
        raise ValueError('signal shorter than half a window (reflect padding)')
    dev = 'cuda'
    wave_off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64)
    frame_off = torch.tensor(np.concatenate([[0], np.cumsum(frames)]), dtype=torch.int32)
    total = int(frame_off[-1])
    flat = torch.cat([w.cpu() if w.is_cuda else w for w in waves]).pin_memory().to(dev, non_blocking=True) \
        if not all(w.is_cuda for w in waves) else torch.cat(waves)
    d_woff, d_foff = wave_off.to(dev), frame_off.to(dev)
    n_utt = len(waves)
    tb = _tables('librosa', n_fft, args.n_mels, args.n_mfcc, SAMPLE_RATE)
    mel_db = torch.empty(total, args.n_mels, device=dev)
    umax = torch.empty(n_utt, device=dev)
    mfcc = args.feature_type == 'mfcc'
    # mfcc: power_to_db(S, amin=1e-10); mfe: amplitude_to_db of the POWER mel spectrogram (the reference's quirk) = 2 * 10 log10(max(S, 1e-5))
    hip.check(lib.las_fe_batch_melspec(hip.p(flat), hip.p(d_woff), hip.p(d_foff), n_utt, total, n_fft, hop, 1, 2, hip.p(tb['window']),
                                       hip.p(tb['cos']), hip.p(tb['sin']), bins, hip.p(tb['mel']), args.n_mels, 2,
                                       1e-10 if mfcc else 1e-5, 1.0 if mfcc else 2.0, hip.p(mel_db), hip.p(umax), st))
    n_out = args.n_mfcc if mfcc else args.n_mels
    F = n_out + (1 if args.energy else 0)
    feats = torch.empty(total, F, device=dev)
    hip.check(lib.las_fe_batch_finish(hip.p(mel_db), args.n_mels, hip.p(d_foff), n_utt, total, hip.p(umax), 80.0,
                                      hip.p(tb['dct']) if mfcc else None, n_out, hip.p(flat), hip.p(d_woff), n_fft, hop,
                                      1 if args.energy else 0, hip.p(feats), F, st))
    if args.deltas:
        out = torch.empty(total, 3 * F, device=dev)
        (t1, l1, h1), (t2, l2, h2) = tb['sg1'], tb['sg2']
        hip.check(lib.las_fe_batch_delta(hip.p(feats), F, hip.p(d_foff), n_utt, total, F, hip.p(t1), hip.p(l1), hip.p(h1),
                                         hip.p(t2), hip.p(l2), hip.p(h2), 9, hip.p(out), 3 * F, st))
        feats = out
    fo = frame_off.tolist()
    return [feats[fo[u]:fo[u + 1]] for u in range(n_utt)]


def calculate_mfcc_op(sample_rate, coeffs, window, step, mels):
    """utils/features_utils.py:5-20: returns ``_mfcc_op(waveform)`` -> CUDA fp32 [frames, coeffs]."""
    def _mfcc_op(input_tensor):
        lib, st = hip.lib(), hip.stream()
        wave = _as_wave(input_tensor)
        N = wave.numel()
        frames = 1 + (N - window) // step                  # no centering / padding
        if frames <= 0:
            raise ValueError('signal shorter than one window')
        bins = window // 2 + 1
        tb = _tables('tf', window, mels, mels, sample_rate)
        dev = wave.device
        spec = torch.empty(frames, bins, device=dev)
        hip.check(lib.las_fe_stft(hip.p(wave), N, window, step, 0, 1, hip.p(tb['window']), hip.p(tb['cos']), hip.p(tb['sin']),
                                  bins, hip.p(spec), bins, frames, st))
        logmel = torch.empty(frames, mels, device=dev)
        hip.check(lib.las_fe_matmul(hip.p(spec), bins, hip.p(tb['mel']), mels, hip.p(logmel), mels, frames, mels, bins, 1, 1e-6, st))
        out = torch.empty(frames, mels, device=dev)
        hip.check(lib.las_fe_matmul(hip.p(logmel), mels, hip.p(tb['dct']), mels, hip.p(out), mels, frames, mels, mels, 0, 0.0, st))
        return out[..., :coeffs]
    return _mfcc_op
