"""Throughput of the batched acoustic front-end (phones-las_amd/frontend.calculate_acoustic_features_batch: STFT -> mel ->
dB -> DCT -> energy -> deltas in three launches over a batch) against the per-utterance path, on synthetic audio:
audio-seconds per second, and the HBM traffic the stages NEED (samples in, features out, the intermediate mel rows once
each way) over the time they take, as a fraction of the 8 TB/s roof.  Env: N (utterances, default 256), SEC (seconds each,
default 8 = T=800 frames), TYPE (mfcc | mfe)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phones_las_amd import frontend  # noqa: E402

N = int(os.environ.get('N', '256'))
SEC = float(os.environ.get('SEC', '8'))
args = argparse.Namespace(feature_type=os.environ.get('TYPE', 'mfcc'), backend='librosa', n_mfcc=13, n_mels=40, window=20, step=10,
                          energy=True, deltas=True)
rng = np.random.default_rng(0)
waves = [torch.from_numpy(rng.standard_normal(int(SEC * 16000)).astype(np.float32) * 0.1).cuda() for _ in range(N)]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


t_batch = timed(lambda: frontend.calculate_acoustic_features_batch(args, waves))
t_single = timed(lambda: [frontend.calculate_acoustic_features(args, w) for w in waves[:32]], reps=2) * N / 32
audio = N * SEC
frames = N * (1 + int(SEC * 16000) // 160)
F = (13 + 1) * 3
need = N * SEC * 16000 * 4 * 2 + frames * (40 * 4 * 2 + (13 + 1) * 4 * 2 + F * 4)      # samples (melspec + rms), mel rows w+r, base w+r, out
print('batched : %.1f ms for %d x %.0f s = %.0f audio-seconds/s (%.0f x real time)' % (t_batch * 1e3, N, SEC, audio / t_batch, audio / t_batch))
print('per utt : %.1f ms (extrapolated from 32)  = %.0f audio-seconds/s; batched is %.1f x faster' % (t_single * 1e3, audio / t_single, t_single / t_batch))
print('needed HBM traffic %.1f MB in %.2f ms = %.1f GB/s = %.4f of the 8 TB/s roof (the stages are L2 / VALU work: a %d-point direct '
      'DFT per frame from a 412 KB twiddle table that lives in L2)' % (need / 1e6, t_batch * 1e3, need / t_batch / 1e9, need / t_batch / 8e12, 320))
# the arithmetic the stages need: the 320-point direct DFT (re and im: 4 flops per sample and bin) + the mel filterbank, fp32
flops = frames * (320 * 161 * 4 + 161 * 40 * 2)
print('arithmetic %.1f GF in %.2f ms (all three launches) >= %.1f TFLOP/s = %.3f of the 157 TFLOP/s fp32 (vector = matrix) roof; '
      'round 3 (one frame per workgroup, tables re-read per frame: LAS_FE_BLOCKED=0): 4.8 ms' % (flops / 1e9, t_batch * 1e3, flops / t_batch / 1e12, flops / t_batch / 157e12))
