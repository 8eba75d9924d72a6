#!/usr/bin/env python3
"""Offline front-end with the flag surface of the reference's preprocess_all.py (preprocess_all.py:193-255): a CSV of
``path<delimiter>language<delimiter>text`` lines -> features on the GPU (phones-las_amd/frontend.py) -> TFRecord of
SequenceExamples (+ vocab.txt, norm.dmp beside it).  Behaviour kept from the reference: per-line try/except skip, norm
statistics = MEAN of the per-utterance mean / std (quirk B5), vocabulary = most common `top_k` tokens.

--backend speechpy (round 6) runs speechpy==2.4's mfe / mfcc / extract_derivative_feature on the same kernels
(phones-las_amd/frontend.py; utterance by utterance, as the reference's forced n_jobs = 1, preprocess_all.py:228-230).

Not on this path: audio decoding other than 16 kHz PCM WAV (the reference calls librosa.load),
lyon features, espeak-ng text->IPA: texts whose language column is not 'arpabet'/'ipa' are refused for --targets phones /
binary_features."""
import argparse
import os
import sys
import wave
from collections import Counter

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SAMPLE_RATE = 16000


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--input_file', required=True, help='File with audio paths and texts.')
    p.add_argument('--output_file', required=True, help='Target TFRecord file name.')
    p.add_argument('--top_k', type=int, default=1000, help='Max size of vocabulary.')
    p.add_argument('--save_norm', action='store_true')
    p.add_argument('--save_vocab', action='store_true')
    p.add_argument('--feature_type', type=str, choices=['mfe', 'mfcc', 'lyon'], default='mfcc')
    p.add_argument('--backend', type=str, choices=['speechpy', 'librosa'], default='librosa')
    p.add_argument('--n_mfcc', type=int, default=13)
    p.add_argument('--n_mels', type=int, default=40)
    p.add_argument('--energy', action='store_true')
    p.add_argument('--window', type=int, default=20)
    p.add_argument('--step', type=int, default=10)
    p.add_argument('--deltas', action='store_true')
    p.add_argument('--n_jobs', type=int, default=1)
    p.add_argument('--targets', type=str, choices=['words', 'phones', 'binary_features', 'chars'], default='words')
    p.add_argument('--binf_map', type=str, default='misc/binf_map.csv')
    p.add_argument('--start', type=int, default=0)
    p.add_argument('--count', type=int, default=-1)
    p.add_argument('--delimiter', type=str, default=',')
    p.add_argument('--gpu_batch', type=int, default=64,
                   help='utterances whose features are computed together on the GPU (three launches per batch instead of six to '
                        'nine per utterance; same features bit for bit)')
    return p.parse_args(argv)


def read_wav(path):
    with wave.open(path, 'rb') as w:
        if w.getframerate() != SAMPLE_RATE or w.getsampwidth() != 2:
            raise ValueError('%s: only 16 kHz 16-bit PCM WAV is decoded here (resample first)' % path)
        data = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2').astype(np.float32) / 32768.0
        if w.getnchannels() > 1:
            data = data.reshape(-1, w.getnchannels()).mean(1)
    return data


def read_audio_and_text(path, text):
    """preprocess_all.py:53-66."""
    text = ' '.join(text.split())
    for ch in ',.:;?!-_':
        text = text.replace(ch, '')
    return read_wav(path), text.lower().split()


def main(args):
    from phones_las_amd import frontend
    from phones_las_amd.utils import tfrecord
    from phones_las_amd.utils.features_utils import save_normalization
    out_dir = os.path.dirname(args.output_file)
    if args.feature_type == 'lyon' or args.backend == 'speechpy':      # preprocess_all.py:228-230
        print('Forcing n_jobs = 1 for selected configuration.')
        args.n_jobs = 1
    lines = open(args.input_file, 'r').readlines()
    count = len(lines) - args.start
    if 0 < args.count < len(lines):
        count = args.count
    lines = lines[args.start:args.start + count]
    vocabulary = Counter()
    means = stds = None
    total = 0
    def features_of(waves):
        """Features of a group of utterances: one batched pass; if that fails (one signal too short for the deltas, say) each
        utterance on its own, so that only the offending line is skipped, as the reference's per-line try / except does."""
        try:
            return [f.cpu().numpy() for f in frontend.calculate_acoustic_features_batch(args, waves)]
        except Exception:  # noqa: BLE001
            out = []
            for w in waves:
                try:
                    out.append(frontend.calculate_acoustic_features(args, w).cpu().numpy())
                except Exception as e:  # noqa: BLE001
                    print('Hopefully recoverable error: %s' % e)
                    out.append(None)
            return out

    with tfrecord.TFRecordWriter(args.output_file) as writer:
        pending = []                     # (waveform, tokens) of the lines read so far: flushed every --gpu_batch utterances

        def flush():
            nonlocal means, stds, total
            if not pending:
                return
            for (_, tokens), feats in zip(pending, features_of([w for w, _ in pending])):
                if feats is None:
                    continue
                if args.save_norm:
                    m, s = feats.mean(0), feats.std(0)
                    means, stds = (m, s) if means is None else (means + m, stds + s)
                    total += 1
                writer.write(tfrecord.make_example(feats, tokens))
            del pending[:]

        for line in lines:
            try:
                filename, language, text = line.split(args.delimiter)
                waveform, tokens = read_audio_and_text(filename, text.strip())
            except Exception as e:  # noqa: BLE001  (the reference skips unreadable lines, preprocess_all.py:177-181)
                print('Failed to read audio or text! Exception: %s' % e)
                continue
            try:
                if args.targets in ('phones', 'binary_features'):
                    if language not in ('arpabet', 'ipa'):
                        raise ValueError('text->IPA (espeak-ng) is not available; give phones with language arpabet/ipa')
                    if args.targets == 'binary_features':
                        raise ValueError('binary-feature targets are not implemented on this path yet')
                elif args.targets == 'chars':
                    tokens = [c for c in ' '.join(tokens)]
            except Exception as e:  # noqa: BLE001
                print('Hopefully recoverable error: %s' % e)
                continue
            # (the vocabulary counts a line's tokens BEFORE its features are formed, as the reference does,
            # preprocess_all.py:147-148: a line whose audio then fails still contributes to vocab.txt -- ADVICE r3)
            vocabulary.update(tokens)
            pending.append((waveform, tokens))
            if len(pending) >= max(1, args.gpu_batch):
                flush()
        flush()
    if args.save_norm and total:
        save_normalization(os.path.join(out_dir, 'norm.dmp'), means / total, stds / total)
    if args.save_vocab:
        with open(os.path.join(out_dir, 'vocab.txt'), 'w') as f:
            for x, _ in vocabulary.most_common(args.top_k):
                f.write(x + '\n')
    return total


if __name__ == '__main__':
    main(parse_args())
