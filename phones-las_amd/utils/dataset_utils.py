"""Input pipeline with the reference's semantics (utils/dataset_utils.py:138-308) as plain Python iterators over
numpy batches: TFRecord -> parse -> [filter max_frames/max_symbols] -> repeat -> shuffle(buffer = batch*500) ->
vocab lookup -> (x - mean)/std -> [<s>]+y / y+[</s>] -> padded batches (x pad 0.0, labels pad EOS id,
drop_remainder=True).  The T2T format needs tensor2tensor and is out of scope.

Deliberate fix of reference quirk B1: ``is_infer`` IS forwarded (no shuffle, remainder kept when inferring), so
predictions stay aligned with the targets file."""
import os
import random

import numpy as np

from . import tfrecord
from .features_utils import load_normalization
from .vocab_utils import SOS, EOS, create_vocab_table

__all__ = ['input_fn', 'process_dataset', 'read_dataset']


def read_dataset(filename, num_channels=39):
    """utils/dataset_utils.py:138-160: yields (inputs [T,F] f32, labels list[str]); ``*.txt`` = list of TFRecord files."""
    files = [filename]
    if filename.endswith('.txt'):
        with open(filename, 'r') as f:
            files = [x.strip() for x in f.readlines() if x.strip()]

    def gen():
        for path in files:
            for rec in tfrecord.tf_record_iterator(path):
                yield tfrecord.parse_sequence_example(rec, num_channels)
    return gen


def _shuffle(it, buffer_size, rng):
    buf = []
    for item in it:
        if len(buf) < buffer_size:
            buf.append(item)
            continue
        i = rng.randrange(buffer_size)
        yield buf[i]
        buf[i] = item
    rng.shuffle(buf)
    for item in buf:
        yield item


def process_dataset(dataset, vocab_table, sos, eos, means=None, stds=None, batch_size=8, num_epochs=1,
                    num_parallel_calls=32, is_infer=False, max_frames=-1, max_symbols=-1, seed=None):
    """utils/dataset_utils.py:163-283.  ``dataset`` is a zero-argument callable returning a fresh iterator (so it can
    be repeated).  Yields (features, labels) dicts of numpy arrays."""
    sos_id, eos_id = vocab_table.lookup([sos])[0], vocab_table.lookup([eos])[0]
    rng = random.Random(seed)

    def examples():
        epoch = 0
        while num_epochs <= 0 or epoch < num_epochs:              # dataset.repeat(num_epochs)
            n = 0
            for inputs, labels in dataset():
                if max_frames > 0 and not (inputs.shape[0] <= max_frames and len(labels) <= max_symbols):
                    continue
                n += 1
                yield inputs, labels
            epoch += 1
            if n == 0:
                return

    stream = examples()
    if not is_infer:
        stream = _shuffle(stream, batch_size * 500, rng)          # shuffle AFTER repeat (quirk B3)

    def encode(inputs, labels):
        ids = np.asarray(vocab_table.lookup(labels), dtype=np.int32)
        if means is not None and stds is not None:
            inputs = (inputs - means) / stds
        x = inputs.astype(np.float32)
        tin = np.concatenate(([sos_id], ids)).astype(np.int32)
        tout = np.concatenate((ids, [eos_id])).astype(np.int32)
        return x, tin, tout

    def batches():
        batch = []
        for inputs, labels in stream:
            batch.append(encode(inputs, labels))
            if len(batch) == batch_size:
                yield collate(batch)
                batch = []
        if batch and is_infer:                                     # drop_remainder=True except when inferring (B1)
            yield collate(batch)

    def collate(batch):
        B = len(batch)
        F = batch[0][0].shape[1]
        T = max_frames if max_frames > 0 else max(b[0].shape[0] for b in batch)
        U = max_symbols if max_frames > 0 else max(b[1].shape[0] for b in batch)
        x = np.zeros((B, T, F), np.float32)
        tin = np.full((B, U), eos_id, np.int32)
        tout = np.full((B, U), eos_id, np.int32)
        sl = np.zeros(B, np.int32)
        tl = np.zeros(B, np.int32)
        for i, (xi, a, b) in enumerate(batch):
            x[i, :xi.shape[0]] = xi
            n = min(a.shape[0], U)                                 # quirk B4: padded shape is [max_symbols]
            tin[i, :n], tout[i, :n] = a[:n], b[:n]
            sl[i], tl[i] = xi.shape[0], n
        return ({'encoder_inputs': x, 'source_sequence_length': sl},
                {'targets_inputs': tin, 'targets_outputs': tout, 'target_sequence_length': tl})

    return batches()


def input_fn(dataset_filename, vocab_filename, norm_filename=None, num_channels=39, batch_size=8, num_epochs=1,
             num_parallel_calls=32, max_frames=-1, max_symbols=-1, take=0, is_infer=False, seed=None):
    """utils/dataset_utils.py:286-308 (same arguments).  Returns an iterator of (features, labels) numpy batches."""
    dataset = read_dataset(dataset_filename, num_channels)
    vocab_table = create_vocab_table(vocab_filename)
    means = stds = None
    if norm_filename is not None and os.path.exists(norm_filename):
        means, stds = load_normalization(norm_filename)
    it = process_dataset(dataset, vocab_table, SOS, EOS, means, stds, batch_size, num_epochs,
                         num_parallel_calls=num_parallel_calls, is_infer=is_infer, max_frames=max_frames,
                         max_symbols=max_symbols, seed=seed)
    if take > 0:
        def limited():
            for i, b in enumerate(it):
                if i >= take:
                    return
                yield b
        return limited()
    return it
