"""The train loop's input path at device speed: the reference's pipeline (utils/dataset_utils.py:138-283:
TFRecordDataset -> parse -> filter -> repeat -> shuffle(batch*500) -> vocab lookup -> (x - mean) / std -> padded batches)
with the per-record work in C (liblas_hip.so: las_tfrecord_index / las_tfrecord_parse_batch, include/las_hip.h) and the
normalisation + bf16 cast + zero padding in one HIP kernel (las_normalize_pad_bf16).

  * every TFRecord file is memory-mapped and indexed ONCE (framing, optional crc32c check, frames / labels per record);
  * repeat / filter / shuffle / batching run on record INDICES with the same generator and the same random draws as
    dataset_utils.process_dataset, so both paths yield the same utterances in the same order for the same seed;
  * a batch is parsed by one C call into pinned host buffers (packed frames, no host-side padding or normalisation),
    copied with non-blocking H2D copies on a copy stream and finished on the device;
  * a background thread keeps `depth` batches in flight; the consumer only makes its stream wait on the batch's event.

The batches are what LasModel.train_step takes: features['encoder_inputs'] is already the listener's bf16 [B, T', F']
layout (T' a multiple of 2^(L-1), F' of 8)."""
import ctypes as C
import mmap
import os
import queue
import random
import threading

import numpy as np
import torch

from .. import hip
from .dataset_utils import _shuffle
from .features_utils import load_normalization
from .vocab_utils import SOS, EOS, create_vocab_table

__all__ = ['IndexedRecords', 'HostBatcher', 'FastInput', 'fast_input_fn', 'shard_indices']


def _fnv1a(data):
    """FNV-1a, 64 bits, as las_vocab_lookup computes it (a hash of 0 is stored as 1: 0 marks an empty slot)."""
    h = 1469598103934665603
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h or 1


class IndexedRecords(object):
    """One or several TFRecord files (``*.txt`` = list of files, utils/dataset_utils.py:155-156), memory-mapped and indexed."""

    def __init__(self, filename, verify_crc=True):
        files = [filename]
        if filename.endswith('.txt'):
            with open(filename, 'r') as f:
                files = [x.strip() for x in f.readlines() if x.strip()]
        lib = hip.lib()
        self.maps, self.file_of, self.offsets, self.lengths = [], [], [], []
        self.n_frames, self.n_labels, self.label_bytes = [], [], []
        for fi, path in enumerate(files):
            size = os.path.getsize(path)
            if size == 0:
                self.maps.append(None)
                continue
            fh = open(path, 'rb')
            mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            buf = np.frombuffer(mm, dtype=np.uint8)
            self.maps.append((fh, mm, buf))
            n = lib.las_tfrecord_index(buf.ctypes.data, size, 0, 0, None, None, None, None, None)
            if n < 0:
                raise IOError('%s: %s' % (path, lib.las_last_error().decode()))
            off, ln = np.empty(n, np.int64), np.empty(n, np.int64)
            nf, nl, lb = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int64)
            n2 = lib.las_tfrecord_index(buf.ctypes.data, size, int(verify_crc), n, off.ctypes.data, ln.ctypes.data,
                                        nf.ctypes.data, nl.ctypes.data, lb.ctypes.data)
            if n2 < 0:
                raise IOError('%s: %s' % (path, lib.las_last_error().decode()))
            if (nf < 0).any():
                raise IOError('%s holds records that are not SequenceExamples with inputs / labels' % path)
            self.file_of.append(np.full(n, fi, np.int32))
            self.offsets.append(off); self.lengths.append(ln)
            self.n_frames.append(nf); self.n_labels.append(nl); self.label_bytes.append(lb)
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
        self.file_of, self.offsets, self.lengths = cat(self.file_of, np.int32), cat(self.offsets, np.int64), cat(self.lengths, np.int64)
        self.n_frames, self.n_labels, self.label_bytes = cat(self.n_frames, np.int32), cat(self.n_labels, np.int32), cat(self.label_bytes, np.int64)
        # absolute addresses: one parse call takes records of several files (offsets relative to address 0)
        base = np.array([m[2].ctypes.data if m is not None else 0 for m in self.maps], dtype=np.int64)
        self.addr = base[self.file_of] + self.offsets if len(self.offsets) else np.zeros(0, np.int64)

    def __len__(self):
        return len(self.offsets)


class HostBatcher(object):
    """The host half of a batch: padded shapes from the INDEXED counts, then one C call that parses the records into packed
    frames + token ids.  Separate from FastInput so that the per-rank work of a data-parallel job can be measured and tested
    without a GPU: `shapes()` looks at the whole global batch (every rank must agree on T', U and the decoder steps), `parse()`
    only at the records it is given -- a rank hands it ITS slice of the global index batch, so host parse time, pinned memory
    and H2D bytes per rank stay those of one replica's batch however many replicas there are (round 2
    parsed and copied the whole global batch on every rank and sliced on the device)."""

    def __init__(self, records, vocab_table, num_channels, max_frames=-1, max_symbols=-1, time_multiple=1):
        self.rec, self.F = records, int(num_channels)
        self.max_frames, self.max_symbols = max_frames, max_symbols
        self.tm = max(1, int(time_multiple))
        # vocab_utils.create_vocab_table semantics on the tokens' utf-8 bytes, looked up in C (las_vocab_lookup): an open
        # addressing table over FNV-1a hashes, built here once (a Python dict lookup per token was 5 of the 7.5 ms a batch took)
        self.sos_id, self.eos_id = vocab_table.lookup([SOS])[0], vocab_table.lookup([EOS])[0]
        self.unk_id = vocab_table.lookup(['\x00 no such token \x00'])[0]
        size = 16
        while size < 4 * max(1, len(vocab_table)):
            size *= 2
        self.vkeys, self.vvals = np.zeros(size, np.uint64), np.zeros(size, np.int32)
        for tok, idx in vocab_table.items():
            h = _fnv1a(tok.encode('utf-8'))
            slot = h & (size - 1)
            while self.vkeys[slot] != 0 and int(self.vkeys[slot]) != h:
                slot = (slot + 1) & (size - 1)
            if self.vkeys[slot] == 0:                     # (setdefault semantics: the first index of a repeated token)
                self.vkeys[slot], self.vvals[slot] = h, idx

    def shapes(self, idx):
        """(T, U, T', decoder steps) of the batch with record indices `idx` -- no record is touched."""
        nfr, nlb = self.rec.n_frames[idx], self.rec.n_labels[idx]
        T = self.max_frames if self.max_frames > 0 else int(nfr.max())
        U = self.max_symbols if self.max_frames > 0 else int(nlb.max()) + 1
        Tp = (T + self.tm - 1) // self.tm * self.tm
        return T, U, Tp, min(int(nlb.max()) + 1, U)

    def parse(self, idx, U, pinned=False):
        """Packed frames [rows, F] fp32, row offsets [B+1], targets_inputs / targets_outputs [B, U] and target lengths [B] of
        the records `idx` (labels: [<s>] + y / y + [</s>], padded with the EOS id, utils/dataset_utils.py:226-264)."""
        lib, rec = hip.lib(), self.rec
        idx = np.asarray(idx, dtype=np.int64)
        B = len(idx)
        nfr, nlb = rec.n_frames[idx], rec.n_labels[idx]
        rows, ntok, nbytes = int(nfr.sum()), int(nlb.sum()), int(rec.label_bytes[idx].sum())
        frames = torch.empty(max(rows, 1), self.F, dtype=torch.float32, pin_memory=pinned)
        row_off = torch.empty(B + 1, dtype=torch.int64, pin_memory=pinned)
        lab = np.empty(max(nbytes, 1), np.uint8)
        tok_off = np.empty(ntok + 1, np.int32)
        counts = np.empty(B, np.int32)
        addr = np.ascontiguousarray(rec.addr[idx])
        lens = np.ascontiguousarray(rec.lengths[idx])
        hip.check(lib.las_tfrecord_parse_batch(0, addr.ctypes.data, lens.ctypes.data, B, self.F, frames.data_ptr(), rows,
                                               row_off.data_ptr(), lab.ctypes.data, nbytes, tok_off.ctypes.data, ntok,
                                               counts.ctypes.data))
        all_ids = np.empty(max(ntok, 1), np.int32)
        hip.check(lib.las_vocab_lookup(lab.ctypes.data, tok_off.ctypes.data, ntok, self.vkeys.ctypes.data, self.vvals.ctypes.data,
                                       len(self.vkeys), self.unk_id, all_ids.ctypes.data))
        tin = np.full((B, U), self.eos_id, np.int32)
        tout = np.full((B, U), self.eos_id, np.int32)
        tl = np.zeros(B, np.int32)
        t0 = 0
        for b in range(B):
            c = int(counts[b])
            ids = all_ids[t0:t0 + c]
            t0 += c
            n = min(c + 1, U)                    # quirk B4 of the slow path: the padded shape is [max_symbols]
            tin[b, 0] = self.sos_id
            tin[b, 1:n] = ids[:n - 1]
            tout[b, :min(c, n)] = ids[:n]
            if c < n:
                tout[b, c] = self.eos_id
            tl[b] = n
        return dict(frames=frames, row_off=row_off, tin=tin, tout=tout, tl=tl, rows=rows)


def shard_indices(idx, rank, world):
    """This rank's contiguous slice of a global index batch (dp.shard_batch's split, applied BEFORE anything is parsed)."""
    if len(idx) % world:
        raise ValueError('global batch %d is not divisible by %d replicas' % (len(idx), world))
    n = len(idx) // world
    return idx[rank * n:(rank + 1) * n]


class FastInput(object):
    """Iterator of (features, labels) dicts of CUDA tensors, produced ahead of the consumer by a background thread."""

    def __init__(self, records, vocab_table, num_channels, batch_size, num_epochs=1, is_infer=False, max_frames=-1,
                 max_symbols=-1, means=None, stds=None, seed=None, time_multiple=1, depth=3, device=None, take=0, shard=None):
        """batch_size: the GLOBAL batch.  shard = (rank, world): every rank walks the same index stream (same seed), derives
        the padded shapes from the whole global batch and parses / copies only its own 1/world of it."""
        self.rec, self.F, self.B = records, int(num_channels), int(batch_size)
        self.num_epochs, self.is_infer = num_epochs, is_infer
        self.max_frames, self.max_symbols = max_frames, max_symbols
        self.tm = max(1, int(time_multiple))
        self.Fp = (self.F + 7) // 8 * 8
        self.take = take
        self.shard = tuple(shard) if shard is not None and shard[1] > 1 else None
        self.dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        self.rng = random.Random(seed)
        self.host = HostBatcher(records, vocab_table, num_channels, max_frames, max_symbols, time_multiple)
        self.mean = self.std = None
        if means is not None and stds is not None:
            self.mean = torch.as_tensor(np.asarray(means, dtype=np.float64)).to(self.dev)
            self.std = torch.as_tensor(np.asarray(stds, dtype=np.float64)).to(self.dev)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.q = queue.Queue(maxsize=max(1, depth))
        self.err = None
        self.thread = threading.Thread(target=self._produce, daemon=True)
        self.thread.start()

    # -- index stream with the semantics (and the random draws) of dataset_utils.process_dataset ------------------------
    def _examples(self):
        n_frames, n_labels = self.rec.n_frames, self.rec.n_labels
        epoch = 0
        while self.num_epochs <= 0 or epoch < self.num_epochs:
            n = 0
            for i in range(len(self.rec)):
                if self.max_frames > 0 and not (n_frames[i] <= self.max_frames and n_labels[i] <= self.max_symbols):
                    continue
                n += 1
                yield i
            epoch += 1
            if n == 0:
                return

    def _index_batches(self):
        stream = self._examples()
        if not self.is_infer:
            stream = _shuffle(stream, self.B * 500, self.rng)
        batch = []
        for i in stream:
            batch.append(i)
            if len(batch) == self.B:
                yield batch
                batch = []
        if batch and self.is_infer:
            yield batch


    def _produce(self):
        try:
            torch.cuda.set_device(self.dev)
            lib = hip.lib()
            produced = 0
            for idx in self._index_batches():
                if self.take > 0 and produced >= self.take:
                    break
                idx = np.asarray(idx, dtype=np.int64)
                T, U, Tp, max_len = self.host.shapes(idx)          # from the whole global batch: all ranks agree
                if self.shard is not None:
                    idx = shard_indices(idx, *self.shard)          # ... and parse / copy only this rank's records
                B = len(idx)
                hb = self.host.parse(idx, U, pinned=True)
                frames, row_off = hb['frames'], hb['row_off']
                labels_host = torch.from_numpy(np.concatenate([hb['tin'].reshape(-1), hb['tout'].reshape(-1), hb['tl']])).pin_memory()
                with torch.cuda.stream(self.copy_stream):
                    d_frames = frames.to(self.dev, non_blocking=True)
                    d_off = row_off.to(self.dev, non_blocking=True)
                    d_lab = labels_host.to(self.dev, non_blocking=True)
                    x = torch.empty(B, Tp, self.Fp, dtype=torch.bfloat16, device=self.dev)
                    sl = torch.empty(B, dtype=torch.int32, device=self.dev)
                    hip.check(lib.las_normalize_pad_bf16(hip.p(d_frames), hip.p(d_off), hip.p(self.mean), hip.p(self.std), self.F,
                                                         hip.p(x), B, Tp, self.Fp, hip.p(sl), hip.stream()))
                    ev = torch.cuda.Event()
                    ev.record()
                n1 = B * U
                feats = {'encoder_inputs': x, 'source_sequence_length': sl}
                labels = {'targets_inputs': d_lab[:n1].view(B, U), 'targets_outputs': d_lab[n1:2 * n1].view(B, U),
                          'target_sequence_length': d_lab[2 * n1:]}
                # the pinned staging buffers stay referenced until the consumer has taken the batch (the copies read them)
                self.q.put((feats, labels, ev, max_len, (frames, row_off, labels_host, d_frames, d_off)))
                produced += 1
        except BaseException as e:          # surfaced in the consumer
            self.err = e
        finally:
            self.q.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is None:
            if self.err is not None:
                raise self.err
            raise StopIteration
        feats, labels, ev, max_len, keep = item
        torch.cuda.current_stream().wait_event(ev)
        for t in list(feats.values()) + list(labels.values()) + list(keep[3:]):
            t.record_stream(torch.cuda.current_stream())
        labels['max_target_length'] = max_len          # host int: train.py's num_steps without a device round trip
        return feats, labels


def fast_input_fn(dataset_filename, vocab_filename, norm_filename=None, num_channels=39, batch_size=8, num_epochs=1,
                  num_parallel_calls=32, max_frames=-1, max_symbols=-1, take=0, is_infer=False, seed=None, time_multiple=1,
                  verify_crc=True, depth=3, shard=None):
    """utils/dataset_utils.py:286-308 (same arguments, plus the listener's time multiple): an iterator of device batches.
    batch_size is the global batch; shard = (rank, world) makes this process parse and copy only its replica's part."""
    records = IndexedRecords(dataset_filename, verify_crc=verify_crc)
    vocab_table = create_vocab_table(vocab_filename)
    means = stds = None
    if norm_filename is not None and os.path.exists(norm_filename):
        means, stds = load_normalization(norm_filename)
    return FastInput(records, vocab_table, num_channels, batch_size, num_epochs, is_infer, max_frames, max_symbols, means, stds,
                     seed, time_multiple, depth, take=take, shard=shard)
