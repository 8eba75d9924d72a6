"""TFRecord files holding tf.train.SequenceExample records, byte-compatible with what the reference writes
(preprocess_all.py:31-50,164-167) and reads (utils/dataset_utils.py:141-158), without TensorFlow or protoc.

Framing: uint64 length | uint32 masked_crc32c(length) | payload | uint32 masked_crc32c(payload),
little-endian, masked = rotr15(crc) + 0xa282ead8.  Payload: SequenceExample{ feature_lists(2): FeatureLists{
feature_list(1): map<string, FeatureList{ feature(1): Feature{ bytes_list(1) | float_list(2) | int64_list(3) } }> } }.
"""
import struct

import numpy as np

from .. import hip

__all__ = ['TFRecordWriter', 'tf_record_iterator', 'make_example', 'parse_sequence_example', 'masked_crc32c']

_MASK_DELTA = 0xa282ead8


def masked_crc32c(data):
    crc = hip.lib().las_crc32c(bytes(data), len(data))
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


# ---- protobuf wire helpers ---------------------------------------------------------------------
def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):                      # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _read_varint(buf, pos):
    res, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        res |= (b & 0x7f) << shift
        if not b & 0x80:
            return res, pos
        shift += 7


def _fields(buf):
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 2:
            ln, pos = _read_varint(buf, pos)
            yield field, wt, buf[pos:pos + ln]
            pos += ln
        elif wt == 0:
            v, pos = _read_varint(buf, pos)
            yield field, wt, v
        elif wt == 5:
            yield field, wt, buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            yield field, wt, buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)


# ---- SequenceExample ------------------------------------------------------------------------------
def _float_feature(values):
    packed = np.asarray(values, dtype='<f4').tobytes()
    return _ld(2, _ld(1, packed))             # Feature.float_list(2) { value(1) packed }


def _bytes_feature(value):
    return _ld(1, _ld(1, value))              # Feature.bytes_list(1) { value(1) }


def make_example(inputs, labels):
    """preprocess_all.py:31-50: 'inputs' = one float_list of F values per frame; 'labels' = one bytes_list token per
    step (list of str) or one float_list per step (binary features).  Returns the serialized SequenceExample."""
    if len(labels) and isinstance(labels[0], str):
        lab = b''.join(_ld(1, _bytes_feature(p.encode())) for p in labels)
    else:
        lab = b''.join(_ld(1, _float_feature(f)) for f in labels)
    inp = b''.join(_ld(1, _float_feature(f)) for f in inputs)
    entries = b''
    for key, fl in (('labels', lab), ('inputs', inp)):
        entries += _ld(1, _ld(1, key.encode()) + _ld(2, fl))          # map entry {key(1), value(2)=FeatureList}
    return _ld(2, entries)                                              # SequenceExample.feature_lists(2)


def _parse_feature(buf):
    for field, wt, val in _fields(buf):
        if field == 1:                        # bytes_list
            return 'bytes', [v for f, _, v in _fields(val) if f == 1]
        if field == 2:                        # float_list (packed or not)
            out = []
            for f, w, v in _fields(val):
                if f == 1 and w == 2:
                    out.append(np.frombuffer(v, dtype='<f4'))
                elif f == 1 and w == 5:
                    out.append(np.frombuffer(v, dtype='<f4'))
            return 'float', (np.concatenate(out) if out else np.zeros(0, np.float32))
        if field == 3:
            vals = []
            for f, w, v in _fields(val):
                if f == 1 and w == 2:
                    p = 0
                    while p < len(v):
                        x, p = _read_varint(v, p)
                        vals.append(x)
                elif f == 1:
                    vals.append(v)
            return 'int64', np.asarray(vals, dtype=np.int64)
    return 'empty', None


def parse_sequence_example(serialized, num_channels=None):
    """tf.parse_single_sequence_example with {'inputs': FixedLenSequenceFeature([num_channels], f32), 'labels':
    FixedLenSequenceFeature([], string)} (utils/dataset_utils.py:141-153).  Returns (inputs [T,F] float32, labels:
    list of str, or [U,nf] float32 for binary-feature targets)."""
    lists = {}
    for field, _, val in _fields(memoryview(serialized).tobytes() if not isinstance(serialized, bytes) else serialized):
        if field != 2:
            continue
        for f2, _, entry in _fields(val):
            if f2 != 1:
                continue
            key, fl = None, b''
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = bytes(v).decode()
                elif f3 == 2:
                    fl = v
            lists[key] = [_parse_feature(feat) for f4, _, feat in _fields(fl) if f4 == 1]
    if 'inputs' not in lists or 'labels' not in lists:
        raise ValueError('SequenceExample lacks the inputs/labels feature lists')
    frames = [v for _, v in lists['inputs']]
    if num_channels is not None:
        for fr in frames:
            if fr.shape[0] != num_channels:
                raise ValueError('inputs frame has %d values, expected num_channels=%d' % (fr.shape[0], num_channels))
    inputs = np.stack(frames).astype(np.float32) if frames else np.zeros((0, num_channels or 0), np.float32)
    kinds = {k for k, _ in lists['labels']}
    if kinds <= {'bytes'}:
        labels = [bytes(v[0]).decode() for _, v in lists['labels']]
    else:
        labels = np.stack([v for _, v in lists['labels']]).astype(np.float32)
    return inputs, labels


# ---- record framing -------------------------------------------------------------------------------
class TFRecordWriter(object):
    def __init__(self, path):
        self._f = open(path, 'wb')

    def write(self, record):
        ln = struct.pack('<Q', len(record))
        self._f.write(ln)
        self._f.write(struct.pack('<I', masked_crc32c(ln)))
        self._f.write(record)
        self._f.write(struct.pack('<I', masked_crc32c(record)))

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def tf_record_iterator(path, verify=False):
    """Yield the payload of every record of a TFRecord file; ``verify`` checks both masked CRCs."""
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError('truncated TFRecord header in %s' % path)
            (ln,), (crc_len,) = struct.unpack('<Q', head[:8]), struct.unpack('<I', head[8:])
            if verify and masked_crc32c(head[:8]) != crc_len:
                raise IOError('corrupt TFRecord length crc in %s' % path)
            data = f.read(ln)
            tail = f.read(4)
            if len(data) < ln or len(tail) < 4:
                raise IOError('truncated TFRecord payload in %s' % path)
            if verify and masked_crc32c(data) != struct.unpack('<I', tail)[0]:
                raise IOError('corrupt TFRecord payload crc in %s' % path)
            yield data
