"""File formats the path consumes, byte-compatible with the reference's writers/readers:
TFRecord framing + SequenceExample, vocab.txt, norm.dmp, and the process_dataset batching semantics."""
import os
import struct

import numpy as np
import pytest


def test_crc32c_known_answers():
    from phones_las_amd import hip
    lib = hip.lib()
    assert lib.las_crc32c(b'123456789', 9) == 0xE3069283            # the standard CRC-32C check value
    assert lib.las_crc32c(b'', 0) == 0
    data = bytes(range(256)) * 5
    ref = 0xffffffff
    for b in data:                                                   # bitwise reference
        ref ^= b
        for _ in range(8):
            ref = (ref >> 1) ^ 0x82F63B78 if ref & 1 else ref >> 1
    assert lib.las_crc32c(data, len(data)) == ref ^ 0xffffffff


def test_sequence_example_known_bytes():
    # hand-assembled protobuf: feature_lists{ feature_list{key:"labels" value{feature{bytes_list{value:"ab"}}}}
    #                                        feature_list{key:"inputs" value{feature{float_list{value:[1.0,2.0]}}}} }
    from phones_las_amd.utils import tfrecord
    ex = tfrecord.make_example(np.array([[1.0, 2.0]], np.float32), ['ab'])
    labels = b'\x0a\x06labels\x12\x08\x0a\x06\x0a\x04\x0a\x02ab'
    floats = struct.pack('<ff', 1.0, 2.0)
    inputs = b'\x0a\x06inputs\x12\x0e\x0a\x0c\x12\x0a\x0a\x08' + floats
    want = b'\x12' + bytes([2 + len(labels) + 2 + len(inputs)]) + b'\x0a' + bytes([len(labels)]) + labels + \
        b'\x0a' + bytes([len(inputs)]) + inputs
    assert ex == want
    x, y = tfrecord.parse_sequence_example(ex, 2)
    assert x.tolist() == [[1.0, 2.0]] and y == ['ab']
    with pytest.raises(ValueError):
        tfrecord.parse_sequence_example(ex, 3)                       # wrong num_channels


def test_tfrecord_roundtrip_and_framing(tmp_path):
    from phones_las_amd.utils import tfrecord
    rng = np.random.default_rng(0)
    path = str(tmp_path / 'a.tfr')
    exs = []
    with tfrecord.TFRecordWriter(path) as w:
        for n in (5, 1, 17):
            x = rng.standard_normal((n, 13)).astype(np.float32)
            y = ['p%d' % i for i in range(n % 4 + 1)]
            exs.append((x, y))
            w.write(tfrecord.make_example(x, y))
    raw = open(path, 'rb').read()
    ln = struct.unpack('<Q', raw[:8])[0]
    assert struct.unpack('<I', raw[8:12])[0] == tfrecord.masked_crc32c(raw[:8])
    assert struct.unpack('<I', raw[12 + ln:16 + ln])[0] == tfrecord.masked_crc32c(raw[12:12 + ln])
    got = [tfrecord.parse_sequence_example(r, 13) for r in tfrecord.tf_record_iterator(path, verify=True)]
    assert len(got) == 3
    for (x, y), (gx, gy) in zip(exs, got):
        assert np.array_equal(x, gx) and y == gy
    bad = bytearray(raw)
    bad[20] ^= 1
    open(path, 'wb').write(bytes(bad))
    with pytest.raises(IOError):
        list(tfrecord.tf_record_iterator(path, verify=True))
    # binary-feature targets: float_list per step
    ex = tfrecord.make_example(np.zeros((2, 3), np.float32), np.array([[0, 1], [1, 1]], np.float32))
    x, y = tfrecord.parse_sequence_example(ex, 3)
    assert y.tolist() == [[0, 1], [1, 1]]


def test_input_fn_batches_like_the_reference(tmp_path):
    from phones_las_amd.utils import tfrecord, input_fn
    from phones_las_amd.utils.features_utils import save_normalization
    rng = np.random.default_rng(1)
    path = str(tmp_path / 'train.tfr')
    vocab = str(tmp_path / 'vocab.txt')
    norm = str(tmp_path / 'norm.dmp')
    open(vocab, 'w').write('a\nb\nc\n')
    means, stds = np.arange(4, dtype=np.float32), np.full(4, 2.0, np.float32)
    save_normalization(norm, means, stds)
    data = []
    with tfrecord.TFRecordWriter(path) as w:
        for i in range(5):
            x = rng.standard_normal((3 + i, 4)).astype(np.float32)
            y = ['a', 'zz', 'c'][:1 + i % 3]
            data.append((x, y))
            w.write(tfrecord.make_example(x, y))
    batches = list(input_fn(path, vocab, norm, num_channels=4, batch_size=2, num_epochs=1, is_infer=True))
    assert len(batches) == 3                                         # remainder kept when inferring
    f, l = batches[0]
    assert f['encoder_inputs'].shape == (2, 4, 4) and f['source_sequence_length'].tolist() == [3, 4]
    assert np.allclose(f['encoder_inputs'][0, :3], (data[0][0] - means) / stds)
    assert float(np.abs(f['encoder_inputs'][0, 3:]).max()) == 0.0    # padded with 0.0
    assert l['targets_inputs'].tolist() == [[1, 3, 2], [1, 3, 0]]    # <s> a </s>(pad) ; <s> a <unk>
    assert l['targets_outputs'].tolist() == [[3, 2, 2], [3, 0, 2]]
    assert l['target_sequence_length'].tolist() == [2, 3]
    # training: drop_remainder, shuffled, repeated
    tr = list(input_fn(path, vocab, norm, num_channels=4, batch_size=2, num_epochs=2, seed=0))
    assert len(tr) == 5                                              # 10 examples -> 5 full batches
    # max_frames filter + static padding (quirk B4)
    mf = list(input_fn(path, vocab, norm, num_channels=4, batch_size=2, num_epochs=1, max_frames=5, max_symbols=2,
                       is_infer=True))
    assert all(f['encoder_inputs'].shape == (len(f['source_sequence_length']), 5, 4) for f, _ in mf)
    assert all(l['targets_inputs'].shape[1] == 2 for _, l in mf)
    # list-of-files variant
    lst = str(tmp_path / 'files.txt')
    open(lst, 'w').write(path + '\n' + path + '\n')
    assert len(list(input_fn(lst, vocab, None, num_channels=4, batch_size=5, num_epochs=1, is_infer=True))) == 2
