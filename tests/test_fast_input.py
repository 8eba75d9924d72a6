"""The C input path (liblas_hip.so: las_tfrecord_index / las_tfrecord_parse_batch, csrc/input.hip) against the pure-Python
TFRecord / SequenceExample code (utils/tfrecord.py) on the same files -- host functions, no GPU needed -- and, on the GPU,
utils/fast_input.FastInput (C parser + prefetch thread + las_normalize_pad_bf16) against utils.input_fn batch by batch."""
import os
import struct
import ctypes as C

import numpy as np
import pytest
import torch


def _corpus(tmp_path, n=7, F=5, seed=0, name='a.tfr', unpacked=False):
    from phones_las_amd.utils import tfrecord as tfr
    rng = np.random.default_rng(seed)
    path = str(tmp_path / name)
    ex = []
    with tfr.TFRecordWriter(path) as w:
        for i in range(n):
            T, U = int(rng.integers(1, 9)), int(rng.integers(0, 5))
            x = rng.standard_normal((T, F)).astype(np.float32)
            y = [['aa', 'b', 'sil', 'zh', 'æ'][int(k)] for k in rng.integers(0, 5, U)]
            rec = tfr.make_example(x, y)
            if unpacked and i % 2:         # the non-packed float encoding (wire type 5 per value) is legal protobuf too
                fl = b''.join(b'\x0d' + struct.pack('<f', v) for v in x[0])
                feat = tfr._ld(2, fl)
                rec_first = tfr._ld(1, feat)
                inp = rec_first + b''.join(tfr._ld(1, tfr._float_feature(f)) for f in x[1:])
                lab = b''.join(tfr._ld(1, tfr._bytes_feature(p.encode())) for p in y)
                entries = b''
                for key, fl_ in (('labels', lab), ('inputs', inp)):
                    entries += tfr._ld(1, tfr._ld(1, key.encode()) + tfr._ld(2, fl_))
                rec = tfr._ld(2, entries)
            w.write(rec)
            ex.append((x, y))
    return path, ex


def test_c_index_and_parse_match_the_python_parser(tmp_path):
    from phones_las_amd import hip
    from phones_las_amd.utils.fast_input import IndexedRecords
    path, ex = _corpus(tmp_path, n=9, unpacked=True)
    rec = IndexedRecords(path, verify_crc=True)
    assert len(rec) == 9
    assert rec.n_frames.tolist() == [x.shape[0] for x, _ in ex] and rec.n_labels.tolist() == [len(y) for _, y in ex]
    assert rec.label_bytes.tolist() == [sum(len(t.encode()) for t in y) for _, y in ex]
    lib = hip.lib()
    idx = np.array([4, 0, 8, 3], dtype=np.int64)
    rows, ntok, nb = int(rec.n_frames[idx].sum()), int(rec.n_labels[idx].sum()), int(rec.label_bytes[idx].sum())
    frames = np.full((rows, 5), np.nan, np.float32)
    off = np.empty(5, np.int64); tok = np.empty(ntok + 1, np.int32); cnt = np.empty(4, np.int32)
    lab = np.empty(max(nb, 1), np.uint8)
    addr, ln = np.ascontiguousarray(rec.addr[idx]), np.ascontiguousarray(rec.lengths[idx])
    hip.check(lib.las_tfrecord_parse_batch(0, addr.ctypes.data, ln.ctypes.data, 4, 5, frames.ctypes.data, rows, off.ctypes.data,
                                           lab.ctypes.data, nb, tok.ctypes.data, ntok, cnt.ctypes.data))
    blob, t0 = lab.tobytes(), 0
    for k, i in enumerate(idx):
        x, y = ex[i]
        assert np.array_equal(frames[off[k]:off[k + 1]], x)
        got = [blob[tok[t0 + j]:tok[t0 + j + 1]].decode() for j in range(cnt[k])]
        assert got == y
        t0 += cnt[k]
    # a frame of another width is an error (tf.parse_single_sequence_example raises as well)
    rc = lib.las_tfrecord_parse_batch(0, addr.ctypes.data, ln.ctypes.data, 4, 6, frames.ctypes.data, rows, off.ctypes.data,
                                      lab.ctypes.data, nb, tok.ctypes.data, ntok, cnt.ctypes.data)
    assert rc != 0 and b'num_channels' in lib.las_last_error()


def test_c_index_detects_corruption_and_truncation(tmp_path):
    from phones_las_amd import hip
    from phones_las_amd.utils.fast_input import IndexedRecords
    path, ex = _corpus(tmp_path, n=4)
    data = bytearray(open(path, 'rb').read())
    bad = str(tmp_path / 'bad.tfr')
    flipped = bytearray(data); flipped[40] ^= 0x10
    open(bad, 'wb').write(flipped)
    with pytest.raises(IOError):
        IndexedRecords(bad, verify_crc=True)
    open(bad, 'wb').write(data[:-3])
    with pytest.raises(IOError):
        IndexedRecords(bad, verify_crc=False)
    # hardware crc32c == the table implementation == the known answer of RFC 3720 ("123456789" -> 0xe3069283)
    assert hip.lib().las_crc32c(b'123456789', 9) == 0xe3069283
    # an empty file and a list of files
    open(str(tmp_path / 'empty.tfr'), 'wb').close()
    p2, ex2 = _corpus(tmp_path, n=3, seed=5, name='b.tfr')
    lst = str(tmp_path / 'files.txt')
    open(lst, 'w').write('%s\n%s\n%s\n' % (path, str(tmp_path / 'empty.tfr'), p2))
    rec = IndexedRecords(lst)
    assert len(rec) == 7 and rec.n_frames.tolist() == [x.shape[0] for x, _ in ex + ex2]


def test_c_vocab_lookup_matches_the_vocab_table(tmp_path):
    """las_vocab_lookup (FNV-1a open-addressing table built by fast_input) against vocab_utils.create_vocab_table: known
    tokens (multi-byte utf-8 among them), an unknown token -> <unk>, the empty token, a token repeated in the vocab file
    (first index wins)."""
    from phones_las_amd import hip
    from phones_las_amd.utils import vocab_utils
    from phones_las_amd.utils.fast_input import _fnv1a
    vocab = str(tmp_path / 'vocab.txt')
    words = ['aa', 'b', 'sil', 'zh', 'æ', 'b', 'ʃ'] + ['w%d' % i for i in range(300)]
    open(vocab, 'w', encoding='utf-8').write('\n'.join(words) + '\n')
    table = vocab_utils.create_vocab_table(vocab)
    size = 16
    while size < 4 * len(table):
        size *= 2
    keys, vals = np.zeros(size, np.uint64), np.zeros(size, np.int32)
    for tok, idx in table.items():
        h = _fnv1a(tok.encode('utf-8'))
        slot = h & (size - 1)
        while keys[slot] != 0 and int(keys[slot]) != h:
            slot = (slot + 1) & (size - 1)
        if keys[slot] == 0:
            keys[slot], vals[slot] = h, idx
    toks = ['zh', 'æ', 'nope', '', 'b', 'w299', 'ʃ', '</s>', 'w0', 'aa ']
    blob = b''.join(t.encode('utf-8') for t in toks)
    offs = np.concatenate([[0], np.cumsum([len(t.encode('utf-8')) for t in toks])]).astype(np.int32)
    lab = np.frombuffer(blob, np.uint8).copy()
    out = np.full(len(toks), -7, np.int32)
    hip.check(hip.lib().las_vocab_lookup(lab.ctypes.data, offs.ctypes.data, len(toks), keys.ctypes.data, vals.ctypes.data, size,
                                         vocab_utils.UNK_ID, out.ctypes.data))
    assert out.tolist() == table.lookup(toks)
    assert out[2] == vocab_utils.UNK_ID and out[3] == vocab_utils.UNK_ID and out[4] == 3 + 1      # 'b': its first line
    assert hip.lib().las_vocab_lookup(lab.ctypes.data, offs.ctypes.data, 1, keys.ctypes.data, vals.ctypes.data, size - 1, 0, out.ctypes.data) != 0


@pytest.mark.gpu
@pytest.mark.parametrize('infer,max_frames', [(False, -1), (True, -1), (False, 6)])
def test_fast_input_batches_equal_the_python_pipeline(tmp_path, infer, max_frames):
    """Same utterances, same order, same padded shapes and -- after the device-side (x - mean) / std -> bf16 -- bit-identical
    features as utils.input_fn + Listener.pad_features, for the same seed (shuffle buffer, repeat, drop_remainder, filters)."""
    import joblib
    from phones_las_amd import utils
    from phones_las_amd.utils.fast_input import fast_input_fn
    path, ex = _corpus(tmp_path, n=23, F=5, seed=3)
    vocab = str(tmp_path / 'vocab.txt')
    open(vocab, 'w', encoding='utf-8').write('aa\nb\nsil\næ\n')          # 'zh' is out of vocabulary -> <unk>
    norm = str(tmp_path / 'norm.dmp')
    joblib.dump([np.linspace(-0.5, 0.5, 5), np.linspace(0.5, 2.0, 5)], norm)
    kw = dict(num_channels=5, batch_size=4, num_epochs=2, max_frames=max_frames, max_symbols=(4 if max_frames > 0 else -1),
              is_infer=infer, seed=11)
    slow = list(utils.input_fn(path, vocab, norm, **kw))
    fast = list(fast_input_fn(path, vocab, norm, time_multiple=4, **kw))
    assert len(slow) == len(fast) and len(slow) >= 5
    for (fs, ls), (ff, lf) in zip(slow, fast):
        assert lf.pop('max_target_length') == int(ls['target_sequence_length'].max())
        for k in ls:
            assert np.array_equal(ls[k], lf[k].cpu().numpy()), k
        assert np.array_equal(fs['source_sequence_length'], ff['source_sequence_length'].cpu().numpy())
        x = torch.from_numpy(fs['encoder_inputs'])
        B, T, F = x.shape
        got = ff['encoder_inputs'].cpu()
        assert got.dtype == torch.bfloat16 and got.shape[0] == B and got.shape[1] % 4 == 0 and got.shape[2] == 8
        assert torch.equal(got[:, :T, :F], x.to(torch.bfloat16))
        assert float(got[:, T:].float().abs().max()) == 0.0 if got.shape[1] > T else True
        assert float(got[:, :, F:].float().abs().max()) == 0.0


@pytest.mark.gpu
def test_train_loop_on_tfrecords_sustains_the_resident_batch_rate(tmp_path):
    """train.py's step loop fed from TFRecords through the C parser + prefetch thread against the same steps
    on one resident batch (what bench.py times), at T=800 / F=40 / B=64 with a reduced model: the record-fed loop must reach
    90 % of the resident-batch rate (measured: see DESIGN.md section 6)."""
    import time
    from phones_las_amd import model_helper as mh, utils
    from phones_las_amd.utils import tfrecord as tfr, params_utils as pu
    from phones_las_amd.utils.fast_input import fast_input_fn
    rng = np.random.default_rng(0)
    B, T, F, U, N = 64, 800, 40, 40, 256
    path = str(tmp_path / 'train.tfr')
    vocab = str(tmp_path / 'vocab.txt')
    toks = ['p%d' % i for i in range(61)]
    open(vocab, 'w').write('\n'.join(toks) + '\n')
    with tfr.TFRecordWriter(path) as w:
        x = rng.standard_normal((T, F)).astype(np.float32)
        for i in range(N):
            y = [toks[int(k)] for k in rng.integers(0, 61, U - 1)]
            w.write(tfr.make_example(np.roll(x, i, 0), y))
    hp = pu.get_default_hparams()
    for k, v in dict(num_channels=F, encoder_layers=2, encoder_units=128, use_pyramidal=True, decoder_layers=1, decoder_units=128,
                     target_vocab_size=64, attention_type='luong', bottom_only=True, pass_hidden_state=True, dropout=0.0,
                     sampling_probability=0.0).items():
        hp.set_hparam(k, v)
    model = mh.LasModel(pu.get_encoder_decoder_hparams(hp))

    def run(batches, n):
        it = iter(batches)
        for _ in range(3):
            f, l = next(it)
            model.train_step(f, l, num_steps=l.pop('max_target_length'))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            f, l = next(it)
            model.train_step(f, l, num_steps=l.pop('max_target_length'))
        torch.cuda.synchronize()
        return n * B / (time.perf_counter() - t0)

    fed = fast_input_fn(path, vocab, None, num_channels=F, batch_size=B, num_epochs=40, seed=1, time_multiple=model.listener.time_multiple)
    f0, l0 = next(iter(fast_input_fn(path, vocab, None, num_channels=F, batch_size=B, num_epochs=1, seed=1,
                                     time_multiple=model.listener.time_multiple)))
    l0.pop('max_target_length')

    def resident():
        while True:
            yield f0, dict(l0, max_target_length=U)
    r_res = run(resident(), 40)
    r_fed = run(fed, 40)
    r_res2 = run(resident(), 40)
    model.check_device_status()
    print('resident batch %.0f / %.0f utt/s, TFRecord-fed %.0f utt/s (%.1f %%)' % (r_res, r_res2, r_fed, 100 * r_fed / max(r_res, r_res2)))
    assert r_fed >= 0.9 * min(r_res, r_res2), (r_fed, r_res, r_res2)


def _big_corpus(tmp_path, n=96, F=40, seed=5):
    from phones_las_amd.utils import tfrecord as tfr
    rng = np.random.default_rng(seed)
    path = str(tmp_path / 'big.tfr')
    with tfr.TFRecordWriter(path) as w:
        for i in range(n):
            T, U = int(rng.integers(150, 400)), int(rng.integers(3, 30))
            w.write(tfr.make_example(rng.standard_normal((T, F)).astype(np.float32), ['w%d' % int(k) for k in rng.integers(0, 50, U)]))
    vocab = str(tmp_path / 'vocab.txt')
    open(vocab, 'w').write('\n'.join('w%d' % i for i in range(50)) + '\n')
    return path, vocab


def _shard_worker(rank, world, port, path, vocab, ret):
    """One rank of a data-parallel job on the HOST side of the input path (no GPU): walks the same index stream as every
    other rank, takes the padded shapes from the whole global batch and parses only its slice."""
    import random
    import time
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from phones_las_amd.utils import vocab_utils
    from phones_las_amd.utils.dataset_utils import _shuffle
    from phones_las_amd.utils.fast_input import IndexedRecords, HostBatcher, shard_indices
    rec = IndexedRecords(path)
    hb = HostBatcher(rec, vocab_utils.create_vocab_table(vocab), 40, time_multiple=4)
    rng = random.Random(1234)                                   # train.py's seed: the same stream on every rank
    order = list(_shuffle(iter(range(len(rec))), 32 * 500, rng))
    out = []
    t_shard = t_global = 0.0
    for k in range(0, len(order) - 31, 32):
        gidx = np.asarray(order[k:k + 32], dtype=np.int64)
        T, U, Tp, steps = hb.shapes(gidx)
        mine = shard_indices(gidx, rank, world)
        t0 = time.perf_counter()
        part = hb.parse(mine, U)
        t_shard += time.perf_counter() - t0
        if rank == 0:                                           # what round 2 did on EVERY rank: the whole global batch
            t0 = time.perf_counter()
            whole = hb.parse(gidx, U)
            t_global += time.perf_counter() - t0
        else:
            whole = None
        out.append((gidx.tolist(), (T, U, Tp, steps), part, whole))
    ret[rank] = (out, t_shard, t_global)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_host_batches_tile_the_global_batch_gloo(tmp_path):
    """world_size 2 on CPU: the ranks agree on the index stream and the padded shapes, the UNION of their
    parsed shards is the batch one rank would have parsed -- frames, row offsets, targets, lengths -- and a rank touches
    half the records: half the rows (deterministic) and clearly less parse time than the global batch (measured ~0.5x)."""
    import socket
    import torch.multiprocessing as mp
    path, vocab = _big_corpus(tmp_path)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_shard_worker, args=(2, port, path, vocab, ret), nprocs=2, join=True)
    (o0, ts0, tg0), (o1, ts1, _) = ret[0], ret[1]
    assert len(o0) == len(o1) == 3
    for (g0, shp0, p0, whole), (g1, shp1, p1, _) in zip(o0, o1):
        assert g0 == g1 and shp0 == shp1                                            # same stream, same padded shapes
        rows0, rows1 = p0['rows'], p1['rows']
        assert rows0 + rows1 == whole['rows'] and 0.3 < rows0 / whole['rows'] < 0.7
        assert torch.equal(torch.cat([p0['frames'][:rows0], p1['frames'][:rows1]]), whole['frames'][:whole['rows']])
        assert torch.equal(torch.cat([p0['row_off'], p1['row_off'][1:] + rows0]), whole['row_off'])
        for k in ('tin', 'tout', 'tl'):
            assert np.array_equal(np.concatenate([p0[k], p1[k]]), whole[k]), k
        assert p0['tin'].shape == (16, shp0[1]) and shp0[2] % 4 == 0 and shp0[3] == int(whole['tl'].max())
    assert ts0 < 0.9 * tg0, (ts0, tg0)             # (half the rows; measured 0.5x -- the bound leaves room for a loaded host)


@pytest.mark.gpu
def test_fast_input_shards_are_the_slices_of_the_unsharded_batches(tmp_path):
    """FastInput(shard=(r, 2)) for r = 0, 1 against the unsharded iterator: the same steps, and each rank's device batch is
    bit-identical to its slice of the global one (features after normalise + bf16 + pad, labels, lengths, decoder steps)."""
    from phones_las_amd.utils.fast_input import fast_input_fn
    path, vocab = _big_corpus(tmp_path, n=40)
    kw = dict(num_channels=40, batch_size=8, num_epochs=2, seed=1234, time_multiple=4)
    whole = list(fast_input_fn(path, vocab, None, **kw))
    parts = [list(fast_input_fn(path, vocab, None, shard=(r, 2), **kw)) for r in range(2)]
    assert len(whole) == len(parts[0]) == len(parts[1]) == 10
    for (fw, lw), (f0, l0), (f1, l1) in zip(whole, *parts):
        assert lw.pop('max_target_length') == l0.pop('max_target_length') == l1.pop('max_target_length')
        for k in fw:
            assert torch.equal(torch.cat([f0[k], f1[k]]), fw[k]), k
        for k in lw:
            assert torch.equal(torch.cat([l0[k], l1[k]]), lw[k]), k
