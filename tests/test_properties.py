"""Property tests of the host logic ("hypothesis on TFRecord framing/CRC, pyramid length arithmetic, edit-distance
trimming"; VERDICT r5 missing #5).  CPU only; the C parser is called through liblas_hip.so as the input path calls it.

  * TFRecord framing + SequenceExample (`preprocess_all.py:31-50` writes, `utils/dataset_utils.py:141-153` reads): what the
    Python writer writes, the C index / parser and the Python reader return unchanged, for any frame count (0 included), any
    feature width, any utf-8 tokens (empty included); a file cut at ANY byte is refused by both readers unless the cut is a record
    boundary; any flipped bit is refused when CRCs are checked.
  * pyramid lengths (`las/ops.py:49-65`: pad to even, len -> len // 2 + len % 2 per level): the oracle's recurrence equals
    ceil(len / 2^k), never exceeds the padded time axis the host computes (`Listener.pad_features`, `HostBatcher.shapes`).
  * edit distance (`utils/metrics_utils.py:8-41`): trimming (merge repeats, cut at the first EOS, drop -1) against a direct
    restatement, the distance against a brute-force recursion, metric properties."""
import functools
import os
import struct
import tempfile

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

SET = dict(deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])

tokens = st.text(alphabet=st.characters(blacklist_categories=('Cs',)), min_size=0, max_size=6)
examples = st.lists(st.tuples(st.integers(0, 6), st.lists(tokens, min_size=0, max_size=5)), min_size=0, max_size=5)


def _write(path, exs, F, seed):
    from phones_las_amd.utils import tfrecord as tfr
    rng = np.random.default_rng(seed)
    out, bounds, pos = [], [0], 0
    with tfr.TFRecordWriter(path) as w:
        for T, toks in exs:
            x = rng.standard_normal((T, F)).astype(np.float32)
            rec = tfr.make_example(x, toks)
            w.write(rec)
            pos += 12 + len(rec) + 4
            bounds.append(pos)
            out.append((x, toks))
    return out, bounds


def _c_index(data, verify=1):
    from phones_las_amd import hip
    lib = hip.lib()
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    n = lib.las_tfrecord_index(buf.ctypes.data, len(data), verify, 0, None, None, None, None, None)
    if n < 0:
        return n, None
    off, ln = np.empty(max(n, 1), np.int64), np.empty(max(n, 1), np.int64)
    nf, nl, lb = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int64)
    n2 = lib.las_tfrecord_index(buf.ctypes.data, len(data), verify, n, off.ctypes.data, ln.ctypes.data, nf.ctypes.data, nl.ctypes.data,
                                lb.ctypes.data)
    assert n2 == n
    return n, (buf, off[:n], ln[:n], nf[:n], nl[:n], lb[:n])


def _c_parse(idx, F):
    from phones_las_amd import hip
    lib = hip.lib()
    buf, off, ln, nf, nl, lb = idx
    n = len(off)
    rows, ntok, nb = int(nf.sum()), int(nl.sum()), int(lb.sum())
    frames = np.full((max(rows, 1), F), np.nan, np.float32)
    row_off = np.empty(n + 1, np.int64); tok = np.empty(ntok + 1, np.int32); cnt = np.empty(max(n, 1), np.int32)
    lab = np.empty(max(nb, 1), np.uint8)
    off, ln = np.ascontiguousarray(off), np.ascontiguousarray(ln)
    hip.check(lib.las_tfrecord_parse_batch(buf.ctypes.data, off.ctypes.data, ln.ctypes.data, n, F, frames.ctypes.data, rows,
                                           row_off.ctypes.data, lab.ctypes.data, nb, tok.ctypes.data, ntok, cnt.ctypes.data))
    blob, t0, out = lab.tobytes(), 0, []
    for k in range(n):
        toks = [blob[tok[t0 + j]:tok[t0 + j + 1]].decode('utf-8') for j in range(cnt[k])]
        out.append((frames[row_off[k]:row_off[k + 1]].copy(), toks))
        t0 += cnt[k]
    return out


@settings(max_examples=60, **SET)
@given(exs=examples, F=st.integers(1, 9), seed=st.integers(0, 2 ** 16))
def test_tfrecord_round_trip_python_writer_c_reader_python_reader(exs, F, seed):
    from phones_las_amd.utils import tfrecord as tfr
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'a.tfr')
        want, bounds = _write(path, exs, F, seed)
        data = open(path, 'rb').read()
        assert len(data) == bounds[-1]
        n, idx = _c_index(data, verify=1)
        assert n == len(want)
        if n:
            assert idx[3].tolist() == [x.shape[0] for x, _ in want] and idx[4].tolist() == [len(t) for _, t in want]
            assert idx[5].tolist() == [sum(len(t.encode('utf-8')) for t in toks) for _, toks in want]
            got = _c_parse(idx, F)
            for (x, toks), (gx, gt) in zip(want, got):
                assert np.array_equal(gx.reshape(-1, F), x.reshape(-1, F)) and gt == toks
        py = [tfr.parse_sequence_example(r, F) for r in tfr.tf_record_iterator(path, verify=True)]
        assert len(py) == len(want)
        for (x, toks), (px, pt) in zip(want, py):
            assert np.array_equal(np.asarray(px, np.float32).reshape(-1, F), x.reshape(-1, F))
            assert [t.decode('utf-8') if isinstance(t, bytes) else t for t in pt] == toks


@settings(max_examples=40, **SET)
@given(exs=examples.filter(lambda e: len(e) > 0), F=st.integers(1, 5), seed=st.integers(0, 2 ** 16), data=st.data())
def test_truncation_at_any_byte_and_any_flipped_bit_are_refused(exs, F, seed, data):
    from phones_las_amd.utils import tfrecord as tfr
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'a.tfr')
        _, bounds = _write(path, exs, F, seed)
        blob = open(path, 'rb').read()
        cut = data.draw(st.integers(0, len(blob) - 1))
        n, _ = _c_index(blob[:cut], verify=1)
        cpath = os.path.join(d, 'cut.tfr')
        open(cpath, 'wb').write(blob[:cut])
        if cut in bounds:
            assert n == bounds.index(cut)
            assert len(list(tfr.tf_record_iterator(cpath, verify=True))) == n
        else:
            assert n < 0
            with pytest.raises(IOError):
                list(tfr.tf_record_iterator(cpath, verify=True))
        at, bit = data.draw(st.integers(0, len(blob) - 1)), data.draw(st.integers(0, 7))
        bad = bytearray(blob)
        bad[at] ^= 1 << bit
        assert _c_index(bad, verify=1)[0] < 0
        open(cpath, 'wb').write(bad)
        with pytest.raises(IOError):
            list(tfr.tf_record_iterator(cpath, verify=True))
        _c_index(bad, verify=0)                       # without the check: any answer, no crash (ASan twin: test_host_sanitized.py)


def _crc32c_bitwise(b):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) bit by bit: independent of the library's table / SSE4.2 code."""
    c = 0xffffffff
    for x in b:
        c ^= x
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
    return c ^ 0xffffffff


@settings(max_examples=100, **SET)
@given(b=st.binary(min_size=0, max_size=300))
def test_crc32c_of_the_library_against_a_bitwise_restatement(b):
    from phones_las_amd import hip
    from phones_las_amd.utils import tfrecord as tfr
    crc = hip.lib().las_crc32c(bytes(b), len(b))
    assert crc == _crc32c_bitwise(b)
    assert tfr.masked_crc32c(b) == (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff       # TFRecord's mask


def test_crc32c_known_answer():
    from phones_las_amd import hip
    assert hip.lib().las_crc32c(b'123456789', 9) == 0xE3069283 == _crc32c_bitwise(b'123456789')    # RFC 3720 B.4 check value


# ---- pyramid length arithmetic --------------------------------------------------------------------------------------------
@settings(max_examples=200, **SET)
@given(lens=st.lists(st.integers(0, 5000), min_size=1, max_size=8), layers=st.integers(1, 6))
def test_pyramid_lengths_are_ceil_division_and_fit_the_padded_axis(lens, layers):
    import torch
    from oracle import las_oracle as O
    T = max(max(lens), 1)
    m = 2 ** (layers - 1)                                 # Listener.time_multiple / HostBatcher.tm
    Tp = (T + m - 1) // m * m                             # Listener.pad_features, HostBatcher.shapes
    assert Tp % m == 0 and 0 <= Tp - T < m
    length = torch.tensor(lens)
    x = torch.zeros(len(lens), T, 2, dtype=O.DT)
    t_axis = Tp
    for k in range(1, layers):                            # las/ops.py:83-85: stack after every layer but the first
        x, length = O.pyramidal_stack(x, length)
        t_axis //= 2                                      # the product's view: an exact halving of the padded axis
        assert length.tolist() == [-(-n // 2 ** k) for n in lens]
        assert x.shape[1] == -(-T // 2 ** k)              # the oracle pads one frame per odd level: ceil as well
        assert x.shape[1] <= t_axis and int(length.max()) <= t_axis
    assert t_axis * m == Tp


# ---- edit distance -----------------------------------------------------------------------------------------------------
ids = st.lists(st.integers(-1, 5), min_size=1, max_size=9)


def _trim_restated(row, eos):
    """utils/metrics_utils.py:8-26 restated literally: diff of [row, eos] marks the LAST element of every run; mask = positions
    before the first eos of [row, eos]; -1 dropped."""
    ext = list(row) + [eos]
    diff = [ext[i + 1] - ext[i] != 0 for i in range(len(row))]
    first = min(i for i, v in enumerate(ext) if v == eos)
    return [v for i, v in enumerate(row) if diff[i] and i < first and v != -1]


@functools.lru_cache(maxsize=None)
def _lev(a, b):
    if not a:
        return len(b)
    if not b:
        return len(a)
    return min(_lev(a[1:], b) + 1, _lev(a, b[1:]) + 1, _lev(a[1:], b[1:]) + (a[0] != b[0]))


@settings(max_examples=300, **SET)
@given(h=ids, t=ids, eos=st.integers(0, 5))
def test_edit_distance_trimming_and_distance(h, t, eos):
    from phones_las_amd.utils import metrics_utils as M
    n = max(len(h), len(t))
    h, t = h + [eos] * (n - len(h)), t + [eos] * (n - len(t))           # dense [B, U] rows are padded with EOS
    hs, ts = M.dense_to_sparse(h, eos), M.dense_to_sparse(t, eos)
    assert hs == _trim_restated(h, eos) and ts == _trim_restated(t, eos)
    assert eos not in hs and -1 not in hs
    if -1 not in h:                                    # (a -1 between two equal ids keeps both: the reference merges BEFORE it drops -1)
        assert all(a != b for a, b in zip(hs, hs[1:]))
    d = M.edit_distance([h], [t], eos)[0]
    if not ts:
        assert d == (float('inf') if hs else 0.0)
    else:
        assert d == _lev(tuple(hs), tuple(ts)) / len(ts)
        assert abs(len(hs) - len(ts)) / len(ts) <= d <= max(len(hs), len(ts)) / len(ts)
    assert M.edit_distance([t], [t], eos)[0] in (0.0,)                  # identity
    # what follows the first EOS never matters
    junk = h + [3, eos, 1]
    assert M.dense_to_sparse(junk[:len(h)] + [eos] + junk[len(h):], eos)[:len(hs)] == hs


@settings(max_examples=100, **SET)
@given(h=ids, t=ids, eos=st.integers(0, 5), perm=st.permutations(list(range(6))))
def test_edit_distance_mapping_is_applied_before_trimming(h, t, eos, perm):
    from phones_las_amd.utils import metrics_utils as M
    n = max(len(h), len(t))
    h = [max(v, 0) for v in h] + [eos] * (n - len(h))
    t = [max(v, 0) for v in t] + [eos] * (n - len(t))
    mapping = list(perm)
    got = M.edit_distance([h], [t], eos, mapping)[0]
    want = M.edit_distance([[mapping[v] for v in h]], [[mapping[v] for v in t]], eos)[0]
    assert got == want or (got != got and want != want)
