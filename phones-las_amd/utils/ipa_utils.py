"""The data-only part of the reference's utils/ipa_utils.py that the model path consumes:
``load_binf2phone`` (utils/ipa_utils.py:313-328) and ``get_mapping`` (utils/ipa_utils.py:289-310).
Text-to-IPA (espeak-ng) is data preparation and out of scope."""
import csv

import numpy as np

from .vocab_utils import UNK, SOS, EOS

__all__ = ['load_binf2phone', 'get_mapping', 'BinfTable']


class BinfTable(object):
    """features x phones 0/1 table with the DataFrame attributes the callers use (.values/.index/.columns/.shape)."""

    def __init__(self, index, columns, values):
        self.index, self.columns = list(index), list(columns)
        self.values = np.asarray(values, dtype=np.float32)
        self.shape = self.values.shape

    def __getitem__(self, cols):
        if isinstance(cols, str):
            return self.values[:, self.columns.index(cols)]
        idx = [self.columns.index(c) for c in cols]
        return BinfTable(self.index, [self.columns[i] for i in idx], self.values[:, idx])


def load_binf2phone(filename, vocab_list=None):
    """CSV (header = phones, first column = feature names) -> table with <unk>/<s>/</s> columns inserted at
    0/1/2 and <s>/</s> rows appended; <unk> column all ones, <s>/</s> one-hot on their own rows."""
    with open(filename, 'r', encoding='utf-8', newline='') as f:
        rows = list(csv.reader(f))
    phones = rows[0][1:]
    feats = [r[0] for r in rows[1:] if r]
    body = np.array([[float(x) for x in r[1:]] for r in rows[1:] if r], dtype=np.float32)
    nf = body.shape[0]
    cols = [UNK, SOS, EOS] + phones
    mat = np.zeros((nf + 2, len(cols)), dtype=np.float32)
    mat[:nf, 3:] = body
    mat[:nf, 0] = 1.0            # <unk> column
    mat[nf, 1] = 1.0             # <s> row / column
    mat[nf + 1, 2] = 1.0         # </s>
    mat[nf:, 0] = 1.0
    t = BinfTable(feats + [SOS, EOS], cols, mat)
    if vocab_list is not None:
        t = t[vocab_list]
    return t


def get_mapping(mapping_path, vocab_path):
    """TIMIT folding (e.g. misc/phones.60-48-39.map + misc/timit-61.txt) -> (new vocab with specials, int map;
    -1 = deleted phone).  'sil' is renamed 'h#' as in the reference."""
    with open(mapping_path, 'r') as f:
        mapping_lines = f.read().strip().replace('sil', 'h#').split('\n')
    with open(vocab_path, 'r') as f:
        vocab = f.read().strip().split('\n')
    mapping, new_vocab = {}, set()
    for line in mapping_lines:
        ph = line.split('\t')
        if len(ph) < 3:
            mapping[ph[0]] = None
        else:
            mapping[ph[0]] = ph[-1]
            new_vocab.add(ph[-1])
    new_vocab = [UNK, SOS, EOS] + sorted(new_vocab)
    int_mapping = [0, 1, 2]
    for p in vocab:
        int_mapping.append(new_vocab.index(mapping[p]) if mapping[p] is not None else -1)
    return new_vocab, int_mapping
