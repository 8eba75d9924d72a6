"""Vocabulary files with the reference's conventions (utils/vocab_utils.py:16-41): ``<unk>``=0, ``<s>``=1,
``</s>``=2, then one token per line of vocab.txt (or a pickled list)."""
import pickle

__all__ = ['create_vocab_table', 'load_vocab', 'UNK', 'SOS', 'EOS', 'UNK_ID', 'SOS_ID', 'EOS_ID']

UNK, SOS, EOS = '<unk>', '<s>', '</s>'
UNK_ID, SOS_ID, EOS_ID = 0, 1, 2


def load_vocab(filename):
    if '.pickle' not in filename:
        with open(filename, 'r', encoding='utf-8') as f:
            vocab_list = [v.strip('\r\n') for v in f]
    else:
        with open(filename, 'rb') as f:
            vocab_list = list(pickle.load(f))
    return [UNK, SOS, EOS] + vocab_list


class VocabTable(dict):
    """index_table_from_tensor(vocab, num_oov_buckets=0, default_value=UNK_ID)."""

    def lookup(self, tokens):
        return [self.get(t, UNK_ID) for t in tokens]


def create_vocab_table(filename):
    table = VocabTable()
    for i, tok in enumerate(load_vocab(filename)):
        table.setdefault(tok, i)
    return table
