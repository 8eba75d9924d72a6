"""Normalised edit distance with the reference's pre-processing (utils/metrics_utils.py:8-41): optional id
mapping, keep the last element of every run of equal ids, cut at the first EOS, drop -1, then
Levenshtein / len(truth) (tf.edit_distance(normalize=True)).  Integer work on a few hundred ids per batch:
done on the host."""
__all__ = ['edit_distance', 'dense_to_sparse']


def _rows(t):
    if hasattr(t, 'detach'):
        return t.detach().cpu().tolist()
    return [list(r) for r in t]


def dense_to_sparse(row, eos_id):
    ext = list(row) + [eos_id]
    first_eos = ext.index(eos_id)
    return [v for i, v in enumerate(row) if ext[i + 1] != v and i < first_eos and v != -1]


def _levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        ai = a[i - 1]
        for j in range(1, len(b) + 1):
            sub = prev[j - 1] + (ai != b[j - 1])
            ins = cur[j - 1] + 1
            dele = prev[j] + 1
            cur[j] = sub if sub < ins and sub < dele else (ins if ins < dele else dele)
        prev = cur
    return prev[-1]


def edit_distance(hypothesis, truth, eos_id, mapping=None):
    out = []
    for h, t in zip(_rows(hypothesis), _rows(truth)):
        if mapping:
            h = [mapping[i] for i in h]
            t = [mapping[i] for i in t]
        hs, ts = dense_to_sparse(h, eos_id), dense_to_sparse(t, eos_id)
        d = _levenshtein(hs, ts)
        if not ts:
            out.append(float('inf') if hs else 0.0)
        else:
            out.append(d / len(ts))
    return out
